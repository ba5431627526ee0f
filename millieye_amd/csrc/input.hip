// Input producer on the GPU: what module3_our_dataset/utils/datasets.py does per frame on
// the CPU between the decoded jpg / radar pickle and the batch tensors Network.forward takes.
//   me_image_pad_resize_u8_f32  ToTensor (u8 HWC -> f32 CHW / 255, datasets.py:203) + pad_to_square (:16-27, :211) +
//                               resize(img, S) = F.interpolate(nearest) (:30-32, :317), one pass, one write.
//   me_radar_heatmap_f32        plot_radar_heatmap (:59-106: three np.histogram2d in float64, per-bin means, range
//                               normalisation) + ToTensor().float() (:267) + pad_to_square (:270) +
//                               F.interpolate(bilinear, align_corners=True) to the map size (:318-321).
// Both are HBM/latency-trivial (a 1600x900 frame is 4.3 MB in, 2 MB out); the point is that the batch never
// exists on the host and the 3 histograms + resize per frame leave the DataLoader's critical path.
#include "common.h"

// every product / sum below is a separately rounded IEEE operation in the reference (numpy float64, aten float32
// scalar kernels): no fused multiply-add contraction
#pragma clang fp contract(off)

namespace {

// ---- image: u8 HWC -> padded square -> nearest resize -> f32 CHW -------------------------------------------
// flip: the padded square is mirrored left-right before the resize (module2_mixed/utils/datasets.py:143-146,
// utils/augmentations.py: horisontal_flip runs on the padded tensor, the resize in collate_fn afterwards)
__global__ __launch_bounds__(256) void image_pad_resize_kernel(const unsigned char* __restrict__ src, int h, int w,
                                                               float* __restrict__ dst, int S, int flip) {
  const int P = h > w ? h : w;                 // padded side
  const int diff = h > w ? h - w : w - h;
  const int pad1 = diff / 2;                   // upper / left padding (datasets.py:20)
  const int pad_top = h <= w ? pad1 : 0, pad_left = h <= w ? 0 : pad1;
  // torch nearest (aten UpSampleKernel nearest_idx): src = min(int(floorf(dst * scale)), in - 1),
  // scale = float(in) / out; identity when in == out
  const float scale = (float)P / (float)S;
  const int total = S * S;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int y = idx / S, x = idx - y * S;
    int py = P == S ? y : (int)floorf((float)y * scale);
    int px = P == S ? x : (int)floorf((float)x * scale);
    py = py < P - 1 ? py : P - 1;
    px = px < P - 1 ? px : P - 1;
    if (flip) px = P - 1 - px;
    const int sy = py - pad_top, sx = px - pad_left;
    float r = 0.f, g = 0.f, b = 0.f;  // pad_value 0 (datasets.py:211)
    if ((unsigned)sy < (unsigned)h && (unsigned)sx < (unsigned)w) {
      const unsigned char* p = src + ((size_t)sy * w + sx) * 3;
      r = (float)p[0] / 255.f;  // ToTensor: .float().div(255) - IEEE division
      g = (float)p[1] / 255.f;
      b = (float)p[2] / 255.f;
    }
    dst[idx] = r;
    dst[total + idx] = g;
    dst[2 * total + idx] = b;
  }
}

// ---- radar heat map ------------------------------------------------------------------------------------------
constexpr int HM_MAX = 64;  // max bins per side (radar_maps_size is 32 in the reference)

// bin index of np.histogramdd: searchsorted(edges, v, side="right") - 1 with edges = linspace(0, hi, nb + 1)
// (edge i = i * step in float64, last edge = hi exactly); v == hi belongs to the last bin; -1 = outside.
__device__ inline int hist_bin(double v, double hi, int nb) {
  if (!(v >= 0.0) || !(v <= hi)) return -1;
  if (v == hi) return nb - 1;
  const double step = hi / (double)nb;
  int b = (int)(v / step);
  if (b > nb - 1) b = nb - 1;
  // make b the largest i with edge_i <= v (the division can be off by one ulp either way)
  while (b > 0 && (double)b * step > v) --b;
  while (b + 1 < nb && (double)(b + 1) * step <= v) ++b;
  return b;
}

struct HeatP {
  const double* points;   // [total, 4] = (u, v, depth, velocity) rows, frames concatenated
  const int* offsets;     // [n + 1] row range of frame i
  const int* sizes;       // [n, 2] = (w, h) of the original image
  float* out;             // [n, 3, ms, ms]
  int maps_size;          // radar_maps_size (32)
  int ms;                 // output side (img_size / 16)
};

__global__ __launch_bounds__(256) void radar_heatmap_kernel(HeatP p) {
  __shared__ float maps[3][HM_MAX * HM_MAX];  // [c][y * bw + x], already float32 (ToTensor().float())
  const int f = blockIdx.x;
  const int img_w = p.sizes[2 * f], img_h = p.sizes[2 * f + 1];
  const int r0 = p.offsets[f], npts = p.offsets[f + 1] - r0;
  // scale = max(img_size) / radar_maps_size; bin_w, bin_h = round(w / scale), round(h / scale): python round =
  // round-half-even on the float64 quotient
  const double scale = (double)(img_w > img_h ? img_w : img_h) / (double)p.maps_size;
  const int bw = (int)rint((double)img_w / scale), bh = (int)rint((double)img_h / scale);
  const int nbins = bw * bh;
  // one thread per bin walks the points in order: the weighted sums are the sequential float64 sums np.bincount makes
  for (int b = threadIdx.x; b < nbins; b += 256) {
    const int by = b / bw, bx = b - by * bw;
    double cnt = 0.0, sd = 0.0, sv = 0.0;
    for (int i = 0; i < npts; ++i) {
      const double* q = p.points + (size_t)(r0 + i) * 4;
      const int ix = hist_bin(q[0], (double)img_w, bw);
      const int iy = hist_bin(q[1], (double)img_h, bh);
      if (ix == bx && iy == by) {
        cnt += 1.0;
        sd += q[2];
        sv += q[3];
      }
    }
    double h1 = sd / (cnt + 1e-6);          // mean depth per bin
    h1 = h1 < 1.0 ? 100.0 : h1;             // empty / too close -> far away
    const double h2 = fabs(sv / (cnt + 1e-6));
    // ranges ((0,5), (12,0), (0,4)) then clip to [0,1]
    double c0 = (cnt - 0.0) / (5.0 - 0.0), c1 = (h1 - 12.0) / (0.0 - 12.0), c2 = (h2 - 0.0) / (4.0 - 0.0);
    c0 = fmin(fmax(c0, 0.0), 1.0);
    c1 = fmin(fmax(c1, 0.0), 1.0);
    c2 = fmin(fmax(c2, 0.0), 1.0);
    maps[0][b] = (float)c0;
    maps[1][b] = (float)c1;
    maps[2][b] = (float)c2;
  }
  __syncthreads();
  // pad_to_square (zeros) + bilinear, align_corners=True (aten upsample_bilinear2d, float32 arithmetic)
  const int P = bh > bw ? bh : bw;
  const int diff = bh > bw ? bh - bw : bw - bh;
  const int pad1 = diff / 2;
  const int pad_top = bh <= bw ? pad1 : 0, pad_left = bh <= bw ? 0 : pad1;
  const int ms = p.ms;
  const float rs = ms > 1 ? (float)(P - 1) / (float)(ms - 1) : 0.f;
  auto at = [&](int c, int y, int x) -> float {
    const int sy = y - pad_top, sx = x - pad_left;
    return ((unsigned)sy < (unsigned)bh && (unsigned)sx < (unsigned)bw) ? maps[c][sy * bw + sx] : 0.f;
  };
  float* out = p.out + (size_t)f * 3 * ms * ms;
  for (int idx = threadIdx.x; idx < 3 * ms * ms; idx += 256) {
    const int c = idx / (ms * ms), rem = idx - c * ms * ms;
    const int oy = rem / ms, ox = rem - oy * ms;
    float v;
    if (P == ms) {
      v = at(c, oy, ox);
    } else {
      const float h1r = rs * (float)oy, w1r = rs * (float)ox;
      const int h1 = (int)h1r, w1 = (int)w1r;
      const int h1p = h1 < P - 1 ? 1 : 0, w1p = w1 < P - 1 ? 1 : 0;
      const float h1l = h1r - (float)h1, w1l = w1r - (float)w1;
      const float h0l = 1.f - h1l, w0l = 1.f - w1l;
      v = h0l * (w0l * at(c, h1, w1) + w1l * at(c, h1, w1 + w1p)) +
          h1l * (w0l * at(c, h1 + h1p, w1) + w1l * at(c, h1 + h1p, w1 + w1p));
    }
    out[idx] = v;
  }
}

}  // namespace

extern "C" {

int me_image_pad_resize_u8_f32(const uint8_t* src, int32_t h, int32_t w, float* dst, int32_t size, void* stream) {
  ME_REQUIRE(src && dst, ME_E_NULLPTR, "me_image_pad_resize_u8_f32: null pointer");
  ME_REQUIRE(h > 0 && w > 0 && size > 0, ME_E_BADARG, "me_image_pad_resize_u8_f32: non-positive size");
  ME_REQUIRE((long long)size * size < (1ll << 30), ME_E_TOOBIG, "me_image_pad_resize_u8_f32: output too large");
  int blocks = (int)me::ceil_div((int64_t)size * size, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(image_pad_resize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, h, w, dst, size, 0);
  return me::check_launch("image_pad_resize_kernel");
}

int me_image_pad_resize_flip_u8_f32(const uint8_t* src, int32_t h, int32_t w, float* dst, int32_t size, int32_t flip,
                                    void* stream) {
  ME_REQUIRE(src && dst, ME_E_NULLPTR, "me_image_pad_resize_flip_u8_f32: null pointer");
  ME_REQUIRE(h > 0 && w > 0 && size > 0, ME_E_BADARG, "me_image_pad_resize_flip_u8_f32: non-positive size");
  ME_REQUIRE((long long)size * size < (1ll << 30), ME_E_TOOBIG, "me_image_pad_resize_flip_u8_f32: output too large");
  int blocks = (int)me::ceil_div((int64_t)size * size, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(image_pad_resize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, h, w, dst, size,
                     flip ? 1 : 0);
  return me::check_launch("image_pad_resize_kernel");
}

int me_radar_heatmap_f32(const double* points, const int32_t* offsets, const int32_t* sizes, int32_t n,
                         int32_t radar_maps_size, float* out, int32_t map_size, void* stream) {
  if (n == 0) return 0;
  ME_REQUIRE(points && offsets && sizes && out, ME_E_NULLPTR, "me_radar_heatmap_f32: null pointer");
  ME_REQUIRE(n > 0 && map_size > 0, ME_E_BADARG, "me_radar_heatmap_f32: bad n / map_size");
  ME_REQUIRE(radar_maps_size >= 1 && radar_maps_size <= HM_MAX, ME_E_BADARG,
             "me_radar_heatmap_f32: radar_maps_size %d outside [1, %d]", radar_maps_size, HM_MAX);
  HeatP p{points, offsets, sizes, out, radar_maps_size, map_size};
  hipLaunchKernelGGL(radar_heatmap_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, p);
  return me::check_launch("radar_heatmap_kernel");
}

}  // extern "C"

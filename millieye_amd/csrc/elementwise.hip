// elementwise.hip - HBM-bound NHWC helpers of the detector graph (gfx950).
//   maxpool (incl. darknet's zero-padded 2x2/stride-1 pool), nearest upsample, pitched add /
//   copy (fallbacks for [shortcut] / [route] when they cannot be fused into a conv epilogue),
//   NHWC->NCHW export, and the YOLO head decode.
// All kernels move 16 bytes per lane when pitches allow (channels % 4 == 0), consecutive
// lanes walk the channel dimension first -> fully coalesced NHWC accesses.
#include <math.h>
#include "common.h"

namespace {

constexpr int kThreads = 256;

inline unsigned grid_for(long long work) {
  long long b = (work + kThreads - 1) / kThreads;
  const long long cap = 256ll * 32;  // 256 CUs x 32 workgroups, grid-stride beyond
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ---- max pooling ----------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(kThreads) void maxpool_kernel(me_pool_desc d) {
  const int cv = d.c / V;
  const long long total = (long long)d.n * d.ho * d.wo * cv;
  for (long long idx = (long long)blockIdx.x * kThreads + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * kThreads) {
    const int c = (int)(idx % cv) * V;
    long long pix = idx / cv;
    const int ox = (int)(pix % d.wo);
    pix /= d.wo;
    const int oy = (int)(pix % d.ho);
    const int nimg = (int)(pix / d.ho);
    float best[V];
#pragma unroll
    for (int v = 0; v < V; ++v) best[v] = -INFINITY;
    for (int ky = 0; ky < d.size; ++ky) {
      const int iy = oy * d.stride - d.pad + ky;
      for (int kx = 0; kx < d.size; ++kx) {
        const int ix = ox * d.stride - d.pad + kx;
        const bool inside = (unsigned)iy < (unsigned)d.h && (unsigned)ix < (unsigned)d.w;
        if (inside) {
          const float* src = d.x + ((long long)(nimg * d.h + iy) * d.w + ix) * d.x_pitch + c;
          if (V == 4) {
            const float4 t = *reinterpret_cast<const float4*>(src);
            best[0] = fmaxf(best[0], t.x);
            best[1 % V] = fmaxf(best[1 % V], t.y);
            best[2 % V] = fmaxf(best[2 % V], t.z);
            best[3 % V] = fmaxf(best[3 % V], t.w);
          } else {
            best[0] = fmaxf(best[0], src[0]);
          }
        } else if (d.zero_ext && iy >= 0 && ix >= 0 && iy <= d.h && ix <= d.w) {
          // ZeroPad2d((0,1,0,1)): one extra row / column of zeros that takes part in the max
#pragma unroll
          for (int v = 0; v < V; ++v) best[v] = fmaxf(best[v], 0.f);
        }
      }
    }
    float* dst = d.y + ((long long)(nimg * d.ho + oy) * d.wo + ox) * d.y_pitch + c;
    if (V == 4)
      *reinterpret_cast<float4*>(dst) = make_float4(best[0], best[1 % V], best[2 % V], best[3 % V]);
    else
      dst[0] = best[0];
  }
}

// ---- nearest upsample -----------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(kThreads) void upsample_kernel(const float* x, long long xp, float* y, long long yp,
                                                            int n, int h, int w, int c, int f) {
  const int cv = c / V;
  const int ho = h * f, wo = w * f;
  const long long total = (long long)n * ho * wo * cv;
  for (long long idx = (long long)blockIdx.x * kThreads + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * kThreads) {
    const int cc = (int)(idx % cv) * V;
    long long pix = idx / cv;
    const int ox = (int)(pix % wo);
    pix /= wo;
    const int oy = (int)(pix % ho);
    const int nimg = (int)(pix / ho);
    const float* src = x + ((long long)(nimg * h + oy / f) * w + ox / f) * xp + cc;
    float* dst = y + ((long long)(nimg * ho + oy) * wo + ox) * yp + cc;
    if (V == 4)
      *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
    else
      dst[0] = src[0];
  }
}

// ---- pitched add / copy ----------------------------------------------------------------------
template <int V, bool ADD>
__global__ __launch_bounds__(kThreads) void addcopy_kernel(const float* a, long long ap, const float* b, long long bp,
                                                           float* y, long long yp, long long pixels, int c) {
  const int cv = c / V;
  const long long total = pixels * cv;
  for (long long idx = (long long)blockIdx.x * kThreads + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * kThreads) {
    const int cc = (int)(idx % cv) * V;
    const long long pix = idx / cv;
    if (V == 4) {
      float4 u = *reinterpret_cast<const float4*>(a + pix * ap + cc);
      if (ADD) {
        const float4 t = *reinterpret_cast<const float4*>(b + pix * bp + cc);
        u.x += t.x; u.y += t.y; u.z += t.z; u.w += t.w;
      }
      *reinterpret_cast<float4*>(y + pix * yp + cc) = u;
    } else {
      float u = a[pix * ap + cc];
      if (ADD) u += b[pix * bp + cc];
      y[pix * yp + cc] = u;
    }
  }
}

// ---- NHWC -> dense NCHW (API boundary only): 32x32 LDS transpose over (pixel, channel) --------
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* x, long long xp, float* y, int hw, int c) {
  __shared__ float tile[32][33];
  const int nimg = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int pix = p0 + r, ch = c0 + tx;
    tile[r][tx] = (pix < hw && ch < c) ? x[((long long)nimg * hw + pix) * xp + ch] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int ch = c0 + r, pix = p0 + tx;
    if (pix < hw && ch < c) y[((long long)nimg * c + ch) * hw + pix] = tile[tx][r];
  }
}

// ---- YOLO decode -------------------------------------------------------------------------------
// one thread per output element (n, a, pixel, k): reads and writes are both contiguous in k.
__global__ __launch_bounds__(kThreads) void yolo_decode_kernel(me_yolo_desc d) {
  // blockIdx.y = (image, anchor); blockIdx.x strides over that anchor's g*g*(5+C) elements: one 32-bit division per
  // element (the flat 64-bit index needed three), reads and writes contiguous in k
  const unsigned per = d.num_classes + 5;
  const unsigned gg = d.g * d.g;
  const unsigned a = blockIdx.y % d.num_anchors, nimg = blockIdx.y / d.num_anchors;
  const unsigned total = gg * per;
  const float* xin = d.x + (long long)nimg * gg * d.x_pitch + a * per;
  float* out = d.out + ((long long)nimg * d.rows_total + d.row_offset + (long long)a * gg) * per;
  const float aw = d.anchors[2 * a], ah = d.anchors[2 * a + 1];
  for (unsigned idx = blockIdx.x * kThreads + threadIdx.x; idx < total; idx += gridDim.x * kThreads) {
    const unsigned pix = idx / per;
    const unsigned k = idx - pix * per;
    const float v = xin[(long long)pix * d.x_pitch + k];
    float o;
    if (k < 2) {
      const float s = 1.f / (1.f + expf(-v));
      const float g = (k == 0) ? (float)(pix % d.g) : (float)(pix / d.g);
      o = (s + g) * d.stride;
    } else if (k < 4) {
      // reference order: exp(t) * (anchor / stride), then * stride (yolov3/models.py:126,162-163,168)
      o = (expf(v) * (k == 2 ? aw : ah)) * d.stride;  // anchors are pre-divided by stride
    } else {
      o = 1.f / (1.f + expf(-v));
    }
    out[idx] = o;
  }
}

}  // namespace

extern "C" {

int me_maxpool_f32(const me_pool_desc* d, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(d && d->x && d->y, ME_E_NULLPTR, "me_maxpool_f32: null pointer");
  ME_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->c > 0 && d->size >= 1 && d->stride >= 1 && d->pad >= 0,
             ME_E_BADARG, "me_maxpool_f32: bad dimensions");
  const int ext = d->zero_ext ? 1 : 0;
  const int ho = (d->h + ext + 2 * d->pad - d->size) / d->stride + 1;
  const int wo = (d->w + ext + 2 * d->pad - d->size) / d->stride + 1;
  ME_REQUIRE(ho == d->ho && wo == d->wo, ME_E_BADARG, "me_maxpool_f32: ho/wo (%d,%d) != derived (%d,%d)", d->ho,
             d->wo, ho, wo);
  ME_REQUIRE(d->x_pitch >= d->c && d->y_pitch >= d->c, ME_E_BADARG, "me_maxpool_f32: pitch < c");
  const bool vec = (d->c % 4 == 0) && (d->x_pitch % 4 == 0) && (d->y_pitch % 4 == 0) && me::aligned16(d->x) &&
                   me::aligned16(d->y);
  const long long work = (long long)d->n * d->ho * d->wo * (vec ? d->c / 4 : d->c);
  if (vec)
    hipLaunchKernelGGL(maxpool_kernel<4>, dim3(grid_for(work)), dim3(kThreads), 0, stream, *d);
  else
    hipLaunchKernelGGL(maxpool_kernel<1>, dim3(grid_for(work)), dim3(kThreads), 0, stream, *d);
  return me::check_launch("maxpool_kernel");
}

int me_upsample_f32(const float* x, int64_t x_pitch, float* y, int64_t y_pitch, int32_t n, int32_t h, int32_t w,
                    int32_t c, int32_t factor, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(x && y, ME_E_NULLPTR, "me_upsample_f32: null pointer");
  ME_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && factor >= 1, ME_E_BADARG, "me_upsample_f32: bad dimensions");
  ME_REQUIRE(x_pitch >= c && y_pitch >= c, ME_E_BADARG, "me_upsample_f32: pitch < c");
  const bool vec = (c % 4 == 0) && (x_pitch % 4 == 0) && (y_pitch % 4 == 0) && me::aligned16(x) && me::aligned16(y);
  const long long work = (long long)n * h * factor * w * factor * (vec ? c / 4 : c);
  if (vec)
    hipLaunchKernelGGL(upsample_kernel<4>, dim3(grid_for(work)), dim3(kThreads), 0, stream, x, (long long)x_pitch, y,
                       (long long)y_pitch, n, h, w, c, factor);
  else
    hipLaunchKernelGGL(upsample_kernel<1>, dim3(grid_for(work)), dim3(kThreads), 0, stream, x, (long long)x_pitch, y,
                       (long long)y_pitch, n, h, w, c, factor);
  return me::check_launch("upsample_kernel");
}

int me_add_f32(const float* a, int64_t a_pitch, const float* b, int64_t b_pitch, float* y, int64_t y_pitch,
               int64_t pixels, int32_t c, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(a && b && y, ME_E_NULLPTR, "me_add_f32: null pointer");
  ME_REQUIRE(pixels > 0 && c > 0 && a_pitch >= c && b_pitch >= c && y_pitch >= c, ME_E_BADARG,
             "me_add_f32: bad dimensions");
  const bool vec = (c % 4 == 0) && (a_pitch % 4 == 0) && (b_pitch % 4 == 0) && (y_pitch % 4 == 0) &&
                   me::aligned16(a) && me::aligned16(b) && me::aligned16(y);
  const long long work = pixels * (vec ? c / 4 : c);
  if (vec)
    hipLaunchKernelGGL((addcopy_kernel<4, true>), dim3(grid_for(work)), dim3(kThreads), 0, stream, a,
                       (long long)a_pitch, b, (long long)b_pitch, y, (long long)y_pitch, (long long)pixels, c);
  else
    hipLaunchKernelGGL((addcopy_kernel<1, true>), dim3(grid_for(work)), dim3(kThreads), 0, stream, a,
                       (long long)a_pitch, b, (long long)b_pitch, y, (long long)y_pitch, (long long)pixels, c);
  return me::check_launch("add_kernel");
}

int me_copy_f32(const float* x, int64_t x_pitch, float* y, int64_t y_pitch, int64_t pixels, int32_t c,
                void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(x && y, ME_E_NULLPTR, "me_copy_f32: null pointer");
  ME_REQUIRE(pixels > 0 && c > 0 && x_pitch >= c && y_pitch >= c, ME_E_BADARG, "me_copy_f32: bad dimensions");
  const bool vec = (c % 4 == 0) && (x_pitch % 4 == 0) && (y_pitch % 4 == 0) && me::aligned16(x) && me::aligned16(y);
  const long long work = pixels * (vec ? c / 4 : c);
  if (vec)
    hipLaunchKernelGGL((addcopy_kernel<4, false>), dim3(grid_for(work)), dim3(kThreads), 0, stream, x,
                       (long long)x_pitch, (const float*)nullptr, 0ll, y, (long long)y_pitch, (long long)pixels, c);
  else
    hipLaunchKernelGGL((addcopy_kernel<1, false>), dim3(grid_for(work)), dim3(kThreads), 0, stream, x,
                       (long long)x_pitch, (const float*)nullptr, 0ll, y, (long long)y_pitch, (long long)pixels, c);
  return me::check_launch("copy_kernel");
}

int me_nhwc_to_nchw_f32(const float* x, int64_t x_pitch, float* y, int32_t n, int32_t h, int32_t w, int32_t c,
                        void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(x && y, ME_E_NULLPTR, "me_nhwc_to_nchw_f32: null pointer");
  ME_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && x_pitch >= c && n <= 65535, ME_E_BADARG,
             "me_nhwc_to_nchw_f32: bad dimensions");
  const int hw = h * w;
  dim3 grid((hw + 31) / 32, (c + 31) / 32, n);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, stream, x, (long long)x_pitch, y, hw, c);
  return me::check_launch("nhwc_to_nchw_kernel");
}

int me_yolo_decode_f32(const me_yolo_desc* d, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(d && d->x && d->out, ME_E_NULLPTR, "me_yolo_decode_f32: null pointer");
  ME_REQUIRE(d->n > 0 && d->g > 0 && d->num_anchors > 0 && d->num_anchors <= 8 && d->num_classes >= 0, ME_E_BADARG,
             "me_yolo_decode_f32: bad dimensions");
  ME_REQUIRE(d->x_pitch >= d->num_anchors * (d->num_classes + 5), ME_E_BADARG, "me_yolo_decode_f32: x_pitch too small");
  ME_REQUIRE(d->row_offset >= 0 && d->row_offset + d->num_anchors * d->g * d->g <= d->rows_total, ME_E_BADARG,
             "me_yolo_decode_f32: rows out of range");
  ME_REQUIRE(d->stride > 0.f, ME_E_BADARG, "me_yolo_decode_f32: stride must be positive");
  const long long per_anchor = (long long)d->g * d->g * (d->num_classes + 5);
  ME_REQUIRE(per_anchor < (1ll << 31) && (long long)d->n * d->num_anchors < 65536, ME_E_TOOBIG,
             "me_yolo_decode_f32: grid too large");
  long long bx = (per_anchor + kThreads - 1) / kThreads;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(yolo_decode_kernel, dim3((unsigned)bx, (unsigned)(d->n * d->num_anchors)), dim3(kThreads), 0, stream,
                     *d);
  return me::check_launch("yolo_decode_kernel");
}

}  // extern "C"

// heads.hip - proposal assembly, RoI pooling and the fused refinement / ensemble heads (gfx950).
//
// Replaces, for inference, the tail of Network.forward (module3_our_dataset/my_models.py):
//   :459-473  per-image python loop building img_boxes            -> gather_class_boxes_kernel
//   :495-496  torchvision ps_roi_align / roi_align                -> ps_sample / roi_sample below
//   :260-284  refinement_head.forward, :202-210 ensemble_head     -> roi_heads_kernel (one launch)
//   :502-539  masks, thresholds, box_regress (:378-391), sort key -> same launch
// RoI pooling semantics follow torchvision 0.6 exactly as restated in oracle/tv_ops.c (loop and
// summation order included; FP contraction is off in the sampling code so pooled values are
// bit-identical with that oracle).  Samples that the border rule maps to 0 are skipped by
// bounding the sample loops to the feature map (adding +0.0f is exact), which also bounds the
// work for degenerate huge boxes.
//
// WHICH KERNELS ARE PRODUCT (three generations live side by side in this file):
//   inference, stage 3 (Network.forward -> me_roi_heads_f32 with me_heads_desc.pool_scratch set):
//       roi_pool10_kernel (a thread per bin, ten channels each)  +  roi_heads_mfma_kernel (32 RoIs per workgroup, net0 on the fp32
//       matrix pipe)  +  compact_sort_kernel.                                                        <- the product path
//   training, stage 3 (train_path.py: save_* pointers set): roi_heads_kernel (8 RoIs per workgroup, VALU chains; writes the saved
//       activations the backward needs) and the *_bwd kernels.                                         <- product for training
//   cross-checks only: roi_heads_kernel as the single fused launch of inference (pool_scratch == NULL; Network._fused_heads,
//       tests/test_gpu_network.py::test_two_launch_heads_equal_the_fused_launch), roi_pool_kernel (MILLIEYE_POOL10=0: a thread per
//       pooled value), MILLIEYE_HEADS_MFMA=0 (VALU heads behind the pooling launch).
//   stage 2 (module2): m2_pool_kernel + m2_heads kernels.
#include <math.h>
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int P = 7;            // pooled size (my_models.py:495-496)
constexpr int PP = P * P;       // 49
constexpr int C_OUT = 10;       // score-map groups
constexpr int FEAT = C_OUT * PP;  // 490
constexpr int HID = 256;
constexpr int RPB = 8;          // RoIs per workgroup

__device__ __forceinline__ float bilinear_nhwc(const float* map, long long pitch, int height, int width, int c,
                                               float y, float x) {
#pragma clang fp contract(off)
  if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) return 0.0f;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else { y_high = y_low + 1; }
  if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else { x_high = x_low + 1; }
  const float ly = y - y_low, lx = x - x_low;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float v1 = map[((long long)y_low * width + x_low) * pitch + c];
  const float v2 = map[((long long)y_low * width + x_high) * pitch + c];
  const float v3 = map[((long long)y_high * width + x_low) * pitch + c];
  const float v4 = map[((long long)y_high * width + x_high) * pitch + c];
  const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
}

// index range [lo, hi] of samples start + ((i + .5f) * bin) / g that can fall inside [-1, limit];
// conservative (2 extra on each side) - the exact test stays in bilinear_nhwc.
__device__ __forceinline__ void sample_range(float start, float bin, int g, float limit, int* lo, int* hi) {
  *lo = 0;
  *hi = g - 1;
  if (g <= 0) return;
  const float step = bin / (float)g;
  if (!(step > 0.f) || !isfinite(start) || !isfinite(step)) {
    // non-finite geometry (inf / NaN boxes): the reference's behaviour is undefined there
    // ((int)NaN, 2^31-iteration loops); take no samples so the launch stays bounded.
    *hi = -1;
    return;
  }
  const float a = (-1.0f - start) / step - 0.5f;
  const float b = (limit - start) / step - 0.5f;
  if (a > 2.f) *lo = (a - 2.f >= (float)g) ? g : (int)(a - 2.f);
  if (b + 2.f < (float)(g - 1)) *hi = (b + 2.f < 0.f) ? -1 : (int)(b + 2.f);
}

__device__ __forceinline__ int grid_of(float extent) {
  // (int)ceil(roi / pooled) like the reference; saturate instead of overflowing for absurd boxes
  const float g = ceilf(extent / (float)P);
  if (!(g < 1.0e9f)) return (g != g) ? 0 : 1000000000;
  if (g < -1.0e9f) return -1000000000;
  return (int)g;
}

// one output of torchvision.ops.roi_align(aligned=False, sampling_ratio=-1), map NHWC [n,h,w,c]
__device__ float roi_sample(const float* map, long long pitch, int height, int width, const float* roi,
                            float scale, int c, int ph, int pw) {
#pragma clang fp contract(off)
  const int b = (int)roi[0];
  const float sw = roi[1] * scale - 0.0f, sh = roi[2] * scale - 0.0f;
  const float ew = roi[3] * scale - 0.0f, eh = roi[4] * scale - 0.0f;
  float roi_w = ew - sw, roi_h = eh - sh;
  roi_w = roi_w > 1.f ? roi_w : 1.f;
  roi_h = roi_h > 1.f ? roi_h : 1.f;
  const float bin_h = roi_h / (float)P, bin_w = roi_w / (float)P;
  const int gh = grid_of(roi_h), gw = grid_of(roi_w);
  const int cnt = gh * gw;
  const float count = (float)(cnt > 1 ? cnt : 1);
  const float* img = map + (long long)b * height * width * pitch;
  const float ybase = sh + ph * bin_h, xbase = sw + pw * bin_w;
  int ylo, yhi, xlo, xhi;
  sample_range(ybase, bin_h, gh, (float)height, &ylo, &yhi);
  sample_range(xbase, bin_w, gw, (float)width, &xlo, &xhi);
  float acc = 0.f;
  for (int iy = ylo; iy <= yhi; ++iy) {
    const float yy = ybase + ((float)(iy + .5f)) * bin_h / (float)gh;
    for (int ix = xlo; ix <= xhi; ++ix) {
      const float xx = xbase + ((float)(ix + .5f)) * bin_w / (float)gw;
      acc += bilinear_nhwc(img, pitch, height, width, c, yy, xx);
    }
  }
  return acc / count;
}

// one output of torchvision.ops.ps_roi_align(sampling_ratio=-1): input channel c_in = (c_out*P+ph)*P+pw
__device__ float ps_sample(const float* map, long long pitch, int height, int width, const float* roi, float scale,
                           int c_in, int ph, int pw) {
#pragma clang fp contract(off)
  const int b = (int)roi[0];
  const float sw = roi[1] * scale - 0.5f, sh = roi[2] * scale - 0.5f;
  const float ew = roi[3] * scale - 0.5f, eh = roi[4] * scale - 0.5f;
  const float roi_w = ew - sw, roi_h = eh - sh;
  const float bin_h = roi_h / (float)P, bin_w = roi_w / (float)P;
  const float hstart = (float)ph * bin_h + sh;
  const float wstart = (float)pw * bin_w + sw;
  const int gh = grid_of(roi_h), gw = grid_of(roi_w);
  const float count = (float)(gh * gw);
  const float* img = map + (long long)b * height * width * pitch;
  int ylo, yhi, xlo, xhi;
  sample_range(hstart, bin_h, gh, (float)height, &ylo, &yhi);
  sample_range(wstart, bin_w, gw, (float)width, &xlo, &xhi);
  float out_sum = 0.f;
  for (int iy = ylo; iy <= yhi; ++iy) {
    const float y = hstart + ((float)(iy + .5f)) * bin_h / (float)gh;
    for (int ix = xlo; ix <= xhi; ++ix) {
      const float x = wstart + ((float)(ix + .5f)) * bin_w / (float)gw;
      out_sum += bilinear_nhwc(img, pitch, height, width, c_in, y, x);
    }
  }
  return out_sum / count;
}

__global__ __launch_bounds__(256) void roi_align_kernel(const float* map, long long pitch, int h, int w, int c,
                                                        const float* rois, int k, float scale, float* out,
                                                        int ps) {
  const long long total = (long long)k * c * PP / (ps ? PP : 1);
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int pw = (int)(idx % P);
    const int ph = (int)((idx / P) % P);
    const int cc = (int)((idx / PP) % (ps ? c / PP : c));
    const int r = (int)(idx / ((long long)PP * (ps ? c / PP : c)));
    out[idx] = ps ? ps_sample(map, pitch, h, w, rois + 5 * r, scale, (cc * P + ph) * P + pw, ph, pw)
                  : roi_sample(map, pitch, h, w, rois + 5 * r, scale, cc, ph, pw);
  }
}

// ---- proposal assembly --------------------------------------------------------------------------
// One workgroup per image.  Rows keep the reference's image-major order, so image i starts at the number of rows the
// images before it contribute: every workgroup counts those itself (<= n * max_det independent loads spread over 256
// threads = one memory latency) instead of one workgroup walking all slots serially (7 dependent rounds of load + three
// barriers at batch 32: 70 us -> see profiles/r02_kernel_evolution.md).
__global__ __launch_bounds__(256) void gather_class_boxes_kernel(const float* det, const int* count, int n,
                                                                 int max_det, int num_classes, int class_idx,
                                                                 int class_num, float* boxes, int* total) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int img = blockIdx.x;
  const int width = 7 + num_classes, cols = 8 + class_num;
  auto taken = [&](int im, int k) {
    return k < count[im] && (class_idx < 0 || det[((long long)im * max_det + k) * width + 6] == (float)class_idx);
  };
  // rows of the images in front of this one
  int before_imgs = 0;
  if (class_idx < 0) {
    for (int j = t; j < img; j += 256) before_imgs += count[j] < max_det ? count[j] : max_det;
  } else {
    const int slots = img * max_det;
    for (int sidx = t; sidx < slots; sidx += 256) {
      const int j = sidx / max_det;
      before_imgs += taken(j, sidx - j * max_det) ? 1 : 0;
    }
  }
  for (int o = 32; o > 0; o >>= 1) before_imgs += __shfl_down(before_imgs, o, 64);
  if (lane == 0) s_wave[wv] = before_imgs;
  __syncthreads();
  if (t == 0) s_base = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  __syncthreads();
  // own rows, 256 slots at a time, order kept by ballot + prefix counts
  for (int k0 = 0; k0 < max_det; k0 += 256) {
    const int k = k0 + t;
    const bool take = k < max_det && taken(img, k);
    const unsigned long long m = __ballot(take);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();  // s_wave is reused
    if (lane == 0) s_wave[wv] = __popcll(m);
    __syncthreads();
    int woff = 0, all = 0;
    for (int q = 0; q < 4; ++q) {
      if (q < wv) woff += s_wave[q];
      all += s_wave[q];
    }
    const int base = s_base;
    if (take) {
      const float* d = det + ((long long)img * max_det + k) * width;
      float* o = boxes + (long long)(base + woff + before) * cols;
      o[0] = (float)img;
      for (int c = 0; c < 7 + class_num; ++c) o[1 + c] = d[c];
    }
    __syncthreads();
    if (t == 0) s_base = base + all;
    __syncthreads();
  }
  if (t == 0 && img == n - 1) *total = s_base;
}

// ---- fused RoI pooling + heads ---------------------------------------------------------------------
__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : 0.1f * v; }
__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }

// scalar tail of one RoI (shared by the fused inference kernel and me_heads_tail_f32)
struct TailW {  // the tail's small weights: ensemble head fc1 [32][2] / bias [32] / fc2 [2][64], radar_net BN + 1x1
  const float *e1w, *e1b, *e2w, *rw2, *rscale, *rshift, *rb2;
};
__device__ __forceinline__ TailW tail_weights(const me_heads_desc& d) {
  return TailW{d.wts.e1w, d.wts.e1b, d.wts.e2w, d.wts.rw2, d.wts.rscale, d.wts.rshift, d.wts.rb2};
}

__device__ __forceinline__ void tail_one(const me_heads_desc& d, const float* sm, const float* roi, int k, int n_img,
                                         const TailW& tw) {
  const float *e1w = tw.e1w, *e1b = tw.e1b, *e2w = tw.e2w;
  const float cls0 = sigmoidf(sm[4]), cls1 = sigmoidf(sm[5]);
  float rad = tw.rb2[0];
#pragma unroll
  for (int o = 0; o < C_OUT; ++o) rad = fmaf(tw.rw2[o], leaky(sm[6 + o] * tw.rscale[o] + tw.rshift[o]), rad);
  const float radar_conf = sigmoidf(rad);
  const float conf = sigmoidf(radar_conf + cls0);  // sigmoid applied twice on purpose (quirk q2)
  const bool is_img = k < n_img;
  float p;
  float c6, c7;
  if (is_img) {
    const float* bx = d.img_boxes + (long long)k * d.box_cols;
    const float yolo0 = bx[5], yolo1 = bx[8];
    // ensemble_head: stack -> fc1 (2->32) + leaky -> flatten(64) -> fc2 (64->2) -> softmax; column 0 (q1)
    float o0 = d.wts.e2b[0], o1 = d.wts.e2b[1];
#pragma unroll 4
    for (int u = 0; u < 32; ++u) {
      const float wa = e1w[2 * u], wb = e1w[2 * u + 1], bb = e1b[u];
      const float h0 = leaky(wa * conf + wb * yolo0 + bb);
      const float h1 = leaky(wa * cls1 + wb * yolo1 + bb);
      o0 += e2w[u] * h0 + e2w[32 + u] * h1;
      o1 += e2w[64 + u] * h0 + e2w[96 + u] * h1;
    }
    const float m = fmaxf(o0, o1);
    const float e0 = expf(o0 - m), e1 = expf(o1 - m);
    p = e0 / (e0 + e1);
    c6 = bx[6];
    c7 = bx[7];
  } else {
    p = conf;
    c6 = cls1;
    c7 = 0.f;
  }
  d.regress_out[4 * k + 0] = sm[0];
  d.regress_out[4 * k + 1] = sm[1];
  d.regress_out[4 * k + 2] = sm[2];
  d.regress_out[4 * k + 3] = sm[3];
  d.refine_out[2 * k + 0] = conf;
  d.refine_out[2 * k + 1] = cls1;
  d.mask1_out[k] = p;
  const float thr = is_img ? d.thr_img : d.thr_radar;
  d.keep[k] = (p > thr) ? 1 : 0;
  d.sort_key[k] = is_img ? p : p / 5.f;
  float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
  if (d.regress) {  // box_regress, my_models.py:378-391
    const float cx = (x1 + x2) / 2, cy = (y1 + y2) / 2, bw = x2 - x1, bh = y2 - y1;
    const float nx = sm[0] * bw + cx, ny = sm[1] * bh + cy;
    const float nw = expf(sm[2]) * bw, nh = expf(sm[3]) * bh;
    x1 = nx - nw / 2; y1 = ny - nh / 2; x2 = nx + nw / 2; y2 = ny + nh / 2;
  }
  float* o = d.out_rows + 8ll * k;
  o[0] = roi[0]; o[1] = x1; o[2] = y1; o[3] = x2; o[4] = y2; o[5] = p; o[6] = c6; o[7] = c7;
}

// RoI pooling of one RoI per workgroup into d.pool_scratch [cap][2 * FEAT] (see me_heads_desc.pool_scratch).  Thread order:
// (bin, channel) with the channel fastest - the ten lanes of a bin share their sample points, so on the radar map (NHWC, 10
// channels) and on a bin-major image map (me_heads_desc.img_bin_major) a sample corner is ONE 40-byte read per ten lanes instead
// of ten cache lines; the values go through LDS to their slots f = c * 49 + bin and leave as contiguous rows.
__global__ __launch_bounds__(256) void roi_pool_kernel(me_heads_desc d) {
  __shared__ float s_box[5];
  __shared__ float s_out[2 * FEAT];
  const int t = threadIdx.x;
  const int n_img = *d.n_img;
  const int k = blockIdx.x;
  if (k >= n_img + d.n_radar) return;
  if (t < 5) s_box[t] = (k < n_img) ? d.img_boxes[(long long)k * d.box_cols + t] : d.radar_boxes[(long long)(k - n_img) * 5 + t];
  __syncthreads();
  for (int i = t; i < 2 * FEAT; i += 256) {
    const int part = i >= FEAT ? 1 : 0, q = i - part * FEAT;
    const int bin = q / C_OUT, c = q - bin * C_OUT;
    const int ph = bin / P, pw = bin - ph * P;
    const int f = c * PP + bin;  // = (c * 7 + ph) * 7 + pw
    float v;
    if (!part) v = ps_sample(d.img_map, d.img_pitch, d.fh, d.fw, s_box, d.spatial_scale, d.img_bin_major ? q : f, ph, pw);
    else v = roi_sample(d.radar_map, d.radar_pitch, d.rh, d.rw, s_box, d.spatial_scale, c, ph, pw);
    s_out[part * FEAT + f] = v;
  }
  __syncthreads();
  float* out = d.pool_scratch + (long long)k * 2 * FEAT;
  for (int f = t; f < 2 * FEAT; f += 256) out[f] = s_out[f];
}

// ---- the same pooling, one thread per BIN and all ten channels --------------------------------------------------------------------
// roi_pool_kernel is VALU-bound: every (bin, channel) thread redoes the bin's sample geometry (two divisions per sample point,
// the range tests, the bilinear weights) - ten times per bin.  Here a thread owns a bin: geometry and weights once per sample
// point, then ten (value * weight) sums whose corner values are contiguous floats.  Per channel the operations and their
// order are those of ps_sample / roi_sample + bilinear_nhwc above (FP contraction off), so the pooled values are the same bits.
__device__ __forceinline__ void bilinear10(const float* img, long long pitch, int height, int width, int cbase, int cstride,
                                           float y, float x, float* acc) {
#pragma clang fp contract(off)
  if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) return;  // (the sample is +0: acc + 0 = acc)
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else { y_high = y_low + 1; }
  if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else { x_high = x_low + 1; }
  const float ly = y - y_low, lx = x - x_low;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  const float* p1 = img + ((long long)y_low * width + x_low) * pitch + cbase;
  const float* p2 = img + ((long long)y_low * width + x_high) * pitch + cbase;
  const float* p3 = img + ((long long)y_high * width + x_low) * pitch + cbase;
  const float* p4 = img + ((long long)y_high * width + x_high) * pitch + cbase;
  float v1[C_OUT], v2[C_OUT], v3[C_OUT], v4[C_OUT];
#pragma unroll
  for (int c = 0; c < C_OUT; ++c) {
    v1[c] = p1[c * cstride];
    v2[c] = p2[c * cstride];
    v3[c] = p3[c * cstride];
    v4[c] = p4[c * cstride];
  }
#pragma unroll
  for (int c = 0; c < C_OUT; ++c) acc[c] += (w1 * v1[c] + w2 * v2[c] + w3 * v3[c] + w4 * v4[c]);
}

// bin (ph, pw) of torchvision.ops.ps_roi_align: channels cbase + c * cstride, c < 10
__device__ void ps_sample10(const float* map, long long pitch, int height, int width, const float* roi, float scale,
                            int cbase, int cstride, int ph, int pw, float* out) {
#pragma clang fp contract(off)
  const int b = (int)roi[0];
  const float sw = roi[1] * scale - 0.5f, sh = roi[2] * scale - 0.5f;
  const float ew = roi[3] * scale - 0.5f, eh = roi[4] * scale - 0.5f;
  const float roi_w = ew - sw, roi_h = eh - sh;
  const float bin_h = roi_h / (float)P, bin_w = roi_w / (float)P;
  const float hstart = (float)ph * bin_h + sh;
  const float wstart = (float)pw * bin_w + sw;
  const int gh = grid_of(roi_h), gw = grid_of(roi_w);
  const float count = (float)(gh * gw);
  const float* img = map + (long long)b * height * width * pitch;
  int ylo, yhi, xlo, xhi;
  sample_range(hstart, bin_h, gh, (float)height, &ylo, &yhi);
  sample_range(wstart, bin_w, gw, (float)width, &xlo, &xhi);
  float acc[C_OUT];
#pragma unroll
  for (int c = 0; c < C_OUT; ++c) acc[c] = 0.f;
  for (int iy = ylo; iy <= yhi; ++iy) {
    const float y = hstart + ((float)(iy + .5f)) * bin_h / (float)gh;
    for (int ix = xlo; ix <= xhi; ++ix) {
      const float x = wstart + ((float)(ix + .5f)) * bin_w / (float)gw;
      bilinear10(img, pitch, height, width, cbase, cstride, y, x, acc);
    }
  }
#pragma unroll
  for (int c = 0; c < C_OUT; ++c) out[c] = acc[c] / count;
}

// bin (ph, pw) of torchvision.ops.roi_align(aligned=False, sampling_ratio=-1): channels 0 .. 9
__device__ void roi_sample10(const float* map, long long pitch, int height, int width, const float* roi, float scale,
                             int ph, int pw, float* out) {
#pragma clang fp contract(off)
  const int b = (int)roi[0];
  const float sw = roi[1] * scale - 0.0f, sh = roi[2] * scale - 0.0f;
  const float ew = roi[3] * scale - 0.0f, eh = roi[4] * scale - 0.0f;
  float roi_w = ew - sw, roi_h = eh - sh;
  roi_w = roi_w > 1.f ? roi_w : 1.f;
  roi_h = roi_h > 1.f ? roi_h : 1.f;
  const float bin_h = roi_h / (float)P, bin_w = roi_w / (float)P;
  const int gh = grid_of(roi_h), gw = grid_of(roi_w);
  const int cnt = gh * gw;
  const float count = (float)(cnt > 1 ? cnt : 1);
  const float* img = map + (long long)b * height * width * pitch;
  const float ybase = sh + ph * bin_h, xbase = sw + pw * bin_w;
  int ylo, yhi, xlo, xhi;
  sample_range(ybase, bin_h, gh, (float)height, &ylo, &yhi);
  sample_range(xbase, bin_w, gw, (float)width, &xlo, &xhi);
  float acc[C_OUT];
#pragma unroll
  for (int c = 0; c < C_OUT; ++c) acc[c] = 0.f;
  for (int iy = ylo; iy <= yhi; ++iy) {
    const float yy = ybase + ((float)(iy + .5f)) * bin_h / (float)gh;
    for (int ix = xlo; ix <= xhi; ++ix) {
      const float xx = xbase + ((float)(ix + .5f)) * bin_w / (float)gw;
      bilinear10(img, pitch, height, width, 0, 1, yy, xx, acc);
    }
  }
#pragma unroll
  for (int c = 0; c < C_OUT; ++c) out[c] = acc[c] / count;
}

// one RoI per workgroup of 128 threads: thread t < 49 = image bin t, 49 <= t < 98 = radar bin t - 49
__global__ __launch_bounds__(128) void roi_pool10_kernel(me_heads_desc d) {
  __shared__ float s_box[5];
  __shared__ float s_out[2 * FEAT];
  const int t = threadIdx.x;
  const int n_img = *d.n_img;
  const int k = blockIdx.x;
  if (k >= n_img + d.n_radar) return;
  if (t < 5) s_box[t] = (k < n_img) ? d.img_boxes[(long long)k * d.box_cols + t] : d.radar_boxes[(long long)(k - n_img) * 5 + t];
  __syncthreads();
  if (t < 2 * PP) {
    const int part = t >= PP ? 1 : 0, bin = t - part * PP;
    const int ph = bin / P, pw = bin - ph * P;
    float o[C_OUT];
    if (!part) {
      if (d.img_bin_major) ps_sample10(d.img_map, d.img_pitch, d.fh, d.fw, s_box, d.spatial_scale, bin * C_OUT, 1, ph, pw, o);
      else ps_sample10(d.img_map, d.img_pitch, d.fh, d.fw, s_box, d.spatial_scale, bin, PP, ph, pw, o);
    } else {
      roi_sample10(d.radar_map, d.radar_pitch, d.rh, d.rw, s_box, d.spatial_scale, ph, pw, o);
    }
#pragma unroll
    for (int c = 0; c < C_OUT; ++c) s_out[part * FEAT + c * PP + bin] = o[c];
  }
  __syncthreads();
  float* out = d.pool_scratch + (long long)k * 2 * FEAT;
  for (int f = t; f < 2 * FEAT; f += 128) out[f] = s_out[f];
}

template <int KC>  // rows of W0 per prefetched chunk (phase B)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KC == 4 ? 4 : 2))) void roi_heads_kernel(me_heads_desc d) {
  __shared__ __attribute__((aligned(16))) float s_feat[RPB][2 * FEAT];  // [r][0:490] image (PS-RoIAlign), [490:980] radar (RoIAlign)
  __shared__ float s_hid[RPB][HID];
  __shared__ float s_small[RPB][16];       // 0-3 reg, 4-5 cls logits(0,1), 6-15 radar conv
  __shared__ float s_roi[RPB][5];

  const int t = threadIdx.x;
  const int n_img = *d.n_img;
  const int total = n_img + d.n_radar;
  const int k0 = blockIdx.x * RPB;
  // slots behind the last RoI keep nothing (me_compact_sort_rows_f32 reads keep[0 : cap]; the caller need not clear it)
  if (t < RPB && k0 + t >= total && k0 + t < d.n_img_cap + d.n_radar) d.keep[k0 + t] = 0;
  if (k0 >= total) return;
  const int nr = (total - k0 < RPB) ? total - k0 : RPB;

  if (t < RPB * 5) {
    const int r = t / 5, c = t % 5;
    float v = 0.f;
    if (r < nr) {
      const int k = k0 + r;
      v = (k < n_img) ? d.img_boxes[(long long)k * d.box_cols + c] : d.radar_boxes[(long long)(k - n_img) * 5 + c];
    }
    s_roi[r][c] = v;
  }
  __syncthreads();

  // phase A: pooled features (sampled here, or read back from the pooling launch)
  if (d.pool_scratch) {
    // the nr rows are contiguous on both sides (2 * FEAT = 1960 floats = 490 float4 per RoI); four loads in flight per lane
    const float4* src = reinterpret_cast<const float4*>(d.pool_scratch + (long long)k0 * 2 * FEAT);
    float4* dst = reinterpret_cast<float4*>(&s_feat[0][0]);
    const int n4 = nr * (2 * FEAT / 4);
    for (int i0 = 0; i0 < n4; i0 += 1024) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256 + t;
        v[u] = i < n4 ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256 + t;
        if (i < n4) dst[i] = v[u];
      }
    }
  } else
  for (int idx = t; idx < nr * 2 * FEAT; idx += 256) {
    const int r = idx / (2 * FEAT), f = idx % (2 * FEAT);
    float v;
    if (f < FEAT) {
      const int pw = f % P, ph = (f / P) % P;
      v = ps_sample(d.img_map, d.img_pitch, d.fh, d.fw, s_roi[r], d.spatial_scale,
                    d.img_bin_major ? (ph * P + pw) * C_OUT + f / PP : f, ph, pw);
    } else {
      const int g = f - FEAT;
      const int pw = g % P, ph = (g / P) % P, c = g / PP;
      v = roi_sample(d.radar_map, d.radar_pitch, d.rh, d.rw, s_roi[r], d.spatial_scale, c, ph, pw);
    }
    s_feat[r][f] = v;
  }
  __syncthreads();

  // phase B: net0 (490 -> 256) + LeakyReLU; thread t = hidden unit t, all RoIs of the workgroup
  {
    float acc[RPB];
#pragma unroll
    for (int r = 0; r < RPB; ++r) acc[r] = 0.f;
    // four features per trip: one 16-byte broadcast read per RoI instead of four 4-byte ones (round 4: the loop issued eight
    // ds_read_b32 per FMA group and was bound by LDS instruction issue); the FMA chain of every (RoI, unit) keeps its k order.
    // The weights come in chunks of KC rows, the next chunk's loads in flight while this one is used: with four loads per trip
    // the lone workgroup of a CU (batch <= 8: fewer workgroups than CUs) waited one L2 latency per trip - ~50 us per launch
    // whatever the batch.
    // whatever the batch.  KC = 4 (many workgroups per CU: the other waves cover the latency, 97 VGPRs) skips the chunks.
    constexpr int NCH = KC >= 16 ? FEAT / KC : 0;  // 15 chunks of 32 = 480 rows, the last 10 rows below
    const float* wcol = d.wts.w0t + t;
    float wa[KC], wb[KC];
    auto fetch = [&](float* wreg, int c) {
#pragma unroll
      for (int u = 0; u < KC; ++u) wreg[u] = wcol[(c * KC + u) * HID];
    };
    auto use = [&](const float* wreg, int c) {
#pragma unroll
      for (int u = 0; u < KC; u += 4) {
#pragma unroll
        for (int r = 0; r < RPB; ++r) {
          const float4 f = *reinterpret_cast<const float4*>(&s_feat[r][c * KC + u]);
          acc[r] = fmaf(wreg[u + 3], f.w, fmaf(wreg[u + 2], f.z, fmaf(wreg[u + 1], f.y, fmaf(wreg[u], f.x, acc[r]))));
        }
      }
    };
    if constexpr (NCH > 0) {
      fetch(wa, 0);
      int c = 0;
      for (; c + 1 < NCH; c += 2) {
        fetch(wb, c + 1);
        use(wa, c);
        if (c + 2 < NCH) fetch(wa, c + 2);
        use(wb, c + 1);
      }
      if (c < NCH) use(wa, c);
    }
    int k = NCH * KC;
    for (; k + 4 <= FEAT; k += 4) {
      const float w0 = d.wts.w0t[(k + 0) * HID + t], w1 = d.wts.w0t[(k + 1) * HID + t];
      const float w2 = d.wts.w0t[(k + 2) * HID + t], w3 = d.wts.w0t[(k + 3) * HID + t];
#pragma unroll
      for (int r = 0; r < RPB; ++r) {
        const float4 f = *reinterpret_cast<const float4*>(&s_feat[r][k]);
        acc[r] = fmaf(w3, f.w, fmaf(w2, f.z, fmaf(w1, f.y, fmaf(w0, f.x, acc[r]))));
      }
    }
    for (; k < FEAT; ++k) {
      const float w = d.wts.w0t[k * HID + t];
#pragma unroll
      for (int r = 0; r < RPB; ++r) acc[r] = fmaf(w, s_feat[r][k], acc[r]);
    }
    const float b = d.wts.b0[t];
#pragma unroll
    for (int r = 0; r < RPB; ++r) s_hid[r][t] = leaky(acc[r] + b);
  }
  __syncthreads();

  // phase C: the small dot products: (r, j) with j < 6 -> net1 rows 0-3 / net2 rows 0-1 over the
  // hidden vector; 6 <= j < 16 -> radar_net 7x7 conv output j-6 over the pooled radar feature
  {
    const int r = t >> 5, j = t & 31;
    if (r < nr && j < 16) {
      float acc = 0.f;
      if (j < 6) {
        const float* wrow = (j < 4) ? d.wts.w1 + j * HID : d.wts.w2 + (j - 4) * HID;
#pragma unroll 32
        for (int k = 0; k < HID; ++k) acc = fmaf(wrow[k], s_hid[r][k], acc);  // (32 weight loads in flight per trip)
        acc += (j < 4) ? d.wts.b1[j] : d.wts.b2[j - 4];
      } else {
        const float* wrow = d.wts.rw + (j - 6) * FEAT;
#pragma unroll 35
        for (int k = 0; k < FEAT; ++k) acc = fmaf(wrow[k], s_feat[r][FEAT + k], acc);
      }
      s_small[r][j] = acc;
    }
  }
  __syncthreads();

  if (d.save_small) {  // training mode: keep what the backward pass needs, the tail runs separately
    for (int idx = t; idx < nr * 2 * FEAT; idx += 256) {
      const int r = idx / (2 * FEAT), f = idx % (2 * FEAT);
      if (f < FEAT) d.save_feat_img[(long long)(k0 + r) * FEAT + f] = s_feat[r][f];
      else d.save_feat_rad[(long long)(k0 + r) * FEAT + (f - FEAT)] = s_feat[r][f];
    }
    for (int idx = t; idx < nr * HID; idx += 256) d.save_hidden[(long long)(k0 + idx / HID) * HID + idx % HID] = s_hid[idx / HID][idx % HID];
    if (t < nr * 16) {
      const int r = t / 16, j = t % 16;
      d.save_small[(long long)(k0 + r) * 16 + j] = s_small[r][j] + (j >= 6 ? d.wts.rb[j - 6] : 0.f);
    }
    return;
  }

  // phase D: one thread per RoI - scalar tail
  if (t < nr) tail_one(d, s_small[t], s_roi[t], k0 + t, n_img, tail_weights(d));
}


// ---- the same heads on the matrix pipe (inference, pooled features in d.pool_scratch) ----------------------------------------
// roi_heads_kernel keeps a hidden unit per thread and reads every RoI's features by LDS broadcast (4 MB of LDS reads per
// workgroup of 8 RoIs), W0 (500 KB) streams from the L2 once per 8 RoIs, and the small dot products of phase C walk their
// weight rows with dependent-latency global loads: ~50 us per workgroup whatever the batch.  Here a workgroup owns RM = 32 RoIs:
// * net0 is the GEMM [32 x 490] x [490 x 256] on v_mfma_f32_32x32x2_f32 (wave v: units 64 v .. 64 v + 63 = two 32 x 32 tiles;
//   A = the features from LDS, one ds_read_b32 per step shared by both tiles; B = W0 rows straight from the L2, 128-byte rows,
//   three chunks of seven steps in rotation).  The MFMA accumulates k in order with fp32 FMAs, like the VALU chain, and W0 is
//   read once per 32 RoIs;
// * the weights of phase C (net1 / net2: 6 x 256, radar_net conv: 10 x 490) and of the ensemble head sit in LDS, fetched with
//   the first features; the chains read weights and features four k at a time (same k order);
// * the image features and - after net0 - the radar features share one LDS buffer.
constexpr int RM = 32;
constexpr int FPI = FEAT + 2;  // 492: rows 16-byte aligned
constexpr int HP = HID + 4;    // 260
constexpr int kHeadsMfmaFloats = RM * FPI + RM * HP + RM * 16 + RM * 5 + 256 + 6 * HID + C_OUT * FPI;
constexpr size_t kHeadsMfmaLds = (size_t)kHeadsMfmaFloats * sizeof(float);

__global__ __launch_bounds__(256) void roi_heads_mfma_kernel(me_heads_desc d, int stop) {
  extern __shared__ __attribute__((aligned(16))) float heads_lds[];
  float* s_f = heads_lds;            // [RM][FPI] image features, then radar features
  float* s_hid = s_f + RM * FPI;     // [RM][HP]
  float* s_small = s_hid + RM * HP;  // [RM][16]: 0-3 reg, 4-5 cls logits, 6-15 radar conv
  float* s_roi = s_small + RM * 16;  // [RM][5]
  float* s_ens = s_roi + RM * 5;     // [256] e1w [32][2], e1b [32], e2w [2][64], rw2 [10], rscale [10], rshift [10], rb2
  float* s_w12 = s_ens + 256;        // [6][HID] net1 rows 0-3, net2 rows 0-1
  float* s_rw = s_w12 + 6 * HID;     // [C_OUT][FPI] radar_net 7x7 conv
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int n_img = *d.n_img;
  const int total = n_img + d.n_radar;
  const int k0 = blockIdx.x * RM;
  if (t < RM && k0 + t >= total && k0 + t < d.n_img_cap + d.n_radar) d.keep[k0 + t] = 0;  // (see roi_heads_kernel)
  if (k0 >= total) return;
  const int nr = (total - k0 < RM) ? total - k0 : RM;
  if (t < RM * 5) {
    const int r = t / 5, c = t % 5;
    float v = 0.f;
    if (r < nr) {
      const int k = k0 + r;
      v = (k < n_img) ? d.img_boxes[(long long)k * d.box_cols + c] : d.radar_boxes[(long long)(k - n_img) * 5 + c];
    }
    s_roi[r * 5 + c] = v;
  }
  // pooled rows -> registers -> LDS (rows behind nr: zeros); part = 0 image half, 1 radar half of a pool_scratch row.  All 32 rows
  // = 64 loads per thread in flight at once (one wave per SIMD: registers are free)
  auto feats_issue = [&](float (*st)[2], int part) {
    const float* src = d.pool_scratch + (long long)k0 * 2 * FEAT + part * FEAT;
#pragma unroll
    for (int u = 0; u < RM; ++u)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int f = t + 256 * h;
        st[u][h] = (u < nr && f < FEAT) ? src[(long long)u * 2 * FEAT + f] : 0.f;
      }
  };
  auto feats_commit = [&](float (*st)[2]) {
#pragma unroll
    for (int u = 0; u < RM; ++u)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int f = t + 256 * h;
        if (f < FEAT) s_f[u * FPI + f] = st[u][h];
      }
  };
  float radar[RM][2];  // the radar rows wait in registers until net0 is done with the image rows
  {
    float img[RM][2];
    feats_issue(img, 0);
    // the small weights, in flight with the features
    float wst[26];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int i = t + 256 * u;  // < 1536
      wst[u] = i < 4 * HID ? d.wts.w1[i] : d.wts.w2[i - 4 * HID];
    }
#pragma unroll
    for (int u = 0; u < 20; ++u) {
      const int i = t + 256 * u;
      wst[6 + u] = i < C_OUT * FEAT ? d.wts.rw[i] : 0.f;
    }
    float ens = 0.f;
    if (t < 64) ens = d.wts.e1w[t];
    else if (t < 96) ens = d.wts.e1b[t - 64];
    else if (t < 224) ens = d.wts.e2w[t - 96];
    else if (t < 234) ens = d.wts.rw2[t - 224];
    else if (t < 244) ens = d.wts.rscale[t - 234];
    else if (t < 254) ens = d.wts.rshift[t - 244];
    else if (t == 254) ens = d.wts.rb2[0];
    feats_issue(radar, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 6; ++u) s_w12[t + 256 * u] = wst[u];
#pragma unroll
    for (int u = 0; u < 20; ++u) {
      const int i = t + 256 * u;
      if (i < C_OUT * FEAT) s_rw[(i / FEAT) * FPI + i % FEAT] = wst[6 + u];
    }
    s_ens[t] = ens;
    feats_commit(img);
  }
  __syncthreads();
  if (stop == 1) return;  // (tools/heads_phases.py: time of the launch up to here)

  // phase B: net0 on the matrix pipe
  {
    const int m = lane & 31, hh = lane >> 5;
    const float* arow = s_f + m * FPI + hh;                             // A[m][2 s + hh]
    const float* brow = d.wts.w0t + (long long)hh * HID + 64 * wv + m;  // B[2 s + hh][64 wv + m] (+ 32: second tile)
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;
    constexpr int KS = 7, NCH = (FEAT / 2) / KS;  // 245 steps = 35 chunks of 7
    float b0[KS][2], b1[KS][2], b2[KS][2], a0[KS], a1[KS], a2[KS];
    auto fetch = [&](float (*b)[2], float* a, int c) {  // both operands of chunk c: W0 rows from the L2, features from LDS
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        const float* p = brow + (long long)(c * KS + u) * 2 * HID;
        b[u][0] = p[0];
        b[u][1] = p[32];
      }
#pragma unroll
      for (int u = 0; u < KS; ++u) a[u] = arow[(c * KS + u) * 2];
      __builtin_amdgcn_sched_barrier(0);  // (keep the loads two chunks ahead of their MFMAs)
    };
    auto use = [&](float (*b)[2], const float* a) {
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][1], acc1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    fetch(b0, a0, 0);
    fetch(b1, a1, 1);
    int c = 0;
    for (; c + 2 < NCH; c += 3) {  // two chunks of loads in flight behind the one in use
      fetch(b2, a2, c + 2);
      use(b0, a0);
      if (c + 3 < NCH) fetch(b0, a0, c + 3);
      use(b1, a1);
      if (c + 4 < NCH) fetch(b1, a1, c + 4);
      use(b2, a2);
    }
    if (c < NCH) use(b0, a0);
    if (c + 1 < NCH) use(b1, a1);
    // register e of lane (m, hh) = RoI (e & 3) + 8 (e >> 2) + 4 hh, unit 64 wv + m (+ 32)
    const float bias0 = d.wts.b0[64 * wv + m], bias1 = d.wts.b0[64 * wv + 32 + m];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = (e & 3) + 8 * (e >> 2) + 4 * hh;
      s_hid[r * HP + 64 * wv + m] = leaky(acc0[e] + bias0);
      s_hid[r * HP + 64 * wv + 32 + m] = leaky(acc1[e] + bias1);
    }
  }
  __syncthreads();  // every wave is done with the image features; the hidden vectors are complete
  if (stop == 2) return;
  feats_commit(radar);

  // phase C: (RoI r = t >> 3, j): j < 6 -> net1 rows 0-3 / net2 rows 0-1 over the hidden vector; 6 <= j < 16 -> radar_net 7x7 conv
  // output j - 6 over the pooled radar feature: lane (t & 7) takes the hidden-vector row (t & 7) < 6, then radar outputs (t & 7)
  // and, for (t & 7) < 2, 8 + (t & 7).
  {
    const int r = t >> 3, jj = t & 7;
    if (r < nr && jj < 6) {
      const float4* wrow = reinterpret_cast<const float4*>(s_w12 + jj * HID);
      const float4* hrow = reinterpret_cast<const float4*>(s_hid + r * HP);
      float acc = 0.f;
#pragma unroll 8
      for (int k = 0; k < HID / 4; ++k) {
        const float4 w = wrow[k], h = hrow[k];
        acc = fmaf(w.w, h.w, fmaf(w.z, h.z, fmaf(w.y, h.y, fmaf(w.x, h.x, acc))));
      }
      s_small[r * 16 + jj] = acc + ((jj < 4) ? d.wts.b1[jj] : d.wts.b2[jj - 4]);
    }
    __syncthreads();  // the radar rows are in LDS
    if (stop == 3) return;
    if (r < nr) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int o = jj + 8 * half;  // radar conv output
        if (o < C_OUT) {
          const float4* wrow = reinterpret_cast<const float4*>(s_rw + o * FPI);
          const float4* frow = reinterpret_cast<const float4*>(s_f + r * FPI);
          float acc = 0.f;
#pragma unroll 8
          for (int k = 0; k < FEAT / 4; ++k) {  // 122 x 4 = 488
            const float4 w = wrow[k], f = frow[k];
            acc = fmaf(w.w, f.w, fmaf(w.z, f.z, fmaf(w.y, f.y, fmaf(w.x, f.x, acc))));
          }
          for (int k = FEAT / 4 * 4; k < FEAT; ++k) acc = fmaf(s_rw[o * FPI + k], s_f[r * FPI + k], acc);
          s_small[r * 16 + 6 + o] = acc;
        }
      }
    }
  }
  __syncthreads();
  if (stop == 4) return;
  if (d.save_small) {  // training mode (round 6): keep what the backward pass needs - the pooled features already sit in pool_scratch
    for (int idx = t; idx < nr * HID; idx += 256) d.save_hidden[(long long)(k0 + idx / HID) * HID + idx % HID] = s_hid[(idx / HID) * HP + idx % HID];
    for (int idx = t; idx < nr * 16; idx += 256) {
      const int r = idx / 16, j = idx % 16;
      d.save_small[(long long)(k0 + r) * 16 + j] = s_small[r * 16 + j] + (j >= 6 ? d.wts.rb[j - 6] : 0.f);
    }
    return;   // (the tail runs separately, once the RoI-wise BatchNorm statistics are known: me_heads_tail_f32)
  }
  // phase D: one thread per RoI - scalar tail (its weights from LDS)
  if (t < nr)
    tail_one(d, s_small + t * 16, s_roi + t * 5, k0 + t, n_img,
             TailW{s_ens, s_ens + 64, s_ens + 96, s_ens + 224, s_ens + 234, s_ens + 244, s_ens + 254});
}


// ---- module 2 (stage 2): PS-RoIAlign + refinement_head + ensemble_head over ALL classes -------------------------
// Reference module2_mixed/my_models.py:299-364: boxes [K, 8 + C] (every class, C = 12), refinement_head((490, 256, C+1))
// = net0 (490->256, LeakyReLU; Dropout is identity in eval), net1 (256->4), net2 (256->C+1, sigmoid);
// ensemble_head((2, 32, 32*(C+1), 2)): stack(refinement_vector, yolo_vector) -> fc1 (2->32, leaky) per class ->
// flatten -> fc2 (32*(C+1) -> 2, **leaky**) -> softmax; masks[:,1] is the new confidence; box_regress on the kept rows.
constexpr int M2_MAXC = 16;  // C + 1 <= 16

struct M2Desc {
  const float* img_map; long long img_pitch; int n, fh, fw; float spatial_scale;
  const float* boxes; const int* n_boxes; int box_cols, ncls1;  // ncls1 = C + 1
  const float *w0t, *b0, *w1, *b1, *w2, *b2, *e1w, *e1b, *e2w, *e2b;
  float thr;
  float *regress_out, *refine_out, *mask_out, *out_rows, *sort_key; unsigned char* keep;
  float* pool_scratch;  // [boxes_cap][FEAT] or NULL (see me_heads_desc.pool_scratch)
};

// PS-RoIAlign of one box per workgroup into d.pool_scratch
__global__ __launch_bounds__(256) void m2_pool_kernel(M2Desc d) {
  __shared__ float s_box[5];
  const int t = threadIdx.x, k = blockIdx.x;
  if (k >= *d.n_boxes) return;
  if (t < 5) s_box[t] = d.boxes[(long long)k * d.box_cols + t];
  __syncthreads();
  float* out = d.pool_scratch + (long long)k * FEAT;
  for (int f = t; f < FEAT; f += 256) {
    const int pw = f % P, ph = (f / P) % P;
    out[f] = ps_sample(d.img_map, d.img_pitch, d.fh, d.fw, s_box, d.spatial_scale, f, ph, pw);
  }
}

__global__ __launch_bounds__(256) void m2_heads_kernel(M2Desc d) {
  __shared__ __attribute__((aligned(16))) float s_feat[RPB][FEAT];
  __shared__ float s_hid[RPB][HID];
  __shared__ float s_small[RPB][4 + M2_MAXC];  // 0-3 reg, 4.. class logits
  __shared__ float s_roi[RPB][5];
  const int t = threadIdx.x;
  const int total = *d.n_boxes;
  const int k0 = blockIdx.x * RPB;
  if (k0 >= total) return;
  const int nr = (total - k0 < RPB) ? total - k0 : RPB;
  if (t < RPB * 5) {
    const int r = t / 5, c = t % 5;
    s_roi[r][c] = (r < nr) ? d.boxes[(long long)(k0 + r) * d.box_cols + c] : 0.f;
  }
  __syncthreads();
  if (d.pool_scratch) {  // pooled by m2_pool_kernel: nr rows of FEAT floats, contiguous on both sides (FEAT * 4 = 1960 bytes)
    const float2* src = reinterpret_cast<const float2*>(d.pool_scratch + (long long)k0 * FEAT);
    float2* dst = reinterpret_cast<float2*>(&s_feat[0][0]);
    const int n2 = nr * (FEAT / 2);
    for (int i0 = 0; i0 < n2; i0 += 1024) {
      float2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256 + t;
        v[u] = i < n2 ? src[i] : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256 + t;
        if (i < n2) dst[i] = v[u];
      }
    }
  } else
  for (int idx = t; idx < nr * FEAT; idx += 256) {
    const int r = idx / FEAT, f = idx % FEAT;
    const int pw = f % P, ph = (f / P) % P;
    s_feat[r][f] = ps_sample(d.img_map, d.img_pitch, d.fh, d.fw, s_roi[r], d.spatial_scale, f, ph, pw);
  }
  __syncthreads();
  {
    float acc[RPB];
#pragma unroll
    for (int r = 0; r < RPB; ++r) acc[r] = 0.f;
    for (int k = 0; k < FEAT; ++k) {
      const float w = d.w0t[k * HID + t];
#pragma unroll
      for (int r = 0; r < RPB; ++r) acc[r] = fmaf(w, s_feat[r][k], acc[r]);
    }
    const float b = d.b0[t];
#pragma unroll
    for (int r = 0; r < RPB; ++r) s_hid[r][t] = leaky(acc[r] + b);
  }
  __syncthreads();
  {
    const int r = t >> 5, j = t & 31;  // 32 lanes per RoI, 4 + ncls1 <= 20 dot products over the hidden vector
    if (r < nr && j < 4 + d.ncls1) {
      const float* wrow = (j < 4) ? d.w1 + j * HID : d.w2 + (j - 4) * HID;
      float acc = 0.f;
      for (int k = 0; k < HID; ++k) acc = fmaf(wrow[k], s_hid[r][k], acc);
      s_small[r][j] = acc + ((j < 4) ? d.b1[j] : d.b2[j - 4]);
    }
  }
  __syncthreads();
  if (t < nr) {
    const int k = k0 + t, nc = d.ncls1;
    const float* sm = s_small[t];
    const float* bx = d.boxes + (long long)k * d.box_cols;
    float o0 = d.e2b[0], o1 = d.e2b[1];
    for (int c = 0; c < nc; ++c) {
      const float rv = sigmoidf(sm[4 + c]);
      const float yv = (c == 0) ? bx[5] : bx[8 + (c - 1)];  // yolo_vector = (obj_conf, class scores)
      d.refine_out[(long long)k * nc + c] = rv;
      for (int u = 0; u < 32; ++u) {
        const float h = leaky(d.e1w[2 * u] * rv + d.e1w[2 * u + 1] * yv + d.e1b[u]);
        o0 += d.e2w[c * 32 + u] * h;
        o1 += d.e2w[nc * 32 + c * 32 + u] * h;
      }
    }
    o0 = leaky(o0);
    o1 = leaky(o1);
    const float m = fmaxf(o0, o1);
    const float e0 = expf(o0 - m), e1 = expf(o1 - m);
    const float p = e1 / (e0 + e1);  // masks[:, 1]
    for (int c = 0; c < 4; ++c) d.regress_out[4ll * k + c] = sm[c];
    d.mask_out[k] = p;
    d.keep[k] = (p > d.thr) ? 1 : 0;
    d.sort_key[k] = p;
    const float x1 = bx[1], y1 = bx[2], x2 = bx[3], y2 = bx[4];
    const float cx = (x1 + x2) / 2, cy = (y1 + y2) / 2, bw = x2 - x1, bh = y2 - y1;
    const float nx = sm[0] * bw + cx, ny = sm[1] * bh + cy;
    const float nw = expf(sm[2]) * bw, nh = expf(sm[3]) * bh;
    float* o = d.out_rows + 8ll * k;
    o[0] = bx[0]; o[1] = nx - nw / 2; o[2] = ny - nh / 2; o[3] = nx + nw / 2; o[4] = ny + nh / 2;
    o[5] = p; o[6] = bx[6]; o[7] = bx[7];
  }
}

// ---- training: stand-alone tail, loss terms, tail backward -----------------------------------------
__global__ __launch_bounds__(256) void heads_tail_kernel(me_heads_desc d, const float* small, int k) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= k) return;
  const int n_img = *d.n_img;
  float roi[5];
  for (int c = 0; c < 5; ++c)
    roi[c] = (i < n_img) ? d.img_boxes[(long long)i * d.box_cols + c] : d.radar_boxes[(long long)(i - n_img) * 5 + c];
  tail_one(d, small + 16ll * i, roi, i, n_img, tail_weights(d));
}

__global__ __launch_bounds__(256) void heads_loss_kernel(const float* mask1, const float* refine,
                                                         const uint8_t* label_pos, const uint8_t* in_focal,
                                                         const uint8_t* in_conf, int k, float alpha, float lam,
                                                         float* terms, float* seed_p, float* seed_conf) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= k) return;
  const bool pos = label_pos[i] != 0;
  float focal = 0.f, dp = 0.f;
  if (in_focal[i]) {  // FocalLoss(alpha, gamma=2, "sum") on [1-p, p] vs one-hot (my_models.py:287-314, 610)
    const float p = mask1[i];
    const float pt = pos ? p : 1.f - p;
    const float a = pos ? alpha : 1.f - alpha;
    const float lg = logf(pt);
    focal = -a * (1.f - pt) * (1.f - pt) * lg;
    const float dpt = a * (2.f * (1.f - pt) * lg - (1.f - pt) * (1.f - pt) / pt);
    dp = pos ? dpt : -dpt;
  }
  float bce = 0.f, dc = 0.f;
  if (in_conf[i]) {  // nn.BCELoss(reduction="sum") / lambda (my_models.py:614-619, 635); log clamped at -100
    const float x = refine[2 * i];
    const float y = pos ? 1.f : 0.f;
    const float l1 = fmaxf(logf(x), -100.f), l0 = fmaxf(logf(1.f - x), -100.f);
    bce = -(y * l1 + (1.f - y) * l0) / lam;
    dc = (x - y) / fmaxf((1.f - x) * x, 1e-12f) / lam;
  }
  terms[2 * i] = focal;
  terms[2 * i + 1] = bce;
  seed_p[i] = dp;
  seed_conf[i] = dc;
}

__global__ __launch_bounds__(256) void heads_tail_bwd_kernel(me_heads_desc d, const float* small, const float* refine,
                                                             const float* mask1, const float* seed_p,
                                                             const float* seed_conf, int k, float* g_o, float* g_hpre,
                                                             float* h_act, float* xin, float* g_z2, float* g_rl,
                                                             float* rl_out, float* g_rlogit, const int* k_dev) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= k) return;
  if (k_dev && i >= *k_dev) {   // captured step (k = the buffers' capacity, the live row count in device memory): zeros behind it
    for (int u = 0; u < 64; ++u) { g_hpre[64ll * i + u] = 0.f; h_act[64ll * i + u] = 0.f; }
    xin[4 * i + 0] = xin[4 * i + 1] = xin[4 * i + 2] = xin[4 * i + 3] = 0.f;
    g_o[2 * i] = g_o[2 * i + 1] = g_z2[2 * i] = g_z2[2 * i + 1] = g_rlogit[i] = 0.f;
    for (int o = 0; o < C_OUT; ++o) { g_rl[10ll * i + o] = 0.f; rl_out[10ll * i + o] = 0.f; }
    return;
  }
  const int n_img = *d.n_img;
  const float* sm = small + 16ll * i;
  const float conf = refine[2 * i], cls1 = refine[2 * i + 1];
  const float cls0 = sigmoidf(sm[4]);
  float rl[C_OUT], rpre[C_OUT];
  float rad = d.wts.rb2[0];
#pragma unroll
  for (int o = 0; o < C_OUT; ++o) {
    rpre[o] = sm[6 + o] * d.wts.rscale[o] + d.wts.rshift[o];
    rl[o] = leaky(rpre[o]);
    rad = fmaf(d.wts.rw2[o], rl[o], rad);
  }
  const float rconf = sigmoidf(rad);
  float d_conf = seed_conf[i], d_cls1 = 0.f;
  float go0 = 0.f, go1 = 0.f;
  float* gh = g_hpre + 64ll * i;
  float* ha = h_act + 64ll * i;
  if (i < n_img) {
    const float* bx = d.img_boxes + (long long)i * d.box_cols;
    const float yolo0 = bx[5], yolo1 = bx[8];
    const float p = mask1[i];
    // p = softmax(o)[0]: dp/do0 = p (1 - p), dp/do1 = -p (1 - p)
    go0 = seed_p[i] * p * (1.f - p);
    go1 = -go0;
    float d_r0 = 0.f, d_r1 = 0.f;
    for (int u = 0; u < 32; ++u) {
      const float wa = d.wts.e1w[2 * u], wb = d.wts.e1w[2 * u + 1], bb = d.wts.e1b[u];
      const float pre0 = wa * conf + wb * yolo0 + bb, pre1 = wa * cls1 + wb * yolo1 + bb;
      ha[u] = leaky(pre0);
      ha[32 + u] = leaky(pre1);
      const float dh0 = d.wts.e2w[u] * go0 + d.wts.e2w[64 + u] * go1;
      const float dh1 = d.wts.e2w[32 + u] * go0 + d.wts.e2w[96 + u] * go1;
      const float dp0 = pre0 > 0.f ? dh0 : 0.1f * dh0, dp1 = pre1 > 0.f ? dh1 : 0.1f * dh1;
      gh[u] = dp0;
      gh[32 + u] = dp1;
      d_r0 += dp0 * wa;
      d_r1 += dp1 * wa;
    }
    d_conf += d_r0;
    d_cls1 = d_r1;
    xin[4 * i + 0] = conf; xin[4 * i + 1] = yolo0; xin[4 * i + 2] = cls1; xin[4 * i + 3] = yolo1;
  } else {
    for (int u = 0; u < 64; ++u) { gh[u] = 0.f; ha[u] = 0.f; }
    xin[4 * i + 0] = xin[4 * i + 1] = xin[4 * i + 2] = xin[4 * i + 3] = 0.f;
  }
  g_o[2 * i] = go0;
  g_o[2 * i + 1] = go1;
  const float d_s = d_conf * conf * (1.f - conf);  // conf = sigmoid(rconf + cls0)
  g_z2[2 * i] = d_s * cls0 * (1.f - cls0);
  g_z2[2 * i + 1] = d_cls1 * cls1 * (1.f - cls1);
  const float d_rlogit = d_s * rconf * (1.f - rconf);
  g_rlogit[i] = d_rlogit;
#pragma unroll
  for (int o = 0; o < C_OUT; ++o) {
    g_rl[10ll * i + o] = d_rlogit * d.wts.rw2[o];
    rl_out[10ll * i + o] = rl[o];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// output tail: compaction + stable descending sort by rank counting (one launch; replaces nonzero + sort + gather)
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSortTile = 2048;
constexpr int kSortRows = 16;  // rows per workgroup; 16 lanes share one row and split the keys between them

// One workgroup = 16 rows x 16 key partitions (partition p takes the keys j with j % 16 == p: the 16 lanes of a row read 16
// consecutive LDS words, the 4 rows of a wave broadcast).  A 6464-row batch-32 step is 404 workgroups / 1616 waves instead
// of one thread per row looping over every key (that version was VALU-bound on 26 CUs: 208 us; this one 15 us).
__global__ __launch_bounds__(256) void compact_sort_kernel(const float* __restrict__ rows, const unsigned char* __restrict__ keep,
                                                           const float* __restrict__ key, int cap, int cols,
                                                           float* __restrict__ out, int* __restrict__ count) {
  __shared__ float sk[kSortTile];
  const int part = threadIdx.x & 15;
  const int i = blockIdx.x * kSortRows + (threadIdx.x >> 4);
  const bool mine = i < cap && keep[i] != 0;
  const float ki = mine ? key[i] : 0.f;
  int rank = 0, total = 0;
  for (int t0 = 0; t0 < cap; t0 += kSortTile) {
    __syncthreads();
    for (int j = threadIdx.x; j < kSortTile; j += 256) {
      const int g = t0 + j;
      sk[j] = (g < cap && keep[g] != 0) ? key[g] : __builtin_nanf("");  // dropped rows never compare greater / equal
    }
    __syncthreads();
    const int lim = (cap - t0 < kSortTile ? cap - t0 : kSortTile);
    for (int k = part; k < lim; k += 16) {
      const float kj = sk[k];
      rank += (kj > ki) || (kj == ki && t0 + k < i);
      total += kj == kj;
    }
  }
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) {
    rank += __shfl_xor(rank, o, 16);
    total += __shfl_xor(total, o, 16);
  }
  if (mine)
    for (int c = part; c < cols; c += 16) out[(long long)rank * cols + c] = rows[(long long)i * cols + c];
  // (system scope: the count word may be pinned host memory that the caller polls)
  if (i == 0 && part == 0) __hip_atomic_store(count, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

extern "C" {

int me_gather_class_boxes_f32(const float* det, const int32_t* count, int32_t n, int32_t max_det,
                              int32_t num_classes, int32_t class_idx, int32_t class_num, float* boxes,
                              int32_t* total, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(det && count && boxes && total, ME_E_NULLPTR, "me_gather_class_boxes_f32: null pointer");
  ME_REQUIRE(n > 0 && max_det > 0 && num_classes >= 1 && class_num >= 0 && class_num <= num_classes, ME_E_BADARG,
             "me_gather_class_boxes_f32: bad dimensions");  // class_idx < 0 gathers every class
  hipLaunchKernelGGL(gather_class_boxes_kernel, dim3(n), dim3(256), 0, stream, det, count, n, max_det, num_classes,
                     class_idx, class_num, boxes, total);
  return me::check_launch("gather_class_boxes_kernel");
}

int me_m2_heads_f32(const float* img_map, int64_t img_pitch, int32_t n, int32_t fh, int32_t fw, float spatial_scale,
                    const float* boxes, const int32_t* n_boxes, int32_t boxes_cap, int32_t box_cols, int32_t class_num,
                    const me_heads_weights* w, float refine_threshold, float* regress_out, float* refine_out,
                    float* mask_out, float* out_rows, uint8_t* keep, float* sort_key, float* pool_scratch, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (boxes_cap == 0) return 0;
  ME_REQUIRE(img_map && boxes && n_boxes && w && regress_out && refine_out && mask_out && out_rows && keep && sort_key,
             ME_E_NULLPTR, "me_m2_heads_f32: null pointer");
  ME_REQUIRE(class_num >= 1 && class_num + 1 <= M2_MAXC && box_cols >= 8 + class_num, ME_E_BADARG,
             "me_m2_heads_f32: class_num %d / box_cols %d", class_num, box_cols);
  ME_REQUIRE(w->w0t && w->b0 && w->w1 && w->b1 && w->w2 && w->b2 && w->e1w && w->e1b && w->e2w && w->e2b, ME_E_NULLPTR,
             "me_m2_heads_f32: null weight pointer");
  M2Desc d{img_map, img_pitch, n, fh, fw, spatial_scale, boxes, n_boxes, box_cols, class_num + 1,
           w->w0t, w->b0, w->w1, w->b1, w->w2, w->b2, w->e1w, w->e1b, w->e2w, w->e2b, refine_threshold,
           regress_out, refine_out, mask_out, out_rows, sort_key, keep, pool_scratch};
  const int blocks = (boxes_cap + RPB - 1) / RPB;
  if (pool_scratch) {
    ME_REQUIRE(me::aligned16(pool_scratch), ME_E_ALIGN, "me_m2_heads_f32: pool_scratch must be 16-byte aligned");
    hipLaunchKernelGGL(m2_pool_kernel, dim3(boxes_cap), dim3(256), 0, stream, d);
    const int rc = me::check_launch("m2_pool_kernel");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(m2_heads_kernel, dim3(blocks), dim3(256), 0, stream, d);
  return me::check_launch("m2_heads_kernel");
}

int me_roi_heads_f32(const me_heads_desc* d, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(d, ME_E_NULLPTR, "me_roi_heads_f32: null descriptor");
  ME_REQUIRE(d->img_map && d->radar_map && d->img_boxes && d->n_img, ME_E_NULLPTR, "me_roi_heads_f32: null input");
  ME_REQUIRE(d->n_radar == 0 || d->radar_boxes, ME_E_NULLPTR, "me_roi_heads_f32: null radar_boxes");
  ME_REQUIRE(d->regress_out && d->refine_out && d->mask1_out && d->out_rows && d->keep && d->sort_key, ME_E_NULLPTR,
             "me_roi_heads_f32: null output");
  const me_heads_weights& w = d->wts;
  ME_REQUIRE(w.w0t && w.b0 && w.w1 && w.b1 && w.w2 && w.b2 && w.rw && w.rscale && w.rshift && w.rw2 && w.rb2 &&
                 w.e1w && w.e1b && w.e2w && w.e2b,
             ME_E_NULLPTR, "me_roi_heads_f32: null weight pointer");
  ME_REQUIRE(d->n > 0 && d->fh > 0 && d->fw > 0 && d->rh > 0 && d->rw > 0 && d->n_img_cap >= 0 && d->n_radar >= 0 &&
                 d->box_cols >= 9,
             ME_E_BADARG, "me_roi_heads_f32: bad dimensions");
  ME_REQUIRE(d->img_pitch >= FEAT && d->radar_pitch >= C_OUT, ME_E_BADARG, "me_roi_heads_f32: map pitch too small");
  const bool train = d->save_small || d->save_feat_img || d->save_feat_rad || d->save_hidden;
  // (with pool_scratch the pooled features ARE the feature saves: [cap][980] = image half | radar half, pitch 980)
  ME_REQUIRE(!train || (d->save_small && d->save_hidden && w.rb && (d->pool_scratch || (d->save_feat_img && d->save_feat_rad))), ME_E_NULLPTR,
             "me_roi_heads_f32: training mode needs save_small, save_hidden, wts.rb and either pool_scratch or both save_feat_* pointers");
  const int cap = d->n_img_cap + d->n_radar;
  if (cap == 0) return 0;
  if (d->pool_scratch) {
    static const int pool10 = getenv("MILLIEYE_POOL10") ? atoi(getenv("MILLIEYE_POOL10")) : 1;
    if (pool10) hipLaunchKernelGGL(roi_pool10_kernel, dim3(cap), dim3(128), 0, stream, *d);
    else hipLaunchKernelGGL(roi_pool_kernel, dim3(cap), dim3(256), 0, stream, *d);
    const int rc = me::check_launch("roi_pool_kernel");
    if (rc) return rc;
  }
  static const int mfma_env = getenv("MILLIEYE_HEADS_MFMA") ? atoi(getenv("MILLIEYE_HEADS_MFMA")) : 1;
  if (mfma_env && d->pool_scratch) {  // pooled rows (inference, and training since round 6): net0 on the matrix pipe, 32 RoIs per workgroup
    static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(roi_heads_mfma_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kHeadsMfmaLds);
    ME_HIP(attr);
    static const int stop = getenv("MILLIEYE_HEADS_STOP") ? atoi(getenv("MILLIEYE_HEADS_STOP")) : 0;  // profiling only
    hipLaunchKernelGGL(roi_heads_mfma_kernel, dim3((cap + RM - 1) / RM), dim3(256), kHeadsMfmaLds, stream, *d, stop);
    return me::check_launch("roi_heads_mfma_kernel");
  }
  // few workgroups (batch <= 16: at most two per CU): deep weight prefetch, 189 VGPRs; more: the four-row loop keeps four waves per SIMD
  const int wgs = (cap + RPB - 1) / RPB;
  static const int kc_env = getenv("MILLIEYE_HEADS_KC") ? atoi(getenv("MILLIEYE_HEADS_KC")) : 0;
  if (kc_env ? kc_env == 32 : wgs <= 512) hipLaunchKernelGGL(roi_heads_kernel<32>, dim3(wgs), dim3(256), 0, stream, *d);
  else hipLaunchKernelGGL(roi_heads_kernel<4>, dim3(wgs), dim3(256), 0, stream, *d);
  return me::check_launch("roi_heads_kernel");
}

static int launch_roi(const float* map, int64_t pitch, int32_t n, int32_t h, int32_t w, int32_t c, const float* rois,
                      int32_t k, int32_t pooled, float scale, float* out, void* stream_, int ps) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(map && (out || k == 0) && (rois || k == 0), ME_E_NULLPTR, "me_roi_align_f32: null pointer");
  ME_REQUIRE(pooled == P, ME_E_BADARG, "me_roi_align_f32: only 7x7 pooling is built (my_models.py:495-496)");
  ME_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && pitch >= c && k >= 0, ME_E_BADARG, "me_roi_align_f32: bad dims");
  ME_REQUIRE(!ps || c % PP == 0, ME_E_BADARG, "me_ps_roi_align_f32: channels %% 49 != 0");
  if (k == 0) return 0;
  const long long total = (long long)k * c * PP / (ps ? PP : 1);
  long long blocks = (total + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(roi_align_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, map, (long long)pitch, h, w, c,
                     rois, k, scale, out, ps);
  return me::check_launch("roi_align_kernel");
}

int me_roi_align_f32(const float* map, int64_t pitch, int32_t n, int32_t h, int32_t w, int32_t c, const float* rois,
                     int32_t k, int32_t pooled, float spatial_scale, float* out, void* stream) {
  return launch_roi(map, pitch, n, h, w, c, rois, k, pooled, spatial_scale, out, stream, 0);
}

int me_ps_roi_align_f32(const float* map, int64_t pitch, int32_t n, int32_t h, int32_t w, int32_t c,
                        const float* rois, int32_t k, int32_t pooled, float spatial_scale, float* out, void* stream) {
  return launch_roi(map, pitch, n, h, w, c, rois, k, pooled, spatial_scale, out, stream, 1);
}

int me_heads_tail_f32(const me_heads_desc* d, const float* small, int32_t k, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(d && small && d->n_img && d->img_boxes, ME_E_NULLPTR, "me_heads_tail_f32: null pointer");
  ME_REQUIRE(d->regress_out && d->refine_out && d->mask1_out && d->out_rows && d->keep && d->sort_key, ME_E_NULLPTR,
             "me_heads_tail_f32: null output");
  ME_REQUIRE(k >= 0, ME_E_BADARG, "me_heads_tail_f32: negative k");
  if (k == 0) return 0;
  hipLaunchKernelGGL(heads_tail_kernel, dim3((k + 255) / 256), dim3(256), 0, stream, *d, small, k);
  return me::check_launch("heads_tail_kernel");
}

int me_heads_loss_f32(const float* mask1, const float* refine, const uint8_t* label_pos, const uint8_t* in_focal,
                      const uint8_t* in_conf, int32_t k, float alpha, float conf_lambda, float* terms, float* seed_p,
                      float* seed_conf, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(k >= 0, ME_E_BADARG, "me_heads_loss_f32: negative k");
  if (k == 0) return 0;
  ME_REQUIRE(mask1 && refine && label_pos && in_focal && in_conf && terms && seed_p && seed_conf, ME_E_NULLPTR,
             "me_heads_loss_f32: null pointer");
  hipLaunchKernelGGL(heads_loss_kernel, dim3((k + 255) / 256), dim3(256), 0, stream, mask1, refine, label_pos,
                     in_focal, in_conf, k, alpha, conf_lambda, terms, seed_p, seed_conf);
  return me::check_launch("heads_loss_kernel");
}

static int launch_heads_tail_bwd(const me_heads_desc* d, const float* small, const float* refine, const float* mask1,
                                 const float* seed_p, const float* seed_conf, int32_t k, const int32_t* k_dev, float* g_o,
                                 float* g_hpre, float* h_act, float* xin, float* g_z2, float* g_rl, float* rl, float* g_rlogit,
                                 void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(k >= 0, ME_E_BADARG, "me_heads_tail_bwd_f32: negative k");
  if (k == 0) return 0;
  ME_REQUIRE(d && small && refine && mask1 && seed_p && seed_conf && g_o && g_hpre && h_act && xin && g_z2 && g_rl &&
                 rl && g_rlogit && d->n_img && d->img_boxes,
             ME_E_NULLPTR, "me_heads_tail_bwd_f32: null pointer");
  hipLaunchKernelGGL(heads_tail_bwd_kernel, dim3((k + 255) / 256), dim3(256), 0, stream, *d, small, refine, mask1,
                     seed_p, seed_conf, k, g_o, g_hpre, h_act, xin, g_z2, g_rl, rl, g_rlogit, k_dev);
  return me::check_launch("heads_tail_bwd_kernel");
}

int me_heads_tail_bwd_f32(const me_heads_desc* d, const float* small, const float* refine, const float* mask1,
                          const float* seed_p, const float* seed_conf, int32_t k, float* g_o, float* g_hpre,
                          float* h_act, float* xin, float* g_z2, float* g_rl, float* rl, float* g_rlogit,
                          void* stream) {
  return launch_heads_tail_bwd(d, small, refine, mask1, seed_p, seed_conf, k, nullptr, g_o, g_hpre, h_act, xin, g_z2, g_rl, rl,
                               g_rlogit, stream);
}

int me_heads_tail_bwd_dev_f32(const me_heads_desc* d, const float* small, const float* refine, const float* mask1,
                              const float* seed_p, const float* seed_conf, int32_t cap, const int32_t* k_dev, float* g_o,
                              float* g_hpre, float* h_act, float* xin, float* g_z2, float* g_rl, float* rl, float* g_rlogit,
                              void* stream) {
  ME_REQUIRE(k_dev != nullptr, ME_E_NULLPTR, "me_heads_tail_bwd_dev_f32: null row count");
  return launch_heads_tail_bwd(d, small, refine, mask1, seed_p, seed_conf, cap, k_dev, g_o, g_hpre, h_act, xin, g_z2, g_rl, rl,
                               g_rlogit, stream);
}

int me_compact_sort_rows_f32(const float* rows, const uint8_t* keep, const float* key, int32_t cap, int32_t cols, float* out,
                             int32_t* count, void* stream) {
  ME_REQUIRE(count != nullptr, ME_E_NULLPTR, "me_compact_sort_rows_f32: null count");
  ME_REQUIRE(cap >= 0 && cols > 0, ME_E_BADARG, "me_compact_sort_rows_f32: bad dimensions");
  if (cap == 0) {
    ME_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), reinterpret_cast<hipStream_t>(stream)));
    return 0;
  }
  ME_REQUIRE(rows && keep && key && out, ME_E_NULLPTR, "me_compact_sort_rows_f32: null pointer");
  hipLaunchKernelGGL(compact_sort_kernel, dim3((unsigned)((cap + kSortRows - 1) / kSortRows)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), rows, keep, key, cap, cols, out, count);
  return me::check_launch("compact_sort_kernel");
}

}  // extern "C"

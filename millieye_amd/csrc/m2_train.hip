// Stage-2 (module2_mixed) training building blocks: small dense layers of the refinement / ensemble heads and the
// stage-2 objective.  Reference: module2_mixed/my_models.py:96-164 (heads), :366-459 (losses).  Everything here is tiny
// next to the detector (K <= 200 * N RoIs): one thread per output element, fixed-order sums (bit-reproducible).
#include <math.h>
#include <stdlib.h>
#include "common.h"

#pragma clang fp contract(off)

namespace {

inline unsigned grid_for(long long work) {
  long long b = (work + 255) / 256;
  if (b > 65535) b = 65535;
  if (b < 1) b = 1;
  return (unsigned)b;
}

__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == ME_ACT_LEAKY) return v > 0.f ? v : 0.1f * v;
  if (act == ME_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  return v;
}

// y[r][o] = act(sum_i x[r][i] * w[o][i] + b[o])  (nn.Linear + activation)
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ X, long long ldx, long long rows, int in_f,
                                                     const float* __restrict__ W, const float* __restrict__ B, int out_f,
                                                     int act, float* Y, long long ldy) {
  const long long total = rows * out_f;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int o = (int)(idx % out_f);
    const long long r = idx / out_f;
    const float* x = X + r * ldx;
    const float* w = W + (long long)o * in_f;
    float acc = 0.f;
    for (int i = 0; i < in_f; ++i) acc = fmaf(x[i], w[i], acc);
    Y[r * ldy + o] = act_fwd(acc + (B ? B[o] : 0.f), act);
  }
}

// y = x * mask * scale  (nn.Dropout in training mode and its backward)
__global__ __launch_bounds__(256) void mask_scale_kernel(const float* __restrict__ X, const unsigned char* __restrict__ M,
                                                         float scale, long long count, float* Y) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256)
    Y[i] = M[i] ? X[i] * scale : 0.f;
}

// The same nn.Linear through LDS tiles (round 6).  linear_kernel above reads W with a stride of in_f floats across the lanes of a
// wave (64 cache lines per load) and walks one dependent chain per thread: 400 us for the [1600 x 490] x [490 x 256] layer of a
// batch-8 step, 626 us per step over the five layers.  Here a workgroup stages a TM x 16 tile of X and a 16 x TN tile of W^T in
// LDS and every thread owns RM x 4 outputs; each output is still ONE fmaf chain over i = 0 .. in_f - 1 in ascending order, so the
// results equal linear_kernel's bit for bit.  TM x TN = 64 x 64 (4 x 4 outputs per thread) for the wide layers, 64 x 16 (1 x 4) for
// out_f <= 16 - the narrow layers have few outputs to spread, so small row tiles keep the launch wide.
template <int TM, int TN, int RM>
__global__ __launch_bounds__(256) void linear_tiled_kernel(const float* __restrict__ X, long long ldx, long long rows, int in_f,
                                                           const float* __restrict__ W, const float* __restrict__ B, int out_f,
                                                           int act, float* Y, long long ldy) {
  static_assert((TM / RM) * (TN / 4) == 256, "RM x 4 outputs per thread");
  __shared__ float Xs[16][TM + 1];
  __shared__ float Ws[16][TN + 1];
  constexpr int NX = TN / 4;   // threads along the outputs
  const int tx = threadIdx.x % NX, ty = threadIdx.x / NX;
  const long long m0 = (long long)blockIdx.y * TM;
  const int n0 = blockIdx.x * TN;
  float acc[RM][4];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  // the loads are unconditional (clamped addresses, the value dropped afterwards: behind a branch each of them waited for its own
  // round trip - 2.3 us per K step) and run one K step ahead of the arithmetic (registers -> LDS behind the barrier)
  float xv[16 * TM / 256], wv[(16 * TN + 255) / 256];
  auto fetch = [&](int k0) {
    const int kn = in_f - k0 < 16 ? in_f - k0 : 16;
#pragma unroll
    for (int u = 0; u < 16 * TM / 256; ++u) {   // X[m][k]: k fastest (16 consecutive floats of a row)
      const int idx = threadIdx.x + 256 * u, kk = idx % 16, mm = idx / 16;
      const long long m = m0 + mm;
      xv[u] = X[(m < rows ? m : rows - 1) * ldx + k0 + (kk < kn ? kk : kn - 1)];
    }
#pragma unroll
    for (int u = 0; u < (16 * TN + 255) / 256; ++u) {   // W[n][k]
      const int idx = threadIdx.x + 256 * u, kk = idx % 16, nn = (idx / 16) % TN;
      const int n = n0 + nn;
      wv[u] = W[(long long)(n < out_f ? n : out_f - 1) * in_f + k0 + (kk < kn ? kk : kn - 1)];
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < in_f; k0 += 16) {
    const int kn = in_f - k0 < 16 ? in_f - k0 : 16;
#pragma unroll
    for (int u = 0; u < 16 * TM / 256; ++u) {
      const int idx = threadIdx.x + 256 * u;
      Xs[idx % 16][idx / 16] = xv[u];
    }
#pragma unroll
    for (int u = 0; u < (16 * TN + 255) / 256; ++u) {
      const int idx = threadIdx.x + 256 * u;
      if (idx < 16 * TN) Ws[idx % 16][idx / 16] = wv[u];
    }
    __syncthreads();
    if (k0 + 16 < in_f) fetch(k0 + 16);
    // (the tail step is NOT padded with zero terms: fmaf(0, 0, -0.f) would flip a negative zero)
    auto fma_step = [&](int kk) {
      float a[RM], b[4];
#pragma unroll
      for (int i = 0; i < RM; ++i) a[i] = Xs[kk][ty * RM + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Ws[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    };
    if (kn == 16) {
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) fma_step(kk);
    } else {
      for (int kk = 0; kk < kn; ++kk) fma_step(kk);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long m = m0 + ty * RM + i;
      const int n = n0 + tx * 4 + j;
      if (m < rows && n < out_f) Y[m * ldy + n] = act_fwd(acc[i][j] + (B ? B[n] : 0.f), act);
    }
}

// nn.Dropout's keep mask on the device: Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the
// generator of torch's CUDA dropout too) keyed by a 64-bit seed the caller draws per step, counter = element quad index; one 32-bit
// word per element, keep iff its top 24 bits / 2^24 < keep_prob.
__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0, unsigned k1) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
  const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(unsigned long long seed, float keep_prob, long long count,
                                                           unsigned char* mask) {
  const long long quads = (count + 3) / 4;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < quads; q += (long long)gridDim.x * 256) {
    unsigned c0 = (unsigned)q, c1 = (unsigned)((unsigned long long)q >> 32), c2 = 0u, c3 = 0u;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c0, c1, c2, c3, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    const unsigned w[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long i = 4 * q + j;
      if (i < count) mask[i] = ((float)(w[j] >> 8) * (1.0f / 16777216.0f) < keep_prob) ? 1 : 0;
    }
  }
}

// Per-RoI terms of the stage-2 loss and their gradients (my_models.py:411-456).
//   o [K,2]       ensemble fc2 output (after its LeakyReLU); masks = softmax(o)
//   refine [K,C1] refinement_vector (sigmoid outputs), regress [K,4]
//   roi = boxes[:, 1:5]; target_location [K,4]; class_label [K,C1-1] (built on the host, with the reference's row quirk)
//   pos / sample [K] u8
// terms [K,5] = (focal, conf_bce, category_bce, smooth_l1_xy, smooth_l1_wh) - unscaled sums per RoI;
// d_o [K,2], d_refine [K,C1], d_regress [K,4] = gradients of
//   loss = focal + (conf + category) / lambda0 + (xy + wh) / lambda1   times gscale.
__global__ __launch_bounds__(256) void m2_loss_kernel(const float* __restrict__ O, const float* __restrict__ R, int c1,
                                                      const float* __restrict__ REG, const float* __restrict__ boxes,
                                                      int box_cols, const float* __restrict__ tloc,
                                                      const float* __restrict__ clabel, const unsigned char* __restrict__ pos,
                                                      const unsigned char* __restrict__ sample, int k, float alpha,
                                                      float lambda0, float lambda1, float gscale, float* terms, float* d_o,
                                                      float* d_r, float* d_reg) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= k) return;
  const bool is_pos = pos[i] != 0, in_s = sample[i] != 0;
  float t_focal = 0.f, t_conf = 0.f, t_cat = 0.f, t_xy = 0.f, t_wh = 0.f;
  float go0 = 0.f, go1 = 0.f;
  for (int c = 0; c < c1; ++c) d_r[(long long)i * c1 + c] = 0.f;
  for (int c = 0; c < 4; ++c) d_reg[4ll * i + c] = 0.f;
  if (in_s) {
    // softmax over the two logits
    const float o0 = O[2ll * i], o1 = O[2ll * i + 1];
    const float m = fmaxf(o0, o1);
    const float e0 = expf(o0 - m), e1 = expf(o1 - m);
    const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
    const float p = is_pos ? p1 : p0;
    const float a = is_pos ? alpha : 1.f - alpha;
    const float lp = logf(p);
    t_focal = -a * ((1.f - p) * (1.f - p)) * lp;
    const float dp = -a * (-2.f * (1.f - p) * lp + (1.f - p) * (1.f - p) / p);  // d focal / d p
    // p = softmax component c*: dp/do_j = p (delta - p_j)
    if (is_pos) {
      go1 = dp * p1 * (1.f - p1);
      go0 = dp * (-p1 * p0);
    } else {
      go0 = dp * p0 * (1.f - p0);
      go1 = dp * (-p0 * p1);
    }
    // confidence BCE (sum) on refinement_vector[:, 0]
    const float r0 = R[(long long)i * c1];
    const float y = is_pos ? 1.f : 0.f;
    t_conf = -(y * fmaxf(logf(r0), -100.f) + (1.f - y) * fmaxf(logf(1.f - r0), -100.f));
    d_r[(long long)i * c1] = gscale * ((r0 - y) / fmaxf(r0 * (1.f - r0), 1e-12f)) / lambda0;
  }
  if (is_pos) {
    for (int c = 1; c < c1; ++c) {
      const float r = R[(long long)i * c1 + c], y = clabel[(long long)i * (c1 - 1) + (c - 1)];
      t_cat += -(y * fmaxf(logf(r), -100.f) + (1.f - y) * fmaxf(logf(1.f - r), -100.f));
      d_r[(long long)i * c1 + c] = gscale * ((r - y) / fmaxf(r * (1.f - r), 1e-12f)) / lambda0;
    }
    // box regression targets (regression_loss, :264-279) and SmoothL1(sum, beta 1) against regress_param
    const float* b = boxes + (long long)i * box_cols;
    const float x = (b[1] + b[3]) / 2, yy = (b[2] + b[4]) / 2, w = b[3] - b[1], h = b[4] - b[2];
    const float* t = tloc + 4ll * i;
    const float xt = (t[0] + t[2]) / 2, yt = (t[1] + t[3]) / 2, wt = t[2] - t[0], ht = t[3] - t[1];
    float tgt[4];
    tgt[0] = (xt - x) / (w + 1e-16f);
    tgt[1] = (yt - yy) / (h + 1e-16f);
    tgt[2] = logf(wt / w + 1e-16f);
    tgt[3] = logf(ht / h + 1e-16f);
    for (int c = 0; c < 4; ++c) {
      const float d = tgt[c] - REG[4ll * i + c];
      const float ad = fabsf(d);
      const float l = ad < 1.f ? 0.5f * d * d : ad - 0.5f;
      if (c < 2) t_xy += l; else t_wh += l;
      const float dl = ad < 1.f ? d : (d > 0.f ? 1.f : -1.f);   // d l / d d; d = target - pred
      d_reg[4ll * i + c] = gscale * (-dl) / lambda1;
    }
  }
  terms[5ll * i + 0] = t_focal; terms[5ll * i + 1] = t_conf; terms[5ll * i + 2] = t_cat;
  terms[5ll * i + 3] = t_xy; terms[5ll * i + 4] = t_wh;
  d_o[2ll * i] = gscale * go0;
  d_o[2ll * i + 1] = gscale * go1;
}

// Tail of the stage-2 forward (my_models.py:341-364), one thread per proposal:
//   masks = softmax(o)  (two logits);  keep = masks[:,1] > threshold;  box_regress (:222-236) on the kept boxes;
//   row = (image_i, x1', y1', x2', y2', masks[:,1], cls_score, cls_pred);  key = masks[:,1]  ->  me_compact_sort_rows_f32.
__global__ __launch_bounds__(256) void m2_rows_kernel(const float* __restrict__ O, const float* __restrict__ REG,
                                                      const float* __restrict__ boxes, int box_cols, int k, float thr,
                                                      float* masks, float* rows, unsigned char* keep, float* key) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= k) return;
  const float o0 = O[2ll * i], o1 = O[2ll * i + 1];
  const float m = fmaxf(o0, o1);
  const float e0 = expf(o0 - m), e1 = expf(o1 - m);
  const float s = e0 + e1;
  const float p0 = e0 / s, p1 = e1 / s;
  masks[2ll * i] = p0;
  masks[2ll * i + 1] = p1;
  const float* b = boxes + (long long)i * box_cols;
  const float* rp = REG + 4ll * i;
  const float cx = (b[1] + b[3]) / 2, cy = (b[2] + b[4]) / 2, bw = b[3] - b[1], bh = b[4] - b[2];
  const float nx = rp[0] * bw + cx, ny = rp[1] * bh + cy, nw = expf(rp[2]) * bw, nh = expf(rp[3]) * bh;
  float* r = rows + 8ll * i;
  r[0] = b[0];
  r[1] = nx - nw / 2;
  r[2] = ny - nh / 2;
  r[3] = nx + nw / 2;
  r[4] = ny + nh / 2;
  r[5] = p1;
  r[6] = b[6];
  r[7] = b[7];
  keep[i] = p1 > thr ? 1 : 0;
  key[i] = p1;
}

// Input of the ensemble head (my_models.py:333-339): x2[i * c1 + c] = (refinement_vector[i][c], yolo_vector[i][c]) with
// yolo_vector = (object confidence, class scores) = columns 5, 8 .. 8 + class_num - 1 of the proposal rows.
__global__ __launch_bounds__(256) void m2_pairs_kernel(const float* __restrict__ R, const float* __restrict__ boxes, int box_cols,
                                                       int k, int c1, float* x2) {
  const long long total = (long long)k * c1;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % c1);
    const long long i = idx / c1;
    x2[2 * idx] = R[idx];
    x2[2 * idx + 1] = boxes[i * box_cols + (c == 0 ? 5 : 7 + c)];
  }
}

}  // namespace

extern "C" {

int me_m2_rows_f32(const float* o, const float* regress, const float* boxes, int32_t box_cols, int32_t k, float threshold,
                   float* masks, float* rows, uint8_t* keep, float* key, void* stream) {
  if (k == 0) return 0;
  ME_REQUIRE(o && regress && boxes && masks && rows && keep && key, ME_E_NULLPTR, "me_m2_rows_f32: null pointer");
  ME_REQUIRE(k > 0 && box_cols >= 8, ME_E_BADARG, "me_m2_rows_f32: bad dimensions");
  hipLaunchKernelGGL(m2_rows_kernel, dim3((k + 255) / 256), dim3(256), 0, (hipStream_t)stream, o, regress, boxes, box_cols, k,
                     threshold, masks, rows, keep, key);
  return me::check_launch("m2_rows_kernel");
}

int me_m2_pairs_f32(const float* refine, const float* boxes, int32_t box_cols, int32_t k, int32_t c1, float* x2, void* stream) {
  if (k == 0) return 0;
  ME_REQUIRE(refine && boxes && x2, ME_E_NULLPTR, "me_m2_pairs_f32: null pointer");
  ME_REQUIRE(k > 0 && c1 >= 2 && box_cols >= 7 + c1, ME_E_BADARG, "me_m2_pairs_f32: bad dimensions");
  hipLaunchKernelGGL(m2_pairs_kernel, dim3(grid_for((long long)k * c1)), dim3(256), 0, (hipStream_t)stream, refine, boxes,
                     box_cols, k, c1, x2);
  return me::check_launch("m2_pairs_kernel");
}


int me_linear_f32(const float* x, int64_t ldx, int64_t rows, int32_t in_features, const float* w, const float* bias,
                  int32_t out_features, int32_t act, float* y, int64_t ldy, void* stream) {
  if (rows == 0) return 0;
  ME_REQUIRE(x && w && y, ME_E_NULLPTR, "me_linear_f32: null pointer");
  ME_REQUIRE(rows > 0 && in_features > 0 && out_features > 0 && ldx >= in_features && ldy >= out_features, ME_E_BADARG,
             "me_linear_f32: bad dimensions");
  ME_REQUIRE(act >= 0 && act <= 2, ME_E_BADARG, "me_linear_f32: unknown activation %d", act);
  static const int naive = getenv("MILLIEYE_M2_LINEAR_NAIVE") ? atoi(getenv("MILLIEYE_M2_LINEAR_NAIVE")) : 0;   // (A/B, same bits)
  if (naive) {
    hipLaunchKernelGGL(linear_kernel, dim3(grid_for((long long)rows * out_features)), dim3(256), 0, (hipStream_t)stream, x,
                       (long long)ldx, (long long)rows, in_features, w, bias, out_features, act, y, (long long)ldy);
  } else if (out_features <= 16) {
    ME_REQUIRE((rows + 63) / 64 <= 65535, ME_E_TOOBIG, "me_linear_f32: too many rows");
    hipLaunchKernelGGL((linear_tiled_kernel<64, 16, 1>), dim3(1, (unsigned)((rows + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x,
                       (long long)ldx, (long long)rows, in_features, w, bias, out_features, act, y, (long long)ldy);
  } else {
    ME_REQUIRE((rows + 63) / 64 <= 65535, ME_E_TOOBIG, "me_linear_f32: too many rows");
    hipLaunchKernelGGL((linear_tiled_kernel<64, 64, 4>), dim3((out_features + 63) / 64, (unsigned)((rows + 63) / 64)), dim3(256), 0,
                       (hipStream_t)stream, x, (long long)ldx, (long long)rows, in_features, w, bias, out_features, act, y,
                       (long long)ldy);
  }
  return me::check_launch("linear_kernel");
}

int me_dropout_mask_u8(uint64_t seed, float keep_prob, int64_t count, uint8_t* mask, void* stream) {
  if (count == 0) return 0;
  ME_REQUIRE(mask != nullptr, ME_E_NULLPTR, "me_dropout_mask_u8: null pointer");
  ME_REQUIRE(count > 0 && keep_prob >= 0.f && keep_prob <= 1.f, ME_E_BADARG, "me_dropout_mask_u8: bad arguments");
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for((count + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (unsigned long long)seed, keep_prob, (long long)count, mask);
  return me::check_launch("dropout_mask_kernel");
}

int me_mask_scale_f32(const float* x, const uint8_t* mask, float scale, int64_t count, float* y, void* stream) {
  if (count == 0) return 0;
  ME_REQUIRE(x && mask && y, ME_E_NULLPTR, "me_mask_scale_f32: null pointer");
  hipLaunchKernelGGL(mask_scale_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, x, mask, scale,
                     (long long)count, y);
  return me::check_launch("mask_scale_kernel");
}

int me_m2_loss_f32(const float* o, const float* refine, int32_t c1, const float* regress, const float* boxes,
                   int32_t box_cols, const float* target_location, const float* class_label, const uint8_t* pos,
                   const uint8_t* sample, int32_t k, float alpha, float lambda0, float lambda1, float grad_scale,
                   float* terms, float* d_o, float* d_refine, float* d_regress, void* stream) {
  if (k == 0) return 0;
  ME_REQUIRE(o && refine && regress && boxes && target_location && class_label && pos && sample && terms && d_o &&
                 d_refine && d_regress, ME_E_NULLPTR, "me_m2_loss_f32: null pointer");
  ME_REQUIRE(k > 0 && c1 >= 2 && box_cols >= 5, ME_E_BADARG, "me_m2_loss_f32: bad dimensions");
  hipLaunchKernelGGL(m2_loss_kernel, dim3((k + 255) / 256), dim3(256), 0, (hipStream_t)stream, o, refine, c1, regress, boxes,
                     box_cols, target_location, class_label, pos, sample, k, alpha, lambda0, lambda1, grad_scale, terms, d_o,
                     d_refine, d_regress);
  return me::check_launch("m2_loss_kernel");
}

}  // extern "C"

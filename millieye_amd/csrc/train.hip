// train.hip - building blocks of the stage-3 training step (gfx950), fp32.
//
// The reference trains the fusion heads through torch autograd (module3_our_dataset/train.py:185-191,
// loss assembled in my_models.py:545-639).  Gradients reach: ensemble_head.fc1/fc2, refinement_head
// (net0, net2 rows 0-1, radar_net), img_cnn_layers (through ps_roi_align backward) and
// radar_cnn_layers (through roi_align backward); the detector is frozen and detached.
// This file provides the native kernels that training step is assembled from
// (millieye_amd/train_path.py owns the fixed graph and the order of launches):
//
//   me_gemm_f32            C = op(A) * op(B) (+C)      Linear / 1x1-conv forward, dgrad and wgrad
//   me_colsum_f32          bias gradients
//   me_bn_train_fwd_f32    batch statistics + normalise + LeakyReLU, running-stat update
//   me_bn_train_bwd_f32    dgamma, dbeta, dx (LeakyReLU backward fused)
//   me_act_bwd_f32         sigmoid / leaky backward (elementwise)
//   me_conv_wgrad_f32      3x3 / 1x1 weight gradient (NHWC gather, no im2col)
//   me_roi_align_bwd_f32 / me_ps_roi_align_bwd_f32   atomic scatter (torchvision backward semantics)
//
// These layers are tiny next to the detector (0.3 GFLOP/frame forward): the kernels favour
// simplicity and determinism (no atomics except the RoI scatters, which torchvision also does
// with atomics) over peak throughput; none of them is on the inference path.
#include <math.h>
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// GEMM: C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] + beta * C.  Row-major, leading dimensions.
// 64x64 tile, 256 threads, 4x4 outputs per thread, K step 16 through LDS.  Reduction over K is
// sequential per output element -> bit-reproducible.
// ---------------------------------------------------------------------------------------------
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, long long lda,
                                                   const float* __restrict__ B, long long ldb, float* C,
                                                   long long ldc, int M, int N, int K, float alpha, float beta) {
  __shared__ float As[16][64 + 1];
  __shared__ float Bs[16][64 + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    for (int idx = threadIdx.x; idx < 16 * 64; idx += 256) {
      const int kk = TA ? idx / 64 : idx % 16, mm = TA ? idx % 64 : idx / 16;
      const int m = m0 + mm, k = k0 + kk;
      float v = 0.f;
      if (m < M && k < K) v = TA ? A[(long long)k * lda + m] : A[(long long)m * lda + k];
      As[kk][mm] = v;
    }
    for (int idx = threadIdx.x; idx < 16 * 64; idx += 256) {
      const int kk = TB ? idx % 16 : idx / 64, nn = TB ? idx / 16 : idx % 64;
      const int n = n0 + nn, k = k0 + kk;
      float v = 0.f;
      if (n < N && k < K) v = TB ? B[(long long)n * ldb + k] : B[(long long)k * ldb + n];
      Bs[kk][nn] = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
      if (m < M && n < N) {
        float* c = C + (long long)m * ldc + n;
        *c = alpha * acc[i][j] + (beta != 0.f ? beta * *c : 0.f);
      }
    }
}

// column sums of X[rows, cols] (pitch ld): out[c] = sum_r X[r][c].  One 1024-thread workgroup per CW = min(64, pow2 >= cols)
// columns; the 1024 / CW row lanes of a column stride over the rows with eight independent partial sums in flight (the old
// version had 4 row lanes and one dependent chain: 130-300 us for the [1600 x 2] ... [5408 x 490] matrices of the stage-3
// backward), then a fixed-order LDS tree -> deterministic.
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ X, long long ld, int rows, int cols, int cw,
                                                      float* out) {
  __shared__ float red[1024];
  const int parts = 1024 / cw;
  const int lc = threadIdx.x % cw, part = threadIdx.x / cw;
  const int c = blockIdx.x * cw + lc;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < cols) {
    const long long step = (long long)parts * ld;
    const float* px = X + (long long)part * ld + c;
    int r = part;
    for (; r + 7 * parts < rows; r += 8 * parts, px += 8 * step) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += px[u * step];
    }
    for (int u = 0; r < rows; r += parts, px += step, ++u) acc[u] += px[0];
  }
  red[threadIdx.x] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  for (int half = parts >> 1; half >= 1; half >>= 1) {
    if (part < half) red[threadIdx.x] += red[threadIdx.x + half * cw];
    __syncthreads();
  }
  if (part == 0 && c < cols) out[c] = red[lc];
}

// ---------------------------------------------------------------------------------------------
// BatchNorm (training mode) over rows of X[rows, C] (pitch ld).
//   stats: mean[c], var[c] (biased) - two-level fixed-order reduction (64 row-chunks)
//   fwd:   y = act((x - mean) * rstd * gamma + beta);  running stats updated like nn.BatchNorm2d
//          (momentum m: running = (1-m)*running + m*stat, unbiased variance for running_var)
//   bwd:   g = dy * act'(y);  dbeta = sum g;  dgamma = sum g * xhat;
//          dx = gamma * rstd * (g - dbeta/rows - xhat * dgamma/rows)
// ---------------------------------------------------------------------------------------------
constexpr int BN_CHUNKS = 64;

__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ X, long long ld, int rows, int C,
                                                         const float* __restrict__ G, long long ldg,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int act, float* p0, float* p1, const int* rows_dev = nullptr) {
  // forward use (G == nullptr): p0 = sum x, p1 = sum x^2 over this block's row chunk
  // backward use: p0 = sum g, p1 = sum g * xhat with g = dy * act'(bn output)
  // rows_dev (captured steps): the live row count in device memory, `rows` is then the buffers' capacity
  // Round 6: a block = 64 channels x 4 row lanes (a wave reads 64 consecutive channels of one row; the four lanes of a channel walk
  // rows r0 + lane, + 4, ... and meet in LDS in a fixed order).  The first form gave every channel ONE thread per chunk: the ten
  // channels of the RoI-wise / radar BatchNorms ran on ten threads per workgroup, 85 dependent loads each (22 us per launch,
  // ten launches per stage-3 step).
  __shared__ double s0s[4][64], s1s[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int chunk = blockIdx.y;
  if (rows_dev) rows = *rows_dev < rows ? *rows_dev : rows;
  const int per = (rows + BN_CHUNKS - 1) / BN_CHUNKS;
  const int r0 = chunk * per, r1 = (r0 + per < rows) ? r0 + per : rows;
  double s0 = 0.0, s1 = 0.0;
  if (c < C) {
    if (G == nullptr) {
      for (int r = r0 + rl; r < r1; r += 4) {
        const float x = X[(long long)r * ld + c];
        s0 += x;
        s1 += (double)x * x;
      }
    } else {
      const float mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
      for (int r = r0 + rl; r < r1; r += 4) {
        const float xh = (X[(long long)r * ld + c] - mu) * rs;
        float g = G[(long long)r * ldg + c];
        if (act == ME_ACT_LEAKY) g = (xh * ga + be > 0.f) ? g : 0.1f * g;
        s0 += g;
        s1 += (double)g * xh;
      }
    }
  }
  s0s[rl][cl] = s0;
  s1s[rl][cl] = s1;
  __syncthreads();
  if (rl == 0 && c < C) {
    p0[(long long)chunk * C + c] = (float)(((s0s[0][cl] + s0s[1][cl]) + s0s[2][cl]) + s0s[3][cl]);
    p1[(long long)chunk * C + c] = (float)(((s1s[0][cl] + s1s[1][cl]) + s1s[2][cl]) + s1s[3][cl]);
  }
}

// the 64 chunk partials of one channel, summed in double by the 64 lanes of one wave (xor butterfly: every lane ends with the same
// sum, the association order is fixed).  Round 6: the two finishing kernels below ran a thread per channel through 64 dependent
// loads - 7-8 us per launch, ten launches per stage-3 step.
__device__ __forceinline__ void wave_sum2(double& s0, double& s1) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    s0 += __shfl_xor(s0, m, 64);
    s1 += __shfl_xor(s1, m, 64);
  }
}

__global__ __launch_bounds__(256) void bn_finish_stats_kernel(const float* p0, const float* p1, int rows, int C,
                                                              float eps, float momentum, float* mean, float* var,
                                                              float* rstd, float* running_mean, float* running_var) {
  static_assert(BN_CHUNKS == 64, "one lane per chunk");
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);   // a wave per channel
  if (c >= C) return;
  double s0 = p0[(long long)lane * C + c], s1 = p1[(long long)lane * C + c];
  wave_sum2(s0, s1);
  if (lane != 0) return;
  const double mu = s0 / rows;
  double v = s1 / rows - mu * mu;
  if (v < 0.0) v = 0.0;
  mean[c] = (float)mu;
  var[c] = (float)v;
  rstd[c] = (float)(1.0 / sqrt(v + (double)eps));
  if (running_mean) {
    const double unb = rows > 1 ? v * rows / (rows - 1) : v;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
  }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ X, long long ld, int rows, int C,
                                                       const float* mean, const float* rstd, const float* gamma,
                                                       const float* beta, int act, float* Y, long long ldy) {
  const long long total = (long long)rows * C;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % C);
    const long long r = idx / C;
    float v = (X[r * ld + c] - mean[c]) * rstd[c] * gamma[c] + beta[c];
    if (act == ME_ACT_LEAKY) v = v > 0.f ? v : 0.1f * v;
    Y[r * ldy + c] = v;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ X, long long ld,
                                                           const float* __restrict__ G, long long ldg, int rows,
                                                           int C, const float* mean, const float* rstd,
                                                           const float* gamma, const float* beta, int act,
                                                           const float* p0, const float* p1, float* dgamma,
                                                           float* dbeta, float* DX, long long lddx, const int* rows_dev = nullptr) {
  // p0[c] / p1[c] hold the channel sums (bn_reduce_partials_kernel folded the chunk partials, fixed order)
  const long long total = (long long)rows * C;   // (the capacity when rows_dev is given: rows behind the live count get dx = 0)
  if (rows_dev) rows = *rows_dev < rows ? *rows_dev : rows;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % C);
    const long long r = idx / C;
    if (r >= rows) {
      if (DX) DX[r * lddx + c] = 0.f;
      continue;
    }
    const double s0 = p0[c], s1 = p1[c];
    const float mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
    const float xh = (X[r * ld + c] - mu) * rs;
    float g = G[r * ldg + c];
    if (act == ME_ACT_LEAKY) g = (xh * ga + be > 0.f) ? g : 0.1f * g;
    if (DX) DX[r * lddx + c] = ga * rs * (g - (float)(s0 / rows) - xh * (float)(s1 / rows));
    if (r == 0) {
      if (dbeta) dbeta[c] = (float)s0;
      if (dgamma) dgamma[c] = (float)s1;
    }
  }
}

__global__ __launch_bounds__(256) void bn_reduce_partials_kernel(float* p0, float* p1, int C) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);   // a wave per channel; rows 1.. of the partials are read before row 0 is
  if (c >= C) return;                                  // written, and only by this wave
  double s0 = p0[(long long)lane * C + c], s1 = p1[(long long)lane * C + c];
  wave_sum2(s0, s1);
  if (lane == 0) {
    p0[c] = (float)s0;
    p1[c] = (float)s1;
  }
}

// elementwise activation backward: dx = dy * act'(y) given the activation OUTPUT y
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ Y, long long ldy,
                                                      const float* __restrict__ G, long long ldg, float* DX,
                                                      long long lddx, long long rows, int C, int act) {
  const long long total = rows * C;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % C);
    const long long r = idx / C;
    const float y = Y[r * ldy + c];
    float g = G[r * ldg + c];
    if (act == ME_ACT_SIGMOID) g *= y * (1.f - y);
    else if (act == ME_ACT_LEAKY) g = y > 0.f ? g : 0.1f * g;
    DX[r * lddx + c] = g;
  }
}

// ---------------------------------------------------------------------------------------------
// conv weight gradient: dW[co][ky][kx][ci] = sum_p dY[p][co] * X[p + tap][ci]  (stride 1/2, zero pad)
// one workgroup = 64 co x 64 ci for one tap; pixels stream through LDS 16 at a time.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ X, long long xp,
                                                         const float* __restrict__ DY, long long dyp, float* DW,
                                                         int n, int h, int w, int cin, int cout, int ks, int stride,
                                                         int pad, int ho, int wo) {
  __shared__ float Ys[16][64 + 1];
  __shared__ float Xs[16][64 + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int tap = blockIdx.z, ky = tap / ks, kx = tap - ky * ks;
  const int co0 = blockIdx.y * 64, ci0 = blockIdx.x * 64;
  const int P = n * ho * wo;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int p0 = 0; p0 < P; p0 += 16) {
    for (int idx = threadIdx.x; idx < 16 * 64; idx += 256) {
      const int pp = idx / 64, cc = idx % 64;
      const int p = p0 + pp;
      float vy = 0.f, vx = 0.f;
      if (p < P) {
        if (co0 + cc < cout) vy = DY[(long long)p * dyp + co0 + cc];
        const int nimg = p / (ho * wo);
        const int rem = p - nimg * ho * wo;
        const int oy = rem / wo, ox = rem - oy * wo;
        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        if ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w && ci0 + cc < cin)
          vx = X[((long long)(nimg * h + iy) * w + ix) * xp + ci0 + cc];
      }
      Ys[pp][cc] = vy;
      Xs[pp][cc] = vx;
    }
    __syncthreads();
#pragma unroll
    for (int pp = 0; pp < 16; ++pp) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = Ys[pp][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Xs[pp][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = co0 + ty * 4 + i, ci = ci0 + tx * 4 + j;
      if (co < cout && ci < cin) DW[((long long)co * ks * ks + tap) * cin + ci] = acc[i][j];
    }
}

// ---------------------------------------------------------------------------------------------
// conv weight gradient on the matrix pipe.  Per filter tap this is a GEMM whose reduction runs over the output
// pixels: dW_tap[co][ci] = sum_p dY[p][co] * X[p (+) tap][ci].  Both operands are "K-major" in memory (channels
// contiguous, pixels strided), so LDS holds [pixel][channel] tiles and every v_mfma_f32_32x32x2_f32 takes one
// ds_read_b32 per operand (lane i, k-half kk reads row 2t+kk, column i; the column is XOR-ed with 32*(row&1) so the
// two rows of a k-step hit different banks).  One workgroup = 64 co x 64 ci of one tap over a slice of the pixels
// (blockIdx.z = tap * splits + split); 4 waves, each a 32x32 accumulator.  Global loads of stage s+1 are issued into
// registers before the MFMAs of stage s (two LDS buffers, one barrier per stage).  Slices write slabs
// [split][cout][k*k][cin] that conv_wgrad_reduce_kernel adds in split order (deterministic).
// ---------------------------------------------------------------------------------------------
typedef float wg_f32x16 __attribute__((ext_vector_type(16)));

// ABL (tuning only, MILLIEYE_ABLATION=1 + MILLIEYE_WGRAD_ABL): 1 = global loads only for the first stage (matrix pipe + LDS +
// barriers), 2 = no MFMAs (memory side + LDS).  26^2 x 8, 256 -> 512, 3x3 (12.8 GFLOP = 81 us of matrix pipe): 142 us as
// built (15 of them the slab sum), 121 without the loads, 70 without the MFMAs; a second stage of loads in flight changed
// nothing (146 us) - the loop is bound by neither load latency nor L2 bytes, it loses ~25 us to LDS issue + barriers and ~22 us
// to imperfect overlap of the loads.
template <bool VEC, int ABL = 0>
__global__ __launch_bounds__(256) void conv_wgrad_mfma_kernel(const float* __restrict__ X, long long xp,
                                                              const float* __restrict__ DY, long long dyp, float* OUT,
                                                              int n, int h, int w, int cin, int cout, int ks, int stride,
                                                              int pad, int ho, int wo, int splits, int px_per_split) {
  __shared__ __attribute__((aligned(16))) float Ys[2][16][64];
  __shared__ __attribute__((aligned(16))) float Xs[2][16][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tap = blockIdx.z / splits, split = blockIdx.z - tap * splits;
  const int ky = tap / ks, kx = tap - ky * ks;
  const int co0 = blockIdx.y * 64, ci0 = blockIdx.x * 64;
  const int P = n * ho * wo;
  const int p_begin = split * px_per_split;
  const int p_end = (p_begin + px_per_split < P) ? p_begin + px_per_split : P;
  const int hw = ho * wo;
  // loader role: pixel row pp of the stage, 4 consecutive channels cc
  const int pp = tid >> 4, cc = (tid & 15) * 4;
  const int sw_cc = cc ^ ((pp & 1) << 5);  // swizzled LDS column

  float4 ry = make_float4(0.f, 0.f, 0.f, 0.f), rx = ry;
  // pixel coordinates of this loader lane, advanced by 16 pixels per stage with carries (round 3: the two integer divisions per
  // lane and stage that decoded p were ~100 vector instructions beside 8 MFMAs - with 8 waves per SIMD the kernel was bound
  // by them, not by the matrix pipe)
  int f_img, f_oy, f_ox;
  {
    const int p = p_begin + pp;
    f_img = p / hw;
    const int rem = p - f_img * hw;
    f_oy = rem / wo;
    f_ox = rem - f_oy * wo;
  }
  auto fetch = [&](int p0) {
    const int p = p0 + pp;
    ry = make_float4(0.f, 0.f, 0.f, 0.f);
    rx = ry;
    if (p < p_end) {
      const float* yrow = DY + (long long)p * dyp + co0 + cc;
      const int nimg = f_img, oy = f_oy, ox = f_ox;
      const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
      const bool inb = (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w;
      const float* xrow = X + ((long long)(nimg * h + iy) * w + ix) * xp + ci0 + cc;
      if (VEC) {
        if (co0 + cc < cout) ry = *reinterpret_cast<const float4*>(yrow);
        if (inb && ci0 + cc < cin) rx = *reinterpret_cast<const float4*>(xrow);
      } else {
        if (co0 + cc + 0 < cout) ry.x = yrow[0];
        if (co0 + cc + 1 < cout) ry.y = yrow[1];
        if (co0 + cc + 2 < cout) ry.z = yrow[2];
        if (co0 + cc + 3 < cout) ry.w = yrow[3];
        if (inb) {
          if (ci0 + cc + 0 < cin) rx.x = xrow[0];
          if (ci0 + cc + 1 < cin) rx.y = xrow[1];
          if (ci0 + cc + 2 < cin) rx.z = xrow[2];
          if (ci0 + cc + 3 < cin) rx.w = xrow[3];
        }
      }
    }
    f_ox += 16;  // the next stage's pixel
    while (f_ox >= wo) {
      f_ox -= wo;
      if (++f_oy == ho) {
        f_oy = 0;
        ++f_img;
      }
    }
  };

  wg_f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int i32 = lane & 31, kk = lane >> 5;
  const int a_col = (wr * 32 + i32) ^ (kk << 5), b_col = (wc * 32 + i32) ^ (kk << 5);

  fetch(p_begin);
  int buf = 0;
  for (int p0 = p_begin; p0 < p_end; p0 += 16) {
    *reinterpret_cast<float4*>(&Ys[buf][pp][sw_cc]) = ry;
    *reinterpret_cast<float4*>(&Xs[buf][pp][sw_cc]) = rx;
    __syncthreads();
    if (p0 + 16 < p_end && (ABL != 1 || p0 == p_begin)) fetch(p0 + 16);  // in flight while the matrix pipe works
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float a = Ys[buf][2 * t + kk][a_col];
      const float b = Xs[buf][2 * t + kk][b_col];
      if (ABL != 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      else acc[t] += a * b;
    }
    buf ^= 1;
  }
  // C layout of the 32x32 MFMA: lane (column i32 = ci, row group kk), element e -> row (e&3) + 8*(e>>2) + 4*kk = co
  float* out = OUT + (long long)split * cout * ks * ks * cin;
  const int ci = ci0 + wc * 32 + i32;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int co = co0 + wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
    if (co < cout && ci < cin) out[((long long)co * ks * ks + tap) * cin + ci] = acc[e];
  }
}

// The same GEMM on larger tiles (round 3): TCO x TCI outputs per workgroup, each of the 4 waves (TCO/2) x (TCI/2) =
// MT x NT accumulators of 32x32.  A 64x64 tile moves 8 KB of operands per 131 KFLOP (16 FLOP/B out of L2 - 5 TB/s of L2
// traffic at the 83 TFLOP/s it reached on the 3x3 layers); 128x128 halves the bytes and the LDS reads per FLOP.  16-byte
// loads only (the launcher keeps the 64x64 kernel above for ragged / unaligned operands).
template <int TCO, int TCI>
__global__ __launch_bounds__(256) void conv_wgrad_tile_kernel(const float* __restrict__ X, long long xp,
                                                              const float* __restrict__ DY, long long dyp, float* OUT, int n,
                                                              int h, int w, int cin, int cout, int ks, int stride, int pad,
                                                              int ho, int wo, int splits, int px_per_split) {
  constexpr int MT = TCO / 64, NT = TCI / 64;      // 32x32 blocks per wave
  constexpr int YV = TCO / 64, XV = TCI / 64;      // float4 loads per lane and stage: 16 px x T/4 quads over 256 lanes
  __shared__ __attribute__((aligned(16))) float Ys[2][16][TCO];
  __shared__ __attribute__((aligned(16))) float Xs[2][16][TCI];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tap = blockIdx.z / splits, split = blockIdx.z - tap * splits;
  const int ky = tap / ks, kx = tap - ky * ks;
  const int co0 = blockIdx.y * TCO, ci0 = blockIdx.x * TCI;
  const int P = n * ho * wo;
  const int p_begin = split * px_per_split;
  const int p_end = (p_begin + px_per_split < P) ? p_begin + px_per_split : P;
  const int hw = ho * wo;
  // loader role: pixel row pp of the stage, channel quads cc + 64 * v
  const int pp = tid >> 4, cc = (tid & 15) * 4;
  const int sw = (pp & 1) << 5;

  float4 ry[YV], rx[XV];
  int f_img, f_oy, f_ox;
  {
    const int p = p_begin + pp;
    f_img = p / hw;
    const int rem = p - f_img * hw;
    f_oy = rem / wo;
    f_ox = rem - f_oy * wo;
  }
  auto fetch = [&](int p0) {
    const int p = p0 + pp;
#pragma unroll
    for (int v = 0; v < YV; ++v) ry[v] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int v = 0; v < XV; ++v) rx[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < p_end) {
      const float* yrow = DY + (long long)p * dyp + co0 + cc;
      const int iy = f_oy * stride - pad + ky, ix = f_ox * stride - pad + kx;
      const bool inb = (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w;
      const float* xrow = X + ((long long)(f_img * h + iy) * w + ix) * xp + ci0 + cc;
#pragma unroll
      for (int v = 0; v < YV; ++v)
        if (co0 + cc + 64 * v < cout) ry[v] = *reinterpret_cast<const float4*>(yrow + 64 * v);
      if (inb) {
#pragma unroll
        for (int v = 0; v < XV; ++v)
          if (ci0 + cc + 64 * v < cin) rx[v] = *reinterpret_cast<const float4*>(xrow + 64 * v);
      }
    }
    f_ox += 16;
    while (f_ox >= wo) {
      f_ox -= wo;
      if (++f_oy == ho) {
        f_oy = 0;
        ++f_img;
      }
    }
  };

  wg_f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int i32 = lane & 31, kk = lane >> 5;
  const int a_base = wr * (TCO / 2) + i32, b_base = wc * (TCI / 2) + i32, ksw = kk << 5;

  fetch(p_begin);
  int buf = 0;
  for (int p0 = p_begin; p0 < p_end; p0 += 16) {
#pragma unroll
    for (int v = 0; v < YV; ++v) *reinterpret_cast<float4*>(&Ys[buf][pp][(cc + 64 * v) ^ sw]) = ry[v];
#pragma unroll
    for (int v = 0; v < XV; ++v) *reinterpret_cast<float4*>(&Xs[buf][pp][(cc + 64 * v) ^ sw]) = rx[v];
    __syncthreads();
    if (p0 + 16 < p_end) fetch(p0 + 16);  // in flight while the matrix pipe works
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = Ys[buf][2 * t + kk][(a_base + 32 * i) ^ ksw];
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = Xs[buf][2 * t + kk][(b_base + 32 * j) ^ ksw];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    buf ^= 1;
  }
  float* out = OUT + (long long)split * cout * ks * ks * cin;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int ci = ci0 + wc * (TCI / 2) + 32 * j + i32;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wr * (TCO / 2) + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * kk;
        if (co < cout && ci < cin) out[((long long)co * ks * ks + tap) * cin + ci] = acc[i][j][e];
      }
    }
}

// Round 5: conv_wgrad_tile_kernel with D stages of global loads in flight per lane, and a half-wide tile.
// (a) The 1x1 filters have 2 - 16 output tiles, so their pixel range is cut into many short slices, one or two workgroups per CU,
//     sixteen 16-pixel stages each - and every stage waited for its own loads (one stage of prefetch = eight MFMAs = 0.2 us of
//     cover for ~1 us of L2 / ~2 us of HBM latency): 27 - 31 us per layer for 9 us of matrix work (tools/wgrad_bench.py, batch 8).
//     With D = 4 register sets the loads of stage s + 4 are issued behind stage s's LDS stores.
// (b) TCI = 32 (cin = 32: 416^2 32 -> 64 / stride 2 and 208^2 32 -> 64): the 64 x 64 tile was half padding (38 TFLOP/s).  The
//     tile is 64 co x 32 ci; the two waves of a co block split the k-steps of a stage (t < 4 / t >= 4) and add their accumulators
//     through LDS at the end - four busy waves on a 64 x 32 tile.
template <int TCO, int TCI, int D>
__global__ __launch_bounds__(256) void conv_wgrad_pipe_kernel(const float* __restrict__ X, long long xp,
                                                              const float* __restrict__ DY, long long dyp, float* OUT, int n,
                                                              int h, int w, int cin, int cout, int ks, int stride, int pad,
                                                              int ho, int wo, int splits, int px_per_split) {
  constexpr bool KH = TCI == 32;                   // K split across the wave pair instead of two ci blocks
  constexpr int TCIL = KH ? 64 : TCI;              // LDS row width of the x tile
  constexpr int MT = TCO / 64, NT = KH ? 1 : TCI / 64;
  constexpr int YV = TCO / 64, XV = KH ? 1 : TCI / 64;
  __shared__ __attribute__((aligned(16))) float Ys[2][16][TCO];
  __shared__ __attribute__((aligned(16))) float Xs[2][16][TCIL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = KH ? 0 : (wave & 1), kh = KH ? (wave & 1) : 0;
  const int tap = blockIdx.z / splits, split = blockIdx.z - tap * splits;
  const int ky = tap / ks, kx = tap - ky * ks;
  const int co0 = blockIdx.y * TCO, ci0 = blockIdx.x * TCI;
  const int P = n * ho * wo;
  const int p_begin = split * px_per_split;
  const int p_end = (p_begin + px_per_split < P) ? p_begin + px_per_split : P;
  const int hw = ho * wo;
  const int pp = tid >> 4, cc = (tid & 15) * 4;
  const int sw = (pp & 1) << 5;
  const bool x_lane = !KH || cc < 32;

  float4 ry[D][YV], rx[D][XV];
  int f_img, f_oy, f_ox;
  {
    const int p = p_begin + pp;
    f_img = p / hw;
    const int rem = p - f_img * hw;
    f_oy = rem / wo;
    f_ox = rem - f_oy * wo;
  }
  int p_next = p_begin;  // first pixel of the next stage to fetch (stages are fetched in order)
  auto fetch = [&](float4(&ryd)[YV], float4(&rxd)[XV]) {
    const int p = p_next + pp;
#pragma unroll
    for (int v = 0; v < YV; ++v) ryd[v] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int v = 0; v < XV; ++v) rxd[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < p_end) {
      const float* yrow = DY + (long long)p * dyp + co0 + cc;
      const int iy = f_oy * stride - pad + ky, ix = f_ox * stride - pad + kx;
      const bool inb = (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w;
      const float* xrow = X + ((long long)(f_img * h + iy) * w + ix) * xp + ci0 + cc;
#pragma unroll
      for (int v = 0; v < YV; ++v)
        if (co0 + cc + 64 * v < cout) ryd[v] = *reinterpret_cast<const float4*>(yrow + 64 * v);
      if (inb && x_lane) {
#pragma unroll
        for (int v = 0; v < XV; ++v)
          if (ci0 + cc + 64 * v < cin) rxd[v] = *reinterpret_cast<const float4*>(xrow + 64 * v);
      }
    }
    p_next += 16;
    f_ox += 16;
    while (f_ox >= wo) {
      f_ox -= wo;
      if (++f_oy == ho) {
        f_oy = 0;
        ++f_img;
      }
    }
  };

  wg_f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int i32 = lane & 31, kk = lane >> 5;
  const int a_base = wr * (TCO / 2) + i32, b_base = (KH ? 0 : wc * (TCI / 2)) + i32, ksw = kk << 5;

#pragma unroll
  for (int u = 0; u < D; ++u) fetch(ry[u], rx[u]);   // (stages behind the slice's end come back as zeros)
  int buf = 0;
  for (int p0 = p_begin; p0 < p_end; p0 += 16 * D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      if (p0 + 16 * u < p_end) {   // (uniform over the workgroup)
#pragma unroll
        for (int v = 0; v < YV; ++v) *reinterpret_cast<float4*>(&Ys[buf][pp][(cc + 64 * v) ^ sw]) = ry[u][v];
        if (x_lane) {
#pragma unroll
          for (int v = 0; v < XV; ++v) *reinterpret_cast<float4*>(&Xs[buf][pp][(cc + 64 * v) ^ sw]) = rx[u][v];
        }
        __syncthreads();
        if (p_next < p_end) fetch(ry[u], rx[u]);  // stage s + D, in flight for the next D stages
#pragma unroll
        for (int t0 = 0; t0 < (KH ? 4 : 8); ++t0) {
          const int t = KH ? t0 + 4 * kh : t0;
          float a[MT], b[NT];
#pragma unroll
          for (int i = 0; i < MT; ++i) a[i] = Ys[buf][2 * t + kk][(a_base + 32 * i) ^ ksw];
#pragma unroll
          for (int j = 0; j < NT; ++j) b[j] = Xs[buf][2 * t + kk][(b_base + 32 * j) ^ ksw];
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        buf ^= 1;
      }
    }
  }
  if constexpr (KH) {
    // the k-halves of a co block meet in LDS: wave (wr, 1) parks its accumulators, wave (wr, 0) adds them (fixed order)
    __syncthreads();
    float* park = &Ys[0][0][0];   // 2 waves x 64 lanes x 16 floats * MT <= 2 * 16 * TCO floats
    if (kh == 1) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) park[((wr * MT + i) * 16 + e) * 64 + lane] = acc[i][0][e];
    }
    __syncthreads();
    if (kh == 1) return;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][0][e] += park[((wr * MT + i) * 16 + e) * 64 + lane];
  }
  float* out = OUT + (long long)split * cout * ks * ks * cin;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int ci = ci0 + (KH ? 0 : wc * (TCI / 2)) + 32 * j + i32;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wr * (TCO / 2) + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * kk;
        if (co < cout && ci < cin) out[((long long)co * ks * ks + tap) * cin + ci] = acc[i][j][e];
      }
    }
}

// Round 5: the per-tap weight gradient on 16-BIT operands (the mixed-precision detector step, millieye_amd/detector_train16.py):
// x and dc are bfloat16 / IEEE half NHWC, the products go through v_mfma_f32_32x32x16_bf16 / _f16 (16x the fp32 matrix rate), the
// accumulators and the slabs are fp32 and the slab sums are train.hip's fixed-order reductions - the gradient is float32.
// A stage is 32 pixels (two MFMA k-steps): 16-byte loads (8 channels), LDS image [pixel][channel] as loaded.  The reduction
// dimension (pixels) is the STRIDED one of both operands, so an MFMA fragment (8 consecutive pixels of one channel) is eight
// 2-byte LDS reads packed in registers - the LDS pipe, not the matrix pipe, bounds the kernel (~1/4 - 1/2 of the 16-bit peak,
// still an order of magnitude over the fp32 kernels); a transposing read (ds_read_b64_tr_b16) is the next step if it matters.
template <int F16>
struct WgH16;
template <>
struct WgH16<0> {
  typedef __bf16 v8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ wg_f32x16 mfma(v8 a, v8 b, wg_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <>
struct WgH16<1> {
  typedef _Float16 v8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ wg_f32x16 mfma(v8 a, v8 b, wg_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// TR = 1 (round 5, gfx950): the fragments come from ds_read_b64_tr_b16.  Both operands sit in LDS as [pixel][channel] rows (the
// NHWC order they are stored in) while an MFMA fragment is eight consecutive PIXELS of one channel - the k-strided case the
// transposed read exists for: per 16-lane group, lane i hands in the address of 4 contiguous channels of pixel row i / 4 and gets
// back 4 consecutive pixels of channel column (lane & 15).  Two such reads build a fragment (pixels 8 kk .. 8 kk + 7 of channel
// lane & 31) where the TR = 0 form issues eight 2-byte reads and four pack operations.  Rows are padded by 64 bytes so that the
// four rows a half-wave touches fall on different banks (pitch 320 B: 80 r mod 64 = 0, 16, 32, 48 dwords; 192 B: 0, 48, 32, 16).
typedef short wg_v4s __attribute__((ext_vector_type(4)));

template <int TCO, int TCI, int F16, int TR = 0>
__global__ __launch_bounds__(256) void conv_wgrad_tile_h16_kernel(const unsigned short* __restrict__ X, long long xp,
                                                                  const unsigned short* __restrict__ DY, long long dyp, float* OUT,
                                                                  int n, int h, int w, int cin, int cout, int ks, int stride, int pad,
                                                                  int ho, int wo, int splits, int px_per_split) {
  using v8 = typename WgH16<F16>::v8;
  constexpr int SP = 32;                          // pixels per stage
  constexpr int MT = TCO / 64, NT = TCI / 64;     // 32x32 blocks per wave (2 x 2 waves)
  constexpr int YL = TCO / 64, XL = TCI / 64;     // 16-byte loads per lane and stage: 32 px x T/8 octets over 256 lanes
  constexpr int YO = TCO / 8, XO = TCI / 8;       // octets per pixel row
  constexpr int PY = TCO + (TR ? 32 : 0), PX = TCI + (TR ? 32 : 0);   // LDS row pitches (elements)
  __shared__ __attribute__((aligned(16))) unsigned short Ys[2][SP][PY];
  __shared__ __attribute__((aligned(16))) unsigned short Xs[2][SP][PX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tap = blockIdx.z / splits, split = blockIdx.z - tap * splits;
  const int ky = tap / ks, kx = tap - ky * ks;
  const int co0 = blockIdx.y * TCO, ci0 = blockIdx.x * TCI;
  const int P = n * ho * wo;
  const int p_begin = split * px_per_split;
  const int p_end = (p_begin + px_per_split < P) ? p_begin + px_per_split : P;
  const int hw = ho * wo;

  // loader lanes: load v of a lane is pixel row (tid + 256 v) / octets-per-row, channel octet (tid + 256 v) % octets-per-row
  int x_px[XL], x_oc[XL], f_img[XL], f_oy[XL], f_ox[XL];
#pragma unroll
  for (int v = 0; v < XL; ++v) {
    const int flat = tid + 256 * v;
    x_px[v] = flat / XO;
    x_oc[v] = (flat - x_px[v] * XO) * 8;
    const int p = p_begin + x_px[v];
    f_img[v] = p / hw;
    const int rem = p - f_img[v] * hw;
    f_oy[v] = rem / wo;
    f_ox[v] = rem - f_oy[v] * wo;
  }
  uint4 ry[YL], rx[XL];
  // (two register sets with the loads issued two stages ahead: measured slower, 47 -> 58 us on the 128 x 128 3x3 layers)
  auto fetch = [&](int p0) {
#pragma unroll
    for (int v = 0; v < YL; ++v) {
      const int flat = tid + 256 * v;
      const int px = flat / YO, oc = (flat - px * YO) * 8;
      const int p = p0 + px;
      ry[v] = make_uint4(0u, 0u, 0u, 0u);
      if (p < p_end && co0 + oc < cout) ry[v] = *reinterpret_cast<const uint4*>(DY + (long long)p * dyp + co0 + oc);
    }
#pragma unroll
    for (int v = 0; v < XL; ++v) {
      const int p = p0 + x_px[v];
      rx[v] = make_uint4(0u, 0u, 0u, 0u);
      if (p < p_end) {
        const int iy = f_oy[v] * stride - pad + ky, ix = f_ox[v] * stride - pad + kx;
        if ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w && ci0 + x_oc[v] < cin)
          rx[v] = *reinterpret_cast<const uint4*>(X + ((long long)(f_img[v] * h + iy) * w + ix) * xp + ci0 + x_oc[v]);
      }
      f_ox[v] += SP;   // the next stage's pixel
      while (f_ox[v] >= wo) {
        f_ox[v] -= wo;
        if (++f_oy[v] == ho) {
          f_oy[v] = 0;
          ++f_img[v];
        }
      }
    }
  };

  wg_f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int i32 = lane & 31, kk = lane >> 5;
  const int a_base = wr * (TCO / 2) + i32, b_base = wc * (TCI / 2) + i32;

  // (a fragment = pixels 8 kk .. 8 kk + 7 of a 16-pixel k-step of one channel: eight 16-bit LDS reads, packed in pairs)
  fetch(p_begin);
  int buf = 0;
  for (int p0 = p_begin; p0 < p_end; p0 += SP) {
#pragma unroll
    for (int v = 0; v < YL; ++v) {
      const int flat = tid + 256 * v;
      const int px = flat / YO, oc = (flat - px * YO) * 8;
      *reinterpret_cast<uint4*>(&Ys[buf][px][oc]) = ry[v];
    }
#pragma unroll
    for (int v = 0; v < XL; ++v) *reinterpret_cast<uint4*>(&Xs[buf][x_px[v]][x_oc[v]]) = rx[v];
    __syncthreads();
    if (p0 + SP < p_end) fetch(p0 + SP);  // in flight while the matrix pipe works
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      v8 a[MT], b[NT];
      if constexpr (TR) {
        using lds_v4 = __attribute__((address_space(3))) wg_v4s;
        const int row = 16 * s + 8 * kk + ((lane & 15) >> 2);           // + 4 for the fragment's second half
        const int col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);        // within the 32-channel block
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const unsigned short* q = &Ys[buf][row][wr * (TCO / 2) + 32 * i + col];
          const wg_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)q);
          const wg_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(q + 4 * PY));
          a[i] = __builtin_bit_cast(v8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const unsigned short* q = &Xs[buf][row][wc * (TCI / 2) + 32 * j + col];
          const wg_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)q);
          const wg_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(q + 4 * PX));
          b[j] = __builtin_bit_cast(v8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
      } else {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        unsigned wd[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          wd[q] = (unsigned)Ys[buf][16 * s + 8 * kk + 2 * q][a_base + 32 * i] |
                  ((unsigned)Ys[buf][16 * s + 8 * kk + 2 * q + 1][a_base + 32 * i] << 16);
        a[i] = __builtin_bit_cast(v8, make_uint4(wd[0], wd[1], wd[2], wd[3]));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        unsigned wd[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          wd[q] = (unsigned)Xs[buf][16 * s + 8 * kk + 2 * q][b_base + 32 * j] |
                  ((unsigned)Xs[buf][16 * s + 8 * kk + 2 * q + 1][b_base + 32 * j] << 16);
        b[j] = __builtin_bit_cast(v8, make_uint4(wd[0], wd[1], wd[2], wd[3]));
      }
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = WgH16<F16>::mfma(a[i], b[j], acc[i][j]);
    }
    buf ^= 1;
  }
  float* out = OUT + (long long)split * cout * ks * ks * cin;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int ci = ci0 + wc * (TCI / 2) + 32 * j + i32;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wr * (TCO / 2) + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * kk;
        if (co < cout && ci < cin) out[((long long)co * ks * ks + tap) * cin + ci] = acc[i][j][e];
      }
    }
}

// kk_cin > 0: write the sum in the parameter's own OIHW layout (i indexes the OHWI slabs: co, tap, ci) - the autograd result of
// the detector's training step without a permute + contiguous launch per layer
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ slabs, float* DW, long long count,
                                                                int splits, int kk, int cin) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
    float v = slabs[i];
    for (int k = 1; k < splits; ++k) v += slabs[(long long)k * count + i];
    if (kk > 0) {
      const long long r = i / cin;
      const int ci = (int)(i - r * cin);
      const long long co = r / kk;
      const int tap = (int)(r - co * kk);
      DW[(co * cin + ci) * kk + tap] = v;
    } else {
      DW[i] = v;
    }
  }
}

// Weight gradient of the network's first convolution (3 input channels, 3x3): on the 64x64 MFMA tile above 3 of 64 input
// channels are real and the K loop is 1.4 M pixels long - 1.4 ms of a Darknet-53 step at batch 8 for 2.4 GFLOP.  Here a
// lane owns 4 output channels x all 27 (tap, ci) products of one pixel lane: 256 threads = 8 channel groups x 32 pixel
// lanes, 108 accumulators per lane, the pixel's 27 inputs and 4 gradients are loaded once per 108 FMAs.  Per block the 32
// pixel lanes are summed through LDS in a fixed order; a second kernel adds the blocks' partials (fixed order too).
constexpr int SW_BLOCKS = 768;

__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ X, long long xp,
                                                         const float* __restrict__ DY, long long dyp, float* PART, int n,
                                                         int h, int w, int cout, int stride, int pad, int ho, int wo) {
  __shared__ float s_r[32][8][28];
  const int cg = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int co0 = blockIdx.y * 32 + cg * 4;
  const bool live = co0 < cout;  // cout % 4 == 0
  const long long P = (long long)n * ho * wo;
  const int hw = ho * wo;
  float acc[4][27];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[j][t] = 0.f;
  if (live) {
    for (long long p = (long long)blockIdx.x * 32 + pl; p < P; p += (long long)gridDim.x * 32) {
      const int img = (int)(p / hw);
      const int rem = (int)(p - (long long)img * hw);
      const int oy = rem / wo, ox = rem - oy * wo;
      const float4 g = *reinterpret_cast<const float4*>(DY + p * dyp + co0);
      float xv[27];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * stride - pad + ky;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = ox * stride - pad + kx;
          const bool inb = (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w;
          const float* px = X + ((long long)(img * h + iy) * w + ix) * xp;
#pragma unroll
          for (int ci = 0; ci < 3; ++ci) xv[(ky * 3 + kx) * 3 + ci] = inb ? px[ci] : 0.f;
        }
      }
#pragma unroll
      for (int t = 0; t < 27; ++t) {
        acc[0][t] += g.x * xv[t];
        acc[1][t] += g.y * xv[t];
        acc[2][t] += g.z * xv[t];
        acc[3][t] += g.w * xv[t];
      }
    }
  }
  // partial of this block in OHWI order: [co][tap][ci] = co * 27 + t
  float* out = PART + ((long long)blockIdx.x * cout) * 27;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 27; ++t) s_r[pl][cg][t] = acc[j][t];
    __syncthreads();
    if (threadIdx.x < 8 * 27) {
      const int g2 = threadIdx.x / 27, t = threadIdx.x - g2 * 27;
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) v += s_r[k][g2][t];
      const int co = blockIdx.y * 32 + g2 * 4 + j;
      if (co < cout) out[(long long)co * 27 + t] = v;
    }
  }
}

// sum of the stem partials: 16 outputs x 16 lanes per block; OHWI (oihw = 0) or OIHW result
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ PART, float* DW, int count, int nblk,
                                                                int oihw) {
  __shared__ float s_p[16][17];
  const int ol = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int o = blockIdx.x * 16 + ol;
  float v = 0.f;
  if (o < count)
    for (int k = part; k < nblk; k += 16) v += PART[(long long)k * count + o];
  s_p[part][ol] = v;
  __syncthreads();
  if (part == 0 && o < count) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += s_p[k][ol];
    int dst = o;
    if (oihw) {  // o = (co * 9 + tap) * 3 + ci  ->  (co * 3 + ci) * 9 + tap
      const int co = o / 27, r = o - co * 27, tap = r / 3, ci = r - tap * 3;
      dst = (co * 3 + ci) * 9 + tap;
    }
    DW[dst] = t;
  }
}

// Round 3: the two shapes that matter, without the per-element divisions and the 4-byte stride-k*k stores of the kernel
// above (25 us per layer on average, 1.9 ms of a Darknet-53 step at batch 8 - a sixth of the weight gradient's time).
// (1) flat sum, 16 bytes per lane, four slabs in flight (1x1 filters and the OHWI result);
__global__ __launch_bounds__(256) void conv_wgrad_reduce_v4_kernel(const float* __restrict__ slabs, float* DW, long long count4,
                                                                   int splits) {
  const float4* S = reinterpret_cast<const float4*>(slabs);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count4; i += (long long)gridDim.x * 256) {
    float4 v = S[i];
    int k = 1;
    for (; k + 3 < splits; k += 4) {
      const float4 a = S[(long long)k * count4 + i], b = S[(long long)(k + 1) * count4 + i],
                   c = S[(long long)(k + 2) * count4 + i], d = S[(long long)(k + 3) * count4 + i];
      v.x = (((v.x + a.x) + b.x) + c.x) + d.x;
      v.y = (((v.y + a.y) + b.y) + c.y) + d.y;
      v.z = (((v.z + a.z) + b.z) + c.z) + d.z;
      v.w = (((v.w + a.w) + b.w) + c.w) + d.w;
    }
    for (; k < splits; ++k) {
      const float4 a = S[(long long)k * count4 + i];
      v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    reinterpret_cast<float4*>(DW)[i] = v;
  }
}

// (2) OHWI slabs -> OIHW parameter layout: a block owns one output channel x 64 input channels x all taps; the slab reads
// are runs of 64 floats per tap, the sums turn through LDS ([ci][tap], odd pitch) and leave as one contiguous run.
__global__ __launch_bounds__(256) void conv_wgrad_reduce_oihw_kernel(const float* __restrict__ slabs, float* DW, long long count,
                                                                     int splits, int kk, int cin, int vec16) {
  extern __shared__ float s_t[];  // [64][kk | 1]
  const int pitch = kk | 1;
  const int co = blockIdx.y, c0 = blockIdx.x * 64;
  const int nc = cin - c0 < 64 ? cin - c0 : 64;
  if (vec16 && nc == 64) {  // (vec16: cin % 4 == 0 and a 16-byte aligned workspace, checked by the host side)
    // 16-byte loads, eight slabs in flight per lane (round 4: the scalar loop below kept 4 KB per workgroup in flight and
    // summed 38 MB of nine-tap slabs at 1.5 - 2.5 TB/s)
    for (int i = threadIdx.x; i < kk * 16; i += 256) {
      const int tap = i >> 4, q4 = (i & 15) * 4;
      const long long idx = ((long long)co * kk + tap) * cin + c0 + q4;
      float4 v = *reinterpret_cast<const float4*>(slabs + idx);
      int k = 1;
      for (; k + 7 < splits; k += 8) {
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4*>(slabs + (long long)(k + u) * count + idx);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          v.x += t[u].x; v.y += t[u].y; v.z += t[u].z; v.w += t[u].w;
        }
      }
      for (; k < splits; ++k) {
        const float4 t = *reinterpret_cast<const float4*>(slabs + (long long)k * count + idx);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      s_t[(q4 + 0) * pitch + tap] = v.x;
      s_t[(q4 + 1) * pitch + tap] = v.y;
      s_t[(q4 + 2) * pitch + tap] = v.z;
      s_t[(q4 + 3) * pitch + tap] = v.w;
    }
  } else
  for (int i = threadIdx.x; i < kk * 64; i += 256) {
    const int tap = i >> 6, cl = i & 63;
    if (cl < nc) {
      const long long idx = ((long long)co * kk + tap) * cin + c0 + cl;
      float v = slabs[idx];
      int k = 1;
      for (; k + 3 < splits; k += 4) {
        const float a = slabs[(long long)k * count + idx], b = slabs[(long long)(k + 1) * count + idx],
                    c = slabs[(long long)(k + 2) * count + idx], d = slabs[(long long)(k + 3) * count + idx];
        v = (((v + a) + b) + c) + d;
      }
      for (; k < splits; ++k) v += slabs[(long long)k * count + idx];
      s_t[cl * pitch + tap] = v;
    }
  }
  __syncthreads();
  float* dst = DW + ((long long)co * cin + c0) * kk;
  for (int i = threadIdx.x; i < nc * kk; i += 256) {
    const int cl = i / kk;  // (kk is small: 9, 25, 49)
    dst[i] = s_t[cl * pitch + (i - cl * kk)];
  }
}

// ---------------------------------------------------------------------------------------------
// RoI pooling backward (torchvision roi_align / ps_roi_align backward semantics, oracle/tv_ops.c):
// every sample scatters grad * w / count to its four bilinear corners with atomicAdd.
// grad_map is NHWC [n,h,w,c] (pitch), zero-filled by the caller.
// ---------------------------------------------------------------------------------------------
constexpr int P7 = 7;

__device__ __forceinline__ int grid_of7(float extent) {
  const float g = ceilf(extent / (float)P7);
  if (!(g < 1.0e9f)) return (g != g) ? 0 : 1000000000;
  if (g < -1.0e9f) return -1000000000;
  return (int)g;
}

__device__ __forceinline__ void scatter_bilinear(float* gmap, long long pitch, int height, int width, int c, float y,
                                                 float x, float gval) {
  if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) return;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else { y_high = y_low + 1; }
  if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else { x_high = x_low + 1; }
  const float ly = y - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
  atomicAdd(gmap + ((long long)y_low * width + x_low) * pitch + c, gval * (hy * hx));
  atomicAdd(gmap + ((long long)y_low * width + x_high) * pitch + c, gval * (hy * lx));
  atomicAdd(gmap + ((long long)y_high * width + x_low) * pitch + c, gval * (ly * hx));
  atomicAdd(gmap + ((long long)y_high * width + x_high) * pitch + c, gval * (ly * lx));
}

__global__ __launch_bounds__(256) void roi_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ rois,
                                                      int k, int c, int h, int w, float scale, float* gmap,
                                                      long long pitch, int ps, const int* k_dev = nullptr, int xsplit = 0) {
  // gout: [k, c_out, 7, 7] with c_out = ps ? c/49 : c
  // xsplit (round 6, batches of >= 4 frames): the RoIs of frame b are scattered by the workgroups of XCD b % 8 only (workgroup ids
  // are dealt round-robin over the eight XCDs).  The proposals of a frame pile up on the same few objects, so their atomics hit the
  // same lines of the gradient map; with the RoIs dealt over all XCDs those lines migrated between the eight L2s for every add
  // (120 us per launch at ~300 RoIs); within one XCD they stay in its L2.  Every XCD's workgroups scan all indices and keep their frames'.
  if (k_dev) k = *k_dev < k ? *k_dev : k;   // (captured steps: k = capacity, the live RoI count in device memory)
  const int cout = ps ? c / 49 : c;
  const long long total = (long long)k * cout * 49;
  const int xcd = blockIdx.x & 7;
  const long long first = xsplit ? (long long)(blockIdx.x >> 3) * 256 + threadIdx.x : (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = xsplit ? (long long)(gridDim.x >> 3) * 256 : (long long)gridDim.x * 256;
  for (long long idx = first; idx < total; idx += stride) {
    const int pw = (int)(idx % P7), ph = (int)((idx / P7) % P7);
    const int cc = (int)((idx / 49) % cout);
    const int r = (int)(idx / (49ll * cout));
    const float* roi = rois + 5 * r;
    const int b = (int)roi[0];
    if (xsplit && (b & 7) != xcd) continue;
    const float off = ps ? 0.5f : 0.0f;
    const float sw = roi[1] * scale - off, sh = roi[2] * scale - off;
    const float ew = roi[3] * scale - off, eh = roi[4] * scale - off;
    float roi_w = ew - sw, roi_h = eh - sh;
    if (!ps) {
      roi_w = roi_w > 1.f ? roi_w : 1.f;
      roi_h = roi_h > 1.f ? roi_h : 1.f;
    }
    const float bin_h = roi_h / (float)P7, bin_w = roi_w / (float)P7;
    const int gh = grid_of7(roi_h), gw = grid_of7(roi_w);
    if (gh <= 0 || gw <= 0 || gh > 4096 || gw > 4096) continue;
    const float count = (float)(gh * gw);
    const float g = gout[idx];
    const int ch = ps ? (cc * P7 + ph) * P7 + pw : cc;
    float* gimg = gmap + (long long)b * h * w * pitch;
    const float ybase = ps ? ((float)ph * bin_h + sh) : (sh + ph * bin_h);
    const float xbase = ps ? ((float)pw * bin_w + sw) : (sw + pw * bin_w);
    for (int iy = 0; iy < gh; ++iy) {
      const float yy = ybase + ((float)(iy + .5f)) * bin_h / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        const float xx = xbase + ((float)(ix + .5f)) * bin_w / (float)gw;
        scatter_bilinear(gimg, pitch, h, w, ch, yy, xx, g / count);
      }
    }
  }
}

// The same scatter through LDS (round 6).  A stage-3 step of 8 frames has ~1200 proposals piled on a few objects: the kernel
// above issued 2.4-9 M global float atomics on the same few hundred lines (126 us per launch in the step's trace).  Here a
// workgroup owns one frame and a slice of the gradient map small enough for LDS, scatters into it with LDS atomics and writes
// the slice out once:
//   PS  (ps_roi_align): output channel (cc*7+ph)*7+pw is fed by bin (ph, pw) only, so workgroup (bin, frame) owns the c/49
//       channels of that bin in all h*w cells - exclusively: one add per non-zero cell, no two workgroups on an address;
//   !PS (roi_align): workgroup (split, frame) takes every split-th RoI of the frame on the whole [h, w, c] map and adds its
//       non-zero cells to the global map with atomics (gridDim.x adds per cell at most).
// The RoIs of the frame are compacted into an LDS list first (2048 candidates per round).  Summation order within a cell is
// the order the LDS atomics retire in - not fixed, like the global atomics of the kernel above.
constexpr int ROI_LDS_LIST = 2048;
constexpr int ROI_LDS_THREADS = 512;

template <bool PS>
__global__ __launch_bounds__(ROI_LDS_THREADS) void roi_bwd_lds_kernel(const float* __restrict__ gout,
                                                                       const float* __restrict__ rois, int k, int c, int h,
                                                                       int w, float scale, float* gmap, long long pitch,
                                                                       const int* k_dev) {
  extern __shared__ float s_map[];   // [h*w][cl]: cl = c/49 channels of one bin (PS) or all c channels (!PS)
  __shared__ int s_list[ROI_LDS_LIST];
  __shared__ int s_n;
  if (k_dev) k = *k_dev < k ? *k_dev : k;
  const int b = blockIdx.y;
  const int cout = PS ? c / 49 : c;   // channels of grad_out
  const int cl = cout;                // channels of the LDS slice
  const int cells = h * w * cl;
  for (int i = threadIdx.x; i < cells; i += ROI_LDS_THREADS) s_map[i] = 0.f;
  const int bin = PS ? (int)blockIdx.x : 0;
  const int bph = bin / P7, bpw = bin % P7;
  const int split = PS ? 0 : (int)blockIdx.x, nsplit = PS ? 1 : (int)gridDim.x;
  const int per_roi = PS ? cout : cout * 49;   // scatter items per RoI handled by this workgroup
  for (int base = 0; base < k; base += ROI_LDS_LIST) {
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int lim = (base + ROI_LDS_LIST < k) ? base + ROI_LDS_LIST : k;
    for (int r = base + threadIdx.x; r < lim; r += ROI_LDS_THREADS) {
      if ((int)rois[5 * r] == b && (r % nsplit) == split) s_list[atomicAdd(&s_n, 1)] = r;
    }
    __syncthreads();
    const int items = s_n * per_roi;
    for (int it = threadIdx.x; it < items; it += ROI_LDS_THREADS) {
      // lanes of a wave: channel fastest, then RoI, the bin slowest - 64 lanes are ~6 RoIs x all channels of ONE bin, so two lanes
      // meet on an LDS address only where two RoIs overlap (with the bin fastest the 49 bins of a small RoI fell into two or three
      // cells of the same channel: 20-way same-address adds, 80 us per launch at 1200 RoIs)
      int r, cc, ph, pw;
      if (PS) {
        r = s_list[it / per_roi];
        cc = it % per_roi; ph = bph; pw = bpw;
      } else {
        const int nr = s_n, b49 = it / (nr * cout), rem = it - b49 * (nr * cout);
        r = s_list[rem / cout];
        cc = rem % cout; ph = b49 / P7; pw = b49 % P7;
      }
      const float* roi = rois + 5 * r;
      const float off = PS ? 0.5f : 0.0f;
      const float sw = roi[1] * scale - off, sh = roi[2] * scale - off;
      const float ew = roi[3] * scale - off, eh = roi[4] * scale - off;
      float roi_w = ew - sw, roi_h = eh - sh;
      if (!PS) {
        roi_w = roi_w > 1.f ? roi_w : 1.f;
        roi_h = roi_h > 1.f ? roi_h : 1.f;
      }
      const float bin_h = roi_h / (float)P7, bin_w = roi_w / (float)P7;
      const int gh = grid_of7(roi_h), gw = grid_of7(roi_w);
      if (gh <= 0 || gw <= 0 || gh > 4096 || gw > 4096) continue;
      const float count = (float)(gh * gw);
      const float g = gout[((long long)r * cout + cc) * 49 + ph * P7 + pw];
      const float ybase = PS ? ((float)ph * bin_h + sh) : (sh + ph * bin_h);
      const float xbase = PS ? ((float)pw * bin_w + sw) : (sw + pw * bin_w);
      for (int iy = 0; iy < gh; ++iy) {
        const float yy = ybase + ((float)(iy + .5f)) * bin_h / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
          const float xx = xbase + ((float)(ix + .5f)) * bin_w / (float)gw;
          scatter_bilinear(s_map, cl, h, w, cc, yy, xx, g / count);
        }
      }
    }
    __syncthreads();
  }
  float* gimg = gmap + (long long)b * h * w * pitch;
  for (int i = threadIdx.x; i < cells; i += ROI_LDS_THREADS) {
    const float v = s_map[i];
    if (v == 0.f) continue;   // (NaN compares unequal and is written)
    const int cell = i / cl, cc = i - cell * cl;
    // (PS: this workgroup alone owns the element; the add without return value does not wait for the memory round trip)
    atomicAdd(gimg + (long long)cell * pitch + (PS ? (cc * P7 + bph) * P7 + bpw : cc), v);
  }
}


// ---------------------------------------------------------------------------------------------
// detector backward building blocks (Darknet.forward(x, targets) -> loss.backward(), eval-mode BatchNorm)
// ---------------------------------------------------------------------------------------------
// conv block y = act(scale * c + shift), c = conv(x, W), (scale, shift) = folded BN(eval) or (1, bias):
// g = dy * act'(y); dc = g * scale; per channel s0 = sum g (= dbeta / dbias), s1 = sum g * xhat (= dgamma) with
// xhat = (z - beta) / gamma recovered from the stored output (z = act^-1(y): LeakyReLU is invertible).
// grid (ceil(C / 64), chunks): a block owns 64 channels x one row chunk; its 256 threads are 64 channels x 4 row lanes
// (a wave reads 64 consecutive channels of one row), the four lanes of a channel are combined through LDS.
__global__ __launch_bounds__(256) void affine_bwd_partial_kernel(const float* __restrict__ Y, long long ldy,
                                                                 const float* __restrict__ G, long long ldg, int rows,
                                                                 int C, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int act, float* p0,
                                                                 float* p1, int chunks, const float* __restrict__ scale,
                                                                 float* DC, long long lddc) {
  // (round 3: dc = g * scale does not depend on the sums, so it is written in this pass - the separate apply pass re-read
  // y and dy of every layer: 0.95 of the 3.65 ms the three kernels took per Darknet-53 step at batch 8)
  __shared__ double s0s[4][64], s1s[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int chunk = blockIdx.y;
  const int per = (rows + chunks - 1) / chunks;
  const int r0 = chunk * per, r1 = (r0 + per < rows) ? r0 + per : rows;
  double s0 = 0.0, s1 = 0.0;
  if (c < C) {
    const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    const float inv_ga = (gamma && ga != 0.f) ? 1.f / ga : 0.f;
    const float scl = scale ? scale[c] : 1.f;
    for (int r = r0 + rl; r < r1; r += 4) {
      const float y = Y[(long long)r * ldy + c];
      float g = G[(long long)r * ldg + c];
      float z = y;
      if (act == ME_ACT_LEAKY) {
        g = y > 0.f ? g : 0.1f * g;
        z = y > 0.f ? y : y * 10.f;
      }
      s0 += g;
      s1 += (double)g * ((z - be) * inv_ga);
      if (DC) DC[(long long)r * lddc + c] = scale ? g * scl : g;
    }
  }
  s0s[rl][cl] = s0;
  s1s[rl][cl] = s1;
  __syncthreads();
  if (rl == 0 && c < C) {
    p0[(long long)chunk * C + c] = (float)(((s0s[0][cl] + s0s[1][cl]) + s0s[2][cl]) + s0s[3][cl]);
    p1[(long long)chunk * C + c] = (float)(((s1s[0][cl] + s1s[1][cl]) + s1s[2][cl]) + s1s[3][cl]);
  }
}

// The same pass with 16-byte accesses and four rows in flight per lane (round 3: the scalar kernel above ran one 4-byte
// load pair per lane and iteration - 1.5 TB/s on the 4 GB the Darknet-53 step moves through it at batch 8).  A block owns
// 64 channels x one row chunk; its 256 threads are 16 channel quads x 16 row lanes, each lane walks rows rl, rl + 16, ...
// four at a time; the 16 row lanes of a channel are combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void affine_bwd_partial_v4_kernel(const float* __restrict__ Y, long long ldy,
                                                                    const float* __restrict__ G, long long ldg, int rows,
                                                                    int C, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, int act, float* p0,
                                                                    float* p1, int chunks, const float* __restrict__ scale,
                                                                    float* DC, long long lddc) {
  __shared__ double s0s[16][64], s1s[16][64];
  const int q = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + q * 4;
  const int chunk = blockIdx.y;
  const int per = (rows + chunks - 1) / chunks;
  const int r0 = chunk * per, r1 = (r0 + per < rows) ? r0 + per : rows;
  double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
  if (c < C) {  // C % 4 == 0: the quad is all in or all out
    float ga[4], be[4], inv_ga[4], scl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ga[j] = gamma ? gamma[c + j] : 1.f;
      be[j] = beta ? beta[c + j] : 0.f;
      inv_ga[j] = (gamma && ga[j] != 0.f) ? 1.f / ga[j] : 0.f;
      scl[j] = scale ? scale[c + j] : 1.f;
    }
    auto one = [&](int r, float4 y4, float4 g4) {
      float y[4] = {y4.x, y4.y, y4.z, y4.w}, g[4] = {g4.x, g4.y, g4.z, g4.w}, d[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float z = y[j];
        if (act == ME_ACT_LEAKY) {
          g[j] = y[j] > 0.f ? g[j] : 0.1f * g[j];
          z = y[j] > 0.f ? y[j] : y[j] * 10.f;
        }
        s0[j] += g[j];
        s1[j] += (double)g[j] * ((z - be[j]) * inv_ga[j]);
        d[j] = scale ? g[j] * scl[j] : g[j];
      }
      if (DC) *reinterpret_cast<float4*>(DC + (long long)r * lddc + c) = make_float4(d[0], d[1], d[2], d[3]);
    };
    int r = r0 + rl;
    for (; r + 48 < r1; r += 64) {
      float4 yv[4], gv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        yv[u] = *reinterpret_cast<const float4*>(Y + (long long)(r + 16 * u) * ldy + c);
        gv[u] = *reinterpret_cast<const float4*>(G + (long long)(r + 16 * u) * ldg + c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(r + 16 * u, yv[u], gv[u]);
    }
    for (; r < r1; r += 16)
      one(r, *reinterpret_cast<const float4*>(Y + (long long)r * ldy + c),
          *reinterpret_cast<const float4*>(G + (long long)r * ldg + c));
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s0s[rl][q * 4 + j] = s0[j];
    s1s[rl][q * 4 + j] = s1[j];
  }
  __syncthreads();
  const int cl = threadIdx.x;
  if (cl < 64 && blockIdx.x * 64 + cl < C) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      a += s0s[k][cl];
      b += s1s[k][cl];
    }
    p0[(long long)chunk * C + blockIdx.x * 64 + cl] = (float)a;
    p1[(long long)chunk * C + blockIdx.x * 64 + cl] = (float)b;
  }
}

// Round 5: the same pass with every workgroup on a CONTIGUOUS range of rows (all channels of a row, up to 1024): the 64-channel
// column blocks above read 256-byte pieces of every row (a quarter of it at 256 channels, a sixteenth at 1024), half their
// lanes idle at 32 channels, and the row chunks were sized by rows alone - 13 x 13 x 1024 at batch 8 ran on 80 workgroups.
// Measured on the Darknet-53 shapes at batch 8 (tools/affine_bench.py): 2.0 TB/s over the 3.7 GB of a step, 1 TB/s on the
// 52 x 52 x 128 layers.  Here Q = min(C / 4, 256) lanes take the channel quads of a row, the other 256 / Q lane groups take
// different rows; a lane keeps four rows of y and dy (8 x 16 bytes) in flight; the per-lane sums go through LDS once per
// workgroup in a fixed order (row lane 0, 1, ...) and leave as one partial row per workgroup, summed by
// affine_bwd_reduce_kernel in chunk order - deterministic, and bit-identical for a given (rows, C) on every run.
template <int Q>
__global__ __launch_bounds__(256) void affine_bwd_rows_kernel(const float* __restrict__ Y, long long ldy,
                                                              const float* __restrict__ G, long long ldg, int rows, int C,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              int act, float* p0, float* p1, int per,
                                                              const float* __restrict__ scale, float* DC, long long lddc) {
  constexpr int RL = 256 / Q;  // row lanes
  __shared__ double s0s[RL][4 * Q], s1s[RL][4 * Q];
  const int q = threadIdx.x % Q, rl = threadIdx.x / Q;
  const int c = blockIdx.x * (4 * Q) + q * 4;   // (blockIdx.x > 0 only beyond 1024 channels)
  const int chunk = blockIdx.y;
  const int r0 = chunk * per, r1 = (r0 + per < rows) ? r0 + per : rows;
  double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
  if (c < C) {  // C % 4 == 0: the quad is all in or all out
    float be[4], inv_ga[4], scl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float ga = gamma ? gamma[c + j] : 1.f;
      be[j] = beta ? beta[c + j] : 0.f;
      inv_ga[j] = (gamma && ga != 0.f) ? 1.f / ga : 0.f;
      scl[j] = scale ? scale[c + j] : 1.f;
    }
    auto one = [&](int r, float4 y4, float4 g4) {
      float y[4] = {y4.x, y4.y, y4.z, y4.w}, g[4] = {g4.x, g4.y, g4.z, g4.w}, d[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float z = y[j];
        if (act == ME_ACT_LEAKY) {
          g[j] = y[j] > 0.f ? g[j] : 0.1f * g[j];
          z = y[j] > 0.f ? y[j] : y[j] * 10.f;
        }
        s0[j] += g[j];
        s1[j] += (double)g[j] * ((z - be[j]) * inv_ga[j]);
        d[j] = scale ? g[j] * scl[j] : g[j];
      }
      if (DC) *reinterpret_cast<float4*>(DC + (long long)r * lddc + c) = make_float4(d[0], d[1], d[2], d[3]);
    };
    int r = r0 + rl;
    for (; r + 3 * RL < r1; r += 4 * RL) {
      float4 yv[4], gv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        yv[u] = *reinterpret_cast<const float4*>(Y + (long long)(r + RL * u) * ldy + c);
        gv[u] = *reinterpret_cast<const float4*>(G + (long long)(r + RL * u) * ldg + c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(r + RL * u, yv[u], gv[u]);
    }
    for (; r < r1; r += RL)
      one(r, *reinterpret_cast<const float4*>(Y + (long long)r * ldy + c),
          *reinterpret_cast<const float4*>(G + (long long)r * ldg + c));
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s0s[rl][q * 4 + j] = s0[j];
    s1s[rl][q * 4 + j] = s1[j];
  }
  __syncthreads();
  for (int cl = threadIdx.x; cl < 4 * Q; cl += 256) {
    const int cg = blockIdx.x * (4 * Q) + cl;
    if (cg >= C) continue;
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int k = 0; k < RL; ++k) {
      a += s0s[k][cl];
      b += s1s[k][cl];
    }
    p0[(long long)chunk * C + cg] = (float)a;
    p1[(long long)chunk * C + cg] = (float)b;
  }
}

// second level of the fixed-order reduction: 64 channels x 16 chunk lanes per workgroup (a lane sums the chunks
// k = lane, lane + 16, ... in double), then a fixed LDS tree - the one-thread-per-channel loop over up to 1024 chunks took 31 us
__global__ __launch_bounds__(1024) void affine_bwd_reduce_kernel(float* p0, float* p1, int C, int chunks, float* dshift,
                                                                 float* dgamma) {
  __shared__ double r0[16][64], r1[16][64];
  const int lc = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lc;
  double s0 = 0.0, s1 = 0.0;
  if (c < C)
    for (int k = part; k < chunks; k += 16) {
      s0 += p0[(long long)k * C + c];
      s1 += p1[(long long)k * C + c];
    }
  r0[part][lc] = s0;
  r1[part][lc] = s1;
  __syncthreads();
  for (int half = 8; half >= 1; half >>= 1) {
    if (part < half) {
      r0[part][lc] += r0[part + half][lc];
      r1[part][lc] += r1[part + half][lc];
    }
    __syncthreads();
  }
  // every lane of the block has finished reading the partials of this block's channels before row 0 is overwritten
  if (part == 0 && c < C) {
    p0[c] = (float)r0[0][lc];
    p1[c] = (float)r1[0][lc];
    if (dshift) dshift[c] = (float)r0[0][lc];
    if (dgamma) dgamma[c] = (float)r1[0][lc];
  }
}

// nearest x2 upsample backward: dx[n,y,x,c] += sum of the 2x2 block of dy
__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const float* __restrict__ DY, long long lddy, float* DX,
                                                            long long lddx, int n, int h, int w, int c) {
  const long long total = (long long)n * h * w * c;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int cc = (int)(idx % c);
    long long pix = idx / c;
    const int x = (int)(pix % w);
    pix /= w;
    const int y = (int)(pix % h);
    const int nimg = (int)(pix / h);
    const long long o = ((long long)(nimg * 2 * h + 2 * y) * (2 * w) + 2 * x);
    const float v = (DY[o * lddy + cc] + DY[(o + 1) * lddy + cc]) + (DY[(o + 2 * w) * lddy + cc] + DY[(o + 2 * w + 1) * lddy + cc]);
    DX[((long long)(nimg * h + y) * w + x) * lddx + cc] += v;
  }
}

// max-pool backward (size k, stride s, zero_ext = the darknet ZeroPad2d((0,1,0,1)) before a stride-1 pool): the
// gradient of every output goes to the FIRST maximal input of its window (aten max_pool2d picks the first in
// row-major order); a padded zero that wins receives nothing.  One thread per INPUT element gathers from the <= k*k
// windows that cover it (no atomics, deterministic).
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ X, long long ldx,
                                                          const float* __restrict__ DY, long long lddy, float* DX,
                                                          long long lddx, int n, int h, int w, int c, int k, int s, int pad,
                                                          int zero_ext, int ho, int wo) {
  const long long total = (long long)n * h * w * c;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int cc = (int)(idx % c);
    long long pix = idx / c;
    const int ix = (int)(pix % w);
    pix /= w;
    const int iy = (int)(pix % h);
    const int nimg = (int)(pix / h);
    float acc = 0.f;
    for (int oy = 0; oy < ho; ++oy) {
      const int y0 = oy * s - pad;
      if (iy < y0 || iy >= y0 + k) continue;
      for (int ox = 0; ox < wo; ++ox) {
        const int x0 = ox * s - pad;
        if (ix < x0 || ix >= x0 + k) continue;
        // argmax of the window, first maximum in (row, col) order; out-of-range cells: -inf (pad) or 0 (zero_ext)
        float best = -INFINITY;
        int by = -1, bx = -1;
        for (int a = 0; a < k; ++a)
          for (int b = 0; b < k; ++b) {
            const int yy = y0 + a, xx = x0 + b;
            float v;
            if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w)
              v = X[((long long)(nimg * h + yy) * w + xx) * ldx + cc];
            else if (zero_ext && yy <= h && xx <= w && yy >= 0 && xx >= 0)
              v = 0.f;
            else
              continue;
            if (v > best) {
              best = v;
              by = yy;
              bx = xx;
            }
          }
        if (by == iy && bx == ix) acc += DY[((long long)(nimg * ho + oy) * wo + ox) * lddy + cc];
      }
    }
    DX[((long long)(nimg * h + iy) * w + ix) * lddx + cc] += acc;
  }
}

// gradient of the YOLO loss (yolov3/models.py:196-214) w.r.t. the raw detection map [N,G,G,A*(5+C)] (NHWC, pitch):
// loss = mse(x) + mse(y) + mse(w) + mse(h) (means over the n_obj object cells) + obj_scale * bce(conf | obj) +
// noobj_scale * bce(conf | noobj) (means over n_obj / n_noobj) + bce(cls | obj) (mean over n_obj * C).
// masks / targets are the [N,A,G,G] tensors build_targets returns (tcls [N,A,G,G,C]).
__global__ __launch_bounds__(256) void yolo_loss_bwd_kernel(const float* __restrict__ raw, long long pitch, int n, int g,
                                                            int na, int nc, const unsigned char* __restrict__ obj,
                                                            const unsigned char* __restrict__ noobj,
                                                            const float* __restrict__ tx, const float* __restrict__ ty,
                                                            const float* __restrict__ tw, const float* __restrict__ th,
                                                            const float* __restrict__ tcls, const float* __restrict__ tconf,
                                                            float n_obj, float n_noobj, float obj_scale, float noobj_scale,
                                                            float gscale, float* draw, long long dpitch,
                                                            const float* __restrict__ result_dev,
                                                            const float* __restrict__ gscale_dev) {
  if (result_dev) {  // me_yolo_loss_bwd_dev_f32: the counts the forward left in result[13] / [14], never read by the host
    n_obj = result_dev[13];
    n_noobj = result_dev[14];
  }
  if (gscale_dev) gscale = *gscale_dev;
  const int per = nc + 5;
  const long long total = (long long)n * na * g * g * per;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int k = (int)(idx % per);
    long long t = idx / per;
    const int gx = (int)(t % g);
    t /= g;
    const int gy = (int)(t % g);
    t /= g;
    const int a = (int)(t % na);
    const int nimg = (int)(t / na);
    const long long cell = (((long long)nimg * na + a) * g + gy) * g + gx;           // [N,A,G,G]
    const long long at = ((long long)(nimg * g + gy) * g + gx);
    const float r = raw[at * pitch + a * per + k];
    const bool is_obj = obj[cell] != 0, is_noobj = noobj[cell] != 0;
    float d = 0.f;
    if (k < 2) {
      if (is_obj) {
        const float sg = 1.f / (1.f + expf(-r));
        d = 2.f * (sg - (k == 0 ? tx[cell] : ty[cell])) / n_obj * sg * (1.f - sg);
      }
    } else if (k < 4) {
      if (is_obj) d = 2.f * (r - (k == 2 ? tw[cell] : th[cell])) / n_obj;
    } else if (k == 4) {
      const float sg = 1.f / (1.f + expf(-r));
      if (is_obj) d += obj_scale * (sg - tconf[cell]) / n_obj;
      if (is_noobj) d += noobj_scale * (sg - tconf[cell]) / n_noobj;
    } else if (is_obj) {
      const float sg = 1.f / (1.f + expf(-r));
      d = (sg - tcls[cell * nc + (k - 5)]) / (n_obj * (float)nc);
    }
    draw[at * dpitch + a * per + k] = d * gscale;
  }
}

inline unsigned grid1d(long long work) {
  long long b = (work + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// IoU labels of the stage-3 proposals (reference my_models.py:317-375 obtain_iou_labels with the call site's always-truthy
// multi_boxes, quirk q5): per proposal the FIRST maximum of the +1-pixel IoU over the targets of the same image and class,
// 0 when there is none.  fp32, the reference's operation order, no FMA contraction: bit-identical with the host restatement
// (train_path.iou_labels_vectorized).  One thread per proposal; also packs what the host needs for the metric and the
// negative sampling into ONE buffer (iou, kept flag, conf_1, conf_2), so the training forward reads the device once.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void iou_labels_kernel(const float* img_boxes, int n_img, int cols, const float* radar_boxes,
                                                         int n_radar, const float* targets, int q, const float* refine,
                                                         const float* mask1, const unsigned char* keep, float* out) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int k = n_img + n_radar;
  if (i >= k) return;
  float img, cls, b0, b1, b2, b3, conf1;
  if (i < n_img) {
    const float* r = img_boxes + (long long)i * cols;
    img = r[0]; cls = r[7]; b0 = r[1]; b1 = r[2]; b2 = r[3]; b3 = r[4]; conf1 = r[5];
  } else {
    const float* r = radar_boxes + (long long)(i - n_img) * 5;
    img = r[0]; cls = 0.f; b0 = r[1]; b1 = r[2]; b2 = r[3]; b3 = r[4]; conf1 = refine[2 * i];
  }
  const float a1 = (b2 - b0 + 1.f) * (b3 - b1 + 1.f);
  float best = -1.f;
  for (int t = 0; t < q; ++t) {
    const float* tg = targets + (long long)t * 6;
    if (tg[0] != img || tg[1] != cls) continue;
    const float ix1 = fmaxf(b0, tg[2]), iy1 = fmaxf(b1, tg[3]);
    const float ix2 = fminf(b2, tg[4]), iy2 = fminf(b3, tg[5]);
    const float inter = fmaxf(ix2 - ix1 + 1.f, 0.f) * fmaxf(iy2 - iy1 + 1.f, 0.f);
    const float a2 = (tg[4] - tg[2] + 1.f) * (tg[5] - tg[3] + 1.f);
    const float iou = inter / (a1 + a2 - inter + 1e-16f);
    if (iou > best) best = iou;
  }
  out[4 * i] = best < 0.f ? 0.f : best;
  out[4 * i + 1] = keep[i] ? 1.f : 0.f;
  out[4 * i + 2] = conf1;
  out[4 * i + 3] = mask1[i];
}

extern "C" {

int me_iou_labels_f32(const float* img_boxes, int32_t n_img, int32_t cols, const float* radar_boxes, int32_t n_radar,
                      const float* targets, int32_t q, const float* refine, const float* mask1, const uint8_t* keep, float* out,
                      void* stream) {
  const int k = n_img + n_radar;
  ME_REQUIRE(n_img >= 0 && n_radar >= 0 && q >= 0 && cols >= 8, ME_E_BADARG, "me_iou_labels_f32: bad sizes");
  if (k == 0) return 0;
  ME_REQUIRE(out && refine && mask1 && keep && (n_img == 0 || img_boxes) && (n_radar == 0 || radar_boxes) && (q == 0 || targets),
             ME_E_NULLPTR, "me_iou_labels_f32: null pointer");
  hipLaunchKernelGGL(iou_labels_kernel, dim3((k + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), img_boxes,
                     n_img, cols, radar_boxes, n_radar, targets, q, refine, mask1, keep, out);
  return me::check_launch("iou_labels_kernel");
}


int me_gemm_f32(int32_t trans_a, int32_t trans_b, int32_t m, int32_t n, int32_t k, float alpha, const float* a,
                int64_t lda, const float* b, int64_t ldb, float beta, float* c, int64_t ldc, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(m >= 0 && n >= 0 && k >= 0, ME_E_BADARG, "me_gemm_f32: negative dimension");
  if (m == 0 || n == 0) return 0;
  ME_REQUIRE(c && (k == 0 || (a && b)), ME_E_NULLPTR, "me_gemm_f32: null pointer");
  dim3 grid((n + 63) / 64, (m + 63) / 64);
  ME_REQUIRE(grid.y <= 65535, ME_E_TOOBIG, "me_gemm_f32: M too large");
  const long long la = lda, lb = ldb, lc = ldc;
  if (!trans_a && !trans_b)
    hipLaunchKernelGGL((gemm_kernel<false, false>), grid, dim3(256), 0, stream, a, la, b, lb, c, lc, m, n, k, alpha, beta);
  else if (trans_a && !trans_b)
    hipLaunchKernelGGL((gemm_kernel<true, false>), grid, dim3(256), 0, stream, a, la, b, lb, c, lc, m, n, k, alpha, beta);
  else if (!trans_a && trans_b)
    hipLaunchKernelGGL((gemm_kernel<false, true>), grid, dim3(256), 0, stream, a, la, b, lb, c, lc, m, n, k, alpha, beta);
  else
    hipLaunchKernelGGL((gemm_kernel<true, true>), grid, dim3(256), 0, stream, a, la, b, lb, c, lc, m, n, k, alpha, beta);
  return me::check_launch("gemm_kernel");
}

int me_colsum_f32(const float* x, int64_t ld, int32_t rows, int32_t cols, float* out, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(out && (x || rows == 0), ME_E_NULLPTR, "me_colsum_f32: null pointer");
  ME_REQUIRE(rows >= 0 && cols > 0, ME_E_BADARG, "me_colsum_f32: bad dimensions");
  int cw = 1;
  while (cw < cols && cw < 64) cw <<= 1;
  // tall matrices: narrower column groups, so that a row lane walks <= ~16 rows (two rounds of its eight partial sums) and the
  // launch has more workgroups - [5408 x 490] was 8 workgroups of 338 rows per lane (a chain of 42 dependent load rounds); down
  // to 8 columns (32 B per row and lane group) the reads still fill half a 64-byte sector
  while (cw > 8 && (long long)rows * cw > 16 * 1024) cw >>= 1;
  hipLaunchKernelGGL(colsum_kernel, dim3((cols + cw - 1) / cw), dim3(1024), 0, stream, x, (long long)ld, rows, cols, cw, out);
  return me::check_launch("colsum_kernel");
}

int64_t me_bn_workspace_bytes(int32_t channels) { return (int64_t)2 * BN_CHUNKS * channels * sizeof(float); }

int me_bn_train_fwd_f32(const float* x, int64_t ldx, int32_t rows, int32_t channels, const float* gamma,
                        const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                        int32_t act, float* y, int64_t ldy, float* save_mean, float* save_var, float* save_rstd,
                        void* workspace, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(x && gamma && beta && y && save_mean && save_var && save_rstd && workspace, ME_E_NULLPTR,
             "me_bn_train_fwd_f32: null pointer");
  ME_REQUIRE(rows > 0 && channels > 0, ME_E_BADARG, "me_bn_train_fwd_f32: bad dimensions");
  float* p0 = reinterpret_cast<float*>(workspace);
  float* p1 = p0 + (long long)BN_CHUNKS * channels;
  const unsigned cb = (channels + 3) / 4;   // the finishing kernels: a wave per channel
  hipLaunchKernelGGL(bn_partial_kernel, dim3((channels + 63) / 64, BN_CHUNKS), dim3(256), 0, stream, x, (long long)ldx, rows, channels,
                     (const float*)nullptr, 0ll, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, 0, p0, p1);
  hipLaunchKernelGGL(bn_finish_stats_kernel, dim3(cb), dim3(256), 0, stream, p0, p1, rows, channels, eps, momentum,
                     save_mean, save_var, save_rstd, running_mean, running_var);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid1d((long long)rows * channels)), dim3(256), 0, stream, x,
                     (long long)ldx, rows, channels, save_mean, save_rstd, gamma, beta, act, y, (long long)ldy);
  return me::check_launch("bn_train_fwd");
}

static int launch_bn_train_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int32_t rows, const int32_t* rows_dev,
                               int32_t channels, const float* gamma, const float* beta, const float* save_mean,
                               const float* save_rstd, int32_t act, float* dx, int64_t lddx, float* dgamma, float* dbeta,
                               void* workspace, void* stream_);

int me_bn_train_bwd_f32(const float* x, int64_t ldx, const float* dy, int64_t lddy, int32_t rows, int32_t channels,
                        const float* gamma, const float* beta, const float* save_mean, const float* save_rstd,
                        int32_t act, float* dx, int64_t lddx, float* dgamma, float* dbeta, void* workspace,
                        void* stream) {
  return launch_bn_train_bwd(x, ldx, dy, lddy, rows, nullptr, channels, gamma, beta, save_mean, save_rstd, act, dx, lddx, dgamma,
                             dbeta, workspace, stream);
}

int me_bn_train_bwd_dev_f32(const float* x, int64_t ldx, const float* dy, int64_t lddy, int32_t rows_cap, const int32_t* rows_dev,
                            int32_t channels, const float* gamma, const float* beta, const float* save_mean,
                            const float* save_rstd, int32_t act, float* dx, int64_t lddx, float* dgamma, float* dbeta,
                            void* workspace, void* stream) {
  ME_REQUIRE(rows_dev != nullptr, ME_E_NULLPTR, "me_bn_train_bwd_dev_f32: null row count");
  return launch_bn_train_bwd(x, ldx, dy, lddy, rows_cap, rows_dev, channels, gamma, beta, save_mean, save_rstd, act, dx, lddx,
                             dgamma, dbeta, workspace, stream);
}

static int launch_bn_train_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int32_t rows, const int32_t* rows_dev,
                               int32_t channels, const float* gamma, const float* beta, const float* save_mean,
                               const float* save_rstd, int32_t act, float* dx, int64_t lddx, float* dgamma, float* dbeta,
                               void* workspace, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(x && dy && gamma && beta && save_mean && save_rstd && workspace, ME_E_NULLPTR,
             "me_bn_train_bwd_f32: null pointer");
  ME_REQUIRE(rows > 0 && channels > 0, ME_E_BADARG, "me_bn_train_bwd_f32: bad dimensions");
  float* p0 = reinterpret_cast<float*>(workspace);
  float* p1 = p0 + (long long)BN_CHUNKS * channels;
  const unsigned cb = (channels + 3) / 4;   // the finishing kernels: a wave per channel
  hipLaunchKernelGGL(bn_partial_kernel, dim3((channels + 63) / 64, BN_CHUNKS), dim3(256), 0, stream, x, (long long)ldx, rows, channels,
                     dy, (long long)lddy, save_mean, save_rstd, gamma, beta, act, p0, p1, rows_dev);
  hipLaunchKernelGGL(bn_reduce_partials_kernel, dim3(cb), dim3(256), 0, stream, p0, p1, channels);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid1d((long long)rows * channels)), dim3(256), 0, stream, x,
                     (long long)ldx, dy, (long long)lddy, rows, channels, save_mean, save_rstd, gamma, beta, act, p0,
                     p1, dgamma, dbeta, dx, (long long)lddx, rows_dev);
  return me::check_launch("bn_train_bwd");
}

// Row chunks of the affine backward.  Aligned operands with C % 4 == 0 take affine_bwd_rows_kernel (Q lanes across a row's
// channel quads, RL = 256 / Q rows side by side): ~8 workgroups per CU (2048), every chunk a multiple of 4 * RL rows and at
// least 8 * RL of them (two trips of the four-deep load loop), at most 2048 chunks.  The others keep the 64-channel column
// blocks with rows / 256 chunks.
static int affine_quads(int channels) {
  int q = 8;
  while (q < 256 && q * 4 < channels) q *= 2;
  return q;  // 8 (<= 32 channels), 16, 32, 64, 128, 256 (>= 1024 channels per column block)
}

static void affine_plan(int rows, int channels, bool rows_kernel, int* chunks, int* per) {
  if (!rows_kernel) {
    int c = rows / 256;
    if (c < 1) c = 1;
    if (c > 1024) c = 1024;
    *chunks = c;
    *per = (rows + c - 1) / c;
    return;
  }
  const int q = affine_quads(channels), rl = 256 / q;
  const int groups = (channels + 4 * q - 1) / (4 * q);
  const int unit = 4 * rl;
  static const int wgs = getenv("MILLIEYE_AFFINE_WGS") ? atoi(getenv("MILLIEYE_AFFINE_WGS")) : 2048;
  int want = (wgs + groups - 1) / groups;              // chunks for ~wgs workgroups
  int p = (rows + want - 1) / want;
  if (p < 2 * unit) p = 2 * unit;
  p = (p + unit - 1) / unit * unit;
  int c = (rows + p - 1) / p;
  if (c > 2048) {
    c = 2048;
    p = ((rows + c - 1) / c + unit - 1) / unit * unit;
    c = (rows + p - 1) / p;
  }
  *chunks = c;
  *per = p;
}

// upper bound of the chunk count over both kernels (the workspace does not know the operands' alignment)
static int affine_chunks_max(int rows, int channels) {
  int c0, p0, c1, p1;
  affine_plan(rows, channels, false, &c0, &p0);
  affine_plan(rows, channels, true, &c1, &p1);
  return c0 > c1 ? c0 : c1;
}

int64_t me_affine_bwd_workspace_bytes(int32_t rows, int32_t channels) {
  return (int64_t)2 * affine_chunks_max(rows, channels) * channels * (int64_t)sizeof(float);
}

int me_affine_act_bwd_f32(const float* y, int64_t ldy, const float* dy, int64_t lddy, int32_t rows, int32_t channels,
                          const float* scale, const float* gamma, const float* beta, int32_t act, float* dc, int64_t lddc,
                          float* dshift, float* dgamma, void* workspace, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(y && dy && dc && workspace, ME_E_NULLPTR, "me_affine_act_bwd_f32: null pointer");
  ME_REQUIRE(rows > 0 && channels > 0, ME_E_BADARG, "me_affine_act_bwd_f32: bad dimensions");
  ME_REQUIRE(act == ME_ACT_LINEAR || act == ME_ACT_LEAKY, ME_E_BADARG, "me_affine_act_bwd_f32: activation %d", act);
  const bool v4 = channels % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0 && lddc % 4 == 0 && me::aligned16(y) &&
                  me::aligned16(dy) && me::aligned16(dc);
  static const int rows_env = getenv("MILLIEYE_AFFINE_ROWS") ? atoi(getenv("MILLIEYE_AFFINE_ROWS")) : 0;  // (A/B: 1 = contiguous rows)
  const bool rows_kernel = v4 && rows_env != 0;
  int chunks, per;
  affine_plan(rows, channels, rows_kernel, &chunks, &per);
  float* p0 = reinterpret_cast<float*>(workspace);
  float* p1 = p0 + (long long)chunks * channels;
  if (rows_kernel) {
    const int q = affine_quads(channels);
    const dim3 grid((channels + 4 * q - 1) / (4 * q), chunks);
#define ME_AFF(Q)                                                                                                         \
  hipLaunchKernelGGL(affine_bwd_rows_kernel<Q>, grid, dim3(256), 0, stream, y, (long long)ldy, dy, (long long)lddy, rows, \
                     channels, gamma, beta, act, p0, p1, per, scale, dc, (long long)lddc)
    switch (q) {
      case 8: ME_AFF(8); break;
      case 16: ME_AFF(16); break;
      case 32: ME_AFF(32); break;
      case 64: ME_AFF(64); break;
      case 128: ME_AFF(128); break;
      default: ME_AFF(256); break;
    }
#undef ME_AFF
  } else if (v4)
    hipLaunchKernelGGL(affine_bwd_partial_v4_kernel, dim3((channels + 63) / 64, chunks), dim3(256), 0, stream, y,
                       (long long)ldy, dy, (long long)lddy, rows, channels, gamma, beta, act, p0, p1, chunks, scale, dc,
                       (long long)lddc);
  else
    hipLaunchKernelGGL(affine_bwd_partial_kernel, dim3((channels + 63) / 64, chunks), dim3(256), 0, stream, y,
                       (long long)ldy, dy, (long long)lddy, rows, channels, gamma, beta, act, p0, p1, chunks, scale, dc,
                       (long long)lddc);
  // dshift == dgamma == NULL: the caller adds the partial rows later (me_affine_bwd_sums_f32, possibly on another stream)
  if (dshift || dgamma)
    hipLaunchKernelGGL(affine_bwd_reduce_kernel, dim3((channels + 63) / 64), dim3(1024), 0, stream, p0, p1, channels, chunks,
                       dshift, dgamma);
  return me::check_launch("affine_act_bwd");
}

int me_affine_bwd_sums_f32(void* workspace, int32_t rows, int32_t channels, int32_t vec4, float* dshift, float* dgamma, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(workspace && (dshift || dgamma), ME_E_NULLPTR, "me_affine_bwd_sums_f32: null pointer");
  ME_REQUIRE(rows > 0 && channels > 0, ME_E_BADARG, "me_affine_bwd_sums_f32: bad dimensions");
  static const int rows_env = getenv("MILLIEYE_AFFINE_ROWS") ? atoi(getenv("MILLIEYE_AFFINE_ROWS")) : 0;
  int chunks, per;
  affine_plan(rows, channels, vec4 != 0 && rows_env != 0, &chunks, &per);   // the first call's plan
  float* p0 = reinterpret_cast<float*>(workspace);
  float* p1 = p0 + (long long)chunks * channels;
  hipLaunchKernelGGL(affine_bwd_reduce_kernel, dim3((channels + 63) / 64), dim3(1024), 0, stream, p0, p1, channels, chunks,
                     dshift, dgamma);
  return me::check_launch("affine_bwd_sums");
}

int me_upsample2_bwd_f32(const float* dy, int64_t lddy, float* dx, int64_t lddx, int32_t n, int32_t h, int32_t w,
                         int32_t c, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(dy && dx, ME_E_NULLPTR, "me_upsample2_bwd_f32: null pointer");
  ME_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0, ME_E_BADARG, "me_upsample2_bwd_f32: bad dimensions");
  hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(grid1d((long long)n * h * w * c)), dim3(256), 0, stream, dy,
                     (long long)lddy, dx, (long long)lddx, n, h, w, c);
  return me::check_launch("upsample2_bwd_kernel");
}

int me_maxpool_bwd_f32(const float* x, int64_t ldx, const float* dy, int64_t lddy, float* dx, int64_t lddx, int32_t n,
                       int32_t h, int32_t w, int32_t c, int32_t size, int32_t stride, int32_t pad, int32_t zero_ext,
                       void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(x && dy && dx, ME_E_NULLPTR, "me_maxpool_bwd_f32: null pointer");
  ME_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && size >= 1 && stride >= 1, ME_E_BADARG, "me_maxpool_bwd_f32: bad dimensions");
  const int ext = zero_ext ? 1 : 0;
  const int ho = (h + ext + 2 * pad - size) / stride + 1, wo = (w + ext + 2 * pad - size) / stride + 1;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid1d((long long)n * h * w * c)), dim3(256), 0, stream, x, (long long)ldx,
                     dy, (long long)lddy, dx, (long long)lddx, n, h, w, c, size, stride, pad, zero_ext, ho, wo);
  return me::check_launch("maxpool_bwd_kernel");
}

int me_yolo_loss_bwd_f32(const float* raw, int64_t pitch, int32_t n, int32_t g, int32_t num_anchors, int32_t num_classes,
                         const uint8_t* obj_mask, const uint8_t* noobj_mask, const float* tx, const float* ty,
                         const float* tw, const float* th, const float* tcls, const float* tconf, float n_obj,
                         float n_noobj, float obj_scale, float noobj_scale, float grad_scale, float* draw,
                         int64_t dpitch, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(raw && obj_mask && noobj_mask && tx && ty && tw && th && tcls && tconf && draw, ME_E_NULLPTR,
             "me_yolo_loss_bwd_f32: null pointer");
  ME_REQUIRE(n > 0 && g > 0 && num_anchors > 0 && num_classes > 0, ME_E_BADARG, "me_yolo_loss_bwd_f32: bad dimensions");
  const long long work = (long long)n * num_anchors * g * g * (num_classes + 5);
  hipLaunchKernelGGL(yolo_loss_bwd_kernel, dim3(grid1d(work)), dim3(256), 0, stream, raw, (long long)pitch, n, g,
                     num_anchors, num_classes, obj_mask, noobj_mask, tx, ty, tw, th, tcls, tconf, n_obj, n_noobj,
                     obj_scale, noobj_scale, grad_scale, draw, (long long)dpitch, (const float*)nullptr, (const float*)nullptr);
  return me::check_launch("yolo_loss_bwd_kernel");
}

// the same pass with n_obj / n_noobj read from the forward's result[16] in device memory and the upstream gradient of the scalar
// loss from a device float (NULL = 1): no host value of the step enters the launch (captured step, millieye_amd/detector_graph.py)
int me_yolo_loss_bwd_dev_f32(const float* raw, int64_t pitch, int32_t n, int32_t g, int32_t num_anchors, int32_t num_classes,
                             const uint8_t* obj_mask, const uint8_t* noobj_mask, const float* tx, const float* ty,
                             const float* tw, const float* th, const float* tcls, const float* tconf,
                             const float* result_device, float obj_scale, float noobj_scale, const float* grad_scale_device,
                             float* draw, int64_t dpitch, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(raw && obj_mask && noobj_mask && tx && ty && tw && th && tcls && tconf && draw && result_device, ME_E_NULLPTR,
             "me_yolo_loss_bwd_dev_f32: null pointer");
  ME_REQUIRE(n > 0 && g > 0 && num_anchors > 0 && num_classes > 0, ME_E_BADARG, "me_yolo_loss_bwd_dev_f32: bad dimensions");
  const long long work = (long long)n * num_anchors * g * g * (num_classes + 5);
  hipLaunchKernelGGL(yolo_loss_bwd_kernel, dim3(grid1d(work)), dim3(256), 0, stream, raw, (long long)pitch, n, g,
                     num_anchors, num_classes, obj_mask, noobj_mask, tx, ty, tw, th, tcls, tconf, 1.f, 1.f,
                     obj_scale, noobj_scale, 1.f, draw, (long long)dpitch, result_device, grad_scale_device);
  return me::check_launch("yolo_loss_bwd_kernel");
}

int me_act_bwd_f32(const float* y, int64_t ldy, const float* dy, int64_t lddy, float* dx, int64_t lddx, int64_t rows,
                   int32_t channels, int32_t act, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(y && dy && dx, ME_E_NULLPTR, "me_act_bwd_f32: null pointer");
  ME_REQUIRE(rows >= 0 && channels > 0, ME_E_BADARG, "me_act_bwd_f32: bad dimensions");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid1d(rows * channels)), dim3(256), 0, stream, y, (long long)ldy, dy,
                     (long long)lddy, dx, (long long)lddx, (long long)rows, channels, act);
  return me::check_launch("act_bwd_kernel");
}

int me_conv_wgrad_f32(const float* x, int64_t x_pitch, const float* dy, int64_t dy_pitch, float* dw, int32_t n,
                      int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t ksize, int32_t stride, int32_t pad,
                      void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(x && dy && dw, ME_E_NULLPTR, "me_conv_wgrad_f32: null pointer");
  ME_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && ksize >= 1 && ksize <= 7 && stride >= 1 && pad >= 0,
             ME_E_BADARG, "me_conv_wgrad_f32: bad dimensions");
  const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (w + 2 * pad - ksize) / stride + 1;
  dim3 grid((cin + 63) / 64, (cout + 63) / 64, ksize * ksize);
  hipLaunchKernelGGL(conv_wgrad_kernel, grid, dim3(256), 0, stream, x, (long long)x_pitch, dy, (long long)dy_pitch, dw,
                     n, h, w, cin, cout, ksize, stride, pad, ho, wo);
  return me::check_launch("conv_wgrad_kernel");
}

// pixel slices so that the grid has ~2048 workgroups; each slice a multiple of 16 pixels and at least WGRAD_MIN_PX of them
// (round 3: the 1x1 layers were cut into up to 256 slices of ~96 pixels - six pipeline stages per workgroup, then a reduction
// over 256 slabs; MILLIEYE_WGRAD_MINPX is the tuning switch that found the floor)
static int wgrad_min_px() {
  static const int v = [] {
    const char* e = getenv("MILLIEYE_WGRAD_MINPX");
    const int x = e ? atoi(e) : 0;
    return x > 0 ? x : 256;
  }();
  return v;
}

// tile choice (output channels x input channels per workgroup), measured on the Darknet-53 shapes at batch 8
// (tools/wgrad_bench.py): 128 x 128 wins where the pixel reduction is long and both dimensions fill the tile (52^2 x 8:
// 153 -> 136 us); on the shorter reductions (26^2, 13^2), the 1x1 filters and the half-wide tiles the fewer, fatter waves
// lose more latency hiding than the halved L2 traffic buys.  MILLIEYE_WGRAD_TILE=64 / 128 forces one kernel (A/B switch).
static void wgrad_tile(long long P, int cin, int cout, int ks, int& tco, int& tci) {
  static const int force = [] {
    const char* e = getenv("MILLIEYE_WGRAD_TILE");
    return e ? atoi(e) : 0;
  }();
  tco = tci = 64;
  if (force == 64) return;
  if (force == 128) {
    tco = cout >= 128 ? 128 : 64;
    tci = cin >= 128 ? 128 : 64;
  } else if (ks >= 3 && P >= 16384 && cin >= 128 && cout >= 128) {
    tco = tci = 128;
  }
}

static int wgrad_target_wgs(int tco, int tci) {
  static const int v = [] {
    const char* e = getenv("MILLIEYE_WGRAD_WGS");
    return e ? atoi(e) : 0;
  }();
  if (v > 0) return v;
  return (tco == 128 && tci == 128) ? 1024 : 2048;
}

static int wgrad_splits(long long P, int cin, int cout, int ks) {
  int tco, tci;
  wgrad_tile(P, cin, cout, ks, tco, tci);
  const long long tiles = (long long)((cin + tci - 1) / tci) * ((cout + tco - 1) / tco) * ks * ks;
  const int target = wgrad_target_wgs(tco, tci);
  long long s = (target + tiles - 1) / tiles;
  const long long max_s = (P + wgrad_min_px() - 1) / wgrad_min_px();
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return (int)s;
}

// me_conv_wgrad_h16: tile and pixel-slice count.  Measured on the Darknet-53 shapes at batch 8 (tools/wgrad_bench.py 8 bf16 under
// MILLIEYE_WGRAD_WGS, profiles/r05_wgrad16_wgs_sweep.txt): the 128 x 128 tiles are fastest at ~430 - 580 workgroups (two per CU: the
// slabs a slice writes and the sum re-reads are a third of the pass), the 64 x 64 tiles at ~1000.
static bool wgrad_h16_big(int cin, int cout) { return cin >= 128 && cout >= 128 && cin % 128 == 0 && cout % 128 == 0; }

static int wgrad_splits_h16(long long P, int cin, int cout, int ks) {
  const int t = wgrad_h16_big(cin, cout) ? 128 : 64;
  const long long tiles = (long long)((cin + t - 1) / t) * ((cout + t - 1) / t) * ks * ks;
  static const int env = getenv("MILLIEYE_WGRAD16_WGS") ? atoi(getenv("MILLIEYE_WGRAD16_WGS")) : 0;
  const int target = env > 0 ? env : (t == 128 ? 512 : 1024);
  long long s = (target + tiles - 1) / tiles;
  const long long max_s = (P + wgrad_min_px() - 1) / wgrad_min_px();
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return (int)s;
}

static bool stem_wgrad_shape(int cin, int cout, int ksize) { return cin == 3 && ksize == 3 && cout % 4 == 0; }

extern "C++" {
namespace me_wg9 {  // wgrad9.hip: all nine taps of a (cout, cin) tile per workgroup (3x3, stride 1, pad 1)
bool shape(int cin, int cout, int ksize, int stride, int pad, int* tco, int* tci);
int splits(int n, int h, int w, int cin, int cout, int* per);
int launch(const float* x, long long xp, const float* dy, long long dyp, float* slabs, int n, int h, int w, int cin, int cout,
           hipStream_t stream, int* splits_out);
}  // namespace me_wg9
}  // extern "C++"

int64_t me_conv_wgrad_workspace_bytes(int32_t n, int32_t ho, int32_t wo, int32_t cin, int32_t cout, int32_t ksize) {
  if (stem_wgrad_shape(cin, cout, ksize)) return (int64_t)SW_BLOCKS * cout * 27 * (int64_t)sizeof(float);
  int s = wgrad_splits((long long)n * ho * wo, cin, cout, ksize);
  const int s16 = wgrad_splits_h16((long long)n * ho * wo, cin, cout, ksize);   // (the 16-bit form slices by its own tiles)
  if (s16 > s) s = s16;
  int64_t need = s > 1 ? (int64_t)s * cout * ksize * ksize * cin * (int64_t)sizeof(float) : 0;
  if (ksize == 3) {  // the nine-tap kernel (stride 1: the input map is the output map) always goes through slabs
    const int s9 = me_wg9::splits(n, ho, wo, cin, cout, nullptr);
    const int64_t need9 = (int64_t)s9 * cout * 9 * cin * (int64_t)sizeof(float);
    if (need9 > need) need = need9;
  }
  return need;
}

static int wgrad_mfma(const float* x, int64_t x_pitch, const float* dy, int64_t dy_pitch, float* dw, int32_t n, int32_t h,
                      int32_t w, int32_t cin, int32_t cout, int32_t ksize, int32_t stride, int32_t pad, void* workspace,
                      int64_t workspace_bytes, void* stream_, int oihw) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(x && dy && dw, ME_E_NULLPTR, "me_conv_wgrad_mfma_f32: null pointer");
  ME_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && ksize >= 1 && ksize <= 7 && stride >= 1 && pad >= 0,
             ME_E_BADARG, "me_conv_wgrad_mfma_f32: bad dimensions");
  const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (w + 2 * pad - ksize) / stride + 1;
  const long long P = (long long)n * ho * wo;
  ME_REQUIRE(P < (1ll << 31), ME_E_TOOBIG, "me_conv_wgrad_mfma_f32: too many output pixels");
  const long long count = (long long)cout * ksize * ksize * cin;
  if (stem_wgrad_shape(cin, cout, ksize) && dy_pitch % 4 == 0 && me::aligned16(dy) && workspace &&
      workspace_bytes >= (int64_t)SW_BLOCKS * count * (int64_t)sizeof(float) && (cout + 31) / 32 < 65536) {
    float* part = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(SW_BLOCKS, (cout + 31) / 32), dim3(256), 0, stream, x, (long long)x_pitch, dy,
                       (long long)dy_pitch, part, n, h, w, cout, stride, pad, ho, wo);
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((unsigned)((count + 15) / 16)), dim3(256), 0, stream, part, dw, (int)count,
                       SW_BLOCKS, oihw);
    return me::check_launch("stem_wgrad_kernel");
  }
  int splits = wgrad_splits(P, cin, cout, ksize);
  bool done9 = false;
  {
    int tco9, tci9;
    if (me_wg9::shape(cin, cout, ksize, stride, pad, &tco9, &tci9) && h == ho && w == wo && workspace) {
      const int s9 = me_wg9::splits(n, h, w, cin, cout, nullptr);
      if (workspace_bytes >= (int64_t)s9 * count * (int64_t)sizeof(float)) {
        int sp = 0;
        const int rc9 = me_wg9::launch(x, x_pitch, dy, dy_pitch, reinterpret_cast<float*>(workspace), n, h, w, cin, cout, stream, &sp);
        if (rc9 > 0) return rc9;
        if (rc9 == 0) {
          splits = sp;
          done9 = true;
        }
      }
    }
  }
  if (splits > 1 && (!workspace || workspace_bytes < (int64_t)splits * count * (int64_t)sizeof(float))) splits = 1;
  ME_REQUIRE(!oihw || ksize == 1 || (workspace && workspace_bytes >= count * (int64_t)sizeof(float)), ME_E_BADARG,
             "me_conv_wgrad_mfma_oihw_f32: needs a workspace of at least one slab (%lld bytes)", count * 4ll);
  int per = (int)((P + splits - 1) / splits);
  per = (per + 15) & ~15;
  ME_REQUIRE((long long)ksize * ksize * splits < 65536, ME_E_TOOBIG, "me_conv_wgrad_mfma_f32: grid too large");
  const bool via_ws = done9 || splits > 1 || (oihw && ksize > 1);  // (a 1x1 filter's OHWI and OIHW layouts coincide)
  float* out = via_ws ? reinterpret_cast<float*>(workspace) : dw;
  const bool vec = (cin % 4 == 0) && (cout % 4 == 0) && (x_pitch % 4 == 0) && (dy_pitch % 4 == 0) && me::aligned16(x) &&
                   me::aligned16(dy);
  int tco, tci;
  wgrad_tile(P, cin, cout, ksize, tco, tci);
  if (!vec) tco = tci = 64;  // (the split count was sized for the larger tile: correct, a few workgroups more)
  dim3 grid((cin + tci - 1) / tci, (cout + tco - 1) / tco, ksize * ksize * splits);
  static const int depth = [] {   // stages of loads in flight per lane (round 5; 0 / 1 = the round-3 kernels, A/B)
    const char* e = getenv("MILLIEYE_WGRAD_DEPTH");
    return e ? atoi(e) : 4;
  }();
  if (done9) {
    // (slabs written by the nine-tap kernel: straight to the reduction)
  } else
#define ME_WG_TILE(A, B)                                                                                              \
  hipLaunchKernelGGL((conv_wgrad_tile_kernel<A, B>), grid, dim3(256), 0, stream, x, (long long)x_pitch, dy,           \
                     (long long)dy_pitch, out, n, h, w, cin, cout, ksize, stride, pad, ho, wo, splits, per)
#define ME_WG_PIPE(A, B, D)                                                                                           \
  hipLaunchKernelGGL((conv_wgrad_pipe_kernel<A, B, D>), grid, dim3(256), 0, stream, x, (long long)x_pitch, dy,        \
                     (long long)dy_pitch, out, n, h, w, cin, cout, ksize, stride, pad, ho, wo, splits, per)
  if (vec && depth > 1 && cin <= 32 && tco == 64) {
    grid.x = 1;
    ME_WG_PIPE(64, 32, 4);   // half-wide tile, the wave pair of a co block splits the k-steps
  } else if (vec && depth > 4 && tco == 128 && tci == 128)   // (depth 5+: the four-deep pipeline on the full-width tiles, A/B only -
    ME_WG_PIPE(128, 128, 4);                                  //  measured neutral on the 1x1 filters, slower on 128 x 128)
  else if (vec && depth > 4 && tco == 128)
    ME_WG_PIPE(128, 64, 4);
  else if (vec && depth > 4 && tci == 128)
    ME_WG_PIPE(64, 128, 4);
  else if (vec && depth > 4)
    ME_WG_PIPE(64, 64, 4);
  else if (vec && tco == 128 && tci == 128)
    ME_WG_TILE(128, 128);
  else if (vec && tco == 128)
    ME_WG_TILE(128, 64);
  else if (vec && tci == 128)
    ME_WG_TILE(64, 128);
  else if (vec) {
    static const int abl = [] {
      const char* on = getenv("MILLIEYE_ABLATION");
      const char* e = getenv("MILLIEYE_WGRAD_ABL");
      return (on && on[0] == '1' && e) ? atoi(e) : 0;
    }();
#define ME_WG_ABL(A)                                                                                                   \
  hipLaunchKernelGGL((conv_wgrad_mfma_kernel<true, A>), grid, dim3(256), 0, stream, x, (long long)x_pitch, dy,          \
                     (long long)dy_pitch, out, n, h, w, cin, cout, ksize, stride, pad, ho, wo, splits, per)
    if (abl == 1) ME_WG_ABL(1);
    else if (abl == 2) ME_WG_ABL(2);
    else ME_WG_ABL(0);
#undef ME_WG_ABL
  }
#undef ME_WG_TILE
#undef ME_WG_PIPE
  else
    hipLaunchKernelGGL(conv_wgrad_mfma_kernel<false>, grid, dim3(256), 0, stream, x, (long long)x_pitch, dy,
                       (long long)dy_pitch, out, n, h, w, cin, cout, ksize, stride, pad, ho, wo, splits, per);
  int rc = me::check_launch("conv_wgrad_mfma_kernel");
  if (rc || !via_ws) return rc;
  const float* slabs = reinterpret_cast<const float*>(workspace);
  if (oihw && ksize > 1 && cout < 65536) {
    const int kk = ksize * ksize;
    hipLaunchKernelGGL(conv_wgrad_reduce_oihw_kernel, dim3((cin + 63) / 64, cout), dim3(256), 64 * (kk | 1) * sizeof(float),
                       stream, slabs, dw, count, splits, kk, cin, ((cin & 3) == 0 && me::aligned16(workspace)) ? 1 : 0);
  } else if (!(oihw && ksize > 1) && count % 4 == 0 && me::aligned16(dw) && me::aligned16(workspace)) {
    hipLaunchKernelGGL(conv_wgrad_reduce_v4_kernel, dim3(grid1d(count / 4)), dim3(256), 0, stream, slabs, dw, count / 4, splits);
  } else {
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(grid1d(count)), dim3(256), 0, stream, slabs, dw, count, splits,
                       (oihw && ksize > 1) ? ksize * ksize : 0, cin);
  }
  return me::check_launch("conv_wgrad_reduce_kernel");
}

int me_conv_wgrad_mfma_f32(const float* x, int64_t x_pitch, const float* dy, int64_t dy_pitch, float* dw, int32_t n,
                           int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t ksize, int32_t stride, int32_t pad,
                           void* workspace, int64_t workspace_bytes, void* stream) {
  return wgrad_mfma(x, x_pitch, dy, dy_pitch, dw, n, h, w, cin, cout, ksize, stride, pad, workspace, workspace_bytes, stream, 0);
}

int me_conv_wgrad_mfma_oihw_f32(const float* x, int64_t x_pitch, const float* dy, int64_t dy_pitch, float* dw, int32_t n,
                                int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t ksize, int32_t stride, int32_t pad,
                                void* workspace, int64_t workspace_bytes, void* stream) {
  return wgrad_mfma(x, x_pitch, dy, dy_pitch, dw, n, h, w, cin, cout, ksize, stride, pad, workspace, workspace_bytes, stream, 1);
}

// 16-bit operands -> float32 gradient in the parameter's own OIHW layout (or OHWI): conv_wgrad_tile_h16_kernel + the slab sums above.
int me_conv_wgrad_h16(const void* x_, int64_t x_pitch, const void* dy_, int64_t dy_pitch, float* dw, int32_t n, int32_t h, int32_t w,
                      int32_t cin, int32_t cout, int32_t ksize, int32_t stride, int32_t pad, void* workspace, int64_t workspace_bytes,
                      int32_t oihw, int32_t half_type, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const unsigned short* x = reinterpret_cast<const unsigned short*>(x_);
  const unsigned short* dy = reinterpret_cast<const unsigned short*>(dy_);
  ME_REQUIRE(x && dy && dw, ME_E_NULLPTR, "me_conv_wgrad_h16: null pointer");
  ME_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && ksize >= 1 && ksize <= 7 && stride >= 1 && pad >= 0,
             ME_E_BADARG, "me_conv_wgrad_h16: bad dimensions");
  ME_REQUIRE(half_type == 0 || half_type == 1, ME_E_BADARG, "me_conv_wgrad_h16: half_type must be 0 (bf16) or 1 (f16)");
  ME_REQUIRE(cin % 8 == 0 && cout % 8 == 0 && x_pitch % 8 == 0 && dy_pitch % 8 == 0 && me::aligned16(x) && me::aligned16(dy),
             ME_E_ALIGN, "me_conv_wgrad_h16: channels and pitches must be multiples of 8 and the tensors 16-byte aligned");
  const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (w + 2 * pad - ksize) / stride + 1;
  const long long P = (long long)n * ho * wo;
  ME_REQUIRE(P < (1ll << 31), ME_E_TOOBIG, "me_conv_wgrad_h16: too many output pixels");
  const long long count = (long long)cout * ksize * ksize * cin;
  int splits = wgrad_splits_h16(P, cin, cout, ksize);
  if (splits > 1) {   // as many pixel slices as the workspace holds slabs for
    const long long fit = workspace ? workspace_bytes / (count * (int64_t)sizeof(float)) : 0;
    if (fit < splits) splits = fit < 1 ? 1 : (int)fit;
  }
  ME_REQUIRE(!oihw || ksize == 1 || (workspace && workspace_bytes >= count * (int64_t)sizeof(float)), ME_E_BADARG,
             "me_conv_wgrad_h16: the OIHW form needs a workspace of at least one slab (%lld bytes)", count * 4ll);
  int per = (int)((P + splits - 1) / splits);
  per = (per + 31) & ~31;   // whole 32-pixel stages
  ME_REQUIRE((long long)ksize * ksize * splits < 65536, ME_E_TOOBIG, "me_conv_wgrad_h16: grid too large");
  const bool via_ws = splits > 1 || (oihw && ksize > 1);
  float* out = via_ws ? reinterpret_cast<float*>(workspace) : dw;
  const bool big = wgrad_h16_big(cin, cout);   // 128 x 128: half the LDS reads per product
  const int tco = big ? 128 : 64, tci = big ? 128 : 64;
  const dim3 grid((cin + tci - 1) / tci, (cout + tco - 1) / tco, ksize * ksize * splits);
  static const int tr_env = getenv("MILLIEYE_WGRAD16_TR") ? atoi(getenv("MILLIEYE_WGRAD16_TR")) : 1;  // (A/B: 0 = 2-byte LDS reads)
#define ME_WG16(A, B, F, T)                                                                                                \
  hipLaunchKernelGGL((conv_wgrad_tile_h16_kernel<A, B, F, T>), grid, dim3(256), 0, stream, x, (long long)x_pitch, dy,       \
                     (long long)dy_pitch, out, n, h, w, cin, cout, ksize, stride, pad, ho, wo, splits, per)
#define ME_WG16_T(A, B, F)     \
  do {                         \
    if (tr_env)                \
      ME_WG16(A, B, F, 1);     \
    else                       \
      ME_WG16(A, B, F, 0);     \
  } while (0)
  if (big) {
    if (half_type) ME_WG16_T(128, 128, 1);
    else ME_WG16_T(128, 128, 0);
  } else {
    if (half_type) ME_WG16_T(64, 64, 1);
    else ME_WG16_T(64, 64, 0);
  }
#undef ME_WG16_T
#undef ME_WG16
  int rc = me::check_launch("conv_wgrad_tile_h16_kernel");
  if (rc || !via_ws) return rc;
  const float* slabs = reinterpret_cast<const float*>(workspace);
  if (oihw && ksize > 1 && cout < 65536) {
    const int kk = ksize * ksize;
    hipLaunchKernelGGL(conv_wgrad_reduce_oihw_kernel, dim3((cin + 63) / 64, cout), dim3(256), 64 * (kk | 1) * sizeof(float),
                       stream, slabs, dw, count, splits, kk, cin, ((cin & 3) == 0 && me::aligned16(workspace)) ? 1 : 0);
  } else if (!(oihw && ksize > 1) && count % 4 == 0 && me::aligned16(dw) && me::aligned16(workspace)) {
    hipLaunchKernelGGL(conv_wgrad_reduce_v4_kernel, dim3(grid1d(count / 4)), dim3(256), 0, stream, slabs, dw, count / 4, splits);
  } else {
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(grid1d(count)), dim3(256), 0, stream, slabs, dw, count, splits,
                       (oihw && ksize > 1) ? ksize * ksize : 0, cin);
  }
  return me::check_launch("conv_wgrad_reduce_kernel");
}

static int launch_roi_bwd(const float* gout, const float* rois, int32_t k, int32_t n, int32_t h, int32_t w, int32_t c,
                          int32_t pooled, float scale, float* gmap, int64_t pitch, void* stream_, int ps,
                          const int32_t* k_dev = nullptr) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(gmap && (k == 0 || (gout && rois)), ME_E_NULLPTR, "me_roi_align_bwd_f32: null pointer");
  ME_REQUIRE(pooled == 7 && n > 0 && h > 0 && w > 0 && c > 0 && pitch >= c && k >= 0, ME_E_BADARG,
             "me_roi_align_bwd_f32: bad dimensions");
  ME_REQUIRE(!ps || c % 49 == 0, ME_E_BADARG, "me_ps_roi_align_bwd_f32: channels %% 49 != 0");
  if (k == 0) return 0;
  const long long total = (long long)k * (ps ? c / 49 : c) * 49;
  static const int xs_env = getenv("MILLIEYE_ROI_BWD_XSPLIT") ? atoi(getenv("MILLIEYE_ROI_BWD_XSPLIT")) : 1;   // (A/B: 0 = off)
  const int xsplit = (xs_env && n >= 4) ? 1 : 0;
  // the scatter through LDS when a frame's slice of the map fits (MILLIEYE_ROI_BWD_LDS=0: the global-atomics kernel)
  static const int lds_env = getenv("MILLIEYE_ROI_BWD_LDS") ? atoi(getenv("MILLIEYE_ROI_BWD_LDS")) : 1;
  const long long slice = (long long)h * w * (ps ? c / 49 : c) * (long long)sizeof(float);
  if (lds_env && slice <= 48 * 1024 && n <= 65535) {   // (frames ride in gridDim.y)
    if (ps) {
      hipLaunchKernelGGL(roi_bwd_lds_kernel<true>, dim3(49, n), dim3(ROI_LDS_THREADS), (size_t)slice, stream, gout, rois, k, c, h,
                         w, scale, gmap, (long long)pitch, k_dev);
    } else {
      static const int ns_env = getenv("MILLIEYE_ROI_BWD_SPLITS") ? atoi(getenv("MILLIEYE_ROI_BWD_SPLITS")) : 0;
      // one workgroup walks its RoIs' items in a chain of dependent loads (list -> RoI -> gradient): the time of a launch follows
      // the items per workgroup (1200 RoIs, 8 frames: 136 / 79 / 43 / 31 us at 4 / 8 / 16 / 32 slices per frame), so enough slices
      // to put a workgroup on every CU; a slice's write-out is at most h*w*c adds
      int nsplit = ns_env > 0 ? ns_env : 256 / n;
      nsplit = nsplit < 1 ? 1 : (nsplit > 64 ? 64 : nsplit);
      hipLaunchKernelGGL(roi_bwd_lds_kernel<false>, dim3(nsplit, n), dim3(ROI_LDS_THREADS), (size_t)slice, stream, gout, rois, k,
                         c, h, w, scale, gmap, (long long)pitch, k_dev);
    }
    return me::check_launch("roi_bwd_lds_kernel");
  }
  unsigned grid = grid1d(total);
  if (xsplit) {   // eight groups of workgroups, each scanning every index: a group needs total / 256 workgroup passes of its own
    long long per = (total + 255) / 256;
    if (per > 1024) per = 1024;
    grid = 8u * (unsigned)(per < 1 ? 1 : per);
  }
  hipLaunchKernelGGL(roi_bwd_kernel, dim3(grid), dim3(256), 0, stream, gout, rois, k, c, h, w, scale, gmap,
                     (long long)pitch, ps, k_dev, xsplit);
  return me::check_launch("roi_bwd_kernel");
}

int me_roi_align_bwd_f32(const float* grad_out, const float* rois, int32_t k, int32_t n, int32_t h, int32_t w,
                         int32_t c, int32_t pooled, float spatial_scale, float* grad_map, int64_t pitch,
                         void* stream) {
  return launch_roi_bwd(grad_out, rois, k, n, h, w, c, pooled, spatial_scale, grad_map, pitch, stream, 0);
}

int me_ps_roi_align_bwd_f32(const float* grad_out, const float* rois, int32_t k, int32_t n, int32_t h, int32_t w,
                            int32_t c, int32_t pooled, float spatial_scale, float* grad_map, int64_t pitch,
                            void* stream) {
  return launch_roi_bwd(grad_out, rois, k, n, h, w, c, pooled, spatial_scale, grad_map, pitch, stream, 1);
}

int me_roi_align_bwd_dev_f32(const float* grad_out, const float* rois, int32_t k_cap, const int32_t* k_dev, int32_t n, int32_t h,
                             int32_t w, int32_t c, int32_t pooled, float spatial_scale, float* grad_map, int64_t pitch,
                             void* stream) {
  ME_REQUIRE(k_dev != nullptr, ME_E_NULLPTR, "me_roi_align_bwd_dev_f32: null RoI count");
  return launch_roi_bwd(grad_out, rois, k_cap, n, h, w, c, pooled, spatial_scale, grad_map, pitch, stream, 0, k_dev);
}

int me_ps_roi_align_bwd_dev_f32(const float* grad_out, const float* rois, int32_t k_cap, const int32_t* k_dev, int32_t n, int32_t h,
                                int32_t w, int32_t c, int32_t pooled, float spatial_scale, float* grad_map, int64_t pitch,
                                void* stream) {
  ME_REQUIRE(k_dev != nullptr, ME_E_NULLPTR, "me_ps_roi_align_bwd_dev_f32: null RoI count");
  return launch_roi_bwd(grad_out, rois, k_cap, n, h, w, c, pooled, spatial_scale, grad_map, pitch, stream, 1, k_dev);
}

}  // extern "C"

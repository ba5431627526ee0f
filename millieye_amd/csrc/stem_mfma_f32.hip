// stem_mfma_f32.hip - the cin = 3 / 3x3 / stride-1 stem (Darknet stem, radar CNN stem) of the fp32 path on the matrix pipe.
//
// conv_stem3_f32 (conv.hip) spends 864 FMAs per pixel on the VALU: 0.30 ms at batch 32 for 775 MB of compulsory traffic
// (32 % of the HBM roofline).  v_mfma_f32_32x32x2_f32 runs at the VALU's FLOP rate but needs no operand shuffling: a wave
// builds the im2col rows of 32 pixels in registers (K = 27 taps padded to 28: lane half hh holds tap 2 s + hh of step s, read
// straight from the frame, next block prefetched) and 14 MFMAs produce 32 pixels x 32 channels in exact fp32 FMA chains
// (the numerics class of the CPU reference).  The accumulator block has lane = output channel: every store instruction
// writes two 128-byte rows, no LDS involved.
#include <stdlib.h>

#include "conv32_common.h"

namespace {

struct StemArgs32 {
  ConvP c;
  unsigned w_m, w_s, hw_m, hw_s;  // magic division by W and by H * W
};

__device__ __forceinline__ unsigned udiv_magic(unsigned n, unsigned m, unsigned s) { return (__umulhi(n, m) + n) >> s; }

__global__ __launch_bounds__(256) void conv_stem3_mfma_f32(StemArgs32 a) {
  const ConvP& p = a.c;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const int H = p.h, W = p.w, hw = H * W;
  const int co = blockIdx.y * 32 + r32;  // cout % 32 == 0

  float wreg[14];
  int toff[14];
  unsigned tdy = 0, tdx = 0, tok = 0;
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    const int k = 2 * s + hh;
    const int tap = k / 3, c = k - tap * 3;
    const int dy = tap / 3, dx = tap - dy * 3;
    const bool ok = k < 27;
    wreg[s] = ok ? p.wgt[(long long)co * 27 + k] : 0.f;
    toff[s] = !ok ? 0 : (p.x_nchw ? (c * hw + (dy - 1) * W + (dx - 1)) : (int)(((dy - 1) * W + (dx - 1)) * p.x_pitch + c));
    tdy |= (unsigned)(ok ? dy : 0) << (2 * s);
    tdx |= (unsigned)(ok ? dx : 0) << (2 * s);
    tok |= (unsigned)ok << s;
  }
  const float sc = p.scale[co], sh = p.shift[co];
  const int nblk = (p.M + 31) >> 5;
  float raw[14];
  auto fetch = [&](int blk) {
    const int m = blk * 32 + r32;
    const bool live = blk < nblk && m < p.M;
    const unsigned mm = live ? (unsigned)m : 0u;
    const unsigned n = udiv_magic(mm, a.hw_m, a.hw_s);
    const unsigned rem = mm - n * (unsigned)hw;
    const unsigned y = udiv_magic(rem, a.w_m, a.w_s);
    const unsigned x = rem - y * (unsigned)W;
    const unsigned vy = (y > 0 ? 1u : 0u) | 2u | (y + 1 < (unsigned)H ? 4u : 0u);
    const unsigned vx = (x > 0 ? 1u : 0u) | 2u | (x + 1 < (unsigned)W ? 4u : 0u);
    const long long base = p.x_nchw ? ((long long)n * 3 * hw + (long long)y * W + x)
                                    : ((long long)(n * (unsigned)hw + y * (unsigned)W + x) * p.x_pitch);
    const float* px = p.x + base;
#pragma unroll
    for (int s = 0; s < 14; ++s) {
      const bool ok = live && ((tok >> s) & 1u) && ((vy >> ((tdy >> (2 * s)) & 3u)) & 1u) && ((vx >> ((tdx >> (2 * s)) & 3u)) & 1u);
      raw[s] = ok ? px[toff[s]] : 0.f;
    }
  };
  const int stride = gridDim.x * 4;
  int blk = blockIdx.x * 4 + wave;
  fetch(blk);
  for (; blk < nblk; blk += stride) {
    float cur[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) cur[s] = raw[s];
    fetch(blk + stride);  // in flight behind the 14 MFMAs and the stores of this block
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 14; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[s], wreg[s], acc, 0, 0, 0);
    // lane = output channel, register e = pixel (e & 3) + 8 (e >> 2) + 4 hh of the block
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const long long m = (long long)blk * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
      if (m < p.M) p.y[m * p.y_pitch + co] = apply_act(acc[e] * sc + sh, p.act);
    }
  }
}

// Row-staged version for NCHW frames (the network input): a workgroup owns RY consecutive image rows.  Its input window -
// 3 channels x (RY + 2) rows x (W + 2) pixels, zero halo - is brought into LDS with coalesced loads (0.8 loads per lane and
// 32-pixel block instead of 14 predicated gathers), and the im2col rows come out of LDS with one add + ds_read_b32 per tap:
// no bounds logic, no 64-bit addresses in the block loop - 156 VGPRs (3 waves / SIMD) became ~80, and the texture addresser
// no longer paces the kernel (conv_stem3_mfma_f32 above: 3.1 TB/s of the 6.8 TB/s a plain fill reaches on this GPU).
constexpr int kStemRows = 4;

__global__ __launch_bounds__(256) void conv_stem3_rows_f32(ConvP p) {
  extern __shared__ __attribute__((aligned(16))) float win[];  // [3][RY + 2][WP]
  constexpr int RY = kStemRows;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const int H = p.h, W = p.w, WP = W + 2;
  const int groups = (H + RY - 1) / RY;
  const int n = blockIdx.x / groups, y0 = (blockIdx.x - n * groups) * RY;
  const int co = blockIdx.y * 32 + r32;  // cout % 32 == 0

  // stage the window: per 256-pixel column chunk the loads of all 3 (RY + 2) planes are in flight together (one memory latency
  // per chunk - a load -> store -> next load loop would serialise ~20 latencies per workgroup)
  const float* __restrict__ img = p.x + (long long)n * 3 * H * W;
  for (int x0 = 0; x0 < WP; x0 += 256) {
    const int xx = x0 + (int)threadIdx.x;
    const bool col_ok = xx >= 1 && xx <= W;
    float v[3 * (RY + 2)];
#pragma unroll
    for (int plane = 0; plane < 3 * (RY + 2); ++plane) {  // plane = c * (RY + 2) + ry
      const int c = plane / (RY + 2), ry = plane % (RY + 2);
      const int iy = y0 + ry - 1;
      v[plane] = (col_ok && (unsigned)iy < (unsigned)H) ? img[((long long)c * H + iy) * W + (xx - 1)] : 0.f;
    }
    if (xx < WP) {
#pragma unroll
      for (int plane = 0; plane < 3 * (RY + 2); ++plane) win[plane * WP + xx] = v[plane];
    }
  }

  float wreg[14];
  int toff[14];  // LDS float offset of this lane's tap of step s, relative to (row 0 of the block, pixel x)
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    const int k = 2 * s + hh;
    const int tap = k / 3, c = k - tap * 3;
    const int dy = tap / 3, dx = tap - dy * 3;
    const bool ok = k < 27;
    wreg[s] = ok ? p.wgt[(long long)co * 27 + k] : 0.f;
    toff[s] = ok ? (c * (RY + 2) + dy) * WP + dx : 0;
  }
  const float sc = p.scale[co], sh = p.shift[co];
  const float slope = p.act == ME_ACT_LEAKY ? 0.1f : 1.0f;
  __syncthreads();

  const int nbx = (W + 31) >> 5;
  const int rows = H - y0 < RY ? H - y0 : RY;
  for (int b = wave; b < rows * nbx; b += 4) {
    const int ry = b / nbx, bx = b - ry * nbx;
    const int x = bx * 32 + r32;
    const int xs = x < W ? x : W - 1;  // ragged last block: lanes beyond the row read a valid pixel, their results are not stored
    const int base = ry * WP + xs;
    float cur[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) cur[s] = win[base + toff[s]];
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 14; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[s], wreg[s], acc, 0, 0, 0);
    // lane = output channel, register e = pixel (e & 3) + 8 (e >> 2) + 4 hh of the block
    const long long m0 = ((long long)n * H + (y0 + ry)) * W + bx * 32 + 4 * hh;
    float* yp = p.y + m0 * p.y_pitch + co;
    const long long ystep = p.y_pitch;
    const int left = W - bx * 32 - 4 * hh;  // pixels of this lane half's first column still inside the row
    if (W - bx * 32 >= 32 && p.act != ME_ACT_SIGMOID) {  // whole block, leaky / linear: no per-pixel tests, leaky as max(t, 0.1 t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float t = acc[e] * sc + sh;
        *yp = fmaxf(t, t * slope);
        yp += (e & 3) == 3 ? 5 * ystep : ystep;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int px = (e & 3) + 8 * (e >> 2);
        if (px < left) *yp = apply_act(acc[e] * sc + sh, p.act);
        yp += (e & 3) == 3 ? 5 * ystep : ystep;
      }
    }
  }
}

void magic_u32s(unsigned d, unsigned* m, unsigned* s) {
  unsigned sh = 0;
  while ((1ull << sh) < d) ++sh;
  *s = sh;
  *m = (unsigned)(((1ull << 32) * ((1ull << sh) - d)) / d + 1);
}

}  // namespace

namespace me32 {

bool stem_mfma_eligible(const ConvP& p) {
  return p.cin == 3 && p.ks == 3 && p.stride == 1 && p.pad == 1 && p.cout % 32 == 0 && p.ups == 1 && !p.res &&
         (long long)p.n * p.h * p.w < (1ll << 31);
}

int launch_stem_mfma(const ConvP& p, hipStream_t stream) {
  const size_t win_bytes = (size_t)3 * (kStemRows + 2) * (p.w + 2) * sizeof(float);
  static const bool rows_ok = [] { const char* e = getenv("MILLIEYE_STEM_ROWS"); return !e || atoi(e) != 0; }();
  if (p.x_nchw && win_bytes <= 60 * 1024 && rows_ok) {  // row-staged version: NCHW frames up to ~1270 pixels wide
    const long long groups = (long long)p.n * ((p.h + kStemRows - 1) / kStemRows);
    if (groups < (1ll << 31)) {
      hipLaunchKernelGGL(conv_stem3_rows_f32, dim3((unsigned)groups, p.cout / 32), dim3(256), win_bytes, stream, p);
      return me::check_launch("conv_stem3_rows_f32");
    }
  }
  StemArgs32 a;
  a.c = p;
  magic_u32s((unsigned)p.w, &a.w_m, &a.w_s);
  magic_u32s((unsigned)(p.h * p.w), &a.hw_m, &a.hw_s);
  const int nblk = (p.M + 31) / 32;
  int grid = (nblk + 3) / 4;
  if (grid > 256 * 8) grid = 256 * 8;
  hipLaunchKernelGGL(conv_stem3_mfma_f32, dim3(grid, p.cout / 32), dim3(256), 0, stream, a);
  return me::check_launch("conv_stem3_mfma_f32");
}

}  // namespace me32

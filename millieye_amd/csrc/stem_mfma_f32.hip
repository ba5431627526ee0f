// stem_mfma_f32.hip - the cin = 3 / 3x3 / stride-1 stem (Darknet stem, radar CNN stem) of the fp32 path on the matrix pipe.
//
// conv_stem3_f32 (conv.hip) spends 864 FMAs per pixel on the VALU: 0.30 ms at batch 32 for 775 MB of compulsory traffic
// (32 % of the HBM roofline).  v_mfma_f32_32x32x2_f32 runs at the VALU's FLOP rate but needs no operand shuffling: a wave
// builds the im2col rows of 32 pixels in registers (K = 27 taps padded to 28: lane half hh holds tap 2 s + hh of step s, read
// straight from the frame, next block prefetched) and 14 MFMAs produce 32 pixels x 32 channels in exact fp32 FMA chains
// (the numerics class of the CPU reference).  The accumulator block has lane = output channel: every store instruction
// writes two 128-byte rows, no LDS involved.
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

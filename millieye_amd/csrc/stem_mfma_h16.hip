// stem_mfma_h16.hip - the cin = 3 stem of the 16-bit storage modes on the matrix pipe (gfx950).
//
// The VALU stem (conv_stem3_h16: one thread = one pixel x 32 channels = 864 FMAs) is VALU-bound: 0.23 ms at batch 32 for
// 420 MB of compulsory traffic (fp32 frames in, 16-bit NHWC out) = 1.8 TB/s.  Here a wave builds the im2col rows of 32 pixels
// in registers - K = 27 taps (ky, kx, c) padded to 32, each lane holds the 8 consecutive taps the MFMA operand layout wants,
// read straight from the frame (L1 / L2 serve the 9-fold overlap) and rounded to the storage type - and two
// v_mfma_f32_32x32x16 do what 27 x 32 FMAs per pixel did.  The 32 x 32 weights (2 KiB) live in registers for the whole
// kernel.  Epilogue: folded BN affine + LeakyReLU in fp32, one rounding, through a per-wave LDS transpose to 16-byte stores
// (64 lanes = 16 pixels x 64 contiguous bytes).
//
// Rounding points of the mode (restated by oracle/darknet_ref.py storage="bf16" / "f16"): the frame and the stem weights are
// rounded once to the storage type, accumulation is fp32.
#include <stdlib.h>

#include "conv16_common.h"

namespace {

struct StemArgs {
  Conv16P c;
  unsigned w_m, w_s, hw_m, hw_s;  // magic division by W and by H * W
};

__device__ __forceinline__ unsigned udiv_magic(unsigned n, unsigned m, unsigned s) { return (__umulhi(n, m) + n) >> s; }

template <int F16>
__global__ __launch_bounds__(256) void conv_stem3_mfma_h16(StemArgs a) {
  using frag = typename H16<F16>::v8;
  using elem16 = std::conditional_t<F16 != 0, _Float16, __bf16>;
  const Conv16P& p = a.c;
  __shared__ __attribute__((aligned(16))) float tbuf_all[4][32 * 36];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const int H = p.h, W = p.w, hw = H * W;
  const float* __restrict__ xf = reinterpret_cast<const float*>(p.x);

  // weights: [32 cout][32 k] in the storage type (k = (ky * 3 + kx) * 3 + c, taps 27..31 are zero), lane = cout
  frag bfr[2];
  {
    const frag* wt = reinterpret_cast<const frag*>(p.wgt_tiled);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) bfr[ks] = wt[r32 * 4 + 2 * ks + hh];
  }
  // this lane's 16 taps: k = 8 * (2 ks + hh) + e  ->  (dy, dx, c) and the element offset relative to the pixel
  int toff[16];
  unsigned tdy = 0, tdx = 0, tok = 0;  // 2 bits per tap: dy, dx in 0..2; 1 bit: k < 27
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int k = 8 * (2 * (t >> 3) + hh) + (t & 7);
    const int tap = k / 3, c = k - tap * 3;
    const int dy = tap / 3, dx = tap - dy * 3;
    const bool ok = k < 27;
    toff[t] = !ok ? 0 : (p.x_nchw ? (c * hw + (dy - 1) * W + (dx - 1)) : (int)(((dy - 1) * W + (dx - 1)) * p.x_pitch + c));
    tdy |= (unsigned)(ok ? dy : 0) << (2 * t);
    tdx |= (unsigned)(ok ? dx : 0) << (2 * t);
    tok |= (unsigned)ok << t;
  }
  const float sc = p.scale[r32], sh = p.shift[r32];
  const float slope = p.act == ME_ACT_LEAKY ? 0.1f : 1.0f;
  float* tbuf = &tbuf_all[wave][0];
  const int prow = lane >> 2, c8 = (lane & 3) * 8;
  unsigned short* __restrict__ yb = reinterpret_cast<unsigned short*>(p.y);

  const int nblk = (p.M + 31) >> 5;
  // raw fp32 taps of one block: fetched one block AHEAD (the wave's only long-latency operation), converted when consumed
  float raw[16];
  auto fetch = [&](int blk) {
    const int m = blk * 32 + r32;
    const bool live = blk < nblk && m < p.M;
    const unsigned mm = live ? (unsigned)m : 0u;
    const unsigned n = udiv_magic(mm, a.hw_m, a.hw_s);
    const unsigned rem = mm - n * (unsigned)hw;
    const unsigned y = udiv_magic(rem, a.w_m, a.w_s);
    const unsigned x = rem - y * (unsigned)W;
    // valid rows / columns of the 3 x 3 window as 3-bit masks
    const unsigned vy = (y > 0 ? 1u : 0u) | 2u | (y + 1 < (unsigned)H ? 4u : 0u);
    const unsigned vx = (x > 0 ? 1u : 0u) | 2u | (x + 1 < (unsigned)W ? 4u : 0u);
    const long long base = p.x_nchw ? ((long long)n * 3 * hw + (long long)y * W + x)
                                    : ((long long)(n * (unsigned)hw + y * (unsigned)W + x) * p.x_pitch);
    const float* px = xf + base;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const bool ok = live && ((tok >> t) & 1u) && ((vy >> ((tdy >> (2 * t)) & 3u)) & 1u) && ((vx >> ((tdx >> (2 * t)) & 3u)) & 1u);
      raw[t] = ok ? px[toff[t]] : 0.f;
    }
  };
  const int stride = gridDim.x * 4;
  int blk = blockIdx.x * 4 + wave;
  fetch(blk);
  for (; blk < nblk; blk += stride) {
    frag afr[2];
#pragma unroll
    for (int t = 0; t < 16; ++t) afr[t >> 3][t & 7] = (elem16)raw[t];
    fetch(blk + stride);  // in flight behind the MFMAs, the transpose and the stores of this block
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    acc = H16<F16>::mfma(afr[0], bfr[0], acc);
    acc = H16<F16>::mfma(afr[1], bfr[1], acc);
    // lane = output channel r32, register e = pixel (e & 3) + 8 (e >> 2) + 4 hh of the block; one wave's LDS operations
    // execute in order, so the patch needs neither a wait nor a second copy
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v = acc[e] * sc + sh;
      v = fmaxf(v, v * slope);
      tbuf[((e & 3) + 8 * (e >> 2) + 4 * hh) * 36 + r32] = v;
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int row = pass * 16 + prow;
      const float4 lo = *reinterpret_cast<const float4*>(tbuf + row * 36 + c8);
      const float4 hi = *reinterpret_cast<const float4*>(tbuf + row * 36 + c8 + 4);
      const long long mo = (long long)blk * 32 + row;
      if (mo < p.M) {
        uint4 o;
        o.x = pack2<F16>(lo.x, lo.y);
        o.y = pack2<F16>(lo.z, lo.w);
        o.z = pack2<F16>(hi.x, hi.y);
        o.w = pack2<F16>(hi.z, hi.w);
        *reinterpret_cast<uint4*>(yb + mo * p.y_pitch + c8) = o;
      }
    }
  }
}

// Row-staged version for NCHW frames (see conv_stem3_rows_f32 in stem_mfma_f32.hip): the workgroup's input window - 3 channels
// x (RY + 2) rows x (W + 2) pixels, zero halo - is staged in LDS by coalesced loads, the 16 taps of a lane come out of LDS with
// one add + ds_read_b32 each.  The kernel above issues ~290 vector + ~90 scalar instructions and 18 gathers per 32-pixel
// block at 3 waves / SIMD (148 VGPRs): 2.8 TB/s of the 6.8 TB/s a plain fill reaches.
constexpr int kStemRows16 = 4;

template <int F16>
__global__ __launch_bounds__(256) void conv_stem3_rows_h16(Conv16P p) {
  using frag = typename H16<F16>::v8;
  using elem16 = std::conditional_t<F16 != 0, _Float16, __bf16>;
  constexpr int RY = kStemRows16;
  extern __shared__ __attribute__((aligned(16))) float lds_rows[];  // [4 waves][32 * 36] transpose patches, then [3][RY + 2][WP]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const int H = p.h, W = p.w, WP = W + 2;
  float* tbuf = lds_rows + wave * (32 * 36);
  float* win = lds_rows + 4 * 32 * 36;
  const int groups = (H + RY - 1) / RY;
  const int n = blockIdx.x / groups, y0 = (blockIdx.x - n * groups) * RY;

  const float* __restrict__ img = reinterpret_cast<const float*>(p.x) + (long long)n * 3 * H * W;
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

  frag bfr[2];
  {
    const frag* wt = reinterpret_cast<const frag*>(p.wgt_tiled);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) bfr[ks] = wt[r32 * 4 + 2 * ks + hh];
  }
  int toff[16];  // LDS float offset of this lane's tap t, relative to (row 0 of the block, pixel x); taps >= 27 have zero weights
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int k = 8 * (2 * (t >> 3) + hh) + (t & 7);
    const int tap = k / 3, c = k - tap * 3;
    const int dy = tap / 3, dx = tap - dy * 3;
    toff[t] = k < 27 ? (c * (RY + 2) + dy) * WP + dx : 0;
  }
  const float sc = p.scale[r32], sh = p.shift[r32];
  const float slope = p.act == ME_ACT_LEAKY ? 0.1f : 1.0f;
  const int prow = lane >> 2, c8 = (lane & 3) * 8;
  unsigned short* __restrict__ yb = reinterpret_cast<unsigned short*>(p.y);
  __syncthreads();

  const int nbx = (W + 31) >> 5;
  const int rows = H - y0 < RY ? H - y0 : RY;
  for (int b = wave; b < rows * nbx; b += 4) {
    const int ry = b / nbx, bx = b - ry * nbx;
    const int x = bx * 32 + r32;
    const int xs = x < W ? x : W - 1;  // ragged last block: lanes beyond the row read a valid pixel, their pixels are not stored
    const int base = ry * WP + xs;
    frag afr[2];
#pragma unroll
    for (int t = 0; t < 16; ++t) afr[t >> 3][t & 7] = (elem16)win[base + toff[t]];
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    acc = H16<F16>::mfma(afr[0], bfr[0], acc);
    acc = H16<F16>::mfma(afr[1], bfr[1], acc);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float v = acc[e] * sc + sh;
      tbuf[((e & 3) + 8 * (e >> 2) + 4 * hh) * 36 + r32] = fmaxf(v, v * slope);
    }
    const long long m0 = ((long long)n * H + (y0 + ry)) * W + bx * 32;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int row = pass * 16 + prow;
      const float4 lo = *reinterpret_cast<const float4*>(tbuf + row * 36 + c8);
      const float4 hi = *reinterpret_cast<const float4*>(tbuf + row * 36 + c8 + 4);
      if (bx * 32 + row < W) {
        uint4 o;
        o.x = pack2<F16>(lo.x, lo.y);
        o.y = pack2<F16>(lo.z, lo.w);
        o.z = pack2<F16>(hi.x, hi.y);
        o.w = pack2<F16>(hi.z, hi.w);
        *reinterpret_cast<uint4*>(yb + (m0 + row) * p.y_pitch + c8) = o;
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

namespace me16 {

bool stem_mfma_eligible(const Conv16P& p) {
  return p.cin == 3 && p.ks == 3 && p.stride == 1 && p.pad == 1 && p.cout == 32 && p.ups == 1 && !p.y_f32 && !p.res &&
         p.wgt_tiled && p.act != ME_ACT_SIGMOID && p.y_pitch % 8 == 0 && (long long)p.n * p.h * p.w < (1ll << 31);
}

int launch_stem_mfma(const Conv16P& p, hipStream_t stream) {
  const size_t lds_bytes = ((size_t)4 * 32 * 36 + (size_t)3 * (kStemRows16 + 2) * (p.w + 2)) * sizeof(float);
  static const bool rows_ok = [] { const char* e = getenv("MILLIEYE_STEM_ROWS"); return !e || atoi(e) != 0; }();
  if (p.x_nchw && lds_bytes <= 64 * 1024 && rows_ok) {  // row-staged version: NCHW frames up to ~600 pixels wide
    const long long groups = (long long)p.n * ((p.h + kStemRows16 - 1) / kStemRows16);
    if (groups < (1ll << 31)) {
      if (p.f16) hipLaunchKernelGGL(conv_stem3_rows_h16<1>, dim3((unsigned)groups), dim3(256), lds_bytes, stream, p);
      else hipLaunchKernelGGL(conv_stem3_rows_h16<0>, dim3((unsigned)groups), dim3(256), lds_bytes, stream, p);
      return me::check_launch("conv_stem3_rows_h16");
    }
  }
  StemArgs a;
  a.c = p;
  magic_u32s((unsigned)p.w, &a.w_m, &a.w_s);
  magic_u32s((unsigned)(p.h * p.w), &a.hw_m, &a.hw_s);
  const int nblk = (p.M + 31) / 32;
  int grid = (nblk + 3) / 4;
  if (grid > 256 * 8) grid = 256 * 8;  // grid-stride, 8 resident workgroups per CU: weights fetched once per wave
  if (p.f16) hipLaunchKernelGGL(conv_stem3_mfma_h16<1>, dim3(grid), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(conv_stem3_mfma_h16<0>, dim3(grid), dim3(256), 0, stream, a);
  return me::check_launch("conv_stem3_mfma_h16");
}

}  // namespace me16

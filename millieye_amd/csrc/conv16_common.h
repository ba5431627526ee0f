// conv16_common.h - types and epilogue helpers shared by the 16-bit convolution kernels (conv_h16.hip: per-tap implicit
// GEMM, stem, pools; conv_p8_h16.hip: the patch-resident big-tile generation).
#pragma once
#include <math.h>
#include <type_traits>
#include <utility>
#include "common.h"
#include "dma.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// The two 16-bit storage types: F16 = 0 bfloat16 (v_mfma_f32_32x32x16_bf16), F16 = 1 IEEE half (v_mfma_f32_32x32x16_f16;
// BASELINE configs[4] "fp16 MFMA convs").  Same kernels, same layouts; only the operand type, the MFMA and the conversions
// differ.  Both round to nearest even.
template <int F16>
struct H16;
template <>
struct H16<0> {
  typedef __bf16 v8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ unsigned short to(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
  static __device__ __forceinline__ float from(unsigned b) { return __uint_as_float(b << 16); }
};
template <>
struct H16<1> {
  typedef _Float16 v8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ unsigned short to(float v) { return __builtin_bit_cast(unsigned short, (_Float16)v); }
  static __device__ __forceinline__ float from(unsigned b) {
    return (float)__builtin_bit_cast(_Float16, (unsigned short)b);
  }
};


struct Conv16P {
  const unsigned short* x;
  const unsigned short* wgt;
  const unsigned short* wgt_tiled;  // [taps][cin/32][cout][32] copy (patch-resident kernels), or nullptr
  const float* scale;
  const float* shift;
  const void* res;
  void* y;
  long long x_pitch, res_pitch, y_pitch;  // elements
  int n, h, w, cin, cout, ks, stride, pad, ho, wo, act, ups, y_f32, x_nchw;
  int M;       // n*ho*wo
  int ktot;    // ks*ks*cin
  int cs;      // stages per filter tap = cin / (32*KSUB)
  int stages;  // ks*ks*cs
  int tiles_m, tiles_n;
  unsigned hw_m, hw_s, wo_m, wo_s;  // magic numbers of the divisions by ho*wo and wo: n / d == (umulhi(n, m) + n) >> s, n < 2^31
  float* partial;  // split-K slabs [splitk][M][cout] (raw fp32 accumulators; patch tiles: [tile][split][BM][BN]), or nullptr
  long long partial_bytes;
  int splitk, sps;
  int f16;         // 0 = bfloat16 storage, 1 = IEEE half
  int vec_epi;     // 16-byte epilogue allowed (bf16 out, no upsample, leaky / linear, pitches % 8, 16-byte aligned)
  int store_mode;  // me::store_mode(): 16-byte epilogue stores plain (0), nt (1) or sc1 write-through (2)
  int mask_cols;        // me_conv16_desc.tap_mask_cols (0: no masks)
  unsigned tapmask[4];  // me_conv16_desc.tap_mask: set taps of the four column classes (conv_igemm_buf_h16<..., MASKED = 1>)
};

__device__ __forceinline__ float act16(float v, int act) {
  if (act == ME_ACT_LEAKY) return v > 0.f ? v : 0.1f * v;
  if (act == ME_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  return v;
}
template <int F16>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  return (unsigned)H16<F16>::to(lo) | ((unsigned)H16<F16>::to(hi) << 16);
}

template <class F, int... J>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, J...>) {
  (f(std::integral_constant<int, J>{}), ...);
}

// one output element through the fused epilogue tail: residual, rounding, (replicated) store
template <int F16>
__device__ __forceinline__ void store_out(const Conv16P& p, int m, int co, float v, int hw) {
  if (p.res) {
    v += p.y_f32 ? reinterpret_cast<const float*>(p.res)[(long long)m * p.res_pitch + co]
                 : H16<F16>::from(reinterpret_cast<const unsigned short*>(p.res)[(long long)m * p.res_pitch + co]);
  }
  if (p.ups == 1) {
    const long long o = (long long)m * p.y_pitch + co;
    if (p.y_f32)
      reinterpret_cast<float*>(p.y)[o] = v;
    else
      reinterpret_cast<unsigned short*>(p.y)[o] = H16<F16>::to(v);
    return;
  }
  const int nimg = m / hw;
  const int rem = m - nimg * hw;
  const int oy = rem / p.wo, ox = rem - oy * p.wo;
  const int W2 = p.wo * 2;
  const long long base = ((long long)nimg * (p.ho * 2) + 2 * oy) * W2 + 2 * ox;
  const long long o[4] = {base * p.y_pitch + co, (base + 1) * p.y_pitch + co, (base + W2) * p.y_pitch + co,
                          (base + W2 + 1) * p.y_pitch + co};
  if (p.y_f32) {
#pragma unroll
    for (int k = 0; k < 4; ++k) reinterpret_cast<float*>(p.y)[o[k]] = v;
  } else {
    const unsigned short b = H16<F16>::to(v);
#pragma unroll
    for (int k = 0; k < 4; ++k) reinterpret_cast<unsigned short*>(p.y)[o[k]] = b;
  }
}

__device__ __forceinline__ unsigned udiv_magic16(unsigned n, unsigned m, unsigned s) { return (__umulhi(n, m) + n) >> s; }

inline void magic16(unsigned d, unsigned* m, unsigned* s) {
  unsigned sh = 0;
  while ((1ull << sh) < d) ++sh;
  *s = sh;
  *m = (unsigned)(((1ull << 32) * ((1ull << sh) - d)) / d + 1);
}

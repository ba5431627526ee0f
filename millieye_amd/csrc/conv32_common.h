// conv32_common.h - parameter block and activation helper shared by the fp32 convolution kernels (conv.hip: per-tap
// implicit GEMM, stems, split-K; conv_p8_f32.hip: the patch-resident big-tile generation).
#pragma once
#include <math.h>
#include "common.h"
#include "dma.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvP {
  const float* x;
  const float* wgt;
  const float* wgt_tiled;  // [taps][cin/16][cout][16] copy (patch-resident kernels), or nullptr
  const float* scale;
  const float* shift;
  const float* res;
  float* y;
  long long x_pitch, res_pitch, y_pitch;
  int n, h, w, cin, cout, ks, stride, pad, ho, wo, act, ups, x_nchw;
  unsigned hw_m, hw_s, wo_m, wo_s;  // magic numbers of the divisions by ho*wo and wo (conv_igemm_buf_f32)
  int M;       // n*ho*wo
  int ktot;    // ks*ks*cin
  int cs;      // channel chunks per tap = ceil(cin / BK)
  int stages;  // ks*ks*cs
  int tiles_m, tiles_n;
  float* partial;  // split-K slabs [splitk][M][cout] (raw accumulators), or nullptr
  int splitk;      // number of K splits (grid.y)
  int sps;         // K stages per split
  int gn;          // conv_igemm_buf_f32: tile columns per panel of the tile walk (tiles_n = row-major)
  int kord;        // conv_igemm_buf_f32: K walk, 0 tap-major / 1 chunk-major
  int* counters;   // per-tile arrival counters of the in-launch slab reduction (zero between launches), or nullptr
  long long counters_len;
  int bulk;        // tail-split launches: tiles [0, bulk) run whole, tiles [bulk, tiles) cut splitk ways (0 otherwise)
  int store_mode;  // me::store_mode(): output stores of the fast epilogue plain (0) or streaming (nt)
  unsigned tapmask[4];  // me_conv_desc.tap_mask: set taps of the four column classes (conv_igemm_buf_f32, tap-major walk)
  int mask_cols;        // output channels per column class, 0 = no masks
};

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == ME_ACT_LEAKY) return v > 0.f ? v : 0.1f * v;
  if (act == ME_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  return v;
}


// conv.hip - fused convolution kernels for gfx950 (MI355X / CDNA4), fp32.
//
//   conv_igemm_f32  : im2col-free implicit GEMM on the f32 matrix cores
//                     (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chains, 157 TFLOP/s peak).
//                     M = n*ho*wo output pixels, N = cout, K = ksize^2 * cin.
//                     The K loop walks (tap, cin-chunk); for one tap the A tile is a set of
//                     BM pixel rows x BK contiguous channels of the NHWC input (coalesced 16 B
//                     loads, zero for padding taps), the B tile BN weight rows x BK.  Tiles are
//                     register-prefetched one stage ahead and double buffered in LDS.
//                     Epilogue (fused): per-channel affine (folded BN / bias), LeakyReLU /
//                     sigmoid, residual add ([shortcut]), nearest x2 replication ([upsample]),
//                     pitched store into a channel slice of a concat buffer ([route]).
//   conv_smallcin_f32: direct 3x3 convolution for cin <= 4 (network stem / radar stem): HBM
//                     bound (arithmetic intensity ~12 flop/B), no MFMA benefit; reads NCHW or
//                     NHWC, writes NHWC.
//
// Replaces the nn.Conv2d + BatchNorm2d + LeakyReLU blocks of
// module3_our_dataset/yolov3/models.py:22-41 (see include/millieye_hip.h).
#include <stdlib.h>

#include "common.h"
#include "dma.h"

#include "conv32_common.h"

namespace me32 {  // conv_p8_f32.hip: patch-resident big tiles (tile ids >= 100)
int launch_p8_tile(const ConvP& p, int tile, hipStream_t stream);
bool ws1x1_f32_eligible(const ConvP& p);  // conv_ws_f32.hip: weight-stationary short-K layers (tile ids 50 / 60)
int launch_ws1x1_f32(const ConvP& p, hipStream_t stream);
bool ws3x3_f32_eligible(const ConvP& p);
int launch_ws3x3_f32(const ConvP& p, hipStream_t stream);
bool stem_mfma_eligible(const ConvP& p);  // stem_mfma_f32.hip
int launch_stem_mfma(const ConvP& p, hipStream_t stream);
}  // namespace me32

namespace {

// ---------------------------------------------------------------------------------------------
// implicit-GEMM MFMA kernel
// ---------------------------------------------------------------------------------------------
// ABL (ablation, tuning only): 0 = production kernel; 1 = global loads / LDS refills skipped after
// the first stage (matrix pipe + LDS reads + barriers only); 2 = MFMAs skipped (memory side only).
template <int BM, int BN, int BK, int WR, int WC, int ABL = 0, int PRIO = 0>
__global__ __launch_bounds__(256) void conv_igemm_f32(ConvP p) {
  static_assert(WR * WC == 4, "4 waves per workgroup");
  constexpr int TM = BM / WR, TN = BN / WC;
  constexpr int MT = TM / 32, NT = TN / 32;
  static_assert(MT >= 1 && NT >= 1, "wave tile must hold at least one 32x32 MFMA tile");
  constexpr int LP = BK + 4;  // LDS row pitch (floats): (LP/4) odd -> conflict-free ds_read_b128
  constexpr int CH = BK / 4;  // 16-byte chunks per row
  constexpr int A_IT = (BM * CH + 255) / 256;
  constexpr int B_IT = (BN * CH + 255) / 256;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][BM*LP]
  float* Bs = smem + 2 * BM * LP;   // [2][BN*LP]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int r32 = lane & 31, hh = lane >> 5;

  // XCD-aware bijective remap: consecutive tile ids stay on one XCD (private L2).
  int tile_m, tile_n;
  {
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    tile_n = wg % p.tiles_n;
    tile_m = wg / p.tiles_n;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // per-thread A-row bookkeeping (constant over the K loop)
  int a_iy0[A_IT], a_ix0[A_IT], a_pix0[A_IT], a_lds[A_IT], a_q[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int id = tid + 256 * i;
    const int row = id / CH, q = id % CH;
    a_lds[i] = row * LP + 4 * q;
    a_q[i] = 4 * q;
    const int m = m0 + row;
    if (id < BM * CH && m < p.M) {
      const int hw = p.ho * p.wo;
      const int nimg = m / hw;
      const int rem = m - nimg * hw;
      const int oy = rem / p.wo, ox = rem - oy * p.wo;
      a_iy0[i] = oy * p.stride - p.pad;
      a_ix0[i] = ox * p.stride - p.pad;
      a_pix0[i] = nimg * p.h * p.w;
    } else {
      a_iy0[i] = -(1 << 28);  // every tap out of bounds -> zeros
      a_ix0[i] = 0;
      a_pix0[i] = 0;
    }
  }
  long long b_off[B_IT];
  int b_lds[B_IT], b_q[B_IT];
  bool b_ok[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int id = tid + 256 * i;
    const int row = id / CH, q = id % CH;
    b_lds[i] = row * LP + 4 * q;
    b_q[i] = 4 * q;
    const int co = n0 + row;
    b_ok[i] = (id < BN * CH) && (co < p.cout);
    b_off[i] = (long long)co * p.ktot + 4 * q;
  }

  float4 a_reg[A_IT], b_reg[B_IT];
  // split-K: this workgroup accumulates K stages [s_begin, s_end) of its tile
  const int sid = blockIdx.y;
  const int s_begin = sid * p.sps;
  const int s_end = (s_begin + p.sps < p.stages) ? s_begin + p.sps : p.stages;
  int tap = s_begin / p.cs, cc = s_begin - (s_begin / p.cs) * p.cs;  // stage -> (tap, channel chunk)

  auto load_stage = [&]() {
    const int ky = tap / p.ks, kx = tap - ky * p.ks;
    const int c0 = cc * BK;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
      const bool ok = ((unsigned)iy < (unsigned)p.h) && ((unsigned)ix < (unsigned)p.w) && (c0 + a_q[i] < p.cin);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) {
        long long off = (long long)(a_pix0[i] + iy * p.w + ix) * p.x_pitch + (c0 + a_q[i]);
        if (ABL == 3) off &= 0x3FFC;  // ablation: same instruction stream, every load L1/L2 resident (64 KiB)
        v = *reinterpret_cast<const float4*>(p.x + off);
      }
      a_reg[i] = v;
    }
    const int koff = tap * p.cin + c0;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b_ok[i] && (c0 + b_q[i] < p.cin)) v = *reinterpret_cast<const float4*>(p.wgt + b_off[i] + koff);
      b_reg[i] = v;
    }
    if (++cc == p.cs) {
      cc = 0;
      ++tap;
    }
  };
  auto store_stage = [&](int buf) {
    float* Ad = As + buf * BM * LP;
    float* Bd = Bs + buf * BN * LP;
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      if (A_IT * 256 == BM * CH || tid + 256 * i < BM * CH) *reinterpret_cast<float4*>(Ad + a_lds[i]) = a_reg[i];
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
      if (B_IT * 256 == BN * CH || tid + 256 * i < BN * CH) *reinterpret_cast<float4*>(Bd + b_lds[i]) = b_reg[i];
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  load_stage();
  store_stage(0);
  __syncthreads();

  const int a_frag = (wr * TM + r32) * LP + 4 * hh;
  const int b_frag = (wc * TN + r32) * LP + 4 * hh;

  for (int s = s_begin; s < s_end; ++s) {
    const int buf = (s - s_begin) & 1;
    const bool more = (s + 1 < s_end);
    if (more && ABL != 1) load_stage();  // global loads in flight while the matrix cores work on `buf`
    const float* Ab = As + buf * BM * LP + a_frag;
    const float* Bb = Bs + buf * BN * LP + b_frag;
    if (PRIO) __builtin_amdgcn_s_setprio(1);  // matrix phase outranks the other waves' load phases
#pragma unroll
    for (int kk = 0; kk < BK; kk += 8) {
      float4 af[MT], bf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LP + kk);
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LP + kk);
      // lanes 0-31 hold k = kk..kk+3, lanes 32-63 hold k = kk+4..kk+7 (same split for A and B):
      // each of the four MFMAs consumes one k from each half -> all 8 k's, each exactly once.
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (ABL == 2) {  // keep the fragment reads alive without touching the matrix pipe
            asm volatile("" ::"v"(af[i].x), "v"(af[i].y), "v"(af[i].z), "v"(af[i].w), "v"(bf[j].x), "v"(bf[j].y),
                         "v"(bf[j].z), "v"(bf[j].w));
            continue;
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    if (more && ABL != 1) store_stage(buf ^ 1);
    __syncthreads();
  }

  // ---- fused epilogue -------------------------------------------------------------------
  // 32x32 C/D layout: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5).
  if (p.splitk > 1) {  // raw partial sums; conv_splitk_reduce_f32 applies the epilogue
    float* slab = p.partial + (long long)sid * p.M * p.cout;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = n0 + wc * TN + j * 32 + r32;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wr * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (co < p.cout && m < p.M) slab[(long long)m * p.cout + co] = acc[i][j][e];
        }
    }
    return;
  }
  const int hw = p.ho * p.wo;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = n0 + wc * TN + j * 32 + r32;
    const bool co_ok = co < p.cout;
    const float sc = co_ok ? p.scale[co] : 0.f;
    const float sh = co_ok ? p.shift[co] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * hh;
        const int m = m0 + wr * TM + i * 32 + row;
        if (!co_ok || m >= p.M) continue;
        float v = apply_act(acc[i][j][e] * sc + sh, p.act);
        if (p.res) v += p.res[(long long)m * p.res_pitch + co];
        if (p.ups == 1) {
          p.y[(long long)m * p.y_pitch + co] = v;
        } else {
          const int nimg = m / hw;
          const int rem = m - nimg * hw;
          const int oy = rem / p.wo, ox = rem - oy * p.wo;
          const int W2 = p.wo * 2;
          const long long base = ((long long)nimg * (p.ho * 2) + 2 * oy) * W2 + 2 * ox;
          p.y[(base)*p.y_pitch + co] = v;
          p.y[(base + 1) * p.y_pitch + co] = v;
          p.y[(base + W2) * p.y_pitch + co] = v;
          p.y[(base + W2 + 1) * p.y_pitch + co] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// implicit-GEMM MFMA kernel, LDS-DMA edition (global_load_lds_dwordx4): tiles go HBM/L2 -> LDS
// without a VGPR round trip, three K stages are in flight (two behind a counted s_waitcnt
// vmcnt), one raw s_barrier per stage.  LDS rows are unpadded 64 B (BK = 16 floats); the
// DMA destination is lane-linear (wave-uniform base + lane * 16 B), so the bank-conflict swizzle
// lives on the per-lane SOURCE address: the 16-byte slot q of row R is stored at slot
// q ^ ((R >> 2) & 3) and ds_read_b128 applies the same XOR (conflict-free for its 16-lane groups).
// Padding taps / ragged rows read from a zero block in global memory.
// ---------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(64))) float g_zero_block[16];

// DABL (ablation, tuning only): 1 = every DMA reads the zero block (same instruction stream, memory system idle).
template <int BM, int BN, int WR, int WC, int DPRIO = 0, int MINW = 1, int DABL = 0>
__global__ __launch_bounds__(64 * WR * WC, MINW) void conv_igemm_dma_f32(ConvP p) {
  constexpr int NW = WR * WC;  // waves per workgroup (4 or 8)
  static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
  constexpr int BK = 16, NST = 3;
  constexpr int TM = BM / WR, TN = BN / WC;
  constexpr int MT = TM / 32, NT = TN / 32;
  constexpr int GA = BM / 16, G = (BM + BN) / 16;  // 16-row groups (1 KiB each): A first, then B
  constexpr int LPW = (G + NW - 1) / NW;            // DMA instructions per wave per stage
  constexpr int STAGE_F = LPW * NW * 256;           // floats per stage buffer (incl. dummy groups)

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const int r32 = lane & 31, hh = lane >> 5;

  int tile_m, tile_n;
  {
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    tile_n = wg % p.tiles_n;
    tile_m = wg / p.tiles_n;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // per-lane source bookkeeping for this wave's groups g = wave + 4 * j (A groups first, then B).
  // Everything that does not depend on the K stage is hoisted: a lane pointer for (tap 0, channel 0),
  // a 9-bit "tap is readable" mask and the channel limit; per stage only a wave-uniform offset is added.
  static_assert(GA % NW == 0, "A groups must split evenly over the waves");
  constexpr int LA = GA / NW;  // A-type DMA instructions per wave
  const int lrow = lane >> 2;  // row inside the 16-row group
  const float* g_ptr[LPW];
  unsigned g_mask[LPW];
  int g_qlim[LPW];
#pragma unroll
  for (int j = 0; j < LPW; ++j) {
    const int g = wave + NW * j;
    const int row = g * 16 + lrow;                       // row inside the (A|B) stage image
    const int q = (lane & 3) ^ ((row >> 2) & 3);         // source 16-byte chunk for this LDS slot
    g_qlim[j] = p.cin - 4 * q;
    g_mask[j] = 0;
    g_ptr[j] = g_zero_block;
    if (j < LA) {
      const int m = m0 + row;
      if (m < p.M) {
        const int hw = p.ho * p.wo;
        const int nimg = m / hw;
        const int rem = m - nimg * hw;
        const int oy = rem / p.wo, ox = rem - oy * p.wo;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        unsigned mask = 0;
        for (int ky = 0; ky < p.ks; ++ky)
          for (int kx = 0; kx < p.ks; ++kx)
            if ((unsigned)(iy0 + ky) < (unsigned)p.h && (unsigned)(ix0 + kx) < (unsigned)p.w)
              mask |= 1u << (ky * p.ks + kx);
        g_mask[j] = mask;
        g_ptr[j] = p.x + ((long long)nimg * p.h * p.w + (long long)iy0 * p.w + ix0) * p.x_pitch + 4 * q;
      }
    } else if (g < G) {
      const int co = n0 + (row - BM);
      if (co < p.cout) {
        g_mask[j] = 0xFFFFFFFFu;
        g_ptr[j] = p.wgt + (long long)co * p.ktot + 4 * q;
      }
    }
  }

  const int sid = blockIdx.y;
  const int s_begin = sid * p.sps;
  const int s_end = (s_begin + p.sps < p.stages) ? s_begin + p.sps : p.stages;
  int tap = s_begin / p.cs, cc = s_begin - (s_begin / p.cs) * p.cs;

  const unsigned wave_lds = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)wave * 1024u);  // SGPR
  auto issue_stage = [&](int slot) {
    const int ky = tap / p.ks, kx = tap - ky * p.ks;
    const int c0 = cc * BK;
    const long long a_off = (long long)(ky * p.w + kx) * p.x_pitch + c0;  // wave-uniform (scalar) offsets
    const long long b_off = (long long)tap * p.cin + c0;
    const unsigned stage_lds = wave_lds + (unsigned)slot * (STAGE_F * 4u);
#pragma unroll
    for (int j = 0; j < LPW; ++j) {
      const bool ok = DABL != 1 && ((g_mask[j] >> tap) & 1u) && (c0 < g_qlim[j]);
      const float* src = ok ? g_ptr[j] + (j < LA ? a_off : b_off) : g_zero_block;
      // LDS byte address of this group's 1 KiB image: wave-uniform; hardware adds lane * 16 B.
      // Issued through inline asm so that hipcc does not track the DMA (it would drain it with
      // s_waitcnt vmcnt(0) in front of every ds_read); completion is counted by hand below.
      const unsigned dst = stage_lds + (unsigned)j * (NW * 1024u);
      unsigned keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(src), "s"(dst)
          : "memory");
    }
    if (++cc == p.cs) {
      cc = 0;
      ++tap;
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nstages = s_end - s_begin;
  issue_stage(0);
  if (nstages > 1) issue_stage(1);

  const int sw = (r32 >> 2) & 3;
  const int a_row = (wr * TM + r32) * BK;
  const int b_row = (BM + wc * TN + r32) * BK;
  const int off0 = ((0 + hh) ^ sw) * 4, off1 = ((2 + hh) ^ sw) * 4;

  for (int s = 0; s < nstages; ++s) {
    // stage s landed? (the only younger DMAs are those of stage s+1)
    if (s + 1 < nstages)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPW) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // everyone's stage-s DMAs landed; everyone finished reading stage s-1
    asm volatile("" ::: "memory");
    if (s + 2 < nstages) issue_stage((s + 2) % NST);  // refills the slot stage s-1 just vacated
    if (DPRIO) __builtin_amdgcn_s_setprio(1);
    const float* Ab = smem + (s % NST) * STAGE_F + a_row;
    const float* Bb = smem + (s % NST) * STAGE_F + b_row;
    // both halves' fragments are requested up front (the stage has landed), so the LDS latency of the
    // second half hides behind the first half's MFMAs
    float4 af[2][MT], bf[2][NT];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int off = half ? off1 : off0;
#pragma unroll
      for (int i = 0; i < MT; ++i) af[half][i] = *reinterpret_cast<const float4*>(Ab + i * 32 * BK + off);
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[half][j] = *reinterpret_cast<const float4*>(Bb + j * 32 * BK + off);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the eight ds_reads ahead of the MFMAs (hipcc sinks them otherwise)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      // k-major issue order: consecutive MFMAs hit different accumulators (a dependent one is MT*NT later)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const float4 av = af[half][i], bv = bf[half][j];
            const float a = kk == 0 ? av.x : kk == 1 ? av.y : kk == 2 ? av.z : av.w;
            const float b = kk == 0 ? bv.x : kk == 1 ? bv.y : kk == 2 ? bv.z : bv.w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
          }
    }
    if (DPRIO) __builtin_amdgcn_s_setprio(0);
  }

  // ---- epilogue (same as conv_igemm_f32) --------------------------------------------------
  if (p.splitk > 1) {
    float* slab = p.partial + (long long)sid * p.M * p.cout;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = n0 + wc * TN + j * 32 + r32;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wr * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (co < p.cout && m < p.M) slab[(long long)m * p.cout + co] = acc[i][j][e];
        }
    }
    return;
  }
  const int hw = p.ho * p.wo;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = n0 + wc * TN + j * 32 + r32;
    const bool co_ok = co < p.cout;
    const float sc = co_ok ? p.scale[co] : 0.f;
    const float sh = co_ok ? p.shift[co] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * hh;
        const int m = m0 + wr * TM + i * 32 + row;
        if (!co_ok || m >= p.M) continue;
        float v = apply_act(acc[i][j][e] * sc + sh, p.act);
        if (p.res) v += p.res[(long long)m * p.res_pitch + co];
        if (p.ups == 1) {
          p.y[(long long)m * p.y_pitch + co] = v;
        } else {
          const int nimg = m / hw;
          const int rem = m - nimg * hw;
          const int oy = rem / p.wo, ox = rem - oy * p.wo;
          const int W2 = p.wo * 2;
          const long long base = ((long long)nimg * (p.ho * 2) + 2 * oy) * W2 + 2 * ox;
          p.y[(base)*p.y_pitch + co] = v;
          p.y[(base + 1) * p.y_pitch + co] = v;
          p.y[(base + W2) * p.y_pitch + co] = v;
          p.y[(base + W2 + 1) * p.y_pitch + co] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Stem, second version (cin == 3, cout a multiple of CT): one thread = one output pixel x CT output channels.
// The weights are wave-uniform, so hipcc fetches them with scalar loads and feeds them to the FMAs as SGPR operands -
// no LDS at all (conv_smallcin_f32 re-reads its weights from LDS for every pixel: 54 ds_read_b128 per thread, the LDS
// port was the limiter at 2.7 TB/s of output).  A wave reads 64 consecutive pixels of the NCHW planes (256 B per load)
// and stores 64 x CT consecutive floats (NHWC rows of adjacent pixels are adjacent).  HBM-bound: in + out bytes.
// ---------------------------------------------------------------------------------------------
template <int CT>
__global__ __launch_bounds__(256) void conv_stem3_f32(ConvP p) {
  constexpr int K = 27;
  const int m = blockIdx.x * 256 + threadIdx.x;
  const int co0 = blockIdx.y * CT;
  if (m >= p.M) return;
  const int hw = p.ho * p.wo;
  const int nimg = m / hw;
  const int rem = m - nimg * hw;
  const int oy = rem / p.wo, ox = rem - oy * p.wo;
  float xin[K];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
      const bool ok = ((unsigned)iy < (unsigned)p.h) && ((unsigned)ix < (unsigned)p.w);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = 0.f;
        if (ok)
          v = p.x_nchw ? p.x[(((long long)nimg * 3 + c) * p.h + iy) * p.w + ix]
                       : p.x[((long long)(nimg * p.h + iy) * p.w + ix) * p.x_pitch + c];
        xin[(ky * 3 + kx) * 3 + c] = v;
      }
    }
  const float* __restrict__ wg = p.wgt + (long long)co0 * K;  // [cout][ky][kx][cin], uniform -> s_load
  float acc[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) a = fmaf(xin[k], wg[j * K + k], a);  // same k order as conv_smallcin_f32
    acc[j] = apply_act(a * p.scale[co0 + j] + p.shift[co0 + j], p.act);
  }
  float* yrow = p.y + (long long)m * p.y_pitch + co0;
#pragma unroll
  for (int j = 0; j < CT; j += 4) *reinterpret_cast<float4*>(yrow + j) = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
}

// ---------------------------------------------------------------------------------------------
// implicit-GEMM MFMA kernel, buffer-addressed LDS-DMA edition (production).  Same tiling, LDS image,
// swizzle and 3-stage pipeline as conv_igemm_dma_f32, but the stage loop carries (almost) no VALU / SALU
// work: on gfx950 every non-MFMA instruction issued on a SIMD costs matrix-pipe time (measured with
// tools/mfma_mix.hip: ~2 cycles per VALU op, ~3 per SALU op against 64 per v_mfma_f32_32x32x2_f32; the
// global_load_lds version spends 53 VALU + 90 SALU per 32 MFMAs on 64-bit per-lane addresses, predicates
// and M0 juggling = 15 % of the pipe).  Here each DMA is `buffer_load_dwordx4 v_off, rsrc, s_off offen lds`:
//   * the per-lane byte offset v_off is fixed per (lane, tap): it only changes when the K walk moves to the
//     next filter tap (2 VALU ops per A-type DMA, every cin/16 stages); weights never change theirs;
//   * the per-stage part (tap position, channel chunk) is one SGPR offset per operand;
//   * zero padding, ragged rows and ragged couts use the buffer range check: such lanes carry the offset
//     0x80000000 >= num_records, the hardware writes zeros into LDS and touches no memory
//     (tools/buflds_probe.hip: out-of-range lanes do zero-fill LDS, and s_off takes part in the check, so all real
//     offsets + s_off stay below 2^31 - the host falls back to conv_igemm_dma_f32 otherwise);
//   * M0 is saved / restored once per stage around all of the wave's DMAs.
// Needs cin % 16 == 0 (no per-chunk channel predicate); other shapes use conv_igemm_dma_f32.
// ---------------------------------------------------------------------------------------------
using namespace me_dma;

// n / d for n < 2^31 with the host's (m, s) = magic_u32(d)
__device__ __forceinline__ unsigned udiv_magic32(unsigned n, unsigned m, unsigned s) { return (__umulhi(n, m) + n) >> s; }

// f(integral_constant<int, 0>{}) ... f(integral_constant<int, N - 1>{}), in order
template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, F, I + 1>(static_cast<F&&>(f));
  }
}

// Tile index (XCD-contiguous, see the remap in the kernel) -> tile coordinates: column panels of p.gn tile columns, row-major
// inside a panel (the last panel may be narrower).  p.gn == p.tiles_n is the plain row-major order.  With a panel whose weight
// rows fit the XCD's L2 the weights are fetched once per XCD instead of once per (drifting) workgroup.
__device__ __forceinline__ void tile_coords(const ConvP& p, int t, int& tile_m, int& tile_n) {
  const int per_panel = p.tiles_m * p.gn;
  const int panel = t / per_panel;
  const int r = t - panel * per_panel;
  const int left = p.tiles_n - panel * p.gn;
  const int width = left < p.gn ? left : p.gn;
  tile_m = r / width;
  tile_n = panel * p.gn + (r - tile_m * width);
}

// BABL (ablation, tuning only): 1 = every DMA lane is out of range (zero fill, no L2 / HBM traffic at all).
// HYB (tail split): a 1-D grid of p.bulk whole tiles (a multiple of 256: every CU gets the same number of them) followed by the
// remaining tiles cut p.splitk ways along K, so that the last partial round of workgroups is made of pieces small enough to
// spread over all CUs.  The pieces write raw accumulators into compact slabs [tail tile][split][BM][BN];
// conv_tail_reduce_f32 sums them in a fixed order and applies the epilogue.
// DEEP: a longer LDS ring (6 stages tap-major / 9 chunk-major instead of 3: five / eight stages of DMAs in flight) for launches
// that put one or two workgroups on a CU (batch 1 / 8): there a stage is bound by the latency of its DMAs, not by the matrix pipe.
// MASKED: the instance with the column-class tap masks (me_conv_desc.tap_mask, ABI 10).  A separate instance on purpose: the two
// scalar tests in the tap walk changed the register allocation of the plain kernel (64 x 64 tile: 63 -> 43 VGPRs, another schedule).
template <int BM, int BN, int WR, int WC, int MINW = 1, int BABL = 0, int HYB = 0, int KORD = 0, int DEEP = 0, int MASKED = 0>
__global__ __launch_bounds__(64 * WR * WC, MINW) void conv_igemm_buf_f32(ConvP p) {
  constexpr int NW = WR * WC;  // waves per workgroup (4 or 8)
  static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
  constexpr int BK = 16, NST = DEEP ? (KORD ? 9 : 6) : 3;
  constexpr int DEPTH = NST - 1;  // stages of DMAs in flight behind the one being multiplied
  constexpr int TM = BM / WR, TN = BN / WC;
  constexpr int MT = TM / 32, NT = TN / 32;
  constexpr int GA = BM / 16, G = (BM + BN) / 16;  // 16-row groups (1 KiB each): A first, then B
  constexpr int LPW = (G + NW - 1) / NW;            // DMA instructions per wave per stage
  constexpr int LA = GA / NW;                       // ... of which A-type
  constexpr int STAGE_F = LPW * NW * 256;           // floats per stage buffer (incl. dummy groups)
  static_assert(GA % NW == 0, "A groups must split evenly over the waves");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const int r32 = lane & 31, hh = lane >> 5;

  int tile_m, tile_n;
  int sid = blockIdx.y;  // K split this workgroup accumulates
  int piece = -1;        // HYB: index of this workgroup's slab (tail pieces), -1 = a whole tile with the fused epilogue
  {
    int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    int first = 0, per = 1;
    if constexpr (HYB) {
      if (bid < p.bulk) {
        nwg = p.bulk;
      } else {
        bid -= p.bulk;  // p.bulk % 8 == 0: bid & 7 is still the XCD
        first = p.bulk;
        per = p.splitk;
        nwg = (nwg - p.bulk) * p.splitk;
      }
    }
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    if constexpr (HYB) {
      if (first || p.bulk == 0) {
        piece = wg;
        sid = wg % per;
        wg = first + wg / per;
      } else {
        sid = 0;
      }
    }
    tile_coords(p, wg, tile_m, tile_n);
    tile_m = __builtin_amdgcn_readfirstlane(tile_m);
    tile_n = __builtin_amdgcn_readfirstlane(tile_n);
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int hw = p.ho * p.wo;

  // descriptors: A = activations rebased to the first image this tile touches, minus a bias of `pad` rows +
  // `pad` pixels so that every lane's tap-0 offset is non-negative; B = the weight rows of this tile.
  const int img0 = (int)udiv_magic32((unsigned)m0, p.hw_m, p.hw_s);
  const long long img_elems = (long long)p.h * p.w * p.x_pitch;
  const long long bias_elems = ((long long)p.pad * p.w + p.pad) * p.x_pitch;
  const u32x4 rsrc_a = make_rsrc(p.x + (long long)img0 * img_elems - bias_elems);
  // B: the weight rows of this tile - from the tiled copy [tap][cin/16][cout][16] when the caller has one (a K stage of
  // the tile is then 64 B x BN contiguous: consecutive L2 sets; rows of the OHWI layout lie ktot*4 bytes apart and a stage of
  // all cout rows lands in 1/16 of the sets), else from the OHWI rows.
  const bool b_tiled = p.wgt_tiled != nullptr;
  const u32x4 rsrc_b = make_rsrc(b_tiled ? p.wgt_tiled + (long long)n0 * 16 : p.wgt + (long long)n0 * p.ktot);
  const unsigned b_row = b_tiled ? 64u : (unsigned)p.ktot * 4u;      // bytes between the rows of two output channels
  const unsigned b_step = b_tiled ? (unsigned)p.cout * 64u : BK * 4u;  // ... between two K stages of one tap

  const int lrow = lane >> 2;  // row inside the 16-row group
  unsigned v_base[LPW];        // in-range byte offset of (lane, tap 0 / k 0)
  constexpr bool KS3 = KORD == 1;  // the chunk-major walk is launched for 3x3 filters only
  unsigned v_pad[LA];          // A-type: bit t set = tap t is padding (or the row is beyond M)
  unsigned v_cur[LPW];         // what the DMA uses: v_base or kOobOffset
#pragma unroll
  for (int j = 0; j < LPW; ++j) {
    const int g = wave + NW * j;
    const int row = g * 16 + lrow;                // row inside the (A|B) stage image
    const int q = (lane & 3) ^ ((row >> 2) & 3);  // source 16-byte chunk for this LDS slot
    v_base[j] = kOobOffset;
    if (j < LA) {
      unsigned padmask = 0xFFFFFFFFu;
      const int m = m0 + row;
      if (m < p.M) {
        // (per-tile setup is paid by every one of the 5-20 thousand tiles of a launch: multiply-shift divisions with the
        //  host's magic numbers, and the tap mask from ks row flags x ks column flags instead of ks * ks box tests)
        const int nimg = (int)udiv_magic32((unsigned)m, p.hw_m, p.hw_s);
        const int rem = m - nimg * hw;
        const int oy = (int)udiv_magic32((unsigned)rem, p.wo_m, p.wo_s), ox = rem - oy * p.wo;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        unsigned ok = 0;
        auto in_w = [&](int kx) { return (unsigned)(ix0 + kx) < (unsigned)p.w ? 1u : 0u; };
        auto in_h = [&](int ky) { return (unsigned)(iy0 + ky) < (unsigned)p.h; };
        if (KS3 || p.ks == 3) {  // straight-line for the filter sizes of the networks (a runtime-ks loop is ~100 instructions)
          const unsigned cols = in_w(0) | in_w(1) << 1 | in_w(2) << 2;
          ok = (in_h(0) ? cols : 0u) | (in_h(1) ? cols << 3 : 0u) | (in_h(2) ? cols << 6 : 0u);
        } else if (p.ks == 1) {
          ok = in_h(0) ? in_w(0) : 0u;
        } else {
          unsigned cols = 0;
          for (int kx = 0; kx < p.ks; ++kx) cols |= in_w(kx) << kx;
          for (int ky = 0; ky < p.ks; ++ky) ok |= (in_h(ky) ? cols : 0u) << (ky * p.ks);
        }
        padmask = ~ok;
        const long long e = (long long)(nimg - img0) * img_elems + ((long long)iy0 * p.w + ix0) * p.x_pitch + bias_elems;
        v_base[j] = (unsigned)(e * 4) + 16u * q;
      }
      v_pad[j] = padmask;
    } else if (g < G) {
      const int co_local = row - BM;
      if (n0 + co_local < p.cout) v_base[j] = (unsigned)co_local * b_row + 16u * q;
    }
    v_cur[j] = BABL == 1 ? kOobOffset : v_base[j];
  }

  int s_begin = sid * p.sps;
  int s_end = (s_begin + p.sps < p.stages) ? s_begin + p.sps : p.stages;
  if constexpr (HYB) {
    if (piece < 0) {
      s_begin = 0;
      s_end = p.stages;
    }
  }
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  constexpr unsigned STAGE_B = STAGE_F * 4u;
  const unsigned wave_lds = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)wave * 1024u);  // SGPR
  const unsigned pitch4 = (unsigned)p.x_pitch * 4u;
  const int sw = (r32 >> 2) & 3;
  const int off0 = ((0 + hh) ^ sw) * 4, off1 = ((2 + hh) ^ sw) * 4;
  const float* a_frag = smem + (wr * TM + r32) * BK;        // this lane's A / B fragment rows in stage slot 0
  const float* b_frag = smem + (BM + wc * TN + r32) * BK;

  auto compute_stage = [&](const float* Ab, const float* Bb) {
    if (MT * NT >= 4) __builtin_amdgcn_s_setprio(1);
    // both halves' fragments are requested up front (the stage has landed), so the LDS latency of the
    // second half hides behind the first half's MFMAs
    float4 af[2][MT], bf[2][NT];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int off = half ? off1 : off0;
#pragma unroll
      for (int i = 0; i < MT; ++i) af[half][i] = *reinterpret_cast<const float4*>(Ab + i * 32 * BK + off);
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[half][j] = *reinterpret_cast<const float4*>(Bb + j * 32 * BK + off);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the ds_reads ahead of the MFMAs (hipcc sinks them otherwise)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      // k-major issue order: consecutive MFMAs hit different accumulators
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const float4 av = af[half][i], bv = bf[half][j];
            const float a = kk == 0 ? av.x : kk == 1 ? av.y : kk == 2 ? av.z : av.w;
            const float b = kk == 0 ? bv.x : kk == 1 ? bv.y : kk == 2 ? bv.z : bv.w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
          }
    }
    if (MT * NT >= 4) __builtin_amdgcn_s_setprio(0);
  };

  // Stage s lives in slot s % 3.  A "step" = wait for stage s (only stage s+1's DMAs are younger), barrier
  // (everyone's stage-s DMAs landed, everyone finished reading stage s-1), refill the slot stage s-1 vacated with
  // stage s+2, multiply stage s.
  if constexpr (KORD == 0) {
    // Tap-major K walk (tap outer, 16-channel chunk inner; wave-uniform): the per-lane work of entering a tap (padding
    // select) is paid once per tap.  A workgroup sweeps its input rows once per tap, so those rows have to survive in L2
    // from one sweep to the next - they do as long as the weights leave room (choose_order).
    int tap = 0, cc = 0, ky = 0, kx = 0;
    // Column-class tap masks (me_conv_desc.tap_mask): the K walk of this tile visits only the set taps of its class.
    unsigned tmask = 0xFFFFFFFFu;
    if constexpr (MASKED) {   // (whole tiles only: the host refuses a K split with masks; a class is a multiple of BN columns wide)
      tmask = p.tapmask[n0 / p.mask_cols] & ((1u << (p.ks * p.ks)) - 1u);
      tap = __builtin_ctz(tmask);
      ky = tap / p.ks;
      kx = tap - ky * p.ks;
      s_begin = 0;
      s_end = __builtin_popcount(tmask) * p.cs;
    } else if (s_begin != 0) {  // K-split pieces only: whole tiles skip the divisions
      tap = s_begin / p.cs;
      cc = s_begin - tap * p.cs;
      ky = tap / p.ks;
      kx = tap - ky * p.ks;
    }
    unsigned a_off = 0, b_off = 0;  // scalar byte offsets of the next stage to issue
    auto enter_tap = [&]() {        // VALU work only here: once per filter tap
#pragma unroll
      for (int j = 0; j < LA; ++j) v_cur[j] = (BABL == 1 || ((v_pad[j] >> (MASKED ? (tap & 31) : tap)) & 1u)) ? kOobOffset : v_base[j];
      a_off = (unsigned)(ky * p.w + kx) * pitch4;
      b_off = (unsigned)tap * (unsigned)p.cs * b_step;
    };
    enter_tap();
    a_off += (unsigned)cc * (BK * 4u);
    b_off += (unsigned)cc * b_step;
    auto issue_stage = [&](unsigned lds_dst) {
      // (readfirstlane: a no-op on values that already live in SGPRs, a guard where the compiler moved the walk to VGPRs)
      dma_stage<LPW, LA, NW * 1024>(v_cur, rsrc_a, rsrc_b, __builtin_amdgcn_readfirstlane(a_off),
                                    __builtin_amdgcn_readfirstlane(b_off), lds_dst);
      a_off += BK * 4u;
      b_off += b_step;
      if (++cc == p.cs) {
        cc = 0;
        do {   // (one trip without masks)
          ++tap;
          if (++kx == p.ks) {
            kx = 0;
            ++ky;
          }
        } while (MASKED && tap < p.ks * p.ks && !((tmask >> tap) & 1u));
        enter_tap();
      }
    };
    const int nstages = s_end - s_begin;
    for (int i = 0; i < DEPTH && i < nstages; ++i) issue_stage(wave_lds + (unsigned)i * STAGE_B);
    // wait until only `younger` stages of this wave's DMAs are outstanding (vmcnt wants an immediate)
    auto wait_landed = [&](int younger) {
      bool done = false;
      static_for<DEPTH>([&](auto r_c) {
        constexpr int R = decltype(r_c)::value;
        if (!done && younger <= R) {
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPW * R) : "memory");
          done = true;
        }
      });
      if (!done) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPW * (DEPTH - 1)) : "memory");
    };
    // The bulk runs NST steps per loop trip with compile-time slots (LDS offsets are immediates, ~19 scalar
    // instructions per step); the last stages go through the generic tail below.
    auto step = [&](auto slot_c) {
      constexpr int SLOT = decltype(slot_c)::value;
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPW * (DEPTH - 1)) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue_stage(wave_lds + ((SLOT + DEPTH) % NST) * STAGE_B);
      compute_stage(a_frag + SLOT * STAGE_F, b_frag + SLOT * STAGE_F);
    };
    int s = 0;
    for (; s + NST <= nstages - DEPTH; s += NST) static_for<NST>(step);
    int slot = 0;  // s is a multiple of NST here
    for (; s < nstages; ++s) {
      wait_landed(nstages - 1 - s);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (s + DEPTH < nstages) issue_stage(wave_lds + (slot == 0 ? NST - 1 : slot - 1) * STAGE_B);
      compute_stage(a_frag + slot * STAGE_F, b_frag + slot * STAGE_F);
      slot = slot == NST - 1 ? 0 : slot + 1;
    }
  } else {
    // Chunk-major K walk, 3x3 filters only (chunk outer, the nine taps inner and unrolled: stage = 9 * chunk + tap, s_begin and
    // s_end are multiples of 9).  The nine taps of a chunk re-read the same 64 bytes of every input row back to back, so the
    // input is fetched into L2 once however little room it has there - the order for the deep layers, whose weight
    // panels own the L2 (choose_order).  Per-tap DMA operands are set up once: the per-lane offsets (padding resolved) in
    // registers, the tap offsets in SGPRs; a stage costs two scalar adds.
    unsigned v_tap[9][LPW];
    unsigned a_tap[9], b_tap[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int j = 0; j < LPW; ++j)
        v_tap[t][j] = j < LA ? ((BABL == 1 || ((v_pad[j < LA ? j : 0] >> t) & 1u)) ? kOobOffset : v_base[j]) : v_cur[j];
      a_tap[t] = (unsigned)((t / 3) * p.w + (t % 3)) * pitch4;
      b_tap[t] = (unsigned)t * (unsigned)p.cs * b_step;
    }
    const int c_begin = s_begin / 9, c_end = s_end / 9;
    unsigned ca = (unsigned)c_begin * (BK * 4u), cb = (unsigned)c_begin * b_step;  // offsets of the chunk being multiplied
    auto issue = [&](auto tap_c, unsigned ca_, unsigned cb_, unsigned lds_dst) {
      constexpr int T = decltype(tap_c)::value;
      dma_stage<LPW, LA, NW * 1024>(v_tap[T], rsrc_a, rsrc_b, __builtin_amdgcn_readfirstlane(a_tap[T] + ca_),
                                    __builtin_amdgcn_readfirstlane(b_tap[T] + cb_), lds_dst);
    };
    auto step = [&](auto tap_c, auto last_c) {
      constexpr int T = decltype(tap_c)::value;
      constexpr bool LAST = decltype(last_c)::value;  // the last chunk of this workgroup: nothing to prefetch behind it
      constexpr int SLOT = T % NST;
      constexpr int YOUNGER = (LAST && 8 - T < DEPTH - 1) ? 8 - T : DEPTH - 1;  // stages issued behind stage T, still in flight
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPW * YOUNGER) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      constexpr unsigned DST = ((SLOT + DEPTH) % NST) * STAGE_B;
      if constexpr (T + DEPTH < 9)
        issue(std::integral_constant<int, (T + DEPTH) % 9>{}, ca, cb, wave_lds + DST);
      else if constexpr (!LAST)
        issue(std::integral_constant<int, (T + DEPTH) % 9>{}, ca + BK * 4u, cb + b_step, wave_lds + DST);
      compute_stage(a_frag + SLOT * STAGE_F, b_frag + SLOT * STAGE_F);
    };
    auto chunk = [&](auto last_c) { static_for<9>([&](auto tap_c) { step(tap_c, last_c); }); };
    static_for<DEPTH>([&](auto tap_c) { issue(tap_c, ca, cb, wave_lds + decltype(tap_c)::value * STAGE_B); });
    for (int c = c_begin; c + 1 < c_end; ++c) {
      chunk(std::false_type{});
      ca += BK * 4u;
      cb += b_step;
    }
    chunk(std::true_type{});
  }

  // ---- epilogue (same as conv_igemm_dma_f32) ----------------------------------------------------
  // K-split workgroups (split-K: grid.y; tail split: pieces) park their raw accumulators in slabs - [split][M][cout], or
  // compact [piece][BM][BN] for the tail split.  With arrival counters (p.counters) the last workgroup of a tile to arrive
  // reads the slabs back in the fixed order 0..k-1 and goes on to the fused epilogue; without, a second launch does that.
  // Hand-over in the write-through form of the in-launch split-K recipe (cdna_hip_programming.md section 6, guideline 16):
  // sc1 slab stores (relaxed agent-scope atomic stores of 4 bytes), drained by every wave, barrier, one lane draws the
  // ticket; the last one's workgroup reads the slabs with sc1 loads.  No L2 write-back / invalidate fences.
  const bool split = HYB ? piece >= 0 : p.splitk > 1;
  if (split) {
    const long long slab_stride = HYB ? (long long)(BM * BN) : (long long)p.M * p.cout;
    float* slab0 = HYB ? p.partial + (long long)(piece - sid) * (BM * BN) : p.partial;  // split 0 of this tile
    auto slab_index = [&](int i, int j, int e, long long& idx) {
      const int lm = wr * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh, lc = wc * TN + j * 32 + r32;
      if (HYB) {
        idx = (long long)lm * BN + lc;
        return true;
      }
      idx = (long long)(m0 + lm) * p.cout + (n0 + lc);
      return m0 + lm < p.M && n0 + lc < p.cout;
    };
    float* mine = slab0 + (long long)sid * slab_stride;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          long long idx;
          if (!slab_index(i, j, e, idx)) continue;
          if (p.counters)  // write-through (sc1) store: visible to every XCD once this wave's vmcnt drains, no L2 write-back fence
            __hip_atomic_store(mine + idx, acc[i][j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else
            mine[idx] = acc[i][j][e];
        }
    if (p.counters == nullptr) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* last_flag = reinterpret_cast<int*>(smem);  // the stage buffers are idle now
    if (tid == 0) {
      int* cnt = p.counters + (HYB ? piece / p.splitk : tile_m * p.tiles_n + tile_n);
      const int ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == p.splitk - 1;
      if (last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // all arrived: ready for the next launch
      *last_flag = last;
    }
    __syncthreads();
    if (*last_flag == 0) return;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          long long idx;
          if (!slab_index(i, j, e, idx)) continue;
          // sc1 loads (coherent at agent scope: served past this XCD's L2 lines that may be stale)
          float a = __hip_atomic_load(slab0 + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int k = 1; k < p.splitk; ++k)
            a += __hip_atomic_load(slab0 + (long long)k * slab_stride + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          acc[i][j][e] = a;
        }
  }
  // Fast path (every tile but the ragged ones at the end of M / cout; no upsample, no sigmoid): no per-element bounds tests,
  // output / residual addresses by pointer increments (rows of an accumulator: 4 consecutive, then a jump of 5), leaky as
  // max(t, 0.1 t).  About 9 vector instructions per output instead of ~30 - they run beside the other waves' MFMAs at a cost
  // of matrix-pipe issue slots, and a 1x1 layer's tile has only 128 MFMAs per wave to amortise them over.
  if (m0 + BM <= p.M && n0 + BN <= p.cout && p.ups == 1 && p.act != ME_ACT_SIGMOID) {
    const float slope = p.act == ME_ACT_LEAKY ? 0.1f : 1.0f;
    const long long ystep = p.y_pitch, rstep = p.res_pitch;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = n0 + wc * TN + j * 32 + r32;
      const float sc = p.scale[co], sh = p.shift[co];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const long long row0 = m0 + wr * TM + i * 32 + 4 * hh;
        float* yp = p.y + row0 * ystep + co;
        if (p.res) {
          const float* rp = p.res + row0 * rstep + co;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float t = acc[i][j][e] * sc + sh;
            me::store4(yp, fmaxf(t, t * slope) + *rp, p.store_mode);
            yp += (e & 3) == 3 ? 5 * ystep : ystep;
            rp += (e & 3) == 3 ? 5 * rstep : rstep;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float t = acc[i][j][e] * sc + sh;
            me::store4(yp, fmaxf(t, t * slope), p.store_mode);
            yp += (e & 3) == 3 ? 5 * ystep : ystep;
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = n0 + wc * TN + j * 32 + r32;
    const bool co_ok = co < p.cout;
    const float sc = co_ok ? p.scale[co] : 0.f;
    const float sh = co_ok ? p.shift[co] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * hh;
        const int m = m0 + wr * TM + i * 32 + row;
        if (!co_ok || m >= p.M) continue;
        float v = apply_act(acc[i][j][e] * sc + sh, p.act);
        if (p.res) v += p.res[(long long)m * p.res_pitch + co];
        if (p.ups == 1) {
          p.y[(long long)m * p.y_pitch + co] = v;
        } else {
          const int nimg = m / hw;
          const int rem = m - nimg * hw;
          const int oy = rem / p.wo, ox = rem - oy * p.wo;
          const int W2 = p.wo * 2;
          const long long base = ((long long)nimg * (p.ho * 2) + 2 * oy) * W2 + 2 * ox;
          p.y[(base)*p.y_pitch + co] = v;
          p.y[(base + 1) * p.y_pitch + co] = v;
          p.y[(base + W2) * p.y_pitch + co] = v;
          p.y[(base + W2 + 1) * p.y_pitch + co] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// split-K second pass: sum the slabs in a fixed order (deterministic), then the fused epilogue.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_splitk_reduce_f32(ConvP p) {
  const long long total = (long long)p.M * p.cout;
  const int hw = p.ho * p.wo;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int co = (int)(idx % p.cout);
    const int m = (int)(idx / p.cout);
    float a = p.partial[idx];
    for (int k = 1; k < p.splitk; ++k) a += p.partial[(long long)k * total + idx];
    float v = apply_act(a * p.scale[co] + p.shift[co], p.act);
    if (p.res) v += p.res[(long long)m * p.res_pitch + co];
    if (p.ups == 1) {
      p.y[(long long)m * p.y_pitch + co] = v;
    } else {
      const int nimg = m / hw;
      const int rem = m - nimg * hw;
      const int oy = rem / p.wo, ox = rem - oy * p.wo;
      const int W2 = p.wo * 2;
      const long long base = ((long long)nimg * (p.ho * 2) + 2 * oy) * W2 + 2 * ox;
      p.y[(base)*p.y_pitch + co] = v;
      p.y[(base + 1) * p.y_pitch + co] = v;
      p.y[(base + W2) * p.y_pitch + co] = v;
      p.y[(base + W2 + 1) * p.y_pitch + co] = v;
    }
  }
}

// tail-split second pass (conv_igemm_buf_f32<HYB = 1>): slabs [tail tile][split][bm][bn] -> fixed-order sum + fused epilogue.
__global__ __launch_bounds__(256) void conv_tail_reduce_f32(ConvP p, int bm, int bn) {
  const int tile_elems = bm * bn;
  const long long total = (long long)(p.tiles_m * p.tiles_n - p.bulk) * tile_elems;
  const int hw = p.ho * p.wo;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int t = (int)(idx / tile_elems);
    const int rem_e = (int)(idx - (long long)t * tile_elems);
    const int lm = rem_e / bn, lc = rem_e - lm * bn;
    int tile_m, tile_n;
    tile_coords(p, p.bulk + t, tile_m, tile_n);
    const int m = tile_m * bm + lm;
    const int co = tile_n * bn + lc;
    if (m >= p.M || co >= p.cout) continue;
    const float* src = p.partial + (long long)t * p.splitk * tile_elems + rem_e;
    float a = src[0];
    for (int k = 1; k < p.splitk; ++k) a += src[(long long)k * tile_elems];
    float v = apply_act(a * p.scale[co] + p.shift[co], p.act);
    if (p.res) v += p.res[(long long)m * p.res_pitch + co];
    if (p.ups == 1) {
      p.y[(long long)m * p.y_pitch + co] = v;
    } else {
      const int nimg = m / hw;
      const int rem = m - nimg * hw;
      const int oy = rem / p.wo, ox = rem - oy * p.wo;
      const int W2 = p.wo * 2;
      const long long base = ((long long)nimg * (p.ho * 2) + 2 * oy) * W2 + 2 * ox;
      p.y[(base)*p.y_pitch + co] = v;
      p.y[(base + 1) * p.y_pitch + co] = v;
      p.y[(base + W2) * p.y_pitch + co] = v;
      p.y[(base + W2 + 1) * p.y_pitch + co] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// direct 3x3 convolution, cin <= 4 (stem).  One thread = one output pixel x 8 output channels
// (cout/8 adjacent lanes share a pixel, so a wave stores one contiguous NHWC span); weights are
// broadcast from LDS.  CIN is the padded channel count (3 or 4); channels >= p.cin are zero.
// ---------------------------------------------------------------------------------------------
constexpr int SC_MAXCOUT = 128;

template <int CIN>
__global__ __launch_bounds__(256) void conv_smallcin_f32(ConvP p) {
  constexpr int K = 9 * CIN;
  __shared__ __attribute__((aligned(16))) float w_lds[K * SC_MAXCOUT];  // [k][cout_pad]
  const int cout_pad = (p.cout + 7) & ~7;
  for (int idx = threadIdx.x; idx < K * cout_pad; idx += 256) {
    const int k = idx / cout_pad, co = idx - k * cout_pad;
    const int t = k / CIN, c = k - t * CIN;
    w_lds[idx] = (co < p.cout && c < p.cin) ? p.wgt[(long long)co * (9 * p.cin) + t * p.cin + c] : 0.f;
  }
  __syncthreads();

  const int tpp = cout_pad >> 3;  // threads per pixel
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const int m = (int)(gid / tpp);
  const int co0 = (int)(gid - (long long)m * tpp) * 8;
  if (m >= p.M) return;
  const int hw = p.ho * p.wo;
  const int nimg = m / hw;
  const int rem = m - nimg * hw;
  const int oy = rem / p.wo, ox = rem - oy * p.wo;

  float xin[K];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
      const bool ok = ((unsigned)iy < (unsigned)p.h) && ((unsigned)ix < (unsigned)p.w);
#pragma unroll
      for (int c = 0; c < CIN; ++c) {
        float v = 0.f;
        if (ok && c < p.cin) {
          if (p.x_nchw)
            v = p.x[(((long long)nimg * p.cin + c) * p.h + iy) * p.w + ix];
          else
            v = p.x[((long long)(nimg * p.h + iy) * p.w + ix) * p.x_pitch + c];
        }
        xin[(ky * 3 + kx) * CIN + c] = v;
      }
    }

  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const float* wbase = w_lds + co0;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float4 w0 = *reinterpret_cast<const float4*>(wbase + k * cout_pad);
    const float4 w1 = *reinterpret_cast<const float4*>(wbase + k * cout_pad + 4);
    const float xv = xin[k];
    acc[0] = fmaf(xv, w0.x, acc[0]); acc[1] = fmaf(xv, w0.y, acc[1]);
    acc[2] = fmaf(xv, w0.z, acc[2]); acc[3] = fmaf(xv, w0.w, acc[3]);
    acc[4] = fmaf(xv, w1.x, acc[4]); acc[5] = fmaf(xv, w1.y, acc[5]);
    acc[6] = fmaf(xv, w1.z, acc[6]); acc[7] = fmaf(xv, w1.w, acc[7]);
  }
  float* yrow = p.y + (long long)m * p.y_pitch + co0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int co = co0 + j;
    if (co < p.cout) acc[j] = apply_act(acc[j] * p.scale[co] + p.shift[co], p.act);
  }
  if (co0 + 8 <= p.cout && (p.y_pitch & 3) == 0) {
    *reinterpret_cast<float4*>(yrow) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(yrow + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (co0 + j < p.cout) yrow[j] = acc[j];
  }
}

// ---------------------------------------------------------------------------------------------
// host side: (tile, split-K) planning + launch
// ---------------------------------------------------------------------------------------------
struct TileCfg {
  int id, bm, bn, bk;
  int occ;    // workgroups resident per CU (LDS / register limited)
  int wps;    // waves per SIMD contributed by one resident workgroup (1 for 4-wave, 2 for 8-wave tiles)
  float eff;  // steady-state efficiency of the tile shape relative to 128x128
};
const TileCfg kTiles[] = {
    // LDS per workgroup = 3 stages x (BM + BN) x 64 B: 48 / 36 / 24 / 36 / 72 KiB
    {1, 128, 128, 16, 3, 1, 1.00f},
    {2, 128, 64, 16, 4, 1, 0.95f},
    {3, 64, 64, 16, 6, 1, 0.85f},
    {4, 128, 32, 16, 4, 1, 0.75f},
};
const TileCfg kExtraTiles[] = {  // forced ids only (engine autotuner, tools/conv_bench.py)
    {5, 256, 128, 16, 2, 2, 1.02f},  {6, 256, 128, 16, 1, 2, 1.0f},   {7, 64, 64, 16, 2, 1, 0.85f},   {21, 128, 128, 16, 3, 1, 1.0f}, {22, 128, 64, 16, 4, 1, 0.9f},
    {23, 64, 64, 16, 8, 1, 0.85f},   {24, 128, 32, 16, 7, 1, 0.75f},  {25, 256, 128, 16, 2, 2, 1.02f},
    {51, 128, 128, 16, 3, 1, 1.0f},  {52, 128, 64, 16, 5, 1, 0.9f},  {53, 64, 64, 16, 8, 1, 0.85f},
    {54, 128, 32, 16, 7, 1, 0.75f},  {55, 128, 128, 32, 2, 1, 1.0f}, {31, 128, 128, 16, 3, 1, 1.0f},
    {11, 128, 128, 16, 3, 1, 1.0f},  {12, 128, 128, 16, 3, 1, 1.0f}, {13, 128, 128, 16, 3, 1, 1.0f},
    {61, 128, 128, 16, 3, 1, 1.0f},  {65, 256, 128, 16, 2, 2, 1.0f},
    {81, 128, 128, 16, 3, 1, 1.0f},  {82, 128, 64, 16, 4, 1, 0.9f},  {83, 64, 64, 16, 8, 1, 0.85f}, {85, 256, 128, 16, 2, 2, 1.0f},
};
constexpr int kMaxSplit = 16;

struct ConvPlan {
  int tile;
  int splitk;
};

const TileCfg* find_tile(int id) {
  for (const TileCfg& t : kTiles)
    if (t.id == id) return &t;
  for (const TileCfg& t : kExtraTiles)
    if (t.id == id) return &t;
  return nullptr;
}

// Makespan model.  The busiest CU gets ceil(blocks / 256) workgroups; up to `occ` of them are
// co-resident and share its matrix pipe, whose utilisation grows with the number of resident waves
// (fitted to tools/conv_bench.py sweeps on MI355X, profiles/r01_conv_bench_*: 0.68 / 0.76 / 0.80 /
// 0.84 for 1 / 2 / 3 / 4+ resident workgroups; rms error of the model 6 %).  Split-K multiplies the
// number of workgroups, divides their length and adds a slab round trip that mostly stays in L2 /
// Infinity Cache (~10 TB/s effective).
ConvPlan plan_conv(const ConvP& p, int forced_tile, int max_split) {
  static const double kUtil[5] = {0.68, 0.68, 0.76, 0.80, 0.84};
  ConvPlan best{1, 1};
  double best_cost = 1e300;
  const double flop_per_clk_cu = 256.0;  // 4 SIMDs x 64 FLOP/clk (v_mfma_f32_32x32x2_f32)
  auto consider = [&](const TileCfg& t) {
    const long long tm = (p.M + t.bm - 1) / t.bm, tn = (p.cout + t.bn - 1) / t.bn;
    const int cs = (p.cin + t.bk - 1) / t.bk;
    const int stages = p.ks * p.ks * cs;
    for (int split = 1; split <= max_split && split <= stages; ++split) {
      const int sps = (stages + split - 1) / split;
      if ((split - 1) * sps >= stages) continue;  // an empty split
      const long long blocks = tm * tn * split;
      const long long per_cu = (blocks + 255) / 256;
      // cycles: full rounds of `occ` resident workgroups, then the remainder at its own utilisation;
      // each workgroup is 2*BM*BN*BK*sps flops through one CU's matrix pipe
      const double wg = 2.0 * t.bm * t.bn * t.bk * sps / flop_per_clk_cu;
      const long long full = per_cu / t.occ, rem = per_cu % t.occ;
      auto util = [&](long long resident) {
        const long long w = resident * t.wps;
        return kUtil[w > 4 ? 4 : w] * t.eff;
      };
      double cost = full * t.occ * wg / util(t.occ);
      if (rem) cost += rem * wg / util(rem);
      cost += 3000.0;  // prologue / epilogue latency of a workgroup chain
      if (split > 1) {
        // slab round trip: stays in L2 / Infinity Cache (~10 TB/s) while small, HBM speed (~3 TB/s) beyond
        const double bytes = (double)(split + 1) * p.M * p.cout * 4.0 * 2.0;
        const double bw = bytes < 256.0e6 ? 1.0e13 : 3.0e12;
        cost += bytes / bw * 2.1e9 + 6000.0;  // + the extra launch (~3 us)
      }
      if (cost < best_cost) {
        best_cost = cost;
        best = ConvPlan{t.id, split};
      }
    }
  };
  if (forced_tile) {
    const TileCfg* t = find_tile(forced_tile);
    if (t) consider(*t);
    best.tile = forced_tile;
  } else {
    for (const TileCfg& t : kTiles) consider(t);
  }
  return best;
}

// K split of conv_igemm_buf_f32: p.sps stages per split; the chunk-major walk splits on chunk boundaries (nine stages).
inline void plan_split(ConvP& p) {
  const int unit = p.kord ? 9 : 1;
  const int units = p.stages / unit;
  if (p.splitk > units) p.splitk = units;
  if (p.splitk < 1) p.splitk = 1;
  p.sps = (units + p.splitk - 1) / p.splitk * unit;
  while (p.splitk > 1 && (p.splitk - 1) * p.sps >= p.stages) --p.splitk;  // no empty split
}

template <int BM, int BN, int BK, int WR, int WC, int ABL = 0, int PRIO = 0>
int launch_igemm(ConvP& p, hipStream_t stream) {
  p.cs = (p.cin + BK - 1) / BK;
  p.stages = p.ks * p.ks * p.cs;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.cout + BN - 1) / BN;
  if (p.splitk > p.stages) p.splitk = p.stages;
  p.sps = (p.stages + p.splitk - 1) / p.splitk;
  while (p.splitk > 1 && (p.splitk - 1) * p.sps >= p.stages) --p.splitk;  // no empty split
  const size_t lds = (size_t)2 * (BM + BN) * (BK + 4) * sizeof(float);
  auto kern = conv_igemm_f32<BM, BN, BK, WR, WC, ABL, PRIO>;
  if (lds > 64 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
      attr_set = true;
    }
  }
  const long long blocks = (long long)p.tiles_m * p.tiles_n;
  ME_REQUIRE(blocks < (1ll << 31), ME_E_TOOBIG, "me_conv2d_f32: grid too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)p.splitk), dim3(256), lds, stream, p);
  int rc = me::check_launch("conv_igemm_f32");
  if (rc || p.splitk == 1) return rc;
  long long rb = ((long long)p.M * p.cout + 255) / 256;
  if (rb > 256 * 16) rb = 256 * 16;
  hipLaunchKernelGGL(conv_splitk_reduce_f32, dim3((unsigned)rb), dim3(256), 0, stream, p);
  return me::check_launch("conv_splitk_reduce_f32");
}

template <int BM, int BN, int WR, int WC, int DPRIO = 0, int MINW = 1, int DABL = 0>
int launch_dma(ConvP& p, hipStream_t stream) {
  ME_REQUIRE(p.ks * p.ks <= 32, ME_E_BADARG, "conv_igemm_dma_f32: filters larger than 5x5 need the register-staged tiles (51-55)");
  constexpr int BK = 16;
  p.cs = (p.cin + BK - 1) / BK;
  p.stages = p.ks * p.ks * p.cs;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.cout + BN - 1) / BN;
  plan_split(p);
  constexpr int NW = WR * WC;
  constexpr int LPW = ((BM + BN) / 16 + NW - 1) / NW;
  const size_t lds = (size_t)3 * LPW * NW * 256 * sizeof(float);
  auto kern = conv_igemm_dma_f32<BM, BN, WR, WC, DPRIO, MINW, DABL>;
  if (lds > 64 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
      attr_set = true;
    }
  }
  const long long blocks = (long long)p.tiles_m * p.tiles_n;
  ME_REQUIRE(blocks < (1ll << 31), ME_E_TOOBIG, "me_conv2d_f32: grid too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)p.splitk), dim3(64 * NW), lds, stream, p);
  int rc = me::check_launch("conv_igemm_dma_f32");
  if (rc || p.splitk == 1) return rc;
  long long rb = ((long long)p.M * p.cout + 255) / 256;
  if (rb > 256 * 16) rb = 256 * 16;
  hipLaunchKernelGGL(conv_splitk_reduce_f32, dim3((unsigned)rb), dim3(256), 0, stream, p);
  return me::check_launch("conv_splitk_reduce_f32");
}

// Can conv_igemm_buf_f32 address this problem?  cin must be a multiple of 16 and every in-range byte offset
// (+ the scalar stage offset) must stay below 2^31 (the descriptor's num_records).
template <int BM>
bool buf_addressable(const ConvP& p) {
  if (p.cin % 16 != 0 || p.x_nchw) return false;
  const long long hw = (long long)p.ho * p.wo;
  const long long span_imgs = (BM - 1) / hw + 2;  // images one tile of BM consecutive output pixels can touch
  const long long img_bytes = (long long)p.h * p.w * p.x_pitch * 4;
  const long long tap_bytes = ((long long)p.ks * p.w + p.ks) * p.x_pitch * 4 + (long long)p.cin * 4;
  const long long a_max = span_imgs * img_bytes + 2 * tap_bytes;
  const long long b_max = 256ll * p.ktot * 4 + (long long)p.ktot * 4;
  const long long b_tiled_max = p.wgt_tiled ? (long long)p.ktot * p.cout * 4 + 256ll * 64 : 0;
  return a_max < (1ll << 31) && b_max < (1ll << 31) && b_tiled_max < (1ll << 31) && (long long)p.x_pitch * 4 < (1ll << 31);
}

// (m, s) with n / d == (umulhi(n, m) + n) >> s for every n < 2^31
void magic_u32(unsigned d, unsigned* m, unsigned* s) {
  unsigned sh = 0;
  while ((1ull << sh) < d) ++sh;
  *s = sh;
  *m = (unsigned)(((1ull << 32) * ((1ull << sh) - d)) / d + 1);
}

// Walk order of conv_igemm_buf_f32 (p.tiles_n set).
//  * 3x3 layers take the chunk-major K walk (nine unrolled taps per 16-channel chunk: two scalar adds per stage instead of
//    the tap-major walk's counters, compares and branches; the input rows are re-read within nine consecutive stages).
//  * Layers whose weights fit the L2 beside everything else keep the row-major tile order.  For the deep layers (weights of
//    several MB, e.g. 18.9 MB for 3x3 512->1024) the workgroups of an XCD drift apart along K, the working set becomes
//    "all the weights" and every workgroup streams its own copy from the Infinity Cache (rocprofv3 FETCH_SIZE: 1.05 GB per
//    launch against 52 MB of operands): there the tiles are walked in column panels whose weight rows (<= kPanelBytes)
//    stay L2-resident while the input - fetched once per chunk thanks to the chunk-major walk - streams past
//    (profiles/r02_layer_traffic_f32.txt: 1045 -> 177 MB per launch at 13x13, 451 -> 134 MB at 26x26, and 2-3 % faster).
template <int BN>
void choose_order(ConvP& p) {
  static const long long kPanelBytes = [] {
    const char* e = getenv("MILLIEYE_PANEL_KB");
    return (e && atoi(e) > 0 ? (long long)atoi(e) : 1300ll) * 1024;
  }();
  static const int kChunkMajor = [] {
    const char* e = getenv("MILLIEYE_KORD");  // experiments: 0 = tap-major everywhere, 1 = chunk-major only with panels
    return e ? atoi(e) : 2;
  }();
  const long long col_bytes = (long long)BN * p.ktot * 4;  // weights of one tile column
  p.gn = p.tiles_n;
  // the chunk-major walk is built for nine taps; with two chunks (cin 32: the 416 / 208 maps) it measured 2 % slower
  p.kord = (p.ks == 3 && p.cs >= 4 && kChunkMajor == 2) ? 1 : 0;
  if (col_bytes * p.tiles_n > kPanelBytes) {
    const long long fit = kPanelBytes / col_bytes;
    p.gn = fit < 1 ? 1 : fit > p.tiles_n ? p.tiles_n : (int)fit;
    if (p.ks == 3 && kChunkMajor >= 1) p.kord = 1;
  }
}

template <int BM, int BN, int WR, int WC, int MINW, int BABL, int HYB, int KORD, int DEEP = 0, int MASKED = 0>
int launch_buf_kernel(const ConvP& p, dim3 grid, hipStream_t stream) {
  constexpr int NW = WR * WC;
  constexpr int LPW = ((BM + BN) / 16 + NW - 1) / NW;
  constexpr int NST = DEEP ? (KORD ? 9 : 6) : 3;
  const size_t lds = (size_t)NST * LPW * NW * 256 * sizeof(float);
  auto kern = conv_igemm_buf_f32<BM, BN, WR, WC, MINW, BABL, HYB, KORD, DEEP, MASKED>;
  if (lds > 64 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
      attr_set = true;
    }
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, stream, p);
  return me::check_launch(HYB ? "conv_igemm_buf_f32<tail split>" : "conv_igemm_buf_f32");
}

template <int BM, int BN, int WR, int WC, int MINW = 1, int BABL = 0, int DEEP = 0>
int launch_buf(ConvP& p, hipStream_t stream) {
  if (p.ks * p.ks > 32) {  // the DMA kernels keep a 32-bit "tap is padding" mask per lane: larger filters (6x6, 7x7)
    if constexpr (WR * WC == 4)  // go through the register-staged kernel, which tests bounds per stage
      return launch_igemm<BM, BN, 16, WR, WC>(p, stream);
    else
      return launch_igemm<128, 128, 16, 2, 2>(p, stream);
  }
  if (p.mask_cols)
    ME_REQUIRE(buf_addressable<BM>(p) && p.mask_cols % BN == 0 && BABL == 0 && DEEP == 0, ME_E_BADARG,
               "me_conv2d_f32: tap masks need cin %% 16 == 0, offsets below 2^31 and a tile width (%d) that divides tap_mask_cols", BN);
  if (!buf_addressable<BM>(p)) return launch_dma<BM, BN, WR, WC, 1, MINW>(p, stream);
  constexpr int BK = 16;
  p.cs = (p.cin + BK - 1) / BK;
  p.stages = p.ks * p.ks * p.cs;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.cout + BN - 1) / BN;
  choose_order<BN>(p);
  if (BABL || p.mask_cols) p.kord = 0;  // the ablation kernels exist for the tap-major walk only
  plan_split(p);
  const long long blocks = (long long)p.tiles_m * p.tiles_n;
  ME_REQUIRE(blocks < (1ll << 31), ME_E_TOOBIG, "me_conv2d_f32: grid too large");
  const dim3 grid((unsigned)blocks, (unsigned)p.splitk);
  if (blocks > p.counters_len || p.splitk == 1) p.counters = nullptr;  // in-launch slab reduction needs a counter per tile
  int rc;
  if constexpr (BABL == 0) {
    if (p.mask_cols) {
      if constexpr (DEEP == 0) rc = launch_buf_kernel<BM, BN, WR, WC, MINW, 0, 0, 0, 0, 1>(p, grid, stream);
      else rc = ME_E_BADARG;
    } else
      rc = p.kord ? launch_buf_kernel<BM, BN, WR, WC, MINW, 0, 0, 1, DEEP>(p, grid, stream)
                  : launch_buf_kernel<BM, BN, WR, WC, MINW, 0, 0, 0, DEEP>(p, grid, stream);
  } else {
    rc = launch_buf_kernel<BM, BN, WR, WC, MINW, BABL, 0, 0>(p, grid, stream);
  }
  if (rc || p.splitk == 1 || p.counters) return rc;
  long long rb = ((long long)p.M * p.cout + 255) / 256;
  if (rb > 256 * 16) rb = 256 * 16;
  hipLaunchKernelGGL(conv_splitk_reduce_f32, dim3((unsigned)rb), dim3(256), 0, stream, p);
  return me::check_launch("conv_splitk_reduce_f32");
}

// Tail-split launch (tile ids 41-45): whole tiles for the rounds that fill every CU, the rest cut p.splitk ways along K.
template <int BM, int BN, int WR, int WC, int MINW = 1, int DEEP = 0>
int launch_buf_tail(ConvP& p, hipStream_t stream, long long ws_bytes) {
  ME_REQUIRE(p.ks * p.ks <= 32 && buf_addressable<BM>(p), ME_E_BADARG,
             "me_conv2d_f32: the tail-split tiles need cin %% 16 == 0 and offsets below 2^31");
  constexpr int BK = 16;
  p.cs = (p.cin + BK - 1) / BK;
  p.stages = p.ks * p.ks * p.cs;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.cout + BN - 1) / BN;
  choose_order<BN>(p);
  const long long tiles = (long long)p.tiles_m * p.tiles_n;
  ME_REQUIRE(tiles < (1ll << 24), ME_E_TOOBIG, "me_conv2d_f32: grid too large");
  plan_split(p);
  p.bulk = (int)(tiles / 256 * 256);
  const long long tail = tiles - p.bulk;
  if (tail == 0 || p.splitk <= 1) {  // nothing to cut: the plain launch
    p.splitk = 1;
    p.bulk = 0;
    return launch_buf<BM, BN, WR, WC, MINW, 0, DEEP>(p, stream);
  }
  const long long need = tail * p.splitk * BM * BN * (long long)sizeof(float);
  ME_REQUIRE(p.partial && ws_bytes >= need, ME_E_BADARG, "me_conv2d_f32: tail split %d needs a workspace of %lld bytes",
             p.splitk, need);
  const dim3 grid((unsigned)(p.bulk + tail * p.splitk));
  if (tail > p.counters_len) p.counters = nullptr;
  int rc = p.kord ? launch_buf_kernel<BM, BN, WR, WC, MINW, 0, 1, 1, DEEP>(p, grid, stream)
                  : launch_buf_kernel<BM, BN, WR, WC, MINW, 0, 1, 0, DEEP>(p, grid, stream);
  if (rc || p.counters) return rc;
  long long rb = (tail * BM * BN + 255) / 256;
  if (rb > 256 * 16) rb = 256 * 16;
  hipLaunchKernelGGL(conv_tail_reduce_f32, dim3((unsigned)rb), dim3(256), 0, stream, p, BM, BN);
  return me::check_launch("conv_tail_reduce_f32");
}

int fill_params(const me_conv_desc* d, ConvP& p) {
  p.store_mode = me::store_mode();
  ME_REQUIRE(d != nullptr, ME_E_NULLPTR, "me_conv2d_f32: null descriptor");
  ME_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, ME_E_BADARG,
             "me_conv2d_f32: non-positive dimension");
  ME_REQUIRE(d->ksize >= 1 && d->stride >= 1 && d->pad >= 0, ME_E_BADARG, "me_conv2d_f32: bad ksize/stride/pad");
  const int ho = (d->h + 2 * d->pad - d->ksize) / d->stride + 1;
  const int wo = (d->w + 2 * d->pad - d->ksize) / d->stride + 1;
  ME_REQUIRE(ho == d->ho && wo == d->wo, ME_E_BADARG, "me_conv2d_f32: ho/wo (%d,%d) != derived (%d,%d)", d->ho,
             d->wo, ho, wo);
  ME_REQUIRE((long long)d->n * d->ho * d->wo < (1ll << 31), ME_E_TOOBIG, "me_conv2d_f32: too many output pixels");
  p.x = d->x; p.wgt = d->wgt; p.wgt_tiled = d->wgt_tiled; p.scale = d->scale; p.shift = d->shift; p.res = d->res; p.y = d->y;
  p.x_pitch = d->x_pitch; p.res_pitch = d->res_pitch; p.y_pitch = d->y_pitch;
  p.n = d->n; p.h = d->h; p.w = d->w; p.cin = d->cin; p.cout = d->cout; p.ks = d->ksize;
  p.stride = d->stride; p.pad = d->pad; p.ho = d->ho; p.wo = d->wo; p.act = d->act; p.ups = d->upsample;
  p.x_nchw = d->x_nchw;
  p.M = d->n * d->ho * d->wo;
  p.ktot = d->ksize * d->ksize * d->cin;
  magic_u32((unsigned)(d->ho * d->wo), &p.hw_m, &p.hw_s);
  magic_u32((unsigned)d->wo, &p.wo_m, &p.wo_s);
  p.cs = p.stages = p.tiles_m = p.tiles_n = p.bulk = p.gn = p.kord = 0;
  p.partial = nullptr;
  p.splitk = 1;
  p.sps = 0;
  p.counters = d->tile_counters_len > 0 ? d->tile_counters : nullptr;
  p.counters_len = p.counters ? d->tile_counters_len : 0;
  p.mask_cols = d->tap_mask_cols > 0 ? d->tap_mask_cols : 0;
  for (int i = 0; i < 4; ++i) p.tapmask[i] = d->tap_mask[i];
  return 0;
}

}  // namespace

extern "C" {

int64_t me_conv2d_flops(const me_conv_desc* d) {
  if (!d) return 0;
  return 2ll * d->n * d->ho * d->wo * (int64_t)d->cout * d->ksize * d->ksize * d->cin;
}

int64_t me_conv2d_workspace_bytes(const me_conv_desc* d) {
  ConvP p;
  if (!d || fill_params(d, p) != 0 || d->cin <= 4) return 0;
  if ((d->tile >= 41 && d->tile <= 45) || d->tile == 47) {  // tail split: compact slabs of the last partial round only
    static const int shape[7][2] = {{128, 128}, {128, 64}, {64, 64}, {128, 32}, {256, 128}, {0, 0}, {64, 64}};
    const int bm = shape[d->tile - 41][0], bn = shape[d->tile - 41][1];
    const long long tiles = (long long)((p.M + bm - 1) / bm) * ((p.cout + bn - 1) / bn);
    const int split = d->split_k > 0 ? d->split_k : 4;
    return (int64_t)(tiles % 256) * split * bm * bn * (int64_t)sizeof(float);
  }
  const ConvPlan plan = plan_conv(p, d->tile, d->split_k > 0 ? d->split_k : kMaxSplit);
  const int split = d->split_k > 0 ? d->split_k : plan.splitk;
  return split > 1 ? (int64_t)split * p.M * p.cout * (int64_t)sizeof(float) : 0;
}

int me_conv2d_f32(const me_conv_desc* d, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ConvP p;
  int rc = fill_params(d, p);
  if (rc) return rc;
  ME_REQUIRE(d->x && d->wgt && d->scale && d->shift && d->y, ME_E_NULLPTR, "me_conv2d_f32: null tensor pointer");
  ME_REQUIRE(d->upsample == 1 || d->upsample == 2, ME_E_BADARG, "me_conv2d_f32: upsample must be 1 or 2");
  ME_REQUIRE(d->act >= 0 && d->act <= 2, ME_E_BADARG, "me_conv2d_f32: unknown activation %d", d->act);
  ME_REQUIRE(d->y_pitch >= d->cout, ME_E_BADARG, "me_conv2d_f32: y_pitch < cout");

  if (d->cin <= 4) {
    ME_REQUIRE(d->ksize == 3, ME_E_BADARG, "me_conv2d_f32: cin <= 4 needs ksize 3 (direct stem kernel)");
    ME_REQUIRE(d->cout <= SC_MAXCOUT, ME_E_TOOBIG, "me_conv2d_f32: small-cin kernel supports cout <= %d", SC_MAXCOUT);
    ME_REQUIRE(d->res == nullptr && d->upsample == 1, ME_E_BADARG,
               "me_conv2d_f32: small-cin kernel has no residual/upsample epilogue");
    ME_REQUIRE(d->x_nchw || d->x_pitch >= d->cin, ME_E_BADARG, "me_conv2d_f32: x_pitch < cin");
    ME_REQUIRE(me::aligned16(d->y), ME_E_ALIGN, "me_conv2d_f32: y not 16-byte aligned");
    if (d->tile == 0 && me32::stem_mfma_eligible(p)) return me32::launch_stem_mfma(p, stream);  // tile 92 / 91: VALU versions
    if (d->cin == 3 && (d->y_pitch & 3) == 0 && d->tile != 91) {  // tile 91 forces the first version (A/B)
      const unsigned mb = (unsigned)((p.M + 255) / 256);
      if (d->cout % 32 == 0) {
        hipLaunchKernelGGL(conv_stem3_f32<32>, dim3(mb, d->cout / 32), dim3(256), 0, stream, p);
        return me::check_launch("conv_stem3_f32");
      }
      if (d->cout % 16 == 0) {
        hipLaunchKernelGGL(conv_stem3_f32<16>, dim3(mb, d->cout / 16), dim3(256), 0, stream, p);
        return me::check_launch("conv_stem3_f32");
      }
    }
    const long long threads = (long long)p.M * (((p.cout + 7) & ~7) >> 3);
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    if (d->cin == 3)
      hipLaunchKernelGGL(conv_smallcin_f32<3>, dim3(blocks), dim3(256), 0, stream, p);
    else
      hipLaunchKernelGGL(conv_smallcin_f32<4>, dim3(blocks), dim3(256), 0, stream, p);
    return me::check_launch("conv_smallcin_f32");
  }

  ME_REQUIRE(!d->x_nchw, ME_E_BADARG, "me_conv2d_f32: NCHW input only supported for cin <= 4");
  ME_REQUIRE(d->cin % 4 == 0, ME_E_BADARG, "me_conv2d_f32: cin %% 4 != 0 (cin=%d)", d->cin);
  ME_REQUIRE(d->x_pitch >= d->cin && d->x_pitch % 4 == 0, ME_E_ALIGN, "me_conv2d_f32: x_pitch must be >= cin, %% 4");
  ME_REQUIRE(me::aligned16(d->x) && me::aligned16(d->wgt), ME_E_ALIGN, "me_conv2d_f32: x / wgt not 16-byte aligned");
  ME_REQUIRE(!d->res || d->res_pitch >= d->cout, ME_E_BADARG, "me_conv2d_f32: res_pitch < cout");
  ME_REQUIRE(d->split_k >= 0 && d->split_k <= 64, ME_E_BADARG, "me_conv2d_f32: split_k out of range");

  if (p.mask_cols) {  // column-class tap masks (ABI 10): whole tiles of the buffer kernel's tap-major walk
    const int taps = d->ksize * d->ksize, classes = d->cout / p.mask_cols;
    ME_REQUIRE(taps <= 16 && d->cout % p.mask_cols == 0 && classes >= 1 && classes <= 4 && d->upsample == 1 && d->split_k <= 1 &&
                   d->tile >= 0 && d->tile <= 5 && d->cin % 16 == 0, ME_E_BADARG,
               "me_conv2d_f32: tap masks need ksize^2 <= 16, 1 - 4 column classes, upsample 1, split_k <= 1, tile 0 - 5, cin %% 16 == 0");
    for (int i = 0; i < classes; ++i)
      ME_REQUIRE((d->tap_mask[i] & ((1u << taps) - 1u)) != 0, ME_E_BADARG, "me_conv2d_f32: tap_mask[%d] selects no tap", i);
    p.splitk = 1;
    p.partial = nullptr;
    int tile = d->tile;
    if (tile == 0) tile = p.mask_cols % 128 == 0 ? 1 : p.mask_cols % 64 == 0 ? 2 : 4;
    switch (tile) {
      case 1: return launch_buf<128, 128, 2, 2>(p, stream);
      case 2: return launch_buf<128, 64, 2, 2>(p, stream);
      case 3: return launch_buf<64, 64, 2, 2>(p, stream);
      case 4: return launch_buf<128, 32, 4, 1>(p, stream);
      default: return launch_buf<256, 128, 4, 2, 4>(p, stream);
    }
  }

  if ((d->tile >= 41 && d->tile <= 45) || d->tile == 47) {  // tail split: split_k = pieces per tile of the last, partial round (0: 4)
    p.splitk = d->split_k > 0 ? d->split_k : 4;
    p.partial = reinterpret_cast<float*>(d->workspace);
    const long long ws = d->workspace ? d->workspace_bytes : 0;
    switch (d->tile) {
      case 41: return launch_buf_tail<128, 128, 2, 2>(p, stream, ws);
      case 42: return launch_buf_tail<128, 64, 2, 2>(p, stream, ws);
      case 43: return launch_buf_tail<64, 64, 2, 2>(p, stream, ws);
      case 44: return launch_buf_tail<128, 32, 4, 1>(p, stream, ws);
      case 47: return launch_buf_tail<64, 64, 2, 2, 1, 1>(p, stream, ws);  // deep LDS ring (small batches)
      default: return launch_buf_tail<256, 128, 4, 2, 4>(p, stream, ws);
    }
  }

  // split-K only when the caller supplied a workspace that can hold the slabs
  const long long slab = (long long)p.M * p.cout * (long long)sizeof(float);
  int max_split = 1;
  if (d->workspace && d->workspace_bytes >= 2 * slab) {
    const long long fit = d->workspace_bytes / slab;
    max_split = fit < kMaxSplit ? (int)fit : kMaxSplit;
  }
  ConvPlan plan = plan_conv(p, d->tile, max_split);
  if (d->split_k > 0) {
    ME_REQUIRE(d->split_k == 1 || (d->workspace && d->workspace_bytes >= d->split_k * slab), ME_E_BADARG,
               "me_conv2d_f32: split_k=%d needs a workspace of %lld bytes", d->split_k, d->split_k * slab);
    plan.splitk = d->split_k;
  }
  p.splitk = plan.splitk;
  p.partial = reinterpret_cast<float*>(d->workspace);
  if (plan.tile >= 100) return me32::launch_p8_tile(p, plan.tile, stream);
  if (plan.tile == 50 || plan.tile == 60) {  // weight-stationary streaming kernels (conv_ws_f32.hip): whole K per wave, no split
    ME_REQUIRE(d->split_k <= 1, ME_E_BADARG, "me_conv2d_f32: tile %d has no split-K form", plan.tile);
    return plan.tile == 50 ? me32::launch_ws1x1_f32(p, stream) : me32::launch_ws3x3_f32(p, stream);
  }
  switch (plan.tile) {
    // production tiles: buffer-addressed LDS-DMA pipeline (falls back to conv_igemm_dma_f32 when cin % 16 != 0
    // or the offsets do not fit the descriptor window)
    case 1: return launch_buf<128, 128, 2, 2>(p, stream);
    case 2: return launch_buf<128, 64, 2, 2>(p, stream);
    case 3: return launch_buf<64, 64, 2, 2>(p, stream);
    case 4: return launch_buf<128, 32, 4, 1>(p, stream);
    case 5: return launch_buf<256, 128, 4, 2, 4>(p, stream);  // 8 waves (512 threads), 4 waves / SIMD
    case 7: return launch_buf<64, 64, 2, 2, 1, 0, 1>(p, stream);  // 64x64 with the deep LDS ring (6 / 9 stages): small batches
    // tuning / ablation variants (forced ids only, tools/conv_bench.py)
    case 21: return launch_dma<128, 128, 2, 2, 1>(p, stream);  // previous generation: global_load_lds + per-lane pointers
    case 22: return launch_dma<128, 64, 2, 2, 1>(p, stream);
    case 23: return launch_dma<64, 64, 2, 2, 1>(p, stream);
    case 24: return launch_dma<128, 32, 4, 1, 1>(p, stream);
    case 25: return launch_dma<256, 128, 4, 2, 1, 4>(p, stream);
    case 6: return launch_dma<256, 128, 4, 2, 1, 1>(p, stream);      // 8 waves, registers unconstrained
    case 81: return launch_buf<128, 128, 2, 2, 1, 1>(p, stream);  // ablation: no memory traffic (all lanes out of range)
    case 82: return launch_buf<128, 64, 2, 2, 1, 1>(p, stream);
    case 83: return launch_buf<64, 64, 2, 2, 1, 1>(p, stream);
    case 85: return launch_buf<256, 128, 4, 2, 4, 1>(p, stream);
    case 61: return launch_dma<128, 128, 2, 2, 1, 1, 1>(p, stream);  // ablation: DMA sources = zero block
    case 65: return launch_dma<256, 128, 4, 2, 1, 4, 1>(p, stream);
    case 51: return launch_igemm<128, 128, 16, 2, 2>(p, stream);  // register-staged, double-buffered LDS
    case 52: return launch_igemm<128, 64, 16, 2, 2>(p, stream);
    case 53: return launch_igemm<64, 64, 16, 2, 2>(p, stream);
    case 54: return launch_igemm<128, 32, 16, 4, 1>(p, stream);
    case 55: return launch_igemm<128, 128, 32, 2, 2>(p, stream);
    case 31: return launch_igemm<128, 128, 16, 2, 2, 0, 1>(p, stream);  // register-staged + s_setprio
    case 11: return launch_igemm<128, 128, 16, 2, 2, 1>(p, stream);     // ablations (wrong results!)
    case 12: return launch_igemm<128, 128, 16, 2, 2, 2>(p, stream);
    case 13: return launch_igemm<128, 128, 16, 2, 2, 3>(p, stream);
    default: ME_REQUIRE(false, ME_E_BADARG, "me_conv2d_f32: unknown tile id %d", plan.tile);
  }
  return 0;
}

}  // extern "C"

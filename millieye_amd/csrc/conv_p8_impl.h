// conv_p8_impl.h - the patch-resident big-tile 3x3 / stride-1 convolution kernel (gfx950), generic over the storage type:
// conv_p8_h16.hip instantiates it for bfloat16 / IEEE half (v_mfma_f32_32x32x16), conv_p8_f32.hip for float32
// (v_mfma_f32_32x32x2_f32, exact fp32 FMA chains).  The design notes are in conv_p8_h16.hip's header.
// A "chunk" is 64 bytes of the channel dimension: 32 channels of a 16-bit type, 16 floats.
#pragma once
#include <type_traits>
#include <utility>
#include "common.h"
#include "dma.h"

namespace me_p8 {
using namespace me_dma;
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <class F, int... J>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, J...>) {
  (f(std::integral_constant<int, J>{}), ...);
}

// layout-neutral description of the convolution (filled by the 16-bit and the fp32 launchers)
struct P8Conv {
  const void* x;
  const void* wgt_tiled;  // [taps][cin / CHUNK][cout][CHUNK]: every (tap, chunk) slab of cout rows x 64 bytes contiguous
  const float* scale;
  const float* shift;
  const void* res;
  void* y;
  long long x_pitch, res_pitch, y_pitch;  // elements
  int n, h, w, cin, cout, act;
  int tiles_m, tiles_n;
  void* partial;  // K split: fp32 slabs [tile][split][BM][BN] of raw accumulators; instrumented builds: time stamps
  int splitk;     // workgroups per tile (each takes `cps` 64-byte chunks of the channel dimension), 1 = whole tiles
  int cps;
  int nmap;        // 1: every XCD owns ONE column of tiles (tile_n = xcd % tiles_n; tiles_n divides 8) and every tiles_n-th..
                   // row of them: its slice of the weights (1 / tiles_n of them) stays in its own 4 MB L2 however far the
                   // workgroups drift apart along K; the grid is 8 x the largest per-XCD count, surplus workgroups exit
  int store_mode;  // epilogue stores: 0 plain, 1 nt (streaming), 2 sc1 (write-through: the line leaves the XCD's L2 at once
                   // instead of waiting dirty for the end-of-kernel release)
};

using me::store16;

struct P8Args {
  P8Conv c;
  int Wp, Ip, halo;       // padded-linear pitches, halo = Wp + 1
  long long Mp;           // n * Ip positions
  unsigned ip_m, ip_s;    // magic division by Ip
  unsigned wp_m, wp_s;    // magic division by Wp
  int lpa;                // patch DMA instructions per wave and chunk = ceil(ceil(rows / 16) / 8)
  int rows;               // BM + 2 * halo
  int stamp_wave;         // instrumented instances (ABL = 9): the wave of workgroup 0 whose s_memtime stamps are recorded
};

__device__ __forceinline__ unsigned udiv_magic(unsigned n, unsigned m, unsigned s) { return (__umulhi(n, m) + n) >> s; }

__device__ __forceinline__ void dma1(unsigned v, u32x4 r, unsigned s, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[r], %[s] offen lds\n\t"
               "s_mov_b32 m0, %[k]"
               : [k] "=&s"(keep)
               : [d] "s"(dst), [r] "s"(r), [s] "s"(s), [v] "v"(v)
               : "memory", "scc");
}

constexpr int kLpaMax = 7;  // patch DMA instructions per wave and chunk (8 waves: up to 896 patch rows; 4 waves: 448)

// PIPE = 1: register double-buffered fragments - the ds_reads of stage s + 1 complete behind the MFMAs of stage s (one
//           workgroup per CU, <= 256 VGPRs);  PIPE = 0: fragments are read inside their stage, MINB = 2 workgroups share
//           a CU (<= 128 VGPRs, <= 80 KB of LDS each): their phases interleave, so the epilogue's memory traffic (residual
//           reads + output writes, 88 MB at 52x52) runs beside the other workgroup's main loop instead of after it.
// ABL (tuning only, wrong results): 1 = all DMA lanes out of range, 3 = no DMA instructions, 4 = 3 + no LDS fragment reads,
//           5 = 4 + no barriers (pure MFMA stream), 6 = full main loop but no epilogue, 7 = weights out of range (patch
//           traffic only), 8 = patch out of range (weight traffic only); 9 = correct results + s_memtime stamps of
//           workgroup 0 / wave 0 into the workspace (per stage: before the wait, after the wait, after the barrier, at the end);
//           10 = correct results + four s_memrealtime stamps (100 MHz, chip-wide clock) of EVERY workgroup's wave 0 (start,
//           main loop start, main loop end, end) and its HW_ID / XCC_ID: dispatch rounds, co-residency, epilogue overlap
// SK: the instance with the K-split slab epilogue (p.splitk > 1).  It is a separate instance on purpose: merely containing
// that path cost the whole-tile kernel 20 % (106 scalar registers + 24 spilled against 56, and a slower main loop: 58.9 vs
// 46.6 us on the 52 x 52 128 -> 256 layer, measured with the stamp instance that never had it - profiles/r03_kernel_evolution.md)
// TG = 3 (PIPE = 1 only, round 3): the weight slabs of a 64-byte channel chunk sit in LDS as three GROUPS of three taps (72 KB for
// BN = 128) and the workgroup synchronises once per group - 6 * MT * NT MFMAs per wave between barriers instead of 2 * MT * NT.
// With a barrier and a DMA wait per tap a stage of the one-workgroup-per-CU tiles took ~1270 cycles for 768 of MFMA issue
// (tools/p8_wgmap.py); the per-stage overhead is now paid once per three taps.  Group g of the next chunk is fetched into slot
// g right behind the barrier that ends the current chunk's group g.
// DS = 1 (round 4): DMA duty split.  A wave's loads return IN ORDER (vmcnt retires oldest first), so when every wave issues
// both streams, a weight slab requested behind the next chunk's patch (an HBM / MALL miss) cannot land before that patch
// does.  With DS the lower half of the waves issues ONLY weight slabs (L2 hits, needed soon), the upper half ONLY the patch
// (needed at the next chunk).  Measured: neutral to +2 % (profiles/r04_kernel_evolution.md section 1 - the waits it removes
// were not what bounds the kernel); kept as autotuner candidates (tile ids 621 / 721 / 731).
// DS = 4 (PIPE = 2): four LOADER waves beside the eight consumers (12 waves, <= 168 VGPRs): waves 8-11 issue every LDS-DMA
// of the workgroup (weights: BN / 64 pieces per stage each; patch: seven pieces per chunk each, spread over taps 0-5,
// pieces behind the patch's end are out-of-range no-ops so that the vmcnt bookkeeping is static) and the consumers' load
// segment is fragment reads only - the ablation without DMA instructions ran the ping-pong loop at 84 % of the matrix pipe.
// RB (PIPE = 2 only): slots of the weight ring; a slab is requested RB - 1 stages before its stage (deeper rings: slower).
template <int WR, int WC, int MT, int NT, int PIPE, int MINB, class DT, int ABL = 0, bool SK = false, int TG = 1, int DS = 0, int RB = 3>
__global__ __launch_bounds__(64 * (WR * WC + (DS == 4 ? 4 : 0)))
    __attribute__((amdgpu_waves_per_eu((WR * WC * MINB + (DS == 4 ? 4 : 0)) / 4, (WR * WC * MINB + (DS == 4 ? 4 : 0)) / 4))) void
    conv3x3_p8_kernel(P8Args a) {
  using frag = typename DT::frag;
  const P8Conv& p = a.c;
  constexpr int NW = WR * WC;
  static_assert(NW == 8 || NW == 4, "8 waves, or 4 (two independent 4-wave workgroups per CU: their barriers interleave)");
  constexpr int TM = 32 * MT, TN = 32 * NT, BM = TM * WR, BN = TN * WC;
  constexpr int NWB = DS ? NW / 2 : NW;  // waves that issue weight slabs (the first NWB)
  constexpr int NWA = DS ? NW / 2 : NW;  // waves that issue the patch (the last NWA)
  constexpr int LPB = BN / 16 / NWB;
  static_assert(LPB >= 1 && BN % 128 == 0, "BN must be a multiple of 128");
  static_assert(TG == 1 || (TG == 3 && PIPE == 1), "tap groups need the register-pipelined variant");
  static_assert(PIPE != 2 || (NW == 8 && MINB == 1 && TG == 1 && (DS == 0 || DS == 2 || DS == 4)), "ping-pong: one 8-wave workgroup per CU");
  static_assert((DS != 2 && DS != 4) || PIPE == 2, "the spread duty splits belong to the ping-pong variant");
  constexpr unsigned B_TAP = BN * 64u;            // one tap's weight slab of a chunk
  constexpr unsigned B_SLOT = TG * B_TAP;         // a ring slot: one tap, or a group of three
  static_assert(RB == 3 || (PIPE == 2 && RB >= 3 && RB <= 9), "deeper rings: ping-pong variant only");
  constexpr unsigned A_BASE = (unsigned)RB * B_SLOT;
  constexpr int NSET = PIPE == 1 ? 2 : 1;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const int r32 = lane & 31, hh = lane >> 5;

  int tile_m, tile_n, piece, sid;
  {
    const int nwg = p.tiles_m * p.tiles_n * p.splitk;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    piece = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int wg = SK ? piece / p.splitk : piece;
    sid = SK ? piece - wg * p.splitk : 0;  // K split: this workgroup multiplies chunks [sid * cps, (sid + 1) * cps) of the tile
    tile_n = wg % p.tiles_n;
    tile_m = wg / p.tiles_n;
    if (!SK && p.nmap) {
      const int G = 8 / p.tiles_n;          // XCDs that share a column
      tile_n = xcd % p.tiles_n;
      tile_m = idx * G + xcd / p.tiles_n;
      piece = tile_m * p.tiles_n + tile_n;
      if (tile_m >= p.tiles_m) return;
    }
  }
  const int W = p.w, H = p.h;
  const long long q0 = (long long)tile_m * BM;
  const int n0 = tile_n * BN;
  const int lpa = a.lpa;
  const unsigned A_SLOT = (unsigned)lpa * (NWA * 1024u);
  // wave-uniform roles.  DS = 1: waves 0-3 weights, 4-7 patch.  DS = 2 (ping-pong): waves 0, 1 of each half fetch weights,
  // waves 2, 3 of each half the patch - and they spread it over the taps (below)
  const bool is_loader = DS == 4 && wave >= NW;   // DS = 4: waves 8-11 only load, waves 0-7 only multiply
  const bool is_b = DS == 4 ? is_loader : DS == 2 ? ((wave >> 1) & 1) == 0 : (!DS || wave < NWB);
  const bool is_a = DS == 4 ? is_loader : DS == 2 ? ((wave >> 1) & 1) == 1 : (!DS || wave >= NW - NWA);
  const int wa = DS == 4 ? wave - NW : DS == 2 ? (wave & 1) + 2 * (wave >> 2) : wave - (NW - NWA);   // patch-wave index (meaningful when is_a)
  const int wbi = DS == 4 ? wave - NW : DS == 2 ? (wave & 1) + 2 * (wave >> 2) : wave;               // weight-wave index (meaningful when is_b)

  // first image the patch can touch: descriptor base, so that per-lane offsets stay small and non-negative
  const long long pq0 = q0 - a.halo;
  const int nb = pq0 > 0 ? (int)udiv_magic((unsigned)pq0, a.ip_m, a.ip_s) : 0;
  const u32x4 rsrc_a = make_rsrc(reinterpret_cast<const unsigned char*>(p.x) + (long long)nb * H * W * p.x_pitch * DT::kBytes);
  const u32x4 rsrc_b = make_rsrc(p.wgt_tiled);  // [tap][chunk][cout][32]: a (tap, chunk) slab of BN rows is contiguous

  unsigned v_a[kLpaMax], v_b[LPB];
  auto setup_a = [&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int r = (wa + NWA * j) * 16 + (lane >> 2);
    const int qd = (lane & 3) ^ ((r >> 2) & 3);
    const long long pq = pq0 + r;
    unsigned off = kOobOffset;
    if (ABL != 1 && ABL != 8 && is_a && j < lpa && r < a.rows && pq >= 0 && pq < a.Mp) {
      const unsigned u = (unsigned)pq;
      const unsigned n = udiv_magic(u, a.ip_m, a.ip_s);
      const unsigned rem = u - n * (unsigned)a.Ip;
      const unsigned y = udiv_magic(rem, a.wp_m, a.wp_s);
      const unsigned x = rem - y * (unsigned)a.Wp;
      if (y < (unsigned)H && x < (unsigned)W)
        off = (unsigned)((((long long)(n - nb) * H + y) * W + x) * p.x_pitch * DT::kBytes) + 16u * qd;
    }
    v_a[j] = off;
  };
  static_for(setup_a, std::make_integer_sequence<int, kLpaMax>{});
  auto setup_b = [&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int row = (wbi + NWB * j) * 16 + (lane >> 2);
    const int qd = (lane & 3) ^ ((row >> 2) & 3);
    v_b[j] = (ABL != 1 && ABL != 7 && is_b) ? (unsigned)(n0 + row) * 64u + 16u * qd : kOobOffset;  // cout % BN == 0: always in range
  };
  static_for(setup_b, std::make_integer_sequence<int, LPB>{});

  // tap -> patch row shift: output row tr reads patch row tr + (1 + dy) * Wp + (1 + dx)
  int tapoff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) tapoff[t] = __builtin_amdgcn_readfirstlane((t / 3) * a.Wp + (t % 3));

  unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.partial);
  int dbg_n = 0;
  auto stamp = [&]() {
    if constexpr (ABL == 9) {
      if (blockIdx.x == 0 && wave == a.stamp_wave && dbg && dbg_n < 2040) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0) dbg[dbg_n] = t;
        ++dbg_n;
      }
    }
  };
  stamp();
  auto wg_stamp = [&](int k) {
    if constexpr (ABL == 10) {
      if (wave == 0 && dbg) {
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) dbg[(long long)blockIdx.x * 6 + k] = t;
      }
    }
  };
  wg_stamp(0);
  if constexpr (ABL == 10) {
    if (wave == 0 && lane == 0 && dbg) {
      dbg[(long long)blockIdx.x * 6 + 4] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
      dbg[(long long)blockIdx.x * 6 + 5] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
    }
  }

  unsigned rowbase[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) rowbase[i] = (unsigned)(wr * TM + i * 32 + r32);
  const int swb = (r32 >> 2) & 3;
  const unsigned char* b_frag = smem16 + (wc * TN + r32) * 64;
  const unsigned b_offk[2] = {(unsigned)(((0 + hh) ^ swb) * 16), (unsigned)(((2 + hh) ^ swb) * 16)};

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int cs_all = p.cin / DT::kChunk;  // 64-byte chunks of the channel dimension
  const int cbase = SK ? __builtin_amdgcn_readfirstlane(sid * p.cps) : 0;
  const int cs = SK ? (cbase + p.cps < cs_all ? p.cps : cs_all - cbase) : cs_all;  // ... this workgroup walks
  const unsigned wave_lds = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)wbi * 1024u);
  const unsigned wave_lds_a = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(wa < 0 ? 0 : wa) * 1024u);
  auto issue_a = [&](int chunk, unsigned slot) {
    if (ABL >= 3 && ABL <= 5) return;
    if (DS && !is_a) return;
    const unsigned dst = wave_lds_a + A_BASE + slot * A_SLOT;
    static_for(
        [&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if (j < lpa) dma1(v_a[j], rsrc_a, (unsigned)(cbase + chunk) * 64u, dst + (unsigned)j * (NWA * 1024u));
        },
        std::make_integer_sequence<int, kLpaMax>{});
  };
  auto issue_a_piece = [&](int chunk, unsigned slot, auto jc) {   // one patch piece of this wave (DS = 2: spread over the taps)
    constexpr int j = decltype(jc)::value;
    if (ABL >= 3 && ABL <= 5) return;
    if (!is_a || j >= lpa) return;
    dma1(v_a[j], rsrc_a, (unsigned)(cbase + chunk) * 64u, wave_lds_a + A_BASE + slot * A_SLOT + (unsigned)j * (NWA * 1024u));
  };
  auto issue_b = [&](int chunk, int tap, unsigned ring) {
    if (ABL >= 3 && ABL <= 5) return;
    if (DS && !is_b) return;
    const unsigned soff = ((unsigned)tap * (unsigned)cs_all + (unsigned)(cbase + chunk)) * (unsigned)p.cout * 64u;
#pragma unroll
    for (int j = 0; j < LPB; ++j) dma1(v_b[j], rsrc_b, soff, wave_lds + ring * B_SLOT + (unsigned)j * (NWB * 1024u));
  };
  auto issue_bg = [&](int chunk, int grp) {   // TG = 3: the three taps of group grp into slot grp
    if (ABL >= 3 && ABL <= 5) return;
    if (DS && !is_b) return;
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) {
      const unsigned soff = ((unsigned)(3 * grp + tt) * (unsigned)cs_all + (unsigned)(cbase + chunk)) * (unsigned)p.cout * 64u;
#pragma unroll
      for (int j = 0; j < LPB; ++j)
        dma1(v_b[j], rsrc_b, soff, wave_lds + (unsigned)grp * B_SLOT + (unsigned)tt * B_TAP + (unsigned)j * (NWB * 1024u));
    }
  };
  // DS: the two roles wait for different things.  Weight waves count weight pieces only; patch waves wait for the whole
  // patch (vmcnt(0)) at the stage whose barrier opens the next chunk, and for nothing but their LDS reads elsewhere.
  auto ds_wait = [&](auto nc, bool patch_due) {   // nc = weight pieces that may stay in flight (compile-time)
    constexpr int N = decltype(nc)::value;
    if (is_b) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
    else if (patch_due) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  using V0 = std::integral_constant<int, 0>;
  using V1 = std::integral_constant<int, LPB>;
  using V2 = std::integral_constant<int, 2 * LPB>;
  using V3 = std::integral_constant<int, 3 * LPB>;
  using V6 = std::integral_constant<int, 6 * LPB>;
  // vmcnt(3 * LPB + extra) / vmcnt(extra) with a runtime (uniform) extra = patch pieces that may stay in flight
  auto wait_g = [&](bool with_group, int extra) {
    if (with_group) {
      switch (extra) {
        case 0: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * LPB) : "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * LPB + 1) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * LPB + 2) : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * LPB + 3) : "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * LPB + 4) : "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * LPB + 5) : "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * LPB + 6) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * LPB + 7) : "memory"); break;
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
  };
  // s_waitcnt takes an immediate: the runtime patch count goes through a uniform switch
  auto wait_b_plus_a = [&]() {
    switch (lpa) {
      case 1: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPB + 1) : "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPB + 2) : "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPB + 3) : "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPB + 4) : "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPB + 5) : "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPB + 6) : "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPB + 7) : "memory"); break;
    }
  };
  constexpr bool NODMA = ABL >= 3 && ABL <= 5;  // (the stamp instances 9 / 10 and the out-of-range ablations 7 / 8 do issue their DMAs:
  // until round 3 this read `ABL != 6`, so the round-2 stamp instance ran without loads - its 1033 cycles / stage were a
  // no-DMA number, see profiles/r03_kernel_evolution.md)

  frag afr[NSET][2][MT], bfr[NSET][2][NT];
  if (ABL == 4 || ABL == 5) {  // ablation: the fragments never change - give them ordinary values
#pragma unroll
    for (int s2 = 0; s2 < NSET; ++s2)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          afr[s2][ks][i] = DT::fill(0.01f * (float)(lane + i));
#pragma unroll
        for (int j = 0; j < NT; ++j)
          bfr[s2][ks][j] = DT::fill(0.02f * (float)(lane - j));
      }
  }
  auto load_frags = [&](auto setc, auto tc, int chunk, int bslot = -1) {   // bslot >= 0: runtime ring slot (PIPE = 2)
    constexpr int SET = decltype(setc)::value, T = decltype(tc)::value;
    if (ABL == 4 || ABL == 5) {
#pragma unroll
      for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(afr[SET][0][i]), "+v"(afr[SET][1][i]));
      return;
    }
    const unsigned char* Ab = smem16 + A_BASE + (unsigned)(chunk & 1) * A_SLOT;
    const unsigned char* Bb = b_frag + (bslot >= 0 ? (unsigned)bslot * B_SLOT : (TG == 3 ? T * B_TAP : (T % 3) * B_SLOT));
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      unsigned rb0 = rowbase[i];
      asm volatile("" : "+v"(rb0));  // opaque per stage: otherwise the 18 * MT fragment addresses are hoisted out of the
                                     // chunk loop and live in registers for the whole kernel; 5 VALU per block instead
      const unsigned R = rb0 + (unsigned)tapoff[T];
      const unsigned o0 = (R << 6) + (((R >> 2) ^ (unsigned)hh) & 3u) * 16u;
      afr[SET][0][i] = *reinterpret_cast<const frag*>(Ab + o0);
      afr[SET][1][i] = *reinterpret_cast<const frag*>(Ab + (o0 ^ 32u));
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < NT; ++j) bfr[SET][ks][j] = *reinterpret_cast<const frag*>(Bb + j * 32 * 64 + b_offk[ks]);
  };
  auto mfmas = [&](auto setc) {
    constexpr int SET = decltype(setc)::value;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) DT::mfma(afr[SET][ks][i], bfr[SET][ks][j], acc[i][j]);
  };
  using Z = std::integral_constant<int, 0>;

  issue_a(0, 0);
  if constexpr (TG == 3) {
    issue_bg(0, 0);
    issue_bg(0, 1);
    issue_bg(0, 2);
    if (NODMA) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else if (DS) ds_wait(V6{}, true);
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(6 * LPB) : "memory");   // the patch and group 0 have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);
  } else if constexpr (PIPE == 2) {
    static_for([&](auto sc) { issue_b(0, decltype(sc)::value, (unsigned)decltype(sc)::value); }, std::make_integer_sequence<int, RB - 1>{});
  } else {
  issue_b(0, 0, 0);
  issue_b(0, 1, 1);
  }
  if constexpr (PIPE == 1 && TG == 1) {
    // ---- software pipeline: iteration s (after its barrier): DMA(s + 3) -> ring slot of stage s; LDS -> registers for
    // stage s + 1 (fragment set P ^ 1); MFMAs of stage s from set P.  9 stages per chunk: the parity flips per chunk. ----
    issue_b(0, 2, 2);
    if (NODMA) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else if (DS) ds_wait(V2{}, true);
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * LPB) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_frags(Z{}, Z{}, 0);
  }

  wg_stamp(1);
  auto stage = [&](auto tc, auto parc, int chunk) {
    constexpr int T = decltype(tc)::value, PAR = decltype(parc)::value;
    constexpr int SET = PIPE ? ((T + PAR) & 1) : 0;
    const bool more_chunks = chunk + 1 < cs;
    if constexpr (PIPE) {
      const bool has_next = T < 8 || more_chunks;  // stage s + 1 exists
      stamp();
      if constexpr (TG == 3) {
        // a barrier only where the NEXT tap opens a new group (T = 2, 5, 8): that group's weights must have landed, and every
        // wave is done reading this group (the fragments of its last tap were fetched during the tap before).  Loads younger
        // than the next group's weights:  T = 2: group 2 of this chunk + the next patch;  T = 5: the next patch + the next
        // chunk's group 0;  T = 8: the next chunk's group 1.
        if constexpr (T % 3 == 2) {
          if (has_next) {
            if (NODMA) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else if (DS) {
              // weight waves: T = 2: group 2 may fly; T = 5: the next chunk's group 0 (if any); T = 8: its group 1.
              // patch waves: the next chunk's patch must be in LDS behind the barrier of T = 8
              if (T == 5 && !more_chunks) ds_wait(V0{}, false);
              else ds_wait(V3{}, T == 8);
            }
            else if (T == 2) wait_g(true, more_chunks ? lpa : 0);
            else if (T == 5) wait_g(more_chunks, more_chunks ? lpa : 0);
            else wait_g(true, 0);
            stamp();
            if (ABL != 5) {
              __builtin_amdgcn_s_barrier();
              asm volatile("" ::: "memory");
            }
          }
        }
      } else {
      // stage s + 1's weights must have landed; younger loads: stage s + 2's weights, and - taps 1, 2 - the next patch
      if (NODMA) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if (DS) {
        if (T >= 7 && !more_chunks) ds_wait(V0{}, false);
        else ds_wait(V1{}, T == 8);   // patch waves: the next chunk's patch is read from stage 8 on
      } else if (T >= 7 && !more_chunks) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else if ((T == 1 || T == 2) && more_chunks) {
        wait_b_plus_a();
      } else {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPB) : "memory");
      }
      stamp();
      if (has_next && ABL != 5) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      }
      stamp();
      // The matrix pipe must not idle while this wave issues its DMAs and next-stage fragment reads: the stage is laid
      // out as 2 * MT groups of NT MFMAs with one slice of that work behind each group (scheduling barriers pin the order):
      //   group g < MT : A fragments of block row g for stage s + 1 (5 VALU + 2 ds_read_b128)
      //   group MT     : weights of stage s + 3 into the ring slot of stage s (its fragments were read one iteration ago),
      //                  at tap 0 also the next chunk's patch
      //   groups > MT  : B fragments for stage s + 1, spread
      constexpr int TN1 = T < 8 ? T + 1 : 0;
      const int cn = T < 8 ? chunk : chunk + 1;
      const unsigned char* Ab = smem16 + A_BASE + (unsigned)(cn & 1) * A_SLOT;
      const unsigned char* Bb = b_frag + (TG == 3 ? TN1 * B_TAP : (TN1 % 3) * B_SLOT);
      constexpr int NS = SET ^ 1;
      static_for(
          [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int ks = g / MT, i = g % MT;
#pragma unroll
            for (int j = 0; j < NT; ++j) DT::mfma(afr[SET][ks][i], bfr[SET][ks][j], acc[i][j]);
            if constexpr (g < MT) {
              if (has_next && ABL != 4 && ABL != 5) {
                unsigned rb0 = rowbase[g];
                asm volatile("" : "+v"(rb0));
                const unsigned R = rb0 + (unsigned)tapoff[TN1];
                const unsigned o0 = (R << 6) + (((R >> 2) ^ (unsigned)hh) & 3u) * 16u;
                afr[NS][0][g] = *reinterpret_cast<const frag*>(Ab + o0);
                afr[NS][1][g] = *reinterpret_cast<const frag*>(Ab + (o0 ^ 32u));
              }
            }
            if constexpr (g == MT) {
              if constexpr (TG == 3) {
                // behind the barrier of T = 2 / 5 / 8 the slot of group T / 3 is free: the next chunk's group goes there
                if constexpr (T % 3 == 2) {
                  if (more_chunks) issue_bg(chunk + 1, T / 3);
                }
              } else {
                constexpr int T3 = (T + 3) % 9;
                const int c3 = chunk + (T + 3 >= 9 ? 1 : 0);
                if (c3 < cs) issue_b(c3, T3, (unsigned)(T3 % 3));
              }
              if (T == 0 && more_chunks) issue_a(chunk + 1, (unsigned)((chunk + 1) & 1));
            }
            if constexpr (g >= MT) {  // 2 * NT B-fragment reads over the MT groups MT .. 2 MT - 1
              if (has_next && ABL != 4 && ABL != 5) {
                constexpr int lo = (g - MT) * (2 * NT) / MT, hi = (g - MT + 1) * (2 * NT) / MT;
#pragma unroll
                for (int r = lo; r < hi; ++r)
                  bfr[NS][r / NT][r % NT] = *reinterpret_cast<const frag*>(Bb + (r % NT) * 32 * 64 + b_offk[r / NT]);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          },
          std::make_integer_sequence<int, 2 * MT>{});
      stamp();
    } else {
      // loads younger than this stage's weights: the next tap's weights, and - at taps 1 and 2 - the next chunk's patch
      // (round 3 measured issuing that patch one piece per tap instead of as a burst at tap 0: slower on every layer -
      // 57.9 vs 56.1 us at 52 x 52, 60.5 vs 58.5 at 26 x 26, 70.7 vs 63.0 at 13 x 13 - the burst has the longest lead time)
      stamp();
      if (NODMA) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if (DS) {
        if (T == 8 && !more_chunks) ds_wait(V0{}, false);
        else ds_wait(V1{}, T == 0);   // patch waves: this chunk's patch (requested at tap 0 of the previous chunk)
      } else if (T == 8 && !more_chunks) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else if ((T == 1 || T == 2) && more_chunks) {
        wait_b_plus_a();
      } else {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPB) : "memory");
      }
      stamp();
      if (ABL != 5) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      stamp();
      {  // weights of stage s + 2
        constexpr int T2 = (T + 2) % 9;
        const int c2 = chunk + (T + 2 >= 9 ? 1 : 0);
        if (c2 < cs) issue_b(c2, T2, (unsigned)(T2 % 3));
      }
      if (T == 0 && more_chunks) issue_a(chunk + 1, (unsigned)((chunk + 1) & 1));
      load_frags(Z{}, tc, chunk);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(Z{});
      stamp();
    }
  };
  // ---- PIPE = 2 (round 4): ping-pong.  The counters of the lockstep variants (profiles/r04_p8_*_sq.txt) show waves stalled on
  // the matrix pipe 37-44 % of their life while that pipe is busy 36-42 % of the time: the two waves of a SIMD want it in the
  // same phase and leave it idle together (DMA issue, fragment reads, waits) in the other.  Here the workgroup's two halves
  // (waves 0-3 / 4-7: one wave per SIMD each) run HALF A STAGE APART: a stage is a load segment L (weight DMA of stage s + 2,
  // this stage's fragment reads) and a compute segment C (its MFMAs, back to back at raised priority) with a barrier behind
  // each; the second half enters one barrier late, so that its L(s) runs beside the first half's C(s) and its C(s) beside
  // the first half's L(s + 1).  One fragment set; time slot 2s: G0 L(s); slot 2s + 1: G0 C(s) | G1 L(s); slot 2s + 2: G1 C(s).
  //   RAW: a wave waits for ITS pieces of slab s + 1 before the barrier that ends slot 2s + 1 (G0 behind C(s), G1 behind L(s)).
  //   WAR: slab s + 2 goes to the ring slot of slab s - 1, last read by G1 in slot 2s - 1; G0 issues it in slot 2s.
  if constexpr (PIPE == 2) {
    constexpr int D = RB - 1;   // prefetch distance in stages
    const int grp = is_loader ? 0 : __builtin_amdgcn_readfirstlane(wave >> 2);   // (loaders keep the first half's barrier count)
    auto wait_vm = [&](auto nc, bool plus_patch) {   // vmcnt(N [+ lpa]); s_waitcnt takes an immediate
      constexpr int N = decltype(nc)::value;
      if (!plus_patch) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); return; }
      switch (lpa) {
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N + 1) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N + 2) : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N + 3) : "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N + 4) : "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N + 5) : "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N + 6) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N + 7) : "memory"); break;
      }
    };
    if (NODMA) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else if (DS == 2 && is_a) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if (DS == 4 && !is_loader) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((D - 1) * LPB) : "memory");   // patch 0 and slab 0 have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (grp == 1) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if constexpr (DS == 4) {
      if (is_loader) {
        // The loader waves' own loop (kept apart from the consumers' unrolled stages: sharing the stage body made the register
        // allocator spill the accumulators).  Slot 2s: this stage's requests (slab s + 2, the next chunk's patch in pieces);
        // end of slot 2s + 1: my pieces of slab s + 1 have landed.  Younger than that slab: the patch pieces of stage s - 1
        // (issued behind it), slab s + 2, the patch pieces of this stage - static counts (a.lpa == 7: pieces past the
        // patch's end are no-ops that still count).
        int lrs = 0;
        auto loader_stage = [&](auto tc, int chunk) {
          constexpr int T = decltype(tc)::value;
          const bool more_chunks = chunk + 1 < cs;
          constexpr int T2 = (T + 2) % 9;
          const int c2 = chunk + (T + 2 >= 9 ? 1 : 0);
          if (c2 < cs) issue_b(c2, T2, (unsigned)(lrs == 0 ? RB - 1 : lrs - 1));
          if (more_chunks) {
            const unsigned sl = (unsigned)((chunk + 1) & 1);
            if constexpr (T == 0) {
              issue_a_piece(chunk + 1, sl, std::integral_constant<int, 0>{});
              issue_a_piece(chunk + 1, sl, std::integral_constant<int, 1>{});
            } else if constexpr (T <= 5) {
              issue_a_piece(chunk + 1, sl, std::integral_constant<int, T + 1>{});
            }
          }
          if (ABL != 5) __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          constexpr int NPP_PREV = T == 0 ? 0 : T == 1 ? 2 : T <= 6 ? 1 : 0;   // patch pieces issued at stage T - 1 / T
          constexpr int NPP_THIS = T == 0 ? 2 : T <= 5 ? 1 : 0;
          if (!NODMA) {
            if (more_chunks) {
              asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPP_PREV + LPB + NPP_THIS) : "memory");
            } else if constexpr (T < 8) {
              asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T <= 6 ? LPB : 0) : "memory");
            }
          }
          if (ABL != 5) __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          lrs = lrs + 1 == RB ? 0 : lrs + 1;
        };
        for (int chunk = 0; chunk < cs; ++chunk)
          static_for([&](auto tc) { loader_stage(tc, chunk); }, std::make_integer_sequence<int, 9>{});
        if (ABL != 5) __builtin_amdgcn_s_barrier();          // the first half's closing barrier
        if (DT::kBytes == 2 && !SK && ABL != 6) __syncthreads();   // the barrier the epilogue opens with, then out
        return;
      }
    }
    int rs = 0;   // ring slot of the current stage; the slab D stages ahead goes to the slot of the stage before this one
    auto stage_pp = [&](auto tc, int chunk) {
      constexpr int T = decltype(tc)::value;
      const bool more_chunks = chunk + 1 < cs;
      // my pieces of the next stage's slab must have landed.  Loads retire in order: younger than that slab are the D - 1
      // slabs behind it (fewer at the very end) and - when it was requested before this chunk's tap-0 patch burst, i.e. for
      // T < D - the next chunk's patch.  Entering a chunk (T = 8) its patch is older than the slab, so it has landed too.
      auto wait_next = [&]() {
        if (NODMA || DS == 4) return;   // (DS = 4: the consumers have nothing in flight; the loaders wait in their own branch)
        if constexpr (DS == 2) {
          // weight waves never queue behind a patch piece; patch waves: everything of the next chunk by the end of tap 8
          if (is_a) {
            if (T == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
          }
          if (!more_chunks) {
            if constexpr (T < 8) wait_vm(std::integral_constant<int, (D - 1 < 7 - T ? D - 1 : 7 - T) * LPB>{}, false);
          } else {
            wait_vm(std::integral_constant<int, (D - 1) * LPB>{}, false);
          }
          return;
        }
        if (!more_chunks) {
          if constexpr (T < 8) wait_vm(std::integral_constant<int, (D - 1 < 7 - T ? D - 1 : 7 - T) * LPB>{}, false);
        } else {
          wait_vm(std::integral_constant<int, (D - 1) * LPB>{}, T < D);
        }
      };
      stamp();
      if constexpr (DS == 4) {
        // (the loaders run their own loop in front of this one; the consumers request nothing)
      } else
      {
        constexpr int T2 = (T + D) % 9;
        const int c2 = chunk + (T + D >= 9 ? 1 : 0);
        if (c2 < cs) issue_b(c2, T2, (unsigned)(rs == 0 ? RB - 1 : rs - 1));
      }
      if constexpr (DS == 4) {
      } else if constexpr (DS == 2) {
        // the next chunk's patch leaves in small steps (two pieces at tap 0, one per tap after): a burst of up to 28 KB of
        // misses in front of the weight slabs was what made the two streams cost more together than the sum of each alone
        if (more_chunks) {
          const unsigned sl = (unsigned)((chunk + 1) & 1);
          if constexpr (T == 0) {
            issue_a_piece(chunk + 1, sl, std::integral_constant<int, 0>{});
            issue_a_piece(chunk + 1, sl, std::integral_constant<int, 1>{});
          } else if constexpr (T <= 5) {
            issue_a_piece(chunk + 1, sl, std::integral_constant<int, T + 1>{});
          }
        }
      } else {
        if (T == 0 && more_chunks) issue_a(chunk + 1, (unsigned)((chunk + 1) & 1));
      }
      load_frags(Z{}, tc, chunk, rs);
      if (grp == 1) wait_next();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      stamp();
      if (ABL != 5) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      stamp();
      __builtin_amdgcn_s_setprio(1);
      mfmas(Z{});
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (grp == 0) wait_next();
      stamp();
      if (ABL != 5) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      rs = rs + 1 == RB ? 0 : rs + 1;
    };
    for (int chunk = 0; chunk < cs; ++chunk)
      static_for([&](auto tc) { stage_pp(tc, chunk); }, std::make_integer_sequence<int, 9>{});
    if (grp == 0 && ABL != 5) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  } else
  {
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int chunk = 0;
    for (; chunk + 2 <= cs; chunk += 2) {  // two chunks = 18 stages per trip: the fragment-set parity is static
      static_for([&](auto tc) { stage(tc, P0{}, chunk); }, std::make_integer_sequence<int, 9>{});
      static_for([&](auto tc) { stage(tc, P1{}, chunk + 1); }, std::make_integer_sequence<int, 9>{});
    }
    if (chunk < cs) static_for([&](auto tc) { stage(tc, P0{}, chunk); }, std::make_integer_sequence<int, 9>{});
  }
  stamp();
  wg_stamp(2);
  if (ABL == 6) {  // ablation: no epilogue (one store so that the loop is not dead code)
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) t += acc[i][j][e];
    if (t == 123.456f) reinterpret_cast<float*>(p.y)[0] = t;
    return;
  }

  if constexpr (ABL == 0 && SK) {  // K split: raw accumulators to this piece's slab; the launcher's reduce pass finishes the tile
    float* slab = reinterpret_cast<float*>(p.partial) + (long long)piece * (BM * BN);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          slab[(wr * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh) * BN + wc * TN + j * 32 + r32] = acc[i][j][e];
    return;
  }

  if constexpr (DT::kBytes == 2) {
    // ---- epilogue: per wave, 32x32 blocks through a private LDS transpose (lane = output channel -> 8 consecutive
    // channels of one pixel per lane: 16-byte residual loads and stores).  Every residual piece of the wave is requested
    // BEFORE the arithmetic starts (one memory latency per wave, not one per block), and there is no s_waitcnt between the
    // transpose's writes and reads: the LDS executes one wave's operations in order.  Rows are decoded from the
    // padded-linear position to the dense NHWC pixel; pad positions and positions behind the last image are skipped. ----
    const float slope = p.act == ME_ACT_LEAKY ? 0.1f : 1.0f;
    constexpr int TP = 36;
    __syncthreads();
    float* tbuf = reinterpret_cast<float*>(smem16) + wave * (2 * 32 * TP);
    const int prow = lane >> 2, c8 = (lane & 3) * 8;
    unsigned short* __restrict__ yb = reinterpret_cast<unsigned short*>(p.y);
    const unsigned short* __restrict__ rb = reinterpret_cast<const unsigned short*>(p.res);
    int mrow[MT][2];  // dense pixel index (< 2^31: checked by fill16), -1 = pad / out of range
  #pragma unroll
    for (int i = 0; i < MT; ++i)
  #pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const long long q = q0 + wr * TM + i * 32 + pass * 16 + prow;
        int m = -1;
        if (q < a.Mp) {
          const unsigned u = (unsigned)q;
          const unsigned n = udiv_magic(u, a.ip_m, a.ip_s);
          const unsigned rem = u - n * (unsigned)a.Ip;
          const unsigned y = udiv_magic(rem, a.wp_m, a.wp_s);
          const unsigned x = rem - y * (unsigned)a.Wp;
          if (y < (unsigned)H && x < (unsigned)W) m = (int)((n * (unsigned)H + y) * (unsigned)W + x);
        }
        mrow[i][pass] = m;
      }
    constexpr int RD = MT <= 2 ? MT : 2;  // residual prefetch depth in block rows (all of them for the small wave tiles)
    uint4 rres[RD][NT][2];
    auto fetch_res = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if (!rb) return;
  #pragma unroll
      for (int j = 0; j < NT; ++j)
  #pragma unroll
        for (int pass = 0; pass < 2; ++pass)
          rres[i % RD][j][pass] = mrow[i][pass] >= 0
                                      ? *reinterpret_cast<const uint4*>(rb + (long long)mrow[i][pass] * p.res_pitch + n0 + wc * TN + j * 32 + c8)
                                      : make_uint4(0u, 0u, 0u, 0u);
    };
    static_for(fetch_res, std::make_integer_sequence<int, RD>{});
    auto block_out = [&](auto ic, auto jc) {
      constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
      float* tb = tbuf + ((i * NT + j) & 1) * (32 * TP);  // two patches per wave: block k + 1 is written while block k drains
      const int cb = n0 + wc * TN + j * 32;
      const float sc = p.scale[cb + r32], sh = p.shift[cb + r32];
  #pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = acc[i][j][e] * sc + sh;
        v = fmaxf(v, v * slope);  // slope 0.1: == v > 0 ? v : 0.1 v;  slope 1: v
        tb[((e & 3) + 8 * (e >> 2) + 4 * hh) * TP + r32] = v;
      }
  #pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 16 + prow;
        const float4 lo = *reinterpret_cast<const float4*>(tb + row * TP + c8);
        const float4 hi = *reinterpret_cast<const float4*>(tb + row * TP + c8 + 4);
        const long long m = mrow[i][pass];
        if (m >= 0) {
          float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          if (rb) {
            const uint4 r4 = rres[i % RD][j][pass];
            const unsigned rr[4] = {r4.x, r4.y, r4.z, r4.w};
  #pragma unroll
            for (int k = 0; k < 4; ++k) {
              v[2 * k] += DT::from16(rr[k] & 0xffffu);
              v[2 * k + 1] += DT::from16(rr[k] >> 16);
            }
          }
          uint4 o;
          o.x = DT::pack2(v[0], v[1]);
          o.y = DT::pack2(v[2], v[3]);
          o.z = DT::pack2(v[4], v[5]);
          o.w = DT::pack2(v[6], v[7]);
          store16(yb + m * p.y_pitch + cb + c8, o, p.store_mode);
        }
      }
    };
    static_for(
        [&](auto ic) {
          constexpr int i = decltype(ic)::value;
          static_for([&](auto jc) { block_out(ic, jc); }, std::make_integer_sequence<int, NT>{});
          if constexpr (i + RD < MT) fetch_res(std::integral_constant<int, i + RD>{});  // into the slot this row just freed
        },
        std::make_integer_sequence<int, MT>{});
  } else {
    // ---- fp32 epilogue: lane = output channel (32 consecutive floats = one 128-byte segment per pixel row), register =
    // pixel; affine + activation + residual in place, rows decoded from the padded-linear position (16 per block row) ----
    float* __restrict__ yf = reinterpret_cast<float*>(p.y);
    const float* __restrict__ rf = reinterpret_cast<const float*>(p.res);
    const bool leaky = p.act == ME_ACT_LEAKY;
    static_for(
        [&](auto ic) {
          constexpr int i = decltype(ic)::value;
          int mrow[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const long long q = q0 + wr * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
            int m = -1;
            if (q < a.Mp) {
              const unsigned u = (unsigned)q;
              const unsigned n = udiv_magic(u, a.ip_m, a.ip_s);
              const unsigned rem = u - n * (unsigned)a.Ip;
              const unsigned y = udiv_magic(rem, a.wp_m, a.wp_s);
              const unsigned x = rem - y * (unsigned)a.Wp;
              if (y < (unsigned)H && x < (unsigned)W) m = (int)((n * (unsigned)H + y) * (unsigned)W + x);
            }
            mrow[e] = m;
          }
          static_for(
              [&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int co = n0 + wc * TN + j * 32 + r32;
                const float sc = p.scale[co], sh = p.shift[co];
                float rv[16];
                if (rf) {
#pragma unroll
                  for (int e = 0; e < 16; ++e) rv[e] = mrow[e] >= 0 ? rf[(long long)mrow[e] * p.res_pitch + co] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                  float v = acc[i][j][e] * sc + sh;
                  if (leaky) v = v > 0.f ? v : 0.1f * v;
                  if (rf) v += rv[e];
                  if (mrow[e] >= 0) yf[(long long)mrow[e] * p.y_pitch + co] = v;
                }
              },
              std::make_integer_sequence<int, NT>{});
        },
        std::make_integer_sequence<int, MT>{});
  }
  if constexpr (ABL == 10) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_stamp(3);
  }
  if constexpr (ABL == 9) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();
    if (blockIdx.x == 0 && wave == a.stamp_wave && lane == 0 && dbg) dbg[2047] = (unsigned long long)dbg_n;
  }
}


}  // namespace me_p8

// wgrad9.hip - weight gradient of a 3x3 / stride-1 / pad-1 convolution with ALL NINE TAPS of a (cout, cin) tile owned by one
// workgroup (gfx950, fp32 on v_mfma_f32_32x32x2_f32).  Round 4; reference semantics: autograd of the conv blocks of
// module3_our_dataset/yolov3/models.py:22-41 (the detector is differentiable in the reference, models.py:181-267).
//
// Why.  The per-tap kernels of train.hip (conv_wgrad_mfma_kernel / conv_wgrad_tile_kernel) give every (tile, tap) its own
// workgroup: dy and x are fetched nine times per tile, every 16-pixel stage moves 8 KB through registers and LDS for eight
// MFMAs per wave, and a barrier closes each of those short stages - 0.47 of the fp32 matrix peak on the detector step
// (profiles/r04_bench_detector_train_b8_pre.json).  Here a stage of 16 pixels feeds 72 MFMAs per wave (nine accumulators):
// one load of dy, one load of 16 NEW input rows, one barrier per 4608 matrix-pipe cycles.
//
// Index space.  Pixels are walked in the PADDED-LINEAR order of the forward patch kernel (conv_p8_h16.hip): row pitch
// Wp = W + 1 (one zero column behind every row), image pitch Ip = (H + 1) * Wp (one zero row behind every image),
// q = img * Ip + y * Wp + x.  The input pixel of tap (ky, kx) for output position q is q + (ky - 1) * Wp + (kx - 1) for
// EVERY q, so the input rows a stage needs are one contiguous window of the same sequence, sliding by 16 per stage: they
// live in an LDS ring of 16-row slots that the loader fills ONCE per row (plus a halo of ~Wp rows in front of a slice) -
// no per-tap gathers, no row-shaped tiles, 13 x 13 maps as efficient as 52 x 52 ones.  Pad positions carry dy = 0 (and
// zero input): they cost (H + 1)(W + 1) / (HW) - 1 of matrix work (4 % at 52 x 52, 16 % at 13 x 13) and nothing else.
//
// Workgroup = 4 waves = (32 WCO) x (32 WCI) weights x 9 taps; wave = one 32 x 32 block per tap = 144 accumulator registers.
// The pixel range is cut into `splits` slices (blockIdx.z); every slice writes a slab [cout][9][cin] of partial sums that
// train.hip's fixed-order reduction adds (deterministic), exactly like the per-tap kernels.
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct W9Args {
  const float* x;
  long long xp;
  const float* dy;
  long long dyp;
  float* out;      // slabs [split][cout][9][cin]
  int n, h, w, cin, cout;
  int Wp, Ip;
  long long Mp;    // n * Ip padded positions
  int per;         // padded positions per slice (a multiple of 16)
  int hb;          // halo in 16-row rounds: 16 * hb >= Wp + 1
  int nr;          // ring slots = 2 * hb + 2 * NG (+ one mirror slot behind them)
};

struct Pos {       // padded-linear coordinates of one loader row: y == H or x == W is a pad position
  int img, y, x;
};

__device__ __forceinline__ Pos pos_of(long long q, int Ip, int Wp) {  // q >= -Ip
  Pos p;
  const long long qq = q + Ip;
  p.img = (int)(qq / Ip) - 1;
  const int rem = (int)(qq - (long long)(p.img + 1) * Ip);
  p.y = rem / Wp;
  p.x = rem - p.y * Wp;
  return p;
}

__device__ __forceinline__ void advance(Pos& p, int by, int Wp, int H) {
  p.x += by;
  while (p.x >= Wp) {
    p.x -= Wp;
    if (++p.y == H + 1) {
      p.y = 0;
      ++p.img;
    }
  }
}

// NG = 2: eight waves; the two halves of the workgroup take the two 16-position halves of a 32-position step, share the input
// ring and add their accumulators through LDS at the end - two waves per SIMD for the same number of slabs (the slab bytes
// are workgroups x 147 KB: with four waves per workgroup, filling the CUs twice meant twice the slab traffic).
template <int WCO, int WCI, int NG>
__global__ __launch_bounds__(256 * NG) void conv_wgrad9_kernel(W9Args a) {
  constexpr int TCO = 32 * WCO, TCI = 32 * WCI, STEP = 16 * NG;
  constexpr int YV = TCO / 64 > 0 ? TCO / 64 : 1;  // float4 loads of dy per lane and stage (columns cc + 64 v)
  static_assert(WCO * WCI == 4 && TCO >= 64, "four waves; at least 64 output channels per tile");
  extern __shared__ __attribute__((aligned(16))) float smem9[];
  float* Ys = smem9;                    // [2][STEP][TCO]
  float* Xr = smem9 + 2 * STEP * TCO;   // [(nr + 1) * 16][TCI]: slot nr mirrors slot 0, so 31 rows from any slot start are contiguous
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave8 >> 2, wave = wave8 & 3;   // grp: which 16 positions of a step this wave multiplies
  const int wr = wave / WCI, wc = wave % WCI;
  const int ci0 = blockIdx.x * TCI, co0 = blockIdx.y * TCO, split = blockIdx.z;
  const int H = a.h, W = a.w, Wp = a.Wp, NR = a.nr, HALO = 16 * a.hb;
  const long long qb = (long long)split * a.per;
  const long long qe = qb + a.per < a.Mp ? qb + a.per : a.Mp;
  const int stages = (int)((qe - qb + STEP - 1) / STEP);   // steps of STEP positions

  // loader role: row pp of a STEP-row step (= NG rounds of 16), channel quad cc (dy: + 64 v; x: lanes with cc < TCI)
  const int pp = tid >> 4, cc = (tid & 15) * 4;
  const bool x_lane = cc < TCI;
  Pos px = pos_of(qb - HALO + pp, a.Ip, Wp);  // input rounds start a halo in front of the slice
  Pos py = pos_of(qb + pp, a.Ip, Wp);
  long long qy = qb + pp;
  float4 rx = make_float4(0.f, 0.f, 0.f, 0.f), ry[YV];

  auto fetch_x = [&]() {
    rx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x_lane && (unsigned)px.img < (unsigned)a.n && px.y < H && px.x < W)
      rx = *reinterpret_cast<const float4*>(a.x + ((long long)(px.img * H + px.y) * W + px.x) * a.xp + ci0 + cc);
    advance(px, STEP, Wp, H);
  };
  auto fetch_y = [&]() {
#pragma unroll
    for (int v = 0; v < YV; ++v) ry[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (qy < qe && py.y < H && py.x < W) {  // (qy < qe <= Mp: the image index is in range)
      const float* row = a.dy + ((long long)(py.img * H + py.y) * W + py.x) * a.dyp + co0 + cc;
#pragma unroll
      for (int v = 0; v < YV; ++v) ry[v] = *reinterpret_cast<const float4*>(row + 64 * v);
    }
    advance(py, STEP, Wp, H);
    qy += STEP;
  };
  auto store_x = [&](int slot0) {   // the NG rounds of one fetch go to slots slot0 .. slot0 + NG - 1 (NR % NG == 0: no wrap inside)
    if (x_lane) {
      const int slot = slot0 + (pp >> 4), row = pp & 15;
      *reinterpret_cast<float4*>(Xr + (slot * 16 + row) * TCI + cc) = rx;
      if (slot == 0) *reinterpret_cast<float4*>(Xr + (NR * 16 + row) * TCI + cc) = rx;
    }
  };
  auto store_y = [&](int buf) {
#pragma unroll
    for (int v = 0; v < YV; ++v) *reinterpret_cast<float4*>(Ys + (buf * STEP + pp) * TCO + cc + 64 * v) = ry[v];
  };

  // per tap: first ring round c and row offset o of the window of stage 0 (both wave-uniform); sl = slot of round s + c
  int sl[9], oo[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int rel = HALO + (t / 3 - 1) * Wp + (t % 3 - 1);  // >= 0: HALO >= Wp + 1
    oo[t] = rel & 15;
    sl[t] = ((rel >> 4) + grp) % NR;
  }

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  const int i32 = lane & 31, kk = lane >> 5;
  const float* a_lane = Ys + (16 * grp + kk) * TCO + wr * 32 + i32;
  const float* b_lane = Xr + kk * TCI + wc * 32 + i32;

  // prologue: rounds 0 .. 2 hb + NG - 1 of the input (everything step 0 reads), four fetches in flight at a time (one by one
  // they cost a memory latency each: 9 - 15 of them in front of the first MFMA), dy of step 0 into registers
  int wslot = 0;
  const int nf = (2 * a.hb + 2 * NG - 1) / NG;   // prologue fetches of NG rounds each (NG * nf <= NR - NG: no wrap yet)
  fetch_y();
  for (int j = 0; j < nf; j += 4) {
    float4 r4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (j + u < nf) fetch_x();   // (uniform condition)
      r4[u] = rx;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (j + u < nf) {
        rx = r4[u];
        store_x(wslot);
        wslot += NG;
      }
    }
  }

  for (int s = 0; s < stages; ++s) {
    const int buf = s & 1;
    store_y(buf);
    if (s > 0) {  // the rounds requested during step s - 1; their slots held rounds no wave reads any more (NR = 2 hb + 2 NG)
      store_x(wslot);
      wslot = wslot + NG == NR ? 0 : wslot + NG;
    }
    __syncthreads();
    if (s + 1 < stages) {  // in flight while the matrix pipe works
      fetch_x();
      fetch_y();
    }
    const float* ab = a_lane + buf * STEP * TCO;
    const float* bb[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) bb[t] = b_lane + (sl[t] * 16 + oo[t]) * TCI;
    // Operands of k-step pair kp + 1 are requested BETWEEN the first nine MFMAs of pair kp, one ds_read2 behind each MFMA
    // (the scheduling barriers pin that order), and have the second nine to land.  Left alone the compiler reads a whole
    // pair, waits, multiplies, and only then reads the next one; issued as a block in front of the MFMAs (first version) the
    // twenty reads of both waves of a SIMD still left the matrix pipe idle while they issued.
    float av[2][2], bv[2][9][2];
    auto load_b = [&](auto setc, int kp, int t) {
      constexpr int S = decltype(setc)::value;
      bv[S][t][0] = bb[t][2 * (2 * kp) * TCI];
      bv[S][t][1] = bb[t][2 * (2 * kp + 1) * TCI];
    };
    auto load_a = [&](auto setc, int kp) {
      constexpr int S = decltype(setc)::value;
      av[S][0] = ab[2 * (2 * kp) * TCO];
      av[S][1] = ab[2 * (2 * kp + 1) * TCO];
    };
    auto pair = [&](auto curc, auto nxtc, int kp_next) {   // kp_next < 0: nothing to prefetch
      constexpr int C = decltype(curc)::value;
      if (kp_next >= 0) load_a(nxtc, kp_next);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[C][0], bv[C][t][0], acc[t], 0, 0, 0);
        if (kp_next >= 0) load_b(nxtc, kp_next, t);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[C][1], bv[C][t][1], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    load_a(S0{}, 0);
#pragma unroll
    for (int t = 0; t < 9; ++t) load_b(S0{}, 0, t);
    __builtin_amdgcn_sched_barrier(0);
    pair(S0{}, S1{}, 1);
    pair(S1{}, S0{}, 2);
    pair(S0{}, S1{}, 3);
    pair(S1{}, S0{}, -1);
#pragma unroll
    for (int t = 0; t < 9; ++t) sl[t] = sl[t] + NG >= NR ? sl[t] + NG - NR : sl[t] + NG;
  }

  if constexpr (NG == 2) {  // second half -> LDS (the ring is dead), first half adds: fixed order, deterministic
    __syncthreads();
    float* xch = smem9 + (wave * 144) * 64 + lane;   // [wave][tap][e][lane]
    if (grp == 1) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) xch[(t * 16 + e) * 64] = acc[t][e];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] += xch[(t * 16 + e) * 64];
  }

  // C layout of the 32x32 MFMA: lane = column (ci), element e -> row (e & 3) + 8 (e >> 2) + 4 kk (co)
  float* out = a.out + (long long)split * a.cout * 9 * a.cin;
  const int ci = ci0 + wc * 32 + i32;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = co0 + wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
      out[((long long)co * 9 + t) * a.cin + ci] = acc[t][e];
    }
}

int target_wgs() {
  static const int v = [] {
    const char* e = getenv("MILLIEYE_WGRAD9_WGS");
    const int x = e ? atoi(e) : 0;
    return x > 0 ? x : 256;
  }();
  return v;
}

}  // namespace

namespace me_wg9 {

// tile of the nine-tap kernel for a layer, or false when the layer stays on the per-tap kernels
bool shape(int cin, int cout, int ksize, int stride, int pad, int* tco, int* tci) {
  static const bool off = [] {
    const char* e = getenv("MILLIEYE_WGRAD9");
    return e && e[0] == '0';
  }();
  if (off || ksize != 3 || stride != 1 || pad != 1 || cin % 64 || cout % 64) return false;
  *tco = 64;
  *tci = 64;
  return true;
}

// slices of the padded pixel range and positions per slice (a multiple of 16)
int splits(int n, int h, int w, int cin, int cout, int* per) {
  int tco, tci;
  if (!shape(cin, cout, 3, 1, 1, &tco, &tci)) return 0;
  const long long Mp = (long long)n * (h + 1) * (w + 1);
  const long long tiles = (long long)(cin / tci) * (cout / tco);
  long long s = (target_wgs() + tiles - 1) / tiles;
  const long long max_s = (Mp + 255) / 256;  // at least 16 stages per slice: the halo prologue is 2 hb + 1 rounds
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  long long p = (Mp + s - 1) / s;
  p = (p + 31) & ~31ll;
  s = (Mp + p - 1) / p;
  if (per) *per = (int)p;
  return (int)s;
}

// slabs [split][cout][9][cin] of partial sums into `slabs`; the caller reduces them.  Returns < 0 when the layer is not
// eligible after all (map too wide for the LDS ring, operands not 16-byte aligned): the caller falls back to the per-tap kernel.
int launch(const float* x, long long xp, const float* dy, long long dyp, float* slabs, int n, int h, int w, int cin, int cout,
           hipStream_t stream, int* splits_out) {
  int tco, tci;
  if (!shape(cin, cout, 3, 1, 1, &tco, &tci)) return -1;
  if (xp % 4 || dyp % 4 || !me::aligned16(x) || !me::aligned16(dy)) return -1;
  W9Args a = {};
  a.x = x; a.xp = xp; a.dy = dy; a.dyp = dyp; a.out = slabs;
  a.n = n; a.h = h; a.w = w; a.cin = cin; a.cout = cout;
  a.Wp = w + 1;
  a.Ip = (h + 1) * a.Wp;
  a.Mp = (long long)n * a.Ip;
  if (a.Mp >= (1ll << 31)) return -1;
  static const int NG = [] {   // wave groups per workgroup: 2 = eight waves merging through LDS (default), 1 = four waves
    const char* e = getenv("MILLIEYE_WGRAD9_NG");
    return (e && e[0] == '1') ? 1 : 2;
  }();
  a.hb = (a.Wp + 1 + 15) / 16;
  a.nr = 2 * a.hb + 2 * NG;
  size_t lds = ((size_t)2 * 16 * NG * tco + (size_t)(a.nr + 1) * 16 * tci) * sizeof(float);
  const size_t xch = (size_t)4 * 144 * 64 * sizeof(float);   // accumulator exchange of the second half
  if (NG == 2 && lds < xch) lds = xch;
  if (lds > 160 * 1024) return -1;
  const int s = splits(n, h, w, cin, cout, &a.per);
  if (a.Ip <= 16 * a.hb) return -1;  // (pos_of assumes the halo is shorter than one image)
  static bool attr_set = false;
  if (!attr_set) {
    ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad9_kernel<2, 2, 1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad9_kernel<2, 2, 2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  if (NG == 2)
    hipLaunchKernelGGL((conv_wgrad9_kernel<2, 2, 2>), dim3(cin / tci, cout / tco, s), dim3(512), lds, stream, a);
  else
    hipLaunchKernelGGL((conv_wgrad9_kernel<2, 2, 1>), dim3(cin / tci, cout / tco, s), dim3(256), lds, stream, a);
  *splits_out = s;
  return me::check_launch("conv_wgrad9_kernel");
}

}  // namespace me_wg9

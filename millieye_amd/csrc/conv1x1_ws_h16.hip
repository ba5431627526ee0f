// conv1x1_ws_h16.hip - the pointwise (1x1, stride 1) convolutions of the 16-bit storage modes as a streaming kernel (gfx950).
//
// Reference blocks: module3_our_dataset/yolov3/models.py:22-41 with size=1 - the 23 bottleneck convolutions of Darknet-53
// and the 1x1 layers of its three heads.  At bf16 matrix rates these layers are bandwidth work: 256 -> 128 channels on the
// 52x52 map of a 32-frame batch is 66 MB of activations for 5.7 GFLOP (8 us of HBM against 2.3 us of MFMAs), and the whole
// weight matrix is 64 KB.  The per-tap implicit-GEMM kernel (conv_h16.hip) spends ~20 us on each of them - prologue, an
// 8..32-stage dependent chain at DMA latency and epilogue per 128x64 tile, 13.5 % MFMA-busy, 22 non-MFMA instructions per
// MFMA (profiles/r02_*).  This kernel turns the problem around:
//
//   * WEIGHTS STAY IN REGISTERS.  A wave owns 32 output channels for the whole K = cin: cin / 4 VGPRs of B-operand
//     fragments (64 for cin 256), loaded once per workgroup.  No weight traffic, no K loop over memory stages.
//   * ACTIVATIONS STREAM THROUGH AN LDS RING.  A workgroup walks row tiles of BMT pixels (all cin channels of a pixel are
//     contiguous in NHWC, so a tile is one linear byte range - or BMT rows at a pitch for [route] slices) with
//     buffer_load ... lds in 1 KiB pieces, NSLOT tiles deep; rows behind the last pixel are zero-filled by the descriptor's
//     range check (num_records = the bytes that are left), so there is no per-lane bounds logic.  One barrier per tile.
//   * BANK CONFLICTS are avoided by an XOR swizzle applied on the SOURCE side of the DMA (destinations are lane-linear):
//     the 16-byte chunk c of row r is stored at chunk position c ^ f(r) of that row, f(r) = (r / RPL) % min(CR, 16) with
//     CR = chunks per row and RPL = rows per 256-byte bank line; the permutation stays inside aligned 256-byte groups, so the
//     global accesses remain whole lines.  A ds_read_b128 lane group (16 lanes, 16 different rows) then hits 16 bank groups.
//   * PERSISTENT GRID: one workgroup per CU, tile t of workgroup b is b + t * gridDim - per-workgroup fixed cost (weights,
//     scale / shift, descriptors) is paid once per launch, not once per 128x64 tile.
//   * Epilogue as in the other 16-bit kernels: accumulator lane = output channel -> private LDS transpose -> 8 consecutive
//     channels of one pixel per lane, one 16-byte store (affine, LeakyReLU / linear; no residual: no reference block puts a
//     shortcut behind a 1x1 layer - such calls stay on the per-tap kernel, like fp32 outputs and fused upsampling).
//
// Tile id 50 of me_conv2d_h16 (csrc/conv_h16.hip dispatches; the engine's autotuner offers it for every eligible layer).
#include <utility>

#include "conv16_common.h"

namespace {
using namespace me_dma;

struct K1Args {
  const unsigned short* x;
  const unsigned short* wgt;  // [cout][cin]
  const float* scale;
  const float* shift;
  unsigned short* y;
  long long x_pitch, y_pitch;  // elements
  int M, cout, act, tiles_m, grid_m, store_mode;
};

template <class F, int... J>
__device__ __forceinline__ void sfor(F&& f, std::integer_sequence<int, J...>) {
  (f(std::integral_constant<int, J>{}), ...);
}

__device__ __forceinline__ void dma_one(unsigned v, u32x4 r, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[r], 0 offen lds\n\t"
               "s_mov_b32 m0, %[k]"
               : [k] "=&s"(keep)
               : [d] "s"(dst), [r] "s"(r), [v] "v"(v)
               : "memory", "scc");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// CIN input channels (compile time: the weight fragments are a register array), WN waves along the output channels (32 each),
// WM waves along the rows (32 each), NSLOT tiles in the LDS ring.
template <int CIN, int WN, int WM, int NSLOT, int F16>
__global__ __launch_bounds__(64 * WN * WM) void conv1x1_ws_kernel(K1Args a) {
  using v8 = typename H16<F16>::v8;
  constexpr int NW = WN * WM, BMT = 32 * WM;
  constexpr int CR = CIN / 8;                    // 16-byte chunks per row
  constexpr int ROWB = CIN * 2, TILEB = BMT * ROWB;
  constexpr int ND = TILEB / 1024 / NW;          // DMA instructions per wave and tile
  static_assert(TILEB % (1024 * NW) == 0 && ND >= 1 && ND <= 8, "tile bytes must split into whole DMA pieces per wave");
  constexpr int KS = CIN / 16;                   // MFMA k-steps
  constexpr int RPL = CR >= 16 ? 1 : 16 / CR;    // rows per 256-byte bank line
  constexpr int FM = CR >= 16 ? 16 : CR;         // swizzle modulus
  constexpr int TP = 36;                         // transpose patch pitch (floats)
  constexpr unsigned RING = (unsigned)NSLOT * TILEB;
  static_assert(NSLOT >= 2 && (NSLOT - 2) * ND <= 60, "ring depth");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem1[];
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % WN, wm = wave / WN;
  const int r32 = lane & 31, hh = lane >> 5;
  const int bid = blockIdx.x;
  const int tile_n = bid / a.grid_m, b_m = bid - tile_n * a.grid_m;
  const int n0 = tile_n * (32 * WN) + wn * 32;   // this wave's 32 output channels

  // ---- weights of this wave: KS fragments (lane: channel n0 + r32, k = 16 ks + 8 hh .. + 8) -------------------------------
  v8 wf[KS];
  {
    const unsigned short* wrow = a.wgt + (long long)(n0 + r32) * CIN + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wf[ks] = *reinterpret_cast<const v8*>(wrow + 16 * ks);
  }
  const float sc = a.scale[n0 + r32], sh = a.shift[n0 + r32];
  const float slope = a.act == ME_ACT_LEAKY ? 0.1f : 1.0f;

  // ---- per-lane DMA source offsets: this wave moves pieces wave * ND .. + ND of a tile; piece q, lane l = chunk q * 64 + l of
  // the tile in LDS order = (row, position); it fetches chunk position ^ f(row) of that row ----------------------------------
  unsigned v_off[ND];
  const unsigned pitchb = (unsigned)(a.x_pitch * 2);
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int q = (wave * ND + i) * 64 + lane;
    const int row = q / CR, pos = q % CR;
    const int f = (row / RPL) % FM;
    v_off[i] = (unsigned)row * pitchb + (unsigned)(pos ^ f) * 16u;
  }
  const unsigned wave_dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(wave * ND) * 1024u);
  const unsigned long long xbase = (unsigned long long)a.x;
  const long long tile_bytes_g = (long long)BMT * pitchb;
  auto issue = [&](int t, int slot) {   // tile t -> ring slot; tiles behind the end: all lanes out of range (zero fill, no traffic)
    const long long row0 = (long long)t * BMT;
    long long left = ((long long)a.M - row0) * (long long)pitchb;
    if (left < 0 || t >= a.tiles_m) left = 0;
    if (left > tile_bytes_g) left = tile_bytes_g;
    const unsigned long long b = xbase + (unsigned long long)(t < a.tiles_m ? row0 : 0) * pitchb;
    u32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane((unsigned)left);
    r.w = 0x00020000u;
    const unsigned dst = wave_dst + (unsigned)slot * TILEB;
    sfor([&](auto ic) { dma_one(v_off[decltype(ic)::value], r, dst + decltype(ic)::value * 1024u); },
         std::make_integer_sequence<int, ND>{});
  };

  // ---- A-fragment addressing: row wm * 32 + r32 of a tile, chunk 2 ks + hh stored at (2 ks + hh) ^ f(row) --------------------
  const int arow = wm * 32 + r32;
  const unsigned a_base = (unsigned)arow * ROWB;
  const unsigned g16 = (unsigned)((hh ^ ((arow / RPL) % FM)) * 16);  // (2 ks + hh) ^ f = (2 ks) ^ (hh ^ f): hh is bit 0
  float* tb = reinterpret_cast<float*>(smem1 + RING) + wave * (32 * TP);
  const int prow = lane >> 2, c8 = (lane & 3) * 8;

  // ---- prologue: NSLOT - 1 tiles in flight -------------------------------------------------------------------------------------
  const int G = a.grid_m;
#pragma unroll
  for (int s = 0; s < NSLOT - 1; ++s) issue(b_m + s * G, s);

  int slot = 0;
  for (int t = b_m; t < a.tiles_m; t += G) {
    // tile t's pieces are the oldest loads in flight; younger: the NSLOT - 2 tiles behind it (stores of earlier epilogues only
    // make this wait more conservative: loads complete in order among themselves)
    wait_vm<(NSLOT - 2) * ND>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {  // refill the slot tile t - G just left (every wave is past its reads: it is behind the barrier)
      const int ps = slot == 0 ? NSLOT - 1 : slot - 1;
      issue(t + (NSLOT - 1) * G, ps);
    }
    const unsigned char* At = smem1 + (unsigned)slot * TILEB + a_base;
    f32x16 acc, acc1;  // two accumulation chains (even / odd k-steps): a lone wave per SIMD is bound by the dependent-MFMA
                       // latency, not by the issue rate
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = acc1[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
      const v8 af0 = *reinterpret_cast<const v8*>(At + (((unsigned)(ks * 32)) ^ g16));
      const v8 af1 = *reinterpret_cast<const v8*>(At + (((unsigned)((ks + 1) * 32)) ^ g16));
      acc = H16<F16>::mfma(af0, wf[ks], acc);
      acc1 = H16<F16>::mfma(af1, wf[ks + 1], acc1);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] += acc1[e];
    // ---- epilogue of this wave's 32 x 32 block ----------------------------------------------------------------------------------
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v = acc[e] * sc + sh;
      v = fmaxf(v, v * slope);
      tb[((e & 3) + 8 * (e >> 2) + 4 * hh) * TP + r32] = v;
    }
    const long long row0 = (long long)t * BMT + wm * 32;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int row = pass * 16 + prow;
      const float4 lo = *reinterpret_cast<const float4*>(tb + row * TP + c8);
      const float4 hi = *reinterpret_cast<const float4*>(tb + row * TP + c8 + 4);
      const long long m = row0 + row;
      if (m < a.M) {
        uint4 o;
        o.x = pack2<F16>(lo.x, lo.y);
        o.y = pack2<F16>(lo.z, lo.w);
        o.z = pack2<F16>(hi.x, hi.y);
        o.w = pack2<F16>(hi.z, hi.w);
        me::store16(a.y + m * a.y_pitch + n0 + c8, o, a.store_mode);
      }
    }
    slot = slot + 1 == NSLOT ? 0 : slot + 1;
  }
  wait_vm<0>();  // (zero-range refills of the last iterations still count)
}

template <int CIN, int WN, int WM, int NSLOT>
int launch_ws(const Conv16P& p, hipStream_t stream) {
  constexpr int NW = WN * WM, BMT = 32 * WM;
  K1Args a;
  a.x = p.x; a.wgt = p.wgt; a.scale = p.scale; a.shift = p.shift; a.y = reinterpret_cast<unsigned short*>(p.y);
  a.x_pitch = p.x_pitch; a.y_pitch = p.y_pitch;
  a.M = p.M; a.cout = p.cout; a.act = p.act; a.store_mode = p.store_mode;
  a.tiles_m = (p.M + BMT - 1) / BMT;
  const int tiles_n = p.cout / (32 * WN);
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  const size_t lds = (size_t)NSLOT * BMT * CIN * 2 + (size_t)NW * 32 * 36 * sizeof(float);
  ME_REQUIRE(lds <= 160 * 1024, ME_E_TOOBIG, "me_conv2d_h16: tile 50 needs %zu bytes of LDS", lds);
  int per_cu = (int)(160 * 1024 / lds);   // resident workgroups per CU by LDS (registers: cin / 4 + ~40 VGPRs per wave)
  const int by_regs = 8 / (NW / 4) / (CIN >= 512 ? 4 : CIN >= 384 ? 3 : CIN >= 256 ? 2 : 1);
  if (per_cu > by_regs) per_cu = by_regs;
  if (per_cu < 1) per_cu = 1;
  int per_n = cus * per_cu / tiles_n;
  if (per_n < 1) per_n = 1;
  a.grid_m = a.tiles_m < per_n ? a.tiles_m : per_n;
  const dim3 grid((unsigned)(a.grid_m * tiles_n)), block(64 * NW);
  if (p.f16) {
    auto kern = conv1x1_ws_kernel<CIN, WN, WM, NSLOT, 1>;
    static bool attr = false;
    if (!attr) {
      ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr = true;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  } else {
    auto kern = conv1x1_ws_kernel<CIN, WN, WM, NSLOT, 0>;
    static bool attr = false;
    if (!attr) {
      ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr = true;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  }
  return me::check_launch("conv1x1_ws_h16");
}

}  // namespace

namespace me16 {

// (cin, cout) pairs with a weight-stationary instance: cout / 32 waves along the channels (<= 8), cin / 4 VGPRs of weights
bool ws1x1_eligible(const Conv16P& p) {
  if (p.ks != 1 || p.stride != 1 || p.pad != 0 || p.ups != 1 || p.x_nchw || p.res || !p.vec_epi || p.y_f32) return false;
  if (p.x_pitch % 8 || !me::aligned16(p.x) || !me::aligned16(p.wgt)) return false;
  if ((long long)p.x_pitch * 2 * 128 >= (1ll << 31)) return false;
  const int c = p.cin, o = p.cout;
  return (c == 64 && o == 32) || (c == 128 && o == 64) || (c == 256 && o == 128) || (c == 384 && o == 128) ||
         (c == 512 && o == 256) || (c == 768 && o == 256) || (c == 256 && o == 256) || (c == 512 && o == 512) ||
         (c == 128 && o == 128);
}

int launch_ws1x1(const Conv16P& p, hipStream_t stream) {
  ME_REQUIRE(ws1x1_eligible(p), ME_E_BADARG,
             "me_conv2d_h16: tile 50 (weight-stationary 1x1) needs a 1x1 / stride 1 layer without residual / upsampling / fp32 "
             "output, 16-byte aligned operands and one of the built (cin, cout) pairs; got %d -> %d", p.cin, p.cout);
  const int c = p.cin, o = p.cout;
  static const int variant = [] {
    const char* e = getenv("MILLIEYE_WS_VARIANT");
    return e ? atoi(e) : 0;
  }();
  if (variant && c == 256 && o == 128) {   // tuning: occupancy / ring-depth variants of the 256 -> 128 instance
    switch (variant) {
      case 1: return launch_ws<256, 4, 1, 3>(p, stream);   // 66 KB: two workgroups per CU
      case 2: return launch_ws<256, 4, 1, 2>(p, stream);   // 50 KB: three workgroups per CU
      case 3: return launch_ws<256, 4, 2, 3>(p, stream);   // 8 waves, 64-row tiles, one workgroup per CU
      case 4: return launch_ws<256, 4, 2, 2>(p, stream);
      case 5: return launch_ws<256, 4, 1, 4>(p, stream);   // 82 KB: one per CU with a shorter ring
      default: break;
    }
  }
  if (variant && c == 512 && o == 256) {
    switch (variant) {
      case 1: return launch_ws<512, 8, 1, 2>(p, stream);
      default: break;
    }
  }
  //                           CIN  WN WM NSLOT        LDS: NSLOT * 32 WM * CIN * 2 + patches
  if (c == 64 && o == 32) return launch_ws<64, 1, 4, 8>(p, stream);      // 8 x 16 KB + 18 KB
  if (c == 128 && o == 64) return launch_ws<128, 2, 2, 8>(p, stream);    // 8 x 16 KB
  if (c == 128 && o == 128) return launch_ws<128, 4, 1, 8>(p, stream);   // 8 x  8 KB
  if (c == 256 && o == 128) return launch_ws<256, 4, 1, 8>(p, stream);   // 8 x 16 KB
  if (c == 256 && o == 256) return launch_ws<256, 8, 1, 6>(p, stream);   // 6 x 16 KB + 37 KB
  if (c == 384 && o == 128) return launch_ws<384, 4, 1, 5>(p, stream);   // 5 x 24 KB
  if (c == 512 && o == 256) return launch_ws<512, 8, 1, 3>(p, stream);   // 3 x 32 KB + 37 KB
  if (c == 512 && o == 512) return launch_ws<512, 8, 1, 3>(p, stream);   // two column tiles of 256 channels
  return launch_ws<768, 8, 1, 2>(p, stream);                             // 2 x 48 KB + 37 KB
}

}  // namespace me16

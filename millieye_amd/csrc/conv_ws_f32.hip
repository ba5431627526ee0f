// conv_ws_f32.hip - weight-stationary fp32 convolutions for the short-K layers of the detector (gfx950).
//   tile id 50 of me_conv2d_f32: 1x1 / stride 1 layers with cin <= 512  (conv1x1_ws_f32_kernel)
//   tile id 60 of me_conv2d_f32: 3x3 / pad 1 / stride 1 or 2 layers with cin 32 / 64  (conv3x3_ws_f32_kernel)
//
// Reference blocks: module3_our_dataset/yolov3/models.py:22-41 (conv + folded BatchNorm + LeakyReLU), :258-260 (shortcut).
// The per-tap implicit GEMM (conv.hip, conv_igemm_buf_f32) pays a fixed price per 64x64 tile - index decode, a prologue at DMA
// latency, the epilogue - worth about nine of its 16-channel K stages; a 1x1 layer with 256 input channels has sixteen stages,
// so the thirty-one 1x1 bottlenecks of Darknet-53 ran at 0.61 of the fp32 matrix peak beside 0.85 for the 3x3 layers
// (profiles/r04_layers_f32_b32.txt), and the 3x3 layers on the 208 / 104 maps (K = 288 / 576) at 0.73 - 0.79.  Here, as in the
// 16-bit kernels conv1x1_ws_h16.hip / conv3x3_ws_h16.hip:
//   * WEIGHTS STAY IN REGISTERS: a wave owns 32 output channels for the whole K - K / 2 VGPRs of B operands of
//     v_mfma_f32_32x32x2_f32 (128 for a 256-channel 1x1, 144 / 288 for the 3x3 filters), loaded once per workgroup;
//   * ACTIVATIONS STREAM THROUGH AN LDS RING by buffer_load ... lds (1 KiB per wave instruction, rows / pixels behind the
//     end or outside the frame zero-filled by the descriptor's range check), XOR bank swizzle on the DMA's source side;
//   * PERSISTENT GRID: per-workgroup fixed cost once per launch; a tile costs one barrier, its DMA issue and its epilogue;
//   * the wait for tile t + 1's pieces sits BETWEEN tile t's MFMAs and tile t's epilogue: everything older than those pieces
//     (the previous epilogue's stores) has had a whole tile of MFMAs to complete, and tile t's stores drain under tile
//     t + 1's MFMAs - a wave alone on its SIMD (cin 512: 256 weight registers) never waits for its own stores.
// Numerics: one accumulation chain per output, channels in the order of conv_igemm_buf_f32 (per 16-channel stage: channels
// {k, k + 4} for k = 0..3, then {8 + k, 12 + k}; 3x3: tap-major for cin 32, chunk-major for cin 64 - what that kernel runs
// for these shapes), same epilogue expression: bit-identical to tile 3 without split-K (tests/test_gpu_ops.py).
#include <utility>

#include "conv32_common.h"

namespace {
using namespace me_dma;

template <class F, int... J>
__device__ __forceinline__ void sforw(F&& f, std::integer_sequence<int, J...>) {
  (f(std::integral_constant<int, J>{}), ...);
}

__device__ __forceinline__ void dma_onew(unsigned v, u32x4 r, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[r], 0 offen lds\n\t"
               "s_mov_b32 m0, %[k]"
               : [k] "=&s"(keep)
               : [d] "s"(dst), [r] "s"(r), [v] "v"(v)
               : "memory", "scc");
}

template <int N>
__device__ __forceinline__ void wait_vmw() {
  static_assert(N >= 0 && N <= 63, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ float f4at(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// ---------------------------------------------------------------------------------------------------------------------------
// 1x1
// ---------------------------------------------------------------------------------------------------------------------------
struct W1Args {
  const float* x;
  const float* wgt;  // [cout][cin]
  const float* scale;
  const float* shift;
  float* y;
  long long x_pitch, y_pitch;  // elements
  int M, cout, act, tiles_m, grid_m, store_mode;
  int abl;  // tuning only (MILLIEYE_WS32_ABL): 1 = every DMA lane out of range (no traffic), 2 = no output stores
};

// CIN input channels, WN waves along the output channels (32 each), WM waves along the rows (32 each), NSLOT tiles in the LDS
// ring, MINW = waves per SIMD the register budget has to allow (2: two workgroups per CU overlap epilogue and MFMAs).
template <int CIN, int WN, int WM, int NSLOT, int MINW>
__global__ __launch_bounds__(64 * WN * WM, MINW) void conv1x1_ws_f32_kernel(W1Args a) {
  constexpr int NW = WN * WM, BMT = 32 * WM;
  constexpr int CR = CIN / 4;                    // 16-byte chunks per row (>= 16: a row is one or more 256-byte bank lines)
  constexpr int ROWB = CIN * 4, TILEB = BMT * ROWB;
  constexpr int ND = TILEB / 1024 / NW;          // DMA instructions per wave and tile
  static_assert(CIN % 64 == 0, "rows must be whole bank lines");
  static_assert(TILEB % (1024 * NW) == 0 && ND >= 1 && ND <= 16, "tile bytes must split into whole DMA pieces per wave");
  static_assert(NSLOT >= 2 && (NSLOT - 2) * ND <= 63, "ring depth");
  constexpr int KG = CIN / 8;                    // groups of 8 channels = one ds_read_b128 + four MFMAs

  extern __shared__ __attribute__((aligned(16))) unsigned char smemw[];
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smemw;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % WN, wm = wave / WN;
  const int r32 = lane & 31, hh = lane >> 5;
  const int bid = blockIdx.x;
  const int tile_n = bid / a.grid_m, b_m = bid - tile_n * a.grid_m;
  const int n0 = tile_n * (32 * WN) + wn * 32;   // this wave's 32 output channels
  const int co = n0 + r32;
  const bool co_ok = co < a.cout;

  // ---- weights of this wave: lane (channel co, k half hh) holds channels 8 g + 4 hh .. + 3 of every group g --------------------
  float4 wf[KG];
  {
    const float* wrow = a.wgt + (long long)(co_ok ? co : 0) * CIN + 4 * hh;
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      wf[g] = *reinterpret_cast<const float4*>(wrow + 8 * g);
      if (!co_ok) wf[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float sc = co_ok ? a.scale[co] : 0.f, sh = co_ok ? a.shift[co] : 0.f;
  const float slope = a.act == ME_ACT_LEAKY ? 0.1f : 1.0f;

  // ---- DMA lanes: this wave moves pieces wave * ND .. + ND of a tile; piece q, lane l = LDS chunk q * 64 + l = (row, position);
  // it fetches chunk position ^ (row % 16) of that row ----------------------------------------------------------------------
  unsigned v_off[ND];
  const unsigned pitchb = (unsigned)(a.x_pitch * 4);
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int q = (wave * ND + i) * 64 + lane;
    const int row = q / CR, pos = q % CR;
    v_off[i] = (unsigned)row * pitchb + (unsigned)(pos ^ (row & 15)) * 16u;
  }
  const unsigned wave_dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(wave * ND) * 1024u);
  const unsigned long long xbase = (unsigned long long)a.x;
  const long long tile_bytes_g = (long long)BMT * pitchb;
  auto issue = [&](int t, int slot) {   // tile t -> ring slot; tiles behind the end: every lane out of range (zero fill, no traffic)
    const long long row0 = (long long)t * BMT;
    long long left = ((long long)a.M - row0) * (long long)pitchb;
    if (left < 0 || t >= a.tiles_m) left = 0;
    if (left > tile_bytes_g) left = tile_bytes_g;
    if (a.abl & 1) left = 0;
    const unsigned long long b = xbase + (unsigned long long)(t < a.tiles_m ? row0 : 0) * pitchb;
    u32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane((unsigned)left);
    r.w = 0x00020000u;
    const unsigned dst = wave_dst + (unsigned)slot * TILEB;
    sforw([&](auto ic) { dma_onew(v_off[decltype(ic)::value], r, dst + decltype(ic)::value * 1024u); },
          std::make_integer_sequence<int, ND>{});
  };

  // ---- A fragments: row wm * 32 + r32 of a tile, chunk 2 g + hh stored at (2 g + hh) ^ (row % 16) ------------------------------
  const int arow = wm * 32 + r32;
  const unsigned a_base = (unsigned)arow * ROWB;
  const unsigned g16 = (unsigned)((hh ^ (arow & 15)) * 16);  // (2 g + hh) ^ f = (2 g) ^ (hh ^ f): hh is bit 0 of the chunk index

  const int G = a.grid_m;
#pragma unroll
  for (int s = 0; s < NSLOT - 1; ++s) issue(b_m + s * G, s);
  wait_vmw<(NSLOT - 2) * ND>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  int slot = 0;
  for (int t = b_m; t < a.tiles_m; t += G) {
    {  // refill the slot tile t - G left (every wave passed the barrier behind its reads)
      const int ps = slot == 0 ? NSLOT - 1 : slot - 1;
      issue(t + (NSLOT - 1) * G, ps);
    }
    const unsigned char* At = smemw + (unsigned)slot * TILEB + a_base;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    {
      // fragment reads run PF groups ahead of their MFMAs (sched_barrier: hipcc otherwise sinks half of them to right in front
      // of their first use, and a wave alone on its SIMD then waits out the LDS latency every four MFMAs)
      constexpr int PF = 2;
      float4 af[PF + 1];
      auto rd = [&](int g) { return *reinterpret_cast<const float4*>(At + (((unsigned)(g * 32)) ^ g16)); };
#pragma unroll
      for (int g = 0; g < PF && g < KG; ++g) af[g] = rd(g);
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        if (g + PF < KG) af[(g + PF) % (PF + 1)] = rd(g + PF);
        __builtin_amdgcn_sched_barrier(0);
        const float4 av = af[g % (PF + 1)];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f4at(av, i), f4at(wf[g], i), acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // tile t + G: this wave's pieces have landed (younger: the refills issued since), then everyone's; everyone is done with
    // this slot.  Older stores are a whole tile of MFMAs old.
    wait_vmw<(NSLOT - 2) * ND>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- epilogue: accumulator lane = output channel, register e = row (e & 3) + 8 (e >> 2) + 4 hh ---------------------------
    {
      const long long row0 = (long long)t * BMT + wm * 32 + 4 * hh;
      float* yp = a.y + row0 * a.y_pitch + co;
      const long long ystep = a.y_pitch;
      const int left = a.M - (int)row0;  // rows e with (e & 3) + 8 (e >> 2) < left exist
      if (co_ok && !(a.abl & 2)) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float tv = acc[e] * sc + sh;
          if ((e & 3) + 8 * (e >> 2) < left) me::store4(yp, fmaxf(tv, tv * slope), a.store_mode);
          yp += (e & 3) == 3 ? 5 * ystep : ystep;
        }
      }
    }
    slot = slot + 1 == NSLOT ? 0 : slot + 1;
  }
  wait_vmw<0>();  // (zero-range refills of the last iterations still count)
}

int ws_abl() {
  static const int v = [] {
    const char* e = getenv("MILLIEYE_WS32_ABL");
    return e ? atoi(e) : 0;
  }();
  return v;
}

int cu_count() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

template <int CIN, int WN, int WM, int NSLOT, int MINW>
int launch_w1(const ConvP& p, hipStream_t stream) {
  constexpr int NW = WN * WM, BMT = 32 * WM;
  W1Args a;
  a.x = p.x; a.wgt = p.wgt; a.scale = p.scale; a.shift = p.shift; a.y = p.y;
  a.x_pitch = p.x_pitch; a.y_pitch = p.y_pitch;
  a.M = p.M; a.cout = p.cout; a.act = p.act; a.store_mode = p.store_mode;
  a.abl = ws_abl();
  a.tiles_m = (p.M + BMT - 1) / BMT;
  const int tiles_n = (p.cout + 32 * WN - 1) / (32 * WN);
  const size_t lds = (size_t)NSLOT * BMT * CIN * 4;
  ME_REQUIRE(lds <= 160 * 1024, ME_E_TOOBIG, "me_conv2d_f32: tile 50 needs %zu bytes of LDS", lds);
  int per_cu = (int)(160 * 1024 / lds);
  const int by_regs = MINW * 4 / NW;   // workgroups per CU the register budget allows
  if (per_cu > by_regs) per_cu = by_regs;
  if (per_cu < 1) per_cu = 1;
  int per_n = cu_count() * per_cu / tiles_n;
  per_n = per_n / 8 * 8;               // the column tiles of one row tile on the same XCD (blockIdx % 8)
  if (per_n < 8) per_n = 8;
  a.grid_m = a.tiles_m < per_n ? a.tiles_m : per_n;
  const dim3 grid((unsigned)(a.grid_m * tiles_n)), block(64 * NW);
  auto kern = conv1x1_ws_f32_kernel<CIN, WN, WM, NSLOT, MINW>;
  static bool attr = false;
  if (!attr) {
    ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  if (getenv("MILLIEYE_WS32_DEBUG")) {
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), 64 * NW, lds);
    fprintf(stderr, "[ws32] 1x1 cin %d: grid %u x %d threads, lds %zu, tiles_m %d grid_m %d tiles_n %d, occupancy API %d blocks/CU\n", CIN,
            grid.x, 64 * NW, lds, a.tiles_m, a.grid_m, tiles_n, nb);
  }
  hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  return me::check_launch("conv1x1_ws_f32");
}

// ---------------------------------------------------------------------------------------------------------------------------
// 3x3 (cin 32 / 64)
// ---------------------------------------------------------------------------------------------------------------------------
struct W3Args {
  const float* x;
  const float* wgt;  // [cout][3][3][cin]
  const float* scale;
  const float* shift;
  const float* res;
  float* y;
  long long x_pitch, res_pitch, y_pitch;  // elements
  int n, h, w, ho, wo, cout, act;
  int tiles_y, tiles_x, tiles_total, grid_m, store_mode;
};

// An output tile is TH x TW = (TMY * MBH) x (TMX * MBW) pixels, an MFMA row block MBH x MBW = 32 pixels of it; its input patch
// ((TH - 1) S + 3) x ((TW - 1) S + 3) pixels lives in one ring slot as [patch pixel][CIN] with the 16-byte chunks of a pixel
// XOR-swizzled (source side of the DMA).  The nine taps are pixel shifts inside the resident patch.
template <int CIN, int WN, int WM, int MBH, int MBW, int TMY, int TMX, int S, int NSLOT, int MINW, int RES>
__global__ __launch_bounds__(64 * WN * WM, MINW) void conv3x3_ws_f32_kernel(W3Args a) {
  static_assert(MBH * MBW == 32, "an MFMA row block is 32 pixels");
  constexpr int NW = WN * WM, TH = TMY * MBH, TW = TMX * MBW, NMB = TMY * TMX;
  static_assert(NMB % WM == 0, "row blocks must split evenly over the wave rows");
  constexpr int MPW = NMB / WM;                       // row blocks per wave and tile
  constexpr int PH = (TH - 1) * S + 3, PWP = (TW - 1) * S + 3, NPIX = PH * PWP;
  constexpr int PIXB = CIN * 4, CPP = CIN / 4;        // bytes / 16-byte chunks per pixel
  static_assert(CPP == 8 || CPP == 16, "cin 32 or 64");
  constexpr int PIECES = (NPIX * PIXB + 1023) / 1024;
  constexpr int ND = (PIECES + NW - 1) / NW;          // DMA instructions per wave and tile
  constexpr unsigned SLOTB = (unsigned)ND * NW * 1024u;
  constexpr int KG = CIN / 8;                         // 8-channel groups per tap
  constexpr bool CHUNK_MAJOR = CIN >= 64;             // the K order conv_igemm_buf_f32 runs for the shape (choose_order)
  static_assert(NSLOT >= 2 && (NSLOT - 2) * ND <= 63 && ND <= 24, "ring depth");

  extern __shared__ __attribute__((aligned(16))) unsigned char smemw3[];
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smemw3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % WN, wm = wave / WN;
  const int r32 = lane & 31, hh = lane >> 5;
  const int bid = blockIdx.x;
  const int tile_n = bid / a.grid_m, b_m = bid - tile_n * a.grid_m;
  const int n0 = tile_n * (32 * WN) + wn * 32;
  const int co = n0 + r32;
  const bool co_ok = co < a.cout;

  auto swz = [](int pidx) { return CPP == 8 ? ((pidx >> 1) & 7) : (pidx & 15); };

  // ---- weights: lane (channel co, k half hh) holds channels 8 g + 4 hh .. + 3 of tap t in wf[t * KG + g] ------------------------
  float4 wf[9 * KG];
  {
    const float* wrow = a.wgt + (long long)(co_ok ? co : 0) * (9 * CIN) + 4 * hh;
#pragma unroll
    for (int k = 0; k < 9 * KG; ++k) {
      wf[k] = *reinterpret_cast<const float4*>(wrow + 8 * k);
      if (!co_ok) wf[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float sc = co_ok ? a.scale[co] : 0.f, sh = co_ok ? a.shift[co] : 0.f;
  const float slope = a.act == ME_ACT_LEAKY ? 0.1f : 1.0f;

  // ---- DMA lanes: piece (wave * ND + i), lane l = chunk q of the slot = (patch pixel, chunk position) -----------------------
  unsigned pyx[ND];  // py | px << 12 | source chunk << 24 | valid << 31
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int q = (wave * ND + i) * 64 + lane;
    const int pidx = q / CPP, pos = q % CPP;
    const int py = pidx / PWP, px = pidx - py * PWP;
    pyx[i] = (unsigned)py | ((unsigned)px << 12) | ((unsigned)(pos ^ swz(pidx)) << 24) | (pidx < NPIX ? 0x80000000u : 0u);
  }
  const unsigned pitchb = (unsigned)(a.x_pitch * 4);
  const unsigned wave_dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(wave * ND) * 1024u);
  const unsigned long long xbase = (unsigned long long)a.x;
  const int per_img = a.tiles_y * a.tiles_x;
  const unsigned img_bytes = (unsigned)a.h * (unsigned)a.w * pitchb;  // < 2^31 (checked by the launcher)

  auto issue = [&](int t, int slot) {
    u32x4 r;
    int iy0 = 0, ix0 = 0;
    unsigned long long b = xbase;
    unsigned recs = 0;  // tiles behind the end: every lane out of range
    if (t < a.tiles_total) {
      const int nimg = t / per_img, rem = t - nimg * per_img;
      const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
      iy0 = ty * (TH * S) - 1;
      ix0 = tx * (TW * S) - 1;
      b = xbase + (unsigned long long)nimg * img_bytes;
      recs = img_bytes;
    }
    r.x = __builtin_amdgcn_readfirstlane((unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane(recs);
    r.w = 0x00020000u;
    const unsigned dst = wave_dst + (unsigned)slot * SLOTB;
    sforw(
        [&](auto ic) {
          constexpr int i = decltype(ic)::value;
          const unsigned e = pyx[i];
          const int iy = iy0 + (int)(e & 0xfffu), ix = ix0 + (int)((e >> 12) & 0xfffu);
          unsigned off = kOobOffset;
          if ((e & 0x80000000u) && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w)
            off = ((unsigned)iy * (unsigned)a.w + (unsigned)ix) * pitchb + ((e >> 24) & 0xfu) * 16u;
          dma_onew(off, r, dst + i * 1024u);
        },
        std::make_integer_sequence<int, ND>{});
  };

  // ---- A fragments: row block mb = wm + j * WM; lane pixel (my, mx) of it; patch pixel of tap (dy, dx) = pb + dy * PWP + dx ----
  int pb[MPW];
#pragma unroll
  for (int j = 0; j < MPW; ++j) {
    const int mb = wm + j * WM;
    const int oy = (mb / TMX) * MBH + r32 / MBW, ox = (mb % TMX) * MBW + r32 % MBW;
    pb[j] = oy * S * PWP + ox * S;
  }

  const int G = a.grid_m;
#pragma unroll
  for (int s = 0; s < NSLOT - 1; ++s) issue(b_m + s * G, s);
  wait_vmw<(NSLOT - 2) * ND>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  int slot = 0;
  for (int t = b_m; t < a.tiles_total; t += G) {
    {
      const int ps = slot == 0 ? NSLOT - 1 : slot - 1;
      issue(t + (NSLOT - 1) * G, ps);
    }
    const int nimg = t / per_img, rem = t - nimg * per_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const unsigned char* At = smemw3 + (unsigned)slot * SLOTB;
    // register e of row block mb = pixel q = (e & 3) + 8 (e >> 2) + 4 hh of the block -> dense output pixel, -1 outside the map
    auto out_pixel = [&](int j, int e) -> long long {
      const int mb = wm + j * WM;
      const int q = (e & 3) + 8 * (e >> 2) + 4 * hh;
      const int oy = ty * TH + (mb / TMX) * MBH + q / MBW, ox = tx * TW + (mb % TMX) * MBW + q % MBW;
      return (oy < a.ho && ox < a.wo) ? ((long long)nimg * a.ho + oy) * a.wo + ox : -1;
    };
    f32x16 acc[MPW];
    float rres[RES ? MPW : 1][16];
#pragma unroll
    for (int j = 0; j < MPW; ++j) {
      if constexpr (RES) {  // the shortcut operand of this block: requested in front of its MFMAs, used behind them
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const long long m = out_pixel(j, e);
          rres[j][e] = (co_ok && m >= 0) ? a.res[m * a.res_pitch + co] : 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
      // step k of the K walk -> (tap, 8-channel group): chunk-major = 16-channel chunk outer, taps inner, the chunk's two groups
      auto tap_of = [](int k) { return CHUNK_MAJOR ? (k / 2) % 9 : k / KG; };
      auto grp_of = [](int k) { return CHUNK_MAJOR ? 2 * (k / 18) + (k & 1) : k % KG; };
      auto rd = [&](int k) {
        const int tap = tap_of(k), g = grp_of(k);
        const int pidx = pb[j] + (tap / 3) * PWP + (tap % 3);
        const unsigned addr = (unsigned)pidx * PIXB + (((unsigned)(g * 32)) ^ (unsigned)((hh ^ swz(pidx)) * 16));
        return *reinterpret_cast<const float4*>(At + addr);
      };
      constexpr int PF = 2, KS = 9 * KG;
      float4 af[PF + 1];
#pragma unroll
      for (int k = 0; k < PF; ++k) af[k] = rd(k);
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        if (k + PF < KS) af[(k + PF) % (PF + 1)] = rd(k + PF);
        __builtin_amdgcn_sched_barrier(0);
        const float4 av = af[k % (PF + 1)];
        const float4 wv = wf[tap_of(k) * KG + grp_of(k)];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4at(av, i), f4at(wv, i), acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    wait_vmw<(NSLOT - 2) * ND>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (co_ok) {
#pragma unroll
      for (int j = 0; j < MPW; ++j) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const long long m = out_pixel(j, e);
          const float tv = acc[j][e] * sc + sh;
          float v = fmaxf(tv, tv * slope);
          if constexpr (RES) v += rres[j][e];
          if (m >= 0) me::store4(a.y + m * a.y_pitch + co, v, a.store_mode);
        }
      }
    }
    slot = slot + 1 == NSLOT ? 0 : slot + 1;
  }
  wait_vmw<0>();
}

template <int CIN, int WN, int WM, int MBH, int MBW, int TMY, int TMX, int S, int NSLOT, int MINW, int RES>
int launch_w3r(const ConvP& p, hipStream_t stream) {
  constexpr int NW = WN * WM, TH = TMY * MBH, TW = TMX * MBW;
  constexpr int PH = (TH - 1) * S + 3, PWP = (TW - 1) * S + 3;
  constexpr int PIECES = (PH * PWP * CIN * 4 + 1023) / 1024, ND = (PIECES + NW - 1) / NW;
  W3Args a;
  a.x = p.x; a.wgt = p.wgt; a.scale = p.scale; a.shift = p.shift; a.res = p.res; a.y = p.y;
  a.x_pitch = p.x_pitch; a.res_pitch = p.res_pitch; a.y_pitch = p.y_pitch;
  a.n = p.n; a.h = p.h; a.w = p.w; a.ho = p.ho; a.wo = p.wo; a.cout = p.cout; a.act = p.act; a.store_mode = p.store_mode;
  a.tiles_y = (p.ho + TH - 1) / TH;
  a.tiles_x = (p.wo + TW - 1) / TW;
  const long long total = (long long)p.n * a.tiles_y * a.tiles_x;
  ME_REQUIRE(total < (1ll << 31), ME_E_TOOBIG, "me_conv2d_f32: tile 60: too many tiles");
  a.tiles_total = (int)total;
  const int tiles_n = (p.cout + 32 * WN - 1) / (32 * WN);
  const size_t lds = (size_t)NSLOT * ND * NW * 1024;
  ME_REQUIRE(lds <= 160 * 1024, ME_E_TOOBIG, "me_conv2d_f32: tile 60 needs %zu bytes of LDS", lds);
  int per_cu = (int)(160 * 1024 / lds);
  const int by_regs = MINW * 4 / NW;
  if (per_cu > by_regs) per_cu = by_regs;
  if (per_cu < 1) per_cu = 1;
  int per_n = cu_count() * per_cu / tiles_n;
  per_n = per_n / 8 * 8;
  if (per_n < 8) per_n = 8;
  a.grid_m = a.tiles_total < per_n ? a.tiles_total : per_n;
  const dim3 grid((unsigned)(a.grid_m * tiles_n)), block(64 * NW);
  auto kern = conv3x3_ws_f32_kernel<CIN, WN, WM, MBH, MBW, TMY, TMX, S, NSLOT, MINW, RES>;
  static bool attr = false;
  if (!attr) {
    ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  return me::check_launch("conv3x3_ws_f32");
}

template <int CIN, int WN, int WM, int MBH, int MBW, int TMY, int TMX, int S, int NSLOT, int MINW>
int launch_w3(const ConvP& p, hipStream_t stream) {
  return p.res ? launch_w3r<CIN, WN, WM, MBH, MBW, TMY, TMX, S, NSLOT, MINW, 1>(p, stream)
               : launch_w3r<CIN, WN, WM, MBH, MBW, TMY, TMX, S, NSLOT, MINW, 0>(p, stream);
}

int ws_variant() {
  static const int v = [] {
    const char* e = getenv("MILLIEYE_WS32_VARIANT");
    return e ? atoi(e) : 0;
  }();
  return v;
}

}  // namespace

namespace me32 {

bool ws1x1_f32_eligible(const ConvP& p) {
  if (p.ks != 1 || p.stride != 1 || p.pad != 0 || p.ups != 1 || p.x_nchw || p.res || p.act == ME_ACT_SIGMOID) return false;
  if (p.x_pitch % 4 || !me::aligned16(p.x) || !me::aligned16(p.wgt)) return false;
  if ((long long)p.x_pitch * 4 * 128 >= (1ll << 31)) return false;
  const int c = p.cin;
  return c == 64 || c == 128 || c == 256 || c == 384 || c == 512;
}

int launch_ws1x1_f32(const ConvP& p, hipStream_t stream) {
  ME_REQUIRE(ws1x1_f32_eligible(p), ME_E_BADARG,
             "me_conv2d_f32: tile 50 (weight-stationary 1x1) needs a 1x1 / stride 1 layer without residual / upsampling / sigmoid, "
             "16-byte aligned operands and cin in {64, 128, 256, 384, 512}; got cin %d", p.cin);
  const int c = p.cin, o = p.cout, v = ws_variant();
  //                                  CIN WN WM NSLOT MINW        LDS: NSLOT * 32 WM * CIN * 4
  if (c == 64) {
    if (o <= 32) return launch_w1<64, 1, 4, 4, 2>(p, stream);      // 4 x 32 KB, one workgroup per CU: the layer is HBM-bound
    if (o <= 64) return launch_w1<64, 2, 2, 3, 2>(p, stream);      // 3 x 16 KB
    return launch_w1<64, 4, 1, 4, 2>(p, stream);                   // 4 x  8 KB
  }
  if (c == 128) {
    if (o <= 64) return launch_w1<128, 2, 2, 2, 2>(p, stream);     // 2 x 32 KB, two workgroups per CU
    return launch_w1<128, 4, 1, 3, 2>(p, stream);                  // 3 x 16 KB
  }
  if (c == 256) {
    if (v == 1) return launch_w1<256, 4, 1, 3, 1>(p, stream);      // one workgroup per CU, three slots
    if (v == 2) return launch_w1<256, 4, 2, 2, 2>(p, stream);      // 8 waves, 64-row tiles
    return launch_w1<256, 4, 1, 2, 2>(p, stream);                  // 2 x 32 KB, two workgroups per CU
  }
  if (c == 384) return launch_w1<384, 4, 1, 2, 1>(p, stream);      // 2 x 48 KB, 192 weight registers
  return launch_w1<512, 4, 1, 2, 1>(p, stream);                    // 2 x 64 KB, 256 weight registers: a wave per SIMD
}

bool ws3x3_f32_eligible(const ConvP& p) {
  if (p.ks != 3 || p.pad != 1 || (p.stride != 1 && p.stride != 2) || p.ups != 1 || p.x_nchw || p.act == ME_ACT_SIGMOID) return false;
  if (p.x_pitch % 4 || !me::aligned16(p.x) || !me::aligned16(p.wgt)) return false;
  if ((long long)p.h * p.w * p.x_pitch * 4 >= (1ll << 31)) return false;
  if (p.h >= 4096 || p.w >= 4096) return false;
  return p.cin == 32 || p.cin == 64;
}

int launch_ws3x3_f32(const ConvP& p, hipStream_t stream) {
  ME_REQUIRE(ws3x3_f32_eligible(p), ME_E_BADARG,
             "me_conv2d_f32: tile 60 (weight-stationary 3x3) needs a 3x3 / pad 1 / stride 1 or 2 layer with cin 32 or 64, no "
             "upsampling / sigmoid; got cin %d stride %d", p.cin, p.stride);
  const int v = ws_variant();
  //                                                    CIN WN WM MBH MBW TMY TMX S NSLOT MINW
  if (p.cin == 32) {
    if (p.stride == 1) {
      if (v == 1) return launch_w3<32, 2, 2, 2, 16, 4, 1, 1, 3, 1>(p, stream);      //  8 x 16 tiles (two row blocks per wave), one workgroup per CU
      return launch_w3<32, 2, 2, 2, 16, 2, 1, 1, 3, 2>(p, stream);                  //  4 x 16 tiles, patch  6 x 18 (14 KB), two workgroups per CU
    }
    if (v == 1) return launch_w3<32, 2, 2, 2, 16, 2, 1, 2, 2, 1>(p, stream);
    return launch_w3<32, 2, 2, 2, 16, 2, 1, 2, 2, 2>(p, stream);                    //  4 x 16 tiles, patch  9 x 33 (38 KB)
  }
  // cin 64: 288 weight registers - a wave per SIMD, one row block per wave and tile (two blocks in flight spill)
  if (p.stride == 1) return launch_w3<64, 4, 1, 4, 8, 1, 1, 1, 3, 1>(p, stream);     //  4 x  8 tiles, patch  6 x 10 (15 KB)
  return launch_w3<64, 4, 1, 4, 8, 1, 1, 2, 3, 1>(p, stream);                       //  4 x  8 tiles, patch  9 x 17 (39 KB)
}

}  // namespace me32

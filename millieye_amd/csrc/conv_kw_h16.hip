// conv_kw_h16.hip - the 16-bit convolution for the SMALL-BATCH layers (batch 1: 169 .. 2704 output positions), tile ids 40 / 41:
// the K split lives INSIDE the workgroup.
//
// Why: at batch 1 a layer of Darknet-53 has 6 .. 85 tiles of 32 positions; the implicit-GEMM kernels fill the chip by cutting K
// over several workgroups per tile, write fp32 slabs and need a second launch to add them up (36 of the 121 launches of a batch-1
// detector run are such reduce launches: 146 us of kernel time + 39 us of launch boundaries, profiles/r04_b1_bf16_timeline.txt).
// Here one workgroup owns a 32 x 32 (or 32 x 64) output tile and its EIGHT waves each take one eighth of the K steps
// (a K step = 16 channels of one tap = one v_mfma_f32_32x32x16); partial tiles meet in LDS, are added in wave order (fixed:
// deterministic) and go through the fused epilogue (affine, LeakyReLU, residual, 2x upsample, 16-bit or fp32 store) - one launch.
//
// Nothing is shared between the waves of a workgroup before the reduction (each wave multiplies its own K slice of both
// operands), so the operands do not go through LDS at all: every lane loads the 16 bytes of its MFMA fragment straight from
// global memory (A: 8 channels of one input pixel, B: 8 channels of one filter tap; a 64-byte line is used by four consecutive
// K steps of the same wave, so the re-touches are L1 / L2 hits), four K steps ahead of the MFMA that consumes them.
// Both operands of a batch-1 layer live in L2 / Infinity Cache (activations <= 1.4 MB; weights stream once per tile column).
#include <cstdlib>

#include "conv16_common.h"

namespace {

template <int F16, int NT, int DEPTH>
__global__ __launch_bounds__(512) void conv_kw_kernel(Conv16P p) {
  using HT = H16<F16>;
  using frag = typename HT::v8;
  // DEPTH = K steps in flight per wave
  extern __shared__ __attribute__((aligned(16))) float red[];   // [8 waves][NT][16][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, hh = lane >> 5;
  const int tile_n = blockIdx.x % p.tiles_n, tile_m = blockIdx.x / p.tiles_n;
  const int HW = p.ho * p.wo;
  const int cs16 = p.cin >> 4;               // K steps per tap
  const int ksteps = p.ks * p.ks * cs16;
  const int per = (ksteps + 7) >> 3;
  const int s0 = __builtin_amdgcn_readfirstlane(wave * per);
  const int s1 = __builtin_amdgcn_readfirstlane(s0 + per < ksteps ? s0 + per : ksteps);

  // A: this lane's output position -> input pixel of tap (0, 0)
  const int m = tile_m * 32 + r32;
  const bool m_ok = m < p.M;
  int iy0 = 0, ix0 = 0;
  long long xbase = 0;
  if (m_ok) {
    const unsigned um = (unsigned)m;
    const unsigned nimg = udiv_magic16(um, p.hw_m, p.hw_s);
    const unsigned rem = um - nimg * (unsigned)HW;
    const unsigned oy = udiv_magic16(rem, p.wo_m, p.wo_s);
    const unsigned ox = rem - oy * (unsigned)p.wo;
    iy0 = (int)oy * p.stride - p.pad;
    ix0 = (int)ox * p.stride - p.pad;
    xbase = (long long)nimg * p.h * p.w;
  }
  // B: this lane's output channels (rows of the OHWI weights), clamped for the ragged last tile (their results are not stored)
  const unsigned short* wrow[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    int co = tile_n * (32 * NT) + j * 32 + r32;
    if (co >= p.cout) co = p.cout - 1;
    wrow[j] = p.wgt + (long long)co * p.ktot + 8 * hh;
  }

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  const frag zero = {};
  auto load_a = [&](int s) -> frag {
    const int tap = s / cs16, c16 = s - tap * cs16;
    const int dy = tap / p.ks, dx = tap - dy * p.ks;
    const int iy = iy0 + dy, ix = ix0 + dx;
    if (m_ok && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w)
      return *reinterpret_cast<const frag*>(p.x + (xbase + (long long)iy * p.w + ix) * p.x_pitch + c16 * 16 + 8 * hh);
    return zero;
  };
  auto load_b = [&](int s, int j) -> frag { return *reinterpret_cast<const frag*>(wrow[j] + s * 16); };

  frag fa[DEPTH], fb[DEPTH][NT];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const int s = s0 + d;
    if (s < s1) {
      fa[d] = load_a(s);
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[d][j] = load_b(s, j);
    }
  }
  for (int s = s0; s < s1; s += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (s + d < s1) {
        const frag a = fa[d];
        frag b[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = fb[d][j];
        const int sn = s + d + DEPTH;
        if (sn < s1) {   // the slot's next occupant is requested before its MFMA issues
          fa[d] = load_a(sn);
#pragma unroll
          for (int j = 0; j < NT; ++j) fb[d][j] = load_b(sn, j);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = HT::mfma(a, b[j], acc[j]);
      }
    }
  }

  // ---- the eight partial tiles meet in LDS ([wave][j][e][lane]: conflict-free both ways) and are added in wave order
  float* mine = red + (size_t)wave * (NT * 16 * 64);
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) mine[(j * 16 + e) * 64 + lane] = acc[j][e];
  __syncthreads();
  const float slope = p.act == ME_ACT_LEAKY ? 0.1f : 1.0f;
  // wave w finishes registers e = 2 w, 2 w + 1 of every column block: rows (e & 3) + 8 (e >> 2) + 4 hh of the tile
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = 2 * wave + q;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += red[(size_t)w * (NT * 16 * 64) + (j * 16 + e) * 64 + lane];
      const int row = (e & 3) + 8 * (e >> 2) + 4 * hh;
      const int mo = tile_m * 32 + row;
      const int co = tile_n * (32 * NT) + j * 32 + r32;
      if (mo < p.M && co < p.cout) {
        v = v * p.scale[co] + p.shift[co];
        if (p.act == ME_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
        else v = fmaxf(v, v * slope);
        store_out<F16>(p, mo, co, v, HW);
      }
    }
}

template <int F16, int NT, int DEPTH>
int launch_kw_t(Conv16P p, hipStream_t stream) {
  p.tiles_m = (p.M + 31) / 32;
  p.tiles_n = (p.cout + 32 * NT - 1) / (32 * NT);
  const long long blocks = (long long)p.tiles_m * p.tiles_n;
  ME_REQUIRE(blocks < (1ll << 31), ME_E_TOOBIG, "me_conv2d_h16: grid too large");
  const size_t lds = (size_t)8 * NT * 16 * 64 * sizeof(float);
  auto kern = conv_kw_kernel<F16, NT, DEPTH>;
  static bool attr_set = false;
  if (!attr_set) {
    ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, stream, p);
  return me::check_launch("conv_kw_h16");
}

}  // namespace

namespace me16 {

// tile 40: 32 positions x 32 channels per workgroup; 41: 32 x 64.  Any filter size / stride / padding, residual, 2x upsample,
// fp32 output (detection maps), ragged channel counts on the output side; cin % 16 == 0.
bool kw_eligible(const Conv16P& p) {
  return !p.x_nchw && p.cin % 16 == 0 && p.cin >= 16 && p.x_pitch % 8 == 0 && me::aligned16(p.x) && me::aligned16(p.wgt) &&
         (long long)p.ks * p.ks * p.cin < (1ll << 30);
}

int launch_kw(const Conv16P& p, int tile, hipStream_t stream) {
  ME_REQUIRE(kw_eligible(p), ME_E_BADARG, "me_conv2d_h16: tile %d (K split over the waves of a workgroup) needs NHWC input, cin %% 16 == 0, "
                                           "16-byte aligned operands", tile);
  const char* e = getenv("MILLIEYE_KW_DEPTH");
  const int depth = e ? atoi(e) : 4;
  if (tile == 41) {
    if (depth == 8) return p.f16 ? launch_kw_t<1, 2, 8>(p, stream) : launch_kw_t<0, 2, 8>(p, stream);
    return p.f16 ? launch_kw_t<1, 2, 4>(p, stream) : launch_kw_t<0, 2, 4>(p, stream);
  }
  if (depth == 8) return p.f16 ? launch_kw_t<1, 1, 8>(p, stream) : launch_kw_t<0, 1, 8>(p, stream);
  if (depth == 12) return p.f16 ? launch_kw_t<1, 1, 12>(p, stream) : launch_kw_t<0, 1, 12>(p, stream);
  return p.f16 ? launch_kw_t<1, 1, 4>(p, stream) : launch_kw_t<0, 1, 4>(p, stream);
}

}  // namespace me16

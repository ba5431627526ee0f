// pack.hip - one launch that turns a convolution block's parameters into everything the fp32 kernels read (gfx950).
//
// A training step changes every weight, so the packed device copies of millieye_amd/engine.py:ConvWeights are rebuilt once
// per layer and step.  Through torch that was ~25 launches per layer (permute + contiguous, the tiled copy, the 180-degree
// rotation for the data gradient, six fp64 element-wise kernels for the BatchNorm fold, copies into the stable buffers):
// ~1900 launches and ~10 ms of a 42 ms Darknet-53 step (profiles/r03_bench_detector_train_b8_kernel_stats.txt).  Here:
//
//   ohwi   [cout][k][k][cin]            the B operand of the implicit GEMM (me_conv_desc.wgt)
//   tiled  [k*k][cin/16][cout][16]      me_conv_desc.wgt_tiled (cin % 16 == 0), optional
//   rot    [cin][k][k][cout]            weights of the data gradient: 180-degree rotated, channels transposed, optional
//   rott   [k*k][cout/16][cin][16]      tiled copy of rot (cout % 16 == 0), optional
//   scale, shift [cout]                 BatchNorm(eval) + bias folded in double precision exactly like the host code did:
//                                       scale = gamma / sqrt(var + eps), shift = beta - mean * scale (+ bias * scale);
//                                       without BatchNorm scale = 1, shift = bias (or 0)
//
// One workgroup = 32 output channels x 32 input channels x all taps, staged in LDS so that every destination is written in
// runs of >= 64 bytes (128 where the channel counts allow).  Reference: the parameter layout of nn.Conv2d / nn.BatchNorm2d (module3_our_dataset/yolov3/models.py:22-41).
#include <math.h>

#include "common.h"
#include "conv16_common.h"

namespace {

typedef me_pack_desc PackArgs;  // (k = ksize)

constexpr int PO = 32, PC = 32;

// (round 3, second pass: 16 x 64 blocks with an o pitch of 64 * kk floats put the 16 lanes of every rot / rott read on one
// LDS bank and left the rot rows in 64-byte runs: 1.8 ms of a Darknet-53 step for 1.2 GB of traffic.  32 x 32 blocks, odd
// pitch: every destination in runs of >= 128 bytes, every LDS access conflict-free.)
__device__ __forceinline__ void pack_block(const PackArgs& a, int bx, int by, float* s_w) {
  const int kk = a.ksize * a.ksize;
  const int pitch = PC * kk + 1;
  const int o0 = by * PO, c0 = bx * PC;
  const int no = a.cout - o0 < PO ? a.cout - o0 : PO, nc = a.cin - c0 < PC ? a.cin - c0 : PC;
  const int t = threadIdx.x;
  // load: for each o the nc * kk floats of channels c0 .. c0 + nc are contiguous in OIHW; twelve loads in flight per lane
  // (one load per lane and round trip made a 3x3 block a chain of 36 memory latencies: 26 us for any 3x3 layer)
  const int run = nc * kk, total = no * run;
  for (int base = 0; base < total; base += 256 * 12) {
    float v[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int e = base + j * 256 + t;
      const int o = e / run, r = e - o * run;
      v[j] = e < total ? a.w[((long long)(o0 + o) * a.cin + c0) * kk + r] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int e = base + j * 256 + t;
      const int o = e / run, r = e - o * run;
      if (e < total) s_w[o * pitch + r] = v[j];
    }
  }
  __syncthreads();
  const int lane = t & 31, grp = t >> 5;  // 8 groups of 32 lanes
  // (ABI 11) 16-bit copies beside (or instead of) the fp32 ones: the same element, rounded once
  unsigned short* const ohwi16 = reinterpret_cast<unsigned short*>(a.ohwi16);
  unsigned short* const rot16 = reinterpret_cast<unsigned short*>(a.rot16);
  unsigned short* const parity16 = reinterpret_cast<unsigned short*>(a.parity16);
  auto to16 = [&](float v) -> unsigned short { return a.half_type ? H16<1>::to(v) : H16<0>::to(v); };
  // ohwi[o][tap][c]: one (o, tap) row of nc floats per group step
  for (int r = grp; r < no * kk; r += 8) {
    const int o = r / kk, tap = r - o * kk;
    if (lane < nc) {
      const float v = s_w[o * pitch + lane * kk + tap];
      const long long at = ((long long)(o0 + o) * kk + tap) * a.cin + c0 + lane;
      if (a.ohwi) a.ohwi[at] = v;
      if (ohwi16) ohwi16[at] = to16(v);
    }
  }
  if (a.tiled) {  // [tap][chunk][o][16]: cin % 16 == 0 -> nc is a multiple of 16; a group step = 2 o rows of 16 floats
    const int chunks = nc / 16;
    const int l = lane & 15, oh = lane >> 4;
    for (int r = grp; r < kk * chunks * ((no + 1) / 2); r += 8) {
      const int op = r % ((no + 1) / 2), r2 = r / ((no + 1) / 2);
      const int ch = r2 % chunks, tap = r2 / chunks;
      const int o = op * 2 + oh;
      if (o < no)
        a.tiled[(((long long)tap * (a.cin / 16) + c0 / 16 + ch) * a.cout + o0 + o) * 16 + l] =
            s_w[o * pitch + (ch * 16 + l) * kk + tap];
    }
  }
  if (a.rot || rot16) {  // rot[c][kk - 1 - tap][o]: one (c, tap) row of no floats per group step
    for (int r = grp; r < nc * kk; r += 8) {
      const int c = r / kk, tap = r - c * kk;
      if (lane < no) {
        const float v = s_w[lane * pitch + c * kk + tap];
        const long long at = ((long long)(c0 + c) * kk + (kk - 1 - tap)) * a.cout + o0 + lane;
        if (a.rot) a.rot[at] = v;
        if (rot16) rot16[at] = to16(v);
      }
    }
  }
  if (a.rot_tiled) {  // rott[kk - 1 - tap][o / 16][c][o % 16]: cout % 16 == 0 -> no is a multiple of 16; a group step = 2 c rows
    const int l = lane & 15, chh = lane >> 4;
    const int ochunks = no / 16;
    for (int r = grp; r < kk * ochunks * ((nc + 1) / 2); r += 8) {
      const int cp = r % ((nc + 1) / 2), r2 = r / ((nc + 1) / 2);
      const int och = r2 % ochunks, tap = r2 / ochunks;
      const int c = cp * 2 + chh;
      if (c < nc)
        a.rot_tiled[(((long long)(kk - 1 - tap) * (a.cout / 16) + o0 / 16 + och) * a.cin + c0 + c) * 16 + l] =
            s_w[(och * 16 + l) * pitch + c * kk + tap];
    }
  }
  if ((a.parity || parity16) && kk == 9) {  // parity[(cls * cin + c) * 4 + ij][o]: one (cls, ij, c) row of no floats per group step
    for (int r = grp; r < 16 * nc; r += 8) {
      const int c = r % nc, q = r / nc;  // q = cls * 4 + ij
      const int py = q >> 3, px = (q >> 2) & 1, i = (q >> 1) & 1, j = q & 1;
      const int ky = py ? (i ? 0 : 2) : (i ? -1 : 1), kx = px ? (j ? 0 : 2) : (j ? -1 : 1);
      const float v = (ky < 0 || kx < 0 || lane >= no) ? 0.f : s_w[lane * pitch + c * kk + ky * 3 + kx];
      if (lane < no) {
        const long long at = (((long long)(q >> 2) * a.cin + c0 + c) * 4 + (q & 3)) * a.cout + o0 + lane;
        if (a.parity) a.parity[at] = v;
        if (parity16) parity16[at] = to16(v);
      }
    }
  }
  if (bx == 0 && t < no) {  // the fold, in double like ConvWeights.refresh did on the host side of torch
    const int o = o0 + t;
    double sc = 1.0, sh = 0.0;
    if (a.gamma) {
      sc = (double)a.gamma[o] / sqrt((double)a.var[o] + (double)a.eps);
      sh = (double)a.beta[o] - (double)a.mean[o] * sc;
      if (a.bias) sh += (double)a.bias[o] * sc;
    } else if (a.bias) {
      sh = (double)a.bias[o];
    }
    a.scale[o] = (float)sc;
    a.shift[o] = (float)sh;
  }
}

__global__ __launch_bounds__(256) void pack_conv_kernel(PackArgs a) {
  extern __shared__ float s_w[];  // [PO][PC * kk + 1]
  pack_block(a, blockIdx.x, blockIdx.y, s_w);
}

// every layer of a network: block b belongs to the last layer whose first_block <= b (binary search over the table)
__global__ __launch_bounds__(256) void pack_conv_batch_kernel(const PackArgs* __restrict__ table, int count) {
  extern __shared__ float s_w[];
  const int b = blockIdx.x;
  int lo = 0, hi = count - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].first_block <= b) lo = mid; else hi = mid - 1;
  }
  const PackArgs a = table[lo];
  const int rel = b - a.first_block;
  pack_block(a, rel % a.blocks_x, rel / a.blocks_x, s_w);
}

}  // namespace

extern "C" {

int me_pack_conv_f32(const float* w_oihw, int32_t cout, int32_t cin, int32_t ksize, const float* bias, const float* gamma,
                     const float* beta, const float* mean, const float* var, float eps, float* ohwi, float* tiled,
                     float* rot, float* rot_tiled, float* scale, float* shift, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(w_oihw && ohwi && scale && shift, ME_E_NULLPTR, "me_pack_conv_f32: null pointer");
  ME_REQUIRE(cout > 0 && cin > 0 && ksize >= 1 && ksize <= 7, ME_E_BADARG, "me_pack_conv_f32: bad dimensions");
  ME_REQUIRE(!gamma || (beta && mean && var), ME_E_NULLPTR, "me_pack_conv_f32: BatchNorm needs gamma, beta, mean and var");
  ME_REQUIRE(!tiled || cin % 16 == 0, ME_E_BADARG, "me_pack_conv_f32: the tiled copy needs cin %% 16 == 0");
  ME_REQUIRE(!rot_tiled || (cout % 16 == 0 && rot), ME_E_BADARG, "me_pack_conv_f32: the rotated tiled copy needs cout %% 16 == 0");
  PackArgs a;
  a.w = w_oihw; a.bias = bias; a.gamma = gamma; a.beta = beta; a.mean = mean; a.var = var;
  a.ohwi = ohwi; a.tiled = tiled; a.rot = rot; a.rot_tiled = rot_tiled; a.scale = scale; a.shift = shift; a.parity = nullptr;
  a.ohwi16 = a.rot16 = a.parity16 = nullptr; a.half_type = 0; a.reserved0 = 0;
  a.cout = cout; a.cin = cin; a.ksize = ksize; a.eps = eps; a.first_block = 0; a.blocks_x = (cin + PC - 1) / PC;
  const size_t lds = (size_t)PO * (PC * ksize * ksize + 1) * sizeof(float);
  ME_REQUIRE(lds <= 64 * 1024, ME_E_TOOBIG, "me_pack_conv_f32: filter too large");
  const long long gy = (cout + PO - 1) / PO;
  ME_REQUIRE(gy < 65536, ME_E_TOOBIG, "me_pack_conv_f32: too many output channels");
  hipLaunchKernelGGL(pack_conv_kernel, dim3((cin + PC - 1) / PC, (unsigned)gy), dim3(256), lds, stream, a);
  return me::check_launch("pack_conv_kernel");
}

int64_t me_pack_conv_plan(me_pack_desc* d, int32_t count) {
  if (!d || count <= 0) { me::set_error("me_pack_conv_plan: no descriptors"); return ME_E_BADARG; }
  long long blocks = 0;
  for (int i = 0; i < count; ++i) {
    me_pack_desc& a = d[i];
    const bool ok = a.w && (a.ohwi || a.ohwi16) && a.scale && a.shift && a.cout > 0 && a.cin > 0 && a.ksize >= 1 &&
                    (a.half_type == 0 || a.half_type == 1) && (!a.parity16 || a.ksize == 3) &&
                    (size_t)PO * (PC * a.ksize * a.ksize + 1) * sizeof(float) <= 64 * 1024 &&
                    (!a.gamma || (a.beta && a.mean && a.var)) && (!a.tiled || a.cin % 16 == 0) &&
                    (!a.rot_tiled || (a.cout % 16 == 0 && a.rot)) && (!a.parity || a.ksize == 3);
    if (!ok) { me::set_error("me_pack_conv_plan: descriptor %d is not valid (see me_pack_conv_f32)", i); return ME_E_BADARG; }
    a.first_block = (int32_t)blocks;
    a.blocks_x = (a.cin + PC - 1) / PC;
    blocks += (long long)a.blocks_x * ((a.cout + PO - 1) / PO);
    if (blocks >= (1ll << 31)) { me::set_error("me_pack_conv_plan: too many blocks"); return ME_E_TOOBIG; }
  }
  return blocks;
}

int me_pack_conv_batch_f32(const me_pack_desc* descs_device, int32_t count, int64_t total_blocks, int32_t max_ksize,
                           void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(descs_device, ME_E_NULLPTR, "me_pack_conv_batch_f32: null pointer");
  ME_REQUIRE(count > 0 && total_blocks > 0 && total_blocks < (1ll << 31) && max_ksize >= 1, ME_E_BADARG,
             "me_pack_conv_batch_f32: bad arguments");
  const size_t lds = (size_t)PO * (PC * max_ksize * max_ksize + 1) * sizeof(float);
  ME_REQUIRE(lds <= 64 * 1024, ME_E_TOOBIG, "me_pack_conv_batch_f32: filter too large");
  hipLaunchKernelGGL(pack_conv_batch_kernel, dim3((unsigned)total_blocks), dim3(256), lds, stream, descs_device, count);
  return me::check_launch("pack_conv_batch_kernel");
}

}  // extern "C"

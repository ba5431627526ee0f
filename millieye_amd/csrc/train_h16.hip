// train_h16.hip - building blocks of the detector's backward in the 16-bit storage modes (round 5; gfx950).
//
// Reference semantics: autograd through the conv blocks of module3_our_dataset/yolov3/models.py:22-41 with the detector's
// parameters as leaves (models.py:181-267; no reference script trains the detector, train.py:170 freezes it).  The mixed-precision
// step of millieye_amd/detector_train16.py keeps ACTIVATIONS and ACTIVATION GRADIENTS in bfloat16 / IEEE half (one RNE rounding
// where a tensor is stored), accumulates in fp32, and keeps parameters, parameter gradients and every per-channel sum in fp32:
//   forward        me_conv2d_h16 (+ me_add_h16 / me_copy_h16 / me_upsample_h16 / me_maxpool_h16), every module output kept
//   data gradient  me_conv2d_h16 on the 180-degree rotated, transposed weights rounded to the storage type
//   this file      me_affine_act_bwd_h16: dc = dy * act'(y) * scale in the storage type, d beta / d gamma in fp32
//   weight grad    the fp32 kernels (train.hip / wgrad9.hip) on fp32 copies of x and dc - exact products of 16-bit values
//
// me_affine_act_bwd_h16 is the 16-bit twin of me_affine_act_bwd_f32 (train.hip): 8 channels (16 bytes) per lane, 64 channels x one
// row chunk per workgroup (8 channel octets x 32 row lanes), four rows of y and dy in flight per lane, per-channel sums in double
// through LDS in a fixed order, one partial row per workgroup, summed by a second launch in chunk order (deterministic).
#include "conv16_common.h"

namespace {

template <int F16>
__global__ __launch_bounds__(256) void affine_bwd_partial_h16_kernel(const unsigned short* __restrict__ Y, long long ldy,
                                                                     const unsigned short* __restrict__ G, long long ldg, int rows,
                                                                     int C, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, int act, float* p0, float* p1,
                                                                     int chunks, const float* __restrict__ scale,
                                                                     unsigned short* DC, long long lddc) {
  __shared__ double s0s[32][64], s1s[32][64];
  const int q = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c = blockIdx.x * 64 + q * 8;
  const int chunk = blockIdx.y;
  const int per = (rows + chunks - 1) / chunks;
  const int r0 = chunk * per, r1 = (r0 + per < rows) ? r0 + per : rows;
  double s0[8], s1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s0[j] = s1[j] = 0.0;
  if (c < C) {  // C % 8 == 0: the octet is all in or all out
    float be[8], inv_ga[8], scl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float ga = gamma ? gamma[c + j] : 1.f;
      be[j] = beta ? beta[c + j] : 0.f;
      inv_ga[j] = (gamma && ga != 0.f) ? 1.f / ga : 0.f;
      scl[j] = scale ? scale[c + j] : 1.f;
    }
    auto one = [&](int r, uint4 yv, uint4 gv) {
      const unsigned yw[4] = {yv.x, yv.y, yv.z, yv.w}, gw[4] = {gv.x, gv.y, gv.z, gv.w};
      unsigned ow[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        unsigned short o[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int j = 2 * h + e;
          const float y = H16<F16>::from((yw[h] >> (16 * e)) & 0xFFFFu);
          float g = H16<F16>::from((gw[h] >> (16 * e)) & 0xFFFFu);
          float z = y;
          if (act == ME_ACT_LEAKY) {
            g = y > 0.f ? g : 0.1f * g;
            z = y > 0.f ? y : y * 10.f;
          }
          s0[j] += g;
          s1[j] += (double)g * ((z - be[j]) * inv_ga[j]);
          o[e] = H16<F16>::to(scale ? g * scl[j] : g);
        }
        ow[h] = (unsigned)o[0] | ((unsigned)o[1] << 16);
      }
      if (DC) *reinterpret_cast<uint4*>(DC + (long long)r * lddc + c) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    };
    int r = r0 + rl;
    for (; r + 96 < r1; r += 128) {
      uint4 yv[4], gv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        yv[u] = *reinterpret_cast<const uint4*>(Y + (long long)(r + 32 * u) * ldy + c);
        gv[u] = *reinterpret_cast<const uint4*>(G + (long long)(r + 32 * u) * ldg + c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(r + 32 * u, yv[u], gv[u]);
    }
    for (; r < r1; r += 32)
      one(r, *reinterpret_cast<const uint4*>(Y + (long long)r * ldy + c), *reinterpret_cast<const uint4*>(G + (long long)r * ldg + c));
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s0s[rl][q * 8 + j] = s0[j];
    s1s[rl][q * 8 + j] = s1[j];
  }
  __syncthreads();
  const int cl = threadIdx.x;
  if (cl < 64 && blockIdx.x * 64 + cl < C) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      a += s0s[k][cl];
      b += s1s[k][cl];
    }
    p0[(long long)chunk * C + blockIdx.x * 64 + cl] = (float)a;
    p1[(long long)chunk * C + blockIdx.x * 64 + cl] = (float)b;
  }
}

// second level: 64 channels x 16 chunk lanes per workgroup, fixed order (the same tree as train.hip's affine_bwd_reduce_kernel)
__global__ __launch_bounds__(1024) void affine_bwd_reduce_h16_kernel(const float* p0, const float* p1, int C, int chunks, float* dshift,
                                                                     float* dgamma) {
  __shared__ double r0[16][64], r1[16][64];
  const int lc = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lc;
  double s0 = 0.0, s1 = 0.0;
  if (c < C)
    for (int k = part; k < chunks; k += 16) {
      s0 += p0[(long long)k * C + c];
      s1 += p1[(long long)k * C + c];
    }
  r0[part][lc] = s0;
  r1[part][lc] = s1;
  __syncthreads();
  for (int half = 8; half >= 1; half >>= 1) {
    if (part < half) {
      r0[part][lc] += r0[part + half][lc];
      r1[part][lc] += r1[part + half][lc];
    }
    __syncthreads();
  }
  if (part == 0 && c < C) {
    if (dshift) dshift[c] = (float)r0[0][lc];
    if (dgamma) dgamma[c] = (float)r1[0][lc];
  }
}

int chunks_of(int rows) {
  // 8 rounds of 32 row lanes per chunk (measured on the Darknet-53 shapes at batch 8, tools/affine_bench.py 8 bf16: 512 rows per
  // chunk 1.15 ms per step, 256 rows 1.06 - 1.09 ms, 128 rows 1.11 ms: the mid-size layers ran on 84 workgroups)
  int c = rows / 256;
  if (c < 1) c = 1;
  if (c > 1024) c = 1024;
  return c;
}

}  // namespace

extern "C" {

int64_t me_affine_bwd_h16_workspace_bytes(int32_t rows, int32_t channels) {
  return (int64_t)2 * chunks_of(rows) * channels * (int64_t)sizeof(float);
}

int me_affine_act_bwd_h16(const void* y, int64_t ldy, const void* dy, int64_t lddy, int32_t rows, int32_t channels,
                          const float* scale, const float* gamma, const float* beta, int32_t act, void* dc, int64_t lddc,
                          float* dshift, float* dgamma, void* workspace, int32_t half_type, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(y && dy && dc && workspace, ME_E_NULLPTR, "me_affine_act_bwd_h16: null pointer");
  ME_REQUIRE(rows > 0 && channels > 0, ME_E_BADARG, "me_affine_act_bwd_h16: bad dimensions");
  ME_REQUIRE(act == ME_ACT_LINEAR || act == ME_ACT_LEAKY, ME_E_BADARG, "me_affine_act_bwd_h16: activation %d", act);
  ME_REQUIRE(half_type == 0 || half_type == 1, ME_E_BADARG, "me_affine_act_bwd_h16: half_type must be 0 (bf16) or 1 (f16)");
  ME_REQUIRE(channels % 8 == 0 && ldy % 8 == 0 && lddy % 8 == 0 && lddc % 8 == 0 && me::aligned16(y) && me::aligned16(dy) &&
                 me::aligned16(dc), ME_E_ALIGN,
             "me_affine_act_bwd_h16: channels and pitches must be multiples of 8 and the tensors 16-byte aligned");
  const int chunks = chunks_of(rows);
  float* p0 = reinterpret_cast<float*>(workspace);
  float* p1 = p0 + (long long)chunks * channels;
  const dim3 grid((channels + 63) / 64, chunks);
  const unsigned short* yy = reinterpret_cast<const unsigned short*>(y);
  const unsigned short* gg = reinterpret_cast<const unsigned short*>(dy);
  unsigned short* dd = reinterpret_cast<unsigned short*>(dc);
  if (half_type)
    hipLaunchKernelGGL(affine_bwd_partial_h16_kernel<1>, grid, dim3(256), 0, stream, yy, (long long)ldy, gg, (long long)lddy, rows,
                       channels, gamma, beta, act, p0, p1, chunks, scale, dd, (long long)lddc);
  else
    hipLaunchKernelGGL(affine_bwd_partial_h16_kernel<0>, grid, dim3(256), 0, stream, yy, (long long)ldy, gg, (long long)lddy, rows,
                       channels, gamma, beta, act, p0, p1, chunks, scale, dd, (long long)lddc);
  // dshift == dgamma == NULL: the caller adds the partial rows later (me_affine_bwd_h16_sums, possibly on another stream - the
  // detector backward's main stream only needs dc to go on)
  if (dshift || dgamma)
    hipLaunchKernelGGL(affine_bwd_reduce_h16_kernel, dim3((channels + 63) / 64), dim3(1024), 0, stream, p0, p1, channels, chunks,
                       dshift, dgamma);
  return me::check_launch("affine_act_bwd_h16");
}

int me_affine_bwd_h16_sums(const void* workspace, int32_t rows, int32_t channels, float* dshift, float* dgamma, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(workspace && (dshift || dgamma), ME_E_NULLPTR, "me_affine_bwd_h16_sums: null pointer");
  ME_REQUIRE(rows > 0 && channels > 0 && channels % 8 == 0, ME_E_BADARG, "me_affine_bwd_h16_sums: bad dimensions");
  const int chunks = chunks_of(rows);
  const float* p0 = reinterpret_cast<const float*>(workspace);
  const float* p1 = p0 + (long long)chunks * channels;
  hipLaunchKernelGGL(affine_bwd_reduce_h16_kernel, dim3((channels + 63) / 64), dim3(1024), 0, stream, p0, p1, channels, chunks,
                     dshift, dgamma);
  return me::check_launch("affine_bwd_h16_sums");
}

}  // extern "C"

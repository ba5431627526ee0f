// Bottom-up decomposition of the conv main loop (tuning aid): a synthetic "stage" loop with the same instruction
// mix as conv_igemm_dma_f32<128,128> (per wave and stage: 8 ds_read_b128, 32 v_mfma_f32_32x32x2_f32, 1 barrier,
// 4 global_load_lds_dwordx4) where each ingredient can be switched off.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_mix.hip -o tools/mfma_mix.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE bits: 1 = ds_reads feed the MFMAs, 2 = s_barrier per stage, 4 = DMA refills (L2-resident source)
template <int MODE, int NWAVE, int PRIO = 0>
__global__ __launch_bounds__(64 * NWAVE) void mix(const float* __restrict__ src, float* out, int stages) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int STAGE_F = 256 * 16;  // 16 KiB per stage (128+128 rows x 16 floats)
  for (int i = tid; i < 3 * STAGE_F; i += 64 * NWAVE) smem[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int r32 = lane & 31, hh = lane >> 5, sw = (r32 >> 2) & 3;
  const int wr = (wave >> 1) & 1, wc = wave & 1;
  const int a_row = (wr * 64 + r32) * 16, b_row = (128 + wc * 64 + r32) * 16;
  const int off0 = ((0 + hh) ^ sw) * 4, off1 = ((2 + hh) ^ sw) * 4;
  const float* gsrc = src + ((size_t)blockIdx.x * 4096 + tid * 4) % (1 << 20);
  const unsigned wave_lds = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)wave * 1024u);
  float4 fa[2][2], fb[2][2];
  for (int h = 0; h < 2; ++h)
    for (int i = 0; i < 2; ++i) {
      fa[h][i] = make_float4(1e-3f * lane, 2e-3f, 3e-3f, 4e-3f);
      fb[h][i] = make_float4(1e-3f, 2e-3f * lane, 3e-3f, 4e-3f);
    }
  constexpr int LPW = 16 / NWAVE;  // 16 one-KiB groups per stage
  if (PRIO == 2) { if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(2); }
  if (PRIO == 4) { if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(1); if ((blockIdx.x >> 9) & 1) __builtin_amdgcn_s_setprio(2);}
  for (int s = 0; s < stages; ++s) {
    if (PRIO == 3) __builtin_amdgcn_s_setprio(3);
    if (MODE & 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
    if (MODE & 2) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (MODE & 4) {
      const unsigned stage_lds = wave_lds + (unsigned)((s + 2) % 3) * (STAGE_F * 4u);
#pragma unroll
      for (int j = 0; j < LPW; ++j) {
        const float* p = gsrc + j * 1024 + (s & 63) * 16384;
        const unsigned dst = stage_lds + (unsigned)j * (NWAVE * 1024u);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
      }
    }
    if (MODE & 8) {  // the real kernel's per-stage bookkeeping: ~40 VALU (64-bit adds, selects) + ~60 SALU
      unsigned long long v = (unsigned long long)gsrc + lane;
      unsigned sc = (unsigned)s;
#pragma unroll
      for (int r = 0; r < 10; ++r) {
        v += (unsigned long long)(sc * 7u + r) << 2;
        v = (v & 1) ? v + (lane << 3) : v ^ 0x40;
        sc = sc * 5u + 3u; sc ^= sc >> 3; sc = sc / 3u + (sc & 15u);
      }
      asm volatile("" :: "v"(v), "s"(sc));
      gsrc += (v == 0x123456789ull) + (sc == 0xfffffffu);
    }
    if (MODE & 16) {
      unsigned t0 = lane + s, t1 = lane ^ s;
#pragma unroll
      for (int r = 0; r < 16; ++r) { t0 = t0 * 3u + r; t1 = (t1 >> 1) + t0; }   // 32+ 32-bit VALU ops
      asm volatile("" :: "v"(t0), "v"(t1));
    }
    if (MODE & 32) {
      unsigned u0 = (unsigned)s, u1 = (unsigned)stages;
#pragma unroll
      for (int r = 0; r < 32; ++r) { u0 = u0 * 5u + u1; u1 = (u1 >> 1) ^ u0; }  // 64+ SALU ops
      asm volatile("" :: "s"(u0), "s"(u1));
    }
    if (MODE & 64) {
      unsigned long long w0 = (unsigned long long)gsrc + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) w0 += ((unsigned long long)(unsigned)(s + r)) << 2;  // 16 v_lshl_add_u64
      asm volatile("" :: "v"(w0));
    }
    if (MODE & 1) {
      const float* Ab = smem + (s % 3) * STAGE_F + a_row;
      const float* Bb = smem + (s % 3) * STAGE_F + b_row;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int off = h ? off1 : off0;
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[h][i] = *reinterpret_cast<const float4*>(Ab + i * 32 * 16 + off);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[h][j] = *reinterpret_cast<const float4*>(Bb + j * 32 * 16 + off);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
    if (PRIO == 3) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float4 av = fa[h][i], bv = fb[h][j];
            const float a = kk == 0 ? av.x : kk == 1 ? av.y : kk == 2 ? av.z : av.w;
            const float b = kk == 0 ? bv.x : kk == 1 ? bv.y : kk == 2 ? bv.z : bv.w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
          }
    if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float sum = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
  out[(size_t)blockIdx.x * 64 * NWAVE + tid] = sum;
}

template <int MODE, int NWAVE, int PRIO = 0>
void run(const float* src, float* out, int wg_per_cu) {
  const int stages = 2000, blocks = 256 * wg_per_cu;
  auto k = mix<MODE, NWAVE, PRIO>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * NWAVE), 48 * 1024, 0, src, out, stages);
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * NWAVE), 48 * 1024, 0, src, out, stages);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double flops = 10.0 * blocks * NWAVE * stages * 32.0 * 2 * 32 * 32 * 2;
  printf("prio=%d mode=%d (%s%s%s) waves/WG=%d WG/CU=%d: %.1f TFLOP/s\n", PRIO, MODE, MODE & 1 ? "lds " : "", MODE & 2 ? "barrier " : "",
         MODE & 4 ? "dma" : "", NWAVE, wg_per_cu, flops / ms / 1e9);
}

int main() {
  float *src, *out;
  hipMalloc(&src, (size_t)(1 << 21) * sizeof(float));
  hipMemset(src, 0, (size_t)(1 << 21) * sizeof(float));
  hipMalloc(&out, (size_t)256 * 4 * 512 * sizeof(float));
  run<3, 4, 0>(src, out, 2);
  run<3 + 16, 4, 0>(src, out, 2);
  run<3 + 32, 4, 0>(src, out, 2);
  run<3 + 64, 4, 0>(src, out, 2);
  run<3, 8, 0>(src, out, 2);
  run<3 + 16, 8, 0>(src, out, 2);
  run<3 + 32, 8, 0>(src, out, 2);
  run<3 + 64, 8, 0>(src, out, 2);
  return 0;
}

// Does hipExtAnyOrderLaunch let a second kernel on the SAME stream start before the first one has finished (gfx950 / ROCm 7.2)?
// hip_ext.h says the flag "is not supported on AMD GFX9xx boards" for hipExtModuleLaunchKernel; this probe measures it.
// Kernel A: a few workgroups spin for ~200 us, the rest exit at once.  Kernel B (launched right behind A): every workgroup stamps its
// start on the chip-wide s_memrealtime clock (100 MHz).  If B's first start is earlier than A's last end, the launches overlapped.
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o tools/anyorder_probe.bin && tools/anyorder_probe.bin
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void spin_kernel(unsigned long long* stamps, int spinners, unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if ((int)blockIdx.x < spinners) {
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  }
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
}

__global__ void stamp_kernel(unsigned long long* stamps) {
  if (threadIdx.x == 0) stamps[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}

int main() {
  const int na = 512, nb = 512;
  unsigned long long *sa, *sb;
  hipMalloc(&sa, 2 * na * sizeof(unsigned long long));
  hipMalloc(&sb, nb * sizeof(unsigned long long));
  hipStream_t st;
  hipStreamCreate(&st);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(spin_kernel, dim3(na), dim3(256), 0, st, sa, 16, 20000ull);  // 16 workgroups spin 200 us
      if (mode == 0)
        hipLaunchKernelGGL(stamp_kernel, dim3(nb), dim3(256), 0, st, sb);
      else
        hipExtLaunchKernelGGL(stamp_kernel, dim3(nb), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, sb);
      hipStreamSynchronize(st);
    }
    unsigned long long ha[2 * na], hb[nb];
    hipMemcpy(ha, sa, sizeof(ha), hipMemcpyDeviceToHost);
    hipMemcpy(hb, sb, sizeof(hb), hipMemcpyDeviceToHost);
    unsigned long long a0 = ~0ull, a1 = 0, b0 = ~0ull, b1 = 0;
    for (int i = 0; i < na; ++i) {
      if (ha[2 * i] < a0) a0 = ha[2 * i];
      if (ha[2 * i + 1] > a1) a1 = ha[2 * i + 1];
    }
    for (int i = 0; i < nb; ++i) {
      if (hb[i] < b0) b0 = hb[i];
      if (hb[i] > b1) b1 = hb[i];
    }
    printf("%s: A runs %.1f us; B first start %.1f us after A's start (%s A's end), B last start %.1f us\n",
           mode ? "hipExtAnyOrderLaunch" : "plain launch       ", (a1 - a0) / 100.0, ((long long)b0 - (long long)a0) / 100.0,
           b0 < a1 ? "BEFORE" : "after", ((long long)b1 - (long long)a0) / 100.0);
  }
  return 0;
}

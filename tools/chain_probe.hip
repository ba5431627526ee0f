// chain_probe.hip - what would ONE launch for a chain of dependent conv layers buy at batch 1?  
//
// Model of the batch-1 detector: L dependent "layers" of T tiles each.  A tile = one 256-thread workgroup that
//   (1) reads its weight slab (WKB KB, private to the (layer, tile), from a buffer that fits the Infinity Cache),
//   (2) reads the outputs of ND tiles of the previous layer (4 KB each: the halo / channel dependency of a conv tile),
//   (3) runs an fp32 MFMA chain of CH instructions per wave (the K loop; 64 cycles each),
//   (4) writes its own 4 KB output.
// Variant A: one launch per layer (what the engine does: the stream orders the layers).
// Variant C: B with write-through (sc0 sc1) payload stores and loads on both sides instead of the release / acquire fences.
// Variant B: ONE launch; workgroups draw (layer, tile) tickets in order from a device-scope counter (placement-independent:
//   every dependency of a ticket was drawn before it, by a workgroup that is resident or done), prefetch the weight slab into
//   registers BEFORE waiting (it depends on nothing), then wait for the ND producer tiles' arrival flags of layer - 1
//   (one lane: relaxed agent-scope poll + s_sleep, bounded), agent-scope acquire + __syncthreads, plain loads; after the
//   output stores: __syncthreads, lane-0 agent-scope release, asm s_waitcnt vmcnt(0), relaxed flag store
//   (MI355X_MICROARCH.md "Correctness boundaries" / cdna_hip_programming.md guideline 16).
// Both variants check every word they consume (the value chain is a running checksum), so a stale read shows up as an error.
// Output: microseconds per layer for A and B over a sweep of T / WKB / CH.
//
// build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -o /tmp/chain_probe tools/chain_probe.hip && /tmp/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Args {
  const float4* w;     // [L][T][wq] float4 (wq = WKB * 64)
  float* act;          // [L + 1][T][1024] floats (4 KB per tile); layer 0 = input
  int* flags;          // [L + 1][T] arrival flags (B only); epoch-valued so that no reset is needed between runs
  unsigned* head;      // ticket counter (B only)
  int* err;
  int L, T, wq, nd, ch, epoch;
};

__device__ __forceinline__ float tile_body(const Args& a, int layer, int tile, const float4* wv, int nw, float seed_in) {
  // MFMA chain (K loop stand-in): CH dependent v_mfma_f32_32x32x2_f32 per wave
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float wsum = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (i < nw) wsum += wv[i].x + wv[i].y + wv[i].z + wv[i].w;
  const float av = seed_in * 1e-3f + 1.0f, bv = wsum * 1e-6f + 1.0f;
  for (int i = 0; i < a.ch; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
  return acc[0] * 1e-9f;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 load_wt(const float* p) {   // sc0 sc1: served past this CU's L1 and this XCD's possibly stale L2 lines
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void store_wt(float* p, f32x4 v) {  // write-through: visible to every XCD once vmcnt drains
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// one tile of one layer; MODE 0 = the launch boundary ordered the producers (variant A), 1 = arrival flags with agent-scope
// release / acquire fences around plain payload accesses (variant B), 2 = arrival flags with write-through (sc0 sc1) payload
// stores and loads on both sides and no fences (variant C: the guide's other valid form, cheaper per hop)
template <int MODE>
__device__ __forceinline__ void run_tile(const Args& a, int layer, int tile) {
  constexpr bool WAIT = MODE != 0;
  const int tid = threadIdx.x;
  // (1) weights: 16 float4 per lane in flight per round, up to 4 rounds kept in registers (64 KB / 256 lanes = 16 float4)
  float4 wv[16];
  const float4* wp = a.w + ((long long)layer * a.T + tile) * a.wq;
  const int nw = a.wq / 256;  // float4 per lane (<= 16)
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (i < nw) wv[i] = wp[i * 256 + tid];
  // (2) dependencies
  const float* in = a.act + (long long)layer * a.T * 1024;  // outputs of layer - 1 live in slot `layer`
  if (WAIT && layer > 0) {
    if (tid == 0) {
      for (int d = 0; d < a.nd; ++d) {
        const int src = (tile + d) % a.T;
        const int* f = a.flags + (long long)layer * a.T + src;
        int spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1 << 13)) {  // bounded (~10 ms); report instead of hanging the box
            atomicExch(a.err, 2);
            break;
          }
        }
      }
      if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  float s = 0.f;
  for (int d = 0; d < a.nd; ++d) {
    const int src = (tile + d) % a.T;
    float4 v;
    if (MODE == 2) {
      const f32x4 t = load_wt(in + (long long)src * 1024 + tid * 4);
      v = make_float4(t[0], t[1], t[2], t[3]);
    } else {
      v = reinterpret_cast<const float4*>(in + (long long)src * 1024)[tid];
    }
    // every word of a producer tile carries (layer * 1000 + src + epoch): anything else is a stale or torn read
    const float want = (float)(layer * 1000 + src + a.epoch);
    if (v.x != want || v.y != want || v.z != want || v.w != want) atomicExch(a.err, 1);
    s += v.x;
  }
  // (3) compute
  const float r = tile_body(a, layer, tile, wv, nw, s);
  // (4) output: the value the consumers check (+ r * 0 keeps the chain alive without changing it)
  float* out = a.act + (long long)(layer + 1) * a.T * 1024 + (long long)tile * 1024;
  const float val = (float)((layer + 1) * 1000 + tile + a.epoch) + r * 0.f;
  if (MODE == 2) {
    f32x4 t = {val, val, val, val};
    store_wt(out + tid * 4, t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave drains its own write-through stores in front of the barrier
  } else {
    reinterpret_cast<float4*>(out)[tid] = make_float4(val, val, val, val);
  }
  if (WAIT) {
    __syncthreads();
    if (tid == 0) {
      if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(a.flags + (long long)(layer + 1) * a.T + tile, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ __launch_bounds__(256) void layer_kernel(Args a, int layer) { run_tile<0>(a, layer, blockIdx.x); }

template <int MODE>
__global__ __launch_bounds__(256) void chain_kernel(Args a) {
  __shared__ unsigned s_ticket;
  const unsigned total = (unsigned)a.L * (unsigned)a.T;
  for (;;) {
    // (the barrier in FRONT of the one-lane ticket draw closes the one-lane publish block at the end of run_tile before the
    //  back edge: without it hipcc merged the two one-lane regions across the back edge and the lanes of wave 0 met the
    //  barriers in different iterations - the first version of this probe hung)
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = atomicAdd(a.head, 1u);
    __syncthreads();
    const unsigned t = __builtin_amdgcn_readfirstlane(s_ticket);
    if (t >= total) break;
    run_tile<MODE>(a, (int)(t / a.T), (int)(t % a.T));
  }
}

__global__ void init_input(Args a) {  // layer-0 input tiles carry the values layer 0's consumers expect
  const int tile = blockIdx.x;
  const float val = (float)(0 * 1000 + tile + a.epoch);
  reinterpret_cast<float4*>(a.act + (long long)tile * 1024)[threadIdx.x] = make_float4(val, val, val, val);
  if (threadIdx.x == 0) a.flags[tile] = a.epoch;
}

int main() {
  const int L = 48;
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipStream_t st;
  CK(hipStreamCreate(&st));
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  printf("chain probe: %d layers, %d CUs; us per layer, A = one launch per layer, B = one launch with ticket queue + arrival flags (release / acquire fences), C = the same with write-through payload instead of fences\n", L, cus);
  printf("%6s %6s %6s %4s | %9s %9s %9s %6s %6s | %s\n", "tiles", "W KB", "mfma", "deps", "A us/lyr", "B us/lyr", "C us/lyr", "B / A", "C / A", "errors");
  const int Ts[] = {48, 176, 256, 704};
  const int WKBs[] = {16, 64};
  const int CHs[] = {32, 128};   // 32 MFMAs = 0.85 us, 128 = 3.4 us of matrix pipe per wave at 2.4 GHz
  for (int T : Ts)
    for (int WKB : WKBs)
      for (int CH : CHs) {
        if ((long long)T * WKB > 16384) continue;  // (a layer's weights beyond 16 MB: not a batch-1 layer of this network)
        Args a = {};
        a.L = L; a.T = T; a.wq = WKB * 64; a.nd = 3; a.ch = CH;
        float4* w; float* act; int* flags; unsigned* head; int* err;
        const size_t wbytes = (size_t)L * T * a.wq * sizeof(float4);
        CK(hipMalloc(&w, wbytes));
        CK(hipMemset(w, 0, wbytes));
        CK(hipMalloc(&act, (size_t)(L + 1) * T * 4096));
        CK(hipMalloc(&flags, (size_t)(L + 1) * T * sizeof(int)));
        CK(hipMemset(flags, 0, (size_t)(L + 1) * T * sizeof(int)));
        CK(hipMalloc(&head, 4));
        CK(hipMalloc(&err, 4));
        CK(hipMemset(err, 0, 4));
        a.w = w; a.act = act; a.flags = flags; a.head = head; a.err = err;
        // resident workgroups of the persistent launch: what fits (occupancy query), at most one per tile slot needed
        int per_cu = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chain_kernel<1>, 256, 0));
        if (per_cu > 4) per_cu = 4;
        int grid_b = per_cu * cus;
        if (grid_b > 2 * T) grid_b = 2 * T;   // two layers' worth of tiles in flight is all the chain can use
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        float ms_a = 1e30f, ms_b = 1e30f, ms_c = 1e30f;
        int epoch = 0;
        const bool dbg = getenv("CHAIN_DEBUG") != nullptr;
        for (int rep = 0; rep < 12; ++rep) {
          if (dbg) printf("  rep %d A...\n", rep);
          // A
          a.epoch = ++epoch;
          hipLaunchKernelGGL(init_input, dim3(T), dim3(256), 0, st, a);
          CK(hipEventRecord(e0, st));
          for (int l = 0; l < L; ++l) hipLaunchKernelGGL(layer_kernel, dim3(T), dim3(256), 0, st, a, l);
          CK(hipEventRecord(e1, st));
          CK(hipStreamSynchronize(st));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (rep >= 2 && ms < ms_a) ms_a = ms;
          if (dbg) printf("  rep %d A %.3f ms, B...\n", rep, ms);
          // B
          a.epoch = ++epoch;
          hipLaunchKernelGGL(init_input, dim3(T), dim3(256), 0, st, a);
          CK(hipMemsetAsync(head, 0, 4, st));
          CK(hipEventRecord(e0, st));
          hipLaunchKernelGGL(chain_kernel<1>, dim3(grid_b), dim3(256), 0, st, a);
          CK(hipEventRecord(e1, st));
          CK(hipStreamSynchronize(st));
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (rep >= 2 && ms < ms_b) ms_b = ms;
          // C
          a.epoch = ++epoch;
          hipLaunchKernelGGL(init_input, dim3(T), dim3(256), 0, st, a);
          CK(hipMemsetAsync(head, 0, 4, st));
          CK(hipEventRecord(e0, st));
          hipLaunchKernelGGL(chain_kernel<2>, dim3(grid_b), dim3(256), 0, st, a);
          CK(hipEventRecord(e1, st));
          CK(hipStreamSynchronize(st));
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (rep >= 2 && ms < ms_c) ms_c = ms;
          if (dbg) printf("  rep %d B %.3f ms\n", rep, ms);
        }
        int herr = 0;
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("%6d %6d %6d %4d | %9.2f %9.2f %9.2f %6.2f %6.2f | %s (grid %d)\n", T, WKB, CH, a.nd, ms_a * 1e3 / L, ms_b * 1e3 / L, ms_c * 1e3 / L,
               ms_b / ms_a, ms_c / ms_a,
               herr == 0 ? "none" : herr == 1 ? "STALE/TORN DATA" : "SPIN LIMIT", grid_b);
        CK(hipFree(w)); CK(hipFree(act)); CK(hipFree(flags)); CK(hipFree(head)); CK(hipFree(err));
      }
  return 0;
}

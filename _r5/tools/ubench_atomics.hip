// Microbenchmark (not part of the product): cost of the scattered hash-update atomics of
// expand_kernel at agent scope (memory-side on the 8-XCD MI355X) vs workgroup scope (executed in
// the issuing XCD's L2), plus random 16-byte gathers.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t rng(uint32_t &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

template <int SCOPE, bool RET>
__global__ void k_min64(u64 *tab, int slots_per_chan, int n_chan, int per_thread) {
  const int c = blockIdx.x % n_chan;
  u64 *t = tab + (size_t)c * slots_per_chan;
  uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  u64 acc = 0;
  for (int i = 0; i < per_thread; ++i) {
    uint32_t r = rng(s);
    u64 v = ((u64)r << 32) | r;
    if (RET) acc += __hip_atomic_fetch_min(&t[r % slots_per_chan], v, __ATOMIC_RELAXED, SCOPE);
    else (void)__hip_atomic_fetch_min(&t[r % slots_per_chan], v, __ATOMIC_RELAXED, SCOPE);
  }
  if (acc == 0x1234567) tab[0] = acc;
}
template <int SCOPE>
__global__ void k_cas32(int *tab, int slots_per_chan, int n_chan, int per_thread) {
  const int c = blockIdx.x % n_chan;
  int *t = tab + (size_t)c * slots_per_chan;
  uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 777u;
  int acc = 0;
  for (int i = 0; i < per_thread; ++i) {
    uint32_t r = rng(s);
    int exp = -1;
    __hip_atomic_compare_exchange_strong(&t[r % slots_per_chan], &exp, (int)(r >> 1), __ATOMIC_RELAXED, __ATOMIC_RELAXED, SCOPE);
    acc += exp;
  }
  if (acc == 0x1234567) tab[0] = acc;
}
__global__ void k_gather16(const int4 *arr, size_t n, int per_thread, int *sink) {
  uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 99u;
  int acc = 0;
  for (int i = 0; i < per_thread; ++i) {
    uint32_t r = rng(s);
    int4 v = arr[(size_t)r % n];
    acc += v.x + v.w;
  }
  if (acc == 0x1234567) *sink = acc;
}
__global__ void k_gather_rows(const int4 *arr, size_t n, int per_thread, int *sink) {  // 4 consecutive 16B per "state"
  uint32_t s = ((blockIdx.x * blockDim.x + threadIdx.x) >> 2) * 2654435761u + 99u;
  const int sub = threadIdx.x & 3;
  int acc = 0;
  for (int i = 0; i < per_thread; ++i) {
    uint32_t r = rng(s);
    int4 v = arr[((size_t)r % (n / 4)) * 4 + sub];
    acc += v.x + v.w;
  }
  if (acc == 0x1234567) *sink = acc;
}

template <class F>
float timeit(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main() {
  const int n_chan = 128, slots = 65536;           // 128 channels x 64k slots
  u64 *tab64; int *tab32; int4 *arcs; int *sink;
  CK(hipMalloc(&tab64, (size_t)n_chan * slots * 8)); CK(hipMemset(tab64, 0xFF, (size_t)n_chan * slots * 8));
  CK(hipMalloc(&tab32, (size_t)n_chan * slots * 4)); CK(hipMemset(tab32, 0xFF, (size_t)n_chan * slots * 4));
  const size_t n_arcs = 10u << 20; CK(hipMalloc(&arcs, n_arcs * 16)); CK(hipMemset(arcs, 1, n_arcs * 16)); CK(hipMalloc(&sink, 4));
  const int blocks = 2048, threads = 256, per = 4;  // 2.1M ops per launch ~ one frame of the batch
  const double ops = (double)blocks * threads * per;
  float t;
  t = timeit([&] { hipLaunchKernelGGL((k_min64<__HIP_MEMORY_SCOPE_AGENT, false>), blocks, threads, 0, 0, tab64, slots, n_chan, per); });
  printf("min64 agent   noret : %.1f us  %.1f Gops/s\n", t * 1e3, ops / t / 1e6);
  t = timeit([&] { hipLaunchKernelGGL((k_min64<__HIP_MEMORY_SCOPE_AGENT, true>), blocks, threads, 0, 0, tab64, slots, n_chan, per); });
  printf("min64 agent   ret   : %.1f us  %.1f Gops/s\n", t * 1e3, ops / t / 1e6);
  t = timeit([&] { hipLaunchKernelGGL((k_min64<__HIP_MEMORY_SCOPE_WORKGROUP, false>), blocks, threads, 0, 0, tab64, slots, n_chan, per); });
  printf("min64 wg      noret : %.1f us  %.1f Gops/s\n", t * 1e3, ops / t / 1e6);
  t = timeit([&] { hipLaunchKernelGGL((k_min64<__HIP_MEMORY_SCOPE_WORKGROUP, true>), blocks, threads, 0, 0, tab64, slots, n_chan, per); });
  printf("min64 wg      ret   : %.1f us  %.1f Gops/s\n", t * 1e3, ops / t / 1e6);
  t = timeit([&] { hipLaunchKernelGGL((k_cas32<__HIP_MEMORY_SCOPE_AGENT>), blocks, threads, 0, 0, tab32, slots, n_chan, per); });
  printf("cas32 agent         : %.1f us  %.1f Gops/s\n", t * 1e3, ops / t / 1e6);
  t = timeit([&] { hipLaunchKernelGGL((k_cas32<__HIP_MEMORY_SCOPE_WORKGROUP>), blocks, threads, 0, 0, tab32, slots, n_chan, per); });
  printf("cas32 wg            : %.1f us  %.1f Gops/s\n", t * 1e3, ops / t / 1e6);
  t = timeit([&] { hipLaunchKernelGGL(k_gather16, blocks, threads, 0, 0, arcs, n_arcs, per, sink); });
  printf("gather 16B random   : %.1f us  %.1f Gops/s  %.1f GB/s useful\n", t * 1e3, ops / t / 1e6, ops * 16 / t / 1e6);
  t = timeit([&] { hipLaunchKernelGGL(k_gather_rows, blocks, threads, 0, 0, arcs, n_arcs, per, sink); });
  printf("gather 64B rows     : %.1f us  %.1f Gops/s  %.1f GB/s useful\n", t * 1e3, ops / t / 1e6, ops * 16 / t / 1e6);
  t = timeit([&] { hipLaunchKernelGGL(k_gather16, blocks, threads, 0, 0, arcs, n_arcs, 16, sink); });
  printf("gather 16B x16/thr  : %.1f us  %.1f Gops/s  %.1f GB/s useful\n", t * 1e3, ops * 4 / t / 1e6, ops * 4 * 16 / t / 1e6);
  t = timeit([&] { hipLaunchKernelGGL((k_min64<__HIP_MEMORY_SCOPE_WORKGROUP, false>), blocks, threads, 0, 0, tab64, slots, n_chan, 16); });
  printf("min64 wg noret x16  : %.1f us  %.1f Gops/s\n", t * 1e3, ops * 4 / t / 1e6);
  t = timeit([&] { hipLaunchKernelGGL((k_min64<__HIP_MEMORY_SCOPE_AGENT, false>), blocks, threads, 0, 0, tab64, slots, n_chan, 16); });
  printf("min64 agent noret x16: %.1f us  %.1f Gops/s\n", t * 1e3, ops * 4 / t / 1e6);
  // how fast can ONE workgroup (one compute unit) issue returning atomics?  (the closure kernel's
  // heaviest channel does ~3.9 k of them from its single 1024-thread workgroup)
  for (int wgs = 1; wgs <= 8; wgs *= 2) {
    t = timeit([&] { hipLaunchKernelGGL((k_min64<__HIP_MEMORY_SCOPE_AGENT, true>), wgs, 1024, 0, 0, tab64, slots, 1, 4 / wgs > 0 ? 4 / wgs : 1); });
    printf("min64 agent ret, %d x 1024 threads, %d per thread (4096 atomics total): %.1f us\n", wgs, 4 / wgs > 0 ? 4 / wgs : 1, t * 1e3);
  }
  t = timeit([&] { hipLaunchKernelGGL(k_gather16, 1, 1024, 0, 0, arcs, n_arcs, 4, sink); });
  printf("gather 16B random, 1 x 1024 threads x 4: %.1f us\n", t * 1e3);
  return 0;
}

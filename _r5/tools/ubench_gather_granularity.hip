// What does a random row gather cost on MI355X: is the memory-side fetch granule 64 B or 128 B?
// Groups of G consecutive lanes read one contiguous chunk of G x 16 B at a random position of a 4 GiB
// table (far beyond L2 and the 256 MiB Infinity Cache); positions aligned to `align` bytes.  Reported:
// chunks/s and useful GB/s.  If 64-B chunks (G=4, align 64) run at ~2x the chunk rate of 128-B chunks
// (G=8, align 128) the granule is 64 B; if they run at the same chunk rate it is 128 B.  The straddling
// variants (align 16) show what an unaligned CSR row costs.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gather_granularity tools/ubench_gather_granularity.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ unsigned rng(unsigned &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
template <int G>
__global__ void __launch_bounds__(256) gather(const int4 *a, unsigned n_units, int unit_slots, int per, int *sink) {
  const unsigned gid = (blockIdx.x * blockDim.x + threadIdx.x) / G, lane = threadIdx.x % G;
  unsigned s = gid * 2654435761u + 99u;
  int acc = 0;
  for (int i = 0; i < per; i += 4) {
    size_t p0 = (size_t)(rng(s) % n_units) * unit_slots + lane, p1 = (size_t)(rng(s) % n_units) * unit_slots + lane;
    size_t p2 = (size_t)(rng(s) % n_units) * unit_slots + lane, p3 = (size_t)(rng(s) % n_units) * unit_slots + lane;
    int4 v0 = a[p0], v1 = a[p1], v2 = a[p2], v3 = a[p3];
    acc += v0.x + v1.y + v2.z + v3.w;
  }
  if (acc == 0x1234567) *sink = acc;
}
template <int G>
static int run(const char *name, const int4 *buf, size_t slots, int align_bytes, int *sink) {
  const int unit_slots = align_bytes / 16, per = 32, blocks = 16384;
  const unsigned n_units = (unsigned)((slots - 64) / unit_slots);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(gather<G>, blocks, 256, 0, 0, buf, n_units, unit_slots, per, sink);
  CK(hipEventRecord(e0));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(gather<G>, blocks, 256, 0, 0, buf, n_units, unit_slots, per, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  const double chunks = (double)blocks * 256 / G * per;
  printf("%-34s chunk %4d B align %4d B: %8.2f G chunks/s  %8.1f GB/s useful  (%.3f ms)\n", name, G * 16, align_bytes,
         chunks / ms / 1e6, chunks * G * 16 / ms / 1e6, ms);
  return 0;
}
int main() {
  const size_t bytes = 4ull << 30;
  int4 *buf; int *sink; CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMemset(buf, 1, bytes));
  CK(hipDeviceSynchronize());
  const size_t slots = bytes / 16;
  run<1>("16 B", buf, slots, 16, sink);
  run<2>("32 B aligned", buf, slots, 32, sink);
  run<4>("64 B aligned", buf, slots, 64, sink);
  run<4>("64 B at any 16 B (straddles)", buf, slots, 16, sink);
  run<8>("128 B aligned", buf, slots, 128, sink);
  run<8>("128 B at 64 B (half straddle)", buf, slots, 64, sink);
  run<8>("128 B at any 16 B", buf, slots, 16, sink);
  run<16>("256 B aligned", buf, slots, 256, sink);
  run<16>("256 B at 128", buf, slots, 128, sink);
  run<16>("256 B at any 16 B", buf, slots, 16, sink);
  return 0;
}

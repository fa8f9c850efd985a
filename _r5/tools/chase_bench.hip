// Development micro-benchmark (not part of the product): latency of a dependent-load chain that walks BACKWARDS through a large
// buffer at a fixed stride -- what best_path_kernel's backpointer walk does (one 16-byte token per hop, a frame's tokens apart).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/chase tools/chase_bench.hip && /tmp/chase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void chase(const int4 *buf, size_t region, int hops, long long *out_cycles, int *sink) {
  const int4 *p = buf + (size_t)blockIdx.x * region;
  if (threadIdx.x != 0) return;
  int t = (int)(region - 1);
  const long long t0 = wall_clock64();
  for (int h = 0; h < hops && t >= 0; ++h) t = p[t].z;
  const long long t1 = wall_clock64();
  out_cycles[blockIdx.x] = t1 - t0;
  sink[blockIdx.x] = t;
}
__global__ void fill(int4 *buf, size_t region, int stride_tok) {
  int4 *p = buf + (size_t)blockIdx.x * region;
  for (size_t i = threadIdx.x; i < region; i += blockDim.x) p[i] = make_int4(0, 0, (int)((long long)i - stride_tok), 0);
}
int main() {
  const int B = 128, hops = 300;
  const size_t region = (size_t)45 << 20 >> 4;   // 45 MB of 16-byte tokens per channel
  int4 *buf; long long *cyc; int *sink;
  CK(hipMalloc(&buf, (size_t)B * region * 16));
  CK(hipMalloc(&cyc, B * 8)); CK(hipMalloc(&sink, B * 4));
  int wc = 0; CK(hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0));
  for (int stride_bytes : {68 * 1024, 17 * 1024, 4 * 1024, 1024, 128}) {
    const int st = stride_bytes / 16;
    hipLaunchKernelGGL(fill, dim3(B), dim3(1024), 0, 0, buf, region, st);
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 1; ++rep) {   // (one run: cold -- a second one finds its 300 lines in L2 / the Infinity Cache: 114 ns per hop)
      hipLaunchKernelGGL(chase, dim3(B), dim3(64), 0, 0, buf, region, hops, cyc, sink);
      CK(hipDeviceSynchronize());
    }
    std::vector<long long> h(B);
    CK(hipMemcpy(h.data(), cyc, B * 8, hipMemcpyDeviceToHost));
    double s = 0; for (auto v : h) s += v;
    printf("stride %6d B: %.1f ns per hop (mean over %d chains of %d hops, wall clock %d kHz)\n", stride_bytes, s / B / hops * 1e6 / wc, B, hops, wc);
  }
  return 0;
}

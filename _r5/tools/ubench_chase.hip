// Single-lane dependent-load latency on gfx950: ds_read (LDS, address space known), flat_load into LDS (generic pointer),
// flat/global load of an L2-resident table, and the issue cost of a dependent ALU chain.  Decides where the determinizer's
// tables should live (wfst_determinize.hip).   hipcc --offload-arch=gfx950 -O3 -o ubench_chase tools/ubench_chase.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void chase(int *g, int n_g, int iters, long long *out, int *sink, int **gen) {
  __shared__ int lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (i * 1237 + 1) & 8191;
  __syncthreads();
  if (threadIdx.x != 0) return;
  int p = 0;
  long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) p = lds[p];
  long long t1 = wall_clock64();
  int *q = *gen ? lds : g;   // generic pointer the compiler cannot resolve
  int a = p & 8191;
  for (int i = 0; i < iters; ++i) a = q[a];
  long long t2 = wall_clock64();
  int b = a & 8191;
  for (int i = 0; i < iters; ++i) b = g[b];          // 32 KB table: L2 (and L1) resident
  long long t3 = wall_clock64();
  int c = b;
  for (int i = 0; i < iters; ++i) c = g[(c * 977 + i) & (n_g - 1)];   // 64 MB table: mostly misses
  long long t4 = wall_clock64();
  int d = c;
  for (int i = 0; i < iters; ++i) d = d * 1664525 + 1013904223;
  long long t5 = wall_clock64();
  int e = d & 8191;
  for (int i = 0; i < iters; ++i) e = __builtin_nontemporal_load(&g[e]);
  long long t6 = wall_clock64();
  out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; out[4] = t5 - t4; out[5] = t6 - t5;
  *sink = e + d;
}
int main() {
  const int n = 16 << 20;
  std::vector<int> h(n);
  for (int i = 0; i < n; ++i) h[i] = i < 8192 ? (i * 1237 + 1) & 8191 : (int)((i * 2654435761u) & (n - 1));
  int *g, *sink; long long *out; int **gen;
  hipMalloc(&g, n * 4); hipMalloc(&sink, 4); hipMalloc(&out, 64); hipMalloc(&gen, 8);
  hipMemcpy(g, h.data(), n * 4, hipMemcpyHostToDevice);
  int *one = (int *)1; hipMemcpy(gen, &one, 8, hipMemcpyHostToDevice);
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    chase<<<1, 256>>>(g, n, iters, out, sink, gen);
    hipDeviceSynchronize();
  }
  long long o[6]; hipMemcpy(o, out, 48, hipMemcpyDeviceToHost);
  const char *name[6] = {"ds_read chase", "flat->LDS chase", "global chase 32KB", "global chase 64MB", "dependent mul-add", "global nt chase 32KB"};
  for (int i = 0; i < 6; ++i) printf("%-24s %.1f ns per step\n", name[i], o[i] * 10.0 / iters);
  return 0;
}

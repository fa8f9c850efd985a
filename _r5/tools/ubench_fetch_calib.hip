// Calibration of rocprofv3's FETCH_SIZE on gfx950 for THIS path's access patterns (the guide:
// "calibrate on a known byte count in your own access pattern before trusting an absolute").
// Known request counts: coalesced 16 B/lane stream, random 16 B gathers, random 8 B gathers, all
// over a 1 GiB buffer (far larger than L2 and the Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ unsigned rng(unsigned &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
__global__ void calib_stream16(const int4 *a, size_t n, int *sink) {
  int acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += a[i].x;
  if (acc == 0x1234567) *sink = acc;
}
__global__ void calib_gather16(const int4 *a, size_t n, int per, int *sink) {
  unsigned s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 99u; int acc = 0;
  for (int i = 0; i < per; ++i) acc += a[(size_t)rng(s) % n].x;
  if (acc == 0x1234567) *sink = acc;
}
__global__ void calib_gather8(const int2 *a, size_t n, int per, int *sink) {
  unsigned s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 7u; int acc = 0;
  for (int i = 0; i < per; ++i) acc += a[(size_t)rng(s) % n].x;
  if (acc == 0x1234567) *sink = acc;
}
int main() {
  const size_t bytes = 1ull << 30;  // 1 GiB
  int4 *buf; int *sink; CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMemset(buf, 1, bytes));
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(calib_stream16, 4096, 256, 0, 0, buf, bytes / 16, sink);                  // 1 GiB streamed
  hipLaunchKernelGGL(calib_gather16, 4096, 256, 0, 0, buf, bytes / 16, 8, sink);               // 8.39M random 16 B
  hipLaunchKernelGGL(calib_gather8, 4096, 256, 0, 0, (const int2 *)buf, bytes / 8, 8, sink);   // 8.39M random 8 B
  CK(hipDeviceSynchronize());
  printf("known: stream16 %zu bytes; gather16 %d requests; gather8 %d requests\n", bytes, 4096 * 256 * 8, 4096 * 256 * 8);
  return 0;
}

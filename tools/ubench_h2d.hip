// Development harness: host -> device rates of page-locked rows as the channel pool hands them over -- 128 copies of 300 KB (one per
// channel), ONE 2-D copy of 128 x 300 KB (pitch = a channel's whole buffer), one contiguous copy of the same bytes.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench_h2d tools/ubench_h2d.hip && tools/ubench_h2d
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
int main() {
  const size_t C = 128, chunk = 25 * 3000 * 4, pitch = 300 * 3000 * 4;
  char *h = nullptr, *d = nullptr;
  CK(hipHostMalloc((void **)&h, C * pitch, hipHostMallocDefault));
  CK(hipMalloc((void **)&d, C * pitch));
  memset(h, 1, C * pitch);
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto timeit = [&](const char *what, auto &&f) {
    for (int w = 0; w < 2; ++w) { f(0); CK(hipStreamSynchronize(s)); }
    double best = 1e9, host = 0;
    for (int r = 0; r < 5; ++r) {
      const auto t0 = std::chrono::steady_clock::now();
      f(r + 1);
      const auto t1 = std::chrono::steady_clock::now();
      CK(hipStreamSynchronize(s));
      const auto t2 = std::chrono::steady_clock::now();
      const double ms = std::chrono::duration<double, std::milli>(t2 - t0).count();
      if (ms < best) { best = ms; host = std::chrono::duration<double, std::milli>(t1 - t0).count(); }
    }
    printf("%-44s %7.3f ms (%5.1f GB/s), host side of the calls %.3f ms\n", what, best, C * chunk / best / 1e6, host);
  };
  timeit("128 x hipMemcpyAsync(300 KB)", [&](int k) { for (size_t c = 0; c < C; ++c) CK(hipMemcpyAsync(d + c * pitch + (k % 12) * chunk, h + c * pitch + (k % 12) * chunk, chunk, hipMemcpyHostToDevice, s)); });
  timeit("1 x hipMemcpy2DAsync(128 rows x 300 KB)", [&](int k) { CK(hipMemcpy2DAsync(d + (k % 12) * chunk, pitch, h + (k % 12) * chunk, pitch, chunk, C, hipMemcpyHostToDevice, s)); });
  timeit("1 x hipMemcpyAsync(38.4 MB contiguous)", [&](int k) { CK(hipMemcpyAsync(d, h + (k % 2) * C * chunk, C * chunk, hipMemcpyHostToDevice, s)); });
  timeit("8 x hipMemcpy2DAsync(16 rows x 300 KB)", [&](int k) { for (size_t b = 0; b < 8; ++b) CK(hipMemcpy2DAsync(d + b * 16 * pitch + (k % 12) * chunk, pitch, h + b * 16 * pitch + (k % 12) * chunk, pitch, chunk, 16, hipMemcpyHostToDevice, s)); });
  timeit("128 x hipMemcpyAsync(576 KB) [bytes x 1.92]", [&](int k) { for (size_t c = 0; c < C; ++c) CK(hipMemcpyAsync(d + c * pitch, h + c * pitch, 48 * 3000 * 4, hipMemcpyHostToDevice, s)); });
  return 0;
}

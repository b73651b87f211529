// tools/ubench/pcie_duplex.hip -- is the host link full duplex, and for which kind of copy?  (developer micro-benchmark)
// H2D by hipMemcpyAsync (SDMA) alone, D2H by hipMemcpyAsync alone, D2H by a KERNEL storing into pinned host memory alone, H2D by a
// kernel loading from pinned host memory alone, and the pairs at once on two streams.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/pcie_duplex.hip -o tools/ubench/pcie_duplex
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void copy_kernel(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t n = 256u << 20;
  void *h1, *h2, *d1, *d2;
  CK(hipHostMalloc(&h1, n, hipHostMallocDefault)); CK(hipHostMalloc(&h2, n, hipHostMallocDefault));
  CK(hipMalloc(&d1, n)); CK(hipMalloc(&d2, n));
  memset(h1, 1, n); memset(h2, 2, n);
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  auto h2d_dma = [&] { return hipMemcpyAsync(d1, h1, n, hipMemcpyHostToDevice, s1); };
  auto d2h_dma = [&] { return hipMemcpyAsync(h2, d2, n, hipMemcpyDeviceToHost, s2); };
  auto d2h_krn = [&](int wgs) { hipLaunchKernelGGL(copy_kernel, dim3(wgs), dim3(256), 0, s2, (const uint4*)d2, (uint4*)h2, n / 16); return hipGetLastError(); };
  auto h2d_krn = [&](int wgs) { hipLaunchKernelGGL(copy_kernel, dim3(wgs), dim3(256), 0, s1, (const uint4*)h1, (uint4*)d1, n / 16); return hipGetLastError(); };
  auto run = [&](const char* name, auto f) {
    double best = 1e9;
    for (int r = 0; r < 5; r++) {
      (void)hipDeviceSynchronize();
      const double t = now();
      f();
      (void)hipDeviceSynchronize();
      best = best < now() - t ? best : now() - t;
    }
    printf("%-44s %.1f ms  %.1f GB/s per direction\n", name, 1e3 * best, n / best / 1e9);
  };
  run("H2D dma", [&] { (void)h2d_dma(); });
  run("D2H dma", [&] { (void)d2h_dma(); });
  run("H2D dma + D2H dma at once", [&] { (void)h2d_dma(); (void)d2h_dma(); });
  for (int wgs : {64, 256, 1024}) {
    char nm[64];
    snprintf(nm, sizeof nm, "D2H kernel (%d wgs)", wgs); run(nm, [&] { (void)d2h_krn(wgs); });
    snprintf(nm, sizeof nm, "H2D kernel (%d wgs)", wgs); run(nm, [&] { (void)h2d_krn(wgs); });
    snprintf(nm, sizeof nm, "H2D dma + D2H kernel (%d wgs) at once", wgs); run(nm, [&] { (void)h2d_dma(); (void)d2h_krn(wgs); });
    snprintf(nm, sizeof nm, "H2D kernel + D2H dma (%d wgs) at once", wgs); run(nm, [&] { (void)h2d_krn(wgs); (void)d2h_dma(); });
    snprintf(nm, sizeof nm, "H2D kernel + D2H kernel (%d wgs) at once", wgs); run(nm, [&] { (void)h2d_krn(wgs); (void)d2h_krn(wgs); });
  }
  return 0;
}

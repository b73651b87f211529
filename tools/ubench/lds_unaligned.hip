// LDS access cost by width and alignment (gfx950): one wavefront per workgroup, W workgroups per CU.
// Each lane reads / writes `width` bytes at (lane * stride + mis) inside its own 1 KB-region pattern like the decoder's rings:
// 16 groups of 4 lanes, group g at g * 832 bytes, lane l of a group at l * width (+ mis).  Reports shader cycles per instruction for
// a dependent read chain (latency) and for back-to-back independent accesses (throughput).
// Build: hipcc --offload-arch=gfx950 -O3 lds_unaligned.hip -o lds_unaligned
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ uint64_t clk() { uint64_t t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

template <int W> struct Vec { uint32_t w[W / 4]; };

template <int W, bool WRITE>
__global__ __launch_bounds__(64) void k_tp(int iters, uint32_t mis, uint32_t gstride, uint64_t* out, uint32_t* sink) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[16 * 1024 + 64];
  for (int i = threadIdx.x; i < (16 * 1024 + 64) / 4; i += 64) ((uint32_t*)lds)[i] = i * 2654435761u;
  __syncthreads();
  const uint32_t g = threadIdx.x >> 2, l = threadIdx.x & 3;
  uint32_t a = g * gstride + l * W + mis;
  Vec<W> acc = {};
  uint64_t t0 = clk();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t ad = (a + k * 64u) & 1023u;
      if (WRITE) { Vec<W> v; for (int q = 0; q < W / 4; q++) v.w[q] = acc.w[q] + k; __builtin_memcpy(lds + g * gstride + ((ad) & 511u) + 0, &v, W); }
      else { Vec<W> v; __builtin_memcpy(&v, lds + ad + (g * gstride & ~1023u), W); for (int q = 0; q < W / 4; q++) acc.w[q] += v.w[q]; }
    }
    a += 7u * W;
  }
  uint64_t t1 = clk();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  uint32_t s = 0; for (int q = 0; q < W / 4; q++) s += acc.w[q];
  if (WRITE) s += ((uint32_t*)lds)[threadIdx.x];
  sink[blockIdx.x * 64 + threadIdx.x] = s;
}
// dependent chain: the loaded value decides the next address
template <int W>
__global__ __launch_bounds__(64) void k_lat(int iters, uint32_t mis, uint64_t* out, uint32_t* sink) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[16 * 1024 + 64];
  for (int i = threadIdx.x; i < (16 * 1024 + 64) / 4; i += 64) ((uint32_t*)lds)[i] = i * 2654435761u;
  __syncthreads();
  uint32_t a = threadIdx.x * 16u + mis;
  uint64_t t0 = clk();
  for (int i = 0; i < iters; i++) {
    Vec<W> v; __builtin_memcpy(&v, lds + a, W);
    a = ((v.w[0] >> 8) & 0x3FF0u) + mis;
  }
  uint64_t t1 = clk();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = a;
}
int main() {
  uint64_t* out; CHK(hipMalloc(&out, 8 * 4096));
  uint32_t* sink; CHK(hipMalloc(&sink, 4 * 64 * 4096));
  std::vector<uint64_t> ho(4096);
  auto report = [&](const char* name, int grid, double per) {
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(ho.data(), out, 8 * grid, hipMemcpyDeviceToHost));
    double s = 0; for (int i = 0; i < grid; i++) s += (double)ho[i];
    printf("%-64s grid %5d: %8.1f ticks per instruction\n", name, grid, s / grid / per);
  };
  const int IT = 500;
  for (int grid : {256, 1024, 3072}) {
    for (uint32_t mis : {0u, 4u, 1u}) {
      char nm[128];
#define TP(W, WR) snprintf(nm, sizeof nm, "%s b%d, misalignment %u, groups 832 B apart, back to back", WR ? "write" : "read", W * 8, mis); \
      hipLaunchKernelGGL((k_tp<W, WR>), dim3(grid), dim3(64), 0, 0, IT, mis, 832u, out, sink); report(nm, grid, IT * 8.0);
      TP(16, false) TP(8, false) TP(4, false) TP(16, true) TP(8, true) TP(4, true)
      snprintf(nm, sizeof nm, "read b128 chain, misalignment %u", mis);
      hipLaunchKernelGGL((k_lat<16>), dim3(grid), dim3(64), 0, 0, IT * 4, mis, out, sink); report(nm, grid, IT * 4.0);
      snprintf(nm, sizeof nm, "read b64 chain, misalignment %u", mis);
      hipLaunchKernelGGL((k_lat<8>), dim3(grid), dim3(64), 0, 0, IT * 4, mis, out, sink); report(nm, grid, IT * 4.0);
    }
  }
  return 0;
}

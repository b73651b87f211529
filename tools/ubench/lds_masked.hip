// Cost of a MISALIGNED LDS access when only some lanes take part (exec-masked), of ds_read2_b32 pairs and of ds_bpermute, under
// load (W wavefronts per CU).  Companion of lds_unaligned.hip.  Build: hipcc --offload-arch=gfx950 -O3 lds_masked.hip -o lds_masked
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ uint64_t clk() { uint64_t t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
template <int W> struct Vec { uint32_t w[W / 4]; };
// every `every`-th lane does a misaligned access of W bytes (read or write), 8 per iteration
template <int W, bool WRITE>
__global__ __launch_bounds__(64) void k_masked(int iters, uint32_t mis, uint32_t every, uint64_t* out, uint32_t* sink) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[16 * 1024 + 64];
  for (int i = threadIdx.x; i < (16 * 1024 + 64) / 4; i += 64) ((uint32_t*)lds)[i] = i * 2654435761u;
  __syncthreads();
  uint32_t a = threadIdx.x * 208u + mis;
  Vec<W> acc = {};
  const bool on = (threadIdx.x % every) == 0;
  uint64_t t0 = clk();
  for (int i = 0; i < iters; i++) {
    if (on) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t ad = (a + k * 64u) & 16383u;
        if (WRITE) { Vec<W> v; for (int q = 0; q < W / 4; q++) v.w[q] = acc.w[q] + k; __builtin_memcpy(lds + ad, &v, W); }
        else { Vec<W> v; __builtin_memcpy(&v, lds + ad, W); for (int q = 0; q < W / 4; q++) acc.w[q] += v.w[q]; }
      }
    }
    a += 7u * 16u;
  }
  uint64_t t1 = clk();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  uint32_t s = 0; for (int q = 0; q < W / 4; q++) s += acc.w[q];
  if (WRITE) s += ((uint32_t*)lds)[threadIdx.x];
  sink[blockIdx.x * 64 + threadIdx.x] = s;
}
// aligned pairs: two dwords at a and a + 4 (the compiler fuses them into ds_read2_b32) + one v_alignbyte
__global__ __launch_bounds__(64) void k_read2(int iters, uint32_t shift, uint64_t* out, uint32_t* sink) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[4096 + 16];
  for (int i = threadIdx.x; i < 4096 + 16; i += 64) lds[i] = i * 2654435761u;
  __syncthreads();
  uint32_t a = threadIdx.x * 13u, acc = 0;
  uint64_t t0 = clk();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t ad = (a + k * 97u) & 4095u;
      const uint32_t d0 = lds[ad], d1 = lds[ad + 1];
      acc += __builtin_amdgcn_alignbyte(d1, d0, shift);
    }
    a += 31u;
  }
  uint64_t t1 = clk();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(64) void k_bperm(int iters, uint64_t* out, uint32_t* sink) {
  uint32_t v = threadIdx.x * 2654435761u, acc = 0;
  uint64_t t0 = clk();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) acc += (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x + k * 5 + i) & 63u) << 2), (int)(v + k));
  }
  uint64_t t1 = clk();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = acc;
}
int main() {
  uint64_t* out; CHK(hipMalloc(&out, 8 * 4096));
  uint32_t* sink; CHK(hipMalloc(&sink, 4 * 64 * 4096));
  std::vector<uint64_t> ho(4096);
  auto report = [&](const char* name, int grid, double per) {
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(ho.data(), out, 8 * grid, hipMemcpyDeviceToHost));
    double s = 0; for (int i = 0; i < grid; i++) s += (double)ho[i];
    printf("%-72s grid %5d: %8.1f ticks per instruction\n", name, grid, s / grid / per);
  };
  const int IT = 500;
  for (int grid : {256, 1024, 3072, 8192}) {
    char nm[160];
    for (uint32_t every : {1u, 4u, 16u, 64u}) {
#define MK(W, WR) snprintf(nm, sizeof nm, "%s b%d misaligned by 1, every %u-th lane takes part", WR ? "write" : "read", W * 8, every); \
      hipLaunchKernelGGL((k_masked<W, WR>), dim3(grid), dim3(64), 0, 0, IT, 1u, every, out, sink); report(nm, grid, IT * 8.0);
      MK(4, false) MK(4, true) MK(16, false) MK(16, true)
    }
    snprintf(nm, sizeof nm, "read b32 ALIGNED, all lanes (reference)");
    hipLaunchKernelGGL((k_masked<4, false>), dim3(grid), dim3(64), 0, 0, IT, 0u, 1u, out, sink); report(nm, grid, IT * 8.0);
    snprintf(nm, sizeof nm, "write b32 ALIGNED, all lanes (reference)");
    hipLaunchKernelGGL((k_masked<4, true>), dim3(grid), dim3(64), 0, 0, IT, 0u, 1u, out, sink); report(nm, grid, IT * 8.0);
    snprintf(nm, sizeof nm, "two aligned dwords (ds_read2_b32) + v_alignbyte");
    hipLaunchKernelGGL(k_read2, dim3(grid), dim3(64), 0, 0, IT, 1u, out, sink); report(nm, grid, IT * 8.0);
    snprintf(nm, sizeof nm, "ds_bpermute_b32");
    hipLaunchKernelGGL(k_bperm, dim3(grid), dim3(64), 0, 0, IT, out, sink); report(nm, grid, IT * 8.0);
  }
  return 0;
}

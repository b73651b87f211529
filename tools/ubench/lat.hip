// Latency micro-benchmarks for the fast-compress chain (one wavefront per workgroup, 1 or more workgroups per CU):
// dependent chains of the operations a step is made of.  Build: hipcc --offload-arch=gfx950 -O3 lat.hip -o lat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t clk() { uint64_t t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

// mode 0: dependent global dword loads, every lane its own chain inside a `span`-byte region (unaligned 1-byte-stride windows like the compressor)
// the next address depends on the loaded value (wave-uniform via readfirstlane, like the compressor's scalar chain)
__global__ void k_gload(const uint8_t* buf, uint32_t span, int iters, uint64_t* out, uint32_t* sink, int uniform) {
  uint32_t pos = (blockIdx.x * 7919u) % (span - 512u);
  uint64_t t0 = clk();
  uint32_t acc = 0;
  for (int i = 0; i < iters; i++) {
    uint32_t v;
    __builtin_memcpy(&v, buf + pos + threadIdx.x, 4);
    acc += v;
    uint32_t nx = uniform ? __builtin_amdgcn_readfirstlane(v) : v;
    pos = (pos + 64u + (nx & 63u)) % (span - 512u);
    if (!uniform) pos = __builtin_amdgcn_readfirstlane(pos);
  }
  uint64_t t1 = clk();
  if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; }
  sink[blockIdx.x * 64 + threadIdx.x] = acc;
}
// random far jumps (each step a different 128-byte line far away): L2 / MALL / HBM latency by span
__global__ void k_gload_far(const uint8_t* buf, uint64_t span, int iters, uint64_t* out, uint32_t* sink) {
  uint64_t pos = ((uint64_t)blockIdx.x * 2654435761ull) % (span - 512u);
  uint64_t t0 = clk();
  uint32_t acc = 0;
  for (int i = 0; i < iters; i++) {
    uint32_t v;
    __builtin_memcpy(&v, buf + pos + threadIdx.x * 4u, 4);
    acc += v;
    uint32_t nx = __builtin_amdgcn_readfirstlane(v);
    pos = (pos * 6364136223846793005ull + 1442695040888963407ull + nx) % (span - 512u);
  }
  uint64_t t1 = clk();
  if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; }
  sink[blockIdx.x * 64 + threadIdx.x] = acc;
}
// LDS dependent reads / atomics with return
__global__ void k_lds(int iters, uint64_t* out, uint32_t* sink, int atomic) {
  __shared__ uint32_t t[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) t[i] = i * 2654435761u;
  __syncthreads();
  uint32_t h = threadIdx.x * 37u;
  uint64_t t0 = clk();
  for (int i = 0; i < iters; i++) {
    uint32_t v = atomic ? atomicMax(&t[h & 8191u], h) : t[h & 8191u];
    h = v * 2654435761u >> 7;
  }
  uint64_t t1 = clk();
  if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; }
  sink[blockIdx.x * 64 + threadIdx.x] = h;
}
// the scalar side: ballot -> ff1 -> readlane -> vector op -> ballot ...
__global__ void k_scalar(int iters, uint64_t* out, uint32_t* sink) {
  uint32_t v = threadIdx.x * 2654435761u;
  uint64_t t0 = clk();
  for (int i = 0; i < iters; i++) {
    uint64_t m = __builtin_amdgcn_ballot_w64((v & 0x10000u) != 0) | (1ull << 63);
    int k = __builtin_ctzll(m);
    uint32_t s = __builtin_amdgcn_readlane(v, k);
    v = v * 2654435761u + s;
  }
  uint64_t t1 = clk();
  if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; }
  sink[blockIdx.x * 64 + threadIdx.x] = v;
}
// dependent ds_bpermute chain
__global__ void k_bperm(int iters, uint64_t* out, uint32_t* sink) {
  uint32_t v = threadIdx.x * 2654435761u;
  uint64_t t0 = clk();
  for (int i = 0; i < iters; i++) {
    v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((v & 63u) << 2), (int)(v + i)) + 0x9E3779B9u;
  }
  uint64_t t1 = clk();
  if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; }
  sink[blockIdx.x * 64 + threadIdx.x] = v;
}
// VGPR-indexing mode: on / v_mov / off with a scalar index that depends on the previous value
__global__ __attribute__((amdgpu_num_vgpr(96))) void k_gpridx(int iters, uint64_t* out, uint32_t* sink, int dep) {
  uint32_t v = threadIdx.x * 2654435761u;
  for (int s = 0; s < 32; s++) asm volatile("s_set_gpr_idx_on %1, gpr_idx(DST)\n\tv_mov_b32 v96, %0\n\ts_set_gpr_idx_off" ::"v"(v + s), "s"(s) : "m0", "v255");
  uint32_t idx = 3;
  uint64_t t0 = clk();
  for (int i = 0; i < iters; i++) {
    uint32_t x;
    asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\tv_mov_b32 %0, v96\n\ts_set_gpr_idx_off" : "=v"(x) : "s"(idx) : "m0");
    v += x;
    if (dep) idx = __builtin_amdgcn_readfirstlane(v) & 31u; else idx = (idx + 7u) & 31u;
  }
  uint64_t t1 = clk();
  if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; }
  sink[blockIdx.x * 64 + threadIdx.x] = v;
}
__global__ void k_empty(int iters, uint64_t* out) {
  uint64_t t0 = clk();
  uint64_t t1 = clk();
  if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; }
}

int main() {
  const uint64_t BIG = 8ull << 30;
  uint8_t* buf; CHK(hipMalloc(&buf, BIG));
  CHK(hipMemset(buf, 0x5a, BIG));
  std::vector<uint32_t> h(1 << 20);
  for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 3);
  for (uint64_t o = 0; o < BIG; o += (4u << 20)) CHK(hipMemcpy(buf + o, h.data(), 4u << 20, hipMemcpyHostToDevice));
  uint64_t* out; CHK(hipMalloc(&out, 8 * 4096));
  uint32_t* sink; CHK(hipMalloc(&sink, 4 * 64 * 4096));
  std::vector<uint64_t> ho(4096);
  auto report = [&](const char* name, int grid, int iters) {
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(ho.data(), out, 8 * grid, hipMemcpyDeviceToHost));
    double s = 0; for (int i = 0; i < grid; i++) s += (double)ho[i];
    printf("%-44s grid %5d: %8.1f ticks per iteration\n", name, grid, s / grid / iters);
  };
  const int IT = 2000;
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k_empty, dim3(256), dim3(64), 0, 0, 1, out); report("empty (two clk() back to back)", 256, 1);
    for (int grid : {256, 1280, 2560}) {
      hipLaunchKernelGGL(k_gload, dim3(grid), dim3(64), 0, 0, buf, 32768u, IT, out, sink, 1); report("global dword, window walk in 32 KB (L1/L2)", grid, IT);
    }
    for (int grid : {256, 1280}) {
      for (uint64_t span : {1ull << 20, 16ull << 20, 128ull << 20, 8ull << 30}) {
        char nm[96]; snprintf(nm, sizeof nm, "global dword, random line in %llu MB", (unsigned long long)(span >> 20));
        hipLaunchKernelGGL(k_gload_far, dim3(grid), dim3(64), 0, 0, buf, span, IT, out, sink); report(nm, grid, IT);
      }
    }
    hipLaunchKernelGGL(k_lds, dim3(256), dim3(64), 0, 0, IT, out, sink, 0); report("LDS read -> mul -> LDS read", 256, IT);
    hipLaunchKernelGGL(k_lds, dim3(256), dim3(64), 0, 0, IT, out, sink, 1); report("LDS atomic max rtn -> mul -> atomic", 256, IT);
    hipLaunchKernelGGL(k_lds, dim3(1024), dim3(64), 0, 0, IT, out, sink, 0); report("LDS read chain, 4 waves per CU", 1024, IT);
    hipLaunchKernelGGL(k_scalar, dim3(256), dim3(64), 0, 0, IT, out, sink); report("ballot -> ff1 -> readlane -> mul/add", 256, IT);
    hipLaunchKernelGGL(k_bperm, dim3(256), dim3(64), 0, 0, IT, out, sink); report("ds_bpermute -> add -> ds_bpermute", 256, IT);
    hipLaunchKernelGGL(k_bperm, dim3(1280), dim3(64), 0, 0, IT, out, sink); report("ds_bpermute chain, 5 waves per CU", 1280, IT);
    hipLaunchKernelGGL(k_gpridx, dim3(256), dim3(64), 0, 0, IT, out, sink, 1); report("gpr_idx on/mov/off, index depends on value", 256, IT);
    hipLaunchKernelGGL(k_gpridx, dim3(256), dim3(64), 0, 0, IT, out, sink, 0); report("gpr_idx on/mov/off, independent index", 256, IT);
  }
  return 0;
}

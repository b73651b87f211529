// Random-line gather throughput of one MI355X: how many 128-byte (or 64-byte) lines per second the memory system delivers to
// wavefronts that each touch RANDOM lines of a footprint -- the access pattern of the LZ4 decoder's match sources and of the HC
// parser's chain walks (both verdicts hang on this ceiling).  Independent loads (addresses from a per-lane LCG, not from
// loaded data), `unroll` of them in flight per lane.
//   gather <footprint MiB> <waves per CU> <lanes per line: 1|4|8|16> <bytes per lane: 4|16> <nt: 0|1> [iters]
// lanes per line = how many adjacent lanes read adjacent pieces of the SAME line (1: 64 different lines per wave instruction;
// 4 x 16 B = one 64-byte half line per 4 lanes, the decoder's shape; 8 x 16 B = a whole 128-byte line per 8 lanes).
// Prints lines/s (distinct line requests per wave instruction = 64 / lanes per line), useful GB/s, and the clock.
// Build: hipcc --offload-arch=gfx950 -O3 gather.hip -o gather
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int LPL, int BPL, bool NT>
__global__ __launch_bounds__(256) void gather_kernel(const uint8_t* buf, uint64_t lines, int iters, uint32_t* sink) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t grp = lane / LPL, sub = lane % LPL;
  uint64_t s = ((uint64_t)(blockIdx.x * 256u + threadIdx.x - sub) * 0x9E3779B97F4A7C15ull) | 1ull;   // same stream for the lanes of a group
  uint32_t acc = 0;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      const uint64_t line = (s >> 20) % lines;
      const uint8_t* p = buf + line * 128u + (uint64_t)(sub * BPL) % 128u;
      if (BPL == 4) {
        uint32_t v;
        if (NT) v = __builtin_nontemporal_load((const uint32_t*)p); else v = *(const uint32_t*)p;
        acc += v;
      } else {
        uint4 v;
        if (NT) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); u4 t = __builtin_nontemporal_load((const u4*)p); v.x = t.x; v.y = t.y; v.z = t.z; v.w = t.w; }
        else v = *(const uint4*)p;
        acc += v.x ^ v.y ^ v.z ^ v.w;
      }
    }
  }
  (void)grp;
  sink[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int LPL, int BPL>
void run(bool nt, const uint8_t* buf, uint64_t lines, int waves_per_cu, int iters, uint32_t* sink, int cus) {
  const int wgs = cus * waves_per_cu / 4;   // 4 wavefronts per workgroup
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  for (int rep = 0; rep < 3; rep++) {
    CHK(hipEventRecord(a));
    if (nt) hipLaunchKernelGGL((gather_kernel<LPL, BPL, true>), dim3(wgs), dim3(256), 0, 0, buf, lines, iters, sink);
    else hipLaunchKernelGGL((gather_kernel<LPL, BPL, false>), dim3(wgs), dim3(256), 0, 0, buf, lines, iters, sink);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    const double instr = (double)wgs * 4.0 * iters * 8.0;                 // wave load instructions
    const double lreq = instr * (64.0 / LPL);                             // distinct line requests
    const double bytes = instr * 64.0 * BPL;
    if (rep == 2)
      printf("footprint %6.0f MiB  waves/CU %2d  lanes/line %2d  bytes/lane %2d  nt %d : %7.2f G lines/s  %8.1f GB/s useful  %8.1f GB/s as 128-B lines  (%.2f ms)\n",
             lines * 128.0 / 1048576.0, waves_per_cu, LPL, BPL, (int)nt, lreq / ms / 1e6, bytes / ms / 1e6, lreq * 128.0 / ms / 1e6, ms);
  }
}

int main(int argc, char** argv) {
  if (argc < 6) { printf("usage: gather <footprint MiB> <waves per CU> <lanes per line> <bytes per lane> <nt> [iters]\n"); return 2; }
  const uint64_t mib = strtoull(argv[1], 0, 10);
  const int wpc = atoi(argv[2]), lpl = atoi(argv[3]), bpl = atoi(argv[4]), nt = atoi(argv[5]);
  const int iters = argc > 6 ? atoi(argv[6]) : 64;
  hipDeviceProp_t pr; CHK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  uint8_t* buf; uint32_t* sink;
  CHK(hipMalloc(&buf, mib << 20)); CHK(hipMemset(buf, 1, mib << 20));
  CHK(hipMalloc(&sink, (size_t)cus * 32 * 64 * 4 + 1024));
  const uint64_t lines = (mib << 20) / 128u;
#define CASE(L, B) if (lpl == L && bpl == B) run<L, B>(nt != 0, buf, lines, wpc, iters, sink, cus)
  CASE(1, 4); CASE(1, 16); CASE(4, 16); CASE(8, 16); CASE(16, 4); CASE(4, 4); CASE(8, 4);
  CHK(hipDeviceSynchronize());
  return 0;
}

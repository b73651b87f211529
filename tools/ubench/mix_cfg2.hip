// BASELINE configs[2]'s decode TRAFFIC and nothing else (round-5 verdict #3: "show with a probe that the ceiling is < 900 GB/s and close
// it, or hand-schedule the ring trip"): what decode_ring_kernel<4, 2048> moves through memory for 16384 x 4 MiB blocks (App. F, match
// window 4096), without a parse and without a dependence of any address on loaded data.  Per lane group of 4 lanes ("a block", 16 per
// wavefront, one wavefront per SIMD: 16384 blocks = 1024 wavefronts, as the ring loop runs them) and 64-byte step of output:
//   * a sequential read of the compressed stream: 30 bytes (ratio 2.115)
//   * a sequential, address-aligned 64-byte store of output (the ring loop stores every output byte once)
//   * with probability P / 100 one FAR match source: 64 bytes at a random byte offset 2 .. 4 KB behind the write pointer (the ring loop's
//     2 KB output ring holds the nearer ones: 61 % of the matches lie beyond it, one match per ~60 output bytes: P = 61)
// with D steps' loads in flight per block (the ring loop: 4 slots).  mix_cfg2 <waves per CU> <P> <D> [blocks = 16384]
// Build: hipcc --offload-arch=gfx950 -O3 mix_cfg2.hip -o mix_cfg2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr uint32_t BLK = 4u << 20, STR = 2u << 20;

template <int D>
__global__ __launch_bounds__(64) void mix_kernel(const uint8_t* stream, uint8_t* out, uint32_t n_blocks, uint32_t P, uint32_t* queue, uint32_t* sink) {
  const uint32_t lane = threadIdx.x & 63u, sub = lane & 3u, grp = lane >> 2;
  uint32_t acc = 0;
  for (;;) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(queue, 16u);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (base >= n_blocks) break;
    const uint32_t b = base + grp;
    if (b >= n_blocks) continue;
    uint8_t* o = out + (uint64_t)b * BLK;
    const uint8_t* s = stream + (uint64_t)b * STR;
    uint64_t rng = ((uint64_t)b * 0x9E3779B97F4A7C15ull) | 1ull;
    for (uint32_t wp = 4096u; wp < BLK; wp += 64u * D) {
      uint4 g[D];
      uint2 sv[D];
#pragma unroll
      for (int d = 0; d < D; d++) {
        const uint32_t w = wp + 64u * d;
        sv[d] = *(const uint2*)(s + ((w >> 1) & ~7u) + sub * 8u);
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t r = (uint32_t)(rng >> 33);
        const bool far = (r % 100u) < P;
        const uint32_t back = 2048u + ((r >> 8) % 2048u);
        typedef uint32_t u4a __attribute__((ext_vector_type(4), aligned(1)));
        g[d] = make_uint4(0, 0, 0, 0);
        if (far) { const u4a t = *(const u4a*)(o + (w - back) + sub * 16u); g[d] = make_uint4(t.x, t.y, t.z, t.w); }
      }
#pragma unroll
      for (int d = 0; d < D; d++) {
        uint4 v = make_uint4(sv[d].x ^ g[d].x, sv[d].y + g[d].y, sv[d].x ^ g[d].z, sv[d].y + g[d].w + wp);
        *(uint4*)(o + wp + 64u * d + sub * 16u) = v;
        acc += v.x;
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int D>
static void run(int wpc, uint32_t n_blocks, uint32_t P, const uint8_t* stream, uint8_t* out, uint32_t* queue, uint32_t* sink, int cus) {
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    CHK(hipMemset(queue, 0, 4));
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL((mix_kernel<D>), dim3(cus * wpc), dim3(64), 0, 0, stream, out, n_blocks, P, queue, sink);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  printf("waves/CU %2d  far sources %3u %% of the steps  depth %d : %8.3f ms  %7.1f GB/s of output\n", wpc, P, D, best, (double)n_blocks * BLK / best / 1e6);
}

int main(int argc, char** argv) {
  if (argc < 4) { printf("usage: mix_cfg2 <waves per CU> <P> <D> [blocks]\n"); return 2; }
  const int wpc = atoi(argv[1]), D = atoi(argv[3]);
  const uint32_t P = (uint32_t)atoi(argv[2]);
  const uint32_t nb = argc > 4 ? (uint32_t)strtoul(argv[4], 0, 10) : 16384u;
  hipDeviceProp_t pr; CHK(hipGetDeviceProperties(&pr, 0));
  uint8_t *stream, *out; uint32_t *queue, *sink;
  CHK(hipMalloc(&stream, (size_t)nb * STR + 4096)); CHK(hipMemset(stream, 3, (size_t)nb * STR + 4096));
  CHK(hipMalloc(&out, (size_t)nb * BLK + 4096)); CHK(hipMemset(out, 0, (size_t)nb * BLK + 4096));
  CHK(hipMalloc(&queue, 64)); CHK(hipMalloc(&sink, 64));
  if (D == 1) run<1>(wpc, nb, P, stream, out, queue, sink, pr.multiProcessorCount);
  if (D == 2) run<2>(wpc, nb, P, stream, out, queue, sink, pr.multiProcessorCount);
  if (D == 4) run<4>(wpc, nb, P, stream, out, queue, sink, pr.multiProcessorCount);
  if (D == 8) run<8>(wpc, nb, P, stream, out, queue, sink, pr.multiProcessorCount);
  CHK(hipDeviceSynchronize());
  return 0;
}

// Which lane wins when several lanes of ONE LDS store instruction write the same address?  (Not documented; asked by the plan for
// 16-bit table entries -- pos16[8192] + fp8[8192] per chain, DESIGN.md section 7 -- whose commit would be plain ds_write_b16 / ds_write_b8
// instead of the 32-bit atomic max: the later position has to win a bucket.)  Every lane stores its own lane number; groups of lanes
// share an address (group sizes 2, 3, 4, 8, 16, 64; members contiguous, strided by 16 and bit-reversed), widths 8 / 16 / 32 bits,
// with all lanes active and with a sparse exec mask; read back: the lane number that landed, per group.  Build:
//   hipcc --offload-arch=gfx950 -O3 lds_write_order.hip -o lds_write_order
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// pattern: 0 contiguous groups (lane / g), 1 strided (lane % (64 / g)), 2 bit-reversed lane / g
__device__ __forceinline__ uint32_t group_of(uint32_t lane, uint32_t g, int pattern) {
  if (pattern == 0) return lane / g;
  if (pattern == 1) return lane % (64u / g);
  return (__brev(lane) >> 26) / g;
}
template <int BITS>
__global__ __launch_bounds__(64) void k_order(uint32_t g, int pattern, uint64_t mask, uint32_t* out, int reps) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[256];
  const uint32_t lane = threadIdx.x;
  const uint32_t grp = group_of(lane, g, pattern);
  uint32_t winner_min = 0xFFFFFFFFu, winner_max = 0;
  for (int r = 0; r < reps; r++) {
    for (int i = lane; i < 256; i += 64) lds[i] = 0xEEEEEEEEu;
    __syncthreads();
    if ((mask >> lane) & 1ull) {
      // (addresses of different groups lie in different dwords AND different banks: grp * 4 bytes)
      if (BITS == 8) ((volatile uint8_t*)lds)[grp * 4u + (r & 3)] = (uint8_t)lane;
      else if (BITS == 16) ((volatile uint16_t*)lds)[grp * 2u + (r & 1)] = (uint16_t)lane;
      else ((volatile uint32_t*)lds)[grp] = lane;
    }
    __syncthreads();
    uint32_t v;
    if (BITS == 8) v = ((volatile uint8_t*)lds)[grp * 4u + (r & 3)];
    else if (BITS == 16) v = ((volatile uint16_t*)lds)[grp * 2u + (r & 1)];
    else v = ((volatile uint32_t*)lds)[grp];
    winner_min = v < winner_min ? v : winner_min;
    winner_max = v > winner_max ? v : winner_max;
    __syncthreads();
  }
  out[lane * 2] = winner_min; out[lane * 2 + 1] = winner_max;
}

int main() {
  uint32_t* d; CHK(hipMalloc(&d, 128 * 4));
  std::vector<uint32_t> h(128);
  const uint64_t masks[3] = {~0ull, 0xAAAAAAAAAAAAAAAAull | 1ull, 0x00FF00FF00FF00FFull};
  int bad = 0;
  for (int bits : {8, 16, 32})
    for (uint32_t g : {2u, 3u, 4u, 8u, 16u, 64u})
      for (int pattern = 0; pattern < 3; pattern++)
        for (int m = 0; m < 3; m++) {
          if (pattern == 1 && 64 % g) continue;
          if (bits == 8) hipLaunchKernelGGL(k_order<8>, dim3(1), dim3(64), 0, 0, g, pattern, masks[m], d, 64);
          else if (bits == 16) hipLaunchKernelGGL(k_order<16>, dim3(1), dim3(64), 0, 0, g, pattern, masks[m], d, 64);
          else hipLaunchKernelGGL(k_order<32>, dim3(1), dim3(64), 0, 0, g, pattern, masks[m], d, 64);
          CHK(hipDeviceSynchronize());
          CHK(hipMemcpy(h.data(), d, 128 * 4, hipMemcpyDeviceToHost));
          // expectation under test: the HIGHEST active lane of a group wins, every time
          int highest = 0, lowest = 0, other = 0, unstable = 0;
          for (uint32_t lane = 0; lane < 64; lane++) {
            if (!((masks[m] >> lane) & 1ull)) continue;
            uint32_t hi = 0, lo = 64; bool any = false;
            for (uint32_t l2 = 0; l2 < 64; l2++) {
              if (!((masks[m] >> l2) & 1ull)) continue;
              const uint32_t ga = pattern == 0 ? lane / g : pattern == 1 ? lane % (64u / g) : 0, gb = pattern == 0 ? l2 / g : pattern == 1 ? l2 % (64u / g) : 0;
              bool same;
              if (pattern == 2) {
                auto br = [](uint32_t x) { uint32_t r = 0; for (int b = 0; b < 6; b++) r |= ((x >> b) & 1u) << (5 - b); return r; };
                same = br(lane) / g == br(l2) / g;
              } else same = ga == gb;
              if (same) { any = true; hi = l2 > hi ? l2 : hi; lo = l2 < lo ? l2 : lo; }
            }
            if (!any) continue;
            const uint32_t wmin = h[lane * 2], wmax = h[lane * 2 + 1];
            if (wmin != wmax) unstable++;
            else if (wmin == hi) highest++;
            else if (wmin == lo) lowest++;
            else other++;
          }
          printf("b%-2d group %-2u pattern %d mask %d: highest-lane wins %2d  lowest %2d  other %2d  unstable %2d\n", bits, g, pattern, m, highest, lowest, other, unstable);
          if (lowest || other || unstable) bad++;
        }
  printf(bad ? "RESULT: %d configurations where the highest active lane did not win every time\n" : "RESULT: the highest active lane of a group won in every configuration, every repetition (%d)\n", bad);
  return 0;
}

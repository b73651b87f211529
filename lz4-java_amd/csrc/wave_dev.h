// wave_dev.h -- gfx950 backend of the "wave" interface used by lz4_fast_core.h.
// One instance per 64-lane wavefront; per-lane types are plain scalars (SIMT), ballots and
// broadcasts are the CDNA wave intrinsics, the match table lives in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lz4hip {

struct WaveDev {
  using VU = uint32_t;
  using VU64 = uint64_t;
  using VB = bool;
  template <bool U16> struct Entry;

  void* lds;  // 32 KB table of this wavefront
  static constexpr bool kAsmLean = true;   // the table is in LDS: the hand-scheduled lean loop (lz4_fast_v2_asm.h) applies

  __device__ __forceinline__ explicit WaveDev(void* l) : lds(l) {}

  // orders this wave's LDS traffic (hardware executes a wave's DS ops in order; this stops the compiler moving them)
  __device__ __forceinline__ static void sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // shader clock, issued after `dep` is available (profiling kernel only)
  __device__ __forceinline__ static uint64_t tick(uint32_t dep) {
    uint64_t t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "s"(__builtin_amdgcn_readfirstlane(dep)) : "memory");
    return t;
  }
  __device__ __forceinline__ static VU lane() { return __lane_id(); }
  __device__ __forceinline__ static VU64 lanemask_lt() { return (1ull << __lane_id()) - 1ull; }
  // (the i1 form folds straight into the compare that produced `b`; HIP's __ballot(int) costs a v_cndmask + v_cmp on top)
  __device__ __forceinline__ static uint64_t ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }
  // per-lane predicate from a wave-uniform lane mask (the SGPR pair is used as the lane mask directly)
  __device__ __forceinline__ static bool lanes(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
  template <class T> __device__ __forceinline__ static T select(bool c, T a, T b) { return c ? a : b; }
  __device__ __forceinline__ static VU64 u64(VU v) { return (uint64_t)v; }
  __device__ __forceinline__ static VU lo32(VU64 v) { return (uint32_t)v; }
  __device__ __forceinline__ static VU clz64(VU64 v) { return (uint32_t)__clzll((long long)v); }

  __device__ __forceinline__ static uint32_t bcast(VU v, int l) { return __builtin_amdgcn_readlane(v, l); }
  __device__ __forceinline__ static uint64_t bcast64(VU64 v, int l) {
    uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, l);
    uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
  }
  template <bool U16> __device__ __forceinline__ static auto bcast_e(typename Entry<U16>::V v, int l) {
    if constexpr (U16) return bcast(v, l);
    else return bcast64(v, l);
  }
  template <bool U16> __device__ __forceinline__ static auto shfl_e(typename Entry<U16>::V v, VU srcl) {
    if constexpr (U16) return (uint32_t)__shfl((int)v, (int)srcl, 64);
    else {
      uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, (int)srcl, 64);
      uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)srcl, 64);
      return ((uint64_t)hi << 32) | lo;
    }
  }

  // ---- global memory (unaligned accesses are single instructions on gfx950/amdhsa) ----
  __device__ __forceinline__ static VU ld8(const uint8_t* b, VU i, bool m) { return m ? (uint32_t)b[i] : 0u; }
  __device__ __forceinline__ static VU ld32(const uint8_t* b, VU i, bool m) {
    uint32_t v = 0;
    if (m) __builtin_memcpy(&v, b + i, 4);
    return v;
  }
  __device__ __forceinline__ static VU64 ld64(const uint8_t* b, VU i, bool m) {
    uint64_t v = 0;
    if (m) __builtin_memcpy(&v, b + i, 8);
    return v;
  }
  // unconditional loads: the caller passes an in-bounds (clamped) index, so no exec-mask branch is needed
  __device__ __forceinline__ static VU ldu8(const uint8_t* b, VU i) { return (uint32_t)b[i]; }
  __device__ __forceinline__ static VU ldu32(const uint8_t* b, VU i) { uint32_t v; __builtin_memcpy(&v, b + i, 4); return v; }
  __device__ __forceinline__ static VU64 ldu64(const uint8_t* b, VU i) { uint64_t v; __builtin_memcpy(&v, b + i, 8); return v; }
  // candidate side of the match fetch (a non-temporal load here measured 4 % SLOWER: ~half of these hit L2)
  __device__ __forceinline__ static VU64 ldu64_cand(const uint8_t* b, VU i) { return ldu64(b, i); }
#ifndef LZ4HIP_PF_KB
#define LZ4HIP_PF_KB 1  /* source prefetch chunk in KB (1, 2 or 4): how far ahead of the parse position the source is touched */
#endif
  // touches src[from, from + LZ4HIP_PF_KB KB) (clamped to n) so that later loads of that span hit L2: 1 KB wave loads, waited for here
  __device__ __forceinline__ static void prefetch4k(const uint8_t* b, uint32_t from, uint32_t n) {
    const uint32_t top = n - 16u;  // n >= 16
    uint4 v[LZ4HIP_PF_KB];
#pragma unroll
    for (int i = 0; i < LZ4HIP_PF_KB; i++) {
      uint32_t a = from + (uint32_t)i * 1024u + __lane_id() * 16u;
      a = a < top ? a : top;
      __builtin_memcpy(&v[i], b + a, 16);
    }
#pragma unroll
    for (int i = 0; i < LZ4HIP_PF_KB; i++) asm volatile("" ::"v"(v[i].x), "v"(v[i].y), "v"(v[i].z), "v"(v[i].w));
  }
  static constexpr uint32_t kPrefetchBytes = LZ4HIP_PF_KB * 1024u;
  __device__ __forceinline__ static VU opaque(VU v) { asm("" : "+v"(v)); return v; }  // identity the optimiser cannot see through
  __device__ __forceinline__ static void consume(VU v) { asm volatile("" ::"v"(v)); }  // keeps a prefetch load alive
  __device__ __forceinline__ static VU vmin(VU a, VU b) { return a < b ? a : b; }
  __device__ __forceinline__ static VU vmax(VU a, VU b) { return a > b ? a : b; }
  __device__ __forceinline__ static VU ctz64v(VU64 v) { return v ? (uint32_t)__builtin_ctzll(v) : 64u; }  // per-lane
  // lane `l` (wave-uniform) of v replaced by the scalar s
  // (v_writelane_b32 has no builtin in this toolchain and needs M0 for a second scalar operand; compare + select it is)
  __device__ __forceinline__ static VU set_lane(VU v, int l, uint32_t s) { return (int)__lane_id() == l ? s : v; }
  // number of set bits of the wave-uniform mask m below this lane
  __device__ __forceinline__ static VU mbcnt(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  }
  __device__ __forceinline__ static VU shfl(VU v, VU srcl) { return (uint32_t)__shfl((int)v, (int)srcl, 64); }
  __device__ __forceinline__ static VU div255(VU a) { return a / 255u; }
  // exclusive prefix sum across the wavefront: 4 row_shr DPP adds + row_bcast15/31 (no LDS crossbar round trips)
  __device__ __forceinline__ static VU excl_scan(VU a) {
    int x = (int)a;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return (uint32_t)x - a;
  }
  __device__ __forceinline__ static VU shr(VU a, VU k) { return a >> (k & 31u); }  // per-lane shift amount
  __device__ __forceinline__ static VU shfl_up1(VU v) { return (uint32_t)__shfl_up((int)v, 1, 64); }  // lane l <- lane l-1
  __device__ __forceinline__ static uint32_t sld32(const uint8_t* b, uint32_t i) {
    uint32_t v;
    __builtin_memcpy(&v, b + i, 4);
    return v;
  }
  __device__ __forceinline__ static void st8(uint8_t* b, VU i, VU v, bool m) {
    // (non-temporal byte stores here were +3.5 % on the bench workload -- the output no longer pushes the source out of L2 --
    // but they reach memory as partial-line writes: WRITE_SIZE 2.1 -> 11.4 GB per launch, profiles/r01j; not worth it)
#ifdef LZ4HIP_NT_OUT   // developer A/B builds
    if (m) __builtin_nontemporal_store((uint8_t)v, b + i);
#else
    if (m) b[i] = (uint8_t)v;
#endif
  }
  __device__ __forceinline__ static void st32(uint8_t* b, VU i, VU v, bool m) {
#ifdef LZ4HIP_NT_OUT
    typedef uint32_t u32u __attribute__((aligned(1)));
    if (m) __builtin_nontemporal_store((uint32_t)v, (u32u*)(b + i));
#else
    if (m) __builtin_memcpy(b + i, &v, 4);
#endif
  }
  // lane `l` of v replaced by the scalar s, both wave-uniform: ONE v_writelane_b32 (set_lane's compare + select is two VALU
  // instructions and needs the lane index in a VGPR)
  __device__ __forceinline__ static VU writelane(VU v, uint32_t s, uint32_t l) {
    // (gfx9 VALU instructions read ONE SGPR over the constant bus: the lane select goes through M0)
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(s), "s"(l) : "m0");
    return v;
  }
  __device__ __forceinline__ static VU shfl_down1(VU v) {   // lane l <- lane l+1 (lane 63: unspecified)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false);   // wave_shl:1
  }
  // bytes [sh, sh+4) of the 8-byte little-endian value {hi, lo}, sh = 0..3 (per lane)
  __device__ __forceinline__ static VU alignbyte(VU hi, VU lo, VU sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
  __device__ __forceinline__ static void st16(uint16_t* b, VU i, VU v, bool m) {
    if (m) b[i] = (uint16_t)v;
  }
  // per-lane evaluation of a scalar function (lanes diverge freely inside f)
  template <class F> __device__ __forceinline__ static VU64 map_lanes64(F f) { return f((uint32_t)__lane_id()); }
  // same, handing each lane its own element of a and m
  template <class F> __device__ __forceinline__ static VU64 map_lanes64v(VU a, bool m, F f) { return f((uint32_t)__lane_id(), a, m); }
  // cooperative byte copy dst[dpos..+len) = src[spos..+len); regions never overlap
  __device__ __forceinline__ static void copy(uint8_t* dst, uint32_t dpos, const uint8_t* src, uint32_t spos, uint32_t len) {
    const uint32_t l = __lane_id();
    uint8_t* d = dst + dpos;
    const uint8_t* s = src + spos;
    const uint32_t body = len & ~15u;
    for (uint32_t i = l * 16u; i < body; i += 1024u) {
      uint4 v;
      __builtin_memcpy(&v, s + i, 16);
      __builtin_memcpy(d + i, &v, 16);
    }
    const uint32_t i = body + l;
    if (i < len) d[i] = s[i];
  }

  // ---- one word per lane to / from a 64-word array in global memory; a 4-word header (finder / writer ring, mail_ring.h) ----
  __device__ __forceinline__ static void st_lanes(uint32_t* base, VU v) { base[__lane_id()] = v; }
  __device__ __forceinline__ static VU ld_lanes(const uint32_t* base) { return base[__lane_id()]; }
  __device__ __forceinline__ static void st_hdr(uint32_t* h, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const uint32_t l = __lane_id();
    if (l < 4u) h[l] = l == 0u ? a : (l == 1u ? b : (l == 2u ? c : d));
  }
  __device__ __forceinline__ static void ld_hdr(const uint32_t* h, uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
    const uint32_t hv = h[__lane_id() & 3u];
    a = __builtin_amdgcn_readlane(hv, 0); b = __builtin_amdgcn_readlane(hv, 1); c = __builtin_amdgcn_readlane(hv, 2); d = __builtin_amdgcn_readlane(hv, 3);
  }

  // ---- LDS table ----
  template <bool U16> __device__ __forceinline__ void lds_fill(uint32_t count, typename Entry<U16>::S val) {
    using S = typename Entry<U16>::S;
    S* t = (S*)lds;
    for (uint32_t i = __lane_id(); i < count; i += 64u) t[i] = val;
  }
  template <bool U16> __device__ __forceinline__ auto lds_rd(VU h, bool m) {
    using S = typename Entry<U16>::S;
    return m ? ((S*)lds)[h] : (S)0;
  }
  template <bool U16> __device__ __forceinline__ auto lds_rdu(VU h) {
    using S = typename Entry<U16>::S;
    return ((S*)lds)[h];
  }
  template <bool U16> __device__ __forceinline__ auto lds_max(VU h, typename Entry<U16>::V v, bool m) {
    using S = typename Entry<U16>::S;
    S old = 0;
    if (m) {
      if constexpr (U16) old = atomicMax(&((uint32_t*)lds)[h], v);
      else old = atomicMax(&((unsigned long long*)lds)[h], (unsigned long long)v);
    }
    return old;
  }
  template <bool U16> __device__ __forceinline__ void lds_wr(VU h, typename Entry<U16>::V v, bool m) {
    using S = typename Entry<U16>::S;
    if (m) ((S*)lds)[h] = v;
  }
};
template <> struct WaveDev::Entry<true> { using S = uint32_t; using V = uint32_t; };
template <> struct WaveDev::Entry<false> { using S = uint64_t; using V = uint64_t; };

}  // namespace lz4hip

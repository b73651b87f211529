// tests/hostsim/wave_host.h -- TEST INFRASTRUCTURE ONLY (never part of liblz4hip.so).
//
// A lock-step 64-lane simulator of the "wave" backend, so the CPU test-suite can execute the very
// source of lz4-java_amd/csrc/lz4_fast_core.h (the algorithm the gfx950 kernel runs) and compare it
// with the oracle before any GPU time is spent.  Per-lane values are 64-element vectors; LDS
// atomics are applied in a PSEUDO-RANDOM lane order on purpose: the algorithm must not depend on
// the order in which the hardware serialises colliding lanes of one DS instruction.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>

namespace hostsim {

template <class T>
struct V {
  T v[64];
  V() { for (int i = 0; i < 64; i++) v[i] = T(); }
  V(T s) { for (int i = 0; i < 64; i++) v[i] = s; }  // broadcast (implicit on purpose)
#define HS_BIN(OP)                                                                  \
  friend V operator OP(const V& a, const V& b) {                                    \
    V r;                                                                            \
    for (int i = 0; i < 64; i++) r.v[i] = (T)(a.v[i] OP b.v[i]);                    \
    return r;                                                                       \
  }
  HS_BIN(+) HS_BIN(-) HS_BIN(*) HS_BIN(&) HS_BIN(|) HS_BIN(^)
#undef HS_BIN
#define HS_CMP(OP)                                                                  \
  friend V<bool> operator OP(const V& a, const V& b) {                              \
    V<bool> r;                                                                      \
    for (int i = 0; i < 64; i++) r.v[i] = a.v[i] OP b.v[i];                         \
    return r;                                                                       \
  }
  HS_CMP(==) HS_CMP(!=) HS_CMP(<) HS_CMP(<=) HS_CMP(>) HS_CMP(>=)
#undef HS_CMP
  friend V operator<<(const V& a, int s) { V r; for (int i = 0; i < 64; i++) r.v[i] = (T)(a.v[i] << s); return r; }
  friend V operator>>(const V& a, int s) { V r; for (int i = 0; i < 64; i++) r.v[i] = (T)(a.v[i] >> s); return r; }
  V operator~() const { V r; for (int i = 0; i < 64; i++) r.v[i] = (T)~v[i]; return r; }
  V<bool> operator!() const { V<bool> r; for (int i = 0; i < 64; i++) r.v[i] = !v[i]; return r; }
};

struct WaveHost {
  using VU = V<uint32_t>;
  using VU64 = V<uint64_t>;
  using VB = V<bool>;
  template <bool U16> struct Entry;

  std::vector<uint64_t> lds;  // fast compressor: 32 KB; HC head table: 32768 x u32 = 128 KB
  uint64_t rng = 0x9E3779B97F4A7C15ull;
  const uint8_t* src_lo = nullptr; const uint8_t* src_hi = nullptr;  // bounds for checking loads
  uint8_t* dst_lo = nullptr; uint8_t* dst_hi = nullptr;              // bounds for checking stores
  bool oob = false;

  WaveHost() : lds(16384, 0) {}
  void bounds(const uint8_t* s, size_t n, uint8_t* d, size_t cap) { src_lo = s; src_hi = s + n; dst_lo = d; dst_hi = d + cap; }

  static void sync() {}
  static uint64_t tick(uint32_t) { return 0; }
  static VU lane() { VU r; for (int i = 0; i < 64; i++) r.v[i] = (uint32_t)i; return r; }
  static VU64 lanemask_lt() { VU64 r; for (int i = 0; i < 64; i++) r.v[i] = (1ull << i) - 1ull; return r; }
  static uint64_t ballot(const VB& b) { uint64_t m = 0; for (int i = 0; i < 64; i++) if (b.v[i]) m |= 1ull << i; return m; }
  static VB lanes(uint64_t m) { VB r; for (int i = 0; i < 64; i++) r.v[i] = (m >> i) & 1u; return r; }
  template <class T> static V<T> select(const VB& c, const V<T>& a, const V<T>& b) {
    V<T> r; for (int i = 0; i < 64; i++) r.v[i] = c.v[i] ? a.v[i] : b.v[i]; return r;
  }
  static VU64 u64(const VU& v) { VU64 r; for (int i = 0; i < 64; i++) r.v[i] = v.v[i]; return r; }
  static VU lo32(const VU64& v) { VU r; for (int i = 0; i < 64; i++) r.v[i] = (uint32_t)v.v[i]; return r; }
  static VU clz64(const VU64& v) { VU r; for (int i = 0; i < 64; i++) r.v[i] = v.v[i] ? (uint32_t)__builtin_clzll(v.v[i]) : 64u; return r; }
  static uint32_t bcast(const VU& v, int l) { return v.v[l]; }
  static uint64_t bcast64(const VU64& v, int l) { return v.v[l]; }
  template <bool U16> static auto bcast_e(const typename Entry<U16>::V& v, int l) { return v.v[l]; }
  template <bool U16> static auto shfl_e(const typename Entry<U16>::V& v, const VU& srcl) {
    typename Entry<U16>::V r;
    for (int i = 0; i < 64; i++) r.v[i] = v.v[srcl.v[i] & 63u];
    return r;
  }

  bool in_ok(const uint8_t* p, size_t k) { if (p < src_lo || p + k > src_hi) { oob = true; return false; } return true; }
  bool out_ok(const uint8_t* p, size_t k) { if (p < dst_lo || p + k > dst_hi) { oob = true; return false; } return true; }

  VU ld8(const uint8_t* b, const VU& i, const VB& m) {
    VU r; for (int l = 0; l < 64; l++) if (m.v[l] && in_ok(b + i.v[l], 1)) r.v[l] = b[i.v[l]]; return r;
  }
  VU ld32(const uint8_t* b, const VU& i, const VB& m) {
    VU r; for (int l = 0; l < 64; l++) if (m.v[l] && in_ok(b + i.v[l], 4)) memcpy(&r.v[l], b + i.v[l], 4); return r;
  }
  VU64 ld64(const uint8_t* b, const VU& i, const VB& m) {
    VU64 r; for (int l = 0; l < 64; l++) if (m.v[l] && in_ok(b + i.v[l], 8)) memcpy(&r.v[l], b + i.v[l], 8); return r;
  }
  VU ldu8(const uint8_t* b, const VU& i) { return ld8(b, i, VB(true)); }
  VU ldu32(const uint8_t* b, const VU& i) { return ld32(b, i, VB(true)); }
  VU64 ldu64(const uint8_t* b, const VU& i) { return ld64(b, i, VB(true)); }
  VU64 ldu64_cand(const uint8_t* b, const VU& i) { return ld64(b, i, VB(true)); }
  static VU vmin(const VU& a, const VU& b) { VU r; for (int l = 0; l < 64; l++) r.v[l] = a.v[l] < b.v[l] ? a.v[l] : b.v[l]; return r; }
  static VU opaque(const VU& v) { return v; }
  static void consume(const VU&) {}
  static void prefetch4k(const uint8_t*, uint32_t, uint32_t) {}
  static constexpr uint32_t kPrefetchBytes = 4096u;
  static VU vmax(const VU& a, const VU& b) { VU r; for (int l = 0; l < 64; l++) r.v[l] = a.v[l] > b.v[l] ? a.v[l] : b.v[l]; return r; }
  static VU ctz64v(const VU64& v) { VU r; for (int i = 0; i < 64; i++) r.v[i] = v.v[i] ? (uint32_t)__builtin_ctzll(v.v[i]) : 64u; return r; }
  static VU set_lane(const VU& v, int l, uint32_t s) { VU r = v; r.v[l] = s; return r; }
  static VU mbcnt(uint64_t m) { VU r; for (int i = 0; i < 64; i++) r.v[i] = (uint32_t)__builtin_popcountll(m & ((1ull << i) - 1ull)); return r; }
  static VU shfl(const VU& v, const VU& srcl) { VU r; for (int i = 0; i < 64; i++) r.v[i] = v.v[srcl.v[i] & 63u]; return r; }
  static VU div255(const VU& a) { VU r; for (int l = 0; l < 64; l++) r.v[l] = a.v[l] / 255u; return r; }
  static VU excl_scan(const VU& a) { VU r; uint32_t acc = 0; for (int l = 0; l < 64; l++) { r.v[l] = acc; acc += a.v[l]; } return r; }
  static VU shr(const VU& a, const VU& k) { VU r; for (int l = 0; l < 64; l++) r.v[l] = a.v[l] >> (k.v[l] & 31u); return r; }
  static VU shfl_up1(const VU& v) { VU r; r.v[0] = v.v[0]; for (int l = 1; l < 64; l++) r.v[l] = v.v[l - 1]; return r; }
  uint32_t sld32(const uint8_t* b, uint32_t i) { uint32_t v = 0; if (in_ok(b + i, 4)) memcpy(&v, b + i, 4); return v; }
  void st8(uint8_t* b, const VU& i, const VU& v, const VB& m) {
    for (int l = 0; l < 64; l++) if (m.v[l] && out_ok(b + i.v[l], 1)) b[i.v[l]] = (uint8_t)v.v[l];
  }
  void st32(uint8_t* b, const VU& i, const VU& v, const VB& m) {
    for (int l = 0; l < 64; l++) if (m.v[l] && out_ok(b + i.v[l], 4)) memcpy(b + i.v[l], &v.v[l], 4);
  }
  static VU writelane(const VU& v, uint32_t s, uint32_t l) { VU r = v; r.v[l & 63u] = s; return r; }
  // finder / writer ring (csrc/mail_ring.h): one word per lane to / from a 64-word array, a 4-word header
  static void st_lanes(uint32_t* base, const VU& v) { for (int l = 0; l < 64; l++) base[l] = v.v[l]; }
  static VU ld_lanes(const uint32_t* base) { VU r; for (int l = 0; l < 64; l++) r.v[l] = base[l]; return r; }
  static void st_hdr(uint32_t* h, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { h[0] = a; h[1] = b; h[2] = c; h[3] = d; }
  static void ld_hdr(const uint32_t* h, uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) { a = h[0]; b = h[1]; c = h[2]; d = h[3]; }
  static VU shfl_down1(const VU& v) { VU r; for (int l = 0; l < 63; l++) r.v[l] = v.v[l + 1]; r.v[63] = 0xDEADBEEFu; return r; }
  static VU alignbyte(const VU& hi, const VU& lo, const VU& sh) {
    VU r; for (int l = 0; l < 64; l++) r.v[l] = (uint32_t)((((uint64_t)hi.v[l] << 32) | lo.v[l]) >> (8u * (sh.v[l] & 3u))); return r;
  }
  void st16(uint16_t* b, const VU& i, const VU& v, const VB& m) {
    for (int l = 0; l < 64; l++) if (m.v[l]) b[i.v[l]] = (uint16_t)v.v[l];
  }
  template <class F> static VU64 map_lanes64v(const VU& a, const VB& m, F f) { VU64 r; for (int l = 0; l < 64; l++) r.v[l] = f((uint32_t)l, a.v[l], m.v[l]); return r; }
  template <class F> static VU64 map_lanes64(F f) { VU64 r; for (int l = 0; l < 64; l++) r.v[l] = f((uint32_t)l); return r; }
  void copy(uint8_t* dst, uint32_t dpos, const uint8_t* src, uint32_t spos, uint32_t len) {
    if (len && in_ok(src + spos, len) && out_ok(dst + dpos, len)) memcpy(dst + dpos, src + spos, len);
  }

  template <bool U16> void lds_fill(uint32_t count, typename Entry<U16>::S val) {
    using S = typename Entry<U16>::S;
    S* t = (S*)lds.data();
    for (uint32_t i = 0; i < count; i++) t[i] = val;
  }
  template <bool U16> auto lds_rd(const VU& h, const VB& m) {
    using S = typename Entry<U16>::S;
    typename Entry<U16>::V r;
    for (int l = 0; l < 64; l++) if (m.v[l]) r.v[l] = ((S*)lds.data())[h.v[l]];
    return r;
  }
  template <bool U16> auto lds_rdu(const VU& h) { return lds_rd<U16>(h, VB(true)); }
  template <bool U16> auto lds_max(const VU& h, const typename Entry<U16>::V& v, const VB& m) {
    using S = typename Entry<U16>::S;
    typename Entry<U16>::V old;
    int order[64];
    for (int i = 0; i < 64; i++) order[i] = i;
    for (int i = 63; i > 0; i--) {  // Fisher-Yates with a splitmix-style step
      rng = rng * 6364136223846793005ull + 1442695040888963407ull;
      int j = (int)((rng >> 33) % (uint64_t)(i + 1));
      int t = order[i]; order[i] = order[j]; order[j] = t;
    }
    for (int q = 0; q < 64; q++) {
      int l = order[q];
      if (!m.v[l]) continue;
      S* p = &((S*)lds.data())[h.v[l]];
      old.v[l] = *p;
      if (v.v[l] > *p) *p = v.v[l];
    }
    return old;
  }
  template <bool U16> void lds_wr(const VU& h, const typename Entry<U16>::V& v, const VB& m) {
    using S = typename Entry<U16>::S;
    for (int l = 0; l < 64; l++) if (m.v[l]) ((S*)lds.data())[h.v[l]] = v.v[l];
  }
};
template <> struct WaveHost::Entry<true> { using S = uint32_t; using V = hostsim::V<uint32_t>; };
template <> struct WaveHost::Entry<false> { using S = uint64_t; using V = hostsim::V<uint64_t>; };

}  // namespace hostsim

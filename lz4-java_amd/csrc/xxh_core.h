// xxh_core.h -- one-shot XXH32 / XXH64 (xxHash 0.6.5), one thread per buffer.
// Replaces XXH32/XXH64 as reached from XXHash32JNI.hash / XXHash64JNI.hash through
// /root/reference/src/jni/net_jpountz_xxhash_XXHashJNI.c:54 and :164; the algorithm is the one the
// reference restates in src/build/source_templates/xxhash32_hash.template:27-83 and
// xxhash64_hash.template:27-102 (primes: XXHashConstants.java:22-32); SURVEY.md App. D.
// Each thread streams its buffer with 16-byte (XXH32: one stripe) / 32-byte (XXH64) loads; the four
// accumulators are independent multiply-rotate chains, i.e. 4-way ILP per thread.
#pragma once
#include <stdint.h>

#ifndef LZ4HIP_DEV
#if defined(__HIPCC__)
#define LZ4HIP_DEV __device__ __forceinline__
#else
#define LZ4HIP_DEV inline
#endif
#endif

namespace lz4hip {

LZ4HIP_DEV uint32_t xrd32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
LZ4HIP_DEV uint64_t xrd64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
LZ4HIP_DEV uint32_t xrotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
LZ4HIP_DEV uint64_t xrotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

LZ4HIP_DEV uint32_t xxh32_one(const uint8_t* p, uint32_t len, uint32_t seed) {
  const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
  const uint8_t* const end = p + len;
  uint32_t h;
  if (len >= 16u) {
    const uint8_t* const limit = end - 16;
    uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
    // four stripes per trip while they last: the 4 x 16-byte loads are issued back to back, so each thread keeps
    // 64 bytes in flight (one thread per buffer means one cache line per lane per load: more bytes per request pays)
    while (p + 64 <= end) {
      uint32_t w[16];
      __builtin_memcpy(w, p, 64);
#pragma unroll
      for (int k = 0; k < 16; k += 4) {
        v1 = xrotl32(v1 + w[k + 0] * P2, 13) * P1;
        v2 = xrotl32(v2 + w[k + 1] * P2, 13) * P1;
        v3 = xrotl32(v3 + w[k + 2] * P2, 13) * P1;
        v4 = xrotl32(v4 + w[k + 3] * P2, 13) * P1;
      }
      p += 64;
    }
    while (p <= limit) {
      uint32_t w[4];
      __builtin_memcpy(w, p, 16);
      v1 = xrotl32(v1 + w[0] * P2, 13) * P1;
      v2 = xrotl32(v2 + w[1] * P2, 13) * P1;
      v3 = xrotl32(v3 + w[2] * P2, 13) * P1;
      v4 = xrotl32(v4 + w[3] * P2, 13) * P1;
      p += 16;
    }
    h = xrotl32(v1, 1) + xrotl32(v2, 7) + xrotl32(v3, 12) + xrotl32(v4, 18);
  } else {
    h = seed + P5;
  }
  h += len;
  while (p + 4 <= end) { h = xrotl32(h + xrd32(p) * P3, 17) * P4; p += 4; }
  while (p < end) { h = xrotl32(h + (uint32_t)(*p) * P5, 11) * P1; p++; }
  h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
  return h;
}

LZ4HIP_DEV uint64_t xxh64_round(uint64_t acc, uint64_t in) {
  return xrotl64(acc + in * 14029467366897019727ull, 31) * 11400714785074694791ull;
}

LZ4HIP_DEV uint64_t xxh64_one(const uint8_t* p, uint32_t len, uint64_t seed) {
  const uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P3 = 1609587929392839161ull,
                 P4 = 9650029242287828579ull, P5 = 2870177450012600261ull;
  const uint8_t* const end = p + len;
  uint64_t h;
  if (len >= 32u) {
    const uint8_t* const limit = end - 32;
    uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
    while (p + 128 <= end) {  // four stripes per trip: 128 bytes in flight per thread
      uint64_t w[16];
      __builtin_memcpy(w, p, 128);
#pragma unroll
      for (int k = 0; k < 16; k += 4) {
        v1 = xxh64_round(v1, w[k + 0]);
        v2 = xxh64_round(v2, w[k + 1]);
        v3 = xxh64_round(v3, w[k + 2]);
        v4 = xxh64_round(v4, w[k + 3]);
      }
      p += 128;
    }
    while (p <= limit) {
      uint64_t w[4];
      __builtin_memcpy(w, p, 32);
      v1 = xxh64_round(v1, w[0]);
      v2 = xxh64_round(v2, w[1]);
      v3 = xxh64_round(v3, w[2]);
      v4 = xxh64_round(v4, w[3]);
      p += 32;
    }
    h = xrotl64(v1, 1) + xrotl64(v2, 7) + xrotl64(v3, 12) + xrotl64(v4, 18);
    h = (h ^ xxh64_round(0, v1)) * P1 + P4;
    h = (h ^ xxh64_round(0, v2)) * P1 + P4;
    h = (h ^ xxh64_round(0, v3)) * P1 + P4;
    h = (h ^ xxh64_round(0, v4)) * P1 + P4;
  } else {
    h = seed + P5;
  }
  h += (uint64_t)len;
  while (p + 8 <= end) { h ^= xxh64_round(0, xrd64(p)); h = xrotl64(h, 27) * P1 + P4; p += 8; }
  if (p + 4 <= end) { h ^= (uint64_t)xrd32(p) * P1; h = xrotl64(h, 23) * P2 + P3; p += 4; }
  while (p < end) { h ^= (uint64_t)(*p) * P5; h = xrotl64(h, 11) * P1; p++; }
  h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
  return h;
}

// ---- streaming (StreamingXXHash32/64: XXH32/XXH64_reset / _update / _digest, xxHash 0.6.5, as reached through
// /root/reference/src/jni/net_jpountz_xxhash_XXHashJNI.c:89-145 and :199-255).  The state of a stream between two updates
// is this record (in device memory): the four accumulators, the bytes of the last, incomplete stripe, the length so far and
// the digest of everything so far (every update leaves it current, so getValue() is a 4/8-byte read, not a launch).
template <class T> struct XxhRec {
  T v[4];
  uint64_t total;
  T seed, digest;
  uint32_t memsize, pad;
  uint8_t mem[4 * sizeof(T)];
};
template <class T> struct XxhOps;
template <> struct XxhOps<uint32_t> {
  static LZ4HIP_DEV uint32_t init(uint32_t seed, uint32_t k) {
    return k == 0u ? seed + 2654435761u + 2246822519u : (k == 1u ? seed + 2246822519u : (k == 2u ? seed : seed - 2654435761u));
  }
  static LZ4HIP_DEV uint32_t round(uint32_t acc, uint32_t in) { return xrotl32(acc + in * 2246822519u, 13) * 2654435761u; }
  static LZ4HIP_DEV uint32_t premul(uint32_t in) { return in * 2246822519u; }   // round(acc, in) == round_pre(acc, premul(in))
  static LZ4HIP_DEV uint32_t round_pre(uint32_t acc, uint32_t pre) { return xrotl32(acc + pre, 13) * 2654435761u; }
  // digest of a stream of `total` bytes: accumulators + the `rem` bytes after the last whole stripe (XXH32_digest)
  static LZ4HIP_DEV uint32_t finish(uint32_t v1, uint32_t v2, uint32_t v3, uint32_t v4, uint32_t seed, const uint8_t* p, uint32_t rem, uint64_t total) {
    const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    uint32_t h = total >= 16u ? xrotl32(v1, 1) + xrotl32(v2, 7) + xrotl32(v3, 12) + xrotl32(v4, 18) : seed + P5;
    h += (uint32_t)total;
    const uint8_t* const end = p + rem;
    while (p + 4 <= end) { h = xrotl32(h + xrd32(p) * P3, 17) * P4; p += 4; }
    while (p < end) { h = xrotl32(h + (uint32_t)(*p) * P5, 11) * P1; p++; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
  }
};
template <> struct XxhOps<uint64_t> {
  static LZ4HIP_DEV uint64_t init(uint64_t seed, uint32_t k) {
    const uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull;
    return k == 0u ? seed + P1 + P2 : (k == 1u ? seed + P2 : (k == 2u ? seed : seed - P1));
  }
  static LZ4HIP_DEV uint64_t round(uint64_t acc, uint64_t in) { return xxh64_round(acc, in); }
  static LZ4HIP_DEV uint64_t premul(uint64_t in) { return in * 14029467366897019727ull; }
  static LZ4HIP_DEV uint64_t round_pre(uint64_t acc, uint64_t pre) { return xrotl64(acc + pre, 31) * 11400714785074694791ull; }
  static LZ4HIP_DEV uint64_t finish(uint64_t v1, uint64_t v2, uint64_t v3, uint64_t v4, uint64_t seed, const uint8_t* p, uint32_t rem, uint64_t total) {
    const uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P3 = 1609587929392839161ull,
                   P4 = 9650029242287828579ull, P5 = 2870177450012600261ull;
    uint64_t h;
    if (total >= 32u) {
      h = xrotl64(v1, 1) + xrotl64(v2, 7) + xrotl64(v3, 12) + xrotl64(v4, 18);
      h = (h ^ xxh64_round(0, v1)) * P1 + P4;
      h = (h ^ xxh64_round(0, v2)) * P1 + P4;
      h = (h ^ xxh64_round(0, v3)) * P1 + P4;
      h = (h ^ xxh64_round(0, v4)) * P1 + P4;
    } else {
      h = seed + P5;
    }
    h += total;
    const uint8_t* const end = p + rem;
    while (p + 8 <= end) { h ^= xxh64_round(0, xrd64(p)); h = xrotl64(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)xrd32(p) * P1; h = xrotl64(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (uint64_t)(*p) * P5; h = xrotl64(h, 11) * P1; p++; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
  }
};

}  // namespace lz4hip

/*
 * oracle/lz4_oracle.c -- TEST INFRASTRUCTURE ONLY (see lz4_oracle.h for the scope statement).
 *
 * Plain-C restatement of liblz4 1.9.3 / xxHash 0.6.5 as reached by lz4-java's JNI shim
 * (/root/reference/src/jni/net_jpountz_lz4_LZ4JNI.c:75,122,169,216,237 and
 *  /root/reference/src/jni/net_jpountz_xxhash_XXHashJNI.c:54,78,164,188).
 * The liblz4 sources are NOT in /root/reference (empty submodule src/lz4); the algorithms are
 * restated from the published block format + SURVEY.md Appendix A (fast), B (HC), C (decode),
 * D (xxhash), and pinned against the reference's prebuilt liblz4-java.so by tests/.
 */
#include "lz4_oracle.h"
#include <string.h>
#include <stdlib.h>

#define MINMATCH 4
#define LASTLITERALS 5
#define MFLIMIT 12
#define LZ4_MIN_LENGTH (MFLIMIT + 1)
#define LZ4_64K_LIMIT (65536 + (MFLIMIT - 1))
#define LZ4_MAX_INPUT 0x7E000000
#define MAX_DISTANCE 65535
#define RUN_MASK 15u
#define ML_MASK 15u

static inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

/* LZ4_compressBound, reached from LZ4JNI.c:237; Java twin LZ4Utils.java:34-41 */
int lz4o_compress_bound(int n) {
  if (n < 0 || (unsigned)n > (unsigned)LZ4_MAX_INPUT) return 0;
  return n + n / 255 + 16;
}

/* ------------------------------------------------------------------------------------------
 * Fast compressor: LZ4_compress_default (LZ4JNI.c:75) == LZ4_compress_fast(acceleration 1),
 * SURVEY.md Appendix A.
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t hash4(const uint8_t* p, int hlog) { return (rd32(p) * 2654435761u) >> (32 - hlog); }
static inline uint32_t hash5(const uint8_t* p, int hlog) {
  return (uint32_t)(((rd64(p) << 24) * 889523592379ull) >> (64 - hlog));
}

/* number of equal bytes between src[a..] and src[b..], a stops at limit */
static int count_eq(const uint8_t* src, int a, int b, int limit) {
  int s = a;
  while (a < limit && src[a] == src[b]) { a++; b++; }
  return a - s;
}

static uint8_t* put_len(uint8_t* op, int len) { /* 255-run encoding of (len) where len >= 0 */
  while (len >= 255) { *op++ = 255; len -= 255; }
  *op++ = (uint8_t)len;
  return op;
}

int lz4o_compress_fast(const uint8_t* src, int n, uint8_t* dst, int cap) {
  uint32_t table[8192];
  if (n < 0 || (unsigned)n > (unsigned)LZ4_MAX_INPUT) return 0;
  const int limited = cap < lz4o_compress_bound(n);
  uint8_t* op = dst;
  uint8_t* const oend = dst + cap;
  if (n == 0) {
    if (limited && cap <= 0) return 0;
    *op = 0;
    return 1;
  }
  const int by_u16 = n < LZ4_64K_LIMIT;
  const int hlog = by_u16 ? 13 : 12;
#define HASHP(p) (by_u16 ? hash4(src + (p), hlog) : hash5(src + (p), hlog))
  const int mflimit_plus_one = n - MFLIMIT + 1;
  const int matchlimit = n - LASTLITERALS;
  int anchor = 0, ip = 0;
  memset(table, 0, sizeof(table));
  if (n < LZ4_MIN_LENGTH) goto last_literals;

  table[HASHP(0)] = 0;
  ip = 1;
  uint32_t fwd_h = HASHP(1);
  for (;;) {
    int m;
    uint8_t* tok;
    { /* find a match: skip heuristic, 64 probes per step increment */
      int fip = ip, step = 1, nb = 1 << 6;
      for (;;) {
        uint32_t h = fwd_h;
        int cur = fip;
        m = (int)table[h];
        ip = fip;
        fip += step;
        step = (nb++ >> 6);
        if (fip > mflimit_plus_one) goto last_literals;
        fwd_h = HASHP(fip);
        table[h] = (uint32_t)cur;
        if (!by_u16 && m + MAX_DISTANCE < cur) continue;
        if (rd32(src + m) == rd32(src + ip)) break;
      }
    }
    /* catch up */
    while (ip > anchor && m > 0 && src[ip - 1] == src[m - 1]) { ip--; m--; }
    { /* literals */
      int lit = ip - anchor;
      tok = op++;
      if (limited && op + lit + (2 + 1 + LASTLITERALS) + lit / 255 > oend) return 0;
      if (lit >= (int)RUN_MASK) { *tok = (uint8_t)(RUN_MASK << 4); op = put_len(op, lit - (int)RUN_MASK); }
      else *tok = (uint8_t)(lit << 4);
      memcpy(op, src + anchor, (size_t)lit);
      op += lit;
    }
  next_match:
    *op++ = (uint8_t)(ip - m);
    *op++ = (uint8_t)((ip - m) >> 8);
    {
      int mc = count_eq(src, ip + MINMATCH, m + MINMATCH, matchlimit);
      ip += mc + MINMATCH;
      if (limited && op + (1 + LASTLITERALS) + (mc + 240) / 255 > oend) return 0;
      if (mc >= (int)ML_MASK) { *tok += ML_MASK; op = put_len(op, mc - (int)ML_MASK); }
      else *tok += (uint8_t)mc;
    }
    anchor = ip;
    if (ip >= mflimit_plus_one) break;
    table[HASHP(ip - 2)] = (uint32_t)(ip - 2);
    { /* test next position */
      uint32_t h = HASHP(ip);
      m = (int)table[h];
      table[h] = (uint32_t)ip;
      if ((by_u16 || m + MAX_DISTANCE >= ip) && rd32(src + m) == rd32(src + ip)) {
        tok = op++;
        *tok = 0;
        goto next_match;
      }
    }
    ip++;
    fwd_h = HASHP(ip);
  }
last_literals: {
    int last = n - anchor;
    if (limited && op + last + 1 + (last + 255 - (int)RUN_MASK) / 255 > oend) return 0;
    if (last >= (int)RUN_MASK) { *op++ = (uint8_t)(RUN_MASK << 4); op = put_len(op, last - (int)RUN_MASK); }
    else *op++ = (uint8_t)(last << 4);
    memcpy(op, src + anchor, (size_t)last);
    op += last;
  }
#undef HASHP
  return (int)(op - dst);
}

/* ------------------------------------------------------------------------------------------
 * Decoder: LZ4_decompress_safe (LZ4JNI.c:216) and LZ4_decompress_fast (LZ4JNI.c:169).
 * SURVEY.md Appendix C.  liblz4 1.9.3 decodes in three tiers (a "fast loop" while >= 64 bytes
 * of output room remain, a two-stage shortcut, and the fully checked path); which malformed
 * streams are accepted and the negative code -(input position)-1 depend on the tier, so the
 * tiers' CHECKS are restated here.  Copies are exact-length (byte-forward), which yields the
 * same dst[0..ret) as liblz4's wild copies on every accepted stream.
 * ---------------------------------------------------------------------------------------- */
#define FASTLOOP_SAFE_DISTANCE 64
#define WILDCOPYLENGTH 8
#define MATCH_SAFEGUARD_DISTANCE 12

static void copy_match(uint8_t* dst, int op, int offset, int length) {
  if (offset == 0) { memset(dst + op, 0, (size_t)length); return; } /* liblz4: zero-fills (see tests) */
  const uint8_t* m = dst + op - offset;
  uint8_t* d = dst + op;
  for (int i = 0; i < length; i++) d[i] = m[i];
}

/* safe != 0: LZ4_decompress_safe semantics (iend known). safe == 0: LZ4_decompress_fast
 * (trusts input; `in_cap` < 0 means unbounded like liblz4, otherwise reads are refused at
 * in_cap and a negative code is returned -- the memory-safe variant the HIP engine implements). */
static int decode_generic(const uint8_t* src, int src_size, uint8_t* dst, int out_size, int safe, int in_cap) {
  int ip = 0, op = 0;
  const int iend = src_size, oend = out_size;
  const int shortiend = iend - (safe ? 14 : 8) - 2;
  const int shortoend = oend - (safe ? 14 : 8) - 18;
  unsigned token;
  int length, offset, cpy;
  if (out_size < 0) return -1;
#define NEED_IN(k) do { if (!safe && in_cap >= 0 && ip + (k) > in_cap) goto output_error; } while (0)
  if (safe && out_size == 0) return (src_size == 1 && src[0] == 0) ? 0 : -1;
  if (!safe && out_size == 0) { NEED_IN(1); return src[0] == 0 ? 1 : -1; }
  if (safe && src_size == 0) return -1;

  if (oend - op >= FASTLOOP_SAFE_DISTANCE) {
    for (;;) { /* ---- tier 1: fast loop ---- */
      NEED_IN(1);
      token = src[ip++];
      length = (int)(token >> 4);
      if (length == (int)RUN_MASK) {
        /* read_variable_length(initial_check = loop_check = safe, limit iend-15); a limit hit
         * inside the loop is NOT an error here: the partial length is used (1.9.3 behaviour) */
        if (safe && ip >= iend - (int)RUN_MASK) goto output_error;
        for (;;) {
          NEED_IN(1);
          unsigned s = src[ip++];
          length += (int)s;
          if (safe && ip >= iend - (int)RUN_MASK) break;
          if (s != 255) break;
        }
        if (length < 0) goto output_error;
        cpy = op + length;
        if (safe) { if (cpy > oend - 32 || ip + length > iend - 32) goto safe_literal_copy; }
        else      { if (cpy > oend - 8) goto safe_literal_copy; }
        NEED_IN(length);
        memcpy(dst + op, src + ip, (size_t)length);
        ip += length; op = cpy;
      } else {
        cpy = op + length;
        if (safe && ip > iend - (16 + 1)) goto safe_literal_copy;
        NEED_IN(length);
        memcpy(dst + op, src + ip, (size_t)length);
        ip += length; op = cpy;
      }
      NEED_IN(2);
      offset = rd16(src + ip); ip += 2;
      length = (int)(token & ML_MASK);
      if (length == (int)ML_MASK) {
        if (safe && offset > op) goto output_error;
        for (;;) { /* read_variable_length(loop_check = safe, limit iend-4) */
          NEED_IN(1);
          unsigned s = src[ip++];
          length += (int)s;
          if (safe && ip >= iend - LASTLITERALS + 1) goto output_error;
          if (s != 255) break;
        }
        if (length < 0) goto output_error;
        length += MINMATCH;
        if (op + length >= oend - FASTLOOP_SAFE_DISTANCE) goto safe_match_copy;
      } else {
        length += MINMATCH;
        if (op + length >= oend - FASTLOOP_SAFE_DISTANCE) goto safe_match_copy;
        if (!safe || offset <= op) {
          if (offset >= 8) {
            if (!safe && in_cap >= 0 && offset > op) goto output_error; /* bounded variant never reads before dst */
            copy_match(dst, op, offset, length); op += length; continue;
          }
        }
      }
      if (safe && offset > op) goto output_error;
      if (!safe && in_cap >= 0 && offset > op) goto output_error;
      copy_match(dst, op, offset, length);
      op += length;
    }
  }

  for (;;) { /* ---- tiers 2+3: shortcut and fully checked path ---- */
    NEED_IN(1);
    token = src[ip++];
    length = (int)(token >> 4);
    if ((safe ? length != (int)RUN_MASK : length <= 8) && (safe ? ip < shortiend : 1) && op <= shortoend) {
      NEED_IN(length + 2);
      memcpy(dst + op, src + ip, (size_t)length);
      op += length; ip += length;
      length = (int)(token & ML_MASK);
      offset = rd16(src + ip); ip += 2;
      if (length != (int)ML_MASK && offset >= 8 && (!safe || offset <= op)) {
        if (!safe && in_cap >= 0 && offset > op) goto output_error;
        copy_match(dst, op, offset, length + MINMATCH);
        op += length + MINMATCH;
        continue;
      }
      goto copy_match_label;
    }
    if (length == (int)RUN_MASK) {
      if (safe && ip >= iend - (int)RUN_MASK) goto output_error;
      for (;;) {
        NEED_IN(1);
        unsigned s = src[ip++];
        length += (int)s;
        if (safe && ip >= iend - (int)RUN_MASK) break;
        if (s != 255) break;
      }
      if (length < 0) goto output_error;
    }
    cpy = op + length;
  safe_literal_copy:
    if ((safe && (cpy > oend - MFLIMIT || ip + length > iend - (2 + 1 + LASTLITERALS))) ||
        (!safe && cpy > oend - WILDCOPYLENGTH)) {
      if (!safe && cpy != oend) goto output_error;
      if (safe && (ip + length != iend || cpy > oend)) goto output_error;
      NEED_IN(length);
      memmove(dst + op, src + ip, (size_t)length);
      ip += length; op += length;
      break;
    }
    NEED_IN(length);
    memcpy(dst + op, src + ip, (size_t)length);
    ip += length; op = cpy;
    NEED_IN(2);
    offset = rd16(src + ip); ip += 2;
    length = (int)(token & ML_MASK);
  copy_match_label:
    if (length == (int)ML_MASK) {
      for (;;) {
        NEED_IN(1);
        unsigned s = src[ip++];
        length += (int)s;
        if (safe && ip >= iend - LASTLITERALS + 1) goto output_error;
        if (s != 255) break;
      }
      if (length < 0) goto output_error;
    }
    length += MINMATCH;
  safe_match_copy:
    if (safe && offset > op) goto output_error;
    if (!safe && in_cap >= 0 && offset > op) goto output_error;
    cpy = op + length;
    if (cpy > oend - MATCH_SAFEGUARD_DISTANCE) {
      if (cpy > oend - LASTLITERALS) goto output_error;
    }
    copy_match(dst, op, offset, length);
    op = cpy;
  }
  return safe ? op : ip;
output_error:
  return -ip - 1;
#undef NEED_IN
}

int lz4o_decompress_safe(const uint8_t* src, int src_len, uint8_t* dst, int cap) {
  return decode_generic(src, src_len, dst, cap, 1, -1);
}
int lz4o_decompress_fast(const uint8_t* src, uint8_t* dst, int dst_len) {
  return decode_generic(src, 0, dst, dst_len, 0, -1);
}
int lz4o_decompress_fast_bounded(const uint8_t* src, int src_cap, uint8_t* dst, int dst_len) {
  return decode_generic(src, 0, dst, dst_len, 0, src_cap < 0 ? 0 : src_cap);
}

/* ------------------------------------------------------------------------------------------
 * XXH32 / XXH64 (xxHash 0.6.5), reached from XXHashJNI.c:54,78 / :164,188; the reference's
 * exact in-tree restatements are src/build/source_templates/xxhash32_hash.template:27-83 and
 * xxhash64_hash.template:27-102, primes XXHashConstants.java:22-32.  SURVEY.md Appendix D.
 * ---------------------------------------------------------------------------------------- */
#define P32_1 2654435761u
#define P32_2 2246822519u
#define P32_3 3266489917u
#define P32_4 668265263u
#define P32_5 374761393u
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

uint32_t lz4o_xxh32(const uint8_t* p, int64_t len, uint32_t seed) {
  const uint8_t* const end = p + len;
  uint32_t h;
  if (len >= 16) {
    const uint8_t* const limit = end - 16;
    uint32_t v1 = seed + P32_1 + P32_2, v2 = seed + P32_2, v3 = seed, v4 = seed - P32_1;
    do {
      v1 = rotl32(v1 + rd32(p) * P32_2, 13) * P32_1; p += 4;
      v2 = rotl32(v2 + rd32(p) * P32_2, 13) * P32_1; p += 4;
      v3 = rotl32(v3 + rd32(p) * P32_2, 13) * P32_1; p += 4;
      v4 = rotl32(v4 + rd32(p) * P32_2, 13) * P32_1; p += 4;
    } while (p <= limit);
    h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
  } else {
    h = seed + P32_5;
  }
  h += (uint32_t)len;
  while (p + 4 <= end) { h = rotl32(h + rd32(p) * P32_3, 17) * P32_4; p += 4; }
  while (p < end) { h = rotl32(h + (*p) * P32_5, 11) * P32_1; p++; }
  h ^= h >> 15; h *= P32_2; h ^= h >> 13; h *= P32_3; h ^= h >> 16;
  return h;
}

#define P64_1 11400714785074694791ull
#define P64_2 14029467366897019727ull
#define P64_3 1609587929392839161ull
#define P64_4 9650029242287828579ull
#define P64_5 2870177450012600261ull
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t xxh64_round(uint64_t acc, uint64_t in) { return rotl64(acc + in * P64_2, 31) * P64_1; }
static inline uint64_t xxh64_merge(uint64_t h, uint64_t v) { h ^= xxh64_round(0, v); return h * P64_1 + P64_4; }

uint64_t lz4o_xxh64(const uint8_t* p, int64_t len, uint64_t seed) {
  const uint8_t* const end = p + len;
  uint64_t h;
  if (len >= 32) {
    const uint8_t* const limit = end - 32;
    uint64_t v1 = seed + P64_1 + P64_2, v2 = seed + P64_2, v3 = seed, v4 = seed - P64_1;
    do {
      v1 = xxh64_round(v1, rd64(p)); p += 8;
      v2 = xxh64_round(v2, rd64(p)); p += 8;
      v3 = xxh64_round(v3, rd64(p)); p += 8;
      v4 = xxh64_round(v4, rd64(p)); p += 8;
    } while (p <= limit);
    h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
    h = xxh64_merge(h, v1); h = xxh64_merge(h, v2); h = xxh64_merge(h, v3); h = xxh64_merge(h, v4);
  } else {
    h = seed + P64_5;
  }
  h += (uint64_t)len;
  while (p + 8 <= end) { h ^= xxh64_round(0, rd64(p)); h = rotl64(h, 27) * P64_1 + P64_4; p += 8; }
  if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P64_1; h = rotl64(h, 23) * P64_2 + P64_3; p += 4; }
  while (p < end) { h ^= (*p) * P64_5; h = rotl64(h, 11) * P64_1; p++; }
  h ^= h >> 33; h *= P64_2; h ^= h >> 29; h *= P64_3; h ^= h >> 32;
  return h;
}

/* ------------------------------------------------------------------------------------------
 * Synthetic block generator, SURVEY.md Appendix F (workload definition, not reference code).
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t splitmix64(uint64_t* s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

void lz4o_gen_block(uint8_t* out, int64_t n, uint64_t seed, uint64_t idx, uint32_t litmax, uint32_t win) {
  uint64_t s = seed ^ (idx * 0x9E3779B97F4A7C15ull);
  int64_t len = 0;
  while (len < n) {
    uint32_t ll = 1 + (uint32_t)(splitmix64(&s) % litmax);
    while (ll > 0) {
      uint64_t w = splitmix64(&s);
      uint32_t k = ll < 8 ? ll : 8;
      for (uint32_t j = 0; j < k; j++) { if (len < n) out[len] = (uint8_t)(w >> (8 * j)); len++; }
      ll -= k;
    }
    if (len >= 16 && len < n) {
      uint32_t ml = 4 + (uint32_t)(splitmix64(&s) % 61);
      uint64_t lim = (uint64_t)len < (uint64_t)win ? (uint64_t)len : (uint64_t)win;
      uint64_t off = 1 + splitmix64(&s) % lim;
      for (uint32_t j = 0; j < ml; j++) { if (len < n) out[len] = out[len - (int64_t)off]; len++; }
    }
  }
}

/*
 * oracle/lz4hc_oracle.c -- TEST INFRASTRUCTURE ONLY (see lz4_oracle.h).
 *
 * Plain-C restatement of LZ4_compress_HC (liblz4 1.9.3, hash-chain strategy, levels 1..9) as reached
 * from LZ4HCJNICompressor.compress through /root/reference/src/jni/net_jpountz_lz4_LZ4JNI.c:122.
 * The liblz4 sources are not in /root/reference (empty submodule src/lz4); this follows SURVEY.md
 * Appendix B and is pinned byte-for-byte against the reference's prebuilt library by
 * tests/test_oracle_hc.py.  Levels 10..12 (the optimal parser) are NOT restated: the function returns
 * -1 for them ("unsupported"), SURVEY.md section 8(f) item 3.
 *
 * State: hashTable u32[32768] (position of the latest occurrence of a 4-byte hash), chainTable
 * u16[65536] (distance to the previous occurrence, capped at 65535), both zeroed; positions are
 * biased by 64 KiB so that "0" means "no entry" (index space idx(p) = p + 65536).
 */
#include "lz4_oracle.h"
#include <stdlib.h>
#include <string.h>

#define MINMATCH 4
#define LASTLITERALS 5
#define MFLIMIT 12
#define MAXD 65535
#define BIAS 65536u
#define OPTIMAL_ML 18

static inline uint16_t rd16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

typedef struct {
  uint32_t hash[32768];
  uint16_t chain[65536];
  const uint8_t* src;       /* block start; position p <-> index p + BIAS */
  uint32_t next_to_update;  /* index */
  int nb_searches, pattern_analysis;
} hc_t;

static inline uint32_t hc_hash(const uint8_t* p) { return (rd32(p) * 2654435761u) >> 17; }

static void hc_insert(hc_t* c, uint32_t target_idx) {
  uint32_t idx = c->next_to_update;
  while (idx < target_idx) {
    const uint32_t h = hc_hash(c->src + (idx - BIAS));
    uint32_t delta = idx - c->hash[h];
    if (delta > MAXD) delta = MAXD;
    c->chain[idx & 0xFFFF] = (uint16_t)delta;
    c->hash[h] = idx;
    idx++;
  }
  c->next_to_update = target_idx;
}

static int count_fwd(const uint8_t* src, int a, int b, int limit) {
  const int s = a;
  while (a < limit && src[a] == src[b]) { a++; b++; }
  return a - s;
}

/* bytes from p on that follow the 4-byte pattern cyclically (byte 0 first), p < end */
static int count_pattern(const uint8_t* src, int p, int end, uint32_t pat) {
  const int s = p;
  while (p < end && src[p] == (uint8_t)(pat >> (8 * ((p - s) & 3)))) p++;
  return p - s;
}

/* bytes before p that follow the pattern backwards (byte 3 first), not below low */
static int reverse_count_pattern(const uint8_t* src, int p, int low, uint32_t pat) {
  const int s = p;
  while (p > low && src[p - 1] == (uint8_t)(pat >> (8 * (3 - ((s - p) & 3))))) p--;
  return s - p;
}

/* LZ4HC_InsertAndGetWiderMatch: best match for position ip that may start as early as ilow.
 * Returns the length; *mpos / *spos receive match and start positions when the result improves. */
static int wider_match(hc_t* c, int ip, int ilow, int ihigh, int longest, int* mpos, int* spos) {
  const uint8_t* src = c->src;
  const uint32_t ip_idx = (uint32_t)ip + BIAS;
  const uint32_t lowest = (BIAS + 65536u > ip_idx) ? BIAS : ip_idx - MAXD;
  const int look_back = ip - ilow;
  int attempts = c->nb_searches;
  const uint32_t pattern = rd32(src + ip);
  int repeat = 0; /* 0 untested, 1 not, 2 confirmed */
  int src_pat_len = 0;
  uint32_t mi;

  hc_insert(c, ip_idx);
  mi = c->hash[hc_hash(src + ip)];

  while (mi >= lowest && attempts > 0) {
    int ml = 0;
    attempts--;
    {
      const int mp = (int)(mi - BIAS);
      if (rd16(src + ilow + longest - 1) == rd16(src + mp - look_back + longest - 1)) {
        if (rd32(src + mp) == pattern) {
          int back = 0;
          if (look_back) {
            const int min = -((ip - ilow) < mp ? (ip - ilow) : mp); /* MAX(iMin-ip, mMin-match) */
            while (back > min && src[ip + back - 1] == src[mp + back - 1]) back--;
          }
          ml = MINMATCH + count_fwd(src, ip + MINMATCH, mp + MINMATCH, ihigh);
          ml -= back;
          if (ml > longest) {
            longest = ml;
            *mpos = mp + back;
            *spos = ip + back;
          }
        }
      }
    }
    {
      const uint32_t dist_next = c->chain[mi & 0xFFFF];
      if (c->pattern_analysis && dist_next == 1) {
        const uint32_t cand = mi - 1;
        if (repeat == 0) {
          if (((pattern & 0xFFFF) == (pattern >> 16)) & ((pattern & 0xFF) == (pattern >> 24))) {
            repeat = 2;
            src_pat_len = count_pattern(src, ip + 4, ihigh, pattern) + 4;
          } else {
            repeat = 1;
          }
        }
        if (repeat == 2 && cand >= lowest) {
          const int cp = (int)(cand - BIAS);
          if (rd32(src + cp) == pattern) {
            const int fwd = count_pattern(src, cp + 4, ihigh, pattern) + 4;
            int back = reverse_count_pattern(src, cp, 0, pattern);
            int cur;
            { /* limit back so that the segment does not start below `lowest` */
              const uint32_t far = cand - (uint32_t)back;
              back = (int)(cand - (far > lowest ? far : lowest));
            }
            cur = back + fwd;
            if (cur >= src_pat_len && fwd <= src_pat_len) {
              mi = cand + (uint32_t)fwd - (uint32_t)src_pat_len; /* best position: full pattern */
            } else {
              mi = cand - (uint32_t)back; /* farthest position of the current segment */
              if (look_back == 0) {
                const int max_ml = cur < src_pat_len ? cur : src_pat_len;
                if (longest < max_ml) {
                  if (ip_idx - mi > MAXD) break;
                  longest = max_ml;
                  *mpos = (int)(mi - BIAS);
                  *spos = ip;
                }
                {
                  const uint32_t d = c->chain[mi & 0xFFFF];
                  if (d > mi) break;
                  mi -= d;
                }
              }
            }
            continue;
          }
        }
      }
    }
    mi -= c->chain[mi & 0xFFFF];
  }
  return longest;
}

typedef struct {
  const uint8_t* src;
  uint8_t* dst;
  int op, oend, limited;
  int anchor;
} enc_t;

/* LZ4HC_encodeSequence; returns 1 on output overflow */
static int encode_sequence(enc_t* e, int* ip, int ml, int ref) {
  uint8_t* const d = e->dst;
  const int tok = e->op++;
  int length = *ip - e->anchor;
  if (e->limited && e->op + length / 255 + length + (2 + 1 + LASTLITERALS) > e->oend) return 1;
  if (length >= 15) {
    int len = length - 15;
    d[tok] = 15 << 4;
    for (; len >= 255; len -= 255) d[e->op++] = 255;
    d[e->op++] = (uint8_t)len;
  } else {
    d[tok] = (uint8_t)(length << 4);
  }
  memcpy(d + e->op, e->src + e->anchor, (size_t)length);
  e->op += length;
  d[e->op++] = (uint8_t)(*ip - ref);
  d[e->op++] = (uint8_t)((*ip - ref) >> 8);
  length = ml - MINMATCH;
  if (e->limited && e->op + length / 255 + (1 + LASTLITERALS) > e->oend) return 1;
  if (length >= 15) {
    d[tok] += 15;
    length -= 15;
    for (; length >= 255; length -= 255) d[e->op++] = 255;
    d[e->op++] = (uint8_t)length;
  } else {
    d[tok] += (uint8_t)length;
  }
  *ip += ml;
  e->anchor = *ip;
  return 0;
}

int lz4o_compress_hc(const uint8_t* src, int n, uint8_t* dst, int cap, int level) {
  static const int searches[10] = {2, 2, 2, 4, 8, 16, 32, 64, 128, 256};
  hc_t* c;
  enc_t e;
  int ip = 0, ml, ml0, ml2, ml3, ref = 0, ref0, ref2 = 0, ref3 = 0, start0, start2 = 0, start3 = 0;
  int result = 0;
  if (n < 0 || (unsigned)n > 0x7E000000u) return 0;
  if (level < 1) level = 9;
  if (level > 12) level = 12;
  if (level > 9) return -1; /* optimal parser: not restated */
  c = (hc_t*)calloc(1, sizeof(hc_t));
  if (!c) return 0;
  c->src = src;
  c->next_to_update = BIAS;
  c->nb_searches = searches[level];
  c->pattern_analysis = c->nb_searches > 128;
  e.src = src; e.dst = dst; e.op = 0; e.oend = cap; e.anchor = 0;
  e.limited = cap < lz4o_compress_bound(n);
  {
    const int mflimit = n - MFLIMIT, matchlimit = n - LASTLITERALS;
    if (n < MFLIMIT + 1) goto last_literals;
    while (ip <= mflimit) {
      { int dummy = ip; ml = wider_match(c, ip, ip, matchlimit, MINMATCH - 1, &ref, &dummy); }
      if (ml < MINMATCH) { ip++; continue; }
      start0 = ip; ref0 = ref; ml0 = ml;
    search2:
      if (ip + ml <= mflimit) ml2 = wider_match(c, ip + ml - 2, ip, matchlimit, ml, &ref2, &start2);
      else ml2 = ml;
      if (ml2 == ml) { /* no better match: encode ML1 */
        if (encode_sequence(&e, &ip, ml, ref)) goto overflow;
        continue;
      }
      if (start0 < ip) {
        if (start2 < ip + ml0) { ip = start0; ref = ref0; ml = ml0; }
      }
      if (start2 - ip < 3) { /* first match too small: removed */
        ml = ml2; ip = start2; ref = ref2;
        goto search2;
      }
    search3:
      if (start2 - ip < OPTIMAL_ML) {
        int correction, new_ml = ml;
        if (new_ml > OPTIMAL_ML) new_ml = OPTIMAL_ML;
        if (ip + new_ml > start2 + ml2 - MINMATCH) new_ml = (start2 - ip) + ml2 - MINMATCH;
        correction = new_ml - (start2 - ip);
        if (correction > 0) { start2 += correction; ref2 += correction; ml2 -= correction; }
      }
      if (start2 + ml2 <= mflimit) ml3 = wider_match(c, start2 + ml2 - 3, start2, matchlimit, ml2, &ref3, &start3);
      else ml3 = ml2;
      if (ml3 == ml2) { /* no better match: encode ML1 and ML2 */
        if (start2 < ip + ml) ml = start2 - ip;
        if (encode_sequence(&e, &ip, ml, ref)) goto overflow;
        ip = start2;
        if (encode_sequence(&e, &ip, ml2, ref2)) goto overflow;
        continue;
      }
      if (start3 < ip + ml + 3) { /* not enough space for match 2: remove it */
        if (start3 >= ip + ml) { /* Seq1 can be written now; Seq3 becomes Seq1 */
          if (start2 < ip + ml) {
            const int correction = ip + ml - start2;
            start2 += correction; ref2 += correction; ml2 -= correction;
            if (ml2 < MINMATCH) { start2 = start3; ref2 = ref3; ml2 = ml3; }
          }
          if (encode_sequence(&e, &ip, ml, ref)) goto overflow;
          ip = start3; ref = ref3; ml = ml3;
          start0 = start2; ref0 = ref2; ml0 = ml2;
          goto search2;
        }
        start2 = start3; ref2 = ref3; ml2 = ml3;
        goto search3;
      }
      /* three ascending matches: write the first one */
      if (start2 < ip + ml) {
        if (start2 - ip < OPTIMAL_ML) {
          int correction;
          if (ml > OPTIMAL_ML) ml = OPTIMAL_ML;
          if (ip + ml > start2 + ml2 - MINMATCH) ml = (start2 - ip) + ml2 - MINMATCH;
          correction = ml - (start2 - ip);
          if (correction > 0) { start2 += correction; ref2 += correction; ml2 -= correction; }
        } else {
          ml = start2 - ip;
        }
      }
      if (encode_sequence(&e, &ip, ml, ref)) goto overflow;
      ip = start2; ref = ref2; ml = ml2;
      start2 = start3; ref2 = ref3; ml2 = ml3;
      goto search3;
    }
  }
last_literals: {
    const int last = n - e.anchor;
    const int ll_add = (last + 255 - 15) / 255;
    if (e.limited && e.op + 1 + ll_add + last > e.oend) goto overflow;
    if (last >= 15) {
      int len = last - 15;
      dst[e.op++] = 15 << 4;
      for (; len >= 255; len -= 255) dst[e.op++] = 255;
      dst[e.op++] = (uint8_t)len;
    } else {
      dst[e.op++] = (uint8_t)(last << 4);
    }
    memcpy(dst + e.op, src + e.anchor, (size_t)last);
    e.op += last;
    result = e.op;
  }
overflow:
  free(c);
  return result;
}

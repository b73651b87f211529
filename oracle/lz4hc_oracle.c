/*
 * oracle/lz4hc_oracle.c -- TEST INFRASTRUCTURE ONLY (see lz4_oracle.h).
 *
 * Plain-C restatement of LZ4_compress_HC (liblz4 1.9.3: hash-chain strategy for levels 1..9, optimal
 * parser for levels 10..12) as reached
 * from LZ4HCJNICompressor.compress through /root/reference/src/jni/net_jpountz_lz4_LZ4JNI.c:122.
 * The liblz4 sources are not in /root/reference (empty submodule src/lz4); this follows SURVEY.md
 * Appendix B and is pinned byte-for-byte against the reference's prebuilt library by
 * tests/test_oracle_hc.py (all twelve levels).
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
  int nb_searches, pattern_analysis, chain_swap;
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
  uint32_t chain_pos = 0; /* matchChainPos: offset inside the current match whose chain is followed (chainSwap, optimal parser) */

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
    if (c->chain_swap && ml == longest) { /* better match => select a better chain (forward-only searches) */
      if (mi + (uint32_t)longest <= ip_idx) {
        const int k_trigger = 4;
        uint32_t dist_to_next = 1;
        const int end = longest - MINMATCH + 1;
        int step = 1, accel = 1 << k_trigger, pos;
        for (pos = 0; pos < end; pos += step) {
          const uint32_t cand_dist = c->chain[(mi + (uint32_t)pos) & 0xFFFF];
          step = (accel++ >> k_trigger);
          if (cand_dist > dist_to_next) {
            dist_to_next = cand_dist;
            chain_pos = (uint32_t)pos;
            accel = 1 << k_trigger;
          }
        }
        if (dist_to_next > 1) {
          if (dist_to_next > mi) break; /* avoid overflow */
          mi -= dist_to_next;
          continue;
        }
      }
    }
    {
      const uint32_t dist_next = c->chain[mi & 0xFFFF];
      if (c->pattern_analysis && dist_next == 1 && chain_pos == 0) {
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
    mi -= c->chain[(mi + chain_pos) & 0xFFFF];
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

/* ---- levels 10..12: LZ4HC_compress_optimal (liblz4 1.9.3 lz4hc.c) ------------------------------------------------ */
#define LZ4_OPT_NUM (1 << 12)
#define TRAILING_LITERALS 3
typedef struct { int price, off, mlen, litlen; } opt_t;

static int literals_price(int litlen) {
  int price = litlen;
  if (litlen >= 15) price += 1 + ((litlen - 15) / 255);
  return price;
}
static int sequence_price(int litlen, int mlen) {
  int price = 1 + 2; /* token + 16-bit offset */
  price += literals_price(litlen);
  if (mlen >= 15 + MINMATCH) price += 1 + ((mlen - (15 + MINMATCH)) / 255);
  return price;
}
/* LZ4HC_FindLongerMatch: forward-only search with pattern analysis and chain swap; returns the length (0: none) */
static int find_longer_match(hc_t* c, int ip, int ihigh, int min_len, int* off) {
  int mpos = 0, spos = ip;
  const int ml = wider_match(c, ip, ip, ihigh, min_len, &mpos, &spos);
  if (ml <= min_len) return 0;
  *off = ip - mpos;
  return ml;
}

static int compress_optimal(hc_t* c, enc_t* e, int n, int sufficient_len, int full_update) {
  opt_t* const opt = (opt_t*)malloc(sizeof(opt_t) * (LZ4_OPT_NUM + TRAILING_LITERALS));
  const int mflimit = n - MFLIMIT, matchlimit = n - LASTLITERALS;
  int ip = 0, overflow = 0;
  if (!opt) return 1;
  if (sufficient_len >= LZ4_OPT_NUM) sufficient_len = LZ4_OPT_NUM - 1;
  while (ip <= mflimit) {
    const int llen = ip - e->anchor;
    int best_mlen, best_off, cur, last_match_pos = 0;
    int first_off = 0;
    const int first_len = find_longer_match(c, ip, matchlimit, MINMATCH - 1, &first_off);
    if (first_len == 0) { ip++; continue; }
    if (first_len > sufficient_len) { /* good enough: immediate encoding */
      if (encode_sequence(e, &ip, first_len, ip - first_off)) { overflow = 1; break; }
      continue;
    }
    { /* prices of the first positions (literals) */
      int rpos;
      for (rpos = 0; rpos < MINMATCH; rpos++) {
        opt[rpos].mlen = 1; opt[rpos].off = 0; opt[rpos].litlen = llen + rpos; opt[rpos].price = literals_price(llen + rpos);
      }
    }
    { /* prices using the initial match */
      int mlen;
      for (mlen = MINMATCH; mlen <= first_len; mlen++) {
        opt[mlen].mlen = mlen; opt[mlen].off = first_off; opt[mlen].litlen = llen; opt[mlen].price = sequence_price(llen, mlen);
      }
    }
    last_match_pos = first_len;
    {
      int add;
      for (add = 1; add <= TRAILING_LITERALS; add++) {
        opt[last_match_pos + add].mlen = 1; opt[last_match_pos + add].off = 0; opt[last_match_pos + add].litlen = add;
        opt[last_match_pos + add].price = opt[last_match_pos].price + literals_price(add);
      }
    }
    /* check further positions */
    for (cur = 1; cur < last_match_pos; cur++) {
      const int cur_pos = ip + cur;
      int new_len, new_off = 0;
      if (cur_pos > mflimit) break;
      if (full_update) {
        /* not useful to search here if the next position has the same (or lower) cost ... unless the cost rises sharply after */
        if (opt[cur + 1].price <= opt[cur].price && opt[cur + MINMATCH].price < opt[cur].price + 3) continue;
      } else {
        if (opt[cur + 1].price <= opt[cur].price) continue;
      }
      if (full_update) new_len = find_longer_match(c, cur_pos, matchlimit, MINMATCH - 1, &new_off);
      else new_len = find_longer_match(c, cur_pos, matchlimit, last_match_pos - cur, &new_off);
      if (!new_len) continue;
      if (new_len > sufficient_len || new_len + cur >= LZ4_OPT_NUM) { /* immediate encoding */
        best_mlen = new_len; best_off = new_off; last_match_pos = cur + 1;
        goto encode;
      }
      { /* before the match: prices with literals at the beginning */
        const int base_litlen = opt[cur].litlen;
        int litlen;
        for (litlen = 1; litlen < MINMATCH; litlen++) {
          const int price = opt[cur].price - literals_price(base_litlen) + literals_price(base_litlen + litlen);
          const int pos = cur + litlen;
          if (price < opt[pos].price) {
            opt[pos].mlen = 1; opt[pos].off = 0; opt[pos].litlen = base_litlen + litlen; opt[pos].price = price;
          }
        }
      }
      { /* prices using the match at position cur */
        int ml;
        for (ml = MINMATCH; ml <= new_len; ml++) {
          const int pos = cur + ml;
          int price, ll;
          if (opt[cur].mlen == 1) {
            ll = opt[cur].litlen;
            price = ((cur > ll) ? opt[cur - ll].price : 0) + sequence_price(ll, ml);
          } else {
            ll = 0;
            price = opt[cur].price + sequence_price(0, ml);
          }
          if (pos > last_match_pos + TRAILING_LITERALS || price <= opt[pos].price) {
            if (ml == new_len && last_match_pos < pos) last_match_pos = pos;
            opt[pos].mlen = ml; opt[pos].off = new_off; opt[pos].litlen = ll; opt[pos].price = price;
          }
        }
      }
      { /* complete the following positions with literals */
        int add;
        for (add = 1; add <= TRAILING_LITERALS; add++) {
          opt[last_match_pos + add].mlen = 1; opt[last_match_pos + add].off = 0; opt[last_match_pos + add].litlen = add;
          opt[last_match_pos + add].price = opt[last_match_pos].price + literals_price(add);
        }
      }
    }
    best_mlen = opt[last_match_pos].mlen;
    best_off = opt[last_match_pos].off;
    cur = last_match_pos - best_mlen;
  encode: /* cur, last_match_pos, best_mlen, best_off are set */
    { /* reverse traversal: the shortest path */
      int candidate_pos = cur, selected_ml = best_mlen, selected_off = best_off;
      for (;;) {
        const int next_ml = opt[candidate_pos].mlen, next_off = opt[candidate_pos].off;
        opt[candidate_pos].mlen = selected_ml;
        opt[candidate_pos].off = selected_off;
        selected_ml = next_ml;
        selected_off = next_off;
        if (next_ml > candidate_pos) break; /* last match elected, first match to encode */
        candidate_pos -= next_ml;
      }
    }
    { /* encode all recorded sequences in order */
      int rpos = 0;
      while (rpos < last_match_pos) {
        const int ml = opt[rpos].mlen, offset = opt[rpos].off;
        if (ml == 1) { ip++; rpos++; continue; } /* literal */
        rpos += ml;
        if (encode_sequence(e, &ip, ml, ip - offset)) { overflow = 1; break; }
      }
      if (overflow) break;
    }
  }
  free(opt);
  return overflow;
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
  c = (hc_t*)calloc(1, sizeof(hc_t));
  if (!c) return 0;
  c->src = src;
  c->next_to_update = BIAS;
  e.src = src; e.dst = dst; e.op = 0; e.oend = cap; e.anchor = 0;
  e.limited = cap < lz4o_compress_bound(n);
  if (level > 9) { /* optimal parser: clTable {96, 64}, {512, 128}, {16384, LZ4_OPT_NUM}; level 12 = "ultra" (fullUpdate) */
    static const int nb[3] = {96, 512, 16384}, target[3] = {64, 128, LZ4_OPT_NUM};
    c->nb_searches = nb[level - 10];
    c->pattern_analysis = 1;
    c->chain_swap = 1;
    if (compress_optimal(c, &e, n, target[level - 10], level == 12)) goto overflow;
    goto last_literals;
  }
  c->nb_searches = searches[level];
  c->pattern_analysis = c->nb_searches > 128;
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

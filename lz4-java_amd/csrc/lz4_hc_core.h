// lz4_hc_core.h -- bit-exact LZ4 HC (liblz4 levels 1..9: hash-chain strategy with lazy evaluation; levels 10..12:
// optimal parser) for one wavefront per block.
//
// Replaces, for the "HIP" family, LZ4_compress_HC as reached from LZ4HCJNICompressor.compress through
// /root/reference/src/jni/net_jpountz_lz4_LZ4JNI.c:122 (liblz4 1.9.3).  Output is
// byte-identical to liblz4's.
//
// Why this is not a port.  liblz4 interleaves "insert positions into {hashTable, chainTable}" with
// "walk the chain of the current position".  But HC inserts EVERY position, in order, before it
// searches from it -- so the chain structure is a pure function of the data, not of the parse:
//     delta[p] = distance from p to the previous position with the same 4-byte hash (0 = none within 65535)
// is exactly what liblz4's chainTable holds for p, and `p - delta[p]` is what its hashTable returns
// when the search at p starts.  Two phases follow:
//   1. HcBuild   one wavefront streams the block 64 positions per step through a 128 KB LDS head
//                table (32768 x u32) and writes delta[] (2 bytes per input byte) to HBM; intra-step
//                hash collisions are resolved exactly with ballots (same scheme as the fast compressor).
//   2. HcParse   the lazy 3-match parse.  Searches are now READ-ONLY walks over delta[], so the 64
//                plain searches of a literal run (positions ip..ip+63) run in the 64 lanes at once and
//                the first lane that finds a match of >= 4 is exactly liblz4's next match; the few
//                "wider" searches of the lazy evaluation are wave-uniform scalar walks.
// Backend W: wave_dev.h on the GPU, tests/hostsim/wave_host.h in the CPU test-suite.
#pragma once
#include <stdint.h>
#include "lz4_fast_core.h"  // LZ4HIP_DEV, ctz64

#ifndef LZ4HIP_DEV
#if defined(__HIPCC__)
#define LZ4HIP_DEV __device__ __forceinline__
#define LZ4HIP_DEV_NOINLINE __device__ __noinline__
#else
#define LZ4HIP_DEV inline
#define LZ4HIP_DEV_NOINLINE inline
#endif
#endif
#ifndef LZ4HIP_DEV_NOINLINE
#if defined(__HIPCC__)
#define LZ4HIP_DEV_NOINLINE __device__ __noinline__
#else
#define LZ4HIP_DEV_NOINLINE inline
#endif
#endif

namespace lz4hip {

constexpr uint32_t HC_BIAS = 65536u;  // liblz4 biases HC indices by 64 KiB so that 0 means "no entry"
constexpr int HC_MAXD = 65535;

LZ4HIP_DEV uint32_t hc_rd32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
LZ4HIP_DEV uint32_t hc_rd16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
LZ4HIP_DEV uint64_t hc_rd64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

// nbSearches per level (liblz4 clTable); level already clamped to 1..12
LZ4HIP_DEV int hc_nb_searches(int level) {
  if (level >= 10) return level == 10 ? 96 : (level == 11 ? 512 : 16384);
  return level <= 2 ? 2 : (1 << (level - 1));
}
constexpr int HC_OPT_NUM = 1 << 12;          // LZ4_OPT_NUM: positions priced by one run of the optimal parser
constexpr int HC_OPT_TRAILING = 3;           // TRAILING_LITERALS
constexpr int HC_OPT_INTS = 4 * (HC_OPT_NUM + HC_OPT_TRAILING);  // per-block workspace of the optimal parser, in ints

// ---------------------------------------------------------------------------------------------------
// phase 1: delta[] builder (one wavefront, 128 KB LDS head table)
// ---------------------------------------------------------------------------------------------------
template <class W>
struct HcBuild {
  using VU = typename W::VU;
  using VU64 = typename W::VU64;
  using VB = typename W::VB;

  LZ4HIP_DEV static void run(W& w, const uint8_t* src, uint32_t n, uint16_t* delta) {
    if (n < 4u) return;
    const uint32_t last = n - 4u;  // last position whose 4-byte hash is readable
    w.template lds_fill<true>(32768u, 0u);
    w.sync();
    const VU j = w.lane();
    // The input words of a step do not depend on the table: they are requested one group of U steps ahead, so the only
    // latency left in a step is the LDS round trip (one wavefront per CU runs here -- the 128 KB head table -- and nothing
    // else would hide a global load).
    constexpr uint32_t U = 4u;
    VU xn[U];
    for (uint32_t u = 0; u < U; u++) xn[u] = w.ldu32(src, W::vmin(j + u * 64u, last));
    for (uint32_t g0 = 0; g0 <= last; g0 += 64u * U) {
      VU x[U];
      for (uint32_t u = 0; u < U; u++) x[u] = xn[u];
      for (uint32_t u = 0; u < U; u++) xn[u] = w.ldu32(src, W::vmin(j + (g0 + 64u * U + u * 64u), last));
      for (uint32_t u = 0; u < U; u++) {
        const uint32_t p0 = g0 + u * 64u;
        if (p0 > last) break;
        const VU p = j + p0;
        const VB valid = p <= last;
        const VU idx = p + HC_BIAS;
        const VU h = (x[u] * 2654435761u) >> 17;
        // ONE LDS round trip per step: the atomic max returns the bucket's value right before this lane's insert.  The lanes of
        // an instruction execute in some order; if the lanes sharing a bucket happened to go in rising position order, every
        // returned value already is the previous occurrence.  If not, the lane that went too late got back a position ABOVE its
        // own -- that flags the bucket, and its group is put in order with ballots below.
        const VU old = w.template lds_max<true>(h, idx, valid);
        VU prev = old;
        uint64_t pend = w.ballot(valid & (old > idx));
        while (pend) {
          const int d = ctz64(pend);
          const uint32_t hd = w.bcast(h, d);
          const VB grp = valid & (h == hd);
          const uint64_t gm = w.ballot(grp);
          const VU64 lower = w.lanemask_lt() & VU64(gm);
          const VB has = grp & (lower != VU64(0));
          const VU srcl = VU(63u) - W::clz64(lower);
          // the group's first lane takes what the bucket held before this step: the value the lane that executed first got back
          const uint64_t fm = w.ballot(grp & (old < p0 + HC_BIAS));
          const uint32_t before = w.bcast(old, ctz64(fm));
          prev = W::select(has, w.template shfl_e<true>(idx, srcl), W::select(grp, VU(before), prev));
          pend &= ~gm;
        }
        const VU dist = idx - prev;
        w.st16(delta, p, W::select(dist > (uint32_t)HC_MAXD, VU(0u), dist), valid);
        w.sync();
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------
// read-only chain search (LZ4HC_InsertAndGetWiderMatch on the static delta[]), scalar per caller
// ---------------------------------------------------------------------------------------------------
struct HcSearch {
  const uint8_t* src;
  const uint16_t* delta;
  int nb_searches;
  bool pattern_analysis;
  bool chain_swap = false;  // levels 10..12 (LZ4HC_FindLongerMatch): a match as long as the best so far re-anchors the walk on the
                            // position inside it whose chain link is the longest

  LZ4HIP_DEV static int count_fwd(const uint8_t* s, int a, int b, int limit) {
    const int st = a;
    while (a + 8 <= limit) {
      const uint64_t x = hc_rd64(s + a) ^ hc_rd64(s + b);
      if (x) return a - st + (int)(__builtin_ctzll(x) >> 3);
      a += 8; b += 8;
    }
    while (a < limit && s[a] == s[b]) { a++; b++; }
    return a - st;
  }
  // bytes from p on that follow the 4-byte pattern cyclically (byte 0 first), p < end
  LZ4HIP_DEV static int count_pattern(const uint8_t* s, int p, int end, uint32_t pat) {
    const int st = p;
    while (p < end && s[p] == (uint8_t)(pat >> (8 * ((p - st) & 3)))) p++;
    return p - st;
  }
  // bytes before p that follow the pattern backwards (byte 3 first), not below low
  LZ4HIP_DEV static int reverse_count_pattern(const uint8_t* s, int p, int low, uint32_t pat) {
    const int st = p;
    while (p > low && s[p - 1] == (uint8_t)(pat >> (8 * (3 - ((st - p) & 3))))) p--;
    return st - p;
  }
  // liblz4's chainTable entry of index mi
  LZ4HIP_DEV uint32_t chain(uint32_t mi) const {
    const uint32_t d = delta[mi - HC_BIAS];
    return d ? d : (uint32_t)HC_MAXD;
  }

  // best match for position ip that may start as early as ilow; returns the length, updates mpos/spos
  LZ4HIP_DEV int wider(int ip, int ilow, int ihigh, int longest, int& mpos, int& spos) const {
    const uint32_t ip_idx = (uint32_t)ip + HC_BIAS;
    const uint32_t lowest = (HC_BIAS + 65536u > ip_idx) ? HC_BIAS : ip_idx - (uint32_t)HC_MAXD;
    const int look_back = ip - ilow;
    int attempts = nb_searches;
    const uint32_t pattern = hc_rd32(src + ip);
    int repeat = 0;  // 0 untested, 1 not, 2 confirmed
    int src_pat_len = 0;
    uint32_t mi;
    {  // what hashTable[hash(ip)] holds when the search starts: the previous occurrence of ip's hash
      const uint32_t d0 = delta[ip];
      mi = d0 ? ip_idx - d0 : 0u;
    }
    uint32_t chain_pos = 0;  // liblz4's matchChainPos
    while (mi >= lowest && attempts > 0) {
      attempts--;
      const int mp = (int)(mi - HC_BIAS);
      int ml = 0;
      // the link to the next node depends on mi alone: requested here, together with the candidate's bytes, so a hop that
      // fails its compare (most of them) costs one memory round trip, not two
      const uint32_t dist_next = chain(mi);
      const uint32_t c32 = hc_rd32(src + mp);   // (likewise: not behind the 2-byte pre-check)
      if (hc_rd16(src + ilow + longest - 1) == hc_rd16(src + mp - look_back + longest - 1)) {
        if (c32 == pattern) {
          int back = 0;
          if (look_back) {
            const int mn = -((ip - ilow) < mp ? (ip - ilow) : mp);
            while (back > mn && src[ip + back - 1] == src[mp + back - 1]) back--;
          }
          ml = 4 + count_fwd(src, ip + 4, mp + 4, ihigh) - back;
          if (ml > longest) { longest = ml; mpos = mp + back; spos = ip + back; }
        }
      }
      if (chain_swap && ml == longest) {  // forward-only searches: pick the chain of the position with the longest link
        if (mi + (uint32_t)longest <= ip_idx) {
          uint32_t dist_to_next = 1;
          const int end = longest - 4 + 1;
          int step = 1, accel = 1 << 4;
          for (int pos = 0; pos < end; pos += step) {
            const uint32_t cd = chain(mi + (uint32_t)pos);
            step = (accel++ >> 4);
            if (cd > dist_to_next) { dist_to_next = cd; chain_pos = (uint32_t)pos; accel = 1 << 4; }
          }
          if (dist_to_next > 1) {
            if (dist_to_next > mi) break;  // avoid overflow
            mi -= dist_to_next;
            continue;
          }
        }
      }
      if (pattern_analysis && dist_next == 1u && chain_pos == 0u) {
        const uint32_t cand = mi - 1u;
        if (repeat == 0) {
          if (((pattern & 0xFFFFu) == (pattern >> 16)) & ((pattern & 0xFFu) == (pattern >> 24))) {
            repeat = 2;
            src_pat_len = count_pattern(src, ip + 4, ihigh, pattern) + 4;
          } else {
            repeat = 1;
          }
        }
        if (repeat == 2 && cand >= lowest) {
          const int cp = (int)(cand - HC_BIAS);
          if (hc_rd32(src + cp) == pattern) {
            const int fwd = count_pattern(src, cp + 4, ihigh, pattern) + 4;
            int back = reverse_count_pattern(src, cp, 0, pattern);
            {
              const uint32_t far = cand - (uint32_t)back;
              back = (int)(cand - (far > lowest ? far : lowest));
            }
            const int cur = back + fwd;
            if (cur >= src_pat_len && fwd <= src_pat_len) {
              mi = cand + (uint32_t)fwd - (uint32_t)src_pat_len;
            } else {
              mi = cand - (uint32_t)back;
              if (look_back == 0) {
                const int max_ml = cur < src_pat_len ? cur : src_pat_len;
                if (longest < max_ml) {
                  if (ip_idx - mi > (uint32_t)HC_MAXD) break;
                  longest = max_ml;
                  mpos = (int)(mi - HC_BIAS);
                  spos = ip;
                }
                const uint32_t d = chain(mi);
                if (d > mi) break;
                mi -= d;
              }
            }
            continue;
          }
        }
      }
      mi -= chain_pos == 0u ? dist_next : chain(mi + chain_pos);
    }
    return longest;
  }
};

// ---------------------------------------------------------------------------------------------------
// phase 2: the lazy parse + sequence emission (one wavefront per block)
// ---------------------------------------------------------------------------------------------------
template <class W>
struct HcParse {
  using VU = typename W::VU;
  using VU64 = typename W::VU64;
  using VB = typename W::VB;

  W& w;
  HcSearch s;
  const uint8_t* src;
  int n;
  uint8_t* dst;
  int cap;
  bool limited;
  int op = 0, anchor = 0;

  // LZ4HC_InsertAndGetWiderMatch for the wave-uniform searches of the lazy evaluation.  s.wider() spends three dependent
  // loads per chain node (link, 2-byte pre-test, 4 bytes, then the counts); here the wave first walks the links alone (one
  // dependent load per node) and parks up to 64 node positions in its lanes, then every lane evaluates its node at once.
  // Same result as s.wider(): liblz4 keeps the FIRST node (most recent position) that beats the running `longest`, and only
  // tests a node if two bytes at offset longest-1 agree (a node that would beat `longest` can fail that when the match may
  // start before ip) -- so the nodes that could improve are replayed in chain order against the running `longest`.
  // Repeated-byte patterns (level 9's pattern analysis may then jump along the chain) take the serial walk.
  LZ4HIP_DEV int wider_wave(int ip, int ilow, int ihigh, int longest, int& mpos, int& spos) {
    const uint32_t pattern = hc_rd32(src + ip);
    const bool rep = ((pattern & 0xFFFFu) == (pattern >> 16)) & ((pattern & 0xFFu) == (pattern >> 24));
    if (s.pattern_analysis && rep) return s.wider(ip, ilow, ihigh, longest, mpos, spos);
    const uint32_t ip_idx = (uint32_t)ip + HC_BIAS;
    const uint32_t lowest = (HC_BIAS + 65536u > ip_idx) ? HC_BIAS : ip_idx - (uint32_t)HC_MAXD;
    const int look_back = ip - ilow;
    const uint8_t* const sp = src;
    int attempts = s.nb_searches;
    uint32_t mi;
    {
      const uint32_t d0 = s.delta[ip];
      mi = d0 ? ip_idx - d0 : 0u;
    }
    int L = longest;
    const VU lane = w.lane();
    while (mi >= lowest && attempts > 0) {
      // walk: up to 64 nodes
      VU node = VU(0u);
      uint32_t cnt = 0;
      while (mi >= lowest && attempts > 0 && cnt < 64u) {
        attempts--;
        node = w.set_lane(node, (int)cnt, mi - HC_BIAS);
        cnt++;
        mi -= s.chain(mi);
      }
      const int L0 = L;
      const VU64 res = w.map_lanes64v(node, lane < cnt, [&](uint32_t, uint32_t c, bool a) -> uint64_t {
        if (!a || hc_rd32(sp + c) != pattern) return 0ull;
        int back = 0;
        if (look_back) {
          const int mn = -((int)look_back < (int)c ? look_back : (int)c);
          while (back > mn && sp[ip + back - 1] == sp[(int)c + back - 1]) back--;
        }
        const int ml = 4 + HcSearch::count_fwd(sp, ip + 4, (int)c + 4, ihigh) - back;
        const uint32_t pre0 = (hc_rd16(sp + ilow + L0 - 1) == hc_rd16(sp + (int)c - look_back + L0 - 1)) ? 1u : 0u;
        return ((uint64_t)(uint32_t)ml << 32) | ((uint64_t)(uint32_t)(-back) << 1) | pre0;
      });
      uint64_t imp = w.ballot(W::lo32(res >> 32) > (uint32_t)L);   // nodes that would beat the `longest` this batch started with
      while (imp) {
        const int l = ctz64(imp);
        imp &= imp - 1u;
        const uint64_t r = w.bcast64(res, l);
        const int ml = (int)(uint32_t)(r >> 32);
        if (ml <= L) continue;
        const int c = (int)w.bcast(node, l);
        const bool pass = (L == L0) ? (r & 1u) != 0u : hc_rd16(sp + ilow + L - 1) == hc_rd16(sp + c - look_back + L - 1);
        if (pass) {
          const int back = -(int)((uint32_t)r >> 1);
          L = ml; mpos = c + back; spos = ip + back;
        }
      }
    }
    return L;
  }

  LZ4HIP_DEV HcParse(W& w_, const uint8_t* src_, int n_, const uint16_t* delta, uint8_t* dst_, int cap_, int level)
      : w(w_), src(src_), n(n_), dst(dst_), cap(cap_) {
    s.src = src_;
    s.delta = delta;
    s.nb_searches = hc_nb_searches(level);
    s.pattern_analysis = level >= 10 || s.nb_searches > 128;
    s.chain_swap = level >= 10;
    limited = (uint32_t)cap < (uint32_t)n + (uint32_t)n / 255u + 16u;
  }

  LZ4HIP_DEV void put_run(uint32_t o, uint32_t len) {  // 255-run encoding of len (>= 0): 255,...,255,rem
    const uint32_t cnt = len / 255u + 1u, rem = len - 255u * (cnt - 1u);
    for (uint32_t base = 0; base < cnt; base += 64u) {
      const VU i = w.lane() + base;
      w.st8(dst, i + o, W::select(i == cnt - 1u, VU(rem), VU(255u)), i < cnt);
    }
  }

  // LZ4HC_encodeSequence; returns true on output overflow
  LZ4HIP_DEV bool encode(int& ip, int ml, int ref) {
    const uint32_t lit = (uint32_t)(ip - anchor), mc = (uint32_t)(ml - 4);
    const uint32_t nlx = lit >= 15u ? (lit - 15u) / 255u + 1u : 0u;
    const uint32_t nmx = mc >= 15u ? (mc - 15u) / 255u + 1u : 0u;
    if (limited) {
      if ((uint64_t)op + 1u + lit / 255u + lit + (2u + 1u + 5u) > (uint64_t)cap) return true;
      if ((uint64_t)op + 1u + nlx + lit + 2u + mc / 255u + (1u + 5u) > (uint64_t)cap) return true;
    }
    const uint32_t token = ((lit < 15u ? lit : 15u) << 4) | (mc < 15u ? mc : 15u);
    const uint32_t o = (uint32_t)op;
    w.st8(dst, VU(o), VU(token), w.lane() == 0u);
    if (nlx) put_run(o + 1u, lit - 15u);
    w.copy(dst, o + 1u + nlx, src, (uint32_t)anchor, lit);
    const uint32_t o2 = o + 1u + nlx + lit, offset = (uint32_t)(ip - ref);
    w.st8(dst, w.lane() + o2, W::select(w.lane() == 0u, VU(offset & 255u), VU(offset >> 8)), w.lane() < 2u);
    if (nmx) put_run(o2 + 2u, mc - 15u);
    op = (int)(o2 + 2u + nmx);
    ip += ml;
    anchor = ip;
    return false;
  }

  // liblz4's `ml = FindBestMatch(ip); if (ml < 4) { ip++; continue; }` loop, 64 positions at a time:
  // returns the first position >= ip (and <= mflimit) whose plain search yields >= 4, or -1.
  LZ4HIP_DEV int first_match(int ip, int mflimit, int matchlimit, int& ml, int& ref) {
    while (ip <= mflimit) {
      const HcSearch& sr = s;
      // Stage 1 (a filter, no effect on the result): a lane whose MOST RECENT chain node already matches 4 bytes has a match
      // of >= 4 for sure, so no lane behind the first such lane can be the answer -- and those are exactly the lanes with the
      // long walks (they sit inside the coming match, where every 4-gram has been seen before).  Only lanes up to it search.
      const uint64_t quick = w.ballot(w.map_lanes64([&](uint32_t l) -> uint64_t {
        const int p = ip + (int)l;
        if (p > mflimit) return 0ull;
        const uint32_t d = sr.delta[p];
        if (d == 0u) return 0ull;
        return hc_rd32(sr.src + p - (int)d) == hc_rd32(sr.src + p) ? 1ull : 0ull;   // (d <= 65535 and p - d >= 0 by construction)
      }) != VU64(0));
      const uint32_t lmax = quick ? (uint32_t)ctz64(quick) : 63u;
      // (handing the search of lane lmax to the whole wave -- wider_wave -- was measured: +13 % on 64 KiB blocks, -6 % on 1 MiB blocks)
      const VU64 res = w.map_lanes64([&](uint32_t l) -> uint64_t {
        const int p = ip + (int)l;
        if (p > mflimit || l > lmax) return 0ull;
        int mp = 0, sp = p;
        const int m = sr.wider(p, p, matchlimit, 3, mp, sp);
        return m >= 4 ? (((uint64_t)(uint32_t)m << 32) | (uint32_t)mp) : 0ull;
      });
      const uint64_t hits = w.ballot(res != VU64(0));
      if (hits) {
        const int l = ctz64(hits);
        const uint64_t r = w.bcast64(res, l);
        ml = (int)(r >> 32);
        ref = (int)(uint32_t)r;
        return ip + l;
      }
      ip += 64;
    }
    return -1;
  }

  // ---- levels 10..12: liblz4's optimal parser (LZ4HC_compress_optimal, lz4-java levels 10..17) -----------------------------
  // From a position with a match, the prices (output bytes) of reaching each of the next <= 4096 positions are relaxed with
  // every match found on the way, then the cheapest path is walked backwards and its sequences are written.  The price table
  // lives in a per-block workspace in HBM (`opt`: 4 ints per position -- price, offset, match length, literal length); the
  // table walk is wave-uniform scalar work (all lanes redundantly, like the lazy parse of levels 1..9), the searches are
  // s.wider() with liblz4's pattern analysis and chain swap, the sequence writer is the cooperative encode() above.
  LZ4HIP_DEV int last_literals() {
    const uint32_t last = (uint32_t)(n - anchor);
    const uint32_t ll_add = (last + 255u - 15u) / 255u;
    if (limited && (uint64_t)op + 1u + ll_add + last > (uint64_t)cap) return 0;
    const uint32_t o = (uint32_t)op;
    w.st8(dst, VU(o), VU((last < 15u ? last : 15u) << 4), w.lane() == 0u);
    if (last >= 15u) put_run(o + 1u, last - 15u);
    w.copy(dst, o + 1u + ll_add, src, (uint32_t)anchor, last);
    return (int)(o + 1u + ll_add + last);
  }

  LZ4HIP_DEV static int lit_price(int litlen) { return litlen >= 15 ? litlen + 1 + (litlen - 15) / 255 : litlen; }
  LZ4HIP_DEV static int seq_price(int litlen, int mlen) { return 3 + lit_price(litlen) + (mlen >= 19 ? 1 + (mlen - 19) / 255 : 0); }
  // LZ4HC_FindLongerMatch: forward-only search; length (0: nothing longer than min_len) and offset
  LZ4HIP_DEV int find_longer(int ip, int matchlimit, int min_len, int& off) const {
    int mp = 0, sp = ip;
    const int m = s.wider(ip, ip, matchlimit, min_len, mp, sp);
    if (m <= min_len) return 0;
    off = ip - mp;
    return m;
  }

  LZ4HIP_DEV int run_opt(int level, int* opt) {
    const int mflimit = n - 12, matchlimit = n - 5;
    int sufficient = level == 10 ? 64 : (level == 11 ? 128 : HC_OPT_NUM);
    if (sufficient >= HC_OPT_NUM) sufficient = HC_OPT_NUM - 1;
    const bool full_update = level >= 12;
    int* const price = opt;                                        // opt[p].price
    int* const o_off = opt + (HC_OPT_NUM + HC_OPT_TRAILING);        // opt[p].off
    int* const o_ml = opt + 2 * (HC_OPT_NUM + HC_OPT_TRAILING);     // opt[p].mlen (1 = literal)
    int* const o_ll = opt + 3 * (HC_OPT_NUM + HC_OPT_TRAILING);     // opt[p].litlen
    int ip = 0;
    if (n >= 13) {
      while (ip <= mflimit) {
        const int llen = ip - anchor;
        int first_off = 0;
        const int first_len = find_longer(ip, matchlimit, 3, first_off);
        if (first_len == 0) { ip++; continue; }
        if (first_len > sufficient) {  // good enough: immediate encoding
          if (encode(ip, first_len, ip - first_off)) return 0;
          continue;
        }
        for (int r = 0; r < 4; r++) { o_ml[r] = 1; o_off[r] = 0; o_ll[r] = llen + r; price[r] = lit_price(llen + r); }
        for (int m = 4; m <= first_len; m++) { o_ml[m] = m; o_off[m] = first_off; o_ll[m] = llen; price[m] = seq_price(llen, m); }
        int last_match_pos = first_len;
        for (int a = 1; a <= HC_OPT_TRAILING; a++) {
          o_ml[last_match_pos + a] = 1; o_off[last_match_pos + a] = 0; o_ll[last_match_pos + a] = a;
          price[last_match_pos + a] = price[last_match_pos] + lit_price(a);
        }
        int best_mlen = 0, best_off = 0, cur;
        bool direct = false;
        for (cur = 1; cur < last_match_pos; cur++) {
          const int cur_pos = ip + cur;
          if (cur_pos > mflimit) break;
          if (full_update) {  // not useful to search here if the next position costs the same or less -- unless the cost rises sharply after
            if (price[cur + 1] <= price[cur] && price[cur + 4] < price[cur] + 3) continue;
          } else {
            if (price[cur + 1] <= price[cur]) continue;
          }
          int new_off = 0;
          const int new_len = find_longer(cur_pos, matchlimit, full_update ? 3 : last_match_pos - cur, new_off);
          if (!new_len) continue;
          if (new_len > sufficient || new_len + cur >= HC_OPT_NUM) {  // immediate encoding
            best_mlen = new_len; best_off = new_off; last_match_pos = cur + 1;
            direct = true;
            break;
          }
          {  // before the match: prices with literals at the beginning
            const int base_ll = o_ll[cur];
            for (int ll = 1; ll < 4; ll++) {
              const int pr = price[cur] - lit_price(base_ll) + lit_price(base_ll + ll);
              const int pos = cur + ll;
              if (pr < price[pos]) { o_ml[pos] = 1; o_off[pos] = 0; o_ll[pos] = base_ll + ll; price[pos] = pr; }
            }
          }
          {  // prices using the match at position cur
            const bool cur_is_lit = o_ml[cur] == 1;
            const int ll = cur_is_lit ? o_ll[cur] : 0;
            const int basep = cur_is_lit ? ((cur > ll) ? price[cur - ll] : 0) : price[cur];
            for (int m = 4; m <= new_len; m++) {
              const int pos = cur + m;
              const int pr = basep + seq_price(ll, m);
              if (pos > last_match_pos + HC_OPT_TRAILING || pr <= price[pos]) {
                if (m == new_len && last_match_pos < pos) last_match_pos = pos;
                o_ml[pos] = m; o_off[pos] = new_off; o_ll[pos] = ll; price[pos] = pr;
              }
            }
          }
          for (int a = 1; a <= HC_OPT_TRAILING; a++) {  // complete the following positions with literals
            o_ml[last_match_pos + a] = 1; o_off[last_match_pos + a] = 0; o_ll[last_match_pos + a] = a;
            price[last_match_pos + a] = price[last_match_pos] + lit_price(a);
          }
        }
        if (!direct) {
          best_mlen = o_ml[last_match_pos];
          best_off = o_off[last_match_pos];
          cur = last_match_pos - best_mlen;
        }
        {  // reverse traversal: the cheapest path, turned into a forward list in place
          int cand = cur, sel_ml = best_mlen, sel_off = best_off;
          for (;;) {
            const int next_ml = o_ml[cand], next_off = o_off[cand];
            o_ml[cand] = sel_ml;
            o_off[cand] = sel_off;
            sel_ml = next_ml;
            sel_off = next_off;
            if (next_ml > cand) break;  // last match elected, first match to encode
            cand -= next_ml;
          }
        }
        for (int r = 0; r < last_match_pos;) {  // encode all recorded sequences in order
          const int m = o_ml[r], off = o_off[r];
          if (m == 1) { ip++; r++; continue; }  // literal
          r += m;
          if (encode(ip, m, ip - off)) return 0;
        }
      }
    }
    return last_literals();
  }

  LZ4HIP_DEV int run() {
    const int mflimit = n - 12, matchlimit = n - 5;
    int ip = 0, ml = 0, ml0, ml2, ml3, ref = 0, ref0, ref2 = 0, ref3 = 0, start0, start2 = 0, start3 = 0;
    if (n >= 13) {
      for (;;) {
        ip = first_match(ip, mflimit, matchlimit, ml, ref);
        if (ip < 0) break;
        start0 = ip; ref0 = ref; ml0 = ml;
      search2:
        if (ip + ml <= mflimit) ml2 = wider_wave(ip + ml - 2, ip, matchlimit, ml, ref2, start2);
        else ml2 = ml;
        if (ml2 == ml) {  // no better match: encode ML1
          if (encode(ip, ml, ref)) return 0;
          continue;
        }
        if (start0 < ip) {
          if (start2 < ip + ml0) { ip = start0; ref = ref0; ml = ml0; }
        }
        if (start2 - ip < 3) {  // first match too small: removed
          ml = ml2; ip = start2; ref = ref2;
          goto search2;
        }
      search3:
        if (start2 - ip < 18) {
          int new_ml = ml;
          if (new_ml > 18) new_ml = 18;
          if (ip + new_ml > start2 + ml2 - 4) new_ml = (start2 - ip) + ml2 - 4;
          const int correction = new_ml - (start2 - ip);
          if (correction > 0) { start2 += correction; ref2 += correction; ml2 -= correction; }
        }
        if (start2 + ml2 <= mflimit) ml3 = wider_wave(start2 + ml2 - 3, start2, matchlimit, ml2, ref3, start3);
        else ml3 = ml2;
        if (ml3 == ml2) {  // no better match: encode ML1 and ML2
          if (start2 < ip + ml) ml = start2 - ip;
          if (encode(ip, ml, ref)) return 0;
          ip = start2;
          if (encode(ip, ml2, ref2)) return 0;
          continue;
        }
        if (start3 < ip + ml + 3) {  // not enough space for match 2: remove it
          if (start3 >= ip + ml) {   // Seq1 can be written now; Seq3 becomes Seq1
            if (start2 < ip + ml) {
              const int correction = ip + ml - start2;
              start2 += correction; ref2 += correction; ml2 -= correction;
              if (ml2 < 4) { start2 = start3; ref2 = ref3; ml2 = ml3; }
            }
            if (encode(ip, ml, ref)) return 0;
            ip = start3; ref = ref3; ml = ml3;
            start0 = start2; ref0 = ref2; ml0 = ml2;
            goto search2;
          }
          start2 = start3; ref2 = ref3; ml2 = ml3;
          goto search3;
        }
        // three ascending matches: write the first one
        if (start2 < ip + ml) {
          if (start2 - ip < 18) {
            if (ml > 18) ml = 18;
            if (ip + ml > start2 + ml2 - 4) ml = (start2 - ip) + ml2 - 4;
            const int correction = ml - (start2 - ip);
            if (correction > 0) { start2 += correction; ref2 += correction; ml2 -= correction; }
          } else {
            ml = start2 - ip;
          }
        }
        if (encode(ip, ml, ref)) return 0;
        ip = start2; ref = ref2; ml = ml2;
        start2 = start3; ref2 = ref3; ml2 = ml3;
        goto search3;
      }
    }
    return last_literals();
  }
};

}  // namespace lz4hip

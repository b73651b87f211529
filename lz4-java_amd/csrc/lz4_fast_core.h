// lz4_fast_core.h -- wave-parallel, bit-exact LZ4 fast compressor (one 64-lane wavefront per block).
//
// Replaces, for the "HIP" family, what LZ4JNICompressor.compress reaches through the JNI shim:
// LZ4_compress_default (/root/reference/src/jni/net_jpountz_lz4_LZ4JNI.c:75), liblz4 1.9.3,
// acceleration 1, noDict; byU16 table for n < 65547, byU32 otherwise.
// Output is byte-identical to liblz4's.
//
// Why this is not a port: liblz4 walks one position at a time through a 16 KB table of positions.
// Here a wavefront evaluates 64 probe positions per step:
//   * the probe positions of a miss-run are a closed-form sequence of the run start (the skip
//     heuristic "64 probes per step increment"), so lane j computes its own position;
//   * the LDS table stores {position, 16-bit fingerprint of the 4 bytes at that position}; a lane is
//     a *tentative* hit iff the fingerprints agree, so candidate bytes (in HBM/L2, never staged) are
//     only fetched for the first tentative lane -- one global read per sequence instead of one per
//     probe;
//   * lanes up to the first tentative hit commit their inserts with one LDS atomic-max (positions
//     only ever grow, so "max" == "latest insert wins", independent of lane order inside the
//     instruction).  The value returned by the atomic reveals intra-step bucket collisions; those
//     (rare) steps are undone and resolved exactly with ballots;
//   * the two post-match table operations of liblz4 (insert ip-2, probe ip) become two extra lanes
//     in front of the next run's probes;
//   * catch-up, match length and the token/length/literal/offset stream are produced with ballots
//     and per-lane byte stores.
//
// The code is written against a "wave" backend W (wave_dev.h on the GPU; tests/hostsim/wave_host.h
// is a lock-step 64-lane simulator used ONLY by the CPU test-suite to run this very source).
// Rule: per-lane values have type W::VU / VU64 / VB; control flow branches only on wave-uniform
// scalars (ballots, broadcasts).
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

LZ4HIP_DEV int ctz64(uint64_t x) { return __builtin_ctzll(x); }
LZ4HIP_DEV int popc64(uint64_t x) { return __builtin_popcountll(x); }

struct FastStats {  // optional counters (host simulator / profiling kernel); the product kernel passes nullptr
  uint64_t steps, slow_steps, false_pos, sequences;
  uint64_t t[8];  // shader-clock cycles per phase of a step (profiling kernel only), see run()
};

template <class W, bool U16>
struct FastCore {
  using VU = typename W::VU;
  using VU64 = typename W::VU64;
  using VB = typename W::VB;
  using E = typename W::template Entry<U16>::S;   // scalar table entry (uint32_t / uint64_t)
  using VE = typename W::template Entry<U16>::V;  // per-lane table entry
  static constexpr int HLOG = U16 ? 13 : 12;
  static constexpr int PSHIFT = U16 ? 16 : 32;
  static constexpr uint32_t MAXD = 65535u;

  W& w;
  const uint8_t* src;
  uint32_t n;
  uint8_t* dst;
  uint32_t cap;
  bool limited;
  uint32_t anchor = 0, op = 0;
  uint32_t mfl1, matchlimit;  // mflimitPlusOne = n-11, matchlimit = n-5
  FastStats* st;

  LZ4HIP_DEV FastCore(W& w_, const uint8_t* s, uint32_t n_, uint8_t* d, uint32_t cap_, FastStats* st_ = nullptr)
      : w(w_), src(s), n(n_), dst(d), cap(cap_), st(st_) {
    limited = cap < n + n / 255u + 16u;
    mfl1 = n - 11u;
    matchlimit = n - 5u;
  }

  // ---- table entry helpers ---------------------------------------------------------------
  LZ4HIP_DEV static VE mk_entry(VU pos, VU fp) {
    if constexpr (U16) return (pos << 16) | fp;
    else return (W::u64(pos) << 32) | W::u64(fp);
  }
  LZ4HIP_DEV static VU e_pos(VE e) {
    if constexpr (U16) return e >> 16;
    else return W::lo32(e >> 32);
  }
  LZ4HIP_DEV static VU e_fp(VE e) {
    if constexpr (U16) return e & 0xFFFFu;
    else return W::lo32(e) & 0xFFFFu;
  }
  LZ4HIP_DEV static uint32_t se_pos(E e) { return (uint32_t)(e >> PSHIFT); }

  // probe k of a miss-run starting at S sits at S + g(k): liblz4's `step = searchMatchNb++ >> 6`
  LZ4HIP_DEV static VU g(VU k) {
    VU T = k + 62u;
    VU M = T >> 6;
    return W::select(k >= 1u, VU(1u), VU(0u)) + 32u * M * (M - 1u) + M * (T - 64u * M + 1u);
  }

  // ---- small output helpers ---------------------------------------------------------------
  LZ4HIP_DEV static uint32_t ext_count(uint32_t len) { return len >= 15u ? (len - 15u) / 255u + 1u : 0u; }

  // writes `cnt` bytes of the 255-run encoding of (len-15) at dst[o..): 255,...,255,rem
  LZ4HIP_DEV void put_ext(uint32_t o, uint32_t len, uint32_t cnt) {
    const uint32_t rem = (len - 15u) - 255u * (cnt - 1u);
    for (uint32_t base = 0; base < cnt; base += 64u) {
      VU i = w.lane() + base;
      w.st8(dst, i + o, W::select(i == cnt - 1u, VU(rem), VU(255u)), i < cnt);
    }
  }

  // last literals: token + run + raw bytes; returns total size or 0
  LZ4HIP_DEV uint32_t emit_last() {
    const uint32_t last = n - anchor;
    if (limited && (uint64_t)op + last + 1u + (last + 255u - 15u) / 255u > cap) return 0;
    const uint32_t nlx = ext_count(last);
    w.st8(dst, VU(op), VU((last < 15u ? last : 15u) << 4), w.lane() == 0u);
    if (nlx) put_ext(op + 1u, last, nlx);
    w.copy(dst, op + 1u + nlx, src, anchor, last);
    return op + 1u + nlx + last;
  }

  // ---- match extension ------------------------------------------------------------------------
  // tail of a forward count: lane granularity ran into `limit`; at most 7 bytes are left to compare
  LZ4HIP_DEV uint32_t count_tail(uint32_t pa_t, uint32_t pb_t, uint32_t limit) {
    const uint32_t tail = pa_t < limit ? limit - pa_t : 0u;
    if (tail == 0) return 0;
    const VB act = w.lane() < tail;
    const VU ca = w.ld8(src, w.lane() + pa_t, act);
    const VU cb = w.ld8(src, w.lane() + pb_t, act);
    const uint64_t bad = w.ballot(act & (ca != cb));
    return bad ? (uint32_t)ctz64(bad) : tail;
  }

  // number of equal bytes src[a+i]==src[b+i], a+i < limit (b < a); 8 bytes per lane, 512 per round
  LZ4HIP_DEV uint32_t count_fwd(uint32_t a, uint32_t b, uint32_t limit) {
    uint32_t cnt = 0;
    for (;;) {
      const VU off = w.lane() * 8u + cnt;
      const VU pa = off + a;
      const VB full = pa + 8u <= limit;
      const VU64 x = w.ldu64(src, W::vmin(pa, n - 8u)) ^ w.ldu64(src, W::vmin(off + b, n - 8u));
      const VB diff = full & (x != VU64(0));
      const uint64_t dm = w.ballot(diff);
      const uint64_t stop = dm | w.ballot(!full);
      if (stop == 0) { cnt += 512u; continue; }
      const int f = ctz64(stop);
      cnt += 8u * (uint32_t)f;
      if ((dm >> f) & 1u) return cnt + (uint32_t)(ctz64(w.bcast64(x, f)) >> 3);
      return cnt + count_tail(a + cnt, b + cnt, limit);
    }
  }

  // catch-up beyond the first 64 bytes (rare): equal bytes before (ip, m), bounded by maxback
  LZ4HIP_DEV uint32_t count_back(uint32_t ip, uint32_t m, uint32_t maxback) {
    uint32_t back = 0;
    while (back < maxback) {
      const VU jj = w.lane() + back;
      const VB act = jj < maxback;
      const VU ca = w.ld8(src, (ip - 1u) - jj, act);
      const VU cb = w.ld8(src, (m - 1u) - jj, act);
      const uint64_t am = w.ballot(act);
      const uint64_t bad = w.ballot(act & (ca != cb));
      if (bad) return back + (uint32_t)ctz64(bad);
      back += (uint32_t)popc64(am);
    }
    return back;
  }

  // ---- deferred emission ---------------------------------------------------------------------
  // A sequence found in step t is written out in step t+1, AFTER step t+1 has issued its own
  // candidate loads: the token/literal/offset stores then overlap the HBM/L2 latency of those loads.
  struct Pending {
    bool have = false, check_lits = false, regs = false;
    uint32_t lit = 0, mc = 0, offset = 0, anchor = 0;
  };
  Pending pend;
  VU pend_bytes;  // regs: lane i holds the literal byte that output byte i of the sequence needs

  LZ4HIP_DEV bool emit_pending() {
    if (!pend.have) return true;
    pend.have = false;
    const uint32_t lit = pend.lit, mc = pend.mc, offset = pend.offset;
    const uint32_t nlx = ext_count(lit), nmx = ext_count(mc);
    if (limited) {
      if (pend.check_lits && (uint64_t)op + 1u + lit + (2u + 1u + 5u) + lit / 255u > cap) return false;
      if ((uint64_t)op + 1u + nlx + lit + 2u + (1u + 5u) + (mc + 240u) / 255u > cap) return false;
    }
    const uint32_t token = ((lit < 15u ? lit : 15u) << 4) | (mc < 15u ? mc : 15u);
    const uint32_t total = 1u + nlx + lit + 2u + nmx;
    if (total <= 64u) {  // common case: one output byte per lane, a single store instruction
      const VU i = w.lane();
      const uint32_t lit0 = 1u + nlx, off0 = lit0 + lit;
      VU b;
      if (pend.regs) b = pend_bytes;
      else b = w.ld8(src, i + (pend.anchor - lit0), (i >= lit0) & (i < off0));
      b = W::select(i == 0u, VU(token), b);
      if (nlx) {
        const uint32_t rem = (lit - 15u) - 255u * (nlx - 1u);
        b = W::select((i >= 1u) & (i < lit0), W::select(i == nlx, VU(rem), VU(255u)), b);
      }
      b = W::select(i == off0, VU(offset & 255u), b);
      b = W::select(i == off0 + 1u, VU(offset >> 8), b);
      if (nmx) {
        const uint32_t rem = (mc - 15u) - 255u * (nmx - 1u);
        b = W::select(i >= off0 + 2u, W::select(i == total - 1u, VU(rem), VU(255u)), b);
      }
      w.st8(dst, i + op, b, i < total);
    } else {
      w.st8(dst, VU(op), VU(token), w.lane() == 0u);
      if (nlx) put_ext(op + 1u, lit, nlx);
      w.copy(dst, op + 1u + nlx, src, pend.anchor, lit);
      const uint32_t o2 = op + 1u + nlx + lit;
      w.st8(dst, w.lane() + o2, W::select(w.lane() == 0u, VU(offset & 255u), VU(offset >> 8)), w.lane() < 2u);
      if (nmx) put_ext(o2 + 2u, mc, nmx);
    }
    op += total;
    return true;
  }

  // ---- the compressor ------------------------------------------------------------------------
  LZ4HIP_DEV uint32_t run() {
    if (n == 0) {
      if (limited && cap == 0) return 0;
      w.st8(dst, VU(0u), VU(0u), w.lane() == 0u);
      return 1;
    }
    if (n < 13u) return emit_last();

    // every bucket starts as {pos 0, fp(bytes at 0)}: liblz4's zeroed table makes position 0 the
    // candidate of an empty bucket, and its explicit first insert (position 0) is then implied.
    uint32_t fp0;
    {
      const uint32_t x0 = w.sld32(src, 0);
      if constexpr (U16) fp0 = ((x0 * 2654435761u) >> 3) & 0xFFFFu;
      else fp0 = (x0 * 2654435761u) >> 16;
    }
    w.template lds_fill<U16>(1u << HLOG, (E)fp0);
    w.sync();

    bool post = false;       // step kind: false = run probes only; true = {insert ip-2, probe ip, run from ip+1}
    uint32_t S = 1, r = 0;   // run start, index of the first run probe of this step
    uint32_t ip = 0;         // post-match position (== anchor) when post
    const VU j = w.lane();
    const VU o8 = j * 8u;

    for (;;) {
      if (st) st->steps++;
      uint64_t tk = st ? w.tick(0u) : 0;
#define LZ4HIP_PHASE(i, dep) do { if (st) { const uint64_t t_ = w.tick(dep); st->t[i] += t_ - tk; tk = t_; } } while (0)
      // ---- [1] positions of this step's 64 slots ----
      const uint32_t nspecial = post ? 2u : 0u;
      const VB isrun = j >= nspecial;
      const VU k = j - nspecial + r;
      VU prun, pnext;
      if (r + 64u - nspecial <= 65u) { prun = k + S; pnext = prun + 1u; }  // probes 0..65 of a run are consecutive
      else { prun = g(k) + S; pnext = g(k + 1u) + S; }
      const VU pos = W::select(isrun, prun, W::select(j == 0u, VU(ip - 2u), VU(ip)));
      const VB valid = (!isrun) | (pnext <= mfl1);

      // ---- [2] input window, hash, fingerprint (invalid lanes read a clamped, harmless address) ----
      VU x32, h, fp;
      if constexpr (U16) {
        x32 = w.ldu32(src, W::vmin(pos, n - 8u));
        const VU prod = x32 * 2654435761u;
        h = prod >> (32 - HLOG);
        fp = (prod >> 3) & 0xFFFFu;
      } else {
        const VU64 x64 = w.ldu64(src, W::vmin(pos, n - 8u));
        x32 = W::lo32(x64);
        h = W::lo32(((x64 << 24) * 889523592379ull) >> (64 - HLOG));
        fp = (x32 * 2654435761u) >> 16;
      }
      LZ4HIP_PHASE(0, w.bcast(h, 0));   // t[0]: positions + input window arrived + hash
      // ---- [3] table lookup, tentative hits, commit ----
      const VE e = w.template lds_rdu<U16>(h);
      const VE newe = mk_entry(pos, fp);
      const VB probe = valid & (isrun | (j == 1u));  // lane 0 of a post step only inserts
      VB tent = probe & (e_fp(e) == fp);
      if constexpr (!U16) tent = tent & (e_pos(e) + MAXD >= pos);

      const uint64_t tmask = w.ballot(tent);
      const uint64_t imask = w.ballot(!valid);
      uint32_t k0 = tmask ? (uint32_t)ctz64(tmask) : 64u;
      const uint32_t kinv = imask ? (uint32_t)ctz64(imask) : 64u;
      uint32_t ncommit = (k0 + 1u < kinv) ? k0 + 1u : kinv;
      bool have_hit = k0 < kinv;
      VB inrange = j < ncommit;
      LZ4HIP_PHASE(1, ncommit);          // t[1]: table read + ballots
      const VE old = w.template lds_max<U16>(h, newe, inrange);

      // ---- [4] speculative candidate fetch: verify + forward + backward extension in ONE round trip ----
      uint32_t hpos = 0, mpos = 0, maxback = 0;
      bool hit_post = false;
      VU64 fa, fb;   // bytes at hpos+8*lane / mpos+8*lane (kept apart: xor-ing here would wait for the loads)
      VU ba, bb;     // bytes before hpos / mpos
      if (have_hit) {
        hpos = w.bcast(pos, (int)k0);
        mpos = se_pos(w.template bcast_e<U16>(e, (int)k0));
        hit_post = post && k0 == 1u;
        maxback = hit_post ? 0u : ((hpos - anchor) < mpos ? (hpos - anchor) : mpos);
        fa = w.ldu64(src, W::vmin(o8 + hpos, n - 8u));
        fb = w.ldu64(src, W::vmin(o8 + mpos, n - 8u));
        if (maxback) {
          const VB bact = j < maxback;
          ba = w.ldu8(src, W::select(bact, (hpos - 1u) - j, VU(0u)));
          bb = w.ldu8(src, W::select(bact, (mpos - 1u) - j, VU(0u)));
        }
      }

      LZ4HIP_PHASE(2, hpos);             // t[2]: commit issue + candidate-fetch issue
      // ---- [5] write out the previous sequence while those loads are in flight ----
      if (!emit_pending()) return 0;
      LZ4HIP_PHASE(3, op);               // t[3]: emission of the previous sequence

      // ---- [6] intra-step bucket collisions (rare): undo, resolve exactly, commit again ----
      const uint64_t det = w.ballot(inrange & (old != e));
      if (det) {
        if (st) st->slow_steps++;
        w.template lds_wr<U16>(h, e, inrange);
        w.sync();
        VE se = e;  // candidate each lane sees under sequential semantics
        uint64_t pendm = det;
        while (pendm) {
          const int d = ctz64(pendm);
          const uint32_t hd = w.bcast(h, d);
          const VB grp = inrange & (h == hd);
          const uint64_t gm = w.ballot(grp);
          const VU64 lower = w.lanemask_lt() & VU64(gm);
          const VB has = grp & (lower != VU64(0));
          const VU srcl = VU(63u) - W::clz64(lower);
          const VE pe = w.template shfl_e<U16>(newe, srcl);
          se = W::select(has, pe, se);
          pendm &= ~gm;
        }
        VB tent2 = inrange & probe & (e_fp(se) == fp);
        if constexpr (!U16) tent2 = tent2 & (e_pos(se) + MAXD >= pos);
        const uint64_t t2 = w.ballot(tent2);
        const bool had = have_hit;
        const uint32_t hpos_old = hpos, mpos_old = mpos;
        if (t2) {
          k0 = (uint32_t)ctz64(t2);
          ncommit = k0 + 1u;
          have_hit = true;
        } else {
          have_hit = false;  // the tentative lane (if any) no longer matches under the sequential candidates
        }
        inrange = j < ncommit;
        (void)w.template lds_max<U16>(h, newe, inrange);
        if (have_hit) {
          hpos = w.bcast(pos, (int)k0);
          mpos = se_pos(w.template bcast_e<U16>(se, (int)k0));
          if (!had || hpos != hpos_old || mpos != mpos_old) {  // the speculation fetched the wrong candidate
            hit_post = post && k0 == 1u;
            maxback = hit_post ? 0u : ((hpos - anchor) < mpos ? (hpos - anchor) : mpos);
            fa = w.ldu64(src, W::vmin(o8 + hpos, n - 8u));
            fb = w.ldu64(src, W::vmin(o8 + mpos, n - 8u));
            if (maxback) {
              const VB bact = j < maxback;
              ba = w.ldu8(src, W::select(bact, (hpos - 1u) - j, VU(0u)));
              bb = w.ldu8(src, W::select(bact, (mpos - 1u) - j, VU(0u)));
            }
          }
        }
      }
      w.sync();  // table updates of this step are ordered before the next step's reads
      LZ4HIP_PHASE(4, (uint32_t)det);    // t[4]: atomic result + collision handling

      // ---- [7] verify the tentative hit (4 bytes) ----
      bool hit = false;
      const VU64 fx = fa ^ fb;
      if (have_hit) {
        hit = (uint32_t)w.bcast64(fx, 0) == 0u;
        if (!hit && st) st->false_pos++;
      }
      LZ4HIP_PHASE(5, (uint32_t)hit);    // t[5]: wait for the candidate bytes
      if (!hit) {
        if (!have_hit && kinv < 64u && kinv == ncommit) return emit_last();  // liblz4's `goto _last_literals`
        // continue the run after the last committed lane
        if (post) { S = ip + 1u; r = ncommit >= 2u ? ncommit - 2u : 0u; post = false; }
        else r += ncommit;
        continue;
      }

      // ---- [8] a match at hpos against mpos: catch-up, length ----
      if (st) st->sequences++;
      uint32_t back = 0;
      if (maxback) {
        const uint64_t bad = w.ballot((j < maxback) & (ba != bb));
        if (bad) back = (uint32_t)ctz64(bad);
        else if (maxback <= 64u) back = maxback;
        else back = 64u + count_back(hpos - 64u, mpos - 64u, maxback - 64u);
      }
      uint32_t cnt;  // equal bytes from hpos on (>= 4)
      {
        const VB full = o8 + (hpos + 8u) <= matchlimit;
        const VU64 xz = W::select(j == 0u, fx & VU64(0xFFFFFFFF00000000ull), fx);
        const uint64_t dm = w.ballot(full & (xz != VU64(0)));
        const uint64_t stop = dm | w.ballot(!full);
        if (stop == 0) {
          cnt = 512u + count_fwd(hpos + 512u, mpos + 512u, matchlimit);
        } else {
          const int f = ctz64(stop);
          cnt = 8u * (uint32_t)f;
          if ((dm >> f) & 1u) cnt += (uint32_t)(ctz64(w.bcast64(xz, f)) >> 3);
          else if (cnt >= 4u) cnt += count_tail(hpos + cnt, mpos + cnt, matchlimit);
          else cnt = 4u + count_tail(hpos + 4u, mpos + 4u, matchlimit);  // lane 0 itself straddles the limit
        }
      }
      const uint32_t mc = back + (cnt - 4u);
      const uint32_t mip = hpos - back;
      pend.have = true;
      pend.lit = mip - anchor;
      pend.mc = mc;
      pend.offset = hpos - mpos;
      pend.anchor = anchor;
      pend.check_lits = !hit_post;
      // literals straight from this step's window registers: lane l (>= 1) of a post step that
      // started its run here sits on position anchor + l - 1
      pend.regs = post && r == 0u && (1u + ext_count(pend.lit) + pend.lit + 2u + ext_count(mc) <= 64u);
      if (pend.regs) {
        const VU b0 = x32 & 0xFFu;
        pend_bytes = pend.lit >= 15u ? w.shfl_up1(b0) : b0;
      }
      ip = mip + 4u + mc;
      anchor = ip;
      if (ip >= mfl1) {
        if (!emit_pending()) return 0;
        return emit_last();
      }
      post = true;
      S = ip + 1u;
      r = 0;
      LZ4HIP_PHASE(6, ip);               // t[6]: catch-up + match length + bookkeeping
    }
#undef LZ4HIP_PHASE
  }
};

}  // namespace lz4hip

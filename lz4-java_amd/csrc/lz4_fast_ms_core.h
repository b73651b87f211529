// lz4_fast_ms_core.h -- window-parallel, bit-exact LZ4 fast compressor: EVERY sequence that starts inside a
// 64-position window is resolved in one step (lz4_fast_core.h resolves one sequence per step).
//
// Same contract as lz4_fast_core.h: byte-identical to LZ4_compress_default of liblz4 1.9.3
// (/root/reference/src/jni/net_jpountz_lz4_LZ4JNI.c:75; algorithm: SURVEY.md App. A).
//
// One wavefront per block sits at ~1 wave per SIMD (the 32 KB table caps residency), so a step costs its
// DEPENDENT instruction chain, scalar instructions included.  The design therefore keeps per-sequence work out of
// the scalar stream and does it once per window across lanes:
//   1. 64 lanes sit on 64 consecutive positions; each hashes its position and reads its bucket {position,
//      16-bit fingerprint} from the LDS table AS IT WAS BEFORE THE STEP;
//   2. every lane whose fingerprint agrees fetches its own candidate (24 bytes: 8 before, 16 from the candidate on)
//      and decides by itself: 4-byte verify, forward length up to 16, backward length up to 8.  The first tentative
//      lane additionally gets a wave-wide 512-byte compare (as in lz4_fast_core.h), so the first long match of a
//      window costs no extra round trip;
//   3. a minimal scalar walk links the sequences: first verified lane from `cur` -> lane + length = the lane of
//      the match end e -> liblz4's two post-match table operations (insert e-2, probe e) are the lanes at e-2 and e
//      of the SAME window -> continue from there.  Lanes inside a match are never probed and never inserted, exactly
//      like the positions liblz4 skips.  Literal lengths, catch-up, match codes and offsets of all linked sequences
//      are then computed in one vector pass, each sequence in the lane where it was found;
//   4. the lanes liblz4 would have inserted (probed lanes + the e-2 lanes) commit with ONE LDS atomic-max.  Step 2
//      used pre-step buckets; that is exact unless two committing lanes share a bucket, which the values returned
//      by the atomic reveal.  Everything decided before the first affected probing lane v stands; the window is cut
//      at v and the next step starts there, where the table (now holding the lanes before v) gives the exact candidate;
//   5. all sequences of the window are written by the NEXT step, after it has issued its loads: output offsets by a
//      wave prefix sum, tokens / lengths / offsets by the lanes that found the sequences, all literal bytes of all
//      sequences with ONE scattered byte store straight from the window registers.
#pragma once
#include "lz4_fast_core.h"

namespace lz4hip {

template <class W, bool U16>
struct FastCoreMS : FastCore<W, U16, DirectOut<W>> {
  using Base = FastCore<W, U16, DirectOut<W>>;
  using VU = typename W::VU;
  using VU64 = typename W::VU64;
  using VB = typename W::VB;
  using E = typename Base::E;
  using VE = typename Base::VE;
  using Base::w; using Base::out; using Base::src; using Base::n; using Base::anchor; using Base::mfl1;
  using Base::matchlimit; using Base::st;
  static constexpr int HLOG = Base::HLOG;
  static constexpr uint32_t MAXD = Base::MAXD;

  LZ4HIP_DEV FastCoreMS(W& w_, DirectOut<W>& out_, const uint8_t* s, uint32_t n_, FastStats* st_ = nullptr) : Base(w_, out_, s, n_, st_) {}

  // ---- the window of the coming step, requested one step ahead ---------------------------------------
  VU sp_pos;
  uint64_t sp_validm = 0, sp_probem = 0;
  VU64 sp_lo, sp_hi, sp_prev;  // bytes [pos, pos+8), [pos+8, pos+16), [pos-8, pos)
  bool sp_multi = true;        // lanes l0.. sit on consecutive positions wbase, wbase+1, ...
  uint32_t sp_l0 = 0, sp_wbase = 0;

  LZ4HIP_DEV void prepare(bool post, uint32_t S, uint32_t r, uint32_t ip) {
    const VU j = w.lane();
    const uint32_t nspecial = post ? 2u : 0u;
    const VB isrun = j >= nspecial;
    const VU k = j - nspecial + r;
    VU prun;
    uint64_t vm;
    if (LZ4HIP_LIKELY(r + 64u - nspecial <= 65u)) {  // probes 0..65 of a run are consecutive
      prun = k + S;
      vm = w.ballot(prun < mfl1);                    // liblz4: forwardIp = ip + 1 <= mflimitPlusOne
      sp_multi = true;
    } else {
      prun = Base::g(k) + S;
      vm = w.ballot(Base::g(k + 1u) + S <= mfl1);
      sp_multi = false;
    }
    sp_pos = W::select(isrun, prun, W::select(j == 0u, VU(ip - 2u), VU(ip)));
    sp_validm = vm | (uint64_t)(post ? 3u : 0u);
    sp_probem = sp_validm & ~(uint64_t)(post ? 1u : 0u);
    sp_l0 = post ? 1u : 0u;
    sp_wbase = post ? ip : S + r;
    const uint32_t top = n - 8u;  // invalid lanes read a clamped, harmless address: no exec-mask branch around the loads
    sp_lo = w.ldu64(src, W::vmin(sp_pos, top));
    sp_hi = w.ldu64(src, W::vmin(sp_pos + 8u, top));
    sp_prev = w.ldu64(src, W::vmin(W::vmax(sp_pos, 8u) - 8u, top));
  }

  // ---- sequences found by a step (each in the lane that found it), written by the next step ----------
  struct Batch {
    uint64_t cm = 0;       // lanes holding a sequence
    bool regs = false;     // literals of the window are in b0 (consecutive window)
    uint32_t l0 = 0;
    uint32_t anchor0 = 0;  // anchor of the first sequence
    uint32_t carry = 0;    // literals of the first sequence in front of the window (cb holds them when <= 64)
  };
  Batch bt;
  VU q_anchor, q_litw, q_mc, q_off;
  VU bt_b0, bt_pos, bt_cb;

  LZ4HIP_DEV bool emit_batch() {
    const uint64_t cm = bt.cm;
    if (cm == 0) return true;
    bt.cm = 0;
    const VU j = w.lane();
    const VB act = w.lanes(cm);
    const VU lit = q_litw & SEQ_LIT_MASK;
    const VU nlx = W::select(lit >= 15u, W::div255(lit - 15u) + 1u, VU(0u));
    const VU nmx = W::select(q_mc >= 15u, W::div255(q_mc - 15u) + 1u, VU(0u));
    const VU size = W::select(act, nlx + lit + nmx + 3u, VU(0u));
    const VU o = w.excl_scan(size) + out.op;  // token position of each sequence
    const uint32_t end = w.bcast(o + size, 63);
    if (LZ4HIP_UNLIKELY(out.limited)) {  // liblz4's two per-sequence output checks
      const VB c1 = ((q_litw & SEQ_NOCHECK) == 0u) & (W::u64(o) + W::u64(lit) + W::u64(W::div255(lit)) + VU64(1u + 8u) > VU64((uint64_t)out.cap));
      const VB c2 = W::u64(o) + W::u64(nlx) + W::u64(lit) + W::u64(W::div255(q_mc + 240u)) + VU64(1u + 2u + 6u) > VU64((uint64_t)out.cap);
      if (w.ballot(act & (c1 | c2))) return false;
    }
    const VU tok = (W::vmin(lit, VU(15u)) << 4) | W::vmin(q_mc, VU(15u));
    w.st8(out.dst, o, tok, act);
    w.st8(out.dst, o + 1u, lit - 15u, act & (nlx == 1u));
    const VU ld = o + nlx + 1u;  // where each literal run goes
    const VU oo = ld + lit;
    w.st8(out.dst, oo, q_off & 0xFFu, act);
    w.st8(out.dst, oo + 1u, q_off >> 8, act);
    w.st8(out.dst, oo + 2u, q_mc - 15u, act & (nmx == 1u));
    uint64_t longs = (w.ballot(nlx > 1u) | w.ballot(nmx > 1u)) & cm;  // length runs of more than one byte (rare)
    while (longs) {
      const int k = ctz64(longs);
      longs &= longs - 1u;
      const uint32_t l = w.bcast(lit, k), c = w.bcast(q_mc, k), ok_ = w.bcast(o, k), nl = w.bcast(nlx, k), nm = w.bcast(nmx, k);
      if (nl > 1u) out.put_ext(ok_ + 1u, l, nl);
      if (nm > 1u) out.put_ext(ok_ + 1u + nl + l + 2u, c, nm);
    }
    const int jf = ctz64(cm);
    if (LZ4HIP_LIKELY(bt.regs)) {
      // every window lane is a literal of at most one sequence: the next one found at or above it
      const VU nh = W::ctz64v(VU64(cm) & ~w.lanemask_lt());  // 64 above the last sequence
      const VU a_k = w.shfl(q_anchor, nh), l_k = w.shfl(lit, nh), d_k = w.shfl(ld, nh);
      const VB islit = (nh < 64u) & (j >= bt.l0) & (bt_pos >= a_k) & (bt_pos < a_k + l_k);
      w.st8(out.dst, d_k + (bt_pos - a_k), bt_b0, islit);
      if (bt.carry) {  // literals of the first sequence that precede the window
        const uint32_t d0 = w.bcast(ld, jf);
        if (LZ4HIP_LIKELY(bt.carry <= 64u)) w.st8(out.dst, j + d0, bt_cb, j < bt.carry);
        else w.copy(out.dst, d0, src, bt.anchor0, bt.carry);
      }
    } else {
      uint64_t rest = cm;
      while (rest) {
        const int k = ctz64(rest);
        rest &= rest - 1u;
        const uint32_t l = w.bcast(lit, k);
        if (l) w.copy(out.dst, w.bcast(ld, k), src, w.bcast(q_anchor, k), l);
      }
    }
    out.op = end;
    return true;
  }

  static constexpr uint32_t WIDE_MORE = 1u << 31;
  // exact forward length (>= 4) of the match (hp, mp) from the wave-wide compare words fx = src[hp+8j..] ^ src[mp+8j..]
  LZ4HIP_DEV uint32_t wide_len(uint32_t hp, uint32_t mp, VU64 fx) {
    const VU j = w.lane();
    const VU o8 = j * 8u;
    const uint64_t fullm = w.ballot(o8 + (hp + 8u) <= matchlimit);
    const VU64 xz = W::select(j == 0u, fx & VU64(0xFFFFFFFF00000000ull), fx);
    const uint64_t dm = w.ballot(xz != VU64(0)) & fullm;
    const uint64_t stop = dm | ~fullm;
    if (LZ4HIP_UNLIKELY(stop == 0)) return 512u | WIDE_MORE;  // all 512 bytes equal: the caller continues with count_fwd
    const int f = ctz64(stop);
    uint32_t cnt = 8u * (uint32_t)f;
    if (LZ4HIP_LIKELY((dm >> f) & 1u)) return cnt + (uint32_t)(ctz64(w.bcast64(xz, f)) >> 3);
    if (cnt < 4u) cnt = 4u;  // lane 0 itself straddles the limit
    return cnt + Base::count_tail(hp + cnt, mp + cnt, matchlimit);
  }

  // ---- the compressor ------------------------------------------------------------------------
  LZ4HIP_DEV uint32_t run() {
    if (n < 13u) return out.emit_last(0u);

    uint32_t fp0;
    {
      const uint32_t x0 = w.sld32(src, 0);
      if constexpr (U16) fp0 = ((x0 * 2654435761u) >> 3) & 0xFFFFu;
      else fp0 = (x0 * 2654435761u) >> 16;
    }
    w.template lds_fill<U16>(1u << HLOG, (E)fp0);  // every bucket starts as {pos 0, fp(bytes at 0)}, see lz4_fast_core.h
    w.sync();

    bool post = false;
    uint32_t S = 1, r = 0, ip = 0;
    const VU j = w.lane();
    q_anchor = VU(0u); q_litw = VU(0u); q_mc = VU(0u); q_off = VU(0u);
    bt_b0 = VU(0u); bt_pos = VU(0u); bt_cb = VU(0u);
    prepare(post, S, r, ip);

    enum { K_RUN = 0, K_POST = 1, K_DONE = 2, K_LAST = 3 };
    uint32_t pf_end = 0;  // source prefetched up to here

    for (;;) {
      if (st) st->steps++;
      uint64_t tk = st ? w.tick(0u) : 0;
#define LZ4HIP_PHASE(i, dep) do { if (st) { const uint64_t t_ = w.tick(dep); st->t[i] += t_ - tk; tk = t_; } } while (0)
      // ---- [1] window (requested one step ahead), hash, bucket as of before this step ----
      const VU pos = sp_pos;
      const uint64_t validm = sp_validm, probem = sp_probem;
      const bool multi = sp_multi;
      const uint32_t l0 = sp_l0, wbase = sp_wbase;
      const VU64 xlo = sp_lo, xhi = sp_hi, xprev = sp_prev;
      const VU x32 = W::lo32(xlo);
      VU h, fp;
      if constexpr (U16) {
        const VU prod = x32 * 2654435761u;
        h = prod >> (32 - HLOG);
        fp = (prod >> 3) & 0xFFFFu;
      } else {
        h = W::lo32(((xlo << 24) * 889523592379ull) >> (64 - HLOG));
        fp = (x32 * 2654435761u) >> 16;
      }
      LZ4HIP_PHASE(0, w.bcast(h, 0));            // t[0]: window arrived + hash
      const VE newe = Base::mk_entry(pos, fp);
      // lane 0 of a post step is liblz4's `putPosition(ip-2)`: committed BEFORE the buckets are read (a wave's LDS operations
      // execute in order), so lane 1 -- the probe of ip -- already sees it
      if (post) (void)w.template lds_max<U16>(h, newe, j == 0u);
      const VE e = w.template lds_rdu<U16>(h);
      const VU cpos = Base::e_pos(e);
      uint64_t tmask = w.ballot(Base::e_fp(e) == fp) & probem;
      if constexpr (!U16) tmask &= w.ballot(cpos + MAXD >= pos);
      LZ4HIP_PHASE(1, (uint32_t)tmask);          // t[1]: bucket read + fingerprint ballot

      // ---- [2] every tentative lane fetches its own candidate (the others re-read their window: L1 hits);
      //          the first tentative lane also gets the wave-wide 512-byte compare ----
      const uint32_t top = n - 8u;
      const VU ca = W::select(w.lanes(tmask), cpos, pos);
      const VU64 clo = w.ldu64_cand(src, W::vmin(ca, top));
      const VU64 chi = w.ldu64_cand(src, W::vmin(ca + 8u, top));
      const VU64 cprev = w.ldu64_cand(src, W::vmin(W::vmax(ca, 8u) - 8u, top));
      const int k0 = tmask ? ctz64(tmask) : 0;
      const uint32_t hp0 = w.bcast(pos, k0), mp0 = w.bcast(cpos, k0);
      const VU64 fa = w.ldu64(src, W::vmin(j * 8u + hp0, top));
      const VU64 fb = w.ldu64_cand(src, W::vmin(j * 8u + mp0, top));
      if (LZ4HIP_UNLIKELY(wbase + W::kPrefetchBytes > pf_end && pf_end < n && n >= 16u)) { w.prefetch4k(src, pf_end, n); pf_end += W::kPrefetchBytes; }  // source -> L2, one chunk ahead
      LZ4HIP_PHASE(2, (uint32_t)tmask);          // t[2]: candidate fetch issue

      // ---- [3] write the previous window's sequences while those loads are in flight ----
      if (!emit_batch()) return 0;
      LZ4HIP_PHASE(3, out.op);                   // t[3]: emission of the previous window

      // ---- [4] per-lane verdicts ----
      const VU64 x = xlo ^ clo;
      const uint64_t M = w.ballot(W::lo32(x) == 0u) & tmask;
      // lenv: bytes known equal so far; longm: verified lanes that may have more (continue with count_fwd from lenv)
      VU lenv, eqb;
      uint64_t longm, bslowm;
      {
        const VU64 y = xhi ^ chi;
        const VU len = W::select(x != VU64(0), W::ctz64v(x) >> 3, (W::ctz64v(y) >> 3) + 8u);  // 4..16 on verified lanes
        const uint64_t nearm = w.ballot(pos + 16u > matchlimit);   // matchlimit within 16 bytes: count from 4 with the exact limit
        longm = (w.ballot(len == 16u) | nearm) & M;
        lenv = W::select(w.lanes(nearm), VU(4u), len);
        eqb = W::clz64(xprev ^ cprev) >> 3;                         // equal bytes right before (position, candidate): 0..8
        bslowm = w.ballot(eqb == 8u) | w.ballot(cpos < 8u) | w.ballot(pos < 8u);  // catch-up not decidable from those 8 bytes
      }
      if ((longm >> k0) & 1u) {  // the first tentative lane has its answer in fa/fb already
        const uint32_t wl = wide_len(hp0, mp0, fa ^ fb);
        lenv = w.set_lane(lenv, k0, wl & ~WIDE_MORE);
        if (LZ4HIP_LIKELY(!(wl & WIDE_MORE))) longm &= ~(1ull << k0);
      }
      LZ4HIP_PHASE(4, (uint32_t)M + w.bcast(lenv, 0));  // t[4]: candidate bytes arrived + per-lane verdicts

      // ---- [5] link the sequences of this window.  The scalar walk only hops hit -> match end -> next hit and notes
      //          the hit lanes; which lanes lie inside matches, which insert, and every sequence's anchor follow in one vector pass ----
      const uint32_t anchor0 = anchor;
      const uint64_t inv = ~validm;
      const uint32_t kinv = inv ? (uint32_t)ctz64(inv) : 64u;  // valid lanes are a prefix
      const uint32_t wpb = wbase - l0;                         // position of lane x is wpb + x (consecutive windows)
      // a match that ends at or beyond lane `lim` ends the window: the window edge, the end of the probing range
      // (liblz4 stops at mflimitPlusOne), or -- non-consecutive window -- any match at all
      const uint32_t lim = !multi ? 0u : (mfl1 - wpb < 64u ? mfl1 - wpb : 64u);
      VU E = j + lenv;   // lane of the match end, were this lane's match taken
      uint32_t cur = l0, jlast = 0, le = 0;
      uint64_t cm = 0;   // lanes that found a sequence
      bool broke = false;
      for (;;) {
        const uint64_t hm = M & (~0ull << cur);
        if (hm == 0) break;
        jlast = (uint32_t)ctz64(hm);
        if (LZ4HIP_UNLIKELY((longm >> jlast) & 1u)) {
          const uint32_t p = w.bcast(pos, (int)jlast), c = w.bcast(cpos, (int)jlast);
          uint32_t len = w.bcast(lenv, (int)jlast);
          len += Base::count_fwd(p + len, c + len, matchlimit);
          lenv = w.set_lane(lenv, (int)jlast, len);
          E = w.set_lane(E, (int)jlast, jlast + len);
        }
        cm |= 1ull << jlast;
        le = w.bcast(E, (int)jlast);
        if (le >= lim) { broke = true; break; }
        cur = le;
      }
      int kind;
      if (broke) {
        anchor = multi ? wpb + le : w.bcast(pos, (int)jlast) + w.bcast(lenv, (int)jlast);
        kind = anchor >= mfl1 ? K_DONE : K_POST;
      } else {
        if (cm) anchor = wpb + le;
        kind = kinv < 64u ? K_LAST : K_RUN;
      }
      // vector pass: EP = match-end lane of the nearest sequence found below this lane
      const VU64 lowc = VU64(cm) & w.lanemask_lt();
      const uint64_t haspm = w.ballot(lowc != VU64(0));
      const VU EP = w.shfl(E, VU(63u) - W::clz64(lowc));
      const VU anchor_v = W::select(w.lanes(haspm), EP + wpb, VU(anchor0));   // anchor each sequence starts from
      uint64_t covered = w.ballot(EP > j) & haspm;                            // lanes inside a match: neither probed nor inserted
      if (broke) covered |= (~0ull << jlast) << 1;
      const uint64_t e2m = multi ? (w.ballot(EP == j + 2u) & haspm) : 0ull;    // lanes that only insert (liblz4's putPosition(ip-2))
      const uint32_t stop = (kind == K_LAST) ? kinv : 64u;
      uint64_t I = ((~0ull << l0) & (stop >= 64u ? ~0ull : ((1ull << stop) - 1ull)) & ~covered) | e2m;  // lanes liblz4 would insert
      LZ4HIP_PHASE(5, (uint32_t)I);              // t[5]: chain walk

      // ---- [6] commit (the returned values are examined after the next window has been requested) ----
      const VE old = w.template lds_max<U16>(h, newe, w.lanes(I));

      // ---- [7] the sequences, one per finding lane ----
      {
        const VU room = pos - anchor_v;
        const VU bmax = W::vmin(room, cpos);
        VU b = W::vmin(eqb, bmax);
        uint64_t bs = bslowm & w.ballot(bmax != 0u) & cm;  // catch-ups the 8 fetched bytes cannot decide (rare)
        while (bs) {
          const int k = ctz64(bs);
          bs &= bs - 1u;
          b = w.set_lane(b, k, Base::count_back(w.bcast(pos, k), w.bcast(cpos, k), w.bcast(bmax, k)));
        }
        q_anchor = anchor_v;
        q_litw = (room - b) | W::select(room == 0u, VU(SEQ_NOCHECK), VU(0u));  // room 0 <=> found by the post-match probe (no output check 1)
        q_mc = b + lenv - 4u;
        q_off = pos - cpos;
      }

      // ---- [8] next window ----
      bool npost = false;
      uint32_t nS = S, nr = r + 64u, nip = ip;
      if (kind == K_POST) { npost = true; nip = anchor; nS = anchor + 1u; nr = 0; }
      else if (kind == K_RUN && (post || cm != 0)) { nS = anchor + 1u; nr = 63u - (anchor - wbase + l0); }
      if (kind == K_RUN || kind == K_POST) prepare(npost, nS, nr, nip);

      // ---- [9] did two committing lanes share a bucket? ----
      const uint64_t det = w.ballot(old != e) & I;
      if (LZ4HIP_UNLIKELY(det != 0)) {
        uint64_t vict = 0, pendm = det;
        while (pendm) {
          const int d = ctz64(pendm);
          const uint32_t hd = w.bcast(h, d);
          const uint64_t gm = w.ballot(h == hd) & I;
          pendm &= ~gm;
          vict |= gm & (gm - 1ull);  // the lowest lane of a group saw the right candidate, the others should have seen the lane below
        }
        vict &= ~e2m;                // insert-only lanes do not look at their candidate
        if (vict) {
          if (st) st->slow_steps++;
          const uint32_t v = (uint32_t)ctz64(vict);
          const uint64_t below = (1ull << v) - 1ull;
          const uint64_t dropped = cm & ~below;
          if (dropped) anchor = w.bcast(q_anchor, ctz64(dropped));  // the anchor as it was when the first dropped sequence was linked
          cm &= below;
          w.template lds_wr<U16>(h, e, w.lanes(I));
          w.sync();
          I &= below;
          (void)w.template lds_max<U16>(h, newe, w.lanes(I));
          const bool he = post || cm != 0;   // a match end (== anchor) precedes lane v in this run
          const uint32_t le = anchor - wbase + l0;
          if (he && multi && v == le) { npost = true; nip = anchor; nS = anchor + 1u; nr = 0; }
          else { npost = false; nS = he ? anchor + 1u : S; nr = he ? v - le - 1u : r + v; }
          kind = npost ? K_POST : K_RUN;
          prepare(npost, nS, nr, nip);
        }
      }
      w.sync();  // table updates of this step are ordered before the next step's reads
      LZ4HIP_PHASE(6, (uint32_t)det);            // t[6]: commit + records + collision handling

      // ---- [10] hand the sequences to the next step ----
      if (st) st->sequences += (uint64_t)popc64(cm);
      bt.cm = cm;
      bt.regs = multi;
      bt.l0 = l0;
      bt.anchor0 = anchor0;
      bt.carry = 0;
      bt_b0 = x32 & 0xFFu;
      bt_pos = pos;
      if (cm && multi && anchor0 < wbase) {
        const uint32_t lit0 = w.bcast(q_litw, ctz64(cm)) & SEQ_LIT_MASK;
        const uint32_t gap = wbase - anchor0;
        bt.carry = lit0 < gap ? lit0 : gap;
        if (bt.carry && bt.carry <= 64u) bt_cb = w.ld8(src, j + anchor0, j < bt.carry);
      }
      if (kind == K_DONE || kind == K_LAST) break;
      post = npost; S = nS; r = nr; ip = nip;
      LZ4HIP_PHASE(7, (uint32_t)sp_validm);      // t[7]: hand-over
    }
#undef LZ4HIP_PHASE
    if (!emit_batch()) return 0;
    return out.emit_last(anchor);
  }
};

}  // namespace lz4hip

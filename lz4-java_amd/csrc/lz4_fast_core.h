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
// developer probe builds only: a byU16 table of 2^LZ4HIP_PROBE_HLOG entries (13 = liblz4's; anything else is NOT bit-exact and
// exists to measure residency: 12 -> 16 KB tables, ten wavefronts per CU)
#ifndef LZ4HIP_PROBE_HLOG
#define LZ4HIP_PROBE_HLOG 13
#endif
// fingerprint bits kept per byU16 table entry (16 = the {pos16, fp16} u32 entries; narrower values exist to measure the
// false-tentative-hit cost of smaller tables: any width is bit-exact, a false tentative hit is ruled out by its candidate bytes)
#ifndef LZ4HIP_FP_BITS
#define LZ4HIP_FP_BITS 16
#endif
#define LZ4HIP_FP_MASK ((1u << LZ4HIP_FP_BITS) - 1u)
#include <stdint.h>
#include <stddef.h>

#ifndef LZ4HIP_DEV
#if defined(__HIPCC__)
#define LZ4HIP_DEV __device__ __forceinline__
#else
#define LZ4HIP_DEV inline
#endif
#endif
#ifndef LZ4HIP_COLD
#if defined(__HIPCC__)
#define LZ4HIP_COLD __device__ __forceinline__   /* (out-of-line calls measured 1.6x SLOWER: stack traffic around the call sites) */
#else
#define LZ4HIP_COLD inline
#endif
#endif
#define LZ4HIP_LIKELY(x) __builtin_expect(!!(x), 1)
#define LZ4HIP_UNLIKELY(x) __builtin_expect(!!(x), 0)

namespace lz4hip {

LZ4HIP_DEV int ctz64(uint64_t x) { return __builtin_ctzll(x); }
LZ4HIP_DEV int popc64(uint64_t x) { return __builtin_popcountll(x); }

struct FastStats {  // optional counters (host simulator / profiling kernel); the product kernel passes nullptr
  uint64_t steps, slow_steps, false_pos, sequences;
  uint64_t t[8];  // shader-clock cycles per phase of a step (profiling kernel only), see run()
};

// ---------------------------------------------------------------------------------------------------
// Output policies.  The match finder (FastCore) hands every sequence to an `Out`:
//   DirectOut  writes the LZ4 stream itself; a sequence found in step t is written in step t+1, after step
//              t+1 has issued its candidate loads, so the stores overlap that latency;
//   ParkOut    (lz4_fast_v2_core.h) parks {match start, length, offset} one lane per sequence and writes 64 at a time.
// (A third policy, descriptors pushed to a ring drained by a second wavefront, measured 8 % slower than DirectOut in round 1
// and was removed.)
// ---------------------------------------------------------------------------------------------------
template <class W>
struct DirectOut {
  using VU = typename W::VU;
  using VU64 = typename W::VU64;
  using VB = typename W::VB;
  static constexpr bool kUsesWindowRegs = true;

  W& w;
  const uint8_t* src;
  uint32_t n;
  uint8_t* dst;
  uint32_t cap;
  bool limited;
  uint32_t op = 0;

  LZ4HIP_DEV DirectOut(W& w_, const uint8_t* s, uint32_t n_, uint8_t* d, uint32_t cap_) : w(w_), src(s), n(n_), dst(d), cap(cap_) {
    limited = cap < n + n / 255u + 16u;
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

  LZ4HIP_DEV void put_ext_at(uint32_t o, uint32_t len, uint32_t cnt) { put_ext(o, len, cnt); }

  // last literals: token + run + raw bytes; returns total size or 0
  LZ4HIP_COLD uint32_t emit_last(uint32_t anchor) {
    const uint32_t last = n - anchor;
    if (limited && (uint64_t)op + last + 1u + (last + 255u - 15u) / 255u > cap) return 0;
    const uint32_t nlx = ext_count(last);
    w.st8(dst, VU(op), VU((last < 15u ? last : 15u) << 4), w.lane() == 0u);
    if (nlx) put_ext(op + 1u, last, nlx);
    w.copy(dst, op + 1u + nlx, src, anchor, last);
    return op + 1u + nlx + last;
  }

  // ---- deferred emission ---------------------------------------------------------------------
  // A sequence found in step t is written out in step t+1, AFTER step t+1 has issued its own
  // candidate loads: the token/literal/offset stores then overlap the HBM/L2 latency of those loads.
  struct Pending {
    bool have = false, check_lits = false, regs = false;
    uint32_t lit = 0, mc = 0, offset = 0, anchor = 0;
  };
  Pending pend;
  VU pend_b0;  // regs: lane l (>= 1) holds literal l-1 of the pending sequence (byte 0 of its window word)

  LZ4HIP_DEV bool emit_pending() {
    if (!pend.have) return true;
    pend.have = false;
    const uint32_t lit = pend.lit, mc = pend.mc, offset = pend.offset;
    const uint32_t token = ((lit < 15u ? lit : 15u) << 4) | (mc < 15u ? mc : 15u);
    if (LZ4HIP_LIKELY(pend.regs && !limited)) {
      // The common case, branch-free, ONE store instruction, no cross-lane traffic (regs implies at most one length byte
      // each and total <= 63: no division, no loop): lane 0 writes the token; lane l >= 1 writes output byte l + nlx, so
      // the literal it needs (literal l-1) is byte 0 of its own window word; when a literal-length byte exists (nlx == 1)
      // the otherwise idle lane 63 writes it at index 1; bytes past the literals come from the 3-byte trailer word
      // {offset, ml-15}.
      const uint32_t nlx = lit >= 15u ? 1u : 0u, nmx = mc >= 15u ? 1u : 0u;
      const uint32_t off0 = 1u + nlx + lit, total = off0 + 2u + nmx;
      const uint32_t trl = offset | ((mc - 15u) << 16);  // third byte only used when nmx == 1
      const VU i = w.lane();
      const VB ext_lane = (i == 63u) & VB(nlx != 0u);
      const VU oi = W::select(ext_lane, VU(1u), W::select(i == 0u, VU(0u), i + nlx));
      const VU tb = W::shr(VU(trl), ((oi - off0) * 8u) & 31u) & 0xFFu;  // garbage for oi < off0 (never selected)
      VU b = W::select(oi >= off0, tb, pend_b0);
      b = W::select(i == 0u, VU(token), b);
      b = W::select(ext_lane, VU(lit - 15u), b);
      w.st8(dst, oi + op, b, (oi < total));
      op += total;
      return true;
    }
    return emit_checked(lit, mc, offset, token);
  }

  // everything else: a limited output buffer (liblz4's two capacity checks), literals that are not in the window registers,
  // long length runs
  LZ4HIP_COLD bool emit_checked(uint32_t lit, uint32_t mc, uint32_t offset, uint32_t token) {
    const uint32_t nlx = ext_count(lit), nmx = ext_count(mc);
    if (limited) {
      if (pend.check_lits && (uint64_t)op + 1u + lit + (2u + 1u + 5u) + lit / 255u > cap) return false;
      if ((uint64_t)op + 1u + nlx + lit + 2u + (1u + 5u) + (mc + 240u) / 255u > cap) return false;
    }
    const uint32_t total = 1u + nlx + lit + 2u + nmx;
    emit_generic(lit, mc, offset, nlx, nmx, token, total);
    op += total;
    return true;
  }

  // sequences whose literals are not in the window registers, or that need more than 63 bytes / long length runs
  LZ4HIP_COLD void emit_generic(uint32_t lit, uint32_t mc, uint32_t offset, uint32_t nlx, uint32_t nmx, uint32_t token, uint32_t total) {
    const VU i = w.lane();
    if (total <= 64u) {
      const uint32_t lit0 = 1u + nlx, off0 = lit0 + lit;
      VU b = w.ld8(src, i + (pend.anchor - lit0), (i >= lit0) & (i < off0));
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
      w.st8(dst, VU(op), VU(token), i == 0u);
      if (nlx) put_ext(op + 1u, lit, nlx);
      w.copy(dst, op + 1u + nlx, src, pend.anchor, lit);
      const uint32_t o2 = op + 1u + nlx + lit;
      w.st8(dst, i + o2, W::select(i == 0u, VU(offset & 255u), VU(offset >> 8)), i < 2u);
      if (nmx) put_ext(o2 + 2u, mc, nmx);
    }
  }


  // ---- the Out interface ----
  LZ4HIP_DEV bool overlap_point() { return emit_pending(); }
  LZ4HIP_DEV void seq(uint32_t lit, uint32_t mc, uint32_t offset, uint32_t anchor, bool check_lits, bool regs, VU b0) {
    pend.have = true;
    pend.lit = lit;
    pend.mc = mc;
    pend.offset = offset;
    pend.anchor = anchor;
    pend.check_lits = check_lits;
    pend.regs = regs;
    pend_b0 = b0;
  }
  LZ4HIP_DEV uint32_t last(uint32_t anchor) {
    if (!emit_pending()) return 0;
    return emit_last(anchor);
  }
};

// flag / mask of a packed literal length (lz4_fast_ms_core.h): bit 29 = liblz4's _next_match path (no literal-capacity check)
constexpr uint32_t SEQ_NOCHECK = 1u << 29, SEQ_LIT_MASK = (1u << 29) - 1u;

// PK (byU32 blocks of at most 4 MiB only -- the largest block of the LZ4 Frame format and its default in the reference's
// LZ4FrameOutputStream): the 4096 entries are 32 bits, {position (22 bits), fingerprint (10 bits)}, 16 KB instead of 32 -- twice the
// match-finder chains per CU (kernels.hip, compress_fast_v2w8_cu_kernel).  A narrower fingerprint only means more tentative hits that
// their candidate bytes rule out; what is accepted and what the table holds are liblz4's at any width.
template <class W, bool U16, class Out = DirectOut<W>, bool PK = false>
struct FastCore {
  static_assert(!(U16 && PK), "compact entries are a byU32 layout");
  static constexpr bool S32 = U16 || PK;           // entries are 32 bits
  static constexpr uint32_t kPackMaxN = 1u << 22;  // PK: positions have 22 bits
  using VU = typename W::VU;
  using VU64 = typename W::VU64;
  using VB = typename W::VB;
  using E = typename W::template Entry<S32>::S;   // scalar table entry (uint32_t / uint64_t)
  using VE = typename W::template Entry<S32>::V;  // per-lane table entry
  static constexpr int HLOG = U16 ? LZ4HIP_PROBE_HLOG : 12;
  static constexpr int PSHIFT = U16 ? 16 : (PK ? 10 : 32);
  static constexpr uint32_t FPM = PK ? 0x3FFu : 0xFFFFu;   // fingerprint bits of an entry
  static constexpr uint32_t MAXD = 65535u;
#ifndef LZ4HIP_SPEC_LANES
#define LZ4HIP_SPEC_LANES 16
#endif
  static constexpr uint32_t kSpecLanes = LZ4HIP_SPEC_LANES;  // lanes (x 8 bytes) of the speculative verify + extension compare

  W& w;
  Out& out;
  const uint8_t* src;
  uint32_t n;
  uint32_t anchor = 0;
  uint32_t mfl1, matchlimit;  // mflimitPlusOne = n-11, matchlimit = n-5
  FastStats* st;
  // Density probe (0 = off): if sequences 32..95 of the block cover fewer than `dense64` input bytes, the block is made of
  // short sequences -- the window-parallel core (lz4_fast_ms_core.h) is the faster one for it; loop() then stops with
  // `bailed` set and the caller leaves the block to that core.
  uint32_t dense64 = 0;
  bool bailed = false, probe_done = false, one_done = false;
  uint32_t p_S = 0, p_ip = 0;

  LZ4HIP_DEV FastCore(W& w_, Out& out_, const uint8_t* s, uint32_t n_, FastStats* st_ = nullptr)
      : w(w_), out(out_), src(s), n(n_), st(st_) {
    mfl1 = n - 11u;
    matchlimit = n - 5u;
  }

  // ---- table entry helpers ---------------------------------------------------------------
  LZ4HIP_DEV static VE mk_entry(VU pos, VU fp) {
    if constexpr (S32) return (pos << PSHIFT) | fp;
    else return (W::u64(pos) << 32) | W::u64(fp);
  }
  LZ4HIP_DEV static VU e_pos(VE e) {
    if constexpr (S32) return e >> PSHIFT;
    else return W::lo32(e >> 32);
  }
  LZ4HIP_DEV static VU e_fp(VE e) {
    if constexpr (S32) return e & FPM;
    else return W::lo32(e) & 0xFFFFu;
  }
  // byU32 fingerprint of the four bytes x (byU16 entries take theirs out of the bucket product)
  LZ4HIP_DEV static VU fp32(VU x) {
    if constexpr (PK) return ((x * 2654435761u) >> 16) & FPM;
    else return (x * 2654435761u) >> 16;
  }
  LZ4HIP_DEV static uint32_t se_pos(E e) { return (uint32_t)(e >> PSHIFT); }

  // probe k of a miss-run starting at S sits at S + g(k): liblz4's `step = searchMatchNb++ >> 6`
  LZ4HIP_DEV static VU g(VU k) {
    VU T = k + 62u;
    VU M = T >> 6;
    return W::select(k >= 1u, VU(1u), VU(0u)) + 32u * M * (M - 1u) + M * (T - 64u * M + 1u);
  }

  // ---- match extension ------------------------------------------------------------------------
  // tail of a forward count: lane granularity ran into `limit`; at most 7 bytes are left to compare
  LZ4HIP_COLD uint32_t count_tail(uint32_t pa_t, uint32_t pb_t, uint32_t limit) {
    const uint32_t tail = pa_t < limit ? limit - pa_t : 0u;
    if (tail == 0) return 0;
    const VB act = w.lane() < tail;
    const VU ca = w.ld8(src, w.lane() + pa_t, act);
    const VU cb = w.ld8(src, w.lane() + pb_t, act);
    const uint64_t bad = w.ballot(act & (ca != cb));
    return bad ? (uint32_t)ctz64(bad) : tail;
  }

  // number of equal bytes src[a+i]==src[b+i], a+i < limit (b < a); 8 bytes per lane, 512 per round
  LZ4HIP_COLD uint32_t count_fwd(uint32_t a, uint32_t b, uint32_t limit) {
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
  LZ4HIP_COLD uint32_t count_back(uint32_t ip, uint32_t m, uint32_t maxback) {
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

  // ---- per-step inputs, prepared one step ahead so the window load overlaps bookkeeping ----------
  VU sp_pos;      // position of each of the 64 slots
  uint64_t sp_validm = 0, sp_probem = 0;  // lane masks (wave-uniform scalars): slot is valid / slot probes the table
  VU64 sp_x64;    // window word at the slot position (U32 mode: 8 bytes; U16 mode: low 4 bytes used)
  VU sp_x32;

  LZ4HIP_DEV void prepare_step(bool post, uint32_t S, uint32_t r, uint32_t ip) {
    const VU j = w.lane();
    const uint32_t nspecial = post ? 2u : 0u;
    const VB sp_isrun = j >= nspecial;
    const VU k = j - nspecial + r;
    VU prun, pnext;
    if (LZ4HIP_LIKELY(r + 64u - nspecial <= 65u)) { prun = k + S; pnext = prun + 1u; }  // probes 0..65 of a run are consecutive
    else { prun = g(k) + S; pnext = g(k + 1u) + S; }
    sp_pos = W::select(sp_isrun, prun, W::select(j == 0u, VU(ip - 2u), VU(ip)));
    // special lanes (0,1 of a post step) are always valid; lane 0 of a post step only inserts
    sp_validm = w.ballot(pnext <= mfl1) | (uint64_t)(post ? 3u : 0u);
    sp_probem = sp_validm & ~(uint64_t)(post ? 1u : 0u);
    // invalid lanes read a clamped, harmless address: no exec-mask branch around the load
    if constexpr (U16) sp_x32 = w.ldu32(src, W::vmin(sp_pos, n - 8u));
    else sp_x64 = w.ldu64(src, W::vmin(sp_pos, n - 8u));
  }

  // ---- the compressor ------------------------------------------------------------------------
  LZ4HIP_DEV uint32_t run() {
    if (n < 13u) return out.last(0u);  // all literals (n == 0: the single token 0x00)

    // every bucket starts as {pos 0, fp(bytes at 0)}: liblz4's zeroed table makes position 0 the
    // candidate of an empty bucket, and its explicit first insert (position 0) is then implied.
    uint32_t fp0;
    {
      const uint32_t x0 = w.sld32(src, 0);
      if constexpr (U16) fp0 = ((x0 * 2654435761u) >> 3) & LZ4HIP_FP_MASK;
      else fp0 = ((x0 * 2654435761u) >> 16) & FPM;
    }
    w.template lds_fill<S32>(1u << HLOG, (E)fp0);
    w.sync();
    if (dense64 == 0u) return loop<0>(false, 1u, 0u, 0u);
    // density probe: a second copy of the loop counts the first 96 sequences, so the main loop stays untouched
    const uint32_t res = loop<1>(false, 1u, 0u, 0u);
    if (!probe_done) return res;   // the block ended (or ran out of output) before the probe did
    if (bailed) return 0u;
    return loop<0>(true, p_S, 0u, p_ip);
  }

  // the step loop from a given parser state (also entered mid-block by lz4_fast_ms_core.h when it hands a block over):
  //   post: step kind: false = run probes only; true = {insert ip-2, probe ip, run from ip+1}
  //   S, r: run start, index of the first run probe of this step;  ip: post-match position (== anchor) when post
  //   MODE: 0 = to the end of the block; 1 = density probe (see dense64); 2 = ONE sequence: returns 0 with `one_done` set and
  //         p_ip = the post-match position once a sequence has been handed to `out` (lz4_fast_v2_core.h uses this loop as the
  //         exact path for the steps its lean loop does not handle), or the block's result if the block ended first
  template <int MODE>
  LZ4HIP_DEV uint32_t loop(bool post, uint32_t S, uint32_t r, uint32_t ip) {
    constexpr bool PROBE = MODE == 1, ONE = MODE == 2;
    uint32_t probe_cd = 32u, probe_anchor = 0;  // (PROBE only) countdown to the next density-probe event
    uint32_t pf_end = PROBE ? 0u : (post ? ip : S) & ~(W::kPrefetchBytes - 1u);  // source prefetched up to here (LZ4HIP_PF_KB KB chunks, as far ahead)
    // the next chunk is due once hpos + kPrefetchBytes > pf_end, i.e. hpos >= pf_trig (never again once pf_end >= n): ONE compare per step
    uint32_t pf_trig = (n >= 16u && pf_end < n) ? (pf_end >= W::kPrefetchBytes ? pf_end - W::kPrefetchBytes + 1u : 0u) : 0xFFFFFFFFu;
    const VU j = w.lane();
    const VU o8 = j * 8u;
    const VU o8s = W::vmin(o8, VU(8u * (kSpecLanes - 1u)));
    prepare_step(post, S, r, ip);

    for (;;) {
      if (st) st->steps++;
      uint64_t tk = st ? w.tick(0u) : 0;
#define LZ4HIP_PHASE(i, dep) do { if (st) { const uint64_t t_ = w.tick(dep); st->t[i] += t_ - tk; tk = t_; } } while (0)
      // ---- [1] this step's slots were prepared (and their window words requested) one step ahead ----
      const VU pos = sp_pos;
      const uint64_t validm = sp_validm, probem = sp_probem;
      VU x32, h, fp;
      if constexpr (U16) {
        x32 = sp_x32;
        const VU prod = x32 * 2654435761u;
        h = prod >> (32 - HLOG);
        fp = (prod >> 3) & LZ4HIP_FP_MASK;
      } else {
        x32 = W::lo32(sp_x64);
        h = W::lo32(((sp_x64 << 24) * 889523592379ull) >> (64 - HLOG));
        fp = fp32(x32);
      }
      LZ4HIP_PHASE(0, w.bcast(h, 0));   // t[0]: input window arrived + hash
      // ---- [2] table lookup, tentative hits, commit ----
      const VE e = w.template lds_rdu<S32>(h);
      const VE newe = mk_entry(pos, fp);
      // (ballots of plain compares are free -- the compare already writes the lane mask; the masks are combined on the scalar side)
      uint64_t tmask = w.ballot(e_fp(e) == fp) & probem;
      if constexpr (!U16) tmask &= w.ballot(e_pos(e) + MAXD >= pos);
      const uint64_t imask = ~validm;
      uint32_t k0 = tmask ? (uint32_t)ctz64(tmask) : 64u;
      const uint32_t kinv = imask ? (uint32_t)ctz64(imask) : 64u;
      uint32_t ncommit = (k0 + 1u < kinv) ? k0 + 1u : kinv;
      bool have_hit = ncommit == k0 + 1u;   // == (k0 < kinv), as a 32-bit equality (the 64-bit ctz results have no scalar "<")
      uint64_t inm = ncommit >= 64u ? ~0ull : ((1ull << ncommit) - 1ull);  // lanes that commit their insert
      LZ4HIP_PHASE(1, ncommit);          // t[1]: table read + ballots
      const VE old = w.template lds_max<S32>(h, newe, w.lanes(inm));

      // ---- [3] speculative candidate fetch: verify + forward + backward extension in ONE round trip.
      // Branch-free: without a tentative hit the loads still go out (lane 0's slot vs position 0, L1 hits). ----
      const uint32_t kk = have_hit ? k0 : 0u;
      uint32_t hpos = w.bcast(pos, (int)kk);
      uint32_t mpos = se_pos(w.template bcast_e<S32>(e, (int)kk));
      bool hit_post = post && k0 == 1u;
      uint32_t maxback = (!have_hit || hit_post) ? 0u : ((hpos - anchor) < mpos ? (hpos - anchor) : mpos);
      // (only the first kSpecLanes lanes take part -- 8 bytes each: the candidate side is a random re-read of the block and
      // every further 128 bytes are one more cache line that mostly misses L2; longer matches take count_fwd's extra round trip)
      // (lanes past kSpecLanes repeat the last taking lane's address: same cache line, no exec-mask region around the loads)
      VU64 fa = w.ldu64(src, W::vmin(o8s + hpos, n - 8u));   // kept apart from fb: xor-ing here would wait for the loads
      VU64 fb = w.ldu64_cand(src, W::vmin(o8s + mpos, n - 8u));
      VU ba, bb;
      uint64_t bactm = maxback >= 64u ? ~0ull : ((1ull << maxback) - 1ull);
      {
        const VB bact = w.lanes(bactm);
        // (opaque index: keeps the {scalar base + 32-bit lane offset} form; the optimiser otherwise folds the subtraction into a
        // 64-bit address per lane)
        ba = w.ldu8(src, W::opaque(W::select(bact, (hpos - 1u) - j, VU(0u))));
        bb = w.ldu8(src, W::opaque(W::select(bact, (mpos - 1u) - j, VU(0u))));
      }
      if (LZ4HIP_UNLIKELY(hpos >= pf_trig)) {
        w.prefetch4k(src, pf_end, n);
        pf_end += W::kPrefetchBytes;
        pf_trig = pf_end < n ? pf_end - W::kPrefetchBytes + 1u : 0xFFFFFFFFu;
      }
      LZ4HIP_PHASE(2, hpos);             // t[2]: commit issue + candidate-fetch issue

      // ---- [4] write out the previous sequence while those loads are in flight ----
      if (!out.overlap_point()) return 0;
      LZ4HIP_PHASE(3, hpos);             // t[3]: emission of the previous sequence

      // ---- [5] intra-step bucket collisions (rare): undo, resolve exactly, commit again ----
      const uint64_t det = w.ballot(old != e) & inm;
      if (LZ4HIP_UNLIKELY(det != 0)) {
        if (st) st->slow_steps++;
        const VB inrange = w.lanes(inm);
        w.template lds_wr<S32>(h, e, inrange);
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
          const VE pe = w.template shfl_e<S32>(newe, srcl);
          se = W::select(has, pe, se);
          pendm &= ~gm;
        }
        uint64_t t2 = w.ballot(e_fp(se) == fp) & inm & probem;
        if constexpr (!U16) t2 &= w.ballot(e_pos(se) + MAXD >= pos);
        const bool had = have_hit;
        const uint32_t hpos_old = hpos, mpos_old = mpos;
        if (t2) {
          k0 = (uint32_t)ctz64(t2);
          ncommit = k0 + 1u;
          have_hit = true;
        } else {
          have_hit = false;  // the tentative lane (if any) no longer matches under the sequential candidates
        }
        inm = ncommit >= 64u ? ~0ull : ((1ull << ncommit) - 1ull);
        (void)w.template lds_max<S32>(h, newe, w.lanes(inm));
        if (have_hit) {
          hpos = w.bcast(pos, (int)k0);
          mpos = se_pos(w.template bcast_e<S32>(se, (int)k0));
          if (!had || hpos != hpos_old || mpos != mpos_old) {  // the speculation fetched the wrong candidate
            hit_post = post && k0 == 1u;
            maxback = hit_post ? 0u : ((hpos - anchor) < mpos ? (hpos - anchor) : mpos);
            fa = w.ldu64(src, W::vmin(o8s + hpos, n - 8u));
            fb = w.ldu64_cand(src, W::vmin(o8s + mpos, n - 8u));
            bactm = maxback >= 64u ? ~0ull : ((1ull << maxback) - 1ull);
            const VB bact = w.lanes(bactm);
            ba = w.ldu8(src, W::select(bact, (hpos - 1u) - j, VU(0u)));
            bb = w.ldu8(src, W::select(bact, (mpos - 1u) - j, VU(0u)));
          }
        }
      }
      w.sync();  // table updates of this step are ordered before the next step's reads
      LZ4HIP_PHASE(4, (uint32_t)det);    // t[4]: atomic result + collision handling

      // ---- [6] verify the tentative hit (4 bytes) ----
      const VU64 fx = fa ^ fb;
      const bool hit = have_hit & ((uint32_t)w.bcast64(fx, 0) == 0u);   // (no short circuit: one branch, and the likely side falls through)
      if (st && have_hit && !hit) st->false_pos++;
      LZ4HIP_PHASE(5, (uint32_t)hit);    // t[5]: wait for the candidate bytes
      if (LZ4HIP_UNLIKELY(!hit)) {
        if (!have_hit && kinv < 64u && kinv == ncommit) return out.last(anchor);  // liblz4's `goto _last_literals`
        // continue the run after the last committed lane
        if (post) { S = ip + 1u; r = ncommit >= 2u ? ncommit - 2u : 0u; post = false; }
        else r += ncommit;
        prepare_step(post, S, r, ip);
        continue;
      }

      // ---- [7] a match at hpos against mpos: forward length first -- it alone decides where the next step starts ----
      if (st) st->sequences++;
      uint32_t cnt;  // equal bytes from hpos on (>= 4)
      {
        constexpr uint64_t specm = (1ull << kSpecLanes) - 1ull;
        const uint64_t fullm = w.ballot(o8 + (hpos + 8u) <= matchlimit);
        const VU64 xz = W::select(j == 0u, fx & VU64(0xFFFFFFFF00000000ull), fx);
        const uint64_t dm = w.ballot(xz != VU64(0)) & fullm & specm;
        const uint64_t stop = dm | (~fullm & specm);
        if (LZ4HIP_UNLIKELY(stop == 0)) {
          cnt = 8u * kSpecLanes + count_fwd(hpos + 8u * kSpecLanes, mpos + 8u * kSpecLanes, matchlimit);
        } else {
          const int f = ctz64(stop);
          cnt = 8u * (uint32_t)f;
          if (LZ4HIP_LIKELY((dm >> f) & 1u)) cnt += (uint32_t)(ctz64(w.bcast64(xz, f)) >> 3);
          else if (cnt >= 4u) cnt += count_tail(hpos + cnt, mpos + cnt, matchlimit);
          else cnt = 4u + count_tail(hpos + 4u, mpos + 4u, matchlimit);  // lane 0 itself straddles the limit
        }
      }
      const uint32_t ip_new = hpos + cnt;
      const VU b0 = x32 & 0xFFu;         // literal bytes of this step (before the slots are re-prepared)
      const bool was_post = post;
      const bool done = ip_new >= mfl1;
      if (!done && !ONE) {                // request the next step's window NOW; the rest of the bookkeeping overlaps it
        post = true;
        S = ip_new + 1u;
        r = 0;
        ip = ip_new;
        prepare_step(true, S, 0u, ip_new);
      }
      // ---- [8] catch-up, pending-sequence record ----
      uint32_t back = 0;
      if (maxback) {
        const uint64_t bad = w.ballot(ba != bb) & bactm;
        if (bad) back = (uint32_t)ctz64(bad);
        else if (LZ4HIP_LIKELY(maxback <= 64u)) back = maxback;
        else back = 64u + count_back(hpos - 64u, mpos - 64u, maxback - 64u);
      }
      const uint32_t mc = back + (cnt - 4u);
      {
        const uint32_t lit = (hpos - back) - anchor;
        // literals straight from this step's window registers: lane l (>= 1) of a post step sits on position
        // anchor + l - 1 (its run started in this step); needs at most one length byte each and lane 63 free
        const bool regs = Out::kUsesWindowRegs && was_post && mc < 270u &&
                          (1u + (lit >= 15u ? 1u : 0u) + lit + 2u + (mc >= 15u ? 1u : 0u) <= 63u);
        out.seq(lit, mc, hpos - mpos, anchor, !hit_post, regs, b0);
      }
      anchor = ip_new;
      LZ4HIP_PHASE(6, ip_new);           // t[6]: match length + catch-up + bookkeeping
      if (done) return out.last(anchor);
      if constexpr (ONE) {
        one_done = true;
        p_ip = ip_new;
        return 0u;
      }
      if constexpr (PROBE) {
        if (--probe_cd == 0u) {
          if (probe_anchor == 0u) { probe_anchor = anchor; probe_cd = 64u; }
          else {  // sequence 96 done: the next step is a post step at ip_new
            if (!out.overlap_point()) return 0u;
            probe_done = true;
            bailed = anchor - probe_anchor < dense64;
            p_S = S; p_ip = ip;
            return 0u;
          }
        }
      }
    }
#undef LZ4HIP_PHASE
  }
};

}  // namespace lz4hip

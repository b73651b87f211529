// lz4_fast_v2_core.h -- the lean fast-compress core: a minimal match-finder chain + batched emission.
//
// Same contract as lz4_fast_core.h: byte-identical to LZ4_compress_default of liblz4 1.9.3
// (/root/reference/src/jni/net_jpountz_lz4_LZ4JNI.c:75; algorithm: SURVEY.md App. A).
//
// Why a second core: one wavefront per block is a SERIAL chain (table state depends on the parse), a CU holds five of
// them (32 KB tables), so throughput = 1 / (dependent instructions + memory round trips of one step).  lz4_fast_core.h
// spends ~200 instructions per sequence, a third of them writing the previous sequence.  Here
//   * the match finder keeps only what the chain needs.  Its loop handles the common step -- a post-match step
//     {insert ip-2, probe ip, probe ip+1 ..} whose first tentative lane verifies, matches < 256 bytes, no bucket
//     collision inside the step, far from the block end -- in ~70 instructions: catch-up compares the window registers
//     against ONE contiguous byte load of the candidate side, the forward compare is 64 lanes x 4 bytes;
//   * every other step (collision, false positive, miss-run beyond 63 probes, long match, block head and tail) is
//     UNDONE (the committed lanes write their old buckets back) and replayed by the exact generic step of
//     lz4_fast_core.h (FastCore::loop<2>, one sequence), so there is one definition of the hard cases;
//   * sequences are not written when found: {match start, match length, offset} are parked one LANE per sequence
//     (v_writelane) and every 64 sequences one vector pass sizes them (DPP prefix sum), checks liblz4's capacity
//     conditions, copies all literal runs (lane per sequence, dword pieces) and stores tokens / length bytes / offsets --
//     ~5 instructions per sequence instead of ~60, and nothing of it sits between two steps of the chain.
#pragma once
#include "lz4_fast_core.h"
#if defined(__HIPCC__)
#include "lz4_fast_v2_asm.h"
#include "lz4_fast_v2_asm32.h"
#endif

#ifndef LZ4HIP_V2_PROBE
#define LZ4HIP_V2_PROBE 0   /* profiling kernel: 0 = all phases, k + 1 = only the interval that ends at phase point k */
#endif

namespace lz4hip {

LZ4HIP_DEV int ctz32(uint32_t x) { return __builtin_ctz(x); }
LZ4HIP_DEV int clz64(uint64_t x) { return __builtin_clzll(x); }

// Output policy: sequences parked in lanes, written 64 at a time.
template <class W>
struct ParkOut {
  using VU = typename W::VU;
  using VU64 = typename W::VU64;
  using VB = typename W::VB;
  static constexpr bool kUsesWindowRegs = false;
  static constexpr uint32_t kNoCheck = 1u << 16;  // flag in p_off: liblz4's _next_match path (no literal-capacity check)
  static constexpr uint32_t kFinal = 1u << 17;    // flag in p_off (raw parking only): the entry is a finished sequence, not a raw hit
  static constexpr bool kRawPark = false;         // the lean loop does liblz4's backward extension itself and parks finished sequences
  static constexpr bool kAsmPark = false;         // (true: the hand-scheduled loop of lz4_fast_v2_asm.h parks into p_ms / p_ml / p_off itself)

  W& w;
  const uint8_t* src;
  uint32_t n;
  uint8_t* dst;
  uint32_t cap;
  bool limited;
  uint32_t op = 0;
  VU p_ms, p_ml, p_off;   // lane s: match start, match length, offset (| kNoCheck) of parked sequence s
  uint32_t cnt = 0;       // parked sequences
  uint32_t prev_end = 0;  // input position after the last written sequence (= anchor of parked sequence 0)
  bool ok = true;         // false: the output does not fit (liblz4 returns 0)
  // density probe of the adaptive scheme (0 = off): sequences 32..95 of the block cover fewer than dense64 input bytes -> `bail`
  // (the block is made of short sequences: the window-parallel core of lz4_fast_ms_core.h is the faster one and redoes it)
  uint32_t dense64 = 0, flushes = 0, mark = 0;
  bool bail = false;

  LZ4HIP_DEV ParkOut(W& w_, const uint8_t* s, uint32_t n_, uint8_t* d, uint32_t cap_) : w(w_), src(s), n(n_), dst(d), cap(cap_) {
    limited = cap < n + n / 255u + 16u;
    p_ms = VU(0u); p_ml = VU(0u); p_off = VU(0u);
  }

  LZ4HIP_DEV static uint32_t ext_count(uint32_t len) { return len >= 15u ? (len - 15u) / 255u + 1u : 0u; }
  // writes `c` bytes of the 255-run encoding of (len-15) at dst[o..): 255,...,255,rem
  LZ4HIP_DEV void put_ext(uint32_t o, uint32_t len, uint32_t c) {
    const uint32_t rem = (len - 15u) - 255u * (c - 1u);
    for (uint32_t base = 0; base < c; base += 64u) {
      VU i = w.lane() + base;
      w.st8(dst, i + o, W::select(i == c - 1u, VU(rem), VU(255u)), i < c);
    }
  }

  LZ4HIP_DEV void park(uint32_t ms, uint32_t ml, uint32_t offx) {
    p_ms = W::writelane(p_ms, ms, cnt);
    p_ml = W::writelane(p_ml, ml, cnt);
    p_off = W::writelane(p_off, offx, cnt);
    if (++cnt == 64u) flush();
  }

  // Raw parking (kRawPark policies): the lean loop parks the bare hit -- {hit position, forward length, offset} -- and liblz4's
  // backward extension ("catch-up": while (ip > anchor && match > 0 && ip[-1] == match[-1]) ip--, match--) is done here, for the
  // 64 parked hits at once, a lane each.  It changes neither where the finder continues (hit + forward length) nor the table, so
  // it is output work.  anchor = end of the previous sequence; a hit AT the anchor is liblz4's _next_match path (kNoCheck).
  // Entries of the exact path arrive finished (kFinal).
  LZ4HIP_DEV void resolve_raw() {
    if (cnt == 0u) return;
    const VU j = w.lane();
    const VB act = j < cnt;
    const VB raw = act & ((p_off & kFinal) == 0u);
    const VU endv = p_ms + p_ml;
    const VU anchor = W::select(j == 0u, VU(prev_end), W::shfl_up1(endv));
    const VU off = p_off & 0xFFFFu;
    const VU mpos = p_ms - off;
    const VU room = p_ms - anchor;
    const VU maxback = W::vmin(room, mpos);
    VU back = VU(0u);
    VB go = raw & (maxback > 0u);
    while (w.ballot(go)) {
      const VU a = w.ld8(src, p_ms - 1u - back, go), b = w.ld8(src, mpos - 1u - back, go);
      const VB eq = go & (a == b);
      back = W::select(eq, back + 1u, back);
      go = eq & (back < maxback);
    }
    p_ms = p_ms - back;
    p_ml = p_ml + back;
    p_off = W::select(raw, off | W::select(room == 0u, VU(kNoCheck), VU(0u)), p_off & ~kFinal);
  }

  // writes the parked sequences
  LZ4HIP_DEV void flush() {
    if (cnt == 0u) return;
    const uint32_t m = cnt;
    cnt = 0u;
    if (!ok) return;
    const VU j = w.lane();
    const VB act = j < m;
    const VU endv = p_ms + p_ml;                                       // input position after each sequence
    const VU anchor = W::select(j == 0u, VU(prev_end), W::shfl_up1(endv));
    const VU lit = p_ms - anchor;
    const VU mc = p_ml - 4u;
    const VU offset = p_off & 0xFFFFu;
    const VU nlx = W::select(lit >= 15u, W::div255(lit - 15u) + 1u, VU(0u));
    const VU nmx = W::select(mc >= 15u, W::div255(mc - 15u) + 1u, VU(0u));
    const VU size = W::select(act, nlx + lit + nmx + 3u, VU(0u));
    const VU o = w.excl_scan(size) + op;                               // where each sequence's token goes
    const uint32_t end = w.bcast(o + size, (int)(m - 1u));
    prev_end = w.bcast(endv, (int)(m - 1u));
    if (dense64 != 0u && flushes < 2u && m == 64u) {
      const uint32_t e31 = w.bcast(endv, 31);
      if (flushes == 0u) mark = e31;
      else if (e31 - mark < dense64) { bail = true; return; }
      flushes++;
    }
    if (limited) {
      // liblz4's two per-sequence checks (o = position of the token)
      const VB c1 = ((p_off & kNoCheck) == 0u) & (W::u64(o) + W::u64(lit) + W::u64(W::div255(lit)) + VU64(1u + 8u) > VU64((uint64_t)cap));
      const VB c2 = W::u64(o) + W::u64(nlx) + W::u64(lit) + W::u64(W::div255(mc + 240u)) + VU64(1u + 2u + 6u) > VU64((uint64_t)cap);
      if (w.ballot(act & (c1 | c2))) { ok = false; return; }
    }
    // literal runs: lane per sequence, whole dwords then the 0..3 byte tail
    const VU ld = o + nlx + 1u;
    // (all loads of a 16-byte group are requested before its stores: one memory round trip per group, not one per dword)
    uint32_t i = 0;
    for (; i < 48u; i += 16u) {
      if (!w.ballot(act & (VU(i + 4u) <= lit))) break;
      VU t[4];
      VB mk[4];
#pragma unroll
      for (uint32_t k = 0; k < 4u; k++) { mk[k] = act & (VU(i + 4u * k + 4u) <= lit); t[k] = w.ld32(src, anchor + (i + 4u * k), mk[k]); }
#pragma unroll
      for (uint32_t k = 0; k < 4u; k++) w.st32(dst, ld + (i + 4u * k), t[k], mk[k]);
    }
    for (;; i += 4u) {                                                 // literal runs beyond 48 bytes
      const VB mk = act & (VU(i + 4u) <= lit);
      if (!w.ballot(mk)) break;
      w.st32(dst, ld + i, w.ld32(src, anchor + i, mk), mk);
    }
    {
      const VU t0 = lit & ~3u;
      VU tb[3];
      VB mk[3];
#pragma unroll
      for (uint32_t k = 0; k < 3u; k++) { mk[k] = act & (t0 + k < lit); tb[k] = w.ld8(src, anchor + t0 + k, mk[k]); }
#pragma unroll
      for (uint32_t k = 0; k < 3u; k++) w.st8(dst, ld + t0 + k, tb[k], mk[k]);
    }
    const VU tok = (W::vmin(lit, VU(15u)) << 4) | W::vmin(mc, VU(15u));
    w.st8(dst, o, tok, act);
    w.st8(dst, o + 1u, lit - 15u, act & (nlx == 1u));
    const VU oo = ld + lit;
    w.st8(dst, oo, offset & 0xFFu, act);
    w.st8(dst, oo + 1u, offset >> 8, act);
    w.st8(dst, oo + 2u, mc - 15u, act & (nmx == 1u));
    uint64_t longs = w.ballot(act & ((nlx > 1u) | (nmx > 1u)));  // length runs of more than one byte (rare)
    while (longs) {
      const int k = ctz64(longs);
      longs &= longs - 1u;
      const uint32_t l = w.bcast(lit, k), c = w.bcast(mc, k), ok_ = w.bcast(o, k), nl = w.bcast(nlx, k), nm = w.bcast(nmx, k);
      if (nl > 1u) put_ext(ok_ + 1u, l, nl);
      if (nm > 1u) put_ext(ok_ + 1u + nl + l + 2u, c, nm);
    }
    op = end;
  }

  // last literals: token + run + raw bytes; returns total size or 0
  LZ4HIP_DEV uint32_t emit_last(uint32_t anchor) {
    const uint32_t last = n - anchor;
    if (limited && (uint64_t)op + last + 1u + (last + 255u - 15u) / 255u > cap) return 0;
    const uint32_t nlx = ext_count(last);
    w.st8(dst, VU(op), VU((last < 15u ? last : 15u) << 4), w.lane() == 0u);
    if (nlx) put_ext(op + 1u, last, nlx);
    w.copy(dst, op + 1u + nlx, src, anchor, last);
    return op + 1u + nlx + last;
  }

  // ---- the Out interface of FastCore ----
  LZ4HIP_DEV bool overlap_point() { return ok; }
  LZ4HIP_DEV void seq(uint32_t lit, uint32_t mc, uint32_t offset, uint32_t anchor, bool check_lits, bool, VU) {
    park(anchor + lit, mc + 4u, offset | (check_lits ? 0u : kNoCheck));
  }
  LZ4HIP_DEV uint32_t last(uint32_t anchor) {
    flush();
    if (!ok) return 0u;
    return emit_last(anchor);
  }
};

// ParkOut with raw parking, in one wavefront: what the writer wavefront of the two-wave kernel does (kernels.hip MailOut +
// mail_writer), runnable in the CPU lane simulator.
template <class W>
struct ParkOutRaw : ParkOut<W> {
  using P = ParkOut<W>;
  using VU = typename W::VU;
  static constexpr bool kRawPark = true;
  LZ4HIP_DEV ParkOutRaw(W& w_, const uint8_t* s, uint32_t n_, uint8_t* d, uint32_t cap_) : P(w_, s, n_, d, cap_) {}
  LZ4HIP_DEV void park(uint32_t ms, uint32_t ml, uint32_t offx) {
    P::p_ms = W::writelane(P::p_ms, ms, P::cnt);
    P::p_ml = W::writelane(P::p_ml, ml, P::cnt);
    P::p_off = W::writelane(P::p_off, offx, P::cnt);
    if (++P::cnt == 64u) { P::resolve_raw(); P::flush(); }
  }
  LZ4HIP_DEV bool overlap_point() { return P::ok; }
  LZ4HIP_DEV void seq(uint32_t lit, uint32_t mc, uint32_t offset, uint32_t anchor, bool check_lits, bool, VU) {
    park(anchor + lit, mc + 4u, offset | (check_lits ? 0u : P::kNoCheck) | P::kFinal);
  }
  LZ4HIP_DEV uint32_t last(uint32_t anchor) {
    P::resolve_raw();
    return P::last(anchor);
  }
};

// byU16 blocks (n < 65547).
// Measured dead end, kept as a note (round 2, profiles/r02_compress_notes.txt): the block's recent 40 KB in a ring of 156 VGPRs the
// compiler does not allocate (amdgpu_num_vgpr + VGPR-indexing mode with a wave-uniform slot index), so that the window, the row at
// the hit and the row at the candidate are register reads + ds_bpermute rotations instead of global loads.  Bit-exact on the GPU,
// memory waits fell from 66 % to 42 % of the wave's cycles -- but a step grew from 110 to 200 instructions, a wavefront issues one
// instruction per ~4.4 cycles whatever its dependencies, and a mode switch (s_set_gpr_idx_on/off) costs ~80 cycles that overlap
// with nothing: 49.5 .. 55.3 GB/s against 59.9 without the ring.
// U16 = true: blocks < 65547 bytes (liblz4's byU16 table, 4-byte hash); false: larger blocks (byU32: 4096 entries of
// {position, fingerprint} in 64 bits, 5-byte hash, candidates more than 65535 bytes back are no hits) -- round 3: the LZ4 Frame
// default block size is 4 MiB (LZ4FrameOutputStream.java:169-171), until then those blocks ran on the one-sequence core alone.
// PK: the compact byU32 entries of FastCore (blocks of at most 4 MiB).
template <class W, class OUT = ParkOut<W>, bool U16 = true, bool PK = false>
struct FastV2 {
  using VU = typename W::VU;
  using VU64 = typename W::VU64;
  using VB = typename W::VB;
  using Gen = FastCore<W, U16, OUT, PK>;
  static constexpr bool S32 = Gen::S32;
  using VE = typename Gen::VE;
  using E = typename Gen::E;
  static constexpr uint32_t kFwdBytes = 256u;                       // forward compare of the lean step: 64 lanes x 4 bytes
  static constexpr uint32_t kTail = 12u + 64u + kFwdBytes + 16u;    // the lean loop stays this far from the block end

  W& w;
  OUT& out;
  const uint8_t* src;
  uint32_t n;
  FastStats* st;

  LZ4HIP_DEV FastV2(W& w_, OUT& out_, const uint8_t* s, uint32_t n_, FastStats* st_ = nullptr) : w(w_), out(out_), src(s), n(n_), st(st_) {}

  LZ4HIP_DEV uint32_t run() {
    if (n < 13u) return out.last(0u);
    {
      const uint32_t x0 = w.sld32(src, 0);
      // every bucket: {pos 0, fp(bytes at 0)}
      if constexpr (U16) w.template lds_fill<true>(1u << Gen::HLOG, (((x0 * 2654435761u) >> 3) & LZ4HIP_FP_MASK));
      else w.template lds_fill<S32>(1u << Gen::HLOG, (E)(((x0 * 2654435761u) >> 16) & Gen::FPM));
      w.sync();
    }
    Gen gen(w, out, src, n, st);
#if defined(__HIP_DEVICE_COMPILE__) && LZ4HIP_V2_ASM_PROF
    const uint64_t prof_t0 = w.tick(n);
    uint64_t prof_lean = 0, prof_calls = 0;
#endif
    // the head of the block is a plain run from position 1: exact path, one sequence
    uint32_t r = gen.template loop<2>(false, 1u, 0u, 0u);
    while (gen.one_done && !out.bail) {
      gen.one_done = false;
#if defined(__HIP_DEVICE_COMPILE__) && LZ4HIP_V2_ASM_PROF
      const uint64_t prof_l0 = w.tick(gen.p_ip);
      const uint32_t ip = lean(gen.p_ip);
      prof_lean += w.tick(ip) - prof_l0; prof_calls++;
#else
      const uint32_t ip = lean(gen.p_ip);
#endif
      gen.anchor = ip;
      const uint64_t tx = st ? w.tick(ip) : 0;      // (profiling kernel: cycles and count of the exact path's sequences)
      r = gen.template loop<2>(true, ip + 1u, 0u, ip);
      if (st) { st->t[6] += w.tick(r) - tx; st->t[7] += 1; }
    }
#if defined(__HIP_DEVICE_COMPILE__) && LZ4HIP_V2_ASM_PROF
    if (w.lane() == 0u) {
      for (int i = 0; i < 6; i++) atomicAdd(&g_asm_prof[i], (unsigned long long)aprof[i]);
      atomicAdd(&g_asm_prof[6], (unsigned long long)prof_lean);
      atomicAdd(&g_asm_prof[7], (unsigned long long)(w.tick(r) - prof_t0));
      atomicAdd(&g_asm_prof[8], 1ull);
      atomicAdd(&g_asm_prof[9], (unsigned long long)prof_calls);
    }
#endif
    return r;
  }

  // The lean loop.  Enters and leaves in the post-match state at `ip` (== anchor; table committed up to the previous step);
  // returns the position at which the exact path has to take the next sequence.
  uint32_t aprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // (LZ4HIP_V2_ASM_PROF developer builds)
  LZ4HIP_DEV uint32_t lean(uint32_t ip) {
    if (n < kTail + 8u) return ip;
    const uint32_t lim = n - kTail;
    const VU j = w.lane();
    const VU cj = W::select(j == 0u, VU(0xFFFFFFFEu), j - 1u);   // lane 0: ip-2 (insert only), lane 1: ip, lane l: ip + l - 1
    const VU j4 = j * 4u;
    uint32_t pf_end = ip & ~(W::kPrefetchBytes - 1u);
    uint64_t l_steps = 0, l_slow = 0, l_seq = 0, l_t[6] = {0, 0, 0, 0, 0, 0}, tk_keep = 0;   // (profiling kernel: registers, added to *st at the end)
    VU prev_fa = VU(0u);
    uint32_t prev_hpos = 0x80000000u;   // (no row yet: ip - prev_hpos is huge)
    while (ip <= lim && !out.bail) {
#if defined(__HIP_DEVICE_COMPILE__) && LZ4HIP_V2_ASM
#ifndef LZ4HIP_V2_ROWLOAD
#define LZ4HIP_V2_ROWLOAD 1   /* 0: without a row the first step after an exact-path sequence is a C++ step (developer A/B) */
#endif
      // The hand-scheduled loop cuts its windows out of the row at the previous hit.  After a sequence of the exact path there is
      // none: fetch the 256 bytes at ip - 2 (one round trip) instead of spending a whole compiler-generated step on getting one.
      if constexpr (LZ4HIP_V2_ROWLOAD && W::kAsmLean && OUT::kAsmPark) {
        if (!st && ip - prev_hpos > (U16 ? 127u : 126u) && ip >= 2u && out.cnt < 63u) {
          prev_hpos = ip - 2u;
          prev_fa = w.ldu32(src, j4 + prev_hpos);
        }
      }
      // The common steps run in the hand-scheduled loop of lz4_fast_v2_asm.h; it comes back when 64 hits are parked, at the loop
      // limit, or in front of a step it does not handle -- nothing of that step is left in the table, and the C++ step below
      // takes it from the same state with every rule.
      if constexpr (U16 && W::kAsmLean && OUT::kAsmPark) {
        if (!st) {
#if LZ4HIP_ASM_DBG & 4
          const uint32_t dbg_ip = ip, dbg_pc = out.cnt, dbg_php = prev_hpos;
#endif
          // (the loop requests the rows of the NEXT hit before it has compared the next position with its limit: that hit lies
          // up to 191 bytes behind the limit it is given, so it gets one that much earlier; the C++ steps do the rest)
          // (the loop sets exec itself and leaves it at all ones: it must be entered by a whole wavefront -- finder wavefronts are
          // whole by construction, kernels.hip launches 64 x k threads and no caller sits inside a divergent region)
          if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap();
          const uint32_t code = lean_asm_run(ip, prev_hpos, pf_end, out.cnt, prev_fa, out.p_ms, out.p_ml, out.p_off,
                                             lim >= 192u ? lim - 192u : 0u, src,
                                             (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)w.lds, n, aprof);
#if LZ4HIP_ASM_DBG & 4
          if (w.lane() == 0u) {
            const unsigned long long k = atomicAdd(&g_asm_prof[15], 1ull);
            if (k < 200) { g_asm_dbg[k * 8 + 0] = dbg_ip; g_asm_dbg[k * 8 + 1] = dbg_php; g_asm_dbg[k * 8 + 2] = dbg_pc; g_asm_dbg[k * 8 + 3] = code;
                           g_asm_dbg[k * 8 + 4] = ip; g_asm_dbg[k * 8 + 5] = prev_hpos; g_asm_dbg[k * 8 + 6] = out.cnt; g_asm_dbg[k * 8 + 7] = 0; }
          }
#endif
          (void)code;
          if (out.cnt == 64u) { out.batch(); continue; }
          if (ip > lim) break;
        }
      }
#ifndef LZ4HIP_V2_ASM32
#define LZ4HIP_V2_ASM32 1   /* 0: byU32 blocks keep the compiler-generated lean loop (developer A/B) */
#endif
      // byU32 blocks: the same loop with 64-bit table entries and the 5-byte hash (lz4_fast_v2_asm32.h)
      if constexpr (!U16 && LZ4HIP_V2_ASM32 && !LZ4HIP_V2_ASM_PROF && W::kAsmLean && OUT::kAsmPark) {
        if (!st) {
          if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap();
          if constexpr (PK)
            (void)lean_asm_run32p(ip, prev_hpos, pf_end, out.cnt, prev_fa, out.p_ms, out.p_ml, out.p_off,
                                  lim >= 192u ? lim - 192u : 0u, src,
                                  (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)w.lds, n);
          else
          (void)lean_asm_run32(ip, prev_hpos, pf_end, out.cnt, prev_fa, out.p_ms, out.p_ml, out.p_off,
                               lim >= 192u ? lim - 192u : 0u, src,
                               (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)w.lds, n);
          if (out.cnt == 64u) { out.batch(); continue; }
          if (ip > lim) break;
        }
      }
#elif defined(LZ4HIP_HOST_ASM_EMU)
      // CPU suite (tests/hostsim/hostsim_asm.cpp): the same calls, the loop's TEXT run by an interpreter of its instructions
      // (tests/hostsim/asm_emu.h) -- the hand-scheduled loops had no check off the GPU before.
      if constexpr (W::kAsmLean && OUT::kAsmPark) {
        if (!st) {
          if (ip - prev_hpos > (U16 ? 127u : 126u) && ip >= 2u && out.cnt < 63u) {
            prev_hpos = ip - 2u;
            prev_fa = w.ldu32(src, j4 + prev_hpos);
          }
          w.template lean_asm_emu<U16, PK>(ip, prev_hpos, pf_end, out.cnt, prev_fa, out.p_ms, out.p_ml, out.p_off, lim >= 192u ? lim - 192u : 0u, src, n);
          if (out.cnt == 64u) { out.batch(); continue; }
          if (ip > lim) break;
        }
      }
#endif
      if (st) l_steps++;
      uint64_t tk = (st && LZ4HIP_V2_PROBE == 0) ? w.tick(ip) : tk_keep;
      // profiling kernel only.  LZ4HIP_V2_PROBE = k + 1 measures ONLY the interval that ends at phase point k (one pair of clock
      // reads per step: the reads themselves cost ~40 cycles each and wait for LDS, so all phases at once distort each other)
#define LZ4HIP_PHASE2(i, dep) do { if (st) { if (LZ4HIP_V2_PROBE == 0) { const uint64_t t_ = w.tick(dep); l_t[i] += t_ - tk; tk = t_; } \
        else if (LZ4HIP_V2_PROBE == (i) + 2 || (LZ4HIP_V2_PROBE == 1 && (i) == 5)) { tk_keep = tk = w.tick(dep); } \
        else if (LZ4HIP_V2_PROBE == (i) + 1) { l_t[i] += w.tick(dep) - tk; } } } while (0)
      const VU pos = cj + ip;
#ifndef LZ4HIP_V2_WIN_FROM_ROW
#define LZ4HIP_V2_WIN_FROM_ROW 1   /* the next window is taken out of the 256 bytes the forward compare fetched at the hit (two lane permutes) */
#endif
      VU x32, x32b = VU(0u);   // bytes [pos, pos + 4) and (byU32: the 5-byte hash) [pos + 4, pos + 8)
      if (LZ4HIP_V2_WIN_FROM_ROW && LZ4HIP_LIKELY(ip - prev_hpos <= (U16 ? 189u : 185u))) {   // bytes [ip - 2, ip + 66 (+ 4)) lie inside prev_fa = row at prev_hpos
        const VU o = cj + (ip - prev_hpos);
        const VU d = o >> 2;
        const VU w1 = W::shfl(prev_fa, d + 1u);
        x32 = W::alignbyte(w1, W::shfl(prev_fa, d), o & 3u);
        if constexpr (!U16) x32b = W::alignbyte(W::shfl(prev_fa, d + 2u), w1, o & 3u);
      } else {
        if constexpr (U16) x32 = w.ldu32(src, pos);
        else { const VU64 x64 = w.ldu64(src, pos); x32 = W::lo32(x64); x32b = W::lo32(x64 >> 32); }
      }
      VU h, fp;
      if constexpr (U16) {
        const VU prod = x32 * 2654435761u;
        h = prod >> (32 - Gen::HLOG);
        fp = (prod >> 3) & LZ4HIP_FP_MASK;
      } else {
        const VU64 x64 = W::u64(x32) | (W::u64(x32b) << 32);
        h = W::lo32(((x64 << 24) * 889523592379ull) >> (64 - Gen::HLOG));
        fp = Gen::fp32(x32);
      }
      LZ4HIP_PHASE2(0, w.bcast(h, 0));     // t[0]: window load + hash
      const VE e = w.template lds_rdu<S32>(h);
      const VE newe = Gen::mk_entry(pos, fp);
      uint64_t tmask = w.ballot(Gen::e_fp(e) == fp) & ~1ull;
      if constexpr (!U16) tmask &= w.ballot(Gen::e_pos(e) + Gen::MAXD >= pos);   // (byU32: a candidate more than 65535 bytes back is no hit)
#ifndef LZ4HIP_COUNT_CAT
#define LZ4HIP_COUNT_CAT 5   /* profiling builds: which way out of the lean loop FastStats::false_pos counts (5 = no hit in the window) */
#endif
      if (LZ4HIP_UNLIKELY(tmask == 0)) { if (st && LZ4HIP_COUNT_CAT == 5) st->false_pos++; break; }   // no tentative hit in 63 probes (nothing committed yet; profiling: counted in false_pos)
      LZ4HIP_PHASE2(1, (uint32_t)tmask);    // t[1]: table read + ballot
#ifndef LZ4HIP_V2_PVERIFY
#define LZ4HIP_V2_PVERIFY 0   /* 1: the candidate bytes of EVERY tentative lane behind the first are gathered in the first lane's round trip */
#endif
      // (parallel verification: with narrow fingerprints most tentative lanes are false; their 4 candidate bytes come back with the
      // first tentative lane's rows, so a ruled-out first lane is followed directly by the first lane whose candidate verifies --
      // one more round trip at most, and to a line the gather has just touched)
      VU c4 = VU(0u);
      if (LZ4HIP_V2_PVERIFY && OUT::kRawPark) c4 = w.ld32(src, Gen::e_pos(e), w.lanes(tmask & (tmask - 1ull)));
      // The hit is the first tentative lane that survives: a tentative lane is ruled out, and the search goes on to the next one
      // in the same window (raw-parking policies only; the others take the exact path as before), when
      //   K  an earlier committing lane shares its bucket with a different fingerprint -- liblz4, inserting position by position,
      //      would have found that lane's entry there, not the one that made this lane tentative;
      //   F  its candidate's first 4 bytes differ (a fingerprint collision).
      // Either way the lanes up to it are inserted positions (committed), exactly as in liblz4's continuing search.
      uint64_t tm = tmask;
      uint32_t lo = 0;                       // lanes below lo are committed
      uint32_t k0, hpos, mpos, cnt;
      uint64_t upto;
      VU fa, bb = VU(0u);
      VB bval = VB(false);
      bool fail = false;
      for (;;) {
        k0 = (uint32_t)ctz64(tm);
        upto = (2ull << k0) - 1ull;                               // lanes 0..k0
        const uint64_t inm = upto & ~((1ull << lo) - 1ull);       // lanes lo..k0 commit now
        const VE old = w.template lds_max<S32>(h, newe, w.lanes(inm));
        hpos = ip + k0 - 1u;
        mpos = Gen::se_pos(w.template bcast_e<S32>(e, (int)k0));
        // candidate fetch: forward 64 x 4 bytes from both positions; backward (policies that extend backwards here): lane l
        // (1..k0-1) holds src[ip+l-1] in its window word and needs src[mpos-k0+l] -- one contiguous byte load
        // (32 or 16 lanes instead of 64 -- fewer candidate lines per fetch -- measured no different: 59.2 / 59.2 / 59.1 GB/s)
        fa = w.ldu32(src, j4 + hpos);
        const VU fb = w.ldu32(src, j4 + mpos);
        if constexpr (!OUT::kRawPark) {   // (raw parking: the backward extension is output work)
          const VU bidx = j + (mpos - k0);
          bval = j + mpos >= VU(k0);
          bb = w.ldu8(src, W::select(bval, bidx, VU(0u)));
        }
        if (LZ4HIP_UNLIKELY(hpos + W::kPrefetchBytes > pf_end)) {
          if (pf_end < n) w.prefetch4k(src, pf_end, n);
          pf_end += W::kPrefetchBytes;
        }
        LZ4HIP_PHASE2(2, mpos);               // t[2]: commit + fetch issue
        const uint64_t det = w.ballot(old != e) & inm;
        LZ4HIP_PHASE2(3, (uint32_t)det);      // t[3]: atomic result
        const VU fx = fa ^ fb;
        const uint64_t dm = w.ballot(fx != 0u);
        cnt = 0;
        if (LZ4HIP_LIKELY(dm != 0)) {
          const int f = ctz64(dm);
          cnt = 4u * (uint32_t)f + ((uint32_t)ctz32(w.bcast(fx, f)) >> 3);
        }
        LZ4HIP_PHASE2(4, cnt);                // t[4]: candidate fetch wait + forward count
        // A bucket shared by two committing lanes (det: a lane got back another lane's entry instead of the one it had read) only
        // matters if liblz4 would have decided differently: the later lane would have seen the earlier lane's entry instead of
        // the old one.  For a lane below the hit that changes nothing unless the two fingerprints agree (its old entry was no
        // tentative hit, or it would be the hit lane); for the hit lane it always does (case K).  The atomic max leaves the bucket
        // holding the later position either way -- the state liblz4 ends up with.  Exactly one lane with a foreign entry means
        // exactly one more lane in its bucket is new to it (lanes committed by an earlier trip of this loop are simply what it
        // should see); two or more: exact path.
        bool bad = dm == 0;                                        // a match of >= 256 bytes: exact path
        bool ruled_out = dm != 0 && cnt < 4u;                      // case F
        if (LZ4HIP_UNLIKELY(det != 0)) {
          if (det & (det - 1u)) {
            bad = true;
            if (st && LZ4HIP_COUNT_CAT == 1) st->false_pos++;
          } else {
            const int dl = ctz64(det);
            const uint32_t od = w.bcast(Gen::e_fp(old), dl), hd = w.bcast(h, dl), fd = w.bcast(fp, dl);
            if (od == fd) { bad = true; if (st && LZ4HIP_COUNT_CAT == 2) st->false_pos++; }
            else if (hd == w.bcast(h, (int)k0)) ruled_out = true;  // case K
          }
        }
        if (st && LZ4HIP_COUNT_CAT == 3 && dm == 0) st->false_pos++;
        if (st && LZ4HIP_COUNT_CAT == 4 && !bad && ruled_out && (tm & ~upto) == 0) st->false_pos++;
        if (LZ4HIP_LIKELY(!bad && !ruled_out)) break;
        if (OUT::kRawPark && !bad) {
          tm &= ~upto;
          if (LZ4HIP_V2_PVERIFY) tm &= w.ballot(c4 == x32);        // tentative lanes whose candidate bytes differ are not hits
          lo = k0 + 1u;
          if (tm) continue;                                        // the next tentative lane of this window
        }
        // undo everything this step committed, exact path
        if (st) l_slow++;
        w.template lds_wr<S32>(h, e, w.lanes(upto));
        w.sync();
        fail = true;
        break;
      }
      if (LZ4HIP_UNLIKELY(fail)) break;
      w.sync();
      if (st) l_seq++;
      if constexpr (OUT::kRawPark) {
        out.park(hpos, cnt, hpos - mpos);
      } else {
        // catch-up: equal bytes below the hit, down to lane 1 (= anchor) and position 0 of the candidate side
        const uint64_t eqm = w.ballot(bval & ((x32 & 0xFFu) == bb));
        const uint64_t range = ((1ull << k0) - 1ull) & ~1ull;      // lanes 1..k0-1
        const uint64_t ne = (~eqm & range) | 1ull;
        const uint32_t back = k0 - 1u - (63u - (uint32_t)clz64(ne));
        out.park(hpos - back, cnt + back, (hpos - mpos) | (k0 == 1u ? OUT::kNoCheck : 0u));
      }
      ip = hpos + cnt;
      prev_fa = fa;
      prev_hpos = hpos;
      LZ4HIP_PHASE2(5, ip);                 // t[5]: catch-up + park (+ flush every 64 sequences)
    }
#undef LZ4HIP_PHASE2
    if (st) { st->steps += l_steps; st->slow_steps += l_slow; st->sequences += l_seq; for (int i = 0; i < 6; i++) st->t[i] += l_t[i]; }
    return ip;
  }
};

}  // namespace lz4hip

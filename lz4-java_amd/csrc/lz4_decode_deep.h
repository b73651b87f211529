// lz4_decode_deep.h -- the deep interior loop of the block decoder (lz4_decode_core.h, PIPE == 2).
//
// Same sequences, same bytes as the other interior loops of decode_block (LZ4_decompress_safe / _fast of liblz4 1.9.3,
// /root/reference/src/jni/net_jpountz_lz4_LZ4JNI.c:216 / :169); what differs is how a trip is executed:
//  * the compressed stream of the block is staged in an LDS ring (Grp::sr_*), so the token parse is a chain of LDS reads -- a
//    wavefront's VECTOR memory operations return in order, and a parse that read the stream from memory waited for every older
//    match-source load as well (one sequence per memory latency and block);
//  * slots hold the sequences whose match source is on its way; a trip fills one and stores the oldest, in sequence order, so the
//    wait of a trip is for a load that was requested a trip (or more) earlier.  Two slots are the default: with the trip this lean
//    the deeper queues measured slower (LZ4HIP_DEEP_SLOTS 2 / 3 / 4 / 6: 821 / 791 / 762 / 524 GB/s on 16384 x 4 MiB blocks) -- more
//    sequences wait, more match sources reach into them, and every slot is code;
//  * everything a trip does in memory is unconditional and whole-step: all lanes of the group load and store their part of 64
//    bytes whatever the lengths; what a step writes past its length is put right by the stores that follow (slots are stored in
//    sequence order), slots that hold no sequence are aimed at bytes that are written again later.  No per-lane predicates, no
//    divergent branches on the main path: ~3x fewer instructions per trip than the compiled two-trip loop, and the compiler can
//    count the younger memory operations a wait may leave outstanding (vmcnt(N) instead of vmcnt(0)).
// Returns with ip / op at the start of a sequence it does not take (length runs of two or more bytes, an invalid offset, the last
// 2 KB of the stream), everything before it stored.
#pragma once
#include <stdint.h>
#ifndef LZ4HIP_UNLIKELY
#define LZ4HIP_UNLIKELY(x) __builtin_expect(!!(x), 0)
#endif

namespace lz4hip {

template <class Grp>
LZ4HIP_DEV bool decode_deep_loop(Grp& g, const uint8_t* src, const int iend, uint8_t* dst, const int oend, int& ip_io, int& op_io, uint8_t* stage) {
  constexpr uint32_t KS = Grp::kStream, PC = Grp::kPiece, STEP = 64u;
  typedef typename Grp::LChunk LChunk;   // a group's 64-byte step (per lane: LB bytes)
  uint32_t ip = (uint32_t)ip_io, op = (uint32_t)op_io;
  const uint32_t ilim = (uint32_t)iend - 306u, olim = (uint32_t)oend - 606u;
  g.sr_begin(stage);
  // the ring holds the stream bytes [.., avail); `fetched` is the end of what has been requested
  uint32_t avail = ip & ~(PC - 1u);
  while (avail < ip + 768u) { g.sr_put(avail, g.sr_fetch(src, avail)); avail += PC; }   // (the caller left >= 2048 stream bytes)
  uint32_t fetched = avail;
  // slots: literals (v) and match source (u) of a sequence that starts at output position sop; lit = its literal length
  // (six sets of registers; LZ4HIP_DEEP_SLOTS of them are used)
  LChunk v0 = LChunk(), u0 = LChunk(), v1 = LChunk(), u1 = LChunk(), v2 = LChunk(), u2 = LChunk(), v3 = LChunk(), u3 = LChunk();
  uint32_t sop0 = op, sop1 = op, sop2 = op, sop3 = op, lit0 = 0, lit1 = 0, lit2 = 0, lit3 = 0;
  LChunk v4 = LChunk(), u4 = LChunk(), v5 = LChunk(), u5 = LChunk();   // (LZ4HIP_DEEP_SLOTS == 6)
  uint32_t sop4 = op, sop5 = op, lit4 = 0, lit5 = 0;
  (void)v4; (void)u4; (void)v5; (void)u5; (void)sop4; (void)sop5; (void)lit4; (void)lit5;
  typename Grp::PieceRegs rf = g.sr_fetch(src, fetched - PC);
  uint32_t rf_pos = fetched - PC;
  uint32_t t4 = g.sr_ld32(ip);

#define LZ4HIP_RETIRE(k) do { g.step_store(dst + sop##k, v##k); g.step_store(dst + sop##k + lit##k, u##k); } while (0)
#ifndef LZ4HIP_DEEP_COND_REFILL
#define LZ4HIP_DEEP_COND_REFILL 1   /* 1: no request when the ring has no room (0: the last piece is requested again -- every trip the same operations; measured: configs[2] 740 -> 765 GB/s, 16384 x 64 KiB text 128 -> 148 with 1) */
#endif
#if LZ4HIP_DEEP_COND_REFILL
#define LZ4HIP_REFILL_FETCH if ((fetched + PC <= (uint32_t)iend) & (fetched + PC <= (ip & ~(PC - 1u)) + KS)) { rf_pos = fetched; rf = g.sr_fetch(src, rf_pos); fetched = rf_pos + PC; }
#define LZ4HIP_REFILL_PUT if (fetched != avail) { g.sr_put(rf_pos, rf); avail = rf_pos + PC; }
#else
#define LZ4HIP_REFILL_FETCH { const bool room = (fetched + PC <= (uint32_t)iend) & (fetched + PC <= (ip & ~(PC - 1u)) + KS); rf_pos = room ? fetched : fetched - PC; rf = g.sr_fetch(src, rf_pos); fetched = rf_pos + PC; }
#define LZ4HIP_REFILL_PUT { g.sr_put(rf_pos, rf); avail = rf_pos + PC; }
#endif
#ifndef LZ4HIP_DEEP_SLOTS
#define LZ4HIP_DEEP_SLOTS 2   /* slots of the pipeline: 2 (one match source of a block waits while the next is requested), 3, 4 or 6 -- measured on configs[2]: 821 / 791 / 762 / 524 GB/s: the leaner trip beats the deeper queue */
#endif
#if LZ4HIP_DEEP_SLOTS == 6
#define LZ4HIP_REST(b, d, e, f, OP) OP(b); OP(d); OP(e); OP(f)
#define LZ4HIP_TRIP(c, a, b, d, e, f, REFILL) LZ4HIP_TRIP_(c, a, LZ4HIP_REST(b, d, e, f, LZ4HIP_RETIRE), LZ4HIP_REST(b, d, e, f, LZ4HIP_AIM), REFILL)
#elif LZ4HIP_DEEP_SLOTS == 2
#define LZ4HIP_TRIP(c, a, REFILL) LZ4HIP_TRIP_(c, a, (void)0, (void)0, REFILL)
#elif LZ4HIP_DEEP_SLOTS == 3
#define LZ4HIP_REST(b, OP) OP(b)
#define LZ4HIP_TRIP(c, a, b, REFILL) LZ4HIP_TRIP_(c, a, LZ4HIP_REST(b, LZ4HIP_RETIRE), LZ4HIP_REST(b, LZ4HIP_AIM), REFILL)
#else
#define LZ4HIP_REST(b, d, OP) OP(b); OP(d)
#define LZ4HIP_TRIP(c, a, b, d, REFILL) LZ4HIP_TRIP_(c, a, LZ4HIP_REST(b, d, LZ4HIP_RETIRE), LZ4HIP_REST(b, d, LZ4HIP_AIM), REFILL)
#endif
#define LZ4HIP_AIM(k) do { sop##k = aim; lit##k = 0u; } while (0)
  // one trip: fills slot c, stores slot a (the oldest of the slots filled by the trips before; RETIRE_REST / AIM_REST: the others, oldest first)
#define LZ4HIP_TRIP_(c, a, RETIRE_REST, AIM_REST, REFILL)                                                                      \
  {                                                                                                                            \
    uint32_t lit = (t4 >> 4) & 15u, ml = t4 & 15u;                                                                             \
    const uint32_t e1 = (t4 >> 8) & 255u;                                                                                      \
    const bool l15 = lit == 15u;                                                                                               \
    lit += l15 ? e1 : 0u;                                                                                                      \
    const uint32_t hdr = l15 ? 2u : 1u;                                                                                        \
    const uint64_t o8 = g.sr_ld64(ip + hdr + lit);                                                                             \
    const uint32_t off = (uint32_t)o8 & 0xFFFFu;                                                                               \
    const bool m15 = ml == 15u;                                                                                                \
    const uint32_t e2 = (uint32_t)(o8 >> 16) & 255u;                                                                           \
    ml += (m15 ? e2 : 0u) + 4u;                                                                                                \
    const uint32_t nxt = (uint32_t)(o8 >> (m15 ? 24 : 16));                                                                    \
    const uint32_t adv = hdr + lit + (m15 ? 3u : 2u);                                                                          \
    const uint32_t mpos = op + lit - off;            /* where the match copies from */                                         \
    if ((l15 & (e1 == 255u)) | (m15 & (e2 == 255u)) | (off > op + lit)) {   /* not for this loop (nothing of it done) */       \
      LZ4HIP_RETIRE(a); RETIRE_REST; break; }                                                                                  \
    if (LZ4HIP_UNLIKELY((lit > STEP) | (ml > STEP) | (mpos + ml > sop##a))) {                                                   \
      /* the slots first: the source reaches into bytes that wait in one, or this sequence is copied the wide way */          \
      LZ4HIP_RETIRE(a); RETIRE_REST;                                                                                           \
      const bool simple = (lit <= STEP) & (ml <= STEP) & (mpos + ml <= op);                                                    \
      uint32_t aim = op;                                                                                                       \
      if (!simple) {                                                                                                           \
        g.copy_lits_wide(dst + op, src + ip + hdr, lit);                                                                       \
        g.copy_match_wide(dst, op + lit, off, ml);                                                                             \
        aim = op + lit + ml;                         /* (empty slots aim behind what has just been written) */                \
      }                                                                                                                        \
      LZ4HIP_AIM(a); AIM_REST;                                                                                                 \
      if (!simple) {                                                                                                           \
        LZ4HIP_AIM(c); op = aim; ip += adv; t4 = nxt;                                                                          \
        if (fetched != avail) { g.sr_put(rf_pos, rf); avail = rf_pos + PC; }   /* (a piece on its way lands before the rotation restarts) */ \
        if (!((ip <= ilim) & (op <= olim) & (ip + 288u <= avail))) break;                                                      \
        continue;                                                                                                              \
      }                                                                                                                        \
    }                                                                                                                          \
    if (REFILL == 1) {   /* the next piece of the stream, when the ring has room for it and the stream has it */              \
      LZ4HIP_REFILL_FETCH                                                                                                      \
    }                                                                                                                          \
    v##c = g.sr_step(ip + hdr);                                                                                                \
    u##c = g.step_load_upto(dst + mpos, ml, dst + mpos - g.lane_bytes());   /* (lanes beyond the match re-read its first bytes: no line of their own) */ \
    sop##c = op; lit##c = lit;                                                                                                 \
    LZ4HIP_RETIRE(a);                                                                                                          \
    if (REFILL == 2) { LZ4HIP_REFILL_PUT }                                                                                     \
    op += lit + ml; ip += adv; t4 = nxt;                                                                                       \
    if (!((ip <= ilim) & (op <= olim) & (ip + 288u <= avail))) { RETIRE_REST; LZ4HIP_RETIRE(c); break; }                       \
  }
  for (;;) {
#if LZ4HIP_DEEP_SLOTS == 6
    LZ4HIP_TRIP(0, 1, 2, 3, 4, 5, 1)
    LZ4HIP_TRIP(1, 2, 3, 4, 5, 0, 0)
    LZ4HIP_TRIP(2, 3, 4, 5, 0, 1, 0)
    LZ4HIP_TRIP(3, 4, 5, 0, 1, 2, 0)
    LZ4HIP_TRIP(4, 5, 0, 1, 2, 3, 0)
    LZ4HIP_TRIP(5, 0, 1, 2, 3, 4, 2)
#elif LZ4HIP_DEEP_SLOTS == 2
    LZ4HIP_TRIP(0, 1, 1)
    LZ4HIP_TRIP(1, 0, 2)
#elif LZ4HIP_DEEP_SLOTS == 3
    LZ4HIP_TRIP(0, 1, 2, 1)
    LZ4HIP_TRIP(1, 2, 0, 0)
    LZ4HIP_TRIP(2, 0, 1, 2)
#else
    LZ4HIP_TRIP(0, 1, 2, 3, 1)
    LZ4HIP_TRIP(1, 2, 3, 0, 0)
    LZ4HIP_TRIP(2, 3, 0, 1, 0)
    LZ4HIP_TRIP(3, 0, 1, 2, 2)
#endif
  }
  // (every way out of a trip has stored what was waiting, oldest first)
#undef LZ4HIP_REFILL_FETCH
#undef LZ4HIP_REFILL_PUT
#undef LZ4HIP_TRIP
#undef LZ4HIP_TRIP_
#undef LZ4HIP_REST
#undef LZ4HIP_AIM
#undef LZ4HIP_RETIRE
  ip_io = (int)ip; op_io = (int)op;
  return (ip <= ilim) & (op <= olim) & (ip + 288u > avail);   // true: left for want of stream bytes in the ring -- come again
}

}  // namespace lz4hip

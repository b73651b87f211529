// lz4_decode_ring.h -- the ring loop of the block decoder (lz4_decode_core.h, PIPE == 3): an interior loop whose trip touches
// memory only for what MUST come from memory.
//
// Same sequences, same bytes as the other interior loops of decode_block (LZ4_decompress_safe / _fast of liblz4 1.9.3,
// /root/reference/src/jni/net_jpountz_lz4_LZ4JNI.c:216 / :169).  What a trip works on:
//  * the compressed stream of the block is staged in an LDS ring (as in the deep loop, lz4_decode_deep.h): the token parse is a chain
//    of LDS reads, the literals come from LDS;
//  * the block's RECENT OUTPUT lives in a second LDS ring of KW bytes (group_dev.h rg_*): sequences are written into the ring, in
//    sequence order; a match whose source lies inside the ring is an LDS -> LDS copy (near match), only a source the ring no longer
//    holds is a load from the block's output in memory (far match) -- and that load is never waited for in the trip that requests
//    it: slots hold the sequences whose match source is on its way, a trip fills one and puts the oldest into the ring;
//  * output leaves the ring as whole, ADDRESS-ALIGNED 64-byte steps (the ring is indexed by address, so the flusher's LDS reads and
//    its stores are aligned): every output byte is stored to memory exactly once, no partial line is ever written twice;
//  * everything a trip does is unconditional and whole-step, as in the deep loop: what a step writes past its length is put right by
//    the steps that follow (sequence order), empty slots are aimed at bytes that are written again.
// With KW = 512 (4 lanes per block, 16 blocks per wavefront) this is the loop for full batches of 64 KiB blocks -- text (offsets of a
// few hundred bytes stay on chip, the rest is pipelined) and far-match data alike (two match sources of a block in flight instead of
// one); with KW = 4096 (8 / 16 lanes) for big blocks with a short match window (BASELINE configs[2]: four matches of five never
// leave the chip).
// Leaves -- with ip / op at the start of a sequence it does not take, everything before it in memory -- on: literal or match
// lengths over 64, length runs of two or more bytes, an invalid offset, the rare source that lies neither in the ring nor in
// flushed memory (only right behind the loop's entry), the end of the staged stream.
#pragma once
#include <stdint.h>
#ifndef LZ4HIP_UNLIKELY
#define LZ4HIP_UNLIKELY(x) __builtin_expect(!!(x), 0)
#endif
#ifndef LZ4HIP_RING_SLOTS
#define LZ4HIP_RING_SLOTS 2   /* 2: one sequence waits for its match source while the next is parsed and requested; 3: two wait */
#endif

namespace lz4hip {

// entry: 64 <= op (the first aligned step of the flusher lies inside the block's slot), ip + 320 <= iend (the ring's first pieces are
// readable), ip <= iend - 306, op <= oend - 606 (the interior loops' distance from both ends).
// returns true when it left for want of staged stream bytes only (the caller comes again).
template <class Grp>
LZ4HIP_DEV bool decode_ring_loop(Grp& g, const uint8_t* src, const int iend, uint8_t* dst, const int oend, int& ip_io, int& op_io, uint8_t* lds) {
  typedef typename Grp::LChunk LChunk;
  const uint32_t KW = g.ring_bytes(), KS = g.ring_stream(), STEP = 64u;
  uint32_t ip = (uint32_t)ip_io, op = (uint32_t)op_io;
  const uint32_t ilim = (uint32_t)iend - 306u, olim = (uint32_t)oend - 606u;
  g.ring_begin(lds, dst);
  // stream ring: holds [.., avail); `fetched` = end of what has been requested
  uint32_t avail = ip & ~(STEP - 1u);
  while (avail < ip + 192u) { g.rs_put(avail, g.rs_fetch(src, avail)); avail += STEP; }
  uint32_t fetched = avail;
  LChunk rf = g.rs_fetch(src, fetched - STEP);
  uint32_t rf_pos = fetched - STEP;
  // output ring: everything below op is in memory.  The flusher works in address-aligned steps, so the ring is given the aligned step
  // that contains op (its bytes below op from memory); fl = first position that is not flushed, a multiple of 64 in address space
  const uint32_t mt0 = op;                                       // memory is valid below max(fl, mt0)
  uint32_t fl = op - ((g.ring_dbase() + op) & (STEP - 1u));
  const uint32_t rlo = fl;                                       // the ring holds nothing below this position
  g.rg_write(fl, g.step_load(dst + fl));
  uint32_t hw = fl + STEP;                                       // end of everything ever written into the ring: it holds [hw - KW, hw)
  // slots: literals v, match source from the ring (ul) / from memory (ug), nr = which of the two is the source; sop = output position
  // of the sequence, lit = its literal length
  LChunk v0 = LChunk(), ul0 = LChunk(), ug0 = LChunk(), v1 = LChunk(), ul1 = LChunk(), ug1 = LChunk(), v2 = LChunk(), ul2 = LChunk(), ug2 = LChunk();
  uint32_t sop0 = op, sop1 = op, sop2 = op, lit0 = 0, lit1 = 0, lit2 = 0;
  bool nr0 = true, nr1 = true, nr2 = true;
  (void)v2; (void)ul2; (void)ug2; (void)sop2; (void)lit2; (void)nr2;
  uint32_t t4 = g.rs_ld32(ip);

#define LZ4HIP_RETIRE(k) do { const LChunk u_ = Grp::pick(nr##k, ul##k, ug##k); g.rg_write(sop##k, v##k); g.rg_write(sop##k + lit##k, u_); \
                              hw = sop##k + lit##k + STEP; } while (0)
#define LZ4HIP_AIM(k) do { sop##k = aim; lit##k = 0u; } while (0)
#define LZ4HIP_FLUSH_STEP do { g.step_store(dst + fl, g.rg_read(fl)); fl += STEP; } while (0)
#define LZ4HIP_FLUSH_ALL do { while ((int32_t)(op - fl) > 0) LZ4HIP_FLUSH_STEP; } while (0)
#define LZ4HIP_REFILL_FETCH if ((fetched + STEP <= (uint32_t)iend) & (fetched + STEP <= (ip & ~(STEP - 1u)) + KS)) { rf_pos = fetched; rf = g.rs_fetch(src, rf_pos); fetched = rf_pos + STEP; }
#define LZ4HIP_REFILL_PUT if (fetched != avail) { g.rs_put(rf_pos, rf); avail = rf_pos + STEP; }
#if LZ4HIP_RING_SLOTS == 2
#define LZ4HIP_TRIP(c, a, REFILL) LZ4HIP_TRIP_(c, a, c, (void)0, (void)0, REFILL)
#else
#define LZ4HIP_TRIP(c, a, b, REFILL) LZ4HIP_TRIP_(c, a, b, LZ4HIP_RETIRE(b), LZ4HIP_AIM(b), REFILL)
#endif
  // one trip: fills slot c, puts slot a (the oldest) into the ring; n = the slot that is the oldest afterwards (its sequence starts
  // where the retired bytes end: the flusher's bound); RETIRE_REST / AIM_REST: the other waiting slots, oldest first
#define LZ4HIP_TRIP_(c, a, n, RETIRE_REST, AIM_REST, REFILL)                                                                    \
  {                                                                                                                            \
    uint32_t lit = (t4 >> 4) & 15u, ml = t4 & 15u;                                                                             \
    const uint32_t e1 = (t4 >> 8) & 255u;                                                                                      \
    const bool l15 = lit == 15u;                                                                                               \
    lit += l15 ? e1 : 0u;                                                                                                      \
    const uint32_t hdr = l15 ? 2u : 1u;                                                                                        \
    const uint64_t o8 = g.rs_ld64(ip + hdr + lit);   /* (a literal run over 64 reads stale ring bytes: the trip leaves below) */ \
    const uint32_t off = (uint32_t)o8 & 0xFFFFu;                                                                               \
    const bool m15 = ml == 15u;                                                                                                \
    const uint32_t e2 = (uint32_t)(o8 >> 16) & 255u;                                                                           \
    ml += (m15 ? e2 : 0u) + 4u;                                                                                                \
    const uint32_t nxt = (uint32_t)(o8 >> (m15 ? 24 : 16));                                                                    \
    const uint32_t adv = hdr + lit + (m15 ? 3u : 2u);                                                                          \
    const uint32_t mpos = op + lit - off;            /* where the match copies from */                                         \
    bool near = (mpos >= rlo) & (mpos + KW >= hw);   /* the ring holds the source (unless it reaches into waiting bytes) */     \
    if (LZ4HIP_UNLIKELY((l15 & (e1 == 255u)) | (m15 & (e2 == 255u)) | (off - 1u >= op + lit) | (lit > STEP) | (ml > STEP) |      \
                        (!near & (mpos + ml > fl) & (mpos + ml > mt0)))) {   /* not for this loop (nothing of it done) */      \
      LZ4HIP_RETIRE(a); RETIRE_REST; LZ4HIP_FLUSH_ALL; break; }                                                                \
    if (LZ4HIP_UNLIKELY(mpos + ml > sop##a)) {                                                                                 \
      /* the source reaches into bytes that wait in a slot, or into this sequence's own bytes: the slots first */              \
      LZ4HIP_RETIRE(a); RETIRE_REST;                                                                                           \
      uint32_t aim = op;                                                                                                       \
      LZ4HIP_AIM(a); AIM_REST;                                                                                                 \
      near = (mpos >= rlo) & (mpos + KW >= hw);                                                                                \
      if (mpos + ml > op) {                          /* ... its own literals or its own output: copied inside the ring */     \
        if (!near) { LZ4HIP_FLUSH_ALL; break; }      /* (a source the ring does not hold that overlaps its match: exact path) */ \
        g.rg_write(op, g.rs_step(ip + hdr));                                                                                   \
        g.rg_replicate(op + lit, off, ml);                                                                                     \
        hw = op + lit + ml > op + STEP ? op + lit + ml : op + STEP;                                                            \
        op += lit + ml; ip += adv; t4 = nxt;                                                                                   \
        aim = op;                                                                                                              \
        LZ4HIP_AIM(c); LZ4HIP_AIM(a); AIM_REST;                                                                                \
        if (fetched != avail) { g.rs_put(rf_pos, rf); avail = rf_pos + STEP; }   /* (a piece on its way lands before the rotation restarts) */ \
        while (op - fl >= STEP) LZ4HIP_FLUSH_STEP;                                                                             \
        if (!((ip <= ilim) & (op <= olim) & (ip + 80u <= avail))) { LZ4HIP_FLUSH_ALL; break; }                                 \
        continue;                                                                                                              \
      }                                                                                                                        \
    }                                                                                                                          \
    if (REFILL == 1) { LZ4HIP_REFILL_FETCH }                                                                                   \
    v##c = g.rs_step(ip + hdr);                                                                                                \
    ul##c = g.rg_read(mpos);                                                                                                   \
    ug##c = g.step_load(dst + (near ? fl : mpos));   /* (a near match loads a step it does not use: every trip the same operations) */ \
    sop##c = op; lit##c = lit; nr##c = near;                                                                                   \
    LZ4HIP_RETIRE(a);                                                                                                          \
    if (sop##n - fl >= STEP) {                       /* a whole aligned step lies below the waiting sequences: to memory */     \
      LZ4HIP_FLUSH_STEP;                                                                                                       \
      while (LZ4HIP_UNLIKELY(sop##n - fl > KW - 256u)) LZ4HIP_FLUSH_STEP;   /* (the ring must keep room for two more sequences) */ \
    }                                                                                                                          \
    if (REFILL == 2) { LZ4HIP_REFILL_PUT }                                                                                     \
    op += lit + ml; ip += adv; t4 = nxt;                                                                                       \
    if (!((ip <= ilim) & (op <= olim) & (ip + 80u <= avail))) { RETIRE_REST; LZ4HIP_RETIRE(c); LZ4HIP_FLUSH_ALL; break; }       \
  }
  for (;;) {
#if LZ4HIP_RING_SLOTS == 2
    LZ4HIP_TRIP(0, 1, 1)
    LZ4HIP_TRIP(1, 0, 2)
#else
    LZ4HIP_TRIP(0, 1, 2, 1)
    LZ4HIP_TRIP(1, 2, 0, 2)
    LZ4HIP_TRIP(2, 0, 1, 0)
#endif
  }
#undef LZ4HIP_TRIP
#undef LZ4HIP_TRIP_
#undef LZ4HIP_REFILL_FETCH
#undef LZ4HIP_REFILL_PUT
#undef LZ4HIP_FLUSH_ALL
#undef LZ4HIP_FLUSH_STEP
#undef LZ4HIP_AIM
#undef LZ4HIP_RETIRE
  // (every way out has put the waiting sequences into the ring, oldest first, and flushed the ring up to op -- the last step may
  // carry ring bytes past op: positions of this block that are written again, op <= oend - 606)
  ip_io = (int)ip; op_io = (int)op;
  return (ip <= ilim) & (op <= olim) & (ip + 80u > avail) & (ip + 320u <= (uint32_t)iend);
}

}  // namespace lz4hip

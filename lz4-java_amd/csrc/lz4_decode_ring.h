// lz4_decode_ring.h -- the ring loop of the block decoder (lz4_decode_core.h, PIPE == 3): an interior loop whose trip touches
// memory only for what MUST come from memory, and has no branch in it.
//
// Same sequences, same bytes as the other interior loops of decode_block (LZ4_decompress_safe / _fast of liblz4 1.9.3,
// /root/reference/src/jni/net_jpountz_lz4_LZ4JNI.c:216 / :169).  What a trip works on:
//  * the compressed stream of the block is staged in an LDS ring (as in the deep loop, lz4_decode_deep.h): the token parse is a chain
//    of LDS reads, the literals come from LDS;
//  * the block's RECENT OUTPUT lives in a second LDS ring of KW bytes (group_dev.h rg_*): sequences are written into the ring, in
//    sequence order; a match whose source lies inside the ring is an LDS -> LDS copy (near match), only a source the ring no longer
//    holds is a load from the block's output in memory (far match) -- and that load is never waited for in the trip that requests
//    it: slots hold the pieces whose match source is on its way, a trip fills one and puts the oldest into the ring;
//  * output leaves the ring as whole, ADDRESS-ALIGNED 64-byte steps (the ring is indexed by address, so the flusher's LDS reads and
//    its stores are aligned): every output byte is stored to memory exactly once, no partial line is ever written twice;
//  * the blocks of a wavefront run in lock step, so a trip costs what the slowest path through it costs: there is ONE path.  A trip
//    emits a piece = {<= 64 literal bytes, <= 64 match bytes that do not reach into waiting or own bytes}; everything irregular is a
//    piece of another shape, not a branch:
//      - a match source that reaches into bytes still waiting in a slot: the block STALLS for this trip (an empty piece);
//      - a source that reaches into the sequence's own literals: the literals alone are this trip's piece, the whole match is
//        carried in {cm, coff} to the next trips;
//      - a match that overlaps its own output (offset < length), or is longer than 64: the piece is min(length, offset, 64) bytes,
//        the rest is carried in {cm, coff} and copied by the next trips -- with the offset DOUBLED after every piece that was a whole
//        period (any multiple of the period is a period), so a run reaches 64 bytes per trip after six pieces;
//      - a literal run of more than 60 bytes: 60 per trip, the rest is carried in {cl, ipl};
//      - the flusher behind, the stream ring short of bytes at a sequence start: stalls as well;
//    what is left -- literal runs of more than ~180 bytes, length runs of two or more bytes, invalid offsets, a source neither the ring nor flushed
//    memory holds (right behind an entry only), the end of the loop's range -- FREEZES the block (it stalls until the loop is left);
//    the wavefront leaves the loop a few trips after the first block froze, each frozen block does its one sequence through memory
//    (or leaves for the exact code of decode_block), re-seeds its rings, and the loop is entered again.
// With KW = 512 (4 lanes per block, 16 blocks per wavefront) this is the loop for full batches of 64 KiB blocks -- text (offsets of a
// few hundred bytes stay on chip, the rest is pipelined) and far-match data alike; with KW = 4096 (8 / 16 lanes) for big blocks with a
// short match window (BASELINE configs[2]: four matches of five never leave the chip).
#pragma once
#include <stdint.h>
#ifndef LZ4HIP_UNLIKELY
#define LZ4HIP_UNLIKELY(x) __builtin_expect(!!(x), 0)
#endif
#ifndef LZ4HIP_RING_SLOTS
#define LZ4HIP_RING_SLOTS 4      /* pieces in the pipeline: a far source has LZ4HIP_RING_SLOTS - 1 trips to arrive (2, 3, 4, 6 or 8; measured on 16384 x 4 MiB: 765 / 845 / 850 GB/s with 3 / 4 / 6) */
#endif
#ifndef LZ4HIP_RING_PATIENCE
#define LZ4HIP_RING_PATIENCE 4   /* trips the wavefront goes on after the first block froze */
#endif

namespace lz4hip {

// entry: 64 <= op (the first aligned step of the flusher lies inside the block's slot), ip + 320 <= iend (the ring's first pieces are
// readable), ip <= iend - 306, op <= oend - 606 (the interior loops' distance from both ends).
// returns true when the block can come again (it left only because the staged stream ran short near the end of the loop's range).
template <class Grp>
LZ4HIP_DEV bool decode_ring_loop(Grp& g, const uint8_t* src, const int iend, uint8_t* dst, const int oend, int& ip_io, int& op_io, uint8_t* lds) {
  typedef typename Grp::LChunk LChunk;
  const uint32_t KW = g.ring_bytes(), KS = g.ring_stream();
  const uint32_t STEP = g.ring_step();     // bytes of a step: what the block's lanes move at once, the flusher's and the refill's unit (64; 16 with a lane per block)
  const uint32_t PIECE = g.ring_piece();   // bytes of literals / of match a trip emits at most (60: a 64-byte step written at an odd address covers 64 - 3; 16)
  uint32_t ip = (uint32_t)ip_io, op = (uint32_t)op_io;
  const uint32_t ilim = (uint32_t)iend - 306u, olim = (uint32_t)oend - 606u;
  g.ring_begin(lds, dst);
  uint32_t avail, fetched, rf_pos, fl, rlo, mt0;
  uint32_t t4 = g.ld32(src + ip);      // the token word at ip
  // what is left of the sequence in front of ip (its token is consumed when its first piece is emitted): cl literal bytes from stream
  // position ipl on, then cm match bytes at offset coff
  uint32_t cl = 0, ipl = 0, cm = 0, coff = 0;
  LChunk rf;
  bool leave = false;
#ifdef LZ4HIP_RING_DBG   /* developer build: what the loop did (tools/ring_stats.py) */
  uint32_t dbg_trips = 0, dbg_stall = 0, dbg_frozen = 0, dbg_seeds = 0, dbg_wait = 0, dbg_wtrips = 0;
#define LZ4HIP_RING_COUNT(stall_, frozen_, wait_) do { dbg_trips++; dbg_wtrips += g.first_active() ? 1u : 0u; dbg_stall += (stall_); dbg_frozen += (frozen_); dbg_wait += (wait_); } while (0)
#else
#define LZ4HIP_RING_COUNT(stall_, frozen_, wait_) do { } while (0)
#endif
  for (;;) {
#ifdef LZ4HIP_RING_DBG
    dbg_seeds++;
#endif
    // ---- (re-)seed the rings: everything below op is in memory ----
    // stream ring: holds [.., avail); `fetched` = end of what has been requested
    avail = ip & ~(STEP - 1u);
    while (avail < ip + 192u) { g.rs_put(avail, g.rs_fetch(src, avail)); avail += STEP; }
    fetched = avail;
    rf = g.rs_fetch(src, fetched - STEP);
    rf_pos = fetched - STEP;
    // output ring: the flusher works in address-aligned steps, so the ring is given the aligned step that contains op (its bytes
    // below op from memory); fl = first position that is not flushed, a multiple of 64 in address space
    mt0 = op;                                              // memory is valid below max(fl, mt0)
    fl = op - ((g.ring_dbase() + op) & (STEP - 1u));
    rlo = fl;                                              // the ring holds nothing below this position
    g.rg_seed(fl, g.step_load(dst + fl));
    // slots: literals v and -- far matches only -- the match source from memory (ug) of a piece; sop = its output position, lit = its
    // literal length, mp = where its match copies from, nr = the source is read from the ring (when the piece is put there)
    LChunk v0 = LChunk(), ug0 = LChunk(), v1 = LChunk(), ug1 = LChunk(), v2 = LChunk(), ug2 = LChunk(), v3 = LChunk(), ug3 = LChunk();
    LChunk v4 = LChunk(), ug4 = LChunk(), v5 = LChunk(), ug5 = LChunk(), v6 = LChunk(), ug6 = LChunk(), v7 = LChunk(), ug7 = LChunk();
    uint32_t sop0 = op, sop1 = op, sop2 = op, sop3 = op, sop4 = op, sop5 = op, sop6 = op, sop7 = op;
    uint32_t lit0 = 0, lit1 = 0, lit2 = 0, lit3 = 0, lit4 = 0, lit5 = 0, lit6 = 0, lit7 = 0;
    uint32_t mp0 = op, mp1 = op, mp2 = op, mp3 = op, mp4 = op, mp5 = op, mp6 = op, mp7 = op;
    bool nr0 = true, nr1 = true, nr2 = true, nr3 = true, nr4 = true, nr5 = true, nr6 = true, nr7 = true;
    (void)v2; (void)ug2; (void)v3; (void)ug3; (void)sop2; (void)sop3; (void)lit2; (void)lit3; (void)mp2; (void)mp3; (void)nr2; (void)nr3;
    (void)v4; (void)ug4; (void)v5; (void)ug5; (void)sop4; (void)sop5; (void)lit4; (void)lit5; (void)mp4; (void)mp5; (void)nr4; (void)nr5;
    (void)v6; (void)ug6; (void)v7; (void)ug7; (void)sop6; (void)sop7; (void)lit6; (void)lit7; (void)mp6; (void)mp7; (void)nr6; (void)nr7;
    bool frozen = false;
    uint32_t since = 0;                                    // trips since the first block of the wavefront froze (wave-uniform)

#define LZ4HIP_REFILL_FETCH if ((fetched + STEP <= (uint32_t)iend) & (fetched + STEP <= ((cl ? ipl : ip) & ~(STEP - 1u)) + KS)) { rf_pos = fetched; rf = g.rs_fetch(src, rf_pos); fetched = rf_pos + STEP; }
#define LZ4HIP_REFILL_PUT if (fetched != avail) { g.rs_put(rf_pos, rf); avail = rf_pos + STEP; }
    // one trip: fills slot c, puts slot a (the oldest: filled LZ4HIP_RING_SLOTS - 1 trips ago) into the ring.
    // Pieces go into the ring in sequence order, and a near source is read from the ring only WHEN ITS PIECE GOES IN -- behind the
    // piece's own literals: everything in front of the match is in the ring by then, so a source may reach into waiting pieces and
    // into its own literals at no cost, and a piece waits for nothing but a far source's load.  (The literals are written and the
    // source is requested at the head of the trip, the match is written at its end: the LDS round trip lies under the parse.)
#define LZ4HIP_TRIP(c, a, REFILL)                                                                                              \
    {                                                                                                                          \
      g.rg_write(sop##a, v##a);                                                                                                \
      const LChunk ul = g.rg_read(mp##a);                                                                                      \
      const bool fdue = sop##a - fl >= STEP;           /* a whole aligned step lies below the waiting pieces: to memory */      \
      const LChunk fx = g.rg_read_al(fl);                                                                                         \
      uint32_t lit = (t4 >> 4) & 15u, ml = t4 & 15u;                                                                           \
      const uint32_t e1 = (t4 >> 8) & 255u;                                                                                    \
      const bool l15 = lit == 15u;                                                                                             \
      lit += l15 ? e1 : 0u;                                                                                                    \
      const uint32_t hdr = l15 ? 2u : 1u;                                                                                      \
      const uint64_t o8 = g.rs_ld64(ip + hdr + lit);   /* (a literal run over 64 reads stale ring bytes: the block freezes) */  \
      uint32_t off = (uint32_t)o8 & 0xFFFFu;                                                                                   \
      const bool m15 = ml == 15u;                                                                                              \
      const uint32_t e2 = (uint32_t)(o8 >> 16) & 255u;                                                                         \
      ml += (m15 ? e2 : 0u) + 4u;                                                                                              \
      const uint32_t nxt = (uint32_t)(o8 >> (m15 ? 24 : 16));                                                                  \
      const uint32_t adv = hdr + lit + (m15 ? 3u : 2u);                                                                        \
      const bool inc = (cl | cm) != 0u;                /* the rest of a sequence is to be copied: no token is parsed */          \
      const uint32_t need = hdr + lit + 8u;            /* stream bytes the parse of this sequence reads */                     \
      const bool odd = !inc & ((l15 & (e1 == 255u)) | (m15 & (e2 == 255u)) | (off - 1u >= op + lit) | (need > KS - 72u) |        \
                               !((ip <= ilim) & (op <= olim)));                                                                \
      const bool hungry = !inc & (ip + (need < 80u ? 80u : need) > avail);   /* the stream ring is short of this sequence: wait for the next piece */ \
      lit = inc ? cl : lit; ml = inc ? cm : ml; off = inc ? coff : off;                                                        \
      const uint32_t lpos = inc ? ipl : ip + hdr;      /* where the literals are read from */                                  \
      const uint32_t le = lit < PIECE ? lit : PIECE;   /* this trip's literals; its match bytes only once the literals are done */ \
      const uint32_t cap = off < PIECE ? off : PIECE;                                                                          \
      const uint32_t me = (le == lit) ? (ml < cap ? ml : cap) : 0u;   /* a piece never reaches into its own match bytes */      \
      const uint32_t mpos = op + le - off;             /* where the match copies from */                                       \
      const bool near = (mpos >= rlo) & (mpos + KW >= op + 2u * STEP);   /* the ring will still hold the source when this piece goes in */ \
      const bool lost = (me != 0u) & !near & (mpos + me > fl) & (mpos + me > mt0);   /* ... nor does flushed memory hold it (yet) */         \
      frozen = odd | (lost & !fdue & (sop##a == op));  /* (a lost source may just be early while pieces wait or the flusher has steps to store) */ \
      const bool stall = odd | lost | hungry | (op - fl > KW - 5u * STEP) | (since >= LZ4HIP_RING_PATIENCE);                         \
      LZ4HIP_RING_COUNT(stall, frozen, !frozen & (since >= LZ4HIP_RING_PATIENCE));                                           \
      if (REFILL == 1) { LZ4HIP_REFILL_FETCH }         /* (a request in one trip, its bytes into the ring in the next: every conditional   \
                                                          memory operation makes the compiler's wait counts more careful) */     \
      v##c = g.rs_step(lpos);                                                                                                  \
      /* (a near or empty piece loads the step that was flushed last and does not use it: every trip issues the same operations,   \
         so the compiler can count how many a wait may leave outstanding) */                                                   \
      const bool nomem = near | stall | (me == 0u);    /* nothing is needed from memory */                                    \
      /* (only the lanes that hold bytes of the match ask memory for them -- every request is a 128-byte line on its way through   \
         the fabric, and the launch is bound by how many of those the memory system serves; the others, like a near or empty       \
         piece, load the step that was flushed last: every trip issues the same operations) */                                   \
      ug##c = g.step_load_upto(dst + mpos, nomem ? 0u : me, dst + ((fl < STEP ? STEP : fl) - STEP));                             \
      sop##c = op; lit##c = stall ? 0u : le; mp##c = mpos; nr##c = nomem;   /* (a stalled trip's piece is empty: it is aimed at bytes that are written again) */ \
      if (fdue) { g.step_store(dst + fl, fx); fl += STEP; }                                                                    \
      if (REFILL == 2) { LZ4HIP_REFILL_PUT }                                                                                   \
      const bool tok = !stall & !inc;                  /* a token is consumed: the rest of its match, if any, travels in cm */   \
      op += stall ? 0u : le + me;                                                                                              \
      ip = tok ? ip + adv : ip;                                                                                                \
      t4 = tok ? nxt : t4;                                                                                                     \
      cl = stall ? cl : lit - le;                                                                                              \
      ipl = stall ? ipl : lpos + le;                                                                                           \
      cm = stall ? cm : ml - me;                                                                                               \
      coff = stall ? coff : (me == off ? 2u * off : off);                                                                      \
      since += ((since != 0u) | g.any(frozen)) ? 1u : 0u;                                                                        \
      g.rg_write(sop##a + lit##a, Grp::pick(nr##a, ul, ug##a));                                                                \
    }
    // (values that come from memory are made to arrive HERE: a wait at the loop's head would be a wait for everything every trip)
    g.settle(t4);
    // the branch-free loop; it is left by the whole wavefront behind a round of trips in which everyone stalled (nothing waits in a
    // slot then: the pieces of such trips are empty).  ONE exit, at the end of the round: a second way back to the loop's head from
    // the middle of it would carry a trip's load across the head, and the compiler would wait for everything there.
    for (;;) {
      const bool last = since >= LZ4HIP_RING_PATIENCE;
#if LZ4HIP_RING_SLOTS == 2
      LZ4HIP_TRIP(0, 1, 1)
      LZ4HIP_TRIP(1, 0, 2)
#elif LZ4HIP_RING_SLOTS == 3
      LZ4HIP_TRIP(0, 1, 1)
      LZ4HIP_TRIP(1, 2, 2)
      LZ4HIP_TRIP(2, 0, 0)
#elif LZ4HIP_RING_SLOTS == 4
      LZ4HIP_TRIP(0, 1, 1)
      LZ4HIP_TRIP(1, 2, 2)
      LZ4HIP_TRIP(2, 3, 1)
      LZ4HIP_TRIP(3, 0, 2)
#elif LZ4HIP_RING_SLOTS == 6
      LZ4HIP_TRIP(0, 1, 1) LZ4HIP_TRIP(1, 2, 2) LZ4HIP_TRIP(2, 3, 1) LZ4HIP_TRIP(3, 4, 2) LZ4HIP_TRIP(4, 5, 1) LZ4HIP_TRIP(5, 0, 2)
#else
      LZ4HIP_TRIP(0, 1, 1) LZ4HIP_TRIP(1, 2, 2) LZ4HIP_TRIP(2, 3, 1) LZ4HIP_TRIP(3, 4, 2) LZ4HIP_TRIP(4, 5, 1) LZ4HIP_TRIP(5, 6, 2) LZ4HIP_TRIP(6, 7, 1) LZ4HIP_TRIP(7, 0, 2)
#endif
      if (last) break;
    }
#undef LZ4HIP_TRIP
#undef LZ4HIP_REFILL_FETCH
#undef LZ4HIP_REFILL_PUT
    // nothing waits in a slot (the last pieces were empty); ring and memory together hold everything below op.  Every block of
    // the wavefront goes through memory and a re-seed here, frozen or not (the ones that only waited for another block skip the
    // sequence step)
    while ((int32_t)(op - fl) > 0) { g.step_store(dst + fl, g.rg_read_al(fl)); fl += STEP; }   // whole steps; the last one may carry ring bytes past op: positions that are written again
    if ((cl | cm) != 0u) {                // the rest of a sequence that was under way: through memory
      g.copy_lits_wide(dst + op, src + ipl, cl);
      op += cl; cl = 0u;
      g.copy_match_wide(dst, op, coff, cm);
      op += cm; cm = 0u;
    } else if (frozen) {
      // one sequence through memory, with the plain interior loop's rules (lz4_decode_core.h): lengths with at most one extension
      // byte, a valid offset, far from both ends; anything else -- and the end of the loop's range -- is for the caller
      uint32_t lit = (t4 >> 4) & 15u, ml = t4 & 15u;
      const uint32_t e1 = (t4 >> 8) & 255u;
      const bool l15 = lit == 15u;
      lit += l15 ? e1 : 0u;
      const uint32_t hdr = l15 ? 2u : 1u;
      if ((l15 & (e1 == 255u)) | !((ip <= ilim) & (op <= olim))) leave = true;
      else {
        const uint64_t o8 = g.ld64(src + ip + hdr + lit);
        const uint32_t off = (uint32_t)o8 & 0xFFFFu;
        const bool m15 = ml == 15u;
        const uint32_t e2 = (uint32_t)(o8 >> 16) & 255u;
        ml += (m15 ? e2 : 0u) + 4u;
        if ((m15 & (e2 == 255u)) | (off - 1u >= op + lit)) leave = true;
        else {
          g.copy_lits_wide(dst + op, src + ip + hdr, lit);
          g.copy_match_wide(dst, op + lit, off, ml);
          op += lit + ml; ip += hdr + lit + (m15 ? 3u : 2u);
          t4 = g.ld32(src + ip);
        }
      }
    }
    if (leave | !((ip <= ilim) & (op <= olim) & (ip + 320u <= (uint32_t)iend))) break;
  }
#ifdef LZ4HIP_RING_DBG
  (void)dbg_trips; (void)dbg_stall; (void)dbg_frozen; (void)dbg_seeds; (void)dbg_wait; (void)dbg_wtrips;   /* (round 5: the counters of the developer build belong to the wave loop now, tools/wave_stats.py) */
#endif
#undef LZ4HIP_RING_COUNT
  ip_io = (int)ip; op_io = (int)op;
  return !leave & (ip <= ilim) & (op <= olim) & (ip + 320u <= (uint32_t)iend);
}

}  // namespace lz4hip

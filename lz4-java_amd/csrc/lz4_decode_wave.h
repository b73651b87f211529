// lz4_decode_wave.h -- the wave loop of the block decoder (lz4_decode_core.h, PIPE == 4): ONE WAVEFRONT PER BLOCK, for launches that
// cannot fill the GPU with blocks (round 5: the 8-GPU shard of BASELINE configs[2] is 2048 x 4 MiB per GPU, the Java single-call path is
// one block).
//
// Same sequences, same bytes as the other interior loops of decode_block (LZ4_decompress_safe / _fast of liblz4 1.9.3,
// /root/reference/src/jni/net_jpountz_lz4_LZ4JNI.c:216 / :169; block independence -- the only parallelism the reference guarantees --
// /root/reference/src/java/net/jpountz/lz4/LZ4FrameOutputStream.java:361-363).  The lane-group loops (4 .. 16 lanes per block, 4 .. 64
// blocks per wavefront in lock step) need tens of thousands of blocks to hide their per-sequence chain: every sequence is a round trip
// to memory for its match source, and the blocks of a wavefront pay for each other's irregular sequences.  Here a block owns a
// wavefront, and the chain of a sequence is ONE LDS round trip:
//  * the compressed stream is staged in an LDS ring of KS bytes (refilled 256 bytes per sequence, a request in one trip, its bytes
//    into the ring in the next), the block's recent output lives in an LDS ring of KW bytes (8 .. 64 KB: whatever the launch leaves a
//    wavefront of the CU's 160 KB) -- a match whose source the ring still holds never leaves the chip; with KW = 64 KB a 64 KiB block
//    is decoded entirely on chip; only a source the ring no longer holds is a load from the block's flushed output in memory;
//  * control flow is WAVE-UNIFORM: the token parse is scalar work on values made uniform with v_readfirstlane, and everything
//    irregular (a literal run over 252 bytes, a match that overlaps its own output or is longer than a piece, a far source, a stream
//    ring short of bytes) is a scalar branch, not a lock-step piece as in the ring loop;
//  * a trip is software-pipelined around its one wait: the header of sequence n + 1 (the 8 bytes at its offset position: offset,
//    match-length byte, the token of n + 2), its literals, the match source of sequence n and the flusher's step are requested
//    together; LDS operations of a wavefront execute in order, so the match source is read BEHIND the store of the sequence's own
//    literals and in front of nothing it could miss;
//  * all lanes move 4 bytes each: a piece is up to 252 bytes of literals or of match.  Stores are ALIGNED dwords (an LDS instruction
//    whose lanes are not aligned is served one lane per cycle, profiles/r04_lds_alignment.txt): lane l >= 1 of a piece written at
//    ring index w holds the piece's bytes [4 l - s, 4 l - s + 4), s = w & 3, fetched that way -- two aligned dwords around the source
//    position funnelled by v_alignbyte -- and lane 0 alone stores its 4 bytes where they belong;
//  * output leaves the ring as whole address-aligned 256-byte steps, every byte stored once; on entry and exit the partial steps
//    are stored byte-exactly (nothing outside [entry op, exit op) is written).
// What is not "simple" -- length runs of two or more extension bytes, invalid offsets, the last ~KB of the stream, a source that
// straddles the loop's entry position -- leaves the loop with ip / op at the sequence start: the exact code of decode_block does that
// one sequence with every check and comes back.
#pragma once
#include <stdint.h>

#ifndef LZ4HIP_WAVE_CONT
#define LZ4HIP_WAVE_CONT 1
#endif
#ifndef LZ4HIP_PRED_LOADS
#define LZ4HIP_PRED_LOADS 1      /* (group_dev.h vrun_load) the parallel loop's copy rounds read their 16-byte pieces only in the lanes whose run has them; 0: every active lane reads all (developer A/B builds) */
#endif
#ifndef LZ4HIP_RUN_TIERS
#define LZ4HIP_RUN_TIERS 1       /* (group_dev.h) 0: no SHORT instance of the parallel loop -- the copy rounds always in their 64-byte form (developer A/B builds) */
#endif
#ifndef LZ4HIP_SHORT_MIN_T
#define LZ4HIP_SHORT_MIN_T 48u   /* starts in a window from which it counts as full of sequences ... */
#endif
#ifndef LZ4HIP_SHORT_AFTER
#define LZ4HIP_SHORT_AFTER 4u    /* ... and so many of them in a row send the block to the loop's SHORT instance (decode_wave_par_loop) */
#endif
#ifndef LZ4HIP_WAVE_OVL
#define LZ4HIP_WAVE_OVL 1        /* a match that overlaps its own output is copied inside the pass (0: it cuts the pass and gets a one-sequence step; developer A/B builds) */
#endif
#ifndef LZ4HIP_WAVE_SKIP
#define LZ4HIP_WAVE_SKIP 1
#endif
#ifndef LZ4HIP_WALK_PAR_MIN
#define LZ4HIP_WALK_PAR_MIN 24u   /* starts in the window before from which a window's walk is done by pointer doubling (65: never; developer A/B builds) */
#endif

namespace lz4hip {

// entry: ip + 1024 <= iend (three 256-byte steps of the stream around ip are readable), ip <= iend - 306, op <= oend - 606 (the
// interior loops' distance from both ends: every check of liblz4's fast loop provably passes for a simple sequence).
// Leaves with ip / op at the start of the first sequence it did not decode; everything below op is in memory then.
template <class Grp>
LZ4HIP_DEV void decode_wave_loop(Grp& g, const uint8_t* src, const int iend, uint8_t* dst, const int oend, int& ip_io, int& op_io, uint8_t* lds) {
  typedef typename Grp::LChunk LChunk;
  typedef typename Grp::WPiece WPiece;
  constexpr uint32_t STEP = 256u, PIECE = 252u;
  const uint32_t KW = g.wv_ring(), KS = g.wv_stream();
  uint32_t ip = (uint32_t)ip_io, op = (uint32_t)op_io;
  const uint32_t ilim = (uint32_t)iend - 306u, olim = (uint32_t)oend - 606u;

  // ---- the first sequence, from memory: if it is not simple the loop is not entered at all (a block of long runs costs the exact
  // path what it always did, not a seed and a flush per sequence) ----
  uint32_t t4 = g.uni(g.ld32(src + ip));
  uint32_t lit = (t4 >> 4) & 15u, hdr = 1u;
  if (lit == 15u) {
    const uint32_t e1 = (t4 >> 8) & 255u;
    if (e1 == 255u) return;
    lit += e1; hdr = 2u;
  }
  uint64_t o8;
  {
    const uint64_t m8 = g.ld64(src + ip + hdr + lit);
    o8 = (uint64_t)g.uni((uint32_t)m8) | ((uint64_t)g.uni((uint32_t)(m8 >> 32)) << 32);
    const uint32_t off = (uint32_t)o8 & 0xFFFFu, e2 = (uint32_t)(o8 >> 16) & 255u;
    if ((((t4 & 15u) == 15u) & (e2 == 255u)) | (off - 1u >= op + lit)) return;
  }

  g.wv_begin(lds, dst);
  const uint32_t db = g.wv_dbase();        // ring index of output position p = (p + db) & (KW - 1): address-aligned steps are ring-aligned
  const uint32_t op0 = op;                 // memory holds everything below op0, the ring nothing
  // stream ring: holds [.., avail); `fetched` = end of what has been requested (the step in front of it may still be on its way, in rf)
  uint32_t avail = ip & ~(STEP - 1u);
  {
    const LChunk a0 = g.rs_fetch(src, avail), a1 = g.rs_fetch(src, avail + STEP), a2 = g.rs_fetch(src, avail + 2u * STEP);
    g.rs_put(avail, a0); g.rs_put(avail + STEP, a1); g.rs_put(avail + 2u * STEP, a2);
    avail += 3u * STEP;
  }
  uint32_t fetched = avail;
  LChunk rf = LChunk();
  // flusher, in ring coordinates (w = position + db): everything below fl is in memory; fl is a multiple of 256
  uint32_t fl = (op + db) & ~(STEP - 1u);
  WPiece vl = g.wv_get_stream(ip + hdr, op + db);   // the literals of the sequence at ip, shaped for their place in the ring
  uint32_t badw = 0u;                      // sign bit: the sequence at ip is not for this loop (found when its token was parsed)

  // All tests of a trip are SIGN BITS of differences (every quantity is below 2^31): a wavefront issues one instruction per ~4
  // cycles whatever its kind, so the ~45 scalar compare / select / mask instructions the boolean form compiled to cost as much as
  // the copies.
  for (;;) {
    // ---- sequence n: {ip, op, t4, lit, hdr} known, its header o8 and its literals vl have arrived ----
    const uint32_t off = (uint32_t)o8 & 0xFFFFu;
    const uint32_t tm = t4 & 15u;
    const uint32_t e2 = (uint32_t)(o8 >> 16) & 255u;
    const bool m15 = tm == 15u;
    const uint32_t ml = tm + 4u + (m15 ? e2 : 0u);
    const uint32_t mpos = op + lit - off;                 // where the match copies from (negative: invalid offset)
    const uint32_t send = mpos + ml;                      // end of the source
    const uint32_t op2 = op + lit + ml;
    // near (farw >= 0): the ring holds the source and will still hold it when the sequence's last piece is written (a piece written
    // at w touches [w, w + 256): the ring loses [.., w + 256 - KW)).  A far source must lie in flushed memory: one that straddles the
    // entry position may not be there yet
    const uint32_t farw = (mpos - op0) | (mpos + KW - (op2 + STEP));
    const uint32_t oddw = badw | (ilim - ip) | (olim - op) | (off - 1u) | mpos | (269u - tm - e2);
    if ((int32_t)oddw < 0) break;
    if (LZ4HIP_UNLIKELY((int32_t)farw < 0)) { if ((int32_t)((fl - (send + db)) & (op0 - send)) < 0) break; }
    // ---- sequence n + 1: its token is in nxt; request its header and (below) its literals ----
    const uint32_t ip2 = ip + hdr + lit + (m15 ? 3u : 2u);
    const uint32_t nxt = (uint32_t)(o8 >> (m15 ? 24 : 16));
    const uint32_t tl2 = (nxt >> 4) & 15u, e1 = (nxt >> 8) & 255u;
    const bool l15 = tl2 == 15u;
    const uint32_t lit2 = tl2 + (l15 ? e1 : 0u), hdr2 = l15 ? 2u : 1u;
    badw = 269u - tl2 - e1;                               // (negative: a literal-length run of two or more bytes)
    // the parse of n + 1 reads up to 12 bytes from the dword below its offset position; its literals lie in front of that
    const uint32_t need2 = ip2 + hdr2 + lit2 + 12u;
    if (LZ4HIP_UNLIKELY(need2 > avail)) {       // the stream ring is short (its newest step still on its way, or long sequences in a row)
      if (fetched != avail) { g.rs_put(fetched - STEP, rf); avail = fetched; }
      while ((need2 > avail) & (fetched + STEP <= (uint32_t)iend) & (fetched + STEP <= (ip & ~(STEP - 1u)) + KS)) {
        g.rs_put(fetched, g.rs_fetch(src, fetched));
        fetched += STEP; avail = fetched;
      }
      if (need2 > avail) badw = 0x80000000u;    // (the end of the stream is near: the loops behind this one do the rest)
    }
    const uint64_t h8 = g.rs_ld64(ip2 + hdr2 + lit2);
    const bool fdue = op + db - fl >= STEP;     // a whole aligned step lies below this sequence: to memory
    LChunk fx = LChunk();
    if (fdue) fx = g.wv_read_al(fl);
    // ---- the literals of n into the ring, those of n + 1 requested ----
    g.wv_put(op + db, vl);
    if (LZ4HIP_UNLIKELY(lit > PIECE)) g.wv_put(op + db + PIECE, g.wv_get_stream(ip + hdr + PIECE, op + db + PIECE));
    const WPiece vl2 = g.wv_get_stream(ip2 + hdr2, op2 + db);
    // ---- the match of n: behind its own literals in the LDS queue ----
    {
      const uint32_t pos = op + lit + db;       // ring coordinates
      if ((int32_t)farw >= 0) {
        g.wv_put(pos, g.wv_get_ring(pos - off, pos));
        if (LZ4HIP_UNLIKELY((int32_t)((off - ml) | (PIECE - ml)) < 0)) {   // overlaps its own output, or more than a piece: the rest in pieces
          uint32_t o = off, n = o < PIECE ? o : PIECE, p = pos, rem = ml;   // (the first piece carried min(ml, off, PIECE) useful bytes)
          do {
            rem -= n; p += n;
            o = n == o ? 2u * o : o;            // a whole period was copied: twice the period is a period
            n = rem < o ? rem : o;
            n = n < PIECE ? n : PIECE;
            g.wv_put(p, g.wv_get_ring(p - o, p));
          } while (rem > n);
        }
      } else {                                  // (off > KW - 1100: no overlap, at most two pieces)
        g.wv_put(pos, g.wv_get_mem(dst + mpos, pos));
        if (ml > PIECE) g.wv_put(pos + PIECE, g.wv_get_mem(dst + mpos + PIECE, pos + PIECE));
      }
    }
    // ---- flusher: the step requested above; a sequence of more than 256 bytes leaves more than one ----
    if (fdue) {
      g.wv_store(dst, fl, fx, op0 + db, 0xFFFFFFFFu);
      fl += STEP;
      while (LZ4HIP_UNLIKELY(op + db - fl >= STEP)) { g.wv_store(dst, fl, g.wv_read_al(fl), op0 + db, 0xFFFFFFFFu); fl += STEP; }
    }
    // ---- stream refill: when the ring has room for a step, the one requested ~ten sequences ago goes in (its wait is free by now:
    // a wait in the trip that requests it would be a memory round trip on the chain) and the next one is requested ----
    if ((fetched + STEP <= (uint32_t)iend) & (fetched + STEP <= (ip2 & ~(STEP - 1u)) + KS)) {
      if (fetched != avail) { g.rs_put(fetched - STEP, rf); avail = fetched; }
      rf = g.rs_fetch(src, fetched);
      fetched += STEP;
    }
    ip = ip2; op = op2; t4 = nxt; lit = lit2; hdr = hdr2;
    o8 = (uint64_t)g.uni((uint32_t)h8) | ((uint64_t)g.uni((uint32_t)(h8 >> 32)) << 32);
    vl = vl2;
  }
  // ---- leave: everything the ring holds below op goes to memory, whole steps first, the rest byte-exactly ----
  while (op + db - fl >= STEP) { g.wv_store(dst, fl, g.wv_read_al(fl), op0 + db, 0xFFFFFFFFu); fl += STEP; }
  if (op + db != fl) g.wv_store(dst, fl, g.wv_read_al(fl), op0 + db, op + db);
  ip_io = (int)ip; op_io = (int)op;
}


// ================================================================================================================================
// The PARALLEL wave loop (decode_block PIPE == 5): SEVERAL SEQUENCES OF THE BLOCK PER TRIP.
//
// Measured on the loop above (gpurun_out/r05a-c; PMC: profiles/r05_wave_notes.txt): a wavefront issues one instruction per 4 cycles
// whatever its kind, a trip of the one-sequence loop is ~198 instructions (131 of them scalar), i.e. ~1100 cycles per sequence and
// ~38 ms per 4 MiB block when a wavefront has a SIMD to itself -- 1.7x the lane-group loops on a nearly empty GPU and no more: with
// one sequence per trip the trip IS the floor.  Here a trip decodes every sequence that starts in a 256-byte window of the stream:
//  1. DISCOVERY, lane-parallel and speculative: lane l assumes a token at each of the window's bytes 4 l .. 4 l + 3 and decodes it
//     (token, literal-length byte, the offset word and match-length byte behind the literals: aligned dword reads, funnelled)
//     into a record {offset, literal length, match length} and the window position of the NEXT token (255: none / not simple);
//  2. WALK: the four next-positions of a lane are one dword, so the chain 0 -> next(0) -> ... is a scalar walk of v_readlane's
//     (8 instructions per sequence, no memory: group_dev.h vwalk, written by hand); the k-th start goes to lane k (v_writelane);
//  3. RECORDS, A LANE PER RUN: lane 2k takes the literals of the k-th sequence, lane 2k + 1 its match.  Both fetch the record of
//     their start from the lane that decoded it (ds_bpermute); the exclusive prefix sum (DPP) of the run lengths is every run's output
//     position; one ballot finds the first run that is not for this round: invalid offset, not simple, or -- the dependency rule -- a
//     match whose source reaches into THIS ROUND'S OWN OUTPUT (a source the ring no longer holds is read from the block's flushed
//     output in memory).  The next round starts there, behind this round's stores; a round that would be empty ends the trip, and its
//     first sequence is handed to the one-sequence step below;
//  4. COPIES, exact: literals stream ring -> output ring, matches output ring -> output ring, 16 bytes at a time and an 8 / 4 / 2 / 1
//     cascade for the rest -- no byte outside a run's own output is written, so the lanes need no order between them.
// What a trip cannot take as its FIRST sequence (a match that overlaps its own output, literal runs over 255 / matches over 259
// bytes) is decoded by `wave_single_step` -- the one-sequence trip above without
// its pipelining; what that cannot take either leaves the loop for the exact code of decode_block.
// ================================================================================================================================

// a match whose source lies in the output ring, wave-wide pieces: pos = where it goes (ring coordinates).  One piece for the usual match; one that overlaps its
// own output (offset < length) replicates by doubling.  Writes up to a piece beyond the match's end ("wild" bytes: whoever produces them later overwrites them)
template <class Grp>
LZ4HIP_DEV void wave_match_ring(Grp& g, const uint32_t pos, const uint32_t off, const uint32_t ml) {
  constexpr uint32_t PIECE = 252u;
  g.wv_put(pos, g.wv_get_ring(pos - off, pos));
  if ((int32_t)((off - ml) | (PIECE - ml)) < 0) {
    uint32_t o = off, n = o < PIECE ? o : PIECE, p = pos, rem = ml;
    do {
      rem -= n; p += n;
      o = n == o ? 2u * o : o;
      n = rem < o ? rem : o;
      n = n < PIECE ? n : PIECE;
      g.wv_put(p, g.wv_get_ring(p - o, p));
    } while (rem > n);
  }
}

// one sequence at ip through the rings, wave-wide pieces (the body of decode_wave_loop, not pipelined).  false: not for this loop.
template <class Grp>
LZ4HIP_DEV bool wave_single_step(Grp& g, const uint8_t* dst, uint32_t& ip, uint32_t& op, const uint32_t op0, const uint32_t fl, const uint32_t db,
                                 const uint32_t ilim, const uint32_t olim) {
  constexpr uint32_t STEP = 256u, PIECE = 252u;
  const uint32_t KW = g.wv_ring();
  const uint64_t t8 = g.rs_ld64(ip);
  const uint32_t t4 = g.uni((uint32_t)t8);
  const uint32_t tl = (t4 >> 4) & 15u, e1 = (t4 >> 8) & 255u;
  const bool l15 = tl == 15u;
  const uint32_t lit = tl + (l15 ? e1 : 0u), hdr = l15 ? 2u : 1u;
  const uint64_t h8 = g.rs_ld64(ip + hdr + lit);
  const uint64_t o8 = (uint64_t)g.uni((uint32_t)h8) | ((uint64_t)g.uni((uint32_t)(h8 >> 32)) << 32);
  const uint32_t off = (uint32_t)o8 & 0xFFFFu, tm = t4 & 15u, e2 = (uint32_t)(o8 >> 16) & 255u;
  const bool m15 = tm == 15u;
  const uint32_t ml = tm + 4u + (m15 ? e2 : 0u);
  const uint32_t mpos = op + lit - off, send = mpos + ml, op2 = op + lit + ml;
  const uint32_t farw = (mpos - op0) | (mpos + KW - (op2 + STEP));
  const uint32_t oddw = (269u - tl - e1) | (ilim - ip) | (olim - op) | (off - 1u) | mpos | (269u - tm - e2);
  if ((int32_t)oddw < 0) return false;
  if ((int32_t)farw < 0) { if ((int32_t)((fl - (send + db)) & (op0 - send)) < 0) return false; }
  g.wv_put(op + db, g.wv_get_stream(ip + hdr, op + db));
  if (lit > PIECE) g.wv_put(op + db + PIECE, g.wv_get_stream(ip + hdr + PIECE, op + db + PIECE));
  const uint32_t pos = op + lit + db;
  if ((int32_t)farw >= 0) {
    wave_match_ring(g, pos, off, ml);
  } else {
    g.wv_put(pos, g.wv_get_mem(dst + mpos, pos));
    if (ml > PIECE) g.wv_put(pos + PIECE, g.wv_get_mem(dst + mpos + PIECE, pos + PIECE));
  }
  ip += hdr + lit + (m15 ? 3u : 2u);
  op = op2;
  return true;
}

// entry: ip + 1536 <= iend, ip <= iend - 306, op <= oend - 606.  Leaves with ip / op at the first sequence it did not decode;
// everything below op is in memory then.
// SHORT: the loop for streams FULL of sequences (text: ~58 starts per window, runs of 2 .. 6 bytes) -- every pass picks the form of its copy rounds by
// its longest run (group_dev.h vcopy_run: pieces for runs shorter than 16 / 32 / up to 64 bytes; an unaligned LDS access costs a cycle per active lane,
// wanted or not: text 161 -> 210-220 GB/s).  A SECOND INSTANCE of the loop, not a test inside one: with the short forms present the compiler's code for the
// 64-byte form is 1.3 .. 6 % slower on every other kind of data, however the choice is made (per round, per window, per pass, behind an empty asm:
// profiles/r06_wave_segments.txt -- a test of what the window produced at the end of the trip was enough for 2 %; a third instance to come back to, 1.6 %).
// The plain instance returns true when LZ4HIP_SHORT_AFTER windows in a row held LZ4HIP_SHORT_MIN_T starts or more, and decode_block goes on in the SHORT
// instance (it starts with empty rings: sources below its entry come from memory, once per block).  48 starts: text has 58-65 per window; a bitmap's
// windows hold 36, of sequences of 26 bytes on average with runs of hundreds among them -- every pass takes the 64-byte form, which the SHORT instance
// does 4.5 % slower -- and App. F data 15.
template <class Grp, bool SHORT>
LZ4HIP_DEV bool decode_wave_par_loop(Grp& g, const uint8_t* src, const int iend, uint8_t* dst, const int oend, int& ip_io, int& op_io, uint8_t* lds) {
  typedef typename Grp::LChunk LChunk;
  typedef typename Grp::VU VU;
  typedef typename Grp::VB VB;
  constexpr uint32_t STEP = 256u, WIN = 256u, TRIPMAX = 2048u;   // a trip's window of the stream; the most output a trip produces
  constexpr uint32_t AHEAD = 512u;         // stream bytes a trip may read from ip on: a sequence that starts at window position <= 250 has its offset word at <= 250 + 2 + 255 and reads 4 bytes there; the copies' 80-byte reads start at <= 252
                                           // (512: what a 1 KB stream ring always holds in front of ip with one refill step on its way)
  const uint32_t KW = g.wv_ring(), KS = g.wv_stream();
  uint32_t ip = (uint32_t)ip_io, op = (uint32_t)op_io;
  const uint32_t ilim = (uint32_t)iend - 306u, olim = (uint32_t)oend - 606u;
  g.wv_begin(lds, dst);
  const uint32_t db = g.wv_dbase();
  const uint32_t op0 = op;
  uint32_t avail = ip & ~(STEP - 1u);
  {
    const LChunk a0 = g.rs_fetch(src, avail), a1 = g.rs_fetch(src, avail + STEP), a2 = g.rs_fetch(src, avail + 2u * STEP), a3 = g.rs_fetch(src, avail + 3u * STEP);
    g.rs_put(avail, a0); g.rs_put(avail + STEP, a1); g.rs_put(avail + 2u * STEP, a2); g.rs_put(avail + 3u * STEP, a3);
    avail += 4u * STEP;
  }
  LChunk rf0 = LChunk(), rf1 = LChunk();   // the steps requested at the top of a trip; they go into the ring at its end
  uint32_t fl = (op + db) & ~(STEP - 1u);
  const VU lane = g.vlane();
  const VU p0 = lane * 4u;
  uint32_t wild = op;                          // end of what one-sequence steps have written into the ring (see `bound`)
  uint32_t Tprev = 0u;                         // starts the window before this one held (which walk this one gets)
  uint32_t full = 0u;                          // windows in a row that were full of sequences
  bool other = false;                          // leave for the loop's SHORT instance
#ifdef LZ4HIP_RING_DBG   /* developer build: what the loop did (tools/wave_stats.py) */
  uint32_t dbg_trips = 0, dbg_seqs = 0, dbg_rounds = 0, dbg_single = 0, dbg_hungry = 0, dbg_T = 0;
#endif

  for (;;) {
    if (ip + 32u > (uint32_t)iend) break;       // (the end rules of step 3 stop in front of the stream's last 32 bytes)
    // ---- the stream ring holds what this trip may read ----
    if (LZ4HIP_UNLIKELY(ip + AHEAD > avail)) {
#ifdef LZ4HIP_RING_DBG
      dbg_hungry++;
#endif
      while ((ip + AHEAD > avail) & (avail + STEP <= (uint32_t)iend) & (avail + STEP <= (ip & ~(STEP - 1u)) + KS)) {
        g.rs_put(avail, g.rs_fetch(src, avail));
        avail += STEP;
      }
      // the stream's last, partial step: its bytes below iend (zeros behind them).  With it the WHOLE stream is in the ring and the windows go on
      // to the block's end -- a sequence the end rules take ends inside the stream; what a window reads behind iend is nothing a pass uses
      if ((ip + AHEAD > avail) & (avail < (uint32_t)iend) & (avail + STEP > (uint32_t)iend) & (avail + STEP <= (ip & ~(STEP - 1u)) + KS)) {
        g.rs_put(avail, g.rs_fetch_upto(src, avail, (uint32_t)iend));
        avail += STEP;
      }
      if ((ip + AHEAD > avail) & (avail < (uint32_t)iend)) break;
    }
    // ---- stream refill: REQUESTED here, at the top of the trip, PUT into the ring at its end -- the loads are a whole trip old when
    // they are waited for, and nothing is carried from trip to trip.  The ring keeps everything from ip & ~255 on: this trip's
    // reads.  One step per trip: what a window consumes (a trip that consumes more finds the ring short and takes the refill above) ----
    // A window whose starts are all taken consumes MORE than a step (its last sequence ends behind it): a second step is requested when the ring
    // holds less than a step beyond what this trip may read -- without it every ~30th trip found the ring short and waited for memory
    uint32_t nf = 0u;
    if ((avail + STEP <= (uint32_t)iend) & (avail + STEP <= (ip & ~(STEP - 1u)) + KS)) {
      rf0 = g.rs_fetch(src, avail);
      nf = 1u;
      if ((avail < ip + AHEAD + STEP) & (avail + 2u * STEP <= (uint32_t)iend) & (avail + 2u * STEP <= (ip & ~(STEP - 1u)) + KS)) {
        rf1 = g.rs_fetch(src, avail + STEP);
        nf = 2u;
      }
    }
    // ---- 1 + 2. discovery and walk.  The speculative part decodes only what the WALK needs: where a sequence that starts at window
    // position p would be followed by the next one -- token, literal-length byte: all inside the window registers, no LDS access;
    // the offset word and the match-length byte behind the literals are fetched in step 3, for the real starts only (a quarter of the
    // instructions, and 2 instead of 8 LDS reads per lane: this was 125 instructions per window).  posv: where sequence k starts,
    // relative to ip.  (The window is read as ALIGNED dwords funnelled in registers: five unaligned reads by 64 lanes kept the CU's
    // LDS busy for ~400 cycles per trip -- SQ_LDS_UNALIGNED_STALL was 80 % of SQ_LDS_IDX_ACTIVE and at 16 wavefronts per CU the LDS,
    // not the wavefronts, set the pace: 4096 x 4 MiB 110 ms, gpurun_out/r05g) ----
    VU posv = VU(0u);
    uint32_t T = 0u;
    VU blo, bhi;
    g.vs_win(ip, blo, bhi);                     // stream bytes [ip + 4 l, + 8)
    {
      VU nxpack = VU(0u);
#pragma unroll
      for (uint32_t j = 0; j < 4u; j++) {
        const VU w = j == 0u ? blo : ((blo >> (8 * (int)j)) | (bhi << (32 - 8 * (int)j)));   // bytes j, j + 1, ..
        const VU tl = (w >> 4) & 15u, e1 = (w >> 8) & 255u;
        const VB l15 = tl == 15u;
        // (a run of two or more length bytes makes this wrong -- and the sequence "not simple" in step 3: the trip ends in front of it)
        const VU nxt = p0 + (j + 3u) + Grp::vsel(l15, VU(1u), VU(0u)) + tl + Grp::vsel(l15, e1, VU(0u)) + Grp::vsel((w & 15u) == 15u, VU(1u), VU(0u));
        nxpack = nxpack | (Grp::vsel(nxt <= 250u, nxt, VU(255u)) << (8 * (int)j));
      }
      // the starts of the sequences in the window, the k-th to lane k: hop by hop (8 scalar instructions a start), or -- a window of text holds
      // ~58 starts: a third of its trip was this walk -- by pointer doubling (group_dev.h vwalk_par: ~190 vector instructions whatever the
      // count) when the window before this one was that full.  Either way the same posv and T
      if (Tprev >= LZ4HIP_WALK_PAR_MIN) Grp::vwalk_par(nxpack, lane, posv, T); else Grp::vwalk(nxpack, posv, T);
      Tprev = T;
    }
    if (!SHORT && LZ4HIP_RUN_TIERS) {           // (windows full of sequences: the loop's SHORT instance goes on from here)
      full = T >= LZ4HIP_SHORT_MIN_T ? full + 1u : 0u;
      if (full >= LZ4HIP_SHORT_AFTER) { other = true; break; }
    }
    // ---- 3. records, output positions: A LANE PER RUN -- lane 2k the literals of a sequence, lane 2k + 1 its match.  (A lane per
    // sequence copied two runs, 16 reads and 16 predicated stores a round: ~220 instructions of a trip of ~900 that is bound by
    // instruction issue.  With a lane per run the same bytes move with half the instructions, the prefix sum of the run lengths is
    // every run's output position, and a match may read its own sequence's literals: they belong to the round before its own.)
    // The wavefront holds 31 sequences that way: a window with more of them (text: up to 64) is taken in PASSES of 31, each pass from
    // the sequence the pass before it ended with -- discovery and walk are paid once per window ----
    const VB isM = (lane & 1u) != 0u;
    uint32_t tk = 0u, lend = 0u;                // sequences of the window taken so far, where the last one ends in the stream
    bool stuck = false, skip = false;
    // A window is taken in SEGMENTS: passes from the current output position on, until a pass is cut (a source the flusher has not written
    // yet, TRIPMAX, a sequence that is not for a pass) -- then the flusher runs, a sequence no pass takes gets its one-sequence step, and
    // the NEXT SEGMENT goes on with the same window's starts (LZ4HIP_WAVE_CONT; 0: a cut ends the trip and the rest of the window is
    // discovered and walked again)
    for (;;) {
    const uint32_t tk0 = tk;
    uint32_t opc = op;                          // the output position behind the sequences taken
    const bool skipnow = skip;                  // (the pass before this segment has seen that its first sequence is for no pass)
    skip = false;
    if (!(LZ4HIP_WAVE_SKIP && skipnow))
    for (;;) {
      const uint32_t np = T - tk < 31u ? T - tk : 31u;   // (31: the rounds' ballot keeps lane 63 out)
      const VU sq = (lane >> 1) + tk;           // this lane's sequence
      const VU pv = Grp::vshfl(posv, sq);       // where it starts, relative to ip
      const uint64_t actm = (1ull << (2u * np)) - 1ull;                        // (np <= 31)
      const VB act = Grp::vlanes(actm);
      // the sequence's header: its first bytes out of the window registers (two lane shuffles + a funnel), the offset word and the
      // match-length byte behind its literals from the stream ring
      const VU sl = pv >> 2;
      const VU hw = Grp::valignbyte(Grp::vshfl(bhi, sl), Grp::vshfl(blo, sl), pv & 3u);
      const VU tl = (hw >> 4) & 15u, tm = hw & 15u, e1 = (hw >> 8) & 255u;
      const VB l15 = tl == 15u;
      const VU lit = tl + Grp::vsel(l15, e1, VU(0u));
      const VU lp = pv + ip + Grp::vsel(l15, VU(2u), VU(1u));                   // stream position of the literals
      const VU ow = g.vs_ld32(lp + lit);
      const VU off = ow & 0xFFFFu, e2 = (ow >> 16) & 255u;
      const VB m15 = tm == 15u;
      const VU mlx = tm + Grp::vsel(m15, e2, VU(0u)), ml = mlx + 4u;
      const VU endp = (lp - ip) + lit + Grp::vsel(m15, VU(3u), VU(2u));          // where the next token lies, relative to ip
      // (lane sets are kept as wave-uniform MASKS -- a ballot per comparison, combined with scalar instructions: a ballot of a compound
      // bool costs the compiler a v_cndmask + v_cmp on top of the same scalar work)
      const uint64_t simplem = Grp::vballot(off != 0u) & Grp::vballot(lit < 255u) & Grp::vballot(mlx < 255u);   // lengths of 255 and more (every run of two or more length bytes) are not for a trip
      const VU len = Grp::vsel(isM, ml, lit);   // this lane's run
      const VU tot = Grp::vsel(act, len, VU(0u));
      const VU ex = Grp::vexcl_scan(tot);
      const VU o = ex + opc;                    // where the run's output starts
      const VU mp = o - off;                    // match lanes: where the match copies from (negative: invalid offset)
      const VU oe = o + tot;
      const VU send = mp + ml;
      // held: the ring has the source and keeps it while the pass is written (it touches at most [opc, bound): the ring loses what
      // lies below bound - KW); a source the ring does not hold is FAR and comes from the block's flushed output in memory -- which
      // has everything below the flusher's position (and below the loop's entry position)
      const uint32_t oe_all = Grp::vreadlane(oe, 2u * np - 1u);
      // A match that OVERLAPS its own output (offset < length: 0.5 % of the sequences of 4 MiB blocks, 0.8 % of App. F's) is no lane's run -- its source never
      // lies below a round's output.  It used to cut the pass (flusher, one-sequence step, a new pass for the rest of the window); now the round loop copies it
      // where it stands, wave-wide, by doubling (wave_match_ring), and goes on with the next lane.  Its pieces write up to a step beyond its end: a pass that
      // holds such a match counts that step into what it touches
      const uint64_t ovlm = Grp::vballot(off < ml);
      const uint32_t tb = (oe_all - op < TRIPMAX ? oe_all : op + TRIPMAX) + 16u + ((LZ4HIP_WAVE_OVL && (ovlm & actm) != 0ull) ? STEP : 0u);
      const uint32_t bound = (int32_t)(wild - tb) > 0 ? wild : tb;   // (... or what a one-sequence step before this trip has touched: its pieces are a whole step wide)
      const uint32_t memlim = fl > op0 + db ? fl - db : op0;
      constexpr uint64_t litm = 0x5555555555555555ull;                        // the even lanes: literal runs
      const uint64_t heldm = Grp::vballot(mp >= VU(op0)) & Grp::vballot((mp + KW) >= VU(bound));
      const uint64_t srcm = Grp::vballot(mp < VU(0x80000000u)) & (heldm | Grp::vballot(send <= VU(memlim)));   // match lanes: the source is valid and can be had
      // THE BLOCK'S END, sequence by sequence (lz4_decode_trio.h has the derivation): the lengths are known per lane, so the rule is liblz4's own
      // fast-loop rule -- literals that end 32 bytes in front of the stream's end, a match that ends more than 64 bytes in front of the
      // output's end -- instead of a blanket 306 / 606 bytes that left ~40 (text: ~110) sequences of every block to decode_block's exact code,
      // a memory round trip or two each: 6-9 % of a 64 KiB block's time.  The first sequence that does not pass ends the loop (its
      // one-sequence step keeps the blanket margins and refuses)
      const uint64_t okbm = actm & simplem & (litm | srcm) & Grp::vballot((lp + lit + 32u) <= VU((uint32_t)iend)) & Grp::vballot((oe + 64u) < VU((uint32_t)oend)) & Grp::vballot((oe - op) <= VU(TRIPMAX));
      const uint64_t farm = ~litm & ~heldm;
      // (mp "negative" -- an offset that reaches in front of the block -- is a huge unsigned number: tested by itself, the sums may wrap)
      // ---- 4. copies in DEPENDENCY ROUNDS, a lane per run, exact.  A round takes the runs from lane `a` on; a literal run has no
      // dependency, a match is taken when its whole source lies below the round's own output.  The first match whose source reaches
      // into it starts the next round, behind this round's stores -- LDS operations of a wavefront execute in order.  A round that
      // would be empty ends the trip: its first run is a match that reaches into its OWN output (or is not for a trip at all) ----
      const VU spv = Grp::vsel(isM, mp + db, lp);   // the run's source: ring coordinates of the match source / stream position of the literals
      // (the round's lane sets are wave-uniform MASKS, combined with scalar instructions: one vector compare per round)
      const uint64_t oddm = g.vodd_mask(o + db, len);
      const uint32_t tier = SHORT ? Grp::vrun_tier(len, actm) : 4u;
      uint32_t a = 0u;
      for (;;) {
        const uint32_t oa = Grp::vreadlane(o, a);
        const uint64_t below = (1ull << a) - 1ull;                            // the lanes the rounds before this one took (a <= 62)
        const uint64_t okm = (okbm & (litm | Grp::vballot(send <= VU(oa)))) | below;
        const uint32_t Te = (uint32_t)__builtin_ctzll(~okm | (1ull << 63));   // (< 64: a pass uses lanes 0 .. 61)
#ifdef LZ4HIP_WAVE_DBG   /* developer build of the simulator: why rounds end (tests/hostsim) */
        g.vnote(act, Grp::vlanes(simplem), !isM | (mp < VU(0x80000000u)), !isM | (send <= VU(oa)) | (lane < a), Grp::vlanes(litm | heldm) | (send <= VU(memlim)), Grp::vlanes(litm | heldm), (oe - op) <= VU(TRIPMAX), Grp::vlanes(okm));
#endif
        if (Te == a) {
          // an empty round.  At the match lane of a sequence every other rule passes -- simple, source valid and held, the block's ends, TRIPMAX -- it is
          // the overlap: the literals are done (the round before ended here), the match goes wave-wide
          if (LZ4HIP_WAVE_OVL && (a & 1u) != 0u && (((okbm & heldm & ovlm) >> a) & 1ull) != 0ull) {
            const uint32_t ml_a = Grp::vreadlane(ml, a);
            if ((int32_t)(bound - (oa + ml_a + STEP)) >= 0) {
              wave_match_ring(g, oa + db, Grp::vreadlane(off, a), ml_a);
              if ((int32_t)(oa + ml_a + STEP - wild) > 0) wild = oa + ml_a + STEP;
              a += 1u;
              if (a >= 2u * np) break;
              continue;
            }
          }
          break;
        }
        g.template vcopy_run<LZ4HIP_PRED_LOADS != 0>(o + db, !isM, spv, len, ((1ull << Te) - 1ull) & ~below, dst, mp, farm, oddm, tier);
#ifdef LZ4HIP_RING_DBG
        dbg_rounds++;
#endif
        a = Te;
        if (a >= 2u * np) break;
      }
      a &= ~1u;                                 // (literals whose match was not taken are written again by whoever takes the sequence: the same bytes)
      if (a == 0u) break;
      tk += a >> 1;
      opc = Grp::vreadlane(oe, a - 1u);
      lend = Grp::vreadlane(endp, a - 1u);
      // a pass that was cut at a sequence NO pass takes -- a length of 255 or more, a match that overlaps its own output (offset < length: its
      // source never lies below a round's output) -- is followed by that sequence's one-sequence step at once: the next segment's pass
      // would set itself up (~130 instructions), copy the literals and take nothing
      if (LZ4HIP_WAVE_SKIP && a < 2u * np) skip = (((~simplem | ovlm) >> a) & 1ull) != 0ull;
      if ((a < 2u * np) | (tk >= T)) break;     // the pass ended early, or the window is done
    }
#ifdef LZ4HIP_RING_DBG
    dbg_trips++; dbg_seqs += tk - tk0; dbg_T += T - tk0; dbg_single += tk == tk0 ? 1u : 0u;
#endif
    if (LZ4HIP_UNLIKELY(tk == tk0)) {
      // no pass takes sequence tk: its one-sequence step.  It reads up to AHEAD bytes from the sequence's own start: in the middle of a
      // window the requested refill step goes into the ring first, and a ring that is still short ends the trip in front of the sequence
      uint32_t ip1 = ip + Grp::vreadlane(posv, tk);
      if (tk != 0u) {
        if (nf != 0u) { g.rs_put(avail, rf0); avail += STEP; if (nf == 2u) { g.rs_put(avail, rf1); avail += STEP; } nf = 0u; }
        if (ip1 + AHEAD > avail) break;
      }
      const uint32_t ipq = ip1;
      if (!wave_single_step(g, dst, ip1, op, op0, fl, db, ilim, olim)) { ip = ipq; stuck = true; break; }
      wild = op + STEP;                         // its wave-wide pieces end up to a step behind the sequence: the ring has lost what lies KW below that
      tk++;
      lend = ip1 - ip;
    } else {
      op = opc;
    }
    if (!LZ4HIP_WAVE_CONT || tk >= T) break;
    // the next segment of this window: the flusher first (a far source wants everything below op in memory)
    while (op + db - fl >= STEP) { g.wv_store(dst, fl, g.wv_read_al(fl), op0 + db, 0xFFFFFFFFu); fl += STEP; }
    }
    if (stuck) break;
    ip += tk < T ? Grp::vreadlane(posv, tk) : lend;   // (every start was taken: the next token lies behind the last sequence)
    // ---- the requested steps into the ring (in front of the flusher's stores: stores count in vmcnt on this part) ----
    if (nf != 0u) {
      g.rs_put(avail, rf0);
      avail += STEP;
      if (nf == 2u) { g.rs_put(avail, rf1); avail += STEP; }
    }
    // ---- flusher: whole aligned steps below op.  (Requested at the TOP of the next trip instead, next to the window read, and stored behind
    // the discovery -- what the trio loop's copier does in front of its wait for a message, +5 % there -- it is 1.5 % SLOWER here: 2048 x 4 MiB
    // 429.7 -> 422.8 GB/s, 4096: 578.8 -> 572.0, gpurun_out/r06j: at two and four wavefronts per SIMD the round trips are hidden by the other
    // wavefronts already, and the three conditional stores in the middle of the trip cost more than they save) ----
    while (op + db - fl >= STEP) { g.wv_store(dst, fl, g.wv_read_al(fl), op0 + db, 0xFFFFFFFFu); fl += STEP; }
  }
  while (op + db - fl >= STEP) { g.wv_store(dst, fl, g.wv_read_al(fl), op0 + db, 0xFFFFFFFFu); fl += STEP; }
  if (op + db != fl) g.wv_store(dst, fl, g.wv_read_al(fl), op0 + db, op + db);
#ifdef LZ4HIP_RING_DBG
  g.wv_stats(dbg_trips, dbg_seqs, dbg_rounds, dbg_single, dbg_hungry, dbg_T, 0u);
#endif
  ip_io = (int)ip; op_io = (int)op;
  return other;
}

}  // namespace lz4hip

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

}  // namespace lz4hip

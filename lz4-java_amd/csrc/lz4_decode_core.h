// lz4_decode_core.h -- LZ4 block decoder for a small lane group (G lanes of a wavefront per block).
//
// Replaces, for the "HIP" family, LZ4_decompress_safe (/root/reference/src/jni/net_jpountz_lz4_LZ4JNI.c:216,
// reached from LZ4JNISafeDecompressor.java:34-43) and LZ4_decompress_fast (LZ4JNI.c:169,
// LZ4JNIFastDecompressor.java:35-44); SURVEY.md App. C.  Return values -- decoded size / bytes
// consumed, or -(input position)-1 -- equal liblz4 1.9.3's on valid AND malformed input: liblz4
// decodes in three tiers (a fast loop while >= 64 output bytes remain, a two-stage shortcut and a
// fully checked path) and which malformed streams are accepted depends on the tier, so the tiers'
// checks are kept.  What is different is the execution shape: every lane of the group runs the
// (cheap, serial) token parse redundantly -- no cross-lane traffic, parse loads are broadcast
// loads -- and the lanes split every literal / match copy between them.  A wavefront holds 64/G
// independent blocks; thousands of groups per CU-set keep HBM busy while each parse chain waits.
//
// Match copies read the block's own output from global memory: a wavefront's vector memory
// operations execute in order, so bytes stored by an earlier instruction of the same wave are
// visible to its later loads (same CU L1, no fence needed).  Within ONE copy instruction lanes
// never read bytes written by that instruction: overlapping matches (offset < bytes per step) use
// the replicate path, which only reads bytes that precede the match.
//
// Backend `Grp` (group_dev.h on the GPU, tests/hostsim/group_host.h in the CPU test-suite):
//   ld8/ld16/ld32/ld64(p)                   uniform loads (all lanes of the group, same address)
//   copy_lits(d, s, len, wild)               d[0..len) = s[0..len); wild: may touch <= 3 bytes past len on both sides
//   copy_match(dst, op, offset, len, wild)   dst[op+i] = dst[op-offset+i], byte-forward semantics;
//                                            offset 0 zero-fills (liblz4 1.9.3 behaviour)
#pragma once
#include <stdint.h>

#ifndef LZ4HIP_DEV
#if defined(__HIPCC__)
#define LZ4HIP_DEV __device__ __forceinline__
#else
#define LZ4HIP_DEV inline
#endif
#endif

#ifndef LZ4HIP_DECODE_REENTER
#define LZ4HIP_DECODE_REENTER 1   // 0: the interior loop is entered once per block (round 1 behaviour; developer A/B builds)
#endif
#include "lz4_decode_deep.h"
#include "lz4_decode_ring.h"
#include "lz4_decode_wave.h"
#include "lz4_decode_pair.h"
#include "lz4_decode_trio.h"
namespace lz4hip {

// SAFE: LZ4_decompress_safe(src, dst, src_size, out_size) -> decoded size or negative.
// !SAFE: LZ4_decompress_fast(src, dst, out_size) -> bytes consumed or negative; `src_size` is then
//        the readable capacity of the source slot and is never exceeded (liblz4 itself trusts the
//        stream blindly; results on valid streams are identical).
// PIPE: 1 = pipelined interior loop (below).  Pays when a wavefront has to make progress on its own (few, large blocks);
//       with the GPU full of blocks the plain loop is as fast.  2 = the deep loop of lz4_decode_deep.h (stream staged in LDS at
//       `stage`, Grp::kStreamLds bytes for this block; two slots: one match source of the block waits while the next is requested); the last 2 KB of the stream
//       are left to loop 1.
//       3 = the ring loop of lz4_decode_ring.h (stream AND recent output in LDS rings at `stage`, Grp::kRingLds bytes; near matches
//       never leave the chip, far ones are pipelined through slots, output leaves as whole aligned 64-byte steps).
//       4 = the wave loop of lz4_decode_wave.h (ONE WAVEFRONT PER BLOCK, Grp = BlockWaveDev: stream ring and an output ring of 8 .. 64 KB
//       in LDS at `stage`, wave-uniform parse, one LDS round trip per sequence): launches of few blocks.
//       5 = the parallel wave loop (same file, same rings): every sequence that starts in a 256-byte window of the stream per trip.
//       7 = the pair loop of lz4_decode_pair.h (TWO WAVEFRONTS PER BLOCK: the caller is the copier, `stage` = the pair's LDS: the rings of 4 / 5 and a mailbox; a second wavefront
//       runs pair_parser_service on the same LDS).
//       8 = the trio loop of lz4_decode_trio.h (THREE WAVEFRONTS PER BLOCK: the caller is the copier; a planner and a scanner wavefront run trio_service on the same LDS).
// STAGE: the interior loop writes through an LDS staging buffer (`stage`, Grp::kStage bytes for this block) and output leaves
//        it as whole 128-byte lines (group_dev.h st_*).
template <class Grp, bool SAFE, int PIPE = 0, bool STAGE = false>
LZ4HIP_DEV int decode_block(Grp& g, const uint8_t* src, int src_size, uint8_t* dst, int out_size, uint8_t* stage = nullptr) {
  int ip = 0, op = 0;
  const int iend = src_size, oend = out_size;  // iend: real end (SAFE) / read bound (!SAFE)
  const int shortiend = iend - (SAFE ? 14 : 8) - 2;
  const int shortoend = oend - (SAFE ? 14 : 8) - 18;
  uint32_t token;
  int length, offset, cpy;

#define LZ4HIP_LEN_CAP 0x7F000000  /* a run this long can never fit; stops 32-bit wrap on absurd input */
#define LZ4HIP_NEED_IN(k) do { if (!SAFE && ip + (int)(k) > iend) goto output_error; } while (0)
  if (out_size < 0 || src_size < 0) return -1;
  if (SAFE && out_size == 0) return (src_size == 1 && g.ld8(src) == 0) ? 0 : -1;
  if (!SAFE && out_size == 0) { LZ4HIP_NEED_IN(1); return g.ld8(src) == 0 ? 1 : -1; }
  if (SAFE && src_size == 0) return -1;

  if (oend - op >= 64) {
  interior:
    // ---- tier 1, interior fast loop.  While the block is far from both buffer ends (ip <= iend-306, op <= oend-606)
    // and the sequence is "simple" -- literal and match lengths need at most ONE extension byte (<= 269 / <= 273),
    // offset <= op -- every check of liblz4's fast loop provably passes (a sequence consumes <= 274 and produces
    // <= 542 bytes), so only those few conditions are tested.  Anything else falls through, with ip/op still at the
    // sequence start, to the exact tier-1 code below, which re-decodes that sequence with all checks. ----
    // The 8 bytes fetched at the offset position also hold the NEXT sequence's token (and its first length byte), so the
    // steady state costs two dependent loads per sequence: {offset word + literals} and {match source}.
    if constexpr (PIPE == 2) {
      if (ip + 2048 <= iend && ip <= iend - 306 && op <= oend - 606)
        if (decode_deep_loop(g, src, iend, dst, oend, ip, op, stage)) goto interior;
    }
    if constexpr (PIPE == 3) {   // the ring loop (lz4_decode_ring.h): stream and recent output in LDS at `stage` (Grp::kRingLds bytes)
      // It wants 64 output bytes in front of it (its flusher stores whole aligned steps).  They are decoded HERE, by the plain
      // interior loop, and not by the exact code below one sequence per round of `goto interior`: the blocks of a wavefront need
      // different numbers of sequences for their first 64 bytes, and blocks that reach the ring loop in different rounds of an
      // enclosing loop run it one after the other, not side by side (measured: 5.4 of 16 blocks per trip, 3x the time).
      if (op < 64 && ip <= iend - 306 && op <= oend - 606) {
        uint32_t t4 = g.ld32(src + ip);
        do {
          int lit = (int)((t4 >> 4) & 15u), ml = (int)(t4 & 15u), hdr = 1;
          if (lit == 15) {
            const uint32_t e = (t4 >> 8) & 255u;
            if (e == 255u) break;
            lit += (int)e;
            hdr = 2;
          }
          const uint64_t o8 = g.ld64(src + ip + hdr + lit);
          const int off = (int)((uint32_t)o8 & 0xFFFFu);
          int adv = hdr + lit + 2;
          uint32_t nxt = (uint32_t)(o8 >> 16);
          if (ml == 15) {
            const uint32_t e = nxt & 255u;
            if (e == 255u) break;
            ml += (int)e;
            adv++;
            nxt = (uint32_t)(o8 >> 24);
          }
          ml += 4;
          if (off > op + lit) break;   // invalid offset: the exact path produces liblz4's error code
          g.copy_lits_wide(dst + op, src + ip + hdr, (uint32_t)lit);
          op += lit;
          g.copy_match_wide(dst, (uint32_t)op, (uint32_t)off, (uint32_t)ml);
          op += ml;
          ip += adv;
          t4 = nxt;
        } while (op < 64 && ip <= iend - 306 && op <= oend - 606);
      }
      if (op >= 64 && ip + 320 <= iend && ip <= iend - 306 && op <= oend - 606)
        if (decode_ring_loop(g, src, iend, dst, oend, ip, op, stage)) goto interior;
    }
    if constexpr (PIPE == 4) {   // the wave loop (lz4_decode_wave.h); what it leaves at ip is done by the exact code below, which comes back
      if (ip + 1024 <= iend && ip <= iend - 306 && op <= oend - 606) decode_wave_loop(g, src, iend, dst, oend, ip, op, stage);
    }
    if constexpr (PIPE == 5 || PIPE == 6) {   // the parallel wave loop: several sequences of the block per trip (6: one window of the stream per trip whatever the backend's ring -- the simulator's way to the form a 1 KB stream ring runs)
      if (ip + 1536 <= iend && ip <= iend - 306 && op <= oend - 606)
        if (decode_wave_par_loop<Grp, false>(g, src, iend, dst, oend, ip, op, stage))             // (true: the stream is full of sequences -- the loop's SHORT instance)
          if (ip + 1536 <= iend && ip <= iend - 306 && op <= oend - 606) (void)decode_wave_par_loop<Grp, true>(g, src, iend, dst, oend, ip, op, stage);
    }
    if constexpr (PIPE == 7) {   // the pair loop (lz4_decode_pair.h): this wavefront copies, its partner wavefront parses the stream a trip or two ahead
      if (ip + 1536 <= iend && ip <= iend - 306 && op <= oend - 606) decode_pair_loop<Grp>(g, src, iend, dst, oend, ip, op, stage);
    }
    if constexpr (PIPE == 8) {   // the trio loop (lz4_decode_trio.h): this wavefront copies, a planner and a scanner wavefront run ahead in the stream
      if (ip + 1536 <= iend && ip <= iend - 306 && op <= oend - 606) decode_trio_loop<Grp>(g, src, iend, dst, oend, ip, op, stage);
    }
    if ((PIPE == 1 || (PIPE == 2 && ip + 2048 > iend) || (PIPE == 3 && ip + 320 > iend) || (PIPE == 4 && ip + 1024 > iend) || ((PIPE == 5 || PIPE == 6 || PIPE == 7 || PIPE == 8) && ip + 1536 > iend)) && ip <= iend - 306 && op <= oend - 606) {   // (2 .. 8: only the tail of the stream)
      // ---- the same loop, software-pipelined.  A wavefront's memory operations retire in order, so a wait for a load also
      // waits for every OLDER store.  Here (a) the next sequence's offset word is requested as soon as this sequence's header
      // is parsed, (b) a "simple" sequence (literals and match take one step each, match source entirely before the
      // sequence's own output) is LOADED in one trip and STORED in the next, behind that trip's loads -- no wait ever covers
      // a store, and both of a trip's waits are for loads issued a trip earlier.  Two register sets alternate (a copy would
      // have to wait for the loads).  Anything not simple, or whose source reaches into bytes still waiting to be stored,
      // flushes the pending stores first. ----
      typename Grp::SeqRegs R0, R1;
      bool have_p = false, p_in0 = false;  // a loaded-not-stored sequence exists; it sits in R0 / R1
      uint32_t p_op = 0, p_lit = 0, p_ml = 0;
      uint32_t t4 = g.ld32(src + ip);
      uint64_t o8 = 0;
      bool have_o8 = false;                // o8 already holds the offset word of the sequence at ip
      auto trip = [&](typename Grp::SeqRegs& cur, typename Grp::SeqRegs& pnd, const bool cur_is0) -> bool {  // false: leave the loop
        int lit = (int)((t4 >> 4) & 15u);
        int ml = (int)(t4 & 15u);
        int hdr = 1;
        if (lit == 15) {
          const uint32_t e = (t4 >> 8) & 255u;
          if (e == 255u) return false;
          lit += (int)e;
          hdr = 2;
        }
        if (!have_o8) o8 = g.ld64(src + ip + hdr + lit);
        const int off = (int)((uint32_t)o8 & 0xFFFFu);
        int adv = hdr + lit + 2;
        uint32_t nxt = (uint32_t)(o8 >> 16);
        if (ml == 15) {
          const uint32_t e = nxt & 255u;
          if (e == 255u) { have_o8 = false; return false; }
          ml += (int)e;
          adv++;
          nxt = (uint32_t)(o8 >> 24);
        }
        ml += 4;
        if (off > op + lit) { have_o8 = false; return false; }  // invalid offset: the exact path produces liblz4's error code
        const uint8_t* lit_src = src + ip + hdr;
        {
          // the next sequence's header is in nxt: if it can run in this loop, request its offset word now (the address the
          // next trip computes from t4 = nxt; ip + adv <= iend - 306 leaves >= 306 readable bytes, the word ends <= 279 in)
          have_o8 = false;
          int lit2 = (int)((nxt >> 4) & 15u), hdr2 = 1;
          bool ok2 = ip + adv <= iend - 306;
          if (lit2 == 15) {
            const uint32_t e2 = (nxt >> 8) & 255u;
            if (e2 == 255u) ok2 = false;
            lit2 += (int)e2;
            hdr2 = 2;
          }
          if (ok2) { o8 = g.ld64(src + ip + adv + hdr2 + lit2); have_o8 = true; }
        }
        // simple: one step each, and the match source [op+lit-off, +ml+slack) ends before this sequence's own output;
        // dep: the source reaches into the bytes the pending sequence has yet to store
        const uint32_t ulit = (uint32_t)lit, uml = (uint32_t)ml, uoff = (uint32_t)off, sl = g.slack();
        const bool simple = ulit <= g.step() && uml <= g.step() && uoff >= ulit + uml + sl;
        const bool dep = have_p && uoff < ulit + uml + sl + p_lit + p_ml;
        if (have_p && (!simple || dep)) { g.seq_store(pnd, dst + p_op, p_lit, p_ml); have_p = false; }
        if (simple) {
          g.seq_load(cur, lit_src, ulit, dst + op + lit - off, uml);
          if (have_p) g.seq_store(pnd, dst + p_op, p_lit, p_ml);
          have_p = true; p_in0 = cur_is0; p_op = (uint32_t)op; p_lit = ulit; p_ml = uml;
          op += lit;
        } else {
          g.copy_lits_wide(dst + op, lit_src, ulit);
          op += lit;
          g.copy_match_wide(dst, (uint32_t)op, uoff, uml);
        }
        op += ml;
        ip += adv;
        t4 = nxt;
        return ip <= iend - 306 && op <= oend - 606;
      };
      while (trip(R0, R1, true) && trip(R1, R0, false)) {}
      if (have_p) g.seq_store(p_in0 ? R0 : R1, dst + p_op, p_lit, p_ml);  // the exact code below reads what is in memory
    }
    if (STAGE && !PIPE && ip <= iend - 306 && op <= oend - 606) {
      // ---- the plain loop below, writing through the staging buffer.  Match sources are read from memory only: a source that
      // reaches past the flushed position (rare unless offsets are short) flushes everything first, and that match is then
      // written straight to memory.  Whole lines are flushed every second sequence -- on a schedule, not when a block happens
      // to have filled a line, so that the blocks sharing a wavefront flush in the same instructions. ----
      g.st_begin(stage, (uint32_t)op);
      uint32_t t4 = g.ld32(src + ip);
      uint32_t it = 0;
      do {
        int lit = (int)((t4 >> 4) & 15u);
        int ml = (int)(t4 & 15u);
        int hdr = 1;
        if (lit == 15) {
          const uint32_t e = (t4 >> 8) & 255u;
          if (e == 255u) break;
          lit += (int)e;
          hdr = 2;
        }
        const uint64_t o8 = g.ld64(src + ip + hdr + lit);
        g.st_lits(dst, (uint32_t)op, src + ip + hdr, (uint32_t)lit);
        const int off = (int)((uint32_t)o8 & 0xFFFFu);
        int adv = hdr + lit + 2;
        uint32_t nxt = (uint32_t)(o8 >> 16);
        if (ml == 15) {
          const uint32_t e = nxt & 255u;
          if (e == 255u) break;  // (the staged literals lie beyond op: never flushed)
          ml += (int)e;
          adv++;
          nxt = (uint32_t)(o8 >> 24);
        }
        ml += 4;
        if (off > op + lit) break;
        op += lit;
        if ((uint32_t)(op - off) + (uint32_t)ml + g.slack() > g.fl) {
          g.st_flush_all(dst, (uint32_t)op);
          g.copy_match_wide(dst, (uint32_t)op, (uint32_t)off, (uint32_t)ml);
          op += ml;
          g.fl = (uint32_t)op;
        } else {
          g.st_match(dst, (uint32_t)op, (uint32_t)off, (uint32_t)ml);
          op += ml;
        }
        ip += adv;
        t4 = nxt;
#ifndef LZ4HIP_STAGE_EVERY
#define LZ4HIP_STAGE_EVERY 2
#endif
        if (++it % LZ4HIP_STAGE_EVERY == 0u) g.st_flush_lines(dst, (uint32_t)op);
      } while (ip <= iend - 306 && op <= oend - 606);
      g.st_flush_all(dst, (uint32_t)op);  // the exact code below reads and writes memory
    }
    if (!STAGE && !PIPE && ip <= iend - 306 && op <= oend - 606) {
      uint32_t t4 = g.ld32(src + ip);  // {token, first literal-length byte, ...} of the sequence at ip
      do {
        int lit = (int)((t4 >> 4) & 15u);
        int ml = (int)(t4 & 15u);
        int hdr = 1;
        if (lit == 15) {
          const uint32_t e = (t4 >> 8) & 255u;
          if (e == 255u) break;
          lit += (int)e;
          hdr = 2;
        }
        const uint64_t o8 = g.ld64(src + ip + hdr + lit);  // {offset lo, hi, [match-length byte], next token, ...}
        g.copy_lits_wide(dst + op, src + ip + hdr, (uint32_t)lit);
        const int off = (int)((uint32_t)o8 & 0xFFFFu);
        int adv = hdr + lit + 2;
        uint32_t nxt = (uint32_t)(o8 >> 16);
        if (ml == 15) {
          const uint32_t e = nxt & 255u;
          if (e == 255u) break;  // (the literals just written are simply written again by the exact path)
          ml += (int)e;
          adv++;
          nxt = (uint32_t)(o8 >> 24);
        }
        ml += 4;
        if (off > op + lit) break;  // invalid offset: let the exact path produce liblz4's error code
        op += lit;
        g.copy_match_wide(dst, (uint32_t)op, (uint32_t)off, (uint32_t)ml);
        op += ml;
        ip += adv;
        t4 = nxt;
      } while (ip <= iend - 306 && op <= oend - 606);
    }
    // ---- tier 1: fast loop.  Software-pipelined: the word holding {offset, first match-length byte}
    // is requested BEFORE the literal copy, and the next sequence's token word BEFORE the match copy,
    // so a sequence costs two dependent memory round trips (literals+offset, then match source)
    // instead of four; the token fetch rides under the match-source latency. ----
    uint32_t w4;  // token + next 3 bytes when they are readable
    if (ip + 4 <= iend) { w4 = g.ld32(src + ip); } else { LZ4HIP_NEED_IN(1); w4 = g.ld8(src + ip); }
    for (;;) {
      token = w4 & 255u;
      ip++;
      length = (int)(token >> 4);
      bool wild;
      if (length == 15) {
        // read_variable_length(limit iend-15, initial+loop checks when SAFE); a limit hit inside
        // the loop is not an error in liblz4 1.9.3: the partial length is used
        if (SAFE && ip >= iend - 15) goto output_error;
        uint32_t s = (w4 >> 8) & 255u;  // readable: ip+4 <= iend held (SAFE: ip < iend-15; !SAFE: checked next)
        if (!SAFE && ip + 3 > iend) { LZ4HIP_NEED_IN(1); s = g.ld8(src + ip); }
        for (;;) {
          ip++;
          length += (int)s;
          if (SAFE && ip >= iend - 15) break;
          if (s != 255u) break;
          LZ4HIP_NEED_IN(1);
          s = g.ld8(src + ip);
          if (length > LZ4HIP_LEN_CAP) goto output_error;
        }
        if ((uint32_t)length > (uint32_t)(oend - op)) goto output_error;  // cannot fit: every tier rejects it at this ip
        cpy = op + length;
        if (SAFE) { if (cpy > oend - 32 || ip + length > iend - 32) goto safe_literal_copy; }
        else      { if (cpy > oend - 8) goto safe_literal_copy; }
        wild = SAFE || (ip + length + 4 <= iend);
      } else {
        cpy = op + length;
        if (SAFE && ip > iend - (16 + 1)) goto safe_literal_copy;
        wild = SAFE || (ip + length + 4 <= iend);
      }
      LZ4HIP_NEED_IN(length);
      {
        const int ipo = ip + length;  // where {offset lo, offset hi, first ML byte, ...} sit
        const bool have_off = SAFE || ipo + 2 <= iend;
        if (ipo + 4 <= iend) w4 = g.ld32(src + ipo); else if (have_off) w4 = g.ld16(src + ipo);
        g.copy_lits(dst + op, src + ip, (uint32_t)length, wild);
        ip = ipo;
        op = cpy;
        if (!have_off) goto output_error;  // the bounded fast decoder ran out of input at the offset
      }
      offset = (int)(w4 & 0xFFFFu);
      ip += 2;
      length = (int)(token & 15u);
      if (length == 15) {
        if (SAFE && offset > op) goto output_error;
        uint32_t s = (w4 >> 16) & 255u;
        if (ip + 2 > iend) { LZ4HIP_NEED_IN(1); s = g.ld8(src + ip); }  // (SAFE: error follows below anyway)
        for (;;) {  // read_variable_length(limit iend-4, loop check when SAFE)
          ip++;
          length += (int)s;
          if (SAFE && ip >= iend - 4) goto output_error;
          if (s != 255u) break;
          LZ4HIP_NEED_IN(1);
          s = g.ld8(src + ip);
          if (length > LZ4HIP_LEN_CAP) goto output_error;
        }
        if ((uint32_t)length > (uint32_t)(oend - op)) goto output_error;  // (offset <= op was checked / is checked first by liblz4 too)
        length += 4;
        if (op + length >= oend - 64) goto safe_match_copy;
      } else {
        length += 4;
        if (op + length >= oend - 64) goto safe_match_copy;
      }
      if (offset > op) goto output_error;  // liblz4 checks this only when SAFE; the HIP engine never reads before dst
      // next token word (SAFE: ip < iend holds after a non-final sequence)
      if (ip + 4 <= iend) { w4 = g.ld32(src + ip); } else { LZ4HIP_NEED_IN(1); w4 = g.ld8(src + ip); }
      g.copy_match(dst, (uint32_t)op, (uint32_t)offset, (uint32_t)length, true);
      op += length;
      // The interior loop left on a sequence it does not handle (a length run of two or more bytes, ...); that sequence has now
      // been decoded with every check: back to the interior loop while the block is still far from both ends.  (Round 1 entered
      // it once per block: data with an occasional long match or literal run fell off the fast path for the rest of the block.)
      if (LZ4HIP_DECODE_REENTER && ip <= iend - 306 && op <= oend - 606) goto interior;
    }
  }

  for (;;) {  // ---- tiers 2+3: shortcut and fully checked path ----
    LZ4HIP_NEED_IN(1);
    token = g.ld8(src + ip);
    ip++;
    length = (int)(token >> 4);
    if ((SAFE ? length != 15 : length <= 8) && (SAFE ? ip < shortiend : true) && op <= shortoend) {
      LZ4HIP_NEED_IN(length + 2);
      g.copy_lits(dst + op, src + ip, (uint32_t)length, false);
      op += length;
      ip += length;
      length = (int)(token & 15u);
      offset = (int)g.ld16(src + ip);
      ip += 2;
      if (length != 15 && offset >= 8 && offset <= op) {
        g.copy_match(dst, (uint32_t)op, (uint32_t)offset, (uint32_t)(length + 4), false);
        op += length + 4;
        continue;
      }
      if (!SAFE && length != 15 && offset >= 8 && offset > op) goto output_error;  // liblz4 would read before dst
      goto copy_match_label;
    }
    if (length == 15) {
      if (SAFE && ip >= iend - 15) goto output_error;
      for (;;) {
        LZ4HIP_NEED_IN(1);
        const uint32_t s = g.ld8(src + ip);
        ip++;
        length += (int)s;
        if (length > LZ4HIP_LEN_CAP) goto output_error;
        if (SAFE && ip >= iend - 15) break;
        if (s != 255u) break;
      }
      if ((uint32_t)length > (uint32_t)(oend - op)) goto output_error;
    }
    cpy = op + length;
  safe_literal_copy:
    if ((SAFE && (cpy > oend - 12 || ip + length > iend - (2 + 1 + 5))) || (!SAFE && cpy > oend - 8)) {
      if (!SAFE && cpy != oend) goto output_error;
      if (SAFE && (ip + length != iend || cpy > oend)) goto output_error;
      LZ4HIP_NEED_IN(length);
      g.copy_lits(dst + op, src + ip, (uint32_t)length, false);
      ip += length;
      op += length;
      break;
    }
    LZ4HIP_NEED_IN(length);
    g.copy_lits(dst + op, src + ip, (uint32_t)length, false);
    ip += length;
    op = cpy;
    LZ4HIP_NEED_IN(2);
    offset = (int)g.ld16(src + ip);
    ip += 2;
    length = (int)(token & 15u);
  copy_match_label:
    if (length == 15) {
      for (;;) {
        LZ4HIP_NEED_IN(1);
        const uint32_t s = g.ld8(src + ip);
        ip++;
        length += (int)s;
        if (SAFE && ip >= iend - 4) goto output_error;
        if (length > LZ4HIP_LEN_CAP) goto output_error;
        if (s != 255u) break;
      }
      if (offset <= op && (uint32_t)length > (uint32_t)(oend - op)) goto output_error;
    }
    length += 4;
  safe_match_copy:
    if (offset > op) goto output_error;
    cpy = op + length;
    if (cpy > oend - 5) goto output_error;  // the last 5 bytes are always literals
    g.copy_match(dst, (uint32_t)op, (uint32_t)offset, (uint32_t)length, false);
    op = cpy;
  }
  return SAFE ? op : ip;
output_error:
  return -ip - 1;
#undef LZ4HIP_NEED_IN
#undef LZ4HIP_LEN_CAP
}

}  // namespace lz4hip

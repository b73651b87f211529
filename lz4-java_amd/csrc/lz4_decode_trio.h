// lz4_decode_trio.h -- the TRIO loop of the block decoder (decode_block PIPE == 8): THREE WAVEFRONTS PER BLOCK -- SCANNER, PLANNER, COPIER.
//
// Same sequences, same bytes as the other interior loops of decode_block (LZ4_decompress_safe / _fast of liblz4 1.9.3,
// /root/reference/src/jni/net_jpountz_lz4_LZ4JNI.c:216 / :169; the single-call path this is for: one block per call,
// /root/reference/src/java/net/jpountz/lz4/LZ4JNISafeDecompressor.java:34-43).  The pair loop (lz4_decode_pair.h) cuts the parallel
// wave trip between everything that is a function of the compressed stream alone and the copies: ~610 against ~220 wave
// instructions, so its pipeline runs at 830 / 610 = 1.35x (measured: profiles/r06_pair_notes.txt).  The parser's share has a seam of
// its own: WHERE the sequences start -- discovery + walk -- is a property of the stream that does not depend on how the trips are
// cut (a chain of tokens is a chain of tokens), while everything behind it (records, prefix sums, ring-held rules, rounds, what to do
// with a sequence the trips cannot take) depends on where the output stands.  So:
//   SCANNER  owns the stream ring (refills from memory), reads window after window and posts, per window, the start positions of its
//            sequences (the walk's result: 64 lanes x one position), their number, the window's stream position and where the next
//            window begins (the end of the window's last sequence: the discovery's own formula for that start, unclamped);
//   PLANNER  takes the windows in order and does what the pair loop's parser does behind its walk -- per pass: the records of up to 31
//            sequences (headers gathered from the stream ring), output positions, held / far / simple masks, the dependency rounds --
//            posting the pair loop's messages (PASS / SINGLE / EXIT).  A trip that is cut (a dependency with an empty round, a far
//            source not flushed yet, 2 KB of output) continues from the NEXT START OF THE SAME WINDOW: nothing is scanned twice;
//   COPIER   is the pair loop's copier (it runs decode_block): rounds of exact lane-per-run copies, one-sequence steps, the flusher.
// Two queues in LDS (scanner -> planner: 288-byte entries; planner -> copier: the pair loop's 512-byte mailbox), ordered by the LDS
// itself as in the pair loop.  The scanner runs on whatever the stream says: behind a sequence that is not simple (a length run of two
// or more bytes) its positions are garbage -- the planner finds that sequence not simple and ends the entry there (EXIT), garbage is
// never used.  Leaving: the planner posts EXIT and raises STOP; the scanner sees STOP at its next wait or window and acknowledges
// (SACK); the copier flushes, waits for the acknowledgement -- nobody touches the rings or the queues any more -- and goes back to
// decode_block's exact code, which does one sequence and starts all three again.
// tests/hostsim runs the three as host threads over one block of "LDS" (tests/test_hostsim.py: the wave tests with par == "trio").
#pragma once
#include <stdint.h>

namespace lz4hip {

constexpr uint32_t TRIO_SCAN_SLOTS = 3u;       // windows in flight between scanner and planner
constexpr uint32_t TRIO_SCAN_BYTES = 288u;     // 64 positions (a dword each) + {count, window position, next window, flags} + pad
enum : uint32_t { PC_SHEAD = 11u, PC_STAIL = 12u, PC_SACK = 13u, PC_STOP = 14u, PC_IPDONE = 15u };   // (behind the pair loop's control words; STOP and IPDONE are one aligned pair: the scanner reads both at the top of every window)
constexpr uint32_t TRIO_END = 1u;              // flags: the scanner's last entry -- the loop ends at its window position

// ---- SCANNER ----
template <class Grp>
LZ4HIP_DEV void trio_scan_loop(Grp& g, const uint8_t* src, const uint32_t iend, uint32_t ip, const uint32_t epoch) {
  typedef typename Grp::LChunk LChunk;
  typedef typename Grp::VU VU;
  typedef typename Grp::VB VB;
  constexpr uint32_t STEP = 256u, AHEAD = 512u;
  const uint32_t KS = g.wv_stream();
  uint32_t shead = 0u, stail_seen = 0u;
  uint32_t avail = ip & ~(STEP - 1u);
  {
    const LChunk a0 = g.rs_fetch(src, avail), a1 = g.rs_fetch(src, avail + STEP), a2 = g.rs_fetch(src, avail + 2u * STEP), a3 = g.rs_fetch(src, avail + 3u * STEP);
    g.rs_put(avail, a0); g.rs_put(avail + STEP, a1); g.rs_put(avail + 2u * STEP, a2); g.rs_put(avail + 3u * STEP, a3);
    avail += 4u * STEP;
  }
  LChunk rf0 = LChunk();
  const VU lane = g.vlane();
  const VU p0 = lane * 4u;
  bool stopped = false;
  uint32_t Tprev = 0u;
  auto post = [&](const VU& posv, uint32_t T, uint32_t wip, uint32_t nextw, uint32_t flags) -> bool {
    while (shead - stail_seen >= TRIO_SCAN_SLOTS) {        // the planner is behind
      if (g.pm_peek(PC_STOP) == epoch) return false;
      stail_seen = g.pm_peek(PC_STAIL);
      if (shead - stail_seen >= TRIO_SCAN_SLOTS) g.pm_nap(0u);
    }
    g.sq_put(shead % TRIO_SCAN_SLOTS, posv, T, wip, nextw, flags);
    shead++;
    g.pm_post(PC_SHEAD, shead);
    return true;
  };
  for (;;) {
    uint32_t stop, ipdone;
    g.pm_peek2(PC_STOP, stop, ipdone);
    if (stop == epoch) { stopped = true; break; }
    if (ip + 32u > iend) break;                 // (the planner's end rules stop in front of the stream's last 32 bytes: below)
    // the stream ring keeps everything from the copier's oldest unfinished message on (it reads literals there; the planner is ahead of
    // it).  IPDONE = where the sequences of the copier's NEXT message begin: with nothing in flight that is this window, and the
    // condition below is the one-wavefront loop's (a 1 KB stream ring cannot hold the previous window's start AND this window's 512 bytes)
    uint32_t keep = ipdone & ~(STEP - 1u);
    if (LZ4HIP_UNLIKELY(ip + AHEAD > avail)) {
      for (;;) {
        while ((ip + AHEAD > avail) & (avail + STEP <= iend) & (avail + STEP <= keep + KS)) {
          g.rs_put(avail, g.rs_fetch(src, avail));
          avail += STEP;
        }
        // the stream's last, partial step: its bytes below iend (zeros behind them).  With it the WHOLE stream is in the ring and the windows
        // go on up to ilim -- a sequence that starts there ends inside the stream; what a window reads behind iend is nothing the planner uses
        if ((ip + AHEAD > avail) & (avail < iend) & (avail + STEP > iend) & (avail + STEP <= keep + KS)) {
          g.rs_put(avail, g.rs_fetch_upto(src, avail, iend));
          avail += STEP;
        }
        if (!((ip + AHEAD > avail) & (avail < iend))) break;             // enough, or the whole stream
        if (g.pm_peek(PC_STOP) == epoch) { stopped = true; break; }
        g.pm_nap(1u);                                                     // the copier still reads where the next step would go
        keep = g.pm_peek(PC_IPDONE) & ~(STEP - 1u);
      }
      if (stopped | ((ip + AHEAD > avail) & (avail < iend))) break;
    }
    uint32_t nf = 0u;
    if ((avail + STEP <= iend) & (avail + STEP <= keep + KS)) {
      rf0 = g.rs_fetch(src, avail);
      nf = 1u;
    }
    // discovery and walk (lz4_decode_wave.h steps 1 + 2)
    VU posv = VU(0u), nxfull[4];
    uint32_t T = 0u;
    VU blo, bhi;
    g.vs_win(ip, blo, bhi);
    {
      VU nxpack = VU(0u);
#pragma unroll
      for (uint32_t j = 0; j < 4u; j++) {
        const VU w = j == 0u ? blo : ((blo >> (8 * (int)j)) | (bhi << (32 - 8 * (int)j)));
        const VU tl = (w >> 4) & 15u, e1 = (w >> 8) & 255u;
        const VB l15 = tl == 15u;
        const VU nxt = p0 + (j + 3u) + Grp::vsel(l15, VU(1u), VU(0u)) + tl + Grp::vsel(l15, e1, VU(0u)) + Grp::vsel((w & 15u) == 15u, VU(1u), VU(0u));
        nxfull[j] = nxt;
        nxpack = nxpack | (Grp::vsel(nxt <= 250u, nxt, VU(255u)) << (8 * (int)j));
      }
      if (Tprev >= LZ4HIP_WALK_PAR_MIN) Grp::vwalk_par(nxpack, lane, posv, T); else Grp::vwalk(nxpack, posv, T);   // (lz4_decode_wave.h: pointer doubling for windows full of sequences)
      Tprev = T;
    }
    // where the next window begins: behind the last sequence that starts in this one (the same formula, not clamped to the window)
    const uint32_t lp = Grp::vreadlane(posv, T - 1u);
    const uint32_t n0 = Grp::vreadlane(nxfull[0], lp >> 2), n1 = Grp::vreadlane(nxfull[1], lp >> 2), n2 = Grp::vreadlane(nxfull[2], lp >> 2), n3 = Grp::vreadlane(nxfull[3], lp >> 2);
    const uint32_t q = lp & 3u;
    const uint32_t used = q == 0u ? n0 : q == 1u ? n1 : q == 2u ? n2 : n3;
    if (!post(posv, T, ip, ip + used, 0u)) { stopped = true; break; }
    if (nf != 0u) {
      g.rs_put(avail, rf0);
      avail += STEP;
    }
    ip += used;
  }
  if (!stopped) (void)post(VU(0u), 0u, ip, ip, TRIO_END);
  g.pm_post(PC_SACK, epoch);
}

// ---- PLANNER: the pair loop's parser behind its walk (lz4_decode_pair.h pair_parse_loop), fed with the scanner's windows ----
template <class Grp>
LZ4HIP_DEV void trio_plan_loop(Grp& g, const uint32_t iend, const uint32_t oend, const uint32_t db, uint32_t op, const uint32_t epoch) {
  typedef typename Grp::VU VU;
  typedef typename Grp::VB VB;
  constexpr uint32_t STEP = 256u, TRIPMAX = 2048u;
  const uint32_t KW = g.wv_ring();
  const uint32_t ilim = iend - 306u, olim = oend - 606u;
  const uint32_t op0 = op;
  uint32_t head = 0u, tail_seen = 0u, stail = 0u;
  auto post = [&](const VU& w0, const VU& w1) {
    while (head - tail_seen >= PAIR_SLOTS) {
      tail_seen = g.pm_peek(PC_TAIL);
      if (head - tail_seen >= PAIR_SLOTS) g.pm_nap(3u);
    }
    g.pm_put(head % PAIR_SLOTS, w0, w1);
    head++;
    g.pm_post(PC_HEAD, head);
  };
  uint32_t fl = (op + db) & ~(STEP - 1u);
  const VU lane = g.vlane();
  const VB isM = (lane & 1u) != 0u;
  uint32_t wild = op;
  uint32_t exit_ip = 0u;
  for (;;) {
    VU posv;
    uint32_t T, wip, nextw, flags;
    while (g.sq_peek_get(PC_SHEAD, stail % TRIO_SCAN_SLOTS, posv, T, wip, nextw, flags) == stail) g.pm_nap(2u);   // (the word and the entry behind it in one round trip)
    if (flags & TRIO_END) { exit_ip = wip; break; }
    uint32_t tk = 0u;
    bool leave = false;
    // segments: each is what the one-wavefront loop calls a trip -- from start tk of this window on, until a pass is cut or the window is done
    while (tk < T) {
      const uint32_t ip = wip + Grp::vreadlane(posv, tk);     // the stream position of the segment's first sequence
      uint32_t tk0 = tk, opc = op;
      for (;;) {
        const uint32_t np = T - tk < 31u ? T - tk : 31u;
        const VU sq = (lane >> 1) + tk;
        const VU pv = Grp::vshfl(posv, sq);                    // where the lane's sequence starts, relative to the window
        const uint64_t actm = (1ull << (2u * np)) - 1ull;
        const VB act = Grp::vlanes(actm);
        const VU hw = g.vs_ld32(pv + wip);                     // its first bytes, from the stream ring
        const VU tl = (hw >> 4) & 15u, tm = hw & 15u, e1 = (hw >> 8) & 255u;
        const VB l15 = tl == 15u;
        const VU lit = tl + Grp::vsel(l15, e1, VU(0u));
        const VU lp = pv + wip + Grp::vsel(l15, VU(2u), VU(1u));
        const VU ow = g.vs_ld32(lp + lit);
        const VU off = ow & 0xFFFFu, e2 = (ow >> 16) & 255u;
        const VB m15 = tm == 15u;
        const VU mlx = tm + Grp::vsel(m15, e2, VU(0u)), ml = mlx + 4u;
        const uint64_t simplem = Grp::vballot(off != 0u) & Grp::vballot(lit < 255u) & Grp::vballot(mlx < 255u);
        const VU len = Grp::vsel(isM, ml, lit);
        const VU tot = Grp::vsel(act, len, VU(0u));
        const VU ex = Grp::vexcl_scan(tot);
        const VU o = ex + opc;
        const VU mp = o - off;
        const VU oe = o + tot;
        const VU send = mp + ml;
        const uint32_t oe_all = Grp::vreadlane(oe, 2u * np - 1u);
        // (lz4_decode_wave.h LZ4HIP_WAVE_OVL: a match that overlaps its own output does not cut the pass -- the copier does it wave-wide where it stands; its
        // pieces write up to a step beyond its end, which a pass that holds one counts into what it touches)
        const uint64_t ovlm = Grp::vballot(off < ml);
        const uint32_t tb = (oe_all - op < TRIPMAX ? oe_all : op + TRIPMAX) + 16u + ((LZ4HIP_WAVE_OVL && (ovlm & actm) != 0ull) ? STEP : 0u);
        const uint32_t bound = (int32_t)(wild - tb) > 0 ? wild : tb;
        const uint32_t memlim = fl > op0 + db ? fl - db : op0;
        constexpr uint64_t litm = 0x5555555555555555ull;
        const uint64_t heldm = Grp::vballot(mp >= VU(op0)) & Grp::vballot((mp + KW) >= VU(bound));
        const uint64_t srcm = Grp::vballot(mp < VU(0x80000000u)) & (heldm | Grp::vballot(send <= VU(memlim)));
        // THE BLOCK'S END, sequence by sequence.  The other interior loops stop 306 stream / 606 output bytes in front of the block's ends,
        // where every check of liblz4's fast loop provably passes for ANY simple sequence, and leave the rest -- ~17 sequences of an
        // App. F block, a memory round trip or two each in decode_block's exact code: ~20 us of a 150-us single-block call -- to it.
        // Here the lengths are known per lane, so the rule is liblz4's own, per sequence (lz4.c LZ4_decompress_generic, fast loop; the
        // exact code restates it in decode_block's tier 1): literals that end 32 bytes in front of the stream's end (covers `ip + length >
        // iend - 32` and, for short runs, `ip > iend - 17`), a match that ends more than 64 bytes in front of the output's end (`op +
        // length >= oend - FASTLOOP_SAFE_DISTANCE`; it implies `cpy > oend - 32` for the literals).  A sequence that passes is one
        // liblz4 decodes in its fast loop without leaving it, with these bytes; the first that does not is where the loop ends (a lane
        // pair: the rounds take whole sequences) and the exact code takes over with liblz4's own branch.  Both decoders: the fast
        // decoder's conditions (`cpy > oend - 8`, the bounded reads) are weaker.
        const uint64_t okbm = actm & simplem & (litm | srcm) & Grp::vballot((lp + lit + 32u) <= VU(iend)) & Grp::vballot((oe + 64u) < VU(oend)) & Grp::vballot((oe - op) <= VU(TRIPMAX));
        const uint64_t farm = ~litm & ~heldm;
        const VU spv = Grp::vsel(isM, mp + db, lp);
        const uint64_t oddm = g.vodd_mask(o + db, len);
        uint32_t a = 0u;
        uint64_t rsm = 0ull, ovm = 0ull;                       // lanes that start a round; rounds that are one overlapping match
        for (;;) {
          const uint32_t oa = Grp::vreadlane(o, a);
          const uint64_t below = (1ull << a) - 1ull;
          const uint64_t okm = (okbm & (litm | Grp::vballot(send <= VU(oa)))) | below;
          const uint32_t Te = (uint32_t)__builtin_ctzll(~okm | (1ull << 63));
          if (Te == a) {
            if (LZ4HIP_WAVE_OVL && (a & 1u) != 0u && (((okbm & heldm & ovlm) >> a) & 1ull) != 0ull) {
              const uint32_t ml_a = Grp::vreadlane(ml, a);
              if ((int32_t)(bound - (oa + ml_a + STEP)) >= 0) {
                rsm |= 1ull << a; ovm |= 1ull << a;
                if ((int32_t)(oa + ml_a + STEP - wild) > 0) wild = oa + ml_a + STEP;
                a += 1u;
                if (a >= 2u * np) break;
                continue;
              }
            }
            break;
          }
          rsm |= 1ull << a;
          a = Te;
          if (a >= 2u * np) break;
        }
        a &= ~1u;
        if (a == 0u) break;
        tk += a >> 1;
        opc = Grp::vreadlane(oe, a - 1u);
        {
          VU w0 = ((o + db) & 0xFFFFu) | (len << 16) | Grp::vsel(Grp::vlanes(farm), VU(PAIR_F_FAR), VU(0u)) | Grp::vsel(Grp::vlanes(oddm), VU(PAIR_F_ODD), VU(0u)) |
                  Grp::vsel(Grp::vlanes(rsm), VU(PAIR_F_ROUND), VU(0u)) | Grp::vsel(Grp::vlanes(ovm), VU(PAIR_F_OVL), VU(0u));
          VU w1 = spv;
          w0 = Grp::vwritelane(w0, PAIR_PASS | (a << 8), 62u);
          w1 = Grp::vwritelane(w1, opc, 62u);
          w1 = Grp::vwritelane(w1, tk < T ? wip + Grp::vreadlane(posv, tk) : nextw, 63u);   // where the NEXT message's sequences begin: the copier's IPDONE once this one is done
          post(w0, w1);
        }
        if ((a < 2u * np) | (tk >= T)) break;
      }
      if (LZ4HIP_UNLIKELY(tk == tk0)) {           // the segment's first sequence is not for a pass: the one-sequence step, or out of the loop
        uint32_t ip1 = ip, op1 = op;
        if (!wave_single_dry(g, ip1, op1, op0, fl, db, ilim, olim)) { exit_ip = ip; leave = true; break; }
        VU w0 = VU(0u), w1 = VU(0u);
        w0 = Grp::vwritelane(w0, PAIR_SINGLE, 62u);
        w1 = Grp::vwritelane(w1, op, 62u);
        w0 = Grp::vwritelane(w0, ip, 63u);
        w1 = Grp::vwritelane(w1, ip1, 63u);
        post(w0, w1);
        op = op1;
        wild = op + STEP;
        tk++;
      } else {
        op = opc;
      }
      fl = (op + db) & ~(STEP - 1u);
    }
    if (leave) break;
    stail++;
    g.pm_post(PC_STAIL, stail);
  }
  VU w0 = VU(0u), w1 = VU(0u);
  w0 = Grp::vwritelane(w0, PAIR_EXIT, 62u);
  w1 = Grp::vwritelane(w1, op, 62u);
  w0 = Grp::vwritelane(w0, exit_ip, 63u);
  w1 = Grp::vwritelane(w1, exit_ip, 63u);
  post(w0, w1);
  g.pm_post(PC_STOP, epoch);
}

// the helper wavefronts' lives: wait for an entry of the block's copier (CMD = the entry's epoch), work until the loop ends, wait again
template <class Grp>
LZ4HIP_DEV void trio_service(Grp& g, uint8_t* lds, const bool scanner) {
  g.pm_begin(lds);
  uint32_t cmd_seen = 0u;
  for (;;) {
    uint32_t c;
    while ((c = g.pm_peek(PC_CMD)) == cmd_seen) g.pm_idle();
    cmd_seen = c;
    if (g.pm_peek(PC_QUIT) != 0u) return;
    const uint32_t ip = g.pm_peek(PC_IP), op = g.pm_peek(PC_OP), iend = g.pm_peek(PC_IEND), oend = g.pm_peek(PC_OEND), db = g.pm_peek(PC_DB);
    g.wv_begin_db(lds, db);
    if (scanner) trio_scan_loop(g, g.pm_ptr(g.pm_peek(PC_SRC_LO), g.pm_peek(PC_SRC_HI)), iend, ip, c);
    else trio_plan_loop(g, iend, oend, db, op, c);
  }
}

// ---- COPIER (decode_block PIPE == 8; the contract of decode_wave_par_loop) ----
template <class Grp>
LZ4HIP_DEV void decode_trio_loop(Grp& g, const uint8_t* src, const int iend, uint8_t* dst, const int oend, int& ip_io, int& op_io, uint8_t* lds) {
  typedef typename Grp::VU VU;
  typedef typename Grp::VB VB;
  constexpr uint32_t STEP = 256u;
  uint32_t ip = (uint32_t)ip_io, op = (uint32_t)op_io;
  const uint32_t ilim = (uint32_t)iend - 306u, olim = (uint32_t)oend - 606u;
  g.wv_begin(lds, dst);
  g.pm_begin(lds);
  const uint32_t db = g.wv_dbase();
  const uint32_t op0 = op;
  uint32_t fl = (op + db) & ~(STEP - 1u);
  // ---- start scanner and planner on this entry (both idle: the last entry's EXIT was taken and its SACK seen) ----
  const uint32_t epoch = g.pm_peek(PC_CMD) + 1u;
  g.pm_post(PC_HEAD, 0u); g.pm_post(PC_TAIL, 0u); g.pm_post(PC_SHEAD, 0u); g.pm_post(PC_STAIL, 0u); g.pm_post(PC_IPDONE, ip);
  g.pm_post(PC_IP, ip); g.pm_post(PC_OP, op); g.pm_post(PC_IEND, (uint32_t)iend); g.pm_post(PC_OEND, (uint32_t)oend); g.pm_post(PC_DB, db);
  g.pm_post(PC_SRC_LO, (uint32_t)(uintptr_t)src); g.pm_post(PC_SRC_HI, (uint32_t)((uint64_t)(uintptr_t)src >> 32));
  g.pm_post(PC_CMD, epoch);
  typedef typename Grp::LChunk LChunk;
  const VB isM = (g.vlane() & 1u) != 0u;
  uint32_t tail = 0u;
  for (;;) {
    // flusher: the whole aligned steps below op -- the messages before this one put them into the ring -- are REQUESTED here, in front of
    // the wait for the next message, and stored once that has arrived: their LDS round trips hide behind it (they stood at the end of
    // every message, one after the other: the copier is the trio's slowest stage on big blocks, tools/trio_stats.py).  Memory holds
    // everything below (op + db) & ~255 before the message's far loads, as the planner assumed
    const uint32_t pend = (op + db - fl) >> 8;
    LChunk fx0 = LChunk(), fx1 = LChunk(), fx2 = LChunk();
    if (pend >= 1u) fx0 = g.wv_read_al(fl);
    if (pend >= 2u) fx1 = g.wv_read_al(fl + STEP);
    if (pend >= 3u) fx2 = g.wv_read_al(fl + 2u * STEP);
    VU w0, w1;
    while (g.pm_peek_get(PC_HEAD, tail % PAIR_SLOTS, w0, w1) == tail) g.pm_nap(4u);   // (the word and the slot behind it in one round trip)
    if (pend >= 1u) { g.wv_store(dst, fl, fx0, op0 + db, 0xFFFFFFFFu); fl += STEP; }
    if (pend >= 2u) { g.wv_store(dst, fl, fx1, op0 + db, 0xFFFFFFFFu); fl += STEP; }
    if (pend >= 3u) { g.wv_store(dst, fl, fx2, op0 + db, 0xFFFFFFFFu); fl += STEP; }
    while (LZ4HIP_UNLIKELY(op + db - fl >= STEP)) { g.wv_store(dst, fl, g.wv_read_al(fl), op0 + db, 0xFFFFFFFFu); fl += STEP; }
    const uint32_t hdr = Grp::vreadlane(w0, 62u), opn = Grp::vreadlane(w1, 62u), mip = Grp::vreadlane(w0, 63u), ipn = Grp::vreadlane(w1, 63u);
    const uint32_t kind = hdr & 255u;
    if (kind == PAIR_PASS) {
      const uint32_t a_end = hdr >> 8;
      const VU dw = w0 & 0xFFFFu, len = (w0 >> 16) & 0x1FFu;
      const uint64_t farm = Grp::vballot((w0 & PAIR_F_FAR) != 0u), oddm = Grp::vballot((w0 & PAIR_F_ODD) != 0u), rsm = Grp::vballot((w0 & PAIR_F_ROUND) != 0u);
      const VU mp = w1 - db;
      const uint64_t ovm = Grp::vballot((w0 & PAIR_F_OVL) != 0u);
      uint32_t a = 0u;
      do {
        const uint64_t later = rsm & ~((2ull << a) - 1ull);
        uint32_t Te = later != 0ull ? (uint32_t)__builtin_ctzll(later) : a_end;
        Te = Te < a_end ? Te : a_end;
        if (LZ4HIP_UNLIKELY(((ovm >> a) & 1ull) != 0ull)) {   // a round that is one match overlapping its own output: wave-wide, by doubling (ring coordinates mod 2^16: the rings divide that)
          const uint32_t dwa = Grp::vreadlane(dw, a);
          wave_match_ring(g, dwa, (dwa - Grp::vreadlane(w1, a)) & 0xFFFFu, Grp::vreadlane(len, a));
        } else
        g.vcopy_run(dw, !isM, w1, len, ((1ull << Te) - 1ull) & ~((1ull << a) - 1ull), dst, mp, farm, oddm);
        a = Te;
      } while (a < a_end);
      op = opn;
    } else if (kind == PAIR_SINGLE) {
      ip = mip; op = opn;
      (void)wave_single_step(g, dst, ip, op, op0, fl, db, ilim, olim);
    } else {
      ip = mip; op = opn;
    }
    tail++;
    g.pm_post(PC_IPDONE, ipn);                  // behind the message's last read of the stream ring: nothing in front of the next message's sequences is needed any more
    g.pm_post(PC_TAIL, tail);
    if (kind == PAIR_EXIT) break;
  }
  while (op + db - fl >= STEP) { g.wv_store(dst, fl, g.wv_read_al(fl), op0 + db, 0xFFFFFFFFu); fl += STEP; }
  if (op + db != fl) g.wv_store(dst, fl, g.wv_read_al(fl), op0 + db, op + db);
  while (g.pm_peek(PC_SACK) != epoch) g.pm_nap(5u);   // the scanner has seen STOP (or its own end): nobody writes the rings or the queues any more
  ip_io = (int)ip; op_io = (int)op;
}
template <class Grp>
LZ4HIP_DEV void trio_quit(Grp& g, uint8_t* lds) {
  g.pm_begin(lds);
  g.pm_post(PC_QUIT, 1u);
  g.pm_post(PC_CMD, g.pm_peek(PC_CMD) + 1u);
}

}  // namespace lz4hip

// lz4_decode_pair.h -- the PAIR loop of the block decoder (decode_block PIPE == 7): TWO WAVEFRONTS PER BLOCK, a PARSER and a COPIER.
//
// Same sequences, same bytes as the other interior loops of decode_block (LZ4_decompress_safe / _fast of liblz4 1.9.3,
// /root/reference/src/jni/net_jpountz_lz4_LZ4JNI.c:216 / :169).  Block independence is the only parallelism the reference guarantees
// (/root/reference/src/java/net/jpountz/lz4/LZ4FrameOutputStream.java:361-363), and a launch that cannot fill the GPU with blocks --
// the 8-GPU shard of BASELINE configs[2] is 2048 x 4 MiB per GPU, the readers' batches, the Java single-call path is ONE block -- is
// bound by how fast one block decodes.  The parallel wave loop (lz4_decode_wave.h, PIPE 5) gives a block one wavefront and decodes
// every sequence of a 256-byte window of the stream per trip; measured (profiles/r05_wave_notes.txt section 4): ~830 wave instructions
// per trip of 12.4 sequences, the wavefront issuing 52 % of its cycles and waiting for LDS round trips 37 % -- one instruction per ~9
// cycles, with most of the CU's wave slots empty at 8 blocks per CU.  A trip is two halves that need nothing from each other:
//   * WHAT to copy is a function of the compressed stream alone: discovery, walk, the sequences' records, the prefix sum of the run
//     lengths (= every run's output position), which match sources the ring still holds, the dependency rounds -- even where the
//     flusher stands is a function of the output position.  None of it reads a decoded byte.
//   * the COPIES (stream ring -> output ring, output ring -> output ring, far sources from flushed memory) and the flusher need the
//     output ring and nothing of the parse but its result.
// So the block gets a second wavefront.  The PARSER runs the first half one or two trips ahead and posts, per pass of a trip, ONE
// message of 64 x 8 bytes through a three-slot mailbox in LDS: lane k = run k of the pass {ring index of its output, length, source,
// flags: far source / slow copy / first run of a dependency round}, lanes 62 / 63 = {kind, lanes taken, output position behind
// the pass, stream position}.  The COPIER takes the messages in order, runs the rounds (group_dev.h vcopy_run: the same exact
// lane-per-run copies as PIPE 5) and the flusher.  What a trip cannot take as its first sequence is a SINGLE message (the copier
// runs wave_single_step, the parser only its checks); what the loop cannot take at all is an EXIT message: the copier -- which is
// the wavefront that runs decode_block -- flushes, does that one sequence with the exact code and starts the parser again.
//
// Protocol (single producer, single consumer; ctl words in LDS): HEAD = messages posted (parser), TAIL = messages done (copier: a
// slot and the stream bytes its literals come from are free once the message is DONE); CMD = entries started (copier), with the
// entry's parameters in the words behind it.  Ordering rests on the LDS executing a wavefront's instructions in order and every
// instruction as a whole: slot data are written in front of HEAD, read behind it; TAIL is written behind the copier's last read.
// The compiler is held to that order by barriers in the backend (group_dev.h pm_*); no wait is involved -- a release at workgroup
// scope would wait for the parser's stream refill (a load from memory) at every post.  The stream ring belongs to the parser, who
// keeps everything from the OLDEST UNFINISHED message's stream position on (the copier reads literals there); the output ring
// belongs to the copier.  tests/hostsim runs both halves as two host threads per block on shared "LDS" with acquire / release
// atomics and random naps (tests/test_hostsim.py::test_pair_decoder_loop*).
#pragma once
#include <stdint.h>

namespace lz4hip {

constexpr uint32_t PAIR_SLOTS = 3u;            // messages in flight between the two wavefronts
constexpr uint32_t PAIR_SLOT_BYTES = 512u;     // 64 lanes x {w0, w1}
constexpr uint32_t PAIR_CTL_WORDS = 16u;
constexpr uint32_t PAIR_MAIL_BYTES = PAIR_SLOTS * PAIR_SLOT_BYTES + PAIR_CTL_WORDS * 4u;
enum : uint32_t { PAIR_PASS = 1u, PAIR_SINGLE = 2u, PAIR_EXIT = 3u };
enum : uint32_t { PC_HEAD = 0u, PC_TAIL = 1u, PC_CMD = 2u, PC_QUIT = 3u, PC_IP = 4u, PC_OP = 5u, PC_IEND = 6u, PC_OEND = 7u, PC_SRC_LO = 8u, PC_SRC_HI = 9u, PC_DB = 10u };
// w0 of a run: ring index of its output (16 bits: rings of up to 64 KB) | length << 16 (9 bits) | flags
constexpr uint32_t PAIR_F_FAR = 1u << 25, PAIR_F_ODD = 1u << 26, PAIR_F_ROUND = 1u << 27, PAIR_F_OVL = 1u << 28;   // (OVL: the trio planner only -- a round that is one match overlapping its own output)

// the checks of wave_single_step (lz4_decode_wave.h) without its copies: is the sequence at ip one for the one-sequence step, and
// where does it end.  Same reads of the stream ring, same rules: the copier's wave_single_step on the same {ip, op, fl} agrees.
template <class Grp>
LZ4HIP_DEV bool wave_single_dry(Grp& g, uint32_t& ip, uint32_t& op, const uint32_t op0, const uint32_t fl, const uint32_t db, const uint32_t ilim, const uint32_t olim) {
  constexpr uint32_t STEP = 256u;
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
  ip += hdr + lit + (m15 ? 3u : 2u);
  op = op2;
  return true;
}

// ---- the PARSER's loop: the trip of decode_wave_par_loop (lz4_decode_wave.h) up to its copies.  entry: ip + 1536 <= iend,
// ip <= iend - 306, op <= oend - 606 (the copier checked); ends with an EXIT message {ip, op of the first sequence that was not
// taken} ----
template <class Grp>
LZ4HIP_DEV void pair_parse_loop(Grp& g, const uint8_t* src, const uint32_t iend, const uint32_t oend, const uint32_t db, uint32_t ip, uint32_t op) {
  typedef typename Grp::LChunk LChunk;
  typedef typename Grp::VU VU;
  typedef typename Grp::VB VB;
  constexpr uint32_t STEP = 256u, TRIPMAX = 2048u, AHEAD = 512u;   // (lz4_decode_wave.h)
  const uint32_t KW = g.wv_ring(), KS = g.wv_stream();
  const uint32_t ilim = iend - 306u, olim = oend - 606u;
  const uint32_t op0 = op;
  uint32_t head = 0u, tail_seen = 0u;          // (both counters start at zero with every entry: the copier reset them before it raised CMD)
  uint32_t qip[PAIR_SLOTS];                    // stream positions of the messages in flight, oldest first (the last `head - tail` entries)
#pragma unroll
  for (uint32_t k = 0; k < PAIR_SLOTS; k++) qip[k] = ip;
  // everything from here on is needed by somebody: the parser itself from ip on, the copier from the oldest unfinished message on
  auto keep_from = [&](uint32_t ip_now) -> uint32_t {
    const uint32_t k = head - tail_seen;       // 0 .. PAIR_SLOTS
    uint32_t lo = ip_now;
#pragma unroll
    for (uint32_t j = 0; j < PAIR_SLOTS; j++) lo = (k == PAIR_SLOTS - j) ? qip[j] : lo;
    return lo & ~(STEP - 1u);
  };
  auto post = [&](const VU& w0, const VU& w1, uint32_t ip_msg) {
    while (head - tail_seen >= PAIR_SLOTS) {   // the mailbox is full: the copier is behind
      tail_seen = g.pm_peek(PC_TAIL);
      if (head - tail_seen >= PAIR_SLOTS) g.pm_nap();
    }
    g.pm_put(head % PAIR_SLOTS, w0, w1);
    head++;
    g.pm_post(PC_HEAD, head);
#pragma unroll
    for (uint32_t j = 0; j + 1u < PAIR_SLOTS; j++) qip[j] = qip[j + 1u];
    qip[PAIR_SLOTS - 1u] = ip_msg;
  };
  uint32_t avail = ip & ~(STEP - 1u);
  {
    const LChunk a0 = g.rs_fetch(src, avail), a1 = g.rs_fetch(src, avail + STEP), a2 = g.rs_fetch(src, avail + 2u * STEP), a3 = g.rs_fetch(src, avail + 3u * STEP);
    g.rs_put(avail, a0); g.rs_put(avail + STEP, a1); g.rs_put(avail + 2u * STEP, a2); g.rs_put(avail + 3u * STEP, a3);
    avail += 4u * STEP;
  }
  LChunk rf0 = LChunk();
  uint32_t fl = (op + db) & ~(STEP - 1u);      // where the copier's flusher stands when it starts on this trip: a function of op
  const VU lane = g.vlane();
  const VU p0 = lane * 4u;
  uint32_t wild = op;

  for (;;) {
    if (!((ip <= ilim) & (op <= olim))) break;
    tail_seen = g.pm_peek(PC_TAIL);
    // ---- the stream ring holds what this trip may read; a step may go in where neither wavefront needs the bytes it replaces ----
    if (LZ4HIP_UNLIKELY(ip + AHEAD > avail)) {
      for (;;) {
        while ((ip + AHEAD > avail) & (avail + STEP <= iend) & (avail + STEP <= keep_from(ip) + KS)) {
          g.rs_put(avail, g.rs_fetch(src, avail));
          avail += STEP;
        }
        if (!((ip + AHEAD > avail) & (avail + STEP <= iend) & (head != tail_seen))) break;   // enough, the end of the stream, or nobody to wait for
        g.pm_nap();                             // the copier still reads literals where the next step would go
        tail_seen = g.pm_peek(PC_TAIL);
      }
      if (ip + AHEAD > avail) break;            // (the end of the stream is near: the loops behind this one do the rest)
    }
    uint32_t nf = 0u;
    if ((avail + STEP <= iend) & (avail + STEP <= keep_from(ip) + KS)) {
      rf0 = g.rs_fetch(src, avail);
      nf = 1u;
    }
    // ---- 1 + 2. discovery and walk (lz4_decode_wave.h) ----
    VU posv = VU(0u);
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
        nxpack = nxpack | (Grp::vsel(nxt <= 250u, nxt, VU(255u)) << (8 * (int)j));
      }
      Grp::vwalk(nxpack, posv, T);
    }
    // ---- 3. records, output positions, the rounds: a lane per run; one message per pass ----
    const VB isM = (lane & 1u) != 0u;
    uint32_t tk = 0u, opc = op, lend = 0u;
    for (;;) {
      const uint32_t np = T - tk < 31u ? T - tk : 31u;
      const VU sq = (lane >> 1) + tk;
      const VU pv = Grp::vshfl(posv, sq);
      const uint64_t actm = (1ull << (2u * np)) - 1ull;
      const VB act = Grp::vlanes(actm);
      const VU sl = pv >> 2;
      const VU hw = Grp::valignbyte(Grp::vshfl(bhi, sl), Grp::vshfl(blo, sl), pv & 3u);
      const VU tl = (hw >> 4) & 15u, tm = hw & 15u, e1 = (hw >> 8) & 255u;
      const VB l15 = tl == 15u;
      const VU lit = tl + Grp::vsel(l15, e1, VU(0u));
      const VU lp = pv + ip + Grp::vsel(l15, VU(2u), VU(1u));
      const VU ow = g.vs_ld32(lp + lit);
      const VU off = ow & 0xFFFFu, e2 = (ow >> 16) & 255u;
      const VB m15 = tm == 15u;
      const VU mlx = tm + Grp::vsel(m15, e2, VU(0u)), ml = mlx + 4u;
      const VU endp = (lp - ip) + lit + Grp::vsel(m15, VU(3u), VU(2u));
      const uint64_t simplem = Grp::vballot(off != 0u) & Grp::vballot(lit < 255u) & Grp::vballot(mlx < 255u);
      const VU len = Grp::vsel(isM, ml, lit);
      const VU tot = Grp::vsel(act, len, VU(0u));
      const VU ex = Grp::vexcl_scan(tot);
      const VU o = ex + opc;
      const VU mp = o - off;
      const VU oe = o + tot;
      const VU send = mp + ml;
      const uint32_t oe_all = Grp::vreadlane(oe, 2u * np - 1u);
      const uint32_t tb = (oe_all - op < TRIPMAX ? oe_all : op + TRIPMAX) + 16u;
      const uint32_t bound = (int32_t)(wild - tb) > 0 ? wild : tb;
      const uint32_t memlim = fl > op0 + db ? fl - db : op0;
      constexpr uint64_t litm = 0x5555555555555555ull;
      const uint64_t heldm = Grp::vballot(mp >= VU(op0)) & Grp::vballot((mp + KW) >= VU(bound));
      const uint64_t srcm = Grp::vballot(mp < VU(0x80000000u)) & (heldm | Grp::vballot(send <= VU(memlim)));
      const uint64_t okbm = actm & simplem & (litm | srcm) & Grp::vballot((pv + ip) <= VU(ilim)) & Grp::vballot(o <= VU(olim)) & Grp::vballot((oe - op) <= VU(TRIPMAX));
      const uint64_t farm = ~litm & ~heldm;
      const VU spv = Grp::vsel(isM, mp + db, lp);
      const uint64_t oddm = g.vodd_mask(o + db, len);
      // the rounds (lz4_decode_wave.h step 4): which lanes start one -- the copies themselves are the copier's
      uint32_t a = 0u;
      uint64_t rsm = 0ull;
      for (;;) {
        const uint32_t oa = Grp::vreadlane(o, a);
        const uint64_t below = (1ull << a) - 1ull;
        const uint64_t okm = (okbm & (litm | Grp::vballot(send <= VU(oa)))) | below;
        const uint32_t Te = (uint32_t)__builtin_ctzll(~okm | (1ull << 63));
        if (Te == a) break;
        rsm |= 1ull << a;
        a = Te;
        if (a >= 2u * np) break;
      }
      a &= ~1u;                                 // (a literal run whose match was not taken stays with its sequence)
      if (a == 0u) break;
      tk += a >> 1;
      opc = Grp::vreadlane(oe, a - 1u);
      lend = Grp::vreadlane(endp, a - 1u);
      {
        VU w0 = ((o + db) & 0xFFFFu) | (len << 16) | Grp::vsel(Grp::vlanes(farm), VU(PAIR_F_FAR), VU(0u)) | Grp::vsel(Grp::vlanes(oddm), VU(PAIR_F_ODD), VU(0u)) |
                Grp::vsel(Grp::vlanes(rsm), VU(PAIR_F_ROUND), VU(0u));
        VU w1 = spv;
        w0 = Grp::vwritelane(w0, PAIR_PASS | (a << 8), 62u);
        w1 = Grp::vwritelane(w1, opc, 62u);
        post(w0, w1, ip);
      }
      if ((a < 2u * np) | (tk >= T)) break;
    }
    if (LZ4HIP_UNLIKELY(tk == 0u)) {
      const uint32_t ip1 = ip, op1 = op;
      if (!wave_single_dry(g, ip, op, op0, fl, db, ilim, olim)) break;
      VU w0 = VU(0u), w1 = VU(0u);
      w0 = Grp::vwritelane(w0, PAIR_SINGLE, 62u);
      w1 = Grp::vwritelane(w1, op1, 62u);
      w0 = Grp::vwritelane(w0, ip1, 63u);
      post(w0, w1, ip1);
      wild = op + STEP;
    } else {
      ip += tk < T ? Grp::vreadlane(posv, tk) : lend;
      op = opc;
    }
    if (nf != 0u) {
      g.rs_put(avail, rf0);
      avail += STEP;
    }
    fl = (op + db) & ~(STEP - 1u);
  }
  VU w0 = VU(0u), w1 = VU(0u);
  w0 = Grp::vwritelane(w0, PAIR_EXIT, 62u);
  w1 = Grp::vwritelane(w1, op, 62u);
  w0 = Grp::vwritelane(w0, ip, 63u);
  post(w0, w1, ip);
}

// the parser wavefront's life: wait for an entry of its block's copier (CMD), parse until the loop ends, wait again; QUIT ends it
template <class Grp>
LZ4HIP_DEV void pair_parser_service(Grp& g, uint8_t* lds) {
  g.pm_begin(lds);
  uint32_t cmd_seen = 0u;
  for (;;) {
    uint32_t c;
    while ((c = g.pm_peek(PC_CMD)) == cmd_seen) g.pm_idle();
    cmd_seen = c;
    if (g.pm_peek(PC_QUIT) != 0u) return;
    const uint32_t ip = g.pm_peek(PC_IP), op = g.pm_peek(PC_OP), iend = g.pm_peek(PC_IEND), oend = g.pm_peek(PC_OEND), db = g.pm_peek(PC_DB);
    const uint8_t* src = g.pm_ptr(g.pm_peek(PC_SRC_LO), g.pm_peek(PC_SRC_HI));
    g.wv_begin_db(lds, db);
    pair_parse_loop(g, src, iend, oend, db, ip, op);
  }
}
template <class Grp>
LZ4HIP_DEV void pair_parser_quit(Grp& g, uint8_t* lds) {   // (the copier wavefront, when its last block is done)
  g.pm_begin(lds);
  g.pm_post(PC_QUIT, 1u);
  g.pm_post(PC_CMD, g.pm_peek(PC_CMD) + 1u);
}

// ---- the COPIER's loop (decode_block PIPE == 7 calls it where PIPE 5 calls decode_wave_par_loop; same entry conditions, same
// contract: leaves with ip / op at the first sequence that was not decoded, everything below op in memory) ----
template <class Grp>
LZ4HIP_DEV void decode_pair_loop(Grp& g, const uint8_t* src, const int iend, uint8_t* dst, const int oend, int& ip_io, int& op_io, uint8_t* lds) {
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
  // ---- start the parser on this entry (it is idle: the last entry's EXIT message has been taken) ----
  g.pm_post(PC_HEAD, 0u); g.pm_post(PC_TAIL, 0u);
  g.pm_post(PC_IP, ip); g.pm_post(PC_OP, op); g.pm_post(PC_IEND, (uint32_t)iend); g.pm_post(PC_OEND, (uint32_t)oend); g.pm_post(PC_DB, db);
  g.pm_post(PC_SRC_LO, (uint32_t)(uintptr_t)src); g.pm_post(PC_SRC_HI, (uint32_t)((uint64_t)(uintptr_t)src >> 32));
  g.pm_post(PC_CMD, g.pm_peek(PC_CMD) + 1u);
  const VB isM = (g.vlane() & 1u) != 0u;
  uint32_t tail = 0u;
  for (;;) {
    while (g.pm_peek(PC_HEAD) == tail) g.pm_nap();
    VU w0, w1;
    g.pm_get(tail % PAIR_SLOTS, w0, w1);
    const uint32_t hdr = Grp::vreadlane(w0, 62u), opn = Grp::vreadlane(w1, 62u);
    const uint32_t kind = hdr & 255u;
    if (kind == PAIR_PASS) {
      const uint32_t a_end = hdr >> 8;
      const VU dw = w0 & 0xFFFFu, len = (w0 >> 16) & 0x1FFu;
      const uint64_t farm = Grp::vballot((w0 & PAIR_F_FAR) != 0u), oddm = Grp::vballot((w0 & PAIR_F_ODD) != 0u), rsm = Grp::vballot((w0 & PAIR_F_ROUND) != 0u);
      const VU mp = w1 - db;                    // (match lanes: the source as an output position -- where a far one lies in memory)
      uint32_t a = 0u;
      do {
        const uint64_t later = rsm & ~((2ull << a) - 1ull);           // the rounds that start behind lane a (a <= 61)
        uint32_t Te = later != 0ull ? (uint32_t)__builtin_ctzll(later) : a_end;
        Te = Te < a_end ? Te : a_end;
        g.vcopy_run(dw, !isM, w1, len, ((1ull << Te) - 1ull) & ~((1ull << a) - 1ull), dst, mp, farm, oddm);
        a = Te;
      } while (a < a_end);
      op = opn;
    } else if (kind == PAIR_SINGLE) {
      ip = Grp::vreadlane(w0, 63u); op = opn;
      (void)wave_single_step(g, dst, ip, op, op0, fl, db, ilim, olim);   // (true: the parser ran the same checks on the same values)
    } else {                                    // PAIR_EXIT
      ip = Grp::vreadlane(w0, 63u); op = opn;
    }
    tail++;
    g.pm_post(PC_TAIL, tail);                   // behind the message's last read of the stream ring
    // ---- flusher: whole aligned steps below op ----
    while (op + db - fl >= STEP) { g.wv_store(dst, fl, g.wv_read_al(fl), op0 + db, 0xFFFFFFFFu); fl += STEP; }
    if (kind == PAIR_EXIT) break;
  }
  if (op + db != fl) g.wv_store(dst, fl, g.wv_read_al(fl), op0 + db, op + db);
  ip_io = (int)ip; op_io = (int)op;
}

}  // namespace lz4hip

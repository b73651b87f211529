// mail_ring.h -- the finder / writer hand-over of the default fast-compress kernel (kernels.hip, compress_fast_v2w_cu_kernel).
//
// A finder wavefront (lz4_fast_v2_core.h) parks bare hits 64 at a time; every full batch -- 3 x 64 words + a header {kind, block,
// count, x} -- goes into a slot of a small ring in global memory, and a WRITER wavefront of the same workgroup takes the slots in
// order and does all the output of the block (liblz4's backward extension of the 64 hits, literal copies, tokens, capacity checks,
// last literals, the block's result word: LZ4_compress_default, /root/reference/src/jni/net_jpountz_lz4_LZ4JNI.c:75).
// Single producer, single consumer, two counters: ctr[0] = slots published by the finder, ctr[1] = slots the writer has taken
// into its registers (a slot may be overwritten as soon as it has been READ, not when its batch has been written).
// Messages: BATCH (64 or fewer parked hits of block b), LAST (block b ends: last literals from x, result word), ABORT (block b
// is left to the window-parallel kernel: forget it), EXIT (the queue is empty).
//
// Written against the wave backend W and a memory policy M { peek, peek_far, acquire, publish, nap_finder, nap_writer, uptr, u32,
// block_begin, result }: the device policy (kernels.hip, MailDev) orders at workgroup scope; the CPU suite's policy
// (tests/hostsim) uses two host threads with acquire / release atomics, a ring of two slots and random naps, and runs THIS source.
#pragma once
#include "lz4_fast_v2_core.h"

#ifndef LZ4HIP_MAIL_RING
#define LZ4HIP_MAIL_RING 4
#endif

namespace lz4hip {

struct BatchArgs;

constexpr uint32_t MAIL_RING = LZ4HIP_MAIL_RING;   // slots per finder/writer pair
constexpr uint32_t MAIL_SLOT_WORDS = 256u;    // 3 x 64 sequence words + header {kind, block, count, x}
enum : uint32_t { MAIL_BATCH = 1u, MAIL_LAST = 2u, MAIL_ABORT = 3u, MAIL_EXIT = 4u };
constexpr uint32_t MAIL_PENDING = 0xFFFFFFFFu;   // MailOut::last(): the writer produces the result

// finder side: the Out policy of FastV2 / FastCore (same parking as ParkOut; a full batch goes to the partner instead of memory)
template <class W, class M>
struct MailOutT {
  using VU = typename W::VU;
  static constexpr bool kUsesWindowRegs = false;
  static constexpr uint32_t kNoCheck = ParkOut<W>::kNoCheck, kFinal = ParkOut<W>::kFinal;
  static constexpr bool kAsmPark = true;   // (lz4_fast_v2_asm.h parks into p_ms / p_ml / p_off and counts in cnt)
  static constexpr bool kRawPark = true;   // the lean loop parks bare hits: liblz4's backward extension is the writer's work too (ParkOut::resolve_raw)
  W& w;
  uint32_t* slots;   // MAIL_RING x MAIL_SLOT_WORDS
  uint32_t* ctr;     // {published by the finder, consumed by the writer}
  uint32_t head, tail_seen = 0, b = 0;
  VU p_ms = VU(0u), p_ml = VU(0u), p_off = VU(0u);
  uint32_t cnt = 0;
  uint32_t dense64 = 0, flushes = 0, mark = 0;
  bool bail = false;

  LZ4HIP_DEV MailOutT(W& w_, uint32_t* slots_, uint32_t* ctr_, uint32_t head_) : w(w_), slots(slots_), ctr(ctr_), head(head_) {}

  LZ4HIP_DEV void post(uint32_t kind, uint32_t m, uint32_t x) {
    for (uint32_t spin = 1; head - tail_seen >= MAIL_RING; spin++) {   // ring full: the writer is behind (it publishes `tail` once a slot is in its registers)
      tail_seen = (spin & 63u) ? M::peek(ctr + 1) : M::peek_far(ctr + 1);
      if (head - tail_seen >= MAIL_RING) M::nap_finder();
    }
    uint32_t* s = slots + (head % MAIL_RING) * MAIL_SLOT_WORDS;
    W::st_lanes(s, p_ms); W::st_lanes(s + 64u, p_ml); W::st_lanes(s + 128u, p_off);
    W::st_hdr(s + 192u, kind, b, m, x);
    head++;
    M::publish(ctr, head);
  }
  LZ4HIP_DEV void park(uint32_t ms, uint32_t ml, uint32_t offx) {
    p_ms = W::writelane(p_ms, ms, cnt);
    p_ml = W::writelane(p_ml, ml, cnt);
    p_off = W::writelane(p_off, offx, cnt);
    if (++cnt == 64u) batch();
  }
  LZ4HIP_DEV void batch() {
    const uint32_t m = cnt;
    cnt = 0u;
    if (m == 0u || bail) return;
    if (dense64 != 0u && flushes < 2u && m == 64u) {   // the density probe of ParkOut::flush (same rule, same moment)
      const uint32_t e31 = w.bcast(p_ms + p_ml, 31);
      if (flushes == 0u) mark = e31;
      else if (e31 - mark < dense64) { bail = true; return; }
      flushes++;
    }
    post(MAIL_BATCH, m, 0u);
  }
  // ---- the Out interface of FastCore ----
  LZ4HIP_DEV bool overlap_point() { return true; }
  LZ4HIP_DEV void seq(uint32_t lit, uint32_t mc, uint32_t offset, uint32_t anchor, bool check_lits, bool, VU) {
    park(anchor + lit, mc + 4u, offset | (check_lits ? 0u : kNoCheck) | kFinal);   // the exact path hands over finished sequences
  }
  LZ4HIP_DEV uint32_t last(uint32_t anchor) {
    batch();
    if (bail) return 0u;
    post(MAIL_LAST, 0u, anchor);
    return MAIL_PENDING;
  }
};

// writer side: one wavefront, all blocks of its finder in order; A = BatchArgs (kernels.h)
template <class W, class M, class A>
LZ4HIP_DEV void mail_writer_t(W& w, const A& a, uint32_t* slots, uint32_t* ctr) {
  using VU = typename W::VU;
  uint32_t tail = 0, cur = 0xFFFFFFFFu, op = 0, prev_end = 0;
  bool ok = true;
  for (;;) {
    for (uint32_t spin = 1; ((spin & 63u) ? M::peek(ctr) : M::peek_far(ctr)) == tail; spin++) M::nap_writer();
    M::acquire();
    const uint32_t* s = slots + (tail % MAIL_RING) * MAIL_SLOT_WORDS;
    const VU ms = W::ld_lanes(s), ml = W::ld_lanes(s + 64u), off = W::ld_lanes(s + 128u);
    uint32_t kind, b, m, x;
    W::ld_hdr(s + 192u, kind, b, m, x);
    tail++;
    M::publish(ctr + 1, tail);   // the slot is in registers: the finder may reuse it
    if (kind == MAIL_EXIT) return;
    if (kind == MAIL_ABORT) { cur = 0xFFFFFFFFu; continue; }
    if (b != cur) { cur = b; op = 0; prev_end = 0; ok = true; }
    const int32_t n = M::u32(a.src_len[b]);
    const int32_t cap = M::u32(a.dst_cap[b]);
    const uint8_t* sp = M::uptr(a.src + a.src_off[b]);
    uint8_t* dp = M::uptr(a.dst + a.dst_off[b]);
    M::block_begin(w, sp, (uint32_t)n, dp, (uint32_t)cap);
    ParkOut<W> out(w, sp, (uint32_t)n, dp, (uint32_t)cap);
    out.op = op; out.prev_end = prev_end; out.ok = ok;
    if (kind == MAIL_BATCH) {
      out.p_ms = ms; out.p_ml = ml; out.p_off = off; out.cnt = m;
      out.resolve_raw();
      out.flush();
      op = out.op; prev_end = out.prev_end; ok = out.ok;
    } else {   // MAIL_LAST
      const uint32_t r = ok ? out.emit_last(x) : 0u;
      M::result(a.out, b, (int32_t)r);
      cur = 0xFFFFFFFFu;
    }
  }
}

// writer side, TWO finders per writer (the ten-chain kernel of packed byU32 blocks has ten finders and six writers in its 1024
// threads): the rings are polled in turn, a message is handled exactly as above with the state of its ring.  A writer is busy for a
// small fraction of the time its finders need to fill a batch, and a ring holds MAIL_RING batches, so a finder rarely waits.
// slots1 == nullptr: one ring only.
template <class W, class M>
struct MailRingSide {
  uint32_t* slots;
  uint32_t* ctr;
  uint32_t tail = 0, cur = 0xFFFFFFFFu, op = 0, prev_end = 0;
  bool ok = true, alive = true;
  template <class A>
  LZ4HIP_DEV void serve(W& w, const A& a) {   // one message is waiting
    using VU = typename W::VU;
    M::acquire();
    const uint32_t* s = slots + (tail % MAIL_RING) * MAIL_SLOT_WORDS;
    const VU ms = W::ld_lanes(s), ml = W::ld_lanes(s + 64u), off = W::ld_lanes(s + 128u);
    uint32_t kind, b, m, x;
    W::ld_hdr(s + 192u, kind, b, m, x);
    tail++;
    M::publish(ctr + 1, tail);
    if (kind == MAIL_EXIT) { alive = false; return; }
    if (kind == MAIL_ABORT) { cur = 0xFFFFFFFFu; return; }
    if (b != cur) { cur = b; op = 0; prev_end = 0; ok = true; }
    const int32_t n = M::u32(a.src_len[b]);
    const int32_t cap = M::u32(a.dst_cap[b]);
    const uint8_t* sp = M::uptr(a.src + a.src_off[b]);
    uint8_t* dp = M::uptr(a.dst + a.dst_off[b]);
    M::block_begin(w, sp, (uint32_t)n, dp, (uint32_t)cap);
    ParkOut<W> out(w, sp, (uint32_t)n, dp, (uint32_t)cap);
    out.op = op; out.prev_end = prev_end; out.ok = ok;
    if (kind == MAIL_BATCH) {
      out.p_ms = ms; out.p_ml = ml; out.p_off = off; out.cnt = m;
      out.resolve_raw();
      out.flush();
      op = out.op; prev_end = out.prev_end; ok = out.ok;
    } else {   // MAIL_LAST
      const uint32_t r = ok ? out.emit_last(x) : 0u;
      M::result(a.out, b, (int32_t)r);
      cur = 0xFFFFFFFFu;
    }
  }
  LZ4HIP_DEV bool waiting(uint32_t spin) const { return alive && ((spin & 63u) ? M::peek(ctr) : M::peek_far(ctr)) != tail; }
};
template <class W, class M, class A>
LZ4HIP_DEV void mail_writer2_t(W& w, const A& a, uint32_t* slots0, uint32_t* ctr0, uint32_t* slots1, uint32_t* ctr1) {
  MailRingSide<W, M> r0, r1;
  r0.slots = slots0; r0.ctr = ctr0;
  r1.slots = slots1; r1.ctr = ctr1; r1.alive = slots1 != nullptr;
  for (uint32_t spin = 1; r0.alive || r1.alive; spin++) {
    const bool w0 = r0.waiting(spin), w1 = r1.waiting(spin);
    if (w0) r0.serve(w, a);
    if (w1) r1.serve(w, a);
    if (!w0 && !w1) M::nap_writer();
  }
}

}  // namespace lz4hip

// tests/hostsim/group_host.h -- TEST INFRASTRUCTURE ONLY.
// Lock-step simulator of a GL-lane group for lz4_decode_core.h: every "instruction" of a copy loop
// performs all lanes' loads first and all lanes' stores afterwards (SIMT order), with the same
// chunking, wild over-copy and replicate logic as lz4-java_amd/csrc/group_dev.h, and records any
// access outside the source / destination slots.
#pragma once
#include <stdint.h>
#include <string.h>
#include <atomic>
#include <thread>
#include <vector>
#include "wave_host.h"

namespace hostsim {

struct GroupHost {
  int GL;
  const uint8_t* src_lo; const uint8_t* src_hi;
  uint8_t* dst_lo; uint8_t* dst_hi;
  bool oob = false;
  GroupHost(int gl, const uint8_t* s, long n, uint8_t* d, long cap) : GL(gl), src_lo(s), src_hi(s + n), dst_lo(d), dst_hi(d + cap) {}

  bool rd_ok(const uint8_t* p, long k) {
    bool in_src = p >= src_lo && p + k <= src_hi, in_dst = p >= dst_lo && p + k <= dst_hi;
    if (!in_src && !in_dst) { oob = true; return false; }
    return true;
  }
  bool wr_ok(uint8_t* p, long k) { if (p < dst_lo || p + k > dst_hi) { oob = true; return false; } return true; }

  uint32_t ld8(const uint8_t* p) { return rd_ok(p, 1) ? *p : 0; }
  uint32_t ld16(const uint8_t* p) { uint16_t v = 0; if (rd_ok(p, 2)) memcpy(&v, p, 2); return v; }
  uint32_t ld32(const uint8_t* p) { uint32_t v = 0; if (rd_ok(p, 4)) memcpy(&v, p, 4); return v; }
  uint64_t ld64(const uint8_t* p) { uint64_t v = 0; if (rd_ok(p, 8)) memcpy(&v, p, 8); return v; }

  void copy_lits(uint8_t* d, const uint8_t* s, uint32_t len, bool wild) {
    if (wild) {
      for (uint32_t base = 0; base < len; base += 4u * GL) {
        uint32_t v[64]; bool act[64];
        for (int l = 0; l < GL; l++) { uint32_t i = base + 4u * l; act[l] = i < len; v[l] = 0; if (act[l] && rd_ok(s + i, 4)) memcpy(&v[l], s + i, 4); }
        for (int l = 0; l < GL; l++) { uint32_t i = base + 4u * l; if (act[l] && wr_ok(d + i, 4)) memcpy(d + i, &v[l], 4); }
      }
    } else {
      for (uint32_t base = 0; base < len; base += GL) {
        uint8_t v[64]; bool act[64];
        for (int l = 0; l < GL; l++) { uint32_t i = base + l; act[l] = i < len; v[l] = 0; if (act[l] && rd_ok(s + i, 1)) v[l] = s[i]; }
        for (int l = 0; l < GL; l++) { uint32_t i = base + l; if (act[l] && wr_ok(d + i, 1)) d[i] = v[l]; }
      }
    }
  }

  // wide variants (interior loop): LB = 64/GL bytes per lane and step, lock-step loads then stores
  void copy_lits_wide(uint8_t* d, const uint8_t* s, uint32_t len) {
    const uint32_t LB = lb();
    for (uint32_t base = 0; base < len; base += LB * GL) {
      uint8_t v[64][32]; bool act[64];
      for (int l = 0; l < GL; l++) { uint32_t i = base + LB * l; act[l] = i < len; if (act[l] && rd_ok(s + i, LB)) memcpy(v[l], s + i, LB); }
      for (int l = 0; l < GL; l++) { uint32_t i = base + LB * l; if (act[l] && wr_ok(d + i, LB)) memcpy(d + i, v[l], LB); }
    }
  }
  void copy_match_wide(uint8_t* dst, uint32_t op, uint32_t offset, uint32_t len) {
    const uint32_t LB = lb();
    if (offset >= LB * GL) {
      uint8_t* d = dst + op;
      const uint8_t* m = d - offset;
      for (uint32_t base = 0; base < len; base += LB * GL) {
        uint8_t v[64][32]; bool act[64];
        for (int l = 0; l < GL; l++) { uint32_t i = base + LB * l; act[l] = i < len; if (act[l] && rd_ok(m + i, LB)) memcpy(v[l], m + i, LB); }
        for (int l = 0; l < GL; l++) { uint32_t i = base + LB * l; if (act[l] && wr_ok(d + i, LB)) memcpy(d + i, v[l], LB); }
      }
    } else {
      copy_match(dst, op, offset, len, true);
    }
  }

  // split sequence copy of the pipelined interior loop (group_dev.h seq_load / seq_store), lock-step per call
  struct SeqRegs { uint8_t v[64][32], u[64][32]; };
  uint32_t step() const { return lb() * (uint32_t)GL; }
  uint32_t lb() const { return GL == 1 ? 16u : (64u / GL < 4u ? 4u : 64u / GL); }   // (GL == 1: the ring loop's lane-per-block form)
  uint32_t slack() const { return lb(); }
  void seq_load(SeqRegs& r, const uint8_t* s, uint32_t lit, const uint8_t* m, uint32_t len) {
    const uint32_t LB = lb();
    for (int l = 0; l < GL; l++) {
      const uint32_t i = LB * l;
      if (i < lit && rd_ok(s + i, LB)) memcpy(r.v[l], s + i, LB);
      if (i < len && rd_ok(m + i, LB)) memcpy(r.u[l], m + i, LB);
    }
  }
  void seq_store(const SeqRegs& r, uint8_t* d, uint32_t lit, uint32_t len) {
    const uint32_t LB = lb();
    for (int l = 0; l < GL; l++) { const uint32_t i = LB * l; if (i < lit && wr_ok(d + i, LB)) memcpy(d + i, r.v[l], LB); }
    for (int l = 0; l < GL; l++) { const uint32_t i = LB * l; if (i < len && wr_ok(d + lit + i, LB)) memcpy(d + lit + i, r.u[l], LB); }
  }

  // output staging (group_dev.h st_*): same chunking, same flush rules; an index outside the staging buffer counts as oob
  static constexpr uint32_t kStage = 576u;
  uint8_t stg_buf[kStage + 64];
  uint8_t* stg = nullptr;
  uint32_t fl = 0;
  bool st_ok(uint32_t idx, uint32_t k) { if (idx + k > kStage) { oob = true; return false; } return true; }
  void st_begin(uint8_t*, uint32_t op) { stg = stg_buf; fl = op; memset(stg_buf, 0xEE, sizeof stg_buf); }
  void st_flush_lines(uint8_t* dst, uint32_t p) {
    const uint32_t LB = lb();
    const uint32_t target = p - (((uint32_t)(uintptr_t)dst + p) & 127u);
    if ((int32_t)(target - fl) <= 0) return;
    const uint32_t nb = target - fl, rem = p - target;
    for (uint32_t base = 0; base < nb; base += LB * GL)
      for (int l = 0; l < GL; l++) { const uint32_t o = base + LB * l; if (o < nb && st_ok(o, LB) && wr_ok(dst + fl + o, LB)) memcpy(dst + fl + o, stg + o, LB); }
    for (uint32_t base = 0; base < rem; base += LB * GL) {
      uint8_t v[64][32]; bool act[64];
      for (int l = 0; l < GL; l++) { const uint32_t o = base + LB * l; act[l] = o < rem && st_ok(nb + o, LB); if (act[l]) memcpy(v[l], stg + nb + o, LB); }
      for (int l = 0; l < GL; l++) { const uint32_t o = base + LB * l; if (act[l]) memcpy(stg + o, v[l], LB); }
    }
    fl = target;
  }
  void st_flush_all(uint8_t* dst, uint32_t p) {
    const uint32_t LB = lb();
    if ((int32_t)(p - fl) <= 0) return;   // (st_lits may have flushed lines past p: a sequence that then leaves the loop has nothing left to flush)
    const uint32_t nb = p - fl;
    for (uint32_t base = 0; base < nb; base += LB * GL)
      for (int l = 0; l < GL; l++) { const uint32_t o = base + LB * l; if (o < nb && st_ok(o, LB) && wr_ok(dst + fl + o, LB)) memcpy(dst + fl + o, stg + o, LB); }
    fl = p;
  }
  void st_lits(uint8_t* dst, uint32_t op, const uint8_t* s, uint32_t len) {
    const uint32_t LB = lb();
    for (uint32_t base = 0; base < len; base += LB * GL) {
      if (op + base - fl + LB * GL + LB > kStage) st_flush_lines(dst, op + base);
      uint8_t v[64][32]; bool act[64];
      for (int l = 0; l < GL; l++) { const uint32_t i = base + LB * l; act[l] = i < len && rd_ok(s + i, LB); if (act[l]) memcpy(v[l], s + i, LB); }
      for (int l = 0; l < GL; l++) { const uint32_t i = base + LB * l; if (act[l] && st_ok(op + i - fl, LB)) memcpy(stg + (op + i - fl), v[l], LB); }
    }
  }
  void st_match(uint8_t* dst, uint32_t op, uint32_t offset, uint32_t len) {
    const uint32_t LB = lb();
    const uint8_t* m = dst + op - offset;
    for (uint32_t base = 0; base < len; base += LB * GL) {
      if (op + base - fl + LB * GL + LB > kStage) st_flush_lines(dst, op + base);
      uint8_t v[64][32]; bool act[64];
      for (int l = 0; l < GL; l++) {
        const uint32_t i = base + LB * l;
        act[l] = i < len;
        if (act[l] && (uint32_t)(op - offset + i) + LB > fl) { oob = true; act[l] = false; }   // the source must be flushed memory
        if (act[l] && !rd_ok(m + i, LB)) act[l] = false;
        if (act[l]) memcpy(v[l], m + i, LB);
      }
      for (int l = 0; l < GL; l++) { const uint32_t i = base + LB * l; if (act[l] && st_ok(op + i - fl, LB)) memcpy(stg + (op + i - fl), v[l], LB); }
    }
  }

  // ---- backend of the deep interior loop (lz4_decode_deep.h; group_dev.h sr_* / step_*): the stream ring in "LDS" (an index
  // outside it counts as oob), whole 64-byte steps by all lanes -- every call is one instruction: all lanes' loads, then all stores
  static constexpr uint32_t kStream = 1024u, kStreamLds = kStream + 16u, kPiece = 256u;
  struct LChunk { uint8_t b[64][32]; };
  struct PieceRegs { uint8_t b[64][64]; };
  uint8_t sr_mem[kStreamLds];
  bool lds_ok(uint32_t idx, uint32_t k) { if (idx + k > kStreamLds) { oob = true; return false; } return true; }
  void sr_begin(uint8_t*) { memset(sr_mem, 0xEE, sizeof sr_mem); }
  uint32_t cb() const { return kPiece / GL < 4u ? 4u : kPiece / GL; }
  PieceRegs sr_fetch(const uint8_t* s, uint32_t pos) {
    PieceRegs r; memset(&r, 0, sizeof r);
    const uint32_t CB = cb();
    for (int l = 0; l < GL; l++) if (rd_ok(s + pos + l * CB, CB)) memcpy(r.b[l], s + pos + l * CB, CB);
    return r;
  }
  void sr_put(uint32_t pos, const PieceRegs& r) {
    const uint32_t CB = cb();
    for (int l = 0; l < GL; l++) {
      const uint32_t q = (pos & (kStream - 1u)) + l * CB;
      if (lds_ok(q, CB)) memcpy(sr_mem + q, r.b[l], CB);
      if (q < 16u) { const uint32_t k = CB < 16u ? CB : 16u; if (lds_ok(kStream + q, k)) memcpy(sr_mem + kStream + q, r.b[l], k); }
    }
  }
  uint32_t sr_ld32(uint32_t p) { uint32_t v = 0; const uint32_t q = p & (kStream - 1u); if (lds_ok(q, 4)) memcpy(&v, sr_mem + q, 4); return v; }
  static inline uint64_t deep_trips = 0;   // (the test asserts that the deep loop really ran)
  uint64_t sr_ld64(uint32_t p) { deep_trips++; uint64_t v = 0; const uint32_t q = p & (kStream - 1u); if (lds_ok(q, 8)) memcpy(&v, sr_mem + q, 8); return v; }
  LChunk sr_step(uint32_t p) {
    LChunk v; memset(&v, 0, sizeof v);
    const uint32_t LB = lb();
    for (int l = 0; l < GL; l++) { const uint32_t q = (p + l * LB) & (kStream - 1u); if (lds_ok(q, LB)) memcpy(v.b[l], sr_mem + q, LB); }
    return v;
  }
  LChunk step_load(const uint8_t* m) {
    LChunk v; memset(&v, 0, sizeof v);
    const uint32_t LB = lb();
    for (int l = 0; l < GL; l++) if (rd_ok(m + l * LB, LB)) memcpy(v.b[l], m + l * LB, LB);
    return v;
  }
  uint32_t lane_bytes() const { return 0; }   // (the simulator's idle lanes read the step's own first chunk: see step_load_upto)
  LChunk step_load_upto(const uint8_t* m, uint32_t len, const uint8_t* idle) {
    LChunk v; memset(&v, 0, sizeof v);
    const uint32_t LB = lb();
    for (int l = 0; l < GL; l++) { const uint8_t* p = ((uint32_t)l * LB < len ? m : idle) + l * LB; if (rd_ok(p, LB)) memcpy(v.b[l], p, LB); }
    return v;
  }
  void step_store(uint8_t* d, const LChunk& v) {
    const uint32_t LB = lb();
    for (int l = 0; l < GL; l++) if (wr_ok(d + l * LB, LB)) memcpy(d + l * LB, v.b[l], LB);
  }

  // ---- backend of the ring loop (lz4_decode_ring.h; group_dev.h rs_* / rg_*): stream ring + output ring in "LDS" with the same
  // layout, the same mirror rules and the same per-lane index arithmetic as the device backend; every call is one instruction (all
  // lanes' loads, then all lanes' stores); an index outside the block's LDS bytes counts as oob.  Ring sizes are chosen per test.
  uint32_t kRs = 256u, kRing = 512u;
  uint8_t ring_mem[1024 + 32 + 8192 + 96 + 64];
  uint8_t* rsb = nullptr; uint8_t* rgb = nullptr;
  uint32_t dbase = 0;
  static inline std::atomic<uint64_t> ring_trips{0};   // (rs_ld64: both wavefronts of a pair bump it)
  static inline uint64_t ring_entries = 0, ring_repl = 0, ring_flush = 0;   // (statistics for the tests)
  uint32_t ring_bytes() const { return kRing; }
  uint32_t ring_step() const { return lb() * (uint32_t)GL; }
  uint32_t ring_piece() const { return GL == 1 ? 16u : lb() * (uint32_t)GL - 4u; }
  void rg_seed(uint32_t pos, const LChunk& v) { rg_write(pos, v); }
  uint32_t ring_stream() const { return kRs; }
  uint32_t ring_dbase() const { return dbase; }
  uint32_t rs_tail() const { return lb() < 16u ? 16u : lb(); }
  uint32_t ring_lds() const { return kRs + rs_tail() + kRing + 3u * lb() + (GL == 1 ? 32u : 0u); }
  bool rl_ok(const uint8_t* p, uint32_t k) {
    if (wave_mode && pair_lds) { if (p < pair_lds || p + k > pair_lds + (kWs + 32u + kWv + 32u)) { oob = true; return false; } return true; }
    if (wave_mode) { if (p < wv_mem.data() || p + k > wv_mem.data() + wv_mem.size()) { oob = true; return false; } return true; }
    if (p < ring_mem || p + k > ring_mem + ring_lds()) { oob = true; return false; } return true;
  }
  void ring_begin(uint8_t*, const uint8_t* dst) {
    ring_entries++;
    memset(ring_mem, 0xEE, sizeof ring_mem);
    rsb = ring_mem; rgb = ring_mem + kRs + rs_tail() + (GL == 1 ? 32u : lb()); dbase = (uint32_t)(uintptr_t)dst & 63u;
  }
  LChunk rs_fetch(const uint8_t* s, uint32_t pos) {
    LChunk r; memset(&r, 0, sizeof r);
    const uint32_t LB = lb();
    for (int l = 0; l < GL; l++) if (rd_ok(s + pos + l * LB, LB)) memcpy(r.b[l], s + pos + l * LB, LB);
    return r;
  }
  LChunk rs_fetch_upto(const uint8_t* s, uint32_t pos, uint32_t iend) {   // (wave backend: a dword per lane; nothing at or behind s + iend is read)
    LChunk r; memset(&r, 0, sizeof r);
    for (int l = 0; l < GL; l++) {
      const uint32_t a = pos + 4u * (uint32_t)l;
      for (uint32_t k = 0; k < 4u; k++) if (a + k < iend && rd_ok(s + a + k, 1)) r.b[l][k] = s[a + k];
    }
    return r;
  }
  void rs_put(uint32_t pos, const LChunk& r) {
    const uint32_t LB = lb();
    for (int l = 0; l < GL; l++) {
      const uint32_t q = (pos & (kRs - 1u)) + l * LB;
      if (rl_ok(rsb + q, LB)) memcpy(rsb + q, r.b[l], LB);
      if (q < rs_tail() && rl_ok(rsb + kRs + q, LB)) memcpy(rsb + kRs + q, r.b[l], LB);
    }
  }
  uint32_t rs_ld32(uint32_t p) { uint32_t v = 0; const uint8_t* q = rsb + (p & (kRs - 1u)); if (rl_ok(q, 4)) memcpy(&v, q, 4); return v; }
  uint64_t rs_ld64(uint32_t p) { ring_trips++; uint64_t v = 0; const uint8_t* q = rsb + (p & (kRs - 1u)); if (rl_ok(q, 8)) memcpy(&v, q, 8); return v; }
  LChunk rs_step(uint32_t p) {
    LChunk v; memset(&v, 0, sizeof v);
    const uint32_t LB = lb();
    for (int l = 0; l < GL; l++) { const uint8_t* q = rsb + ((p + l * LB) & (kRs - 1u)); if (rl_ok(q, LB)) memcpy(v.b[l], q, LB); }
    return v;
  }
  LChunk rg_read(uint32_t pos) {
    LChunk v; memset(&v, 0, sizeof v);
    const uint32_t LB = lb();
    for (int l = 0; l < GL; l++) { const uint8_t* q = rgb + ((pos + dbase + l * LB) & (kRing - 1u)); if (rl_ok(q, LB)) memcpy(v.b[l], q, LB); }
    return v;
  }
  LChunk rg_read_al(uint32_t pos) { if ((pos + dbase) & (ring_step() - 1u)) oob = true; return rg_read(pos); }   // (the flusher's steps are aligned)
  // one chunk of LB bytes to ring index x with the device backend's mirror rule
  void rg_put(uint32_t x, const uint8_t* c) {
    const uint32_t LB = lb();
    const uint32_t t = (x + LB) & (kRing - 1u);
    if (rl_ok((rgb - LB) + t, LB)) memcpy((rgb - LB) + t, c, LB);
    if (t < 2u * LB && rl_ok((rgb - LB) + t + kRing, LB)) memcpy((rgb - LB) + t + kRing, c, LB);
  }
  // the device backend stores a step as ALIGNED chunks (lanes 1..: the step's bytes [l LB - s, l LB - s + LB) at the dword-aligned
  // index below) plus lane 0's own bytes at the position: it covers [pos, pos + 64 - s), s = ring index & 3 -- exactly that here
  void rg_write(uint32_t pos, const LChunk& v) {
    const uint32_t LB = lb();
    const uint32_t w = pos + dbase, s = w & 3u;
    if (GL == 1) {   // the device backend's window store puts exactly the lane's 16 bytes at the position (and keeps what lies below)
      for (uint32_t i = 0; i < 16u; i++) {
        const uint32_t x = (w + i) & (kRing - 1u);
        if (rl_ok(rgb + x, 1)) rgb[x] = v.b[0][i];
        if (x < 32u && rl_ok(rgb + kRing + x, 1)) rgb[kRing + x] = v.b[0][i];
      }
      return;
    }
    uint8_t flat[64 + 16];
    for (int l = 0; l < GL; l++) memcpy(flat + 4 + l * LB, v.b[l], LB);
    memset(flat, 0xDD, 4);   // (what lane 1 gets "from the lane below" never reaches the ring: lane 0's own store covers those bytes)
    rg_put(w, flat + 4);     // lane 0
    for (int l = 1; l < GL; l++) rg_put((w & ~3u) + l * LB, flat + 4 + l * LB - s);
  }
  static bool any(bool x) { return x; }   // (the simulated "wavefront" is this one group)
  static void settle(uint32_t&) {}
  static LChunk pick(bool first, const LChunk& a, const LChunk& b) { return first ? a : b; }

  // ---- backend of the wave loop (lz4_decode_wave.h; group_dev.h BlockWaveDev): one wavefront (GL == 64) per block, a dword per lane; the
  // stream ring (kWs bytes + 16 tail) and the output ring (16 pad | kWv bytes | 16 tail) in "LDS" with the device backend's layout,
  // mirror rule and per-lane index arithmetic; every call is one instruction (all lanes' loads, then all lanes' stores); an index
  // outside the wavefront's LDS bytes counts as oob.  rs_fetch / rs_put / rs_ld64 above serve this ring too (rsb = wsb, kRs = kWs).
  bool wave_mode = false;
  uint32_t kWv = 8192u, kWs = 2048u;
  std::vector<uint8_t> wv_mem;
  uint8_t* wsb = nullptr; uint8_t* wrb = nullptr;
  uint32_t wdb = 0;
  static inline uint64_t wave_trips = 0, wave_entries = 0, wave_far = 0, wave_mirror = 0;
  uint32_t wv_ring() const { return kWv; }
  uint32_t wv_stream() const { return kWs; }
  uint32_t wv_dbase() const { return wdb; }
  static uint32_t uni(uint32_t x) { return x; }
  void wv_begin(uint8_t*, const uint8_t* dst) {
    wave_entries++;
    wave_mode = true;
    if (pair_lds) {   // the pair loop: both wavefronts' backends work on the one block of "LDS" the harness made (the copier enters while the parser is idle: the rings may be poisoned, the mailbox behind them may not)
      memset(pair_lds, 0xEE, kWs + 32u + kWv + 32u);
      wsb = pair_lds;
    } else {
      wv_mem.assign(kWs + 32u + kWv + 32u, 0xEE);
      wsb = wv_mem.data();
    }
    wrb = wsb + kWs + 32u; wdb = (uint32_t)(uintptr_t)dst & 255u;
    rsb = wsb; kRs = kWs;
  }
  // ---- the pair loop (lz4_decode_pair.h): a parser and a copier wavefront per block = two of these backends, each on its own host
  // thread, over ONE block of "LDS" (rings + mailbox).  The mailbox's control words are acquire / release atomics -- what the device
  // gets from the LDS executing a wavefront's instructions in order; every peek / post naps at random so that the interleavings vary ----
  static constexpr uint32_t kMailSlots = 3u, kMailSlotBytes = 512u, kMailBytes = kMailSlots * kMailSlotBytes + 64u;
  uint8_t* pair_lds = nullptr;
  uint8_t* pmb = nullptr;
  uint64_t nap_rng = 0x2545F4914F6CDD1Dull;
  uint32_t pair_lds_bytes() const { return kWs + 32u + kWv + 32u + kMailBytes; }
  void wv_begin_db(uint8_t*, uint32_t db) { wave_mode = true; wsb = pair_lds; wrb = wsb + kWs + 32u; wdb = db; rsb = wsb; kRs = kWs; }
  void pm_begin(uint8_t*) { pmb = pair_lds + kWs + 32u + kWv + 32u; }
  std::atomic<uint32_t>* pm_ctl(uint32_t i) { return reinterpret_cast<std::atomic<uint32_t>*>(pmb + kMailSlots * kMailSlotBytes + 4u * i); }
  void pm_jitter(uint32_t one_in) {
    nap_rng ^= nap_rng << 13; nap_rng ^= nap_rng >> 7; nap_rng ^= nap_rng << 17;
    if (nap_rng % one_in == 0u) std::this_thread::yield();
  }
  uint32_t pm_peek(uint32_t i) { pm_jitter(16); return pm_ctl(i)->load(std::memory_order_acquire); }
  void pm_post(uint32_t i, uint32_t v) { pm_jitter(8); pm_ctl(i)->store(v, std::memory_order_release); pm_jitter(8); }
  void pm_put(uint32_t slot, const V<uint32_t>& w0, const V<uint32_t>& w1) {
    if (slot >= kMailSlots) { oob = true; return; }
    for (int l = 0; l < 64; l++) { memcpy(pmb + slot * kMailSlotBytes + 8u * l, &w0.v[l], 4); memcpy(pmb + slot * kMailSlotBytes + 8u * l + 4u, &w1.v[l], 4); }
  }
  uint32_t pm_peek_get(uint32_t i, uint32_t slot, V<uint32_t>& w0, V<uint32_t>& w1) {   // (the word first, the slot behind it)
    const uint32_t v = pm_peek(i);
    pm_get(slot, w0, w1);
    return v;
  }
  void pm_get(uint32_t slot, V<uint32_t>& w0, V<uint32_t>& w1) {
    if (slot >= kMailSlots) { oob = true; return; }
    for (int l = 0; l < 64; l++) { memcpy(&w0.v[l], pmb + slot * kMailSlotBytes + 8u * l, 4); memcpy(&w1.v[l], pmb + slot * kMailSlotBytes + 8u * l + 4u, 4); }
  }
  // the trio loop's scan queue (lz4_decode_trio.h: scanner -> planner), behind the mailbox
  static constexpr uint32_t kScanSlots = 3u, kScanBytes = 288u;
  uint32_t trio_lds_bytes() const { return pair_lds_bytes() + kScanSlots * kScanBytes; }
  void sq_put(uint32_t slot, const V<uint32_t>& posv, uint32_t T, uint32_t wip, uint32_t nextw, uint32_t flags) {
    if (slot >= kScanSlots) { oob = true; return; }
    uint8_t* e = pmb + kMailBytes + slot * kScanBytes;
    for (int l = 0; l < 64; l++) memcpy(e + 4 * l, &posv.v[l], 4);
    const uint32_t h[4] = {T, wip, nextw, flags};
    memcpy(e + 256, h, 16);
  }
  void pm_peek2(uint32_t i, uint32_t& a, uint32_t& b) { if (i & 1u) oob = true; a = pm_peek(i); b = pm_peek(i + 1u); }   // (the device reads the two words with one aligned 8-byte load: they are published one at a time, so either order of the two loads is one the device can see)
  uint32_t sq_peek_get(uint32_t i, uint32_t slot, V<uint32_t>& posv, uint32_t& T, uint32_t& wip, uint32_t& nextw, uint32_t& flags) {
    const uint32_t v = pm_peek(i);
    sq_get(slot, posv, T, wip, nextw, flags);
    return v;
  }
  void sq_get(uint32_t slot, V<uint32_t>& posv, uint32_t& T, uint32_t& wip, uint32_t& nextw, uint32_t& flags) {
    if (slot >= kScanSlots) { oob = true; return; }
    const uint8_t* e = pmb + kMailBytes + slot * kScanBytes;
    for (int l = 0; l < 64; l++) memcpy(&posv.v[l], e + 4 * l, 4);
    uint32_t h[4]; memcpy(h, e + 256, 16);
    T = h[0]; wip = h[1]; nextw = h[2]; flags = h[3];
  }
  void pm_nap(uint32_t = 7u) { pair_naps++; std::this_thread::yield(); }
  void pm_idle() { std::this_thread::yield(); }
  static const uint8_t* pm_ptr(uint32_t lo, uint32_t hi) { return (const uint8_t*)(uintptr_t)(((uint64_t)hi << 32) | lo); }
  static inline std::atomic<uint64_t> pair_naps{0}, pair_entries{0};
  struct WPiece { uint8_t b[64][4]; uint32_t dl[64]; };
  uint32_t wv_delta(int l, uint32_t w) const { return l == 0 ? 0u : 4u * (uint32_t)l - (w & 3u); }
  WPiece wv_get(uint8_t* base, uint32_t mask, uint32_t sp, uint32_t w) {
    WPiece v; memset(&v, 0, sizeof v);
    for (int l = 0; l < GL; l++) {
      v.dl[l] = wv_delta(l, w);
      const uint32_t a = sp + v.dl[l], idx = a & mask;
      if (rl_ok(base + (idx & ~3u), 8)) memcpy(v.b[l], base + idx, 4);   // (the device reads the two aligned dwords around the index)
    }
    return v;
  }
  WPiece wv_get_stream(uint32_t sp, uint32_t w) { wave_trips++; return wv_get(wsb, kWs - 1u, sp, w); }
  WPiece wv_get_ring(uint32_t sw, uint32_t w) { return wv_get(wrb, kWv - 1u, sw, w); }
  WPiece wv_get_mem(const uint8_t* m, uint32_t w) {
    wave_far++;
    WPiece v; memset(&v, 0, sizeof v);
    for (int l = 0; l < GL; l++) { v.dl[l] = wv_delta(l, w); if (rd_ok(m + v.dl[l], 4)) memcpy(v.b[l], m + v.dl[l], 4); }
    return v;
  }
  void wv_put(uint32_t w, const WPiece& c) {
    const uint32_t u = (w + 4u) & (kWv - 1u);
    const bool ends = (u < 20u) | (u > kWv - 264u);    // the device's wave-uniform test in front of the mirror stores
    for (int l = 0; l < GL; l++) {
      if (c.dl[l] != wv_delta(l, w)) oob = true;       // the piece was shaped for another ring index
      const uint32_t t = (w + 4u + c.dl[l]) & (kWv - 1u);
      uint8_t* a = (wrb - 4) + t;
      if (l != 0 && ((uintptr_t)(a - wrb) & 3u)) oob = true;   // lanes 1.. store aligned dwords
      if (rl_ok(a, 4)) memcpy(a, c.b[l], 4);
      if (t < 20u) {
        wave_mirror++;
        if (!ends) oob = true;                        // a mirror store the device would have skipped
        if (rl_ok(a + kWv, 4)) memcpy(a + kWv, c.b[l], 4);
      }
    }
  }
  // ---- lane-parallel side (the parallel trips): per-lane values are 64-element vectors, every call is one instruction ----
  typedef V<uint32_t> VU;
  typedef V<bool> VB;
  static inline uint64_t par_trips = 0, par_seqs = 0, par_single = 0, par_far = 0, par_rounds = 0, par_windows = 0;
  static VU vlane() { VU r; for (int i = 0; i < 64; i++) r.v[i] = (uint32_t)i; return r; }
  static VU vsel(const VB& c, const VU& a, const VU& b) { VU r; for (int i = 0; i < 64; i++) r.v[i] = c.v[i] ? a.v[i] : b.v[i]; return r; }
  static VU valignbyte(const VU& hi, const VU& lo, const VU& sh) { VU r; for (int l = 0; l < 64; l++) { const uint64_t x = ((uint64_t)hi.v[l] << 32) | lo.v[l]; r.v[l] = (uint32_t)(x >> (8u * (sh.v[l] & 3u))); } return r; }
  static VB vlanes(uint64_t m) { VB r; for (int i = 0; i < 64; i++) r.v[i] = (m >> i) & 1u; return r; }
  static uint64_t vballot(const VB& b) { uint64_t m = 0; for (int i = 0; i < 64; i++) if (b.v[i]) m |= 1ull << i; return m; }
  static uint32_t vreadlane(const VU& v, uint32_t i) { return v.v[i & 63u]; }
  static VU vwritelane(const VU& v, uint32_t s, uint32_t i) { VU r = v; r.v[i & 63u] = s; return r; }
  static VU vshfl(const VU& v, const VU& srcl) { VU r; for (int i = 0; i < 64; i++) r.v[i] = v.v[srcl.v[i] & 63u]; return r; }
  static void vwalk(const VU& nx, VU& posv, uint32_t& T) {   // (group_dev.h vwalk, hand-written there: the same hops in the same groups)
    uint32_t s = 0u;
    auto hop = [&]() {
      posv = vwritelane(posv, s, T);
      T++;
      const uint32_t d = vreadlane(nx, s >> 2);
      s = (d >> ((s & 3u) * 8u)) & 255u;
    };
    bool done = false;
    while (T <= 60u) {
      hop(); hop(); hop(); hop();
      if (s == 255u) { done = true; break; }
    }
    if (!done) while ((s != 255u) & (T < 64u)) hop();
    for (uint32_t k = 0; k < 64u; k++) if (posv.v[k] == 255u) { if (k < T) T = k; break; }
  }
  // group_dev.h vwalk_par: the same starts by pointer doubling -- the device's levels and gathers, and the result compared with the plain
  // definition of the walk (a mismatch counts as a failure of the simulated wavefront: walk_mismatch, folded into every decode's verdict)
  static inline std::atomic<uint64_t> walk_mismatch{0}, walk_par_calls{0}, short_rounds{0};
  static uint32_t vgather8(const VU& tab, uint32_t q) { return (tab.v[(q >> 2) & 63u] >> ((q & 3u) * 8u)) & 255u; }
  static void vwalk_par(const VU& nx, const VU& lane, VU& posv, uint32_t& T) {
    walk_par_calls++;
    VU J = nx, pos;
    for (int l = 0; l < 64; l++) pos.v[l] = (lane.v[l] & 1u) ? vgather8(J, 0u) : 0u;
    for (uint32_t k = 1; k < 6u; k++) {
      VU Jn;
      for (int l = 0; l < 64; l++) {
        uint32_t w = 0;
        for (uint32_t j = 0; j < 4u; j++) w |= vgather8(J, (J.v[l] >> (8u * j)) & 255u) << (8u * j);
        Jn.v[l] = w;
      }
      J = Jn;
      for (int l = 0; l < 64; l++) if ((lane.v[l] >> k) & 1u) pos.v[l] = vgather8(J, pos.v[l]);
    }
    uint32_t Tn = 64u;
    for (uint32_t l = 0; l < 64u; l++) if (pos.v[l] == 255u) { Tn = l; break; }
    VU pref = posv; uint32_t Tref = T;
    vwalk(nx, pref, Tref);                       // the definition
    bool same = Tn == Tref;
    for (uint32_t l = 0; same && l < Tref; l++) same = pos.v[l] == pref.v[l];
    if (!same) walk_mismatch++;
    posv = pos; T = Tn;
  }
  static VU vexcl_scan(const VU& a) { par_trips++; VU r; uint32_t acc = 0; for (int l = 0; l < 64; l++) { r.v[l] = acc; acc += a.v[l]; } return r; }
  static inline uint64_t why[8] = {0};
  void vnote(const VB& act, const VB& a, const VB& b, const VB& c, const VB& d, const VB& e, const VB& f, const VB& ok) {
    int k = 0; while (k < 64 && ok.v[k]) k++;
    if (k == 64) return;
    if (!act.v[k]) why[0]++; else if (!a.v[k]) why[1]++; else if (!b.v[k]) why[2]++; else if (!c.v[k]) why[3]++; else if (!d.v[k]) why[4]++; else if (!e.v[k]) why[5]++; else if (!f.v[k]) why[6]++; else why[7]++;
  }
  static VU vmin(const VU& a, const VU& b) { VU r; for (int l = 0; l < 64; l++) r.v[l] = a.v[l] < b.v[l] ? a.v[l] : b.v[l]; return r; }
  void vs_win(uint32_t ip, VU& lo, VU& hi) {   // (the device reads the three aligned dwords at ((ip & ~3) + 4 l) & mask)
    par_windows++;
    for (int l = 0; l < 64; l++) {
      const uint32_t a = ((ip & ~3u) + 4u * (uint32_t)l) & (kWs - 1u);
      uint8_t b[12] = {0};
      if (rl_ok(wsb + a, 12)) memcpy(b, wsb + a, 12);
      memcpy(&lo.v[l], b + (ip & 3u), 4); memcpy(&hi.v[l], b + (ip & 3u) + 4, 4);
    }
  }
  VU vs_ld32(const VU& p) {
    VU r;
    for (int l = 0; l < 64; l++) { const uint32_t idx = p.v[l] & (kWs - 1u); if (rl_ok(wsb + (idx & ~3u), 8)) memcpy(&r.v[l], wsb + idx, 4); }
    return r;
  }
  // one store instruction of n bytes per active lane, with the device backend's mirror rule; all lanes' data were read before
  void vput(uint32_t n, const VU& w, const uint8_t (*v)[16], const VB& m) {
    for (int l = 0; l < 64; l++) {
      if (!m.v[l]) continue;
      const uint32_t t = (w.v[l] + 16u) & (kWv - 1u);
      uint8_t* a = (wrb - 16) + t;
      if (rl_ok(a, n)) memcpy(a, v[l], n);
      if (t < 32u && rl_ok(a + kWv, n)) memcpy(a + kWv, v[l], n);
    }
  }
  // src 0: stream ring, 1: output ring, 2: memory (mem + sp)
  void vcopy(int src, const VU& dw, const VU& sp, const VU& len, const VB& go, const uint8_t* mem = nullptr) {
    const uint8_t* sb = src == 0 ? wsb : src == 1 ? wrb : mem;
    const uint32_t sm = src == 0 ? kWs - 1u : src == 1 ? kWv - 1u : 0xFFFFFFFFu;
    uint8_t v[64][16];
    auto step = [&](uint32_t n, const VU& c, const VB& m) {
      memset(v, 0, sizeof v);
      for (int l = 0; l < 64; l++) if (m.v[l]) { const uint8_t* q = sb + ((sp.v[l] + c.v[l]) & sm); if (src == 2 ? rd_ok(q, n) : rl_ok(q, n)) memcpy(v[l], q, n); }
      vput(n, dw + c, v, m);
    };
    for (uint32_t c = 0;; c += 16u) {
      VB m; bool any = false;
      for (int l = 0; l < 64; l++) { m.v[l] = go.v[l] && c + 16u <= len.v[l]; any |= m.v[l]; }
      if (!any) break;
      step(16u, VU(c), m);
    }
    VU c;
    for (int l = 0; l < 64; l++) c.v[l] = len.v[l] & ~15u;
    for (uint32_t n = 8u; n >= 1u; n >>= 1) {
      VB m;
      for (int l = 0; l < 64; l++) m.v[l] = go.v[l] && (len.v[l] & n) != 0u;
      step(n, c, m);
      for (int l = 0; l < 64; l++) if (m.v[l]) c.v[l] += n;
    }
  }
  // one run per lane (group_dev.h vcopy_run): the usual round reads EVERYTHING (the sources of all its lanes) before it stores
  // anything, and stores no mirror copies -- the rule that decides it is the device's
  uint64_t vodd_mask(const VU& dw, const VU& len) const {
    uint64_t m = 0;
    for (int l = 0; l < 64; l++) { const uint32_t x = dw.v[l] & (kWv - 1u); if (len.v[l] > 64u || x < 16u || x + len.v[l] + 16u > kWv) m |= 1ull << l; }
    return m;
  }
  static uint32_t vrun_tier(const VU& len, uint64_t actm) {
    uint32_t t = 0u;
    for (int l = 0; l < 64; l++) if ((actm >> l) & 1ull) { if (len.v[l] >= 32u) return 4u; if (len.v[l] >= 16u) t = 1u; }
    return t;
  }
  template <bool PRED = false>   // (the device's choice between reading a round's pieces in every active lane and only where the run has them: the same bytes)
  void vcopy_run(const VU& dw, const VB& from_stream, const VU& sp, const VU& len, uint64_t gom, const uint8_t* mem, const VU& mpos, uint64_t farm, uint64_t oddm,
                 uint32_t tier = 4u) {
    // (the device picks a round's form -- pieces of 16 bytes read and stored: none / one / four -- by the tier: a lane of the round whose run is
    // longer than its tier says would lose bytes there)
    for (int l = 0; l < 64; l++) if ((gom >> l) & 1ull) { if (len.v[l] >= 16u && tier == 0u) oob = true; if (len.v[l] >= 32u && tier != 4u) oob = true; }
    if (tier != 4u) short_rounds++;
    const VB go = vlanes(gom), far = vlanes(farm);
    if ((oddm ^ vodd_mask(dw, len)) & gom) oob = true;   // (the caller's mask must be the rule below, for the lanes of the round: the pair loop's copier gets the mask from the parser, and lanes 62 / 63 of its message are header words)
    par_rounds++;
    for (int l = 0; l < 64; l++) { par_seqs += (go.v[l] && !from_stream.v[l]) ? 1u : 0u; par_far += (go.v[l] && far.v[l]) ? 1u : 0u; }
    bool odd = false;
    for (int l = 0; l < 64; l++) {
      const uint32_t x = dw.v[l] & (kWv - 1u);
      odd |= go.v[l] && (len.v[l] > 64u || x < 16u || x + len.v[l] + 16u > kWv);
    }
    VB gs, gr, gf;
    for (int l = 0; l < 64; l++) { gs.v[l] = go.v[l] && from_stream.v[l]; gr.v[l] = go.v[l] && !from_stream.v[l] && !far.v[l]; gf.v[l] = go.v[l] && far.v[l]; }
    if (odd) { vcopy(0, dw, sp, len, gs); vcopy(1, dw, sp, len, gr); vcopy(2, dw, mpos, len, gf, mem); return; }
    static uint8_t bf[64][80];
    for (int l = 0; l < 64; l++) {
      if (!go.v[l]) continue;
      // (the device reads 64 + 15 bytes from the source whatever the length: every index it forms must lie in the LDS bytes / the block's slot)
      const uint8_t* base = from_stream.v[l] ? wsb : wrb;
      const uint32_t mask = from_stream.v[l] ? kWs - 1u : kWv - 1u;
      auto run = [&](const uint8_t* b, uint32_t m, uint32_t s0, bool is_mem) {
        for (uint32_t c = 0; c < 64u; c += 16u) { const uint8_t* q = b + ((s0 + c) & m); if (is_mem ? rd_ok(q, 16) : rl_ok(q, 16)) memcpy(bf[l] + c, q, 16); }
        uint32_t c = len.v[l] & ~15u;
        for (uint32_t n = 8u; n >= 1u; n >>= 1) { const uint8_t* q = b + ((s0 + c) & m); if ((is_mem ? rd_ok(q, n) : rl_ok(q, n)) && (len.v[l] & n)) memcpy(bf[l] + c, q, n); c += len.v[l] & n; }
      };
      run(base, mask, sp.v[l], false);
      if (far.v[l]) run(mem, 0xFFFFFFFFu, mpos.v[l], true);
    }
    for (int l = 0; l < 64; l++) {
      if (!go.v[l]) continue;
      uint8_t* a = wrb + (dw.v[l] & (kWv - 1u));
      if (rl_ok(a, len.v[l])) memcpy(a, bf[l], len.v[l]);
    }
  }
  LChunk wv_read_al(uint32_t fw) {
    if (fw & 255u) oob = true;
    LChunk v; memset(&v, 0, sizeof v);
    for (int l = 0; l < GL; l++) { const uint8_t* q = wrb + ((fw + 4u * (uint32_t)l) & (kWv - 1u)); if (rl_ok(q, 4)) memcpy(v.b[l], q, 4); }
    return v;
  }
  void wv_store(uint8_t* dst, uint32_t fw, const LChunk& c, uint32_t lo, uint32_t hi) {
    for (int l = 0; l < GL; l++)
      for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t a = fw + 4u * (uint32_t)l + k;
        if (a >= lo && a < hi) { uint8_t* p = dst + (intptr_t)(int32_t)(a - wdb); if (wr_ok(p, 1)) *p = c.b[l][k]; }
      }
  }

  void copy_match(uint8_t* dst, uint32_t op, uint32_t offset, uint32_t len, bool wild) {
    uint8_t* d = dst + op;
    const uint8_t* m = d - offset;
    if (wild && offset >= 4u * GL) {
      for (uint32_t base = 0; base < len; base += 4u * GL) {
        uint32_t v[64]; bool act[64];
        for (int l = 0; l < GL; l++) { uint32_t i = base + 4u * l; act[l] = i < len; v[l] = 0; if (act[l] && rd_ok(m + i, 4)) memcpy(&v[l], m + i, 4); }
        for (int l = 0; l < GL; l++) { uint32_t i = base + 4u * l; if (act[l] && wr_ok(d + i, 4)) memcpy(d + i, &v[l], 4); }
      }
    } else if (offset == 0) {
      for (uint32_t i = 0; i < len; i++) if (wr_ok(d + i, 1)) d[i] = 0;
    } else {
      uint32_t r[64];
      for (int l = 0; l < GL; l++) r[l] = (uint32_t)l < offset ? (uint32_t)l : (uint32_t)l % offset;
      const uint32_t stp = (uint32_t)GL < offset ? (uint32_t)GL : (uint32_t)GL % offset;
      for (uint32_t base = 0; base < len; base += GL) {
        uint8_t v[64]; bool act[64];
        for (int l = 0; l < GL; l++) { uint32_t i = base + l; act[l] = i < len; v[l] = 0; if (act[l] && rd_ok(m + r[l], 1)) v[l] = m[r[l]]; }
        for (int l = 0; l < GL; l++) {
          uint32_t i = base + l;
          if (act[l] && wr_ok(d + i, 1)) d[i] = v[l];
          r[l] += stp; if (r[l] >= offset) r[l] -= offset;
        }
      }
    }
  }
};

}  // namespace hostsim

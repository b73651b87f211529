// tests/hostsim/hostsim.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the product's algorithm cores (lz4-java_amd/csrc/*_core.h) against the lock-step host
// backends so `pytest -m "not gpu"` can check the ALGORITHMS the gfx950 kernels run against the
// oracle.  Nothing here is linked into liblz4hip.so; the product has no CPU path.
#include <stdint.h>
#include <stdlib.h>
#include "../../lz4-java_amd/csrc/lz4_fast_core.h"
#include "../../lz4-java_amd/csrc/lz4_fast_ms_core.h"
#include "../../lz4-java_amd/csrc/lz4_fast_v2_core.h"
#include "wave_host.h"
#include "../../lz4-java_amd/csrc/lz4_decode_core.h"
#include "group_host.h"
#include "../../lz4-java_amd/csrc/lz4_hc_core.h"
#define LZ4HIP_MAIL_RING 2   /* two slots: every third post wraps and waits for the writer */
#include "../../lz4-java_amd/csrc/mail_ring.h"
#include <atomic>
#include <thread>
#include <vector>

// ---- the finder / writer ring of the default compress kernel (csrc/mail_ring.h) with HOST threads: one thread plays the finder
// wavefront, one the writer wavefront of a pair; several pairs share the block queue like the wavefronts of a workgroup do.
namespace hostsim {
struct SimBatch {   // the fields mail_writer_t reads of kernels.h BatchArgs
  const uint8_t* src; const uint64_t* src_off; const int32_t* src_len;
  uint8_t* dst; const uint64_t* dst_off; const int32_t* dst_cap;
  int32_t* out; uint32_t n;
};
struct MailHost {
  static thread_local uint64_t rng;     // naps of random length shake the interleavings
  static std::atomic<int> oob;
  static void jitter() {
    rng = rng * 6364136223846793005ull + 1442695040888963407ull;
    const int k = (int)((rng >> 33) & 7u);
    if (k == 0) std::this_thread::sleep_for(std::chrono::microseconds(20));
    else if (k < 4) std::this_thread::yield();
  }
  static uint32_t peek(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
  static uint32_t peek_far(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
  static void acquire() { __atomic_thread_fence(__ATOMIC_ACQUIRE); }
  static void publish(uint32_t* p, uint32_t v) { jitter(); __atomic_thread_fence(__ATOMIC_RELEASE); __atomic_store_n(p, v, __ATOMIC_RELAXED); }
  static void nap_finder() { jitter(); }
  static void nap_writer() { jitter(); }
  template <class T> static T* uptr(T* q) { return q; }
  static int32_t u32(int32_t v) { return v; }
  static void block_begin(WaveHost& w, const uint8_t* s, uint32_t n, uint8_t* d, uint32_t cap) { if (w.oob) oob = 1; w.bounds(s, n, d, cap); }
  static void result(int32_t* out, uint32_t b, int32_t r) { out[b] = r; }
};
thread_local uint64_t MailHost::rng = 1;
std::atomic<int> MailHost::oob{0};
}  // namespace hostsim

extern "C" {

// returns compressed size (0 = does not fit), or -1000 if the simulated wave touched memory
// outside [src, src+n) / [dst, dst+cap)
int sim_compress_fast(const uint8_t* src, int n, uint8_t* dst, int cap, uint64_t* stats4, uint64_t rng_seed) {
  if (n < 0 || (uint32_t)n > 0x7E000000u || cap < 0) return 0;
  hostsim::WaveHost w;
  if (rng_seed) w.rng = rng_seed;
  w.bounds(src, (size_t)n, dst, (size_t)cap);
  lz4hip::FastStats st = {0, 0, 0, 0};
  uint32_t r;
  lz4hip::DirectOut<hostsim::WaveHost> out(w, src, (uint32_t)n, dst, (uint32_t)cap);
  if (n < 65547) {
    lz4hip::FastCore<hostsim::WaveHost, true> c(w, out, src, (uint32_t)n, &st);
    r = c.run();
  } else {
    lz4hip::FastCore<hostsim::WaveHost, false> c(w, out, src, (uint32_t)n, &st);
    r = c.run();
  }
  if (stats4) { stats4[0] = st.steps; stats4[1] = st.slow_steps; stats4[2] = st.false_pos; stats4[3] = st.sequences; }
  if (w.oob) return -1000;
  return (int)r;
}

// one-sequence-per-step core with the density probe of the adaptive scheme: -2 = the core left the block to the
// window-parallel core (sequences 32..95 cover fewer than dense64 bytes)
int sim_compress_fast_probe(const uint8_t* src, int n, uint8_t* dst, int cap, uint32_t dense64) {
  if (n < 0 || (uint32_t)n > 0x7E000000u || cap < 0) return 0;
  hostsim::WaveHost w;
  w.bounds(src, (size_t)n, dst, (size_t)cap);
  lz4hip::DirectOut<hostsim::WaveHost> out(w, src, (uint32_t)n, dst, (uint32_t)cap);
  uint32_t r;
  bool bailed;
  if (n < 65547) { lz4hip::FastCore<hostsim::WaveHost, true> c(w, out, src, (uint32_t)n); c.dense64 = dense64; r = c.run(); bailed = c.bailed; }
  else { lz4hip::FastCore<hostsim::WaveHost, false> c(w, out, src, (uint32_t)n); c.dense64 = dense64; r = c.run(); bailed = c.bailed; }
  if (w.oob) return -1000;
  return bailed ? -2 : (int)r;
}

// lean core + parked/batched emission (lz4_fast_v2_core.h); blocks >= 65547 bytes: the generic loop over the same output policy
int sim_compress_fast_v2(const uint8_t* src, int n, uint8_t* dst, int cap, uint64_t* stats4, uint64_t rng_seed) {
  if (n < 0 || (uint32_t)n > 0x7E000000u || cap < 0) return 0;
  hostsim::WaveHost w;
  if (rng_seed) w.rng = rng_seed;
  w.bounds(src, (size_t)n, dst, (size_t)cap);
  lz4hip::FastStats st = {0, 0, 0, 0};
  uint32_t r;
  lz4hip::ParkOut<hostsim::WaveHost> out(w, src, (uint32_t)n, dst, (uint32_t)cap);
  if (n < 65547) {
    lz4hip::FastV2<hostsim::WaveHost> c(w, out, src, (uint32_t)n, &st);
    r = c.run();
  } else {
    lz4hip::FastV2<hostsim::WaveHost, lz4hip::ParkOut<hostsim::WaveHost>, false> c(w, out, src, (uint32_t)n, &st);
    r = c.run();
  }
  if (stats4) { stats4[0] = st.steps; stats4[1] = st.slow_steps; stats4[2] = st.false_pos; stats4[3] = st.sequences; }
  if (w.oob) return -1000;
  return (int)r;
}

// lean core with RAW parking (bare hits parked, liblz4's backward extension done 64 hits at a time at write time): the writer
// wavefront's half of the two-wave kernel, in one wave
int sim_compress_fast_v2raw(const uint8_t* src, int n, uint8_t* dst, int cap, uint64_t* stats4, uint64_t seed) {
  if (n < 0 || (uint32_t)n > 0x7E000000u || cap < 0) return 0;
  hostsim::WaveHost w;
  if (seed) w.rng = seed;
  w.bounds(src, (size_t)n, dst, (size_t)cap);
  lz4hip::FastStats st{};
  uint32_t r;
  lz4hip::ParkOutRaw<hostsim::WaveHost> out(w, src, (uint32_t)n, dst, (uint32_t)cap);
  if (n < 65547) {
    lz4hip::FastV2<hostsim::WaveHost, lz4hip::ParkOutRaw<hostsim::WaveHost>> c(w, out, src, (uint32_t)n, &st);
    r = c.run();
  } else {
    lz4hip::FastV2<hostsim::WaveHost, lz4hip::ParkOutRaw<hostsim::WaveHost>, false> c(w, out, src, (uint32_t)n, &st);
    r = c.run();
  }
  if (stats4) { stats4[0] = st.steps; stats4[1] = st.slow_steps; stats4[2] = st.false_pos; stats4[3] = st.sequences; }
  if (w.oob) return -1000;
  return (int)r;
}

// the same with the compact byU32 entries ({position 22 bits, fingerprint 10 bits}: blocks of 65547 bytes .. 4 MiB) -- what the
// eight-chain kernel of such batches runs; other sizes: -3
int sim_compress_fast_v2pk(const uint8_t* src, int n, uint8_t* dst, int cap, uint64_t* stats4, uint64_t seed) {
  if (n < 65547 || n > (1 << 22) || cap < 0) return -3;
  hostsim::WaveHost w;
  if (seed) w.rng = seed;
  w.bounds(src, (size_t)n, dst, (size_t)cap);
  lz4hip::FastStats st{};
  lz4hip::ParkOutRaw<hostsim::WaveHost> out(w, src, (uint32_t)n, dst, (uint32_t)cap);
  lz4hip::FastV2<hostsim::WaveHost, lz4hip::ParkOutRaw<hostsim::WaveHost>, false, true> c(w, out, src, (uint32_t)n, &st);
  const uint32_t r = c.run();
  if (stats4) { stats4[0] = st.steps; stats4[1] = st.slow_steps; stats4[2] = st.false_pos; stats4[3] = st.sequences; }
  if (w.oob) return -1000;
  return (int)r;
}

// lean core with the density probe of the adaptive scheme: -2 = left to the window-parallel core
int sim_compress_fast_v2_probe(const uint8_t* src, int n, uint8_t* dst, int cap, uint32_t dense64) {
  if (n < 0 || (uint32_t)n > 0x7E000000u || cap < 0) return 0;
  hostsim::WaveHost w;
  w.bounds(src, (size_t)n, dst, (size_t)cap);
  lz4hip::ParkOut<hostsim::WaveHost> out(w, src, (uint32_t)n, dst, (uint32_t)cap);
  out.dense64 = dense64;
  uint32_t r;
  if (n < 65547) { lz4hip::FastV2<hostsim::WaveHost> c(w, out, src, (uint32_t)n); r = c.run(); }
  else { lz4hip::FastV2<hostsim::WaveHost, lz4hip::ParkOut<hostsim::WaveHost>, false> c(w, out, src, (uint32_t)n); r = c.run(); }
  if (w.oob) return -1000;
  return out.bail ? -2 : (int)r;
}

// window-parallel core (lz4_fast_ms_core.h): every sequence of a 64-position window per step
int sim_compress_fast_ms(const uint8_t* src, int n, uint8_t* dst, int cap, uint64_t* stats4, uint64_t rng_seed) {
  if (n < 0 || (uint32_t)n > 0x7E000000u || cap < 0) return 0;
  hostsim::WaveHost w;
  if (rng_seed) w.rng = rng_seed;
  w.bounds(src, (size_t)n, dst, (size_t)cap);
  lz4hip::FastStats st = {0, 0, 0, 0};
  uint32_t r;
  lz4hip::DirectOut<hostsim::WaveHost> out(w, src, (uint32_t)n, dst, (uint32_t)cap);
  if (n < 65547) {
    lz4hip::FastCoreMS<hostsim::WaveHost, true> c(w, out, src, (uint32_t)n, &st);
    r = c.run();
  } else {
    lz4hip::FastCoreMS<hostsim::WaveHost, false> c(w, out, src, (uint32_t)n, &st);
    r = c.run();
  }
  if (stats4) { stats4[0] = st.steps; stats4[1] = st.slow_steps; stats4[2] = st.false_pos; stats4[3] = st.sequences; }
  if (w.oob) return -1000;
  return (int)r;
}

void sim_ring_stats(unsigned long long* out3) { out3[0] = hostsim::GroupHost::ring_trips; out3[1] = hostsim::GroupHost::ring_entries; out3[2] = hostsim::GroupHost::ring_repl; }
unsigned long long sim_ring_trips() { return hostsim::GroupHost::ring_trips; }   // offset words the ring loop has parsed so far
void sim_wave_why(unsigned long long* out8) { for (int i = 0; i < 8; i++) out8[i] = hostsim::GroupHost::why[i]; }
void sim_wave_par_stats(unsigned long long* out3) { out3[0] = hostsim::GroupHost::par_trips; out3[1] = hostsim::GroupHost::par_seqs; out3[2] = hostsim::GroupHost::par_far; }
void sim_wave_par_rounds(unsigned long long* out2) { out2[0] = hostsim::GroupHost::par_rounds; out2[1] = hostsim::GroupHost::par_windows; }
void sim_wave_stats(unsigned long long* out4) { out4[0] = hostsim::GroupHost::wave_trips; out4[1] = hostsim::GroupHost::wave_entries; out4[2] = hostsim::GroupHost::wave_far; out4[3] = hostsim::GroupHost::wave_mirror; }
unsigned long long sim_pair_naps() { return hostsim::GroupHost::pair_naps.load(); }
unsigned long long sim_deep_trips() { return hostsim::GroupHost::deep_trips; }   // offset words the deep decoder loop has parsed so far

// safe != 0: (src_size = compressed length) -> decoded size; safe == 0: (src_size = readable
// capacity) -> bytes consumed.  `gl` = lanes per group (4..64).  -1000000 = out-of-slot access.
int sim_decompress(const uint8_t* src, int src_size, uint8_t* dst, int out_size, int safe, int gl) {
  const bool pipe = (gl & 0x100) != 0;   // bit 8 of gl: the pipelined interior loop
  const bool stage = (gl & 0x200) != 0;  // bit 9: output staging
  const bool deep = (gl & 0x400) != 0;   // bit 10: the deep interior loop (lz4_decode_deep.h; groups of up to 16 lanes)
  const bool ring = (gl & 0x800) != 0;   // bit 11: the ring loop (lz4_decode_ring.h); bits 12..15: log2 of the output ring's bytes (default 512)
  const int ring_log = (gl >> 12) & 15;
  const bool wave = (gl & 0x10000) != 0; // bit 16: the wave loop (lz4_decode_wave.h; 64 lanes); bits 17..21: log2 of its output ring's bytes (default 8 KB), bit 22: a 1 KB stream ring
  const int wave_log = (gl >> 17) & 31;
  const bool wave_ks1k = (gl & 0x400000) != 0;
  const int gl0 = gl;
  gl &= 0xFF;
  hostsim::GroupHost g(gl, src, src_size, dst, out_size);
  if (ring_log) g.kRing = 1u << ring_log;
  if (wave_log) g.kWv = 1u << wave_log;
  if (wave_ks1k) g.kWs = 1024u;
  int r;
  const bool wave_par = (gl0 & 0x800000) != 0;   // bit 23: the parallel wave loop (several sequences of the block per trip)
  if (wave && (gl0 & 0x2000000) != 0) {          // bit 25: the TRIO loop (lz4_decode_trio.h): copier, planner and scanner wavefronts = three host threads over one block of "LDS"
    hostsim::GroupHost gp(gl, src, src_size, dst, out_size), gs(gl, src, src_size, dst, out_size);
    gp.kWv = gs.kWv = g.kWv; gp.kWs = gs.kWs = g.kWs;
    std::vector<uint64_t> lds((g.trio_lds_bytes() + 7u) / 8u, 0xEEEEEEEEEEEEEEEEull);
    g.pair_lds = gp.pair_lds = gs.pair_lds = (uint8_t*)lds.data();
    memset(g.pair_lds + g.pair_lds_bytes() - 64u, 0, 64);   // the control words
    static std::atomic<uint64_t> seed3{0xC2B2AE3D27D4EB4Full};
    g.nap_rng = seed3.fetch_add(0xD1B54A32D192ED03ull) | 1u; gp.nap_rng = g.nap_rng * 0x2545F4914F6CDD1Dull | 1u; gs.nap_rng = gp.nap_rng * 0x9E3779B97F4A7C15ull | 1u;
    std::thread planner([&] { lz4hip::trio_service(gp, gp.pair_lds, false); });
    std::thread scanner([&] { lz4hip::trio_service(gs, gs.pair_lds, true); });
    r = safe ? lz4hip::decode_block<hostsim::GroupHost, true, 8>(g, src, src_size, dst, out_size, g.pair_lds)
             : lz4hip::decode_block<hostsim::GroupHost, false, 8>(g, src, src_size, dst, out_size, g.pair_lds);
    lz4hip::trio_quit(g, g.pair_lds);
    planner.join(); scanner.join();
    if (g.oob || gp.oob || gs.oob || hostsim::GroupHost::walk_mismatch.load() != 0) return -1000000;
    return r;
  }
  if (wave && (gl0 & 0x1000000) != 0) {          // bit 24: the PAIR loop (lz4_decode_pair.h): a copier and a parser wavefront = two host threads over one block of "LDS"
    hostsim::GroupHost gp(gl, src, src_size, dst, out_size);
    gp.kWv = g.kWv; gp.kWs = g.kWs;
    std::vector<uint64_t> lds((g.pair_lds_bytes() + 7u) / 8u, 0xEEEEEEEEEEEEEEEEull);
    g.pair_lds = gp.pair_lds = (uint8_t*)lds.data();
    memset(g.pair_lds + g.pair_lds_bytes() - 64u, 0, 64);   // the control words (the kernel zeroes them in front of its barrier)
    static std::atomic<uint64_t> seed{0x9E3779B97F4A7C15ull};
    g.nap_rng = seed.fetch_add(0xD1B54A32D192ED03ull) | 1u; gp.nap_rng = g.nap_rng * 0x2545F4914F6CDD1Dull | 1u;
    std::thread parser([&] { lz4hip::pair_parser_service(gp, gp.pair_lds); });
    r = safe ? lz4hip::decode_block<hostsim::GroupHost, true, 7>(g, src, src_size, dst, out_size, g.pair_lds)
             : lz4hip::decode_block<hostsim::GroupHost, false, 7>(g, src, src_size, dst, out_size, g.pair_lds);
    lz4hip::pair_parser_quit(g, g.pair_lds);
    parser.join();
    if (g.oob || gp.oob) return -1000000;
    return r;
  }
  if (wave && wave_par && wave_ks1k) r = safe ? lz4hip::decode_block<hostsim::GroupHost, true, 6>(g, src, src_size, dst, out_size, g.stg_buf)
                                              : lz4hip::decode_block<hostsim::GroupHost, false, 6>(g, src, src_size, dst, out_size, g.stg_buf);
  else if (wave && wave_par) r = safe ? lz4hip::decode_block<hostsim::GroupHost, true, 5>(g, src, src_size, dst, out_size, g.stg_buf)
                                 : lz4hip::decode_block<hostsim::GroupHost, false, 5>(g, src, src_size, dst, out_size, g.stg_buf);
  else if (wave) r = safe ? lz4hip::decode_block<hostsim::GroupHost, true, 4>(g, src, src_size, dst, out_size, g.stg_buf)
                     : lz4hip::decode_block<hostsim::GroupHost, false, 4>(g, src, src_size, dst, out_size, g.stg_buf);
  else if (ring) r = safe ? lz4hip::decode_block<hostsim::GroupHost, true, 3>(g, src, src_size, dst, out_size, g.stg_buf)
                     : lz4hip::decode_block<hostsim::GroupHost, false, 3>(g, src, src_size, dst, out_size, g.stg_buf);
  else if (deep) r = safe ? lz4hip::decode_block<hostsim::GroupHost, true, 2>(g, src, src_size, dst, out_size, g.stg_buf)
                     : lz4hip::decode_block<hostsim::GroupHost, false, 2>(g, src, src_size, dst, out_size, g.stg_buf);
  else if (stage) r = safe ? lz4hip::decode_block<hostsim::GroupHost, true, 0, true>(g, src, src_size, dst, out_size, g.stg_buf)
                      : lz4hip::decode_block<hostsim::GroupHost, false, 0, true>(g, src, src_size, dst, out_size, g.stg_buf);
  else if (pipe) r = safe ? lz4hip::decode_block<hostsim::GroupHost, true, 1>(g, src, src_size, dst, out_size)
                     : lz4hip::decode_block<hostsim::GroupHost, false, 1>(g, src, src_size, dst, out_size);
  else r = safe ? lz4hip::decode_block<hostsim::GroupHost, true>(g, src, src_size, dst, out_size)
                : lz4hip::decode_block<hostsim::GroupHost, false>(g, src, src_size, dst, out_size);
  if (g.oob || hostsim::GroupHost::walk_mismatch.load() != 0) return -1000000;
  return r;
}
unsigned long long sim_walk_par_calls() { return hostsim::GroupHost::walk_par_calls.load(); }
unsigned long long sim_short_rounds() { return hostsim::GroupHost::short_rounds.load(); }   // copy rounds in a short form (the wave loop's SHORT instance)

// LZ4 HC (levels 1..12): phase 1 (delta[] build) + phase 2 (lazy parse, or the optimal parser for 10..12) in the lock-step
// simulator.  returns the compressed size, 0 (does not fit) or -1000 (out-of-slot access)
int sim_compress_hc(const uint8_t* src, int n, uint8_t* dst, int cap, int level, uint64_t rng_seed) {
  if (n < 0 || (uint32_t)n > 0x7E000000u || cap < 0) return 0;
  if (level < 1) level = 9;
  if (level > 12) level = 12;
  hostsim::WaveHost w;
  if (rng_seed) w.rng = rng_seed;
  w.bounds(src, (size_t)n, dst, (size_t)cap);
  std::vector<uint16_t> delta((size_t)n + 8, 0xFFFF);
  lz4hip::HcBuild<hostsim::WaveHost>::run(w, src, (uint32_t)n, delta.data());
  lz4hip::HcParse<hostsim::WaveHost> p(w, src, n, delta.data(), dst, cap, level);
  std::vector<int> opt(level >= 10 ? (size_t)lz4hip::HC_OPT_INTS : 1u, 0x55555555);
  const int r = level >= 10 ? p.run_opt(level, opt.data()) : p.run();
  if (w.oob) return -1000;
  return r;
}

// The two-wave kernel's shape on the host: `pairs` finder threads draw blocks from one queue (the body of
// compress_fast_v2w_cu_kernel: lean core for n < 65547, the exact core over the same output policy otherwise; a block the density
// probe rejects is an ABORT message and an entry of routed[]), each with its own writer thread behind a two-slot ring.
// out[b] = compressed size (0 = does not fit), -2 = routed (listed in routed[0 .. *n_routed)).  returns 0, or -1000 on an
// out-of-slot access of any simulated wavefront.
int sim_mail_ring(const uint8_t* src, const uint64_t* src_off, const int32_t* src_len, uint8_t* dst, const uint64_t* dst_off,
                  const int32_t* dst_cap, int32_t* out, uint32_t n, uint32_t pairs, uint32_t dense64, uint32_t* routed, uint32_t* n_routed,
                  uint64_t seed) {
  using W = hostsim::WaveHost;
  using M = hostsim::MailHost;
  using Out = lz4hip::MailOutT<W, M>;
  hostsim::SimBatch a{src, src_off, src_len, dst, dst_off, dst_cap, out, n};
  std::atomic<uint32_t> q{0}, nr{0};
  M::oob = 0;
  std::vector<std::vector<uint32_t>> slots(pairs, std::vector<uint32_t>(lz4hip::MAIL_RING * lz4hip::MAIL_SLOT_WORDS, 0xDEADBEEFu));
  std::vector<std::vector<uint32_t>> ctr(pairs, std::vector<uint32_t>(2, 0u));
  std::vector<std::thread> th;
  for (uint32_t p = 0; p < pairs; p++) {
    th.emplace_back([&, p] {   // the writer wavefront
      M::rng = seed * 977u + p * 2u + 1u;
      W w;
      lz4hip::mail_writer_t<W, M>(w, a, slots[p].data(), ctr[p].data());
      if (w.oob) M::oob = 1;
    });
    th.emplace_back([&, p] {   // the finder wavefront
      M::rng = seed * 977u + p * 2u + 2u;
      W w;
      w.rng = seed ? seed + p : w.rng;
      uint32_t head = 0, tail_seen = 0;
      for (;;) {
        const uint32_t b = q.fetch_add(1);
        Out o(w, slots[p].data(), ctr[p].data(), head);
        o.tail_seen = tail_seen;
        if (b >= a.n) { o.post(lz4hip::MAIL_EXIT, 0u, 0u); break; }
        o.b = b;
        const int32_t bn = a.src_len[b], cap = a.dst_cap[b];
        if (bn >= 0 && (uint32_t)bn <= 0x7E000000u && cap >= 0) {
          const uint8_t* s = a.src + a.src_off[b];
          w.bounds(s, (size_t)bn, nullptr, 0);      // (a finder never writes the output)
          o.dense64 = routed ? dense64 : 0u;
          if (bn < 65547) { lz4hip::FastV2<W, Out> c(w, o, s, (uint32_t)bn); (void)c.run(); }
          else { lz4hip::FastV2<W, Out, false> c(w, o, s, (uint32_t)bn); (void)c.run(); }
          if (o.bail) {
            o.post(lz4hip::MAIL_ABORT, 0u, 0u);
            routed[nr.fetch_add(1)] = b;
            a.out[b] = -2;
          }
        } else {
          a.out[b] = 0;
        }
        head = o.head; tail_seen = o.tail_seen;
      }
      if (w.oob) M::oob = 1;
    });
  }
  for (auto& t : th) t.join();
  if (n_routed) *n_routed = nr.load();
  return M::oob ? -1000 : 0;
}

// The ten-chain kernel's shape (compress_fast_v2wp_cu_kernel): `finders` finder threads, `writers` writer threads, writer j
// serving the rings of finder j and finder j + writers (mail_writer2_t).  Blocks of 65547 bytes .. 4 MiB run with the packed table
// entries that kernel uses, the others with the usual cores.  Results as sim_mail_ring.
int sim_mail_ring_shared(const uint8_t* src, const uint64_t* src_off, const int32_t* src_len, uint8_t* dst, const uint64_t* dst_off,
                         const int32_t* dst_cap, int32_t* out, uint32_t n, uint32_t finders, uint32_t writers, uint32_t dense64,
                         uint32_t* routed, uint32_t* n_routed, uint64_t seed) {
  using W = hostsim::WaveHost;
  using M = hostsim::MailHost;
  using Out = lz4hip::MailOutT<W, M>;
  if (writers == 0 || finders < writers || finders > 2 * writers) return -1;
  hostsim::SimBatch a{src, src_off, src_len, dst, dst_off, dst_cap, out, n};
  std::atomic<uint32_t> q{0}, nr{0};
  M::oob = 0;
  std::vector<std::vector<uint32_t>> slots(finders, std::vector<uint32_t>(lz4hip::MAIL_RING * lz4hip::MAIL_SLOT_WORDS, 0xDEADBEEFu));
  std::vector<std::vector<uint32_t>> ctr(finders, std::vector<uint32_t>(2, 0u));
  std::vector<std::thread> th;
  for (uint32_t j = 0; j < writers; j++)
    th.emplace_back([&, j] {
      M::rng = seed * 977u + j * 2u + 1u;
      W w;
      const uint32_t k = j + writers;
      lz4hip::mail_writer2_t<W, M>(w, a, slots[j].data(), ctr[j].data(), k < finders ? slots[k].data() : nullptr, k < finders ? ctr[k].data() : nullptr);
      if (w.oob) M::oob = 1;
    });
  for (uint32_t p = 0; p < finders; p++)
    th.emplace_back([&, p] {
      M::rng = seed * 977u + p * 2u + 2u;
      W w;
      w.rng = seed ? seed + p : w.rng;
      uint32_t head = 0, tail_seen = 0;
      for (;;) {
        const uint32_t b = q.fetch_add(1);
        Out o(w, slots[p].data(), ctr[p].data(), head);
        o.tail_seen = tail_seen;
        if (b >= a.n) { o.post(lz4hip::MAIL_EXIT, 0u, 0u); break; }
        o.b = b;
        const int32_t bn = a.src_len[b], cap = a.dst_cap[b];
        if (bn >= 0 && (uint32_t)bn <= 0x7E000000u && cap >= 0) {
          const uint8_t* s = a.src + a.src_off[b];
          w.bounds(s, (size_t)bn, nullptr, 0);
          o.dense64 = routed ? dense64 : 0u;
          if (bn < 65547) { lz4hip::FastV2<W, Out> c(w, o, s, (uint32_t)bn); (void)c.run(); }
          else if (bn <= (1 << 22)) { lz4hip::FastV2<W, Out, false, true> c(w, o, s, (uint32_t)bn); (void)c.run(); }
          else { lz4hip::FastV2<W, Out, false> c(w, o, s, (uint32_t)bn); (void)c.run(); }
          if (o.bail) {
            o.post(lz4hip::MAIL_ABORT, 0u, 0u);
            routed[nr.fetch_add(1)] = b;
            a.out[b] = -2;
          }
        } else {
          a.out[b] = 0;
        }
        head = o.head; tail_seen = o.tail_seen;
      }
      if (w.oob) M::oob = 1;
    });
  for (auto& t : th) t.join();
  if (n_routed) *n_routed = nr.load();
  return M::oob ? -1000 : 0;
}

}  // extern "C"

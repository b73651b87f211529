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
#include <vector>

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
    lz4hip::FastCore<hostsim::WaveHost, false, lz4hip::ParkOut<hostsim::WaveHost>> c(w, out, src, (uint32_t)n, &st);
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
    lz4hip::FastCore<hostsim::WaveHost, false, lz4hip::ParkOutRaw<hostsim::WaveHost>> c(w, out, src, (uint32_t)n, &st);
    r = c.run();
  }
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
  else { lz4hip::FastCore<hostsim::WaveHost, false, lz4hip::ParkOut<hostsim::WaveHost>> c(w, out, src, (uint32_t)n); r = c.run(); }
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

// safe != 0: (src_size = compressed length) -> decoded size; safe == 0: (src_size = readable
// capacity) -> bytes consumed.  `gl` = lanes per group (4..64).  -1000000 = out-of-slot access.
int sim_decompress(const uint8_t* src, int src_size, uint8_t* dst, int out_size, int safe, int gl) {
  const bool pipe = (gl & 0x100) != 0;   // bit 8 of gl: the pipelined interior loop
  const bool stage = (gl & 0x200) != 0;  // bit 9: output staging
  gl &= 0xFF;
  hostsim::GroupHost g(gl, src, src_size, dst, out_size);
  int r;
  if (stage) r = safe ? lz4hip::decode_block<hostsim::GroupHost, true, false, true>(g, src, src_size, dst, out_size, g.stg_buf)
                      : lz4hip::decode_block<hostsim::GroupHost, false, false, true>(g, src, src_size, dst, out_size, g.stg_buf);
  else if (pipe) r = safe ? lz4hip::decode_block<hostsim::GroupHost, true, true>(g, src, src_size, dst, out_size)
                     : lz4hip::decode_block<hostsim::GroupHost, false, true>(g, src, src_size, dst, out_size);
  else r = safe ? lz4hip::decode_block<hostsim::GroupHost, true>(g, src, src_size, dst, out_size)
                : lz4hip::decode_block<hostsim::GroupHost, false>(g, src, src_size, dst, out_size);
  if (g.oob) return -1000000;
  return r;
}

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

}  // extern "C"

// tests/hostsim/hostsim_asm.cpp -- TEST INFRASTRUCTURE ONLY (never part of liblz4hip.so).
//
// The lean fast-compress core (lz4-java_amd/csrc/lz4_fast_v2_core.h) in the lock-step lane simulator WITH its hand-scheduled loop:
// the loop's text -- lean_asm_text.inc, generated from the preprocessed device headers by gen_asm_text.py -- is run by the ISA
// interpreter of asm_emu.h wherever the GPU build runs the assembled loop (LZ4HIP_HOST_ASM_EMU selects those call sites).
// Three kinds: byU16 blocks (lz4_fast_v2_asm.h), byU32 blocks with 64-bit entries (lz4_fast_v2_asm32.h), byU32 blocks of at most
// 4 MiB with packed 32-bit entries (same file, shared body).
#define LZ4HIP_HOST_ASM_EMU 1
#include <stdint.h>
#include <stdio.h>
#include "wave_host.h"
#include "asm_emu.h"
#ifndef LZ4HIP_ASM_TEXT_INC
#define LZ4HIP_ASM_TEXT_INC "lean_asm_text.inc"   /* (variant builds of the suite: the text generated with other -D flags) */
#endif
#include LZ4HIP_ASM_TEXT_INC
#include "../../lz4-java_amd/csrc/lz4_fast_v2_core.h"

namespace hostsim {

namespace { thread_local AsmEmu* g_emu[3] = {nullptr, nullptr, nullptr}; }

struct WaveHostAsm : WaveHost {
  static constexpr bool kAsmLean = true;
  uint64_t asm_calls = 0, asm_insns = 0, asm_parked = 0;
  bool asm_error = false;
  uint64_t poison = 0x1234567ull;

  template <bool U16, bool PK>
  void lean_asm_emu(uint32_t& ip, uint32_t& php, uint32_t& pfe, uint32_t& pc, VU& pfa, VU& pms, VU& pml, VU& pof, uint32_t lim,
                    const uint8_t* src, uint32_t n) {
    // (one interpreter per loop and per LIBRARY: a static inside this template would be one object for every variant of the
    // simulator library loaded into the process -- such statics are unique symbols -- and a variant would run another one's text)
    AsmEmu*& emu = g_emu[U16 ? 0 : (PK ? 2 : 1)];
    if (!emu) {
      emu = new AsmEmu();
      if (!emu->load(U16 ? kLeanAsmU16 : (PK ? kLeanAsmU32P : kLeanAsmU32))) { fprintf(stderr, "asm_emu: %s\n", emu->error.c_str()); asm_error = true; }
    }
    AsmEmu& e = *emu;
    if (asm_error) { oob = true; return; }
    // every register the statement does not set itself holds garbage on the GPU: poison them, so that a read of an unset register
    // shows up as a wrong byte here
    for (int r = 0; r < 256; r++) for (int l = 0; l < 64; l++) { poison = poison * 6364136223846793005ull + 1442695040888963407ull; e.v[r][l] = (uint32_t)(poison >> 32); }
    for (int r = 0; r < 128; r++) { poison = poison * 6364136223846793005ull + 1442695040888963407ull; e.s[r] = (uint32_t)(poison >> 32); }
    e.vcc = poison; e.scc = (poison >> 7) & 1u; e.m0 = (uint32_t)(poison >> 9);
    e.exec = ~0ull;
    e.rng = rng; rng = rng * 6364136223846793005ull + 1442695040888963407ull;
    e.s[104] = ip; e.s[105] = php; e.s[106] = pfe; e.s[107] = pc; e.s[109] = lim; e.s[110] = n; e.s[111] = n - 16u;
    e.s[112] = 2654435761u; e.s[113] = 0x1bbcdcbbu;
    const uint64_t sp = (uint64_t)(uintptr_t)src;
    e.s[114] = (uint32_t)sp; e.s[115] = (uint32_t)(sp >> 32);
    for (int l = 0; l < 64; l++) {
      e.v[200][l] = pfa.v[l]; e.v[201][l] = pms.v[l]; e.v[202][l] = pml.v[l]; e.v[203][l] = pof.v[l];
      e.v[204][l] = 0u; e.v[205][l] = (uint32_t)l; e.v[206][l] = 4u * (uint32_t)l; e.v[207][l] = 16u * (uint32_t)l;
    }
    e.lds = (uint8_t*)lds.data(); e.lds_bytes = (!U16 && PK) ? 16384u : 32768u;
    e.g_lo = src; e.g_hi = src + n;
    const uint64_t before = e.executed;
    const uint32_t pc0 = pc;
    const uint32_t ip0 = ip, php0 = php;
    if (!e.run()) { fprintf(stderr, "asm_emu: %s\n", e.error.c_str()); asm_error = true; oob = true; return; }
    asm_calls++; asm_insns += e.executed - before;
    ip = e.s[104]; php = e.s[105]; pfe = e.s[106]; pc = e.s[107];
    asm_parked += pc - pc0;
    if (getenv("ASM_EMU_TRACE")) {
      fprintf(stderr, "asm: ip %u php %d pc %u -> ip %u php %d pc %u code %u  parked:", ip0, (int)(ip0 - php0), pc0, ip, (int)(ip - php), pc, e.s[108]);
      for (uint32_t k = pc0; k < pc && k < 64; k++) fprintf(stderr, " [%u +%u -%u]", e.v[201][k], e.v[202][k], e.v[203][k]);
      fprintf(stderr, "\n");
    }
    if (e.exec != ~0ull) { fprintf(stderr, "asm_emu: the loop left exec = %016llx\n", (unsigned long long)e.exec); asm_error = true; oob = true; }
    for (int l = 0; l < 64; l++) { pfa.v[l] = e.v[200][l]; pms.v[l] = e.v[201][l]; pml.v[l] = e.v[202][l]; pof.v[l] = e.v[203][l]; }
  }
};

// the finder's output policy of the GPU kernels (mail_ring.h MailOutT) without the ring: bare hits parked by the loop itself,
// resolved and written 64 at a time (what the writer wavefront does)
template <class W>
struct ParkOutAsm : lz4hip::ParkOutRaw<W> {
  using P = lz4hip::ParkOutRaw<W>;
  static constexpr bool kAsmPark = true;
  ParkOutAsm(W& w_, const uint8_t* s, uint32_t n_, uint8_t* d, uint32_t cap_) : P(w_, s, n_, d, cap_) {}
  void batch() { P::resolve_raw(); P::flush(); }
};

}  // namespace hostsim

extern "C" {

// kind: 0 = by size (byU16 below 65547 bytes, packed byU32 up to 4 MiB, 64-bit entries above); 1 = 64-bit entries for every byU32 block.
// stats3: calls of the loop, instructions interpreted, hits parked by the loop.  returns the compressed size (0 = does not fit),
// -1000 = an access outside the block / its table, -2000 = the interpreter stopped (unknown instruction, endless loop ...)
int sim_asm_compress(const uint8_t* src, int n, uint8_t* dst, int cap, int kind, uint64_t* stats3, uint64_t seed) {
  if (n < 0 || (uint32_t)n > 0x7E000000u || cap < 0) return 0;
  using W = hostsim::WaveHostAsm;
  W w;
  if (seed) { w.rng = seed; w.poison = seed * 31u + 7u; }
  w.bounds(src, (size_t)n, dst, (size_t)cap);
  hostsim::ParkOutAsm<W> out(w, src, (uint32_t)n, dst, (uint32_t)cap);
  uint32_t r;
  if (n < 65547) { lz4hip::FastV2<W, hostsim::ParkOutAsm<W>> c(w, out, src, (uint32_t)n); r = c.run(); }
  else if (kind == 0 && n <= (1 << 22)) { lz4hip::FastV2<W, hostsim::ParkOutAsm<W>, false, true> c(w, out, src, (uint32_t)n); r = c.run(); }
  else { lz4hip::FastV2<W, hostsim::ParkOutAsm<W>, false> c(w, out, src, (uint32_t)n); r = c.run(); }
  if (stats3) { stats3[0] = w.asm_calls; stats3[1] = w.asm_insns; stats3[2] = w.asm_parked; }
  if (w.asm_error) return -2000;
  if (w.oob) return -1000;
  return (int)r;
}

}  // extern "C"

// ---- the hand-written walk of the parallel wave decoder (lz4-java_amd/csrc/group_dev.h vwalk): its TEXT in the interpreter ----
// nx: the packed next-position bytes of a window (four per lane); out: posv (64 lanes) and the count.  Returns 0, or -1 when the
// interpreter stopped on something it does not know.
extern "C" int sim_wave_walk_asm(const uint32_t* nx, uint32_t t_in, uint32_t* posv_out, uint32_t* t_out) {
  static hostsim::AsmEmu* emu = nullptr;
  if (!emu) {
    emu = new hostsim::AsmEmu();
    if (!emu->load(kWaveWalkAsm)) { fprintf(stderr, "asm_emu (walk): %s\n", emu->error.c_str()); return -1; }
  }
  hostsim::AsmEmu& e = *emu;
  uint64_t poison = 0x9E3779B97F4A7C15ull ^ nx[0];
  for (int r = 0; r < 256; r++) for (int l = 0; l < 64; l++) { poison = poison * 6364136223846793005ull + 1442695040888963407ull; e.v[r][l] = (uint32_t)(poison >> 32); }
  for (int r = 0; r < 128; r++) { poison = poison * 6364136223846793005ull + 1442695040888963407ull; e.s[r] = (uint32_t)(poison >> 32); }
  e.vcc = poison; e.scc = (poison >> 7) & 1u; e.m0 = (uint32_t)(poison >> 9); e.exec = ~0ull;
  e.s[116] = t_in; e.s[117] = 0u;                       // T, s (the statement's "+s" operands: the caller's values)
  for (int l = 0; l < 64; l++) { e.v[208][l] = 0u; e.v[209][l] = nx[l]; }   // posv starts at zero (lz4_decode_wave.h), nx
  if (!e.run()) { fprintf(stderr, "asm_emu (walk): %s\n", e.error.c_str()); return -1; }
  for (int l = 0; l < 64; l++) posv_out[l] = e.v[208][l];
  *t_out = e.s[116];
  return 0;
}

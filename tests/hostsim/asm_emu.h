// tests/hostsim/asm_emu.h -- TEST INFRASTRUCTURE ONLY (never part of liblz4hip.so).
//
// An interpreter for the gfx950 instructions the hand-scheduled match-finder loops are written in
// (lz4-java_amd/csrc/lz4_fast_v2_asm.h, lz4_fast_v2_asm32.h, lz4_fast_v2_asm_body.inc): the CPU suite runs the very TEXT the GPU
// assembles -- extracted from the preprocessed headers when the simulator library is built (tests/hostsim/gen_asm_text.py), so it
// cannot drift -- inside the same lock-step compressor the C++ cores run in (lz4_fast_v2_core.h, LZ4HIP_HOST_ASM_EMU), with the
// table in the simulated LDS and the block in host memory, and compares the compressed bytes with the reference library
// (tests/test_hostsim.py::test_asm_loop_*).  Until round 4 these loops were checked on the GPU only.
//
// What is modelled: 64 lanes, exec / vcc / scc / m0, 256 VGPRs, 128 SGPRs, the LDS bytes, global loads with bounds checks; every
// instruction of the three loops (about sixty opcodes; anything else stops the run with an error, so a new instruction in the
// loop fails the test until it is added here).  What is not: time -- waits are no-ops, loads land at once -- and hardware hazards.
// LDS atomics of one instruction are applied in a PSEUDO-RANDOM lane order (like the C++ simulator's): the loop's collision
// handling must not depend on the order in which the hardware serialises the lanes of a bucket.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <string>
#include <vector>

namespace hostsim {

struct AsmEmu {
  enum Kind : uint8_t { K_NONE, K_V, K_S, K_VCC, K_EXEC, K_M0, K_IMM, K_LABEL };
  struct Opnd { Kind k = K_NONE; int r = 0, n = 1; int64_t imm = 0; };
  struct Insn { std::string op; Opnd o[5]; int no = 0; int offset = 0; bool sdwa_w0 = false; int target = -1; std::string text; };

  std::vector<Insn> prog;
  uint32_t v[256][64];
  uint32_t s[128];
  uint64_t vcc = 0, exec = ~0ull;
  bool scc = false;
  uint32_t m0 = 0;
  uint8_t* lds = nullptr; size_t lds_bytes = 0;
  const uint8_t* g_lo = nullptr; const uint8_t* g_hi = nullptr;   // readable global range
  uint64_t rng = 0x243F6A8885A308D3ull;
  bool shuffle = true;
  std::string error;
  uint64_t executed = 0;

  // named operands of the asm statements -> registers of the interpreter
  static const std::map<std::string, std::string>& names() {
    static const std::map<std::string, std::string> m = {
        {"ip", "s104"}, {"php", "s105"}, {"pfe", "s106"}, {"pc", "s107"}, {"code", "s108"}, {"lim", "s109"}, {"n", "s110"}, {"ntop", "s111"},
        {"kmul", "s112"}, {"plo", "s113"}, {"src", "s[114:115]"}, {"pfa", "v200"}, {"pms", "v201"}, {"pml", "v202"}, {"pof", "v203"},
        {"tbl", "v204"}, {"lane", "v205"}, {"j4", "v206"}, {"j16", "v207"},
        // the walk of the parallel wave decoder (lz4-java_amd/csrc/group_dev.h vwalk)
        {"T", "s116"}, {"s", "s117"}, {"t1", "s118"}, {"t2", "s119"}, {"pv", "v208"}, {"nx", "v209"}};
    return m;
  }

  static bool parse_opnd(const std::string& t, Opnd& o) {
    if (t.empty()) return false;
    if (t == "vcc") { o.k = K_VCC; return true; }
    if (t == "exec") { o.k = K_EXEC; return true; }
    if (t == "m0") { o.k = K_M0; return true; }
    if ((t[0] == 'v' || t[0] == 's') && t.size() > 1 && (isdigit((unsigned char)t[1]) || t[1] == '[')) {
      o.k = t[0] == 'v' ? K_V : K_S;
      if (t[1] == '[') { int a = 0, b = 0; if (sscanf(t.c_str() + 2, "%d:%d", &a, &b) != 2) return false; o.r = a; o.n = b - a + 1; }
      else { o.r = atoi(t.c_str() + 1); o.n = 1; }
      return true;
    }
    if (isdigit((unsigned char)t[0]) || t[0] == '-') { o.k = K_IMM; o.imm = (int64_t)strtoll(t.c_str(), nullptr, 0); return true; }
    return false;
  }

  // text of one asm statement (operands as %[name], labels with %=)
  bool load(const char* text) {
    prog.clear(); error.clear();
    std::map<std::string, int> labels;
    std::vector<std::pair<int, std::string>> fix;
    std::string all(text);
    size_t pos = 0;
    while (pos < all.size()) {
      size_t e = all.find('\n', pos);
      if (e == std::string::npos) e = all.size();
      std::string line = all.substr(pos, e - pos);
      pos = e + 1;
      // named operands
      for (size_t p; (p = line.find("%[")) != std::string::npos;) {
        const size_t q = line.find(']', p);
        const std::string nm = line.substr(p + 2, q - p - 2);
        auto it = names().find(nm);
        if (it == names().end()) { error = "unknown operand " + nm; return false; }
        line.replace(p, q - p + 1, it->second);
      }
      // trim + comments
      const size_t c = line.find("/*"); if (c != std::string::npos) line = line.substr(0, c);
      size_t a = line.find_first_not_of(" \t"); if (a == std::string::npos) continue;
      line = line.substr(a, line.find_last_not_of(" \t") - a + 1);
      if (line.empty()) continue;
      if (line.back() == ':') { labels[line.substr(0, line.size() - 1)] = (int)prog.size(); continue; }
      Insn in; in.text = line;
      const size_t sp = line.find_first_of(" \t");
      in.op = line.substr(0, sp);
      std::string rest = sp == std::string::npos ? "" : line.substr(sp + 1);
      // modifiers
      for (size_t p; (p = rest.find("offset:")) != std::string::npos;) { in.offset = atoi(rest.c_str() + p + 7); size_t q = rest.find_first_of(" \t", p); rest.erase(p, (q == std::string::npos ? rest.size() : q) - p); }
      if (rest.find("src0_sel:WORD_0") != std::string::npos) in.sdwa_w0 = true;
      for (const char* mod : {"src0_sel:WORD_0", "src1_sel:DWORD", "src0_sel:DWORD"}) { size_t p = rest.find(mod); if (p != std::string::npos) rest.erase(p, strlen(mod)); }
      if (in.op == "s_waitcnt") { prog.push_back(in); continue; }
      // operands
      size_t p = 0;
      while (p < rest.size()) {
        size_t q = p; int depth = 0;
        while (q < rest.size() && (rest[q] != ',' || depth)) { if (rest[q] == '[') depth++; if (rest[q] == ']') depth--; q++; }
        std::string t = rest.substr(p, q - p);
        size_t a2 = t.find_first_not_of(" \t");
        if (a2 != std::string::npos) {
          t = t.substr(a2, t.find_last_not_of(" \t") - a2 + 1);
          // constant expressions of the form "64 - 18" do not occur in the product text
          if (in.no >= 5) { error = "too many operands: " + line; return false; }
          if (in.op.rfind("s_branch", 0) == 0 || in.op.rfind("s_cbranch", 0) == 0) { fix.push_back({(int)prog.size(), t}); in.o[in.no].k = K_LABEL; in.no++; }
          else if (!parse_opnd(t, in.o[in.no])) { error = "operand '" + t + "' in: " + line; return false; }
          else in.no++;
        }
        p = q + 1;
      }
      if (in.op.size() > 4 && (in.op.substr(in.op.size() - 4) == "_e32" || in.op.substr(in.op.size() - 4) == "_e64")) in.op = in.op.substr(0, in.op.size() - 4);
      prog.push_back(in);
    }
    for (auto& f : fix) {
      auto it = labels.find(f.second);
      if (it == labels.end()) { error = "label " + f.second; return false; }
      prog[f.first].target = it->second;
    }
    return true;
  }

  // ---- operand access ----
  uint64_t rd_s(const Opnd& o, bool b64) const {
    switch (o.k) {
      case K_S: return b64 ? ((uint64_t)s[o.r] | ((uint64_t)s[o.r + 1] << 32)) : s[o.r];
      case K_VCC: return b64 ? vcc : (uint32_t)vcc;
      case K_EXEC: return b64 ? exec : (uint32_t)exec;
      case K_M0: return m0;
      case K_IMM: return b64 ? (uint64_t)o.imm : (uint32_t)o.imm;   // (inline constants are sign-extended to 64 bits)
      default: return 0;
    }
  }
  void wr_s(const Opnd& o, uint64_t x, bool b64) {
    switch (o.k) {
      case K_S: s[o.r] = (uint32_t)x; if (b64) s[o.r + 1] = (uint32_t)(x >> 32); break;
      case K_VCC: vcc = b64 ? x : ((vcc & ~0xFFFFFFFFull) | (uint32_t)x); break;
      case K_EXEC: exec = b64 ? x : ((exec & ~0xFFFFFFFFull) | (uint32_t)x); break;
      case K_M0: m0 = (uint32_t)x; break;
      default: break;
    }
  }
  uint32_t rd_v(const Opnd& o, int lane, int word = 0) const {   // a VALU source: VGPR, SGPR or constant
    if (o.k == K_V) return v[o.r + word][lane];
    if (o.k == K_IMM) return word ? (uint32_t)((uint64_t)o.imm >> 32) : (uint32_t)o.imm;
    return (uint32_t)(rd_s(o, true) >> (32 * word));
  }
  bool active(int lane) const { return (exec >> lane) & 1ull; }
  uint32_t next_rand() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng >> 33); }
  bool lds_ok(uint64_t a, uint32_t bytes) { if (a + bytes > lds_bytes || (a & (bytes - 1u))) { error = "LDS access out of range or misaligned"; return false; } return true; }
  bool g_ok(const uint8_t* p, uint32_t bytes) { if (p < g_lo || p + bytes > g_hi) { error = "global load outside the block"; return false; } return true; }

  // runs the loaded statement from its first instruction to its end; false: error (see `error`)
  bool run() {
    int pc = 0;
    uint64_t budget = 400000000ull;
    while (pc < (int)prog.size()) {
      if (!budget--) { error = "instruction budget exhausted (endless loop?)"; return false; }
      const Insn& in = prog[pc];
      const std::string& op = in.op;
      const Opnd* o = in.o;
      int next = pc + 1;
      executed++;
#define VALU1(expr) do { for (int l = 0; l < 64; l++) if (active(l)) { const uint32_t a = rd_v(o[1], l); (void)a; v[o[0].r][l] = (expr); } } while (0)
#define VALU2(expr) do { for (int l = 0; l < 64; l++) if (active(l)) { const uint32_t a = rd_v(o[1], l), b = rd_v(o[2], l); (void)a; (void)b; v[o[0].r][l] = (expr); } } while (0)
#define VALU3(expr) do { for (int l = 0; l < 64; l++) if (active(l)) { const uint32_t a = rd_v(o[1], l), b = rd_v(o[2], l), c = rd_v(o[3], l); (void)a; (void)b; (void)c; v[o[0].r][l] = (expr); } } while (0)
      if (op == "s_waitcnt") {}
      // ---- VALU ----
      else if (op == "v_add_u32") VALU2(a + b);
      else if (op == "v_and_b32") VALU2(a & b);
      else if (op == "v_xor_b32") VALU2(a ^ b);
      else if (op == "v_min_u32") VALU2(a < b ? a : b);
      else if (op == "v_mov_b32") VALU1(a);
      else if (op == "v_mul_lo_u32") VALU2(a * b);
      else if (op == "v_mul_u32_u24") VALU2((a & 0xFFFFFFu) * (b & 0xFFFFFFu));
      else if (op == "v_lshrrev_b32") VALU2(b >> (a & 31u));
      else if (op == "v_lshl_add_u32") VALU3((a << (b & 31u)) + c);
      else if (op == "v_lshl_or_b32") VALU3((a << (b & 31u)) | c);
      else if (op == "v_add3_u32") VALU3(a + b + c);
      else if (op == "v_bfe_u32") VALU3((c & 31u) ? ((a >> (b & 31u)) & ((1u << (c & 31u)) - 1u)) : 0u);
      else if (op == "v_alignbyte_b32") VALU3((uint32_t)((((uint64_t)a << 32) | b) >> (8u * (c & 3u))));
      else if (op == "v_alignbit_b32") VALU3((uint32_t)((((uint64_t)a << 32) | b) >> (c & 31u)));
      else if (op == "v_ffbl_b32") VALU1(a ? (uint32_t)__builtin_ctz(a) : 0xFFFFFFFFu);
      else if (op == "v_mov_b64") { for (int l = 0; l < 64; l++) if (active(l)) { v[o[0].r][l] = rd_v(o[1], l, 0); v[o[0].r + 1][l] = rd_v(o[1], l, 1); } }
      else if (op == "v_mad_u64_u32") {   // vdst[2], sdst (carry, unused), a, b, c (64 bits)
        for (int l = 0; l < 64; l++) if (active(l)) {
          const uint64_t c64 = o[4].k == K_V ? ((uint64_t)v[o[4].r][l] | ((uint64_t)v[o[4].r + 1][l] << 32)) : (uint64_t)o[4].imm;
          const uint64_t r = (uint64_t)rd_v(o[2], l) * (uint64_t)rd_v(o[3], l) + c64;
          v[o[0].r][l] = (uint32_t)r; v[o[0].r + 1][l] = (uint32_t)(r >> 32);
        }
      }
      else if (op == "v_cmp_ne_u32" || op == "v_cmp_eq_u32" || op == "v_cmp_ge_u32" || op == "v_cmp_eq_u32_sdwa" || op == "v_cmp_ne_u64") {
        uint64_t m = 0;
        for (int l = 0; l < 64; l++) if (active(l)) {
          bool r;
          if (op == "v_cmp_ne_u64") {
            const uint64_t a = (uint64_t)rd_v(o[1], l, 0) | ((uint64_t)rd_v(o[1], l, 1) << 32), b = (uint64_t)rd_v(o[2], l, 0) | ((uint64_t)rd_v(o[2], l, 1) << 32);
            r = a != b;
          } else {
            uint32_t a = rd_v(o[1], l); const uint32_t b = rd_v(o[2], l);
            if (in.sdwa_w0) a &= 0xFFFFu;
            r = op == "v_cmp_ne_u32" ? a != b : op == "v_cmp_ge_u32" ? a >= b : a == b;
          }
          if (r) m |= 1ull << l;
        }
        wr_s(o[0], m, true);
      }
      else if (op == "v_readlane_b32") wr_s(o[0], v[o[1].r][rd_s(o[2], false) & 63u], false);
      else if (op == "v_writelane_b32") v[o[0].r][rd_s(o[2], false) & 63u] = (uint32_t)rd_s(o[1], false);
      // ---- LDS ----
      else if (op == "ds_bpermute_b32") {
        uint32_t t[64];
        for (int l = 0; l < 64; l++) t[l] = v[o[2].r][((v[o[1].r][l] + (uint32_t)in.offset) >> 2) & 63u];
        for (int l = 0; l < 64; l++) if (active(l)) v[o[0].r][l] = t[l];
      }
      else if (op == "ds_read_b32" || op == "ds_read_b64") {
        const uint32_t w = op == "ds_read_b64" ? 8u : 4u;
        for (int l = 0; l < 64; l++) if (active(l)) {
          const uint64_t a = (uint64_t)v[o[1].r][l] + (uint32_t)in.offset;
          if (!lds_ok(a, w)) return false;
          memcpy(&v[o[0].r][l], lds + a, 4);
          if (w == 8u) memcpy(&v[o[0].r + 1][l], lds + a + 4, 4);
        }
      }
      else if (op == "ds_write_b32" || op == "ds_write_b64") {
        const uint32_t w = op == "ds_write_b64" ? 8u : 4u;
        for (int l = 0; l < 64; l++) if (active(l)) {   // (ascending lanes: the highest lane wins an address, as on the hardware -- profiles/r04_lds_write_order.txt)
          const uint64_t a = (uint64_t)v[o[0].r][l] + (uint32_t)in.offset;
          if (!lds_ok(a, w)) return false;
          memcpy(lds + a, &v[o[1].r][l], 4);
          if (w == 8u) memcpy(lds + a + 4, &v[o[1].r + 1][l], 4);
        }
      }
      else if (op == "ds_max_rtn_u32" || op == "ds_max_rtn_u64") {
        const bool w64 = op == "ds_max_rtn_u64";
        int order[64];
        for (int l = 0; l < 64; l++) order[l] = l;
        if (shuffle) for (int l = 63; l > 0; l--) { const int k = (int)(next_rand() % (uint32_t)(l + 1)); const int t = order[l]; order[l] = order[k]; order[k] = t; }
        for (int i = 0; i < 64; i++) {
          const int l = order[i];
          if (!active(l)) continue;
          const uint64_t a = (uint64_t)v[o[1].r][l] + (uint32_t)in.offset;
          if (!lds_ok(a, w64 ? 8u : 4u)) return false;
          if (w64) {
            uint64_t old; memcpy(&old, lds + a, 8);
            const uint64_t val = (uint64_t)v[o[2].r][l] | ((uint64_t)v[o[2].r + 1][l] << 32);
            const uint64_t nw = old > val ? old : val; memcpy(lds + a, &nw, 8);
            v[o[0].r][l] = (uint32_t)old; v[o[0].r + 1][l] = (uint32_t)(old >> 32);
          } else {
            uint32_t old; memcpy(&old, lds + a, 4);
            const uint32_t val = v[o[2].r][l];
            const uint32_t nw = old > val ? old : val; memcpy(lds + a, &nw, 4);
            v[o[0].r][l] = old;
          }
        }
      }
      // ---- global ----
      else if (op == "global_load_dword" || op == "global_load_dwordx4") {
        const uint32_t words = op == "global_load_dword" ? 1u : 4u;
        const uint8_t* base = (const uint8_t*)(uintptr_t)rd_s(o[2], true);
        for (int l = 0; l < 64; l++) if (active(l)) {
          const uint8_t* p = base + v[o[1].r][l];
          if (!g_ok(p, 4u * words)) return false;
          for (uint32_t k = 0; k < words; k++) memcpy(&v[o[0].r + k][l], p + 4u * k, 4);
        }
      }
      // ---- SALU ----
      else if (op == "s_mov_b32") wr_s(o[0], rd_s(o[1], false), false);
      else if (op == "s_mov_b64") wr_s(o[0], rd_s(o[1], true), true);
      else if (op == "s_add_u32") { const uint64_t r = (uint64_t)(uint32_t)rd_s(o[1], false) + (uint32_t)rd_s(o[2], false); wr_s(o[0], (uint32_t)r, false); scc = (r >> 32) != 0; }
      else if (op == "s_sub_u32") { const uint32_t a = (uint32_t)rd_s(o[1], false), b = (uint32_t)rd_s(o[2], false); wr_s(o[0], a - b, false); scc = b > a; }
      else if (op == "s_and_b32") { const uint32_t r = (uint32_t)rd_s(o[1], false) & (uint32_t)rd_s(o[2], false); wr_s(o[0], r, false); scc = r != 0; }
      else if (op == "s_and_b64") { const uint64_t r = rd_s(o[1], true) & rd_s(o[2], true); wr_s(o[0], r, true); scc = r != 0; }
      else if (op == "s_or_b64") { const uint64_t r = rd_s(o[1], true) | rd_s(o[2], true); wr_s(o[0], r, true); scc = r != 0; }
      else if (op == "s_andn2_b64") { const uint64_t r = rd_s(o[1], true) & ~rd_s(o[2], true); wr_s(o[0], r, true); scc = r != 0; }
      else if (op == "s_lshl_b64") { const uint64_t r = rd_s(o[1], true) << (rd_s(o[2], false) & 63u); wr_s(o[0], r, true); scc = r != 0; }
      else if (op == "s_lshr_b64") { const uint64_t r = rd_s(o[1], true) >> (rd_s(o[2], false) & 63u); wr_s(o[0], r, true); scc = r != 0; }
      else if (op == "s_ff1_i32_b64") { const uint64_t a = rd_s(o[1], true); wr_s(o[0], a ? (uint32_t)__builtin_ctzll(a) : 0xFFFFFFFFu, false); }
      else if (op == "s_bcnt1_i32_b64") { const uint32_t r = (uint32_t)__builtin_popcountll(rd_s(o[1], true)); wr_s(o[0], r, false); scc = r != 0; }
      else if (op == "s_brev_b32") { uint32_t a = (uint32_t)rd_s(o[1], false), r = 0; for (int b = 0; b < 32; b++) r |= ((a >> b) & 1u) << (31 - b); wr_s(o[0], r, false); }
      else if (op == "s_cselect_b32") wr_s(o[0], scc ? rd_s(o[1], false) : rd_s(o[2], false), false);
      else if (op == "s_cmp_gt_u32" || op == "s_cmpk_gt_u32") scc = (uint32_t)rd_s(o[0], false) > (uint32_t)rd_s(o[1], false);
      else if (op == "s_cmp_le_u32") scc = (uint32_t)rd_s(o[0], false) <= (uint32_t)rd_s(o[1], false);
      else if (op == "s_cmpk_eq_u32") scc = (uint32_t)rd_s(o[0], false) == (uint32_t)rd_s(o[1], false);
      else if (op == "s_cmpk_lg_u32") scc = (uint32_t)rd_s(o[0], false) != (uint32_t)rd_s(o[1], false);
      else if (op == "s_lshr_b32") { const uint32_t r = (uint32_t)rd_s(o[1], false) >> ((uint32_t)rd_s(o[2], false) & 31u); wr_s(o[0], r, false); scc = r != 0; }
      else if (op == "s_lshl_b32") { const uint32_t r = (uint32_t)rd_s(o[1], false) << ((uint32_t)rd_s(o[2], false) & 31u); wr_s(o[0], r, false); scc = r != 0; }
      else if (op == "s_min_u32") { const uint32_t a = (uint32_t)rd_s(o[1], false), b = (uint32_t)rd_s(o[2], false); wr_s(o[0], a < b ? a : b, false); scc = a < b; }
      else if (op == "s_cmp_ge_u32") scc = (uint32_t)rd_s(o[0], false) >= (uint32_t)rd_s(o[1], false);
      else if (op == "s_cmp_lt_u32") scc = (uint32_t)rd_s(o[0], false) < (uint32_t)rd_s(o[1], false);
      else if (op == "s_cmp_eq_u32") scc = (uint32_t)rd_s(o[0], false) == (uint32_t)rd_s(o[1], false);
      else if (op == "s_cmp_lg_u32") scc = (uint32_t)rd_s(o[0], false) != (uint32_t)rd_s(o[1], false);
      else if (op == "s_cmp_eq_u64") scc = rd_s(o[0], true) == rd_s(o[1], true);
      else if (op == "s_branch") next = in.target;
      else if (op == "s_cbranch_scc0") { if (!scc) next = in.target; }
      else if (op == "s_cbranch_scc1") { if (scc) next = in.target; }
      else if (op == "s_cbranch_vccz") { if (vcc == 0) next = in.target; }
      else { error = "instruction not modelled: " + in.text; return false; }
#undef VALU1
#undef VALU2
#undef VALU3
      pc = next;
    }
    return true;
  }
};

}  // namespace hostsim

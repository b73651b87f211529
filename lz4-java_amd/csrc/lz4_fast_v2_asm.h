// lz4_fast_v2_asm.h -- the common step of the lean match finder (lz4_fast_v2_core.h, FastV2::lean) in hand-scheduled gfx950 ISA.
//
// Same contract as the C++ loop it stands in front of (byte-identical output to LZ4_compress_default of liblz4 1.9.3,
// /root/reference/src/jni/net_jpountz_lz4_LZ4JNI.c:75): the block below runs post-match steps
//   {insert ip-2, probe ip, probe ip+1 ..}
// for as long as they are the COMMON kind -- the window lies in the 256-byte row fetched at the previous hit, the first tentative
// lane (or, after ruled-out ones, a later one of the same window) verifies, its match is shorter than 256 bytes and no two
// committing lanes share a bucket -- and parks each bare hit {position, forward length, offset} in a lane of the parked registers.
// Anything else is UNDONE (the lanes that committed write their old buckets back) and left to the C++ step, which starts from
// the same state and knows every rule; so the hard cases keep their one definition and this file holds no policy.
//
// Why by hand (profiles/r03_compress_notes.txt):
//   * the compiler's loop is 110 instructions per step, 65 of them scalar control flow; a wavefront issues one instruction per
//     ~4.4 cycles whatever the dependencies, so that is ~480 of the step's ~1265 cycles.  This loop is ~60 (+ 25 for the lookahead);
//   * the hardware has ONE in-order counter for vector memory operations and the compiler waits with vmcnt(0) around anything
//     divergent, so a load that is issued "for later" is waited for at once.  Here the waits are counted by hand, which makes a
//     LOOKAHEAD affordable: the candidate lines of the tentative lanes behind the hit (the rest of this window and the 64
//     positions after it, looked up in the table while the rows are in flight) are requested one step before the step that
//     needs them, and nobody waits for them -- the next step's row fetch then finds its candidate line on its way or in L1
//     instead of paying a full Infinity-Cache / HBM round trip (the candidate side is a random line of the block; misses cost
//     ~340 cycles per step on average).  A prefetch is a hint: a stale table entry costs a useless line, never a wrong byte.
//
// The lookahead was built (candidate lines of the window's other tentative lanes, and of the 64 positions behind the window,
// requested behind the rows and never waited for) and measured: -2 % .. -4 % (profiles/r03_compress_notes.txt) -- the phase
// timers show why: the rows arrive ~480 cycles after their request on average, i.e. most candidate lines already come from L2,
// and a line in L1 is not much closer than that.  What the hand-written loop buys is issue time and the order of things:
// requests first, bookkeeping in their shadow.
//
// Exit codes: 1 = ip > lim; 2 = the step at `ip` is for the C++ loop (nothing of it is left in the table); 3 = 64 hits parked.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef LZ4HIP_V2_ASM
#define LZ4HIP_V2_ASM 1
#endif
// The loop below is gfx950 machine code (register numbers, wait counters and instruction forms of that ISA): the library is built for
// that one target (lz4-java_amd/build.sh), and a device pass for anything else must not get as far as assembling it.
#if LZ4HIP_V2_ASM && defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "lz4_fast_v2_asm.h is hand-scheduled gfx950 ISA: build with --offload-arch=gfx950 (or -DLZ4HIP_V2_ASM=0 for the compiler-generated lean loop)"
#endif

#ifndef LZ4HIP_ASM_DBG
#define LZ4HIP_ASM_DBG 0
#endif
#ifndef LZ4HIP_V2_PF
#define LZ4HIP_V2_PF 0   /* developer A/B: candidate lines of the next windows' tentative slots requested at the end of the shadow */
#endif
#ifndef LZ4HIP_FP_BITS
#define LZ4HIP_FP_BITS 16   /* (lz4_fast_core.h: developer builds with narrower byU16 fingerprints) */
#endif
#ifndef LZ4HIP_V2_RETRY
#define LZ4HIP_V2_RETRY 0   /* 1 (not the product yet: verified in the CPU suite's ISA interpreter only, tests/test_hostsim.py): a false hit -- fingerprints agree, bytes differ -- is
                               handled inside the loop instead of leaving it; the lookups then start one position earlier (slot 0 = hit + 1) */
#endif
#ifndef LZ4HIP_V2_ASM_PROF
#define LZ4HIP_V2_ASM_PROF 0   /* developer builds: shader-clock time per phase of the hand-scheduled step, summed into g_asm_prof */
#endif

namespace lz4hip {

#if LZ4HIP_ASM_DBG & 4
__device__ unsigned int g_asm_dbg[1600];
#endif
#if LZ4HIP_V2_ASM_PROF
// [0..3] cycles: window + hash + table read | hit selection + commit + row requests | waiting for the rows | count + park;
// [4] steps, [5] exits with code 2, [6] cycles inside lean(), [7] cycles inside run(), [8] blocks, [9] exact-path calls
__device__ unsigned long long g_asm_prof[16];
#define LZ4HIP_TICK0 "  s_memtime s[92:93]\n  s_waitcnt lgkmcnt(0)\n"
#define LZ4HIP_TICK(acc) "  s_memtime s[94:95]\n  s_waitcnt lgkmcnt(0)\n  s_sub_u32 s96, s94, s92\n  s_add_u32 %[" acc "], %[" acc "], s96\n  s_mov_b32 s92, s94\n"
#define LZ4HIP_COUNT(acc) "  s_add_u32 %[" acc "], %[" acc "], 1\n"
#else
#define LZ4HIP_TICK0 ""
#define LZ4HIP_TICK(acc) ""
#define LZ4HIP_COUNT(acc) ""
#endif

#if LZ4HIP_V2_RETRY && LZ4HIP_V2_ASM_PROF
#error "LZ4HIP_V2_RETRY and LZ4HIP_V2_ASM_PROF use the same scalar registers"
#endif
// where slot 0 of a lookup set lies behind the hit in flight, and from it the insert-only slot / the first probe slot of a match of length L
#if LZ4HIP_V2_RETRY
#define LZ4HIP_LEAN_S0 "1"
#define LZ4HIP_LEAN_INS "-3"
#define LZ4HIP_LEAN_PRB "-1"
#define LZ4HIP_RETRY_CLOBBERS , "s92", "s93", "s96", "s97"
#else
#define LZ4HIP_LEAN_S0 "2"
#define LZ4HIP_LEAN_INS "-4"
#define LZ4HIP_LEAN_PRB "-2"
#define LZ4HIP_RETRY_CLOBBERS
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define LZ4HIP_STR2(x) #x
#define LZ4HIP_STR(x) LZ4HIP_STR2(x)
// The loop is software-pipelined around the one thing that cannot be made shorter, the fetch of the two 256-byte rows at the
// hit and at its candidate (~480 cycles, mostly L2):
//   * WHILE the rows of hit n are in flight the wavefront looks up the table for the 128 positions [h_n + 2, h_n + 130) -- any
//     next window {insert ip-2, probe ip .. ip+62} with ip = h_n + cnt, cnt = 4..65, lies inside -- two positions per lane
//     ("slot" i of lane i & 63, group A = slots 0..63, group B = 64..127), out of the row of the PREVIOUS hit: window words
//     (lane permutes), hashes, table entries, fingerprints, new entries and the two tentative masks are all there before the
//     rows arrive.  The lookups follow this step's commit in the wavefront's LDS order, so they see the table liblz4 has when it
//     starts probing at ip;
//   * WHEN the rows arrive the critical path to the next request is scalar: forward length -> the window as two lane masks of
//     the slots -> first tentative slot -> its candidate position (one v_readlane) -> the two row requests;
//   * commit (LDS atomic max per group), collision test, parking of the previous hit and the next 128 lookups follow in the
//     shadow of the new request.
// Exits are taken at "clean" points only -- the last validated hit parked, nothing of a later window left in the table
// (a speculative commit is undone from the saved entries) -- so the C++ step can always continue.
//
// registers of the block (fixed, declared as clobbers):
//   v48 shift  v49 permute address  v50..v53 row words / window words / hash products
//   group A: v54 LDS address  v55 entry  v56 new entry  v57 candidate position      group B: v58 .. v61 likewise
//   v62/v63 values the atomics returned   v64..v67 saved {address, entry} of both groups (undo of the commit in flight)
//   v68 row at the hit  v69 row at the candidate  v70 forward length per lane  v71 address scratch  v72..v75 source touch
//   v76/v77 fingerprints  v78 slot positions
//   s70 position of slot 0   s71 hit slot   s72/s73 hit and candidate position of the hit in flight   s74 its length
//   s75..s77 scratch (s76: first slot of the window)   s[78:79]/s[80:81] tentative slots A/B   s[82:83]/s[84:85] committing slots
//   s[86:89] scratch masks   s90 effective loop limit (0: leave at the next clean point)   s91 a validated hit waits to be parked
//   s[94:95] 63 ones
#if LZ4HIP_V2_ASM_PROF
#undef LZ4HIP_TICK0
#undef LZ4HIP_TICK
#define LZ4HIP_TICK0 "  s_memtime s[92:93]\n  s_waitcnt lgkmcnt(0)\n"
#define LZ4HIP_TICK(acc) "  s_memtime s[96:97]\n  s_waitcnt lgkmcnt(0)\n  s_sub_u32 s98, s96, s92\n  s_add_u32 %[" acc "], %[" acc "], s98\n  s_mov_b32 s92, s96\n"
#endif
// the 128 lookups: sreg = byte offset of slot 0 in the row %[pfa], s70 = position of slot 0.  Part 1 requests the row words;
// part 2 (after the wait that also brings the commit's results back) hashes them and reads the table
#define LZ4HIP_BUILD_E1(sreg) \
      "  v_add_u32 v48, " sreg ", %[lane]\n" \
      "  v_and_b32 v49, -4, v48\n" \
      "  v_and_b32 v48, 3, v48\n" \
      "  ds_bpermute_b32 v50, v49, %[pfa]\n" \
      "  ds_bpermute_b32 v51, v49, %[pfa] offset:4\n" \
      "  ds_bpermute_b32 v52, v49, %[pfa] offset:64\n" \
      "  ds_bpermute_b32 v53, v49, %[pfa] offset:68\n" \
      "  v_add_u32 v78, s70, %[lane]\n"
#define LZ4HIP_BUILD_E2 \
      "  v_alignbyte_b32 v50, v51, v50, v48\n" \
      "  v_alignbyte_b32 v52, v53, v52, v48\n" \
      "  v_mul_lo_u32 v51, v50, %[kmul]\n" \
      "  v_mul_lo_u32 v53, v52, %[kmul]\n" \
      "  v_lshrrev_b32 v54, 19, v51\n" \
      "  v_lshrrev_b32 v58, 19, v53\n" \
      "  v_lshl_add_u32 v54, v54, 2, %[tbl]\n" \
      "  v_lshl_add_u32 v58, v58, 2, %[tbl]\n" \
      "  ds_read_b32 v55, v54\n" \
      "  ds_read_b32 v59, v58\n" \
      "  v_bfe_u32 v76, v51, 3, " LZ4HIP_STR(LZ4HIP_FP_BITS) "\n" \
      "  v_bfe_u32 v77, v53, 3, " LZ4HIP_STR(LZ4HIP_FP_BITS) "\n" \
      "  v_lshl_or_b32 v56, v78, 16, v76\n" \
      "  v_add_u32 v78, 64, v78\n" \
      "  v_lshl_or_b32 v60, v78, 16, v77\n" \
      "  s_waitcnt lgkmcnt(0)\n" \
      "  v_cmp_eq_u32_sdwa s[78:79], v55, v76 src0_sel:WORD_0 src1_sel:DWORD\n" \
      "  v_cmp_eq_u32_sdwa s[80:81], v59, v77 src0_sel:WORD_0 src1_sel:DWORD\n" \
      "  v_lshrrev_b32 v57, 16, v55\n" \
      "  v_lshrrev_b32 v61, 16, v59\n"
// first tentative slot of the window whose first probe slot is s75 (its insert-only slot is s76 = s75 - 2): slot -> s71,
// candidate position -> s77, hit position -> s75, rows requested; `nohit` = label for "no tentative slot among the 63 probes"
#define LZ4HIP_SELECT(tag, nohit) \
      "  s_lshl_b64 s[86:87], s[94:95], s75\n" \
      "  s_and_b64 s[88:89], s[86:87], s[78:79]\n" \
      "  s_cbranch_scc0 L_selB" tag "_%=\n" \
      "  s_ff1_i32_b64 s71, s[88:89]\n" \
      "  v_readlane_b32 s77, v57, s71\n" \
      "L_req" tag "_%=:\n" \
      "  s_add_u32 s75, s70, s71\n" \
      "  v_add_u32 v71, s77, %[j4]\n" \
      "  global_load_dword v69, v71, %[src]\n" \
      "  v_add_u32 v71, s75, %[j4]\n" \
      "  global_load_dword v68, v71, %[src]\n"
#define LZ4HIP_SELECT_B(tag, nohit) \
      "L_selB" tag "_%=:\n" \
      "  s_sub_u32 s77, 62, s76\n" \
      "  s_lshr_b64 s[88:89], s[94:95], s77\n" \
      "  s_and_b64 s[88:89], s[88:89], s[80:81]\n" \
      "  s_cbranch_scc0 " nohit "_%=\n" \
      "  s_ff1_i32_b64 s71, s[88:89]\n" \
      "  v_readlane_b32 s77, v61, s71\n" \
      "  s_add_u32 s71, s71, 64\n" \
      "  s_branch L_req" tag "_%=\n"
#define LZ4HIP_PARK \
      "  s_mov_b32 m0, %[pc]\n" \
      "  s_sub_u32 s86, s72, s73\n" \
      "  v_writelane_b32 %[pms], s72, m0\n" \
      "  v_writelane_b32 %[pml], s74, m0\n" \
      "  v_writelane_b32 %[pof], s86, m0\n" \
      "  s_add_u32 %[pc], %[pc], 1\n" \
      LZ4HIP_COUNT("c0")
__device__ __forceinline__ uint32_t lean_asm_run(uint32_t& ip, uint32_t& php, uint32_t& pfe, uint32_t& pc, uint32_t& pfa,
                                                 uint32_t& pms, uint32_t& pml, uint32_t& pof, uint32_t lim, const uint8_t* src,
                                                 uint32_t tbl, uint32_t n, uint32_t* pr) {
  uint32_t code;
#if LZ4HIP_V2_ASM_PROF
  uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, c0 = 0, c1 = 0;
#endif
  const uint32_t lane = __lane_id();
  const uint32_t j4 = lane * 4u, j16 = lane * 16u;
  const uint32_t kmul = 2654435761u, ntop = n - 16u;
  asm volatile(
#define LZ4HIP_LEAN_E1(sreg) LZ4HIP_BUILD_E1(sreg)
#define LZ4HIP_LEAN_E2 LZ4HIP_BUILD_E2
#define LZ4HIP_LEAN_ROWLIM "125"
#define LZ4HIP_LEAN_FPMASK "0xffff"
#include "lz4_fast_v2_asm_body.inc"
#undef LZ4HIP_LEAN_E1
#undef LZ4HIP_LEAN_E2
#undef LZ4HIP_LEAN_ROWLIM
#undef LZ4HIP_LEAN_FPMASK
      : [ip] "+s"(ip), [php] "+s"(php), [pfe] "+s"(pfe), [pc] "+s"(pc), [pfa] "+v"(pfa), [pms] "+v"(pms), [pml] "+v"(pml),
        [pof] "+v"(pof), [code] "=&s"(code)
#if LZ4HIP_V2_ASM_PROF
        , [t0] "+s"(t0), [t1] "+s"(t1), [t2] "+s"(t2), [t3] "+s"(t3), [c0] "+s"(c0), [c1] "+s"(c1)
#endif
      : [lim] "s"(lim), [src] "s"(src), [tbl] "v"(tbl), [n] "s"(n), [ntop] "s"(ntop), [kmul] "s"(kmul), [lane] "v"(lane), [j4] "v"(j4),
        [j16] "v"(j16)
      : "memory", "vcc", "scc", "m0", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59",
        "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74",
        "v75", "v76", "v77", "v78", "v79", "v80", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82",
        "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s94", "s95" LZ4HIP_RETRY_CLOBBERS
#if LZ4HIP_V2_ASM_PROF
        , "s92", "s93", "s96", "s97", "s98"
#endif
  );
#if LZ4HIP_V2_ASM_PROF
  pr[0] += t0; pr[1] += t1; pr[2] += t2; pr[3] += t3; pr[4] += c0; pr[5] += c1;
#endif
  return code;
}
#endif  // __HIP_DEVICE_COMPILE__

}  // namespace lz4hip

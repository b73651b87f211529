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
//   v100 shift  v101 permute address  v102..v105 row words / window words / hash products
//   group A: v106 LDS address  v107 entry  v108 new entry  v109 candidate position      group B: v110 .. v113 likewise
//   v114/v115 values the atomics returned   v116..v119 saved {address, entry} of both groups (undo of the commit in flight)
//   v120 row at the hit  v121 row at the candidate  v122 forward length per lane  v123 address scratch  v124..v127 source touch
//   v128/v129 fingerprints  v130 slot positions
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
// the 128 lookups: s75 = byte offset of slot 0 in the row %[pfa], s70 = position of slot 0.  Part 1 requests the row words;
// part 2 (after the wait that also brings the commit's results back) hashes them and reads the table
#define LZ4HIP_BUILD_E1 \
      "  v_add_u32 v100, s75, %[lane]\n" \
      "  v_and_b32 v101, -4, v100\n" \
      "  v_and_b32 v100, 3, v100\n" \
      "  ds_bpermute_b32 v102, v101, %[pfa]\n" \
      "  ds_bpermute_b32 v103, v101, %[pfa] offset:4\n" \
      "  ds_bpermute_b32 v104, v101, %[pfa] offset:64\n" \
      "  ds_bpermute_b32 v105, v101, %[pfa] offset:68\n" \
      "  v_add_u32 v130, s70, %[lane]\n"
#define LZ4HIP_BUILD_E2 \
      "  v_alignbyte_b32 v102, v103, v102, v100\n" \
      "  v_alignbyte_b32 v104, v105, v104, v100\n" \
      "  v_mul_lo_u32 v103, v102, %[kmul]\n" \
      "  v_mul_lo_u32 v105, v104, %[kmul]\n" \
      "  v_lshrrev_b32 v106, 19, v103\n" \
      "  v_lshrrev_b32 v110, 19, v105\n" \
      "  v_lshl_add_u32 v106, v106, 2, %[tbl]\n" \
      "  v_lshl_add_u32 v110, v110, 2, %[tbl]\n" \
      "  ds_read_b32 v107, v106\n" \
      "  ds_read_b32 v111, v110\n" \
      "  v_bfe_u32 v128, v103, 3, 16\n" \
      "  v_bfe_u32 v129, v105, 3, 16\n" \
      "  v_lshl_or_b32 v108, v130, 16, v128\n" \
      "  v_add_u32 v130, 64, v130\n" \
      "  v_lshl_or_b32 v112, v130, 16, v129\n" \
      "  s_waitcnt lgkmcnt(0)\n" \
      "  v_cmp_eq_u32_sdwa s[78:79], v107, v128 src0_sel:WORD_0 src1_sel:DWORD\n" \
      "  v_cmp_eq_u32_sdwa s[80:81], v111, v129 src0_sel:WORD_0 src1_sel:DWORD\n" \
      "  v_lshrrev_b32 v109, 16, v107\n" \
      "  v_lshrrev_b32 v113, 16, v111\n"
// first tentative slot of the window whose first probe slot is s75 (its insert-only slot is s76 = s75 - 2): slot -> s71,
// candidate position -> s77, hit position -> s75, rows requested; `nohit` = label for "no tentative slot among the 63 probes"
#define LZ4HIP_SELECT(tag, nohit) \
      "  s_lshl_b64 s[86:87], s[94:95], s75\n" \
      "  s_and_b64 s[88:89], s[86:87], s[78:79]\n" \
      "  s_cbranch_scc0 L_selB" tag "_%=\n" \
      "  s_ff1_i32_b64 s71, s[88:89]\n" \
      "  v_readlane_b32 s77, v109, s71\n" \
      "L_req" tag "_%=:\n" \
      "  s_add_u32 s75, s70, s71\n" \
      "  v_add_u32 v123, s77, %[j4]\n" \
      "  global_load_dword v121, v123, %[src]\n" \
      "  v_add_u32 v123, s75, %[j4]\n" \
      "  global_load_dword v120, v123, %[src]\n"
#define LZ4HIP_SELECT_B(tag, nohit) \
      "L_selB" tag "_%=:\n" \
      "  s_sub_u32 s77, 62, s76\n" \
      "  s_lshr_b64 s[88:89], s[94:95], s77\n" \
      "  s_and_b64 s[88:89], s[88:89], s[80:81]\n" \
      "  s_cbranch_scc0 " nohit "_%=\n" \
      "  s_ff1_i32_b64 s71, s[88:89]\n" \
      "  v_readlane_b32 s77, v113, s71\n" \
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
      // ---- entry: post-match state at ip, the row %[pfa] holds the block's bytes from php on
      "  s_cmp_gt_u32 %[ip], %[lim]\n"
      "  s_cbranch_scc1 L_x1_%=\n"
      "  s_cmp_eq_u32 %[pc], 63\n"
      "  s_cbranch_scc1 L_x2_%=\n"
      "  s_sub_u32 s75, %[ip], %[php]\n"
      "  s_add_u32 s75, s75, -2\n"
      "  s_cmpk_gt_u32 s75, 125\n"
      "  s_cbranch_scc1 L_x2_%=\n"
      "  s_lshr_b64 s[94:95], -1, 1\n"
      "  s_add_u32 s70, %[ip], -2\n"
      "  s_mov_b32 s91, 0\n"
#if LZ4HIP_ASM_DBG & 2
      "  s_mov_b32 s90, 0\n"
#else
      "  s_mov_b32 s90, %[lim]\n"
#endif
      LZ4HIP_BUILD_E1
      "  s_waitcnt lgkmcnt(0)\n"
      LZ4HIP_BUILD_E2
      "  s_mov_b32 s76, 0\n"
      "  s_mov_b32 s75, 2\n"
      LZ4HIP_TICK0
      LZ4HIP_SELECT("p", "L_x2")
      "  s_branch L_commit_%=\n"
      LZ4HIP_SELECT_B("p", "L_x2")
      // ---- top of the loop: the rows of the hit at s72 (candidate s73) arrive
      "L_arrive_%=:\n"
      LZ4HIP_TICK("t1")
#if LZ4HIP_V2_PF
      "  s_waitcnt vmcnt(2)\n"
#else
      "  s_waitcnt vmcnt(0)\n"
#endif
      LZ4HIP_TICK("t2")
      "L_arrived_%=:\n"
      "  v_xor_b32 v122, v120, v121\n"
      "  v_cmp_ne_u32_e32 vcc, 0, v122\n"
      "  v_ffbl_b32 v122, v122\n"
      "  v_lshrrev_b32 v122, 3, v122\n"
      "  v_lshl_add_u32 v122, %[lane], 2, v122\n"
      "  v_mov_b32 %[pfa], v120\n"            // the row at the hit: the next windows are cut from it (before v120 is requested again)
      "  s_ff1_i32_b64 s75, vcc\n"
      "  v_readlane_b32 s74, v122, s75\n"     // forward length
      "  s_add_u32 s76, s74, -4\n"
      "  s_add_u32 s75, s74, -2\n"
      LZ4HIP_SELECT("l", "L_nohit")
      // what has to hold for the hit at s72 before anything of the next window is committed: it differs from its candidate
      // within 256 bytes, is 4 .. 65 bytes long (the next window lies among the 128 slots), and the loop goes on
      "  s_cbranch_vccz L_f1_%=\n"
      "  s_cmpk_gt_u32 s76, 61\n"
      "  s_cbranch_scc1 L_odd_%=\n"
      "  s_add_u32 %[ip], s72, s74\n"
      "  s_cmp_gt_u32 %[ip], s90\n"
      "  s_cbranch_scc1 L_clean_%=\n"
      "  s_mov_b32 %[php], s72\n"
      // ---- in the shadow of the request: the window's slots up to the hit slot s71 commit their inserts
      "L_commit_%=:\n"
      LZ4HIP_TICK("t0")
      "  s_lshl_b64 s[82:83], 1, s76\n"
      "  s_or_b64 s[82:83], s[82:83], s[86:87]\n"
      "  s_lshl_b64 s[88:89], -2, s71\n"
      "  s_cmp_lt_u32 s71, 64\n"
      "  s_cbranch_scc0 L_maskB_%=\n"
      "  s_andn2_b64 s[82:83], s[82:83], s[88:89]\n"
      "  s_mov_b64 s[84:85], 0\n"
      "L_masked_%=:\n"
      "  s_mov_b64 exec, s[82:83]\n"
      "  ds_max_rtn_u32 v114, v106, v108\n"
      "  s_mov_b64 exec, s[84:85]\n"
      "  ds_max_rtn_u32 v115, v110, v112\n"
      "  s_mov_b64 exec, -1\n"
      // the next 128 lookups are cut from the row at php, slot 0 = (new hit) + 2; their row words are requested first, the
      // bookkeeping runs while they (and the commit's results) are on their way
      "  s_sub_u32 s86, s75, %[php]\n"
      "  s_add_u32 s70, s75, 2\n"
      "  s_add_u32 s86, s86, 2\n"
      "  v_add_u32 v100, s86, %[lane]\n"
      "  v_and_b32 v101, -4, v100\n"
      "  v_and_b32 v100, 3, v100\n"
      "  ds_bpermute_b32 v102, v101, %[pfa]\n"
      "  ds_bpermute_b32 v103, v101, %[pfa] offset:4\n"
      "  ds_bpermute_b32 v104, v101, %[pfa] offset:64\n"
      "  ds_bpermute_b32 v105, v101, %[pfa] offset:68\n"
      "  v_add_u32 v130, s70, %[lane]\n"
      "  s_cmpk_gt_u32 s86, 125\n"
      "  s_cselect_b32 s90, 0, s90\n"          // the row does not reach: leave at the next clean point
      // the hit validated above is parked: {position, forward length, offset}
      "  s_cmp_eq_u32 s91, 0\n"
      "  s_cbranch_scc1 L_nopark_%=\n"
      LZ4HIP_PARK
      "  s_cmp_eq_u32 %[pc], 63\n"
      "  s_cselect_b32 s90, 0, s90\n"          // the parked registers will be full: leave at the next clean point
      "L_nopark_%=:\n"
      "  s_mov_b32 s91, 1\n"
      "  s_mov_b32 s72, s75\n"
      "  s_mov_b32 s73, s77\n"
      "  v_mov_b32 v116, v106\n"
      "  v_mov_b32 v118, v110\n"
      "  s_waitcnt lgkmcnt(0)\n"
      // a slot that got back another slot's entry: two committing slots share a bucket
      "  v_cmp_ne_u32_e64 s[86:87], v114, v107\n"
      "  v_cmp_ne_u32_e64 s[88:89], v115, v111\n"
      "  s_and_b64 s[86:87], s[86:87], s[82:83]\n"
      "  s_and_b64 s[88:89], s[88:89], s[84:85]\n"
      "  s_or_b64 vcc, s[86:87], s[88:89]\n"
      "  s_cbranch_scc1 L_coll_%=\n"
      "L_cont_%=:\n"
      "  v_mov_b32 v117, v107\n"
      "  v_mov_b32 v119, v111\n"
      LZ4HIP_BUILD_E2
#if LZ4HIP_V2_PF
      // (A/B) the candidate lines of every tentative slot among the 128: the next hit's candidate is one of them
      "  s_mov_b64 exec, s[78:79]\n"
      "  global_load_dword v131, v109, %[src]\n"
      "  s_mov_b64 exec, s[80:81]\n"
      "  global_load_dword v132, v113, %[src]\n"
      "  s_mov_b64 exec, -1\n"
#endif
      "  s_add_u32 s75, s72, 1024\n"
      "  s_cmp_gt_u32 s75, %[pfe]\n"
      "  s_cbranch_scc0 L_arrive_%=\n"
      // the source is touched 1 KB ahead of the parse (one 1 KB wave load per 1 KB of progress; nobody waits for it: it is
      // older than the next rows)
      "  s_cmp_ge_u32 %[pfe], %[n]\n"
      "  s_cbranch_scc1 L_touched_%=\n"
      "  v_add_u32 v123, %[pfe], %[j16]\n"
      "  v_min_u32 v123, %[ntop], v123\n"
      "  global_load_dwordx4 v[124:127], v123, %[src]\n"
      "  s_add_u32 %[pfe], %[pfe], 1024\n"
#if LZ4HIP_V2_PF
      "  s_waitcnt vmcnt(3)\n"
#else
      "  s_waitcnt vmcnt(1)\n"
#endif
      "  s_branch L_arrived_%=\n"
      "L_touched_%=:\n"
      "  s_add_u32 %[pfe], %[pfe], 1024\n"
      "  s_branch L_arrive_%=\n"
      LZ4HIP_SELECT_B("l", "L_nohit")
      "L_maskB_%=:\n"                          // the hit slot is in group B: all window slots of A commit, B's up to the hit
      "  s_sub_u32 s86, 62, s76\n"
      "  s_lshr_b64 s[84:85], s[94:95], s86\n"
      "  s_andn2_b64 s[84:85], s[84:85], s[88:89]\n"
      "  s_branch L_masked_%=\n"
      // ---- clean exits: the hit in s72..s74 is validated and not parked yet; nothing of the next window is committed
      "L_nohit_%=:\n"                           // no tentative slot among the next window's 63 probes
      "  s_cbranch_vccz L_f1_%=\n"
      "  s_cmpk_gt_u32 s76, 61\n"
      "  s_cbranch_scc1 L_odd_%=\n"
      "L_long_%=:\n"
      "  s_add_u32 %[ip], s72, s74\n"
      "L_clean_%=:\n"
      "  s_mov_b32 %[php], s72\n"
      LZ4HIP_PARK
      "  s_cmp_gt_u32 %[ip], %[lim]\n"
      "  s_cselect_b32 %[code], 1, 2\n"
      "  s_branch L_out_%=\n"
      "L_odd_%=:\n"
      "  s_cmp_lt_u32 s74, 4\n"
      "  s_cbranch_scc0 L_long_%=\n"           // a match of more than 65 bytes: the next window is not among the 128 slots
      // the hit's candidate differs within its first four bytes, or is equal for 256: its commit is undone from the saved entries
      "L_f1_%=:\n"
      "  s_brev_b32 %[php], 1\n"               // (the row in %[pfa] starts at that hit, possibly at ip itself: no row for the C++ step)
      "  s_mov_b64 exec, s[82:83]\n"
      "  ds_write_b32 v116, v117\n"
      "  s_mov_b64 exec, s[84:85]\n"
      "  ds_write_b32 v118, v119\n"
      "  s_mov_b64 exec, -1\n"
      "  s_branch L_x2_%=\n"
      // Two committing slots share a bucket (a slot got back another slot's entry).  The rule of the C++ step: with exactly ONE
      // such slot, the foreign entry's fingerprint different from that slot's own, and the bucket not the hit slot's, liblz4 --
      // inserting position by position -- would not have decided differently (the slot's old entry was no hit and neither is
      // the foreign one) and the atomic max has left the bucket as liblz4 leaves it: carry on.  Everything else is undone and
      // left to the C++ step.
      "L_coll_%=:\n"
#if LZ4HIP_ASM_DBG & 1
      "  s_branch L_undo_%=\n"
#endif
      "  s_bcnt1_i32_b64 s75, vcc\n"
      "  s_cmp_eq_u32 s75, 1\n"
      "  s_cbranch_scc0 L_undo_%=\n"
      "  s_cmp_eq_u64 s[86:87], 0\n"
      "  s_cbranch_scc1 L_collB_%=\n"
      "  s_ff1_i32_b64 s75, s[86:87]\n"
      "  v_readlane_b32 s76, v114, s75\n"
      "  v_readlane_b32 s77, v128, s75\n"
      "  v_readlane_b32 s75, v106, s75\n"
      "  s_branch L_coll2_%=\n"
      "L_collB_%=:\n"
      "  s_ff1_i32_b64 s75, s[88:89]\n"
      "  v_readlane_b32 s76, v115, s75\n"
      "  v_readlane_b32 s77, v129, s75\n"
      "  v_readlane_b32 s75, v110, s75\n"
      "L_coll2_%=:\n"
      "  s_and_b32 s76, s76, 0xffff\n"
      "  s_cmp_eq_u32 s76, s77\n"
      "  s_cbranch_scc1 L_undo_%=\n"
      "  v_readlane_b32 s76, v106, s71\n"
      "  v_readlane_b32 s77, v110, s71\n"
      "  s_cmp_lt_u32 s71, 64\n"
      "  s_cselect_b32 s76, s76, s77\n"
      "  s_cmp_eq_u32 s75, s76\n"
      "  s_cbranch_scc0 L_cont_%=\n"
      "L_undo_%=:\n"
      "  s_mov_b64 exec, s[82:83]\n"
      "  ds_write_b32 v106, v107\n"
      "  s_mov_b64 exec, s[84:85]\n"
      "  ds_write_b32 v110, v111\n"
      "  s_mov_b64 exec, -1\n"
      "L_x2_%=:\n"
      LZ4HIP_COUNT("c1")
      "  s_mov_b32 %[code], 2\n"
      "  s_branch L_out_%=\n"
      "L_x1_%=:\n"
      "  s_mov_b32 %[code], 1\n"
      "L_out_%=:\n"
      "  s_waitcnt vmcnt(0) lgkmcnt(0)\n"
      : [ip] "+s"(ip), [php] "+s"(php), [pfe] "+s"(pfe), [pc] "+s"(pc), [pfa] "+v"(pfa), [pms] "+v"(pms), [pml] "+v"(pml),
        [pof] "+v"(pof), [code] "=&s"(code)
#if LZ4HIP_V2_ASM_PROF
        , [t0] "+s"(t0), [t1] "+s"(t1), [t2] "+s"(t2), [t3] "+s"(t3), [c0] "+s"(c0), [c1] "+s"(c1)
#endif
      : [lim] "s"(lim), [src] "s"(src), [tbl] "v"(tbl), [n] "s"(n), [ntop] "s"(ntop), [kmul] "s"(kmul), [lane] "v"(lane), [j4] "v"(j4),
        [j16] "v"(j16)
      : "memory", "vcc", "scc", "m0", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",
        "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126",
        "v127", "v128", "v129", "v130", "v131", "v132", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82",
        "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s94", "s95"
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

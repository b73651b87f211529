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

#ifndef LZ4HIP_V2_ASM_PROF
#define LZ4HIP_V2_ASM_PROF 0   /* developer builds: shader-clock time per phase of the hand-scheduled step, summed into g_asm_prof */
#endif

namespace lz4hip {

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
// registers of the block (fixed, declared as clobbers):
//   v100 shift (o & 3)      v101 4 * (o >> 2)         v102/v103 row words     v104 window word x32    v105 hash product
//   v106 LDS address        v107 table entry e        v108 fingerprint        v109 new entry          v110 atomic's old value
//   v111 address scratch    v112 row at the hit       v113 row at candidate   v114 first-set-bit      v115/v119 prefetch sinks
//   v116..v118 lookahead window / address / fingerprint                    v120..v123 source touch
//   s70 ip - prev_hpos      s71 k0   s72 hpos   s73 mpos   s74 cnt   s75 first differing lane   s76/s77 scratch
//   s[78:79] tentative lanes left   s[80:81] all tentative lanes   s[82:83] lanes 0..k0   s[84:85] lanes committing now
//   s[86:87] lanes committed by earlier tries   s[88:89] scratch mask   s[90:91] lookahead mask
__device__ __forceinline__ uint32_t lean_asm_run(uint32_t& ip, uint32_t& php, uint32_t& pfe, uint32_t& pc, uint32_t& pfa,
                                                 uint32_t& pms, uint32_t& pml, uint32_t& pof, uint32_t lim, const uint8_t* src,
                                                 uint32_t tbl, uint32_t n, uint32_t* pr) {
  uint32_t code;
#if LZ4HIP_V2_ASM_PROF
  uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, c0 = 0, c1 = 0;
#endif
  const uint32_t lane = __lane_id();
  const uint32_t cj = lane == 0u ? 0xFFFFFFFEu : lane - 1u;
  const uint32_t j4 = lane * 4u, j16 = lane * 16u;
  const uint32_t kmul = 2654435761u, ntop = n - 16u;
  asm volatile(
      "L_top_%=:\n"
      "  s_cmp_gt_u32 %[ip], %[lim]\n"
      "  s_cbranch_scc1 L_exit1_%=\n"
      "  s_sub_u32 s70, %[ip], %[php]\n"
      "  s_cmpk_gt_u32 s70, 189\n"
      "  s_cbranch_scc1 L_exit2_%=\n"
      // window: bytes [ip - 2, ip + 66) out of the row at the previous hit
      "  v_add_u32 v100, s70, %[cj]\n"
      "  v_and_b32 v101, -4, v100\n"
      "  v_and_b32 v100, 3, v100\n"
      "  ds_bpermute_b32 v102, v101, %[pfa]\n"
      "  ds_bpermute_b32 v103, v101, %[pfa] offset:4\n"
      LZ4HIP_TICK0
      "L_win_%=:\n"
      "  v_add_u32 v109, %[ip], %[cj]\n"
      "  s_waitcnt lgkmcnt(0)\n"
      "  v_alignbyte_b32 v104, v103, v102, v100\n"
      "  v_mul_lo_u32 v105, v104, %[kmul]\n"
      "  v_lshrrev_b32 v106, 19, v105\n"
      "  v_lshl_add_u32 v106, v106, 2, %[tbl]\n"
      "  ds_read_b32 v107, v106\n"
      "  v_bfe_u32 v108, v105, 3, 16\n"
      "  v_lshl_or_b32 v109, v109, 16, v108\n"
      "  s_mov_b64 s[86:87], 0\n"
      "  s_waitcnt lgkmcnt(0)\n"
      LZ4HIP_TICK("t0")
      "  v_cmp_eq_u32_sdwa s[78:79], v107, v108 src0_sel:WORD_0 src1_sel:DWORD\n"
      "  v_lshrrev_b32 v115, 16, v107\n"
      "  s_and_b64 s[78:79], s[78:79], -2\n"
      "  s_cbranch_scc0 L_exit2_%=\n"
      // the first tentative lane k0: its rows are requested first (the candidate side is the slow one), then lanes .. k0 commit
      "L_try_%=:\n"
      "  s_ff1_i32_b64 s71, s[78:79]\n"
      "  v_readlane_b32 s73, v115, s71\n"
      "  s_add_u32 s72, %[ip], s71\n"
      "  s_lshl_b64 s[82:83], -2, s71\n"
      "  v_add_u32 v111, s73, %[j4]\n"
      "  global_load_dword v113, v111, %[src]\n"
      "  s_add_u32 s72, s72, -1\n"
      "  v_add_u32 v111, s72, %[j4]\n"
      "  global_load_dword v112, v111, %[src]\n"
      "  s_not_b64 s[82:83], s[82:83]\n"
      "  s_andn2_b64 s[84:85], s[82:83], s[86:87]\n"
      "  s_mov_b64 exec, s[84:85]\n"
      "  ds_max_rtn_u32 v110, v106, v109\n"
      "  s_mov_b64 exec, -1\n"
      "  s_add_u32 s77, s72, 1024\n"
      "  s_cmp_gt_u32 s77, %[pfe]\n"
      "  s_cbranch_scc1 L_touch_%=\n"
      "L_back_%=:\n"
      LZ4HIP_TICK("t1")
      // the atomic's result: a lane that got back another lane's entry
      "  s_waitcnt lgkmcnt(0)\n"
      "  v_cmp_ne_u32_e64 s[88:89], v110, v107\n"
      "  s_and_b64 s[88:89], s[88:89], s[84:85]\n"
      "  s_cbranch_scc1 L_coll_%=\n"
      "L_cont_%=:\n"
      "  s_waitcnt vmcnt(0)\n"
      LZ4HIP_TICK("t2")
      "L_post_%=:\n"
      "  v_xor_b32 v113, v112, v113\n"
      "  v_cmp_ne_u32_e32 vcc, 0, v113\n"
      "  v_ffbl_b32 v114, v113\n"
      "  s_cbranch_vccz L_undo_%=\n"
      "  s_ff1_i32_b64 s75, vcc\n"
      "  v_readlane_b32 s76, v114, s75\n"
      "  s_lshr_b32 s76, s76, 3\n"
      "  s_lshl2_add_u32 s74, s75, s76\n"
      "  s_cmp_lt_u32 s74, 4\n"
      "  s_cbranch_scc1 L_ruled_%=\n"
      // a hit of s74 bytes at s72.  The next step's window words are requested from the new row at once (the next step starts at
      // s72 + s74, so its window lies s74 bytes into the row); parking {position, forward length, offset} and the loop's exit
      // tests run while they are on their way.
      "  v_add_u32 v100, s74, %[cj]\n"
      "  v_and_b32 v101, -4, v100\n"
      "  v_and_b32 v100, 3, v100\n"
      "  ds_bpermute_b32 v102, v101, v112\n"
      "  ds_bpermute_b32 v103, v101, v112 offset:4\n"
      "  s_mov_b32 m0, %[pc]\n"
      "  s_sub_u32 s77, s72, s73\n"
      "  v_writelane_b32 %[pms], s72, m0\n"
      "  v_writelane_b32 %[pml], s74, m0\n"
      "  v_writelane_b32 %[pof], s77, m0\n"
      "  s_add_u32 %[ip], s72, s74\n"
      "  s_mov_b32 %[php], s72\n"
      "  v_mov_b32 %[pfa], v112\n"
      "  s_add_u32 %[pc], %[pc], 1\n"
      LZ4HIP_COUNT("c0")
      "  s_cmp_eq_u32 %[pc], 64\n"
      "  s_cbranch_scc1 L_exit3_%=\n"
      "  s_cmpk_gt_u32 s74, 189\n"
      "  s_cbranch_scc1 L_exit2_%=\n"
      "  s_cmp_gt_u32 %[ip], %[lim]\n"
      "  s_cbranch_scc1 L_exit1_%=\n"
      LZ4HIP_TICK("t3")
      "  s_branch L_win_%=\n"
      // the hit lane's candidate differs in its first four bytes (a fingerprint collision): the lanes up to it are inserted
      // positions, the search goes on to the next tentative lane of the window
      "L_ruled_%=:\n"
      "  s_mov_b64 s[86:87], s[82:83]\n"
      "  s_andn2_b64 s[78:79], s[78:79], s[82:83]\n"
      "  s_cbranch_scc1 L_try_%=\n"
      // everything else: the lanes that committed write their old buckets back
      "L_undo_%=:\n"
      "  s_mov_b64 exec, s[82:83]\n"
      "  ds_write_b32 v106, v107\n"
      "  s_mov_b64 exec, -1\n"
      "L_exit2_%=:\n"
      LZ4HIP_COUNT("c1")
      "  s_mov_b32 %[code], 2\n"
      "  s_branch L_out_%=\n"
      // Two committing lanes share a bucket (a lane got back another lane's entry).  The rule of the C++ step: with exactly ONE
      // such lane, and the foreign entry's fingerprint different from that lane's own, liblz4 -- inserting position by position --
      // would not have decided differently for it (its old entry was no hit and neither is the foreign one), and the atomic max
      // has left the bucket as liblz4 leaves it: carry on -- unless the bucket is the hit lane's (the entry that made it tentative
      // is not the one liblz4 would have found there): that lane is ruled out.  Everything else is for the exact path.
      "L_coll_%=:\n"
      "  s_bcnt1_i32_b64 s76, s[88:89]\n"
      "  s_cmp_eq_u32 s76, 1\n"
      "  s_cbranch_scc0 L_undo_%=\n"
      "  s_ff1_i32_b64 s75, s[88:89]\n"
      "  v_readlane_b32 s76, v110, s75\n"
      "  v_readlane_b32 s77, v108, s75\n"
      "  s_and_b32 s76, s76, 0xffff\n"
      "  s_cmp_eq_u32 s76, s77\n"
      "  s_cbranch_scc1 L_undo_%=\n"
      "  v_readlane_b32 s76, v106, s75\n"
      "  v_readlane_b32 s77, v106, s71\n"
      "  s_cmp_eq_u32 s76, s77\n"
      "  s_cbranch_scc0 L_cont_%=\n"
      "  s_branch L_ruled_%=\n"
      // the source is touched 1 KB ahead of the parse (one 1 KB wave load per 1 KB of progress)
      "L_touch_%=:\n"
      "  s_cmp_ge_u32 %[pfe], %[n]\n"
      "  s_cbranch_scc1 L_touched_%=\n"
      "  v_add_u32 v111, %[pfe], %[j16]\n"
      "  v_min_u32 v111, %[ntop], v111\n"
      "  global_load_dwordx4 v[120:123], v111, %[src]\n"
      "  s_add_u32 %[pfe], %[pfe], 1024\n"
      "  s_waitcnt lgkmcnt(0)\n"
      "  v_cmp_ne_u32_e64 s[88:89], v110, v107\n"
      "  s_and_b64 s[88:89], s[88:89], s[84:85]\n"
      "  s_cbranch_scc1 L_coll_%=\n"
      "  s_waitcnt vmcnt(1)\n"      // the two rows; the touch stays in flight (it is older than the next step's rows)
      "  s_branch L_post_%=\n"
      "L_touched_%=:\n"
      "  s_add_u32 %[pfe], %[pfe], 1024\n"
      "  s_branch L_back_%=\n"
      "L_exit3_%=:\n"
      "  s_mov_b32 %[code], 3\n"
      "  s_branch L_out_%=\n"
      "L_exit1_%=:\n"
      "  s_mov_b32 %[code], 1\n"
      "L_out_%=:\n"
      "  s_waitcnt vmcnt(0) lgkmcnt(0)\n"
      : [ip] "+s"(ip), [php] "+s"(php), [pfe] "+s"(pfe), [pc] "+s"(pc), [pfa] "+v"(pfa), [pms] "+v"(pms), [pml] "+v"(pml),
        [pof] "+v"(pof), [code] "=&s"(code)
#if LZ4HIP_V2_ASM_PROF
        , [t0] "+s"(t0), [t1] "+s"(t1), [t2] "+s"(t2), [t3] "+s"(t3), [c0] "+s"(c0), [c1] "+s"(c1)
#endif
      : [lim] "s"(lim), [src] "s"(src), [tbl] "v"(tbl), [n] "s"(n), [ntop] "s"(ntop), [kmul] "s"(kmul), [cj] "v"(cj), [j4] "v"(j4),
        [j16] "v"(j16)
      : "memory", "vcc", "scc", "m0", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",
        "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "s70", "s71", "s72", "s73",
        "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91"
#if LZ4HIP_V2_ASM_PROF
        , "s92", "s93", "s94", "s95", "s96"
#endif
  );
#if LZ4HIP_V2_ASM_PROF
  pr[0] += t0; pr[1] += t1; pr[2] += t2; pr[3] += t3; pr[4] += c0; pr[5] += c1;
#endif
  return code;
}
#endif  // __HIP_DEVICE_COMPILE__

}  // namespace lz4hip

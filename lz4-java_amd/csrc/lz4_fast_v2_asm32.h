// lz4_fast_v2_asm32.h -- the hand-scheduled lean step of lz4_fast_v2_asm.h for byU32 blocks (n >= 65547: liblz4's 4096-entry table,
// 5-byte hash, candidates more than 65535 bytes back are no hits; the LZ4 Frame default block is 4 MiB,
// /root/reference/src/java/net/jpountz/lz4/LZ4FrameOutputStream.java:169-171; LZ4_compress_default behind
// /root/reference/src/jni/net_jpountz_lz4_LZ4JNI.c:75).
//
// Same loop, same contract, same exits as lz4_fast_v2_asm.h (read that file first: the pipeline, the clean exits and the undo rule
// are explained there); what differs is the table side of a slot:
//   * entry = 64 bits {fingerprint (low dword, 16 bits used), position (high dword)}: ds_read_b64 / ds_max_rtn_u64 / ds_write_b64,
//     the candidate position is the high dword as it is, collisions are 64-bit compares;
//   * bucket = bits 28..39 of (5 bytes at the slot) * 889523592379 (= liblz4's ((sequence << 24) * prime5bytes) >> 52):
//     one 32 x 32 -> 64 multiply of the low four bytes + the two byte products that reach bits 32..39;
//     the fifth byte comes out of the row words the four-byte window is cut from (a window of 5 bytes at byte offset o of a dword pair
//     never leaves the pair);
//   * a slot is tentative if the fingerprints agree AND its entry lies within 65535 bytes.
// The FastV2<.., false> step in C++ (lz4_fast_v2_core.h) is the definition of every rule here and takes over at every exit.
#pragma once
#include "lz4_fast_v2_asm.h"

namespace lz4hip {

#if defined(__HIP_DEVICE_COMPILE__)
// registers of the block (fixed, declared as clobbers):
//   v100 shift  v101 permute address  v134/v135 row words of group A  v136/v137 of group B  v102..v105 window words / byte products
//   group A: v106 LDS address  v[108:109] new entry {fp, pos}  v[110:111] entry read      group B: v107, v[112:113], v[114:115]
//   v[116:117] / v[118:119] what the atomics returned   v120/v121 saved addresses, v[122:123] / v[124:125] saved entries (undo)
//   v126 row at the hit  v127 row at the candidate  v128 forward length per lane  v129 address scratch  v[130:133] source touch
//   v[140:143] products  v144/v145 scratch
//   scalars as in lz4_fast_v2_asm.h
#define LZ4HIP32_BUILD_E1 \
      "  v_add_u32 v100, s75, %[lane]\n" \
      "  v_and_b32 v101, -4, v100\n" \
      "  v_and_b32 v100, 3, v100\n" \
      "  ds_bpermute_b32 v134, v101, %[pfa]\n" \
      "  ds_bpermute_b32 v135, v101, %[pfa] offset:4\n" \
      "  ds_bpermute_b32 v136, v101, %[pfa] offset:64\n" \
      "  ds_bpermute_b32 v137, v101, %[pfa] offset:68\n"
#define LZ4HIP32_BUILD_E2 \
      "  v_alignbyte_b32 v102, v135, v134, v100\n" \
      "  v_alignbyte_b32 v104, v137, v136, v100\n" \
      "  v_alignbyte_b32 v103, v135, v135, v100\n" \
      "  v_alignbyte_b32 v105, v137, v137, v100\n" \
      "  v_mad_u64_u32 v[140:141], s[86:87], v102, %[plo], 0\n" \
      "  v_mad_u64_u32 v[142:143], s[86:87], v104, %[plo], 0\n" \
      "  v_mul_u32_u24 v144, 0xcf, v102\n" \
      "  v_mul_u32_u24 v145, 0xcf, v104\n" \
      "  v_mul_u32_u24 v103, 0xbb, v103\n" \
      "  v_mul_u32_u24 v105, 0xbb, v105\n" \
      "  v_add3_u32 v141, v141, v144, v103\n" \
      "  v_add3_u32 v143, v143, v145, v105\n" \
      "  v_alignbit_b32 v106, v141, v140, 28\n" \
      "  v_alignbit_b32 v107, v143, v142, 28\n" \
      "  v_and_b32 v106, 0xfff, v106\n" \
      "  v_and_b32 v107, 0xfff, v107\n" \
      "  v_lshl_add_u32 v106, v106, 3, %[tbl]\n" \
      "  v_lshl_add_u32 v107, v107, 3, %[tbl]\n" \
      "  ds_read_b64 v[110:111], v106\n" \
      "  ds_read_b64 v[114:115], v107\n" \
      "  v_mul_lo_u32 v108, v102, %[kmul]\n" \
      "  v_mul_lo_u32 v112, v104, %[kmul]\n" \
      "  v_add_u32 v109, s70, %[lane]\n" \
      "  v_lshrrev_b32 v108, 16, v108\n" \
      "  v_lshrrev_b32 v112, 16, v112\n" \
      "  v_add_u32 v113, 64, v109\n" \
      "  s_waitcnt lgkmcnt(0)\n" \
      "  v_cmp_eq_u32_e64 s[78:79], v110, v108\n" \
      "  v_add_u32 v144, 0xffff, v111\n" \
      "  v_cmp_eq_u32_e64 s[80:81], v114, v112\n" \
      "  v_add_u32 v145, 0xffff, v115\n" \
      "  v_cmp_ge_u32_e64 s[86:87], v144, v109\n" \
      "  v_cmp_ge_u32_e64 s[88:89], v145, v113\n" \
      "  s_and_b64 s[78:79], s[78:79], s[86:87]\n" \
      "  s_and_b64 s[80:81], s[80:81], s[88:89]\n"
#define LZ4HIP32_PARK \
      "  s_mov_b32 m0, %[pc]\n" \
      "  s_sub_u32 s86, s72, s73\n" \
      "  v_writelane_b32 %[pms], s72, m0\n" \
      "  v_writelane_b32 %[pml], s74, m0\n" \
      "  v_writelane_b32 %[pof], s86, m0\n" \
      "  s_add_u32 %[pc], %[pc], 1\n"
#define LZ4HIP32_SELECT(tag, nohit) \
      "  s_lshl_b64 s[86:87], s[94:95], s75\n" \
      "  s_and_b64 s[88:89], s[86:87], s[78:79]\n" \
      "  s_cbranch_scc0 L_selB" tag "_%=\n" \
      "  s_ff1_i32_b64 s71, s[88:89]\n" \
      "  v_readlane_b32 s77, v111, s71\n" \
      "L_req" tag "_%=:\n" \
      "  s_add_u32 s75, s70, s71\n" \
      "  v_add_u32 v129, s77, %[j4]\n" \
      "  global_load_dword v127, v129, %[src]\n" \
      "  v_add_u32 v129, s75, %[j4]\n" \
      "  global_load_dword v126, v129, %[src]\n"
#define LZ4HIP32_SELECT_B(tag, nohit) \
      "L_selB" tag "_%=:\n" \
      "  s_sub_u32 s77, 62, s76\n" \
      "  s_lshr_b64 s[88:89], s[94:95], s77\n" \
      "  s_and_b64 s[88:89], s[88:89], s[80:81]\n" \
      "  s_cbranch_scc0 " nohit "_%=\n" \
      "  s_ff1_i32_b64 s71, s[88:89]\n" \
      "  v_readlane_b32 s77, v115, s71\n" \
      "  s_add_u32 s71, s71, 64\n" \
      "  s_branch L_req" tag "_%=\n"

__device__ __forceinline__ uint32_t lean_asm_run32(uint32_t& ip, uint32_t& php, uint32_t& pfe, uint32_t& pc, uint32_t& pfa,
                                                   uint32_t& pms, uint32_t& pml, uint32_t& pof, uint32_t lim, const uint8_t* src,
                                                   uint32_t tbl, uint32_t n) {
  uint32_t code;
  const uint32_t lane = __lane_id();
  const uint32_t j4 = lane * 4u, j16 = lane * 16u;
  const uint32_t kmul = 2654435761u, plo = 0x1bbcdcbbu, ntop = n - 16u;
  asm volatile(
      // ---- entry: post-match state at ip, the row %[pfa] holds the block's bytes from php on
      "  s_cmp_gt_u32 %[ip], %[lim]\n"
      "  s_cbranch_scc1 L_x1_%=\n"
      "  s_cmp_eq_u32 %[pc], 63\n"
      "  s_cbranch_scc1 L_x2_%=\n"
      "  s_sub_u32 s75, %[ip], %[php]\n"
      "  s_add_u32 s75, s75, -2\n"
      "  s_cmpk_gt_u32 s75, 124\n"
      "  s_cbranch_scc1 L_x2_%=\n"
      "  s_lshr_b64 s[94:95], -1, 1\n"
      "  s_add_u32 s70, %[ip], -2\n"
      "  s_mov_b32 s91, 0\n"
      "  s_mov_b32 s90, %[lim]\n"
      LZ4HIP32_BUILD_E1
      "  s_waitcnt lgkmcnt(0)\n"
      LZ4HIP32_BUILD_E2
      "  s_mov_b32 s76, 0\n"
      "  s_mov_b32 s75, 2\n"
      LZ4HIP32_SELECT("p", "L_x2")
      "  s_branch L_commit_%=\n"
      LZ4HIP32_SELECT_B("p", "L_x2")
      // ---- top of the loop: the rows of the hit at s72 (candidate s73) arrive
      "L_arrive_%=:\n"
      "  s_waitcnt vmcnt(0)\n"
      "L_arrived_%=:\n"
      "  v_xor_b32 v128, v126, v127\n"
      "  v_cmp_ne_u32_e32 vcc, 0, v128\n"
      "  v_ffbl_b32 v128, v128\n"
      "  v_lshrrev_b32 v128, 3, v128\n"
      "  v_lshl_add_u32 v128, %[lane], 2, v128\n"
      "  v_mov_b32 %[pfa], v126\n"
      "  s_ff1_i32_b64 s75, vcc\n"
      "  v_readlane_b32 s74, v128, s75\n"     // forward length
      "  s_add_u32 s76, s74, -4\n"
      "  s_add_u32 s75, s74, -2\n"
      LZ4HIP32_SELECT("l", "L_nohit")
      "  s_cbranch_vccz L_f1_%=\n"
      "  s_cmpk_gt_u32 s76, 61\n"
      "  s_cbranch_scc1 L_odd_%=\n"
      "  s_add_u32 %[ip], s72, s74\n"
      "  s_cmp_gt_u32 %[ip], s90\n"
      "  s_cbranch_scc1 L_clean_%=\n"
      "  s_mov_b32 %[php], s72\n"
      // ---- in the shadow of the request: the window's slots up to the hit slot s71 commit their inserts
      "L_commit_%=:\n"
      "  s_lshl_b64 s[82:83], 1, s76\n"
      "  s_or_b64 s[82:83], s[82:83], s[86:87]\n"
      "  s_lshl_b64 s[88:89], -2, s71\n"
      "  s_cmp_lt_u32 s71, 64\n"
      "  s_cbranch_scc0 L_maskB_%=\n"
      "  s_andn2_b64 s[82:83], s[82:83], s[88:89]\n"
      "  s_mov_b64 s[84:85], 0\n"
      "L_masked_%=:\n"
      "  s_mov_b64 exec, s[82:83]\n"
      "  ds_max_rtn_u64 v[116:117], v106, v[108:109]\n"
      "  s_mov_b64 exec, s[84:85]\n"
      "  ds_max_rtn_u64 v[118:119], v107, v[112:113]\n"
      "  s_mov_b64 exec, -1\n"
      "  s_sub_u32 s86, s75, %[php]\n"
      "  s_add_u32 s70, s75, 2\n"
      "  s_add_u32 s86, s86, 2\n"
      "  v_add_u32 v100, s86, %[lane]\n"
      "  v_and_b32 v101, -4, v100\n"
      "  v_and_b32 v100, 3, v100\n"
      "  ds_bpermute_b32 v134, v101, %[pfa]\n"
      "  ds_bpermute_b32 v135, v101, %[pfa] offset:4\n"
      "  ds_bpermute_b32 v136, v101, %[pfa] offset:64\n"
      "  ds_bpermute_b32 v137, v101, %[pfa] offset:68\n"
      "  s_cmpk_gt_u32 s86, 124\n"
      "  s_cselect_b32 s90, 0, s90\n"          // the row does not reach: leave at the next clean point
      "  s_cmp_eq_u32 s91, 0\n"
      "  s_cbranch_scc1 L_nopark_%=\n"
      LZ4HIP32_PARK
      "  s_cmp_eq_u32 %[pc], 63\n"
      "  s_cselect_b32 s90, 0, s90\n"          // the parked registers will be full: leave at the next clean point
      "L_nopark_%=:\n"
      "  s_mov_b32 s91, 1\n"
      "  s_mov_b32 s72, s75\n"
      "  s_mov_b32 s73, s77\n"
      "  v_mov_b32 v120, v106\n"
      "  v_mov_b32 v121, v107\n"
      "  s_waitcnt lgkmcnt(0)\n"
      // a slot that got back another slot's entry: two committing slots share a bucket
      "  v_cmp_ne_u64_e64 s[86:87], v[116:117], v[110:111]\n"
      "  v_cmp_ne_u64_e64 s[88:89], v[118:119], v[114:115]\n"
      "  s_and_b64 s[86:87], s[86:87], s[82:83]\n"
      "  s_and_b64 s[88:89], s[88:89], s[84:85]\n"
      "  s_or_b64 vcc, s[86:87], s[88:89]\n"
      "  s_cbranch_scc1 L_coll_%=\n"
      "L_cont_%=:\n"
      "  v_mov_b64 v[122:123], v[110:111]\n"
      "  v_mov_b64 v[124:125], v[114:115]\n"
      LZ4HIP32_BUILD_E2
      "  s_add_u32 s75, s72, 1024\n"
      "  s_cmp_gt_u32 s75, %[pfe]\n"
      "  s_cbranch_scc0 L_arrive_%=\n"
      // the source is touched 1 KB ahead of the parse (nobody waits for it: it is older than the next rows)
      "  s_cmp_ge_u32 %[pfe], %[n]\n"
      "  s_cbranch_scc1 L_touched_%=\n"
      "  v_add_u32 v129, %[pfe], %[j16]\n"
      "  v_min_u32 v129, %[ntop], v129\n"
      "  global_load_dwordx4 v[130:133], v129, %[src]\n"
      "  s_add_u32 %[pfe], %[pfe], 1024\n"
      "  s_waitcnt vmcnt(1)\n"
      "  s_branch L_arrived_%=\n"
      "L_touched_%=:\n"
      "  s_add_u32 %[pfe], %[pfe], 1024\n"
      "  s_branch L_arrive_%=\n"
      LZ4HIP32_SELECT_B("l", "L_nohit")
      "L_maskB_%=:\n"                          // the hit slot is in group B: all window slots of A commit, B's up to the hit
      "  s_sub_u32 s86, 62, s76\n"
      "  s_lshr_b64 s[84:85], s[94:95], s86\n"
      "  s_andn2_b64 s[84:85], s[84:85], s[88:89]\n"
      "  s_branch L_masked_%=\n"
      // ---- clean exits: the hit in s72..s74 is validated and not parked yet; nothing of the next window is committed
      "L_nohit_%=:\n"
      "  s_cbranch_vccz L_f1_%=\n"
      "  s_cmpk_gt_u32 s76, 61\n"
      "  s_cbranch_scc1 L_odd_%=\n"
      "L_long_%=:\n"
      "  s_add_u32 %[ip], s72, s74\n"
      "L_clean_%=:\n"
      "  s_mov_b32 %[php], s72\n"
      LZ4HIP32_PARK
      "  s_cmp_gt_u32 %[ip], %[lim]\n"
      "  s_cselect_b32 %[code], 1, 2\n"
      "  s_branch L_out_%=\n"
      "L_odd_%=:\n"
      "  s_cmp_lt_u32 s74, 4\n"
      "  s_cbranch_scc0 L_long_%=\n"
      // the hit's candidate differs within its first four bytes, or is equal for 256: its commit is undone from the saved entries
      "L_f1_%=:\n"
      "  s_brev_b32 %[php], 1\n"
      "  s_mov_b64 exec, s[82:83]\n"
      "  ds_write_b64 v120, v[122:123]\n"
      "  s_mov_b64 exec, s[84:85]\n"
      "  ds_write_b64 v121, v[124:125]\n"
      "  s_mov_b64 exec, -1\n"
      "  s_branch L_x2_%=\n"
      // two committing slots share a bucket: the rule of lz4_fast_v2_asm.h (one such slot, foreign fingerprint different from the
      // slot's own, bucket not the hit slot's: carry on; everything else is undone and left to the C++ step)
      "L_coll_%=:\n"
      "  s_bcnt1_i32_b64 s75, vcc\n"
      "  s_cmp_eq_u32 s75, 1\n"
      "  s_cbranch_scc0 L_undo_%=\n"
      "  s_cmp_eq_u64 s[86:87], 0\n"
      "  s_cbranch_scc1 L_collB_%=\n"
      "  s_ff1_i32_b64 s75, s[86:87]\n"
      "  v_readlane_b32 s76, v116, s75\n"
      "  v_readlane_b32 s77, v108, s75\n"
      "  v_readlane_b32 s75, v106, s75\n"
      "  s_branch L_coll2_%=\n"
      "L_collB_%=:\n"
      "  s_ff1_i32_b64 s75, s[88:89]\n"
      "  v_readlane_b32 s76, v118, s75\n"
      "  v_readlane_b32 s77, v112, s75\n"
      "  v_readlane_b32 s75, v107, s75\n"
      "L_coll2_%=:\n"
      "  s_and_b32 s76, s76, 0xffff\n"
      "  s_cmp_eq_u32 s76, s77\n"
      "  s_cbranch_scc1 L_undo_%=\n"
      "  v_readlane_b32 s76, v106, s71\n"
      "  v_readlane_b32 s77, v107, s71\n"
      "  s_cmp_lt_u32 s71, 64\n"
      "  s_cselect_b32 s76, s76, s77\n"
      "  s_cmp_eq_u32 s75, s76\n"
      "  s_cbranch_scc0 L_cont_%=\n"
      "L_undo_%=:\n"
      "  s_mov_b64 exec, s[82:83]\n"
      "  ds_write_b64 v106, v[110:111]\n"
      "  s_mov_b64 exec, s[84:85]\n"
      "  ds_write_b64 v107, v[114:115]\n"
      "  s_mov_b64 exec, -1\n"
      "L_x2_%=:\n"
      "  s_mov_b32 %[code], 2\n"
      "  s_branch L_out_%=\n"
      "L_x1_%=:\n"
      "  s_mov_b32 %[code], 1\n"
      "L_out_%=:\n"
      "  s_waitcnt vmcnt(0) lgkmcnt(0)\n"
      : [ip] "+s"(ip), [php] "+s"(php), [pfe] "+s"(pfe), [pc] "+s"(pc), [pfa] "+v"(pfa), [pms] "+v"(pms), [pml] "+v"(pml),
        [pof] "+v"(pof), [code] "=&s"(code)
      : [lim] "s"(lim), [src] "s"(src), [tbl] "v"(tbl), [n] "s"(n), [ntop] "s"(ntop), [kmul] "s"(kmul), [plo] "s"(plo), [lane] "v"(lane),
        [j4] "v"(j4), [j16] "v"(j16)
      : "memory", "vcc", "scc", "m0", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",
        "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126",
        "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v140", "v141", "v142", "v143", "v144",
        "v145", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86",
        "s87", "s88", "s89", "s90", "s91", "s94", "s95");
  return code;
}
#endif  // __HIP_DEVICE_COMPILE__

}  // namespace lz4hip

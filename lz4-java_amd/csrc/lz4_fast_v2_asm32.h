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
//   v48 shift  v49 permute address  v82/v83 row words of group A  v84/v85 of group B  v50..v53 window words / byte products
//   group A: v54 LDS address  v[56:57] new entry {fp, pos}  v[58:59] entry read      group B: v55, v[60:61], v[62:63]
//   v[64:65] / v[66:67] what the atomics returned   v68/v69 saved addresses, v[70:71] / v[72:73] saved entries (undo)
//   v74 row at the hit  v75 row at the candidate  v76 forward length per lane  v77 address scratch  v[78:81] source touch
//   v[88:91] products  v92/v93 scratch
//   scalars as in lz4_fast_v2_asm.h
#define LZ4HIP32_BUILD_E1 \
      "  v_add_u32 v48, s75, %[lane]\n" \
      "  v_and_b32 v49, -4, v48\n" \
      "  v_and_b32 v48, 3, v48\n" \
      "  ds_bpermute_b32 v82, v49, %[pfa]\n" \
      "  ds_bpermute_b32 v83, v49, %[pfa] offset:4\n" \
      "  ds_bpermute_b32 v84, v49, %[pfa] offset:64\n" \
      "  ds_bpermute_b32 v85, v49, %[pfa] offset:68\n"
#define LZ4HIP32_BUILD_E2 \
      "  v_alignbyte_b32 v50, v83, v82, v48\n" \
      "  v_alignbyte_b32 v52, v85, v84, v48\n" \
      "  v_alignbyte_b32 v51, v83, v83, v48\n" \
      "  v_alignbyte_b32 v53, v85, v85, v48\n" \
      "  v_mad_u64_u32 v[88:89], s[86:87], v50, %[plo], 0\n" \
      "  v_mad_u64_u32 v[90:91], s[86:87], v52, %[plo], 0\n" \
      "  v_mul_u32_u24 v92, 0xcf, v50\n" \
      "  v_mul_u32_u24 v93, 0xcf, v52\n" \
      "  v_mul_u32_u24 v51, 0xbb, v51\n" \
      "  v_mul_u32_u24 v53, 0xbb, v53\n" \
      "  v_add3_u32 v89, v89, v92, v51\n" \
      "  v_add3_u32 v91, v91, v93, v53\n" \
      "  v_alignbit_b32 v54, v89, v88, 28\n" \
      "  v_alignbit_b32 v55, v91, v90, 28\n" \
      "  v_and_b32 v54, 0xfff, v54\n" \
      "  v_and_b32 v55, 0xfff, v55\n" \
      "  v_lshl_add_u32 v54, v54, 3, %[tbl]\n" \
      "  v_lshl_add_u32 v55, v55, 3, %[tbl]\n" \
      "  ds_read_b64 v[58:59], v54\n" \
      "  ds_read_b64 v[62:63], v55\n" \
      "  v_mul_lo_u32 v56, v50, %[kmul]\n" \
      "  v_mul_lo_u32 v60, v52, %[kmul]\n" \
      "  v_add_u32 v57, s70, %[lane]\n" \
      "  v_lshrrev_b32 v56, 16, v56\n" \
      "  v_lshrrev_b32 v60, 16, v60\n" \
      "  v_add_u32 v61, 64, v57\n" \
      "  s_waitcnt lgkmcnt(0)\n" \
      "  v_cmp_eq_u32_e64 s[78:79], v58, v56\n" \
      "  v_add_u32 v92, 0xffff, v59\n" \
      "  v_cmp_eq_u32_e64 s[80:81], v62, v60\n" \
      "  v_add_u32 v93, 0xffff, v63\n" \
      "  v_cmp_ge_u32_e64 s[86:87], v92, v57\n" \
      "  v_cmp_ge_u32_e64 s[88:89], v93, v61\n" \
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
      "  v_readlane_b32 s77, v59, s71\n" \
      "L_req" tag "_%=:\n" \
      "  s_add_u32 s75, s70, s71\n" \
      "  v_add_u32 v77, s77, %[j4]\n" \
      "  global_load_dword v75, v77, %[src]\n" \
      "  v_add_u32 v77, s75, %[j4]\n" \
      "  global_load_dword v74, v77, %[src]\n"
#define LZ4HIP32_SELECT_B(tag, nohit) \
      "L_selB" tag "_%=:\n" \
      "  s_sub_u32 s77, 62, s76\n" \
      "  s_lshr_b64 s[88:89], s[94:95], s77\n" \
      "  s_and_b64 s[88:89], s[88:89], s[80:81]\n" \
      "  s_cbranch_scc0 " nohit "_%=\n" \
      "  s_ff1_i32_b64 s71, s[88:89]\n" \
      "  v_readlane_b32 s77, v63, s71\n" \
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
      "  v_xor_b32 v76, v74, v75\n"
      "  v_cmp_ne_u32_e32 vcc, 0, v76\n"
      "  v_ffbl_b32 v76, v76\n"
      "  v_lshrrev_b32 v76, 3, v76\n"
      "  v_lshl_add_u32 v76, %[lane], 2, v76\n"
      "  v_mov_b32 %[pfa], v74\n"
      "  s_ff1_i32_b64 s75, vcc\n"
      "  v_readlane_b32 s74, v76, s75\n"     // forward length
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
      "  ds_max_rtn_u64 v[64:65], v54, v[56:57]\n"
      "  s_mov_b64 exec, s[84:85]\n"
      "  ds_max_rtn_u64 v[66:67], v55, v[60:61]\n"
      "  s_mov_b64 exec, -1\n"
      "  s_sub_u32 s86, s75, %[php]\n"
      "  s_add_u32 s70, s75, 2\n"
      "  s_add_u32 s86, s86, 2\n"
      "  v_add_u32 v48, s86, %[lane]\n"
      "  v_and_b32 v49, -4, v48\n"
      "  v_and_b32 v48, 3, v48\n"
      "  ds_bpermute_b32 v82, v49, %[pfa]\n"
      "  ds_bpermute_b32 v83, v49, %[pfa] offset:4\n"
      "  ds_bpermute_b32 v84, v49, %[pfa] offset:64\n"
      "  ds_bpermute_b32 v85, v49, %[pfa] offset:68\n"
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
      "  v_mov_b32 v68, v54\n"
      "  v_mov_b32 v69, v55\n"
      "  s_waitcnt lgkmcnt(0)\n"
      // a slot that got back another slot's entry: two committing slots share a bucket
      "  v_cmp_ne_u64_e64 s[86:87], v[64:65], v[58:59]\n"
      "  v_cmp_ne_u64_e64 s[88:89], v[66:67], v[62:63]\n"
      "  s_and_b64 s[86:87], s[86:87], s[82:83]\n"
      "  s_and_b64 s[88:89], s[88:89], s[84:85]\n"
      "  s_or_b64 vcc, s[86:87], s[88:89]\n"
      "  s_cbranch_scc1 L_coll_%=\n"
      "L_cont_%=:\n"
      "  v_mov_b64 v[70:71], v[58:59]\n"
      "  v_mov_b64 v[72:73], v[62:63]\n"
      LZ4HIP32_BUILD_E2
      "  s_add_u32 s75, s72, 1024\n"
      "  s_cmp_gt_u32 s75, %[pfe]\n"
      "  s_cbranch_scc0 L_arrive_%=\n"
      // the source is touched 1 KB ahead of the parse (nobody waits for it: it is older than the next rows)
      "  s_cmp_ge_u32 %[pfe], %[n]\n"
      "  s_cbranch_scc1 L_touched_%=\n"
      "  v_add_u32 v77, %[pfe], %[j16]\n"
      "  v_min_u32 v77, %[ntop], v77\n"
      "  global_load_dwordx4 v[78:81], v77, %[src]\n"
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
      "  ds_write_b64 v68, v[70:71]\n"
      "  s_mov_b64 exec, s[84:85]\n"
      "  ds_write_b64 v69, v[72:73]\n"
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
      "  v_readlane_b32 s76, v64, s75\n"
      "  v_readlane_b32 s77, v56, s75\n"
      "  v_readlane_b32 s75, v54, s75\n"
      "  s_branch L_coll2_%=\n"
      "L_collB_%=:\n"
      "  s_ff1_i32_b64 s75, s[88:89]\n"
      "  v_readlane_b32 s76, v66, s75\n"
      "  v_readlane_b32 s77, v60, s75\n"
      "  v_readlane_b32 s75, v55, s75\n"
      "L_coll2_%=:\n"
      "  s_and_b32 s76, s76, 0xffff\n"
      "  s_cmp_eq_u32 s76, s77\n"
      "  s_cbranch_scc1 L_undo_%=\n"
      "  v_readlane_b32 s76, v54, s71\n"
      "  v_readlane_b32 s77, v55, s71\n"
      "  s_cmp_lt_u32 s71, 64\n"
      "  s_cselect_b32 s76, s76, s77\n"
      "  s_cmp_eq_u32 s75, s76\n"
      "  s_cbranch_scc0 L_cont_%=\n"
      "L_undo_%=:\n"
      "  s_mov_b64 exec, s[82:83]\n"
      "  ds_write_b64 v54, v[58:59]\n"
      "  s_mov_b64 exec, s[84:85]\n"
      "  ds_write_b64 v55, v[62:63]\n"
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
      : "memory", "vcc", "scc", "m0", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59",
        "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74",
        "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v88", "v89", "v90", "v91", "v92",
        "v93", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86",
        "s87", "s88", "s89", "s90", "s91", "s94", "s95");
  return code;
}

#if !LZ4HIP_V2_ASM_PROF
// ---- byU32 blocks of at most 4 MiB with compact entries (FastCore PK: {position 22 bits, fingerprint 10 bits}, 16 KB per table) ----
// 32-bit entries with the position on top: the loop of lz4_fast_v2_asm.h as it is (same registers, same commit, same exits); only the
// lookups differ -- the bucket is the 5-byte hash above, the fingerprint 10 bits of the 4-byte product, and a slot is tentative iff
// the fingerprints agree and its entry lies within 65535 bytes.
//   registers: those of lz4_fast_v2_asm.h minus its topmost one (the kernel of these blocks runs ten pairs per CU, and the ten
//   wavefronts of a workgroup land 3 / 3 / 2 / 2 on the SIMDs, so six must fit on one: 80 VGPRs; the loop ends at v78 and leaves v79
//   to the compiler's scalar spills).  The row-word requests ARE that file's (LZ4HIP_BUILD_E1); the lookups keep the four-byte
//   windows where the fingerprints go (v76 / v77) and borrow v[62:63] and v[70:71] (products) and v50..v53 (byte products, then old
//   fingerprints and distance limits) -- all free between the collision test and the next arrival.
#define LZ4HIP32P_BUILD_E2 \
      "  v_alignbyte_b32 v76, v51, v50, v48\n" \
      "  v_alignbyte_b32 v77, v53, v52, v48\n" \
      "  v_alignbyte_b32 v51, v51, v51, v48\n"     /* the fifth byte of a slot in the low byte */ \
      "  v_alignbyte_b32 v53, v53, v53, v48\n" \
      "  v_mad_u64_u32 v[62:63], s[86:87], v76, %[plo], 0\n" \
      "  v_mad_u64_u32 v[70:71], s[86:87], v77, %[plo], 0\n" \
      "  v_mul_u32_u24 v50, 0xcf, v76\n" \
      "  v_mul_u32_u24 v52, 0xcf, v77\n" \
      "  v_mul_u32_u24 v51, 0xbb, v51\n" \
      "  v_mul_u32_u24 v53, 0xbb, v53\n" \
      "  v_add3_u32 v63, v63, v50, v51\n" \
      "  v_add3_u32 v71, v71, v52, v53\n" \
      "  v_alignbit_b32 v54, v63, v62, 28\n" \
      "  v_alignbit_b32 v58, v71, v70, 28\n" \
      "  v_and_b32 v54, 0xfff, v54\n" \
      "  v_and_b32 v58, 0xfff, v58\n" \
      "  v_lshl_add_u32 v54, v54, 2, %[tbl]\n" \
      "  v_lshl_add_u32 v58, v58, 2, %[tbl]\n" \
      "  ds_read_b32 v55, v54\n" \
      "  ds_read_b32 v59, v58\n" \
      "  v_mul_lo_u32 v76, v76, %[kmul]\n" \
      "  v_mul_lo_u32 v77, v77, %[kmul]\n" \
      "  v_bfe_u32 v76, v76, 16, 10\n" \
      "  v_bfe_u32 v77, v77, 16, 10\n" \
      "  v_lshl_or_b32 v56, v78, 10, v76\n" \
      "  v_lshl_or_b32 v60, v78, 10, v77\n" \
      "  v_add_u32 v60, 0x10000, v60\n"        /* a slot of group B lies 64 positions behind its lane's slot of group A */ \
      "  s_waitcnt lgkmcnt(0)\n" \
      "  v_and_b32 v50, 0x3ff, v55\n" \
      "  v_and_b32 v52, 0x3ff, v59\n" \
      "  v_lshrrev_b32 v57, 10, v55\n" \
      "  v_lshrrev_b32 v61, 10, v59\n" \
      "  v_cmp_eq_u32_e64 s[78:79], v50, v76\n" \
      "  v_cmp_eq_u32_e64 s[80:81], v52, v77\n" \
      "  v_add_u32 v51, 0xffff, v57\n" \
      "  v_add_u32 v53, 0xffbf, v61\n"          /* 65535 - 64 */ \
      "  v_cmp_ge_u32_e64 s[86:87], v51, v78\n" \
      "  v_cmp_ge_u32_e64 s[88:89], v53, v78\n" \
      "  s_and_b64 s[78:79], s[78:79], s[86:87]\n" \
      "  s_and_b64 s[80:81], s[80:81], s[88:89]\n"

__device__ __forceinline__ uint32_t lean_asm_run32p(uint32_t& ip, uint32_t& php, uint32_t& pfe, uint32_t& pc, uint32_t& pfa,
                                                    uint32_t& pms, uint32_t& pml, uint32_t& pof, uint32_t lim, const uint8_t* src,
                                                    uint32_t tbl, uint32_t n) {
  uint32_t code;
  const uint32_t lane = __lane_id();
  const uint32_t j4 = lane * 4u, j16 = lane * 16u;
  const uint32_t kmul = 2654435761u, plo = 0x1bbcdcbbu, ntop = n - 16u;
  asm volatile(
#define LZ4HIP_LEAN_E1(sreg) LZ4HIP_BUILD_E1(sreg)
#define LZ4HIP_LEAN_E2 LZ4HIP32P_BUILD_E2
#define LZ4HIP_LEAN_ROWLIM "124"
#define LZ4HIP_LEAN_FPMASK "0x3ff"
#include "lz4_fast_v2_asm_body.inc"
#undef LZ4HIP_LEAN_E1
#undef LZ4HIP_LEAN_E2
#undef LZ4HIP_LEAN_ROWLIM
#undef LZ4HIP_LEAN_FPMASK
      : [ip] "+s"(ip), [php] "+s"(php), [pfe] "+s"(pfe), [pc] "+s"(pc), [pfa] "+v"(pfa), [pms] "+v"(pms), [pml] "+v"(pml),
        [pof] "+v"(pof), [code] "=&s"(code)
      : [lim] "s"(lim), [src] "s"(src), [tbl] "v"(tbl), [n] "s"(n), [ntop] "s"(ntop), [kmul] "s"(kmul), [plo] "s"(plo), [lane] "v"(lane),
        [j4] "v"(j4), [j16] "v"(j16)
      : "memory", "vcc", "scc", "m0", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61",
        "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78",
#if LZ4HIP_V2_RETRY
        "v79", "v80",
#endif
        "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87",
        "s88", "s89", "s90", "s91", "s94", "s95" LZ4HIP_RETRY_CLOBBERS);
  return code;
}
#endif
#endif  // __HIP_DEVICE_COMPILE__

}  // namespace lz4hip

// group_dev.h -- gfx950 backend of the lane-group interface used by lz4_decode_core.h.
// GL lanes (a power of two, 4..64) of one wavefront decode one block; the wavefront's other
// 64/GL groups decode other blocks in the same instruction stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef LZ4HIP_NT_MATCH
#define LZ4HIP_NT_MATCH 0   // 1: non-temporal loads for match sources (developer A/B builds; round 1 default)
#endif
#if LZ4HIP_NT_MATCH
#define LZ4HIP_MATCH_LOAD(p) __builtin_nontemporal_load(p)
#else
#define LZ4HIP_MATCH_LOAD(p) (*(p))
#endif
namespace lz4hip {
#ifdef LZ4HIP_RING_DBG
__device__ unsigned long long g_ring_stat[8];   // developer build: trips, stalled trips, frozen trips, (re-)seeds, trips spent waiting for another block, loop entries
#endif

// KW: bytes of the output ring of the ring loop (lz4_decode_ring.h; 0 = the other loops)
template <int GL, int KW = 0>
struct GroupDev {
  uint32_t l;  // lane index inside the group
  __device__ __forceinline__ GroupDev() : l(threadIdx.x & (GL - 1)) {}

  __device__ __forceinline__ static uint32_t ld8(const uint8_t* p) { return *p; }
  __device__ __forceinline__ static uint32_t ld16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
  __device__ __forceinline__ static uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
  __device__ __forceinline__ static uint64_t ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

  // d[0..len) = s[0..len); wild: 4 bytes per lane, may read/write up to 3 bytes past len
  __device__ __forceinline__ void copy_lits(uint8_t* d, const uint8_t* s, uint32_t len, bool wild) const {
    if (wild) {
      for (uint32_t i = l * 4u; i < len; i += 4u * GL) {
        uint32_t v;
        __builtin_memcpy(&v, s + i, 4);
        __builtin_memcpy(d + i, &v, 4);
      }
    } else {
      for (uint32_t i = l; i < len; i += GL) d[i] = s[i];
    }
  }

  // Wide variants for the decoder's interior loop (>= 300 bytes of slack on every side): each lane moves
  // LB = 64/GL bytes per step, so a step always covers 64 bytes whatever the group size -- smaller groups put more
  // blocks behind every instruction of the (issue-bound) loop.  May touch up to LB-1 bytes past len.
  static constexpr uint32_t LB = GL == 1 ? 16u : (64u / GL < 4u ? 4u : 64u / GL);   // (GL == 1: the ring loop's lane-per-block form, 16-byte steps)
  template <int N> struct Chunk { uint32_t w[N]; };
  typedef uint32_t vecLB __attribute__((ext_vector_type(LB / 4), aligned(1)));
  __device__ __forceinline__ static void store_out(uint8_t* p, const Chunk<LB / 4>& v) {
    __builtin_memcpy(p, &v, LB);  // (non-temporal stores measured 2x SLOWER: the output is re-read as match source)
  }
  __device__ __forceinline__ void copy_lits_wide(uint8_t* d, const uint8_t* s, uint32_t len) const {
    for (uint32_t i = l * LB; i < len; i += LB * GL) {
      Chunk<LB / 4> v;
      __builtin_memcpy(&v, s + i, LB);
      store_out(d + i, v);
    }
  }
  __device__ __forceinline__ void copy_match_wide(uint8_t* dst, uint32_t op, uint32_t offset, uint32_t len) const {
    if (offset >= LB * GL) {  // a step reads [m+i0, m+i0+64) <= d+i0: only bytes stored by earlier instructions
      uint8_t* d = dst + op;
      const uint8_t* m = d - offset;
      for (uint32_t i = l * LB; i < len; i += LB * GL) {
        Chunk<LB / 4> v;
        // (round 1 loaded match sources non-temporally -- random lines that would displace the streams in L2, +3..8 % then.
        // With output staging and whole-line stores in place that reversed: plain loads are +6 % on App. F blocks, +4 % on 4 MiB
        // blocks and +20..50 % on text, where a line is the source of several matches; LZ4HIP_NT_MATCH keeps the A/B)
        const vecLB t = LZ4HIP_MATCH_LOAD((const vecLB*)(m + i));
        __builtin_memcpy(&v, &t, LB);
        store_out(d + i, v);
      }
    } else {
      copy_match(dst, op, offset, len, true);
    }
  }

  // Split sequence copy for the pipelined interior loop (lz4_decode_core.h PIPE): a "simple" sequence -- literals and match
  // each fit one step, the match does not overlap its own output -- is LOADED in one trip of the loop and STORED in the next,
  // behind the next sequence's loads.  A wavefront's memory operations retire in order, so a wait for a load also waits for
  // every older store; with the stores of sequence t issued after the loads of sequence t+1, no wait ever covers a store.
  struct SeqRegs { Chunk<LB / 4> v, u; };
  __device__ __forceinline__ static constexpr uint32_t step() { return LB * GL; }   // bytes one step covers
  __device__ __forceinline__ static constexpr uint32_t slack() { return LB; }   // bytes a wild step may touch past the end
  __device__ __forceinline__ void seq_load(SeqRegs& r, const uint8_t* s, uint32_t lit, const uint8_t* m, uint32_t len) const {
    const uint32_t i = l * LB;
    if (i < lit) __builtin_memcpy(&r.v, s + i, LB);
    if (i < len) {
      const vecLB t = LZ4HIP_MATCH_LOAD((const vecLB*)(m + i));
      __builtin_memcpy(&r.u, &t, LB);
    }
  }
  __device__ __forceinline__ void seq_store(const SeqRegs& r, uint8_t* d, uint32_t lit, uint32_t len) const {
    const uint32_t i = l * LB;
    if (i < lit) store_out(d + i, r.v);
    if (i < len) store_out(d + lit + i, r.u);
  }

  // ---- output staging in LDS (lz4_decode_core.h STAGE): the interior loop's output goes to a per-block staging buffer and
  // leaves it as whole 128-byte lines.  Why: with every block of a big batch in flight, the partially written line of each
  // block is evicted from L2 between the small stores that fill it and reaches HBM several times (WRITE_SIZE 11.8 GB for
  // 4.3 GB of output, profiles/r01n).  Staging holds the output bytes [fl, fl + kStage); everything below fl is in memory.
  // Match sources are read from memory only, so the caller flushes everything first when a source reaches past fl. ----
#ifndef LZ4HIP_RUN_TIERS
#define LZ4HIP_RUN_TIERS 1    /* the wave loops' copy rounds in three forms by their longest run (< 16 / < 32 / <= 64 bytes); 0: always the 64-byte form (developer A/B builds) */
#endif
#ifndef LZ4HIP_PRED_CASCADE
#define LZ4HIP_PRED_CASCADE 1   /* 0: only the 16-byte pieces of a round are predicated (developer A/B builds) */
#endif
#ifndef LZ4HIP_TIER_GUARD
#define LZ4HIP_TIER_GUARD 0   /* 1: the short forms take their operands through an empty asm -- their address math stays in their branch instead of in front of the round loop (developer A/B builds) */
#endif
#ifndef LZ4HIP_KSTAGE
#define LZ4HIP_KSTAGE 576   /* bytes of staging per block (developer A/B builds: smaller = more blocks resident per CU, more flushes) */
#endif
  static constexpr uint32_t kStage = LZ4HIP_KSTAGE;   // >= 127 (open line) + what two sequences of the interior loop typically add + a step + slack
  uint8_t* stg = nullptr;
  uint32_t fl = 0;
  __device__ __forceinline__ void st_begin(uint8_t* lds, uint32_t op) { stg = lds; fl = op; }
  // whole lines below p (p = end of the valid staged bytes) go to memory; the open line moves to the front
  __device__ __forceinline__ void st_flush_lines(uint8_t* dst, uint32_t p) {
    const uint32_t target = p - (((uint32_t)(uintptr_t)dst + p) & 127u);   // last position <= p at a 128-byte address
    if ((int32_t)(target - fl) <= 0) return;
    const uint32_t nb = target - fl, rem = p - target;
    for (uint32_t o = l * LB; o < nb; o += LB * GL) {   // (may carry up to LB-1 staged bytes past target: written again by the next flush)
      Chunk<LB / 4> v;
      __builtin_memcpy(&v, stg + o, LB);
      store_out(dst + fl + o, v);
    }
    for (uint32_t o = l * LB; o < rem; o += LB * GL) {  // shift down by nb (forward, one instruction's loads before its stores)
      Chunk<LB / 4> v;
      __builtin_memcpy(&v, stg + nb + o, LB);
      __builtin_memcpy(stg + o, &v, LB);
    }
    fl = target;
  }
  // everything below p goes to memory (up to LB-1 bytes past p are touched: output positions that are written again later)
  __device__ __forceinline__ void st_flush_all(uint8_t* dst, uint32_t p) {
    if ((int32_t)(p - fl) <= 0) return;   // (st_lits may have flushed lines past p: a sequence that then leaves the loop has nothing left to flush)
    const uint32_t nb = p - fl;
    for (uint32_t o = l * LB; o < nb; o += LB * GL) {
      Chunk<LB / 4> v;
      __builtin_memcpy(&v, stg + o, LB);
      store_out(dst + fl + o, v);
    }
    fl = p;
  }
  __device__ __forceinline__ void st_lits(uint8_t* dst, uint32_t op, const uint8_t* s, uint32_t len) {
    for (uint32_t base = 0; base < len; base += LB * GL) {
      if (op + base - fl + LB * GL + LB > kStage) st_flush_lines(dst, op + base);
      const uint32_t i = base + l * LB;
      if (i < len) {
        Chunk<LB / 4> v;
        __builtin_memcpy(&v, s + i, LB);
        __builtin_memcpy(stg + (op + i - fl), &v, LB);
      }
    }
  }
  // staged[op+i] = memory[op-offset+i]; the caller guarantees offset >= step() and (op - offset) + len + slack() <= fl
  __device__ __forceinline__ void st_match(uint8_t* dst, uint32_t op, uint32_t offset, uint32_t len) {
    const uint8_t* m = dst + op - offset;
    for (uint32_t base = 0; base < len; base += LB * GL) {
      if (op + base - fl + LB * GL + LB > kStage) st_flush_lines(dst, op + base);
      const uint32_t i = base + l * LB;
      if (i < len) {
        Chunk<LB / 4> v;
        const vecLB t = LZ4HIP_MATCH_LOAD((const vecLB*)(m + i));
        __builtin_memcpy(&v, &t, LB);
        __builtin_memcpy(stg + (op + i - fl), &v, LB);
      }
    }
  }

  // ---- backend of the deep interior loop (lz4_decode_deep.h): the block's window of the compressed stream in an LDS ring of
  // kStream bytes, refilled in 256-byte pieces (each lane CB bytes), its first 16 bytes repeated behind the end so that no read
  // wraps; whole 64-byte steps loaded / stored by all lanes of the group. ----
#ifndef LZ4HIP_KSTREAM
#define LZ4HIP_KSTREAM 1024   /* stream bytes per block in LDS (power of two, >= 1024) */
#endif
  static constexpr uint32_t kStream = LZ4HIP_KSTREAM;
  static constexpr uint32_t kStreamLds = kStream + 16u;
  static constexpr uint32_t kPiece = 256u;
  static constexpr uint32_t CB = kPiece / GL < 4u ? 4u : kPiece / GL;   // bytes of a piece per lane (the deep loop runs with GL <= 16)
  struct PieceRegs { uint32_t w[CB / 4]; };
  typedef Chunk<LB / 4> LChunk;
  uint8_t* srb = nullptr;
  __device__ __forceinline__ void sr_begin(uint8_t* lds) { srb = lds; }
  __device__ __forceinline__ PieceRegs sr_fetch(const uint8_t* src, uint32_t pos) const {   // pos: multiple of 256; the whole piece is readable
    PieceRegs r;
    __builtin_memcpy(&r, src + pos + l * CB, CB);
    return r;
  }
  __device__ __forceinline__ void sr_put(uint32_t pos, const PieceRegs& r) {
    const uint32_t q = (pos & (kStream - 1u)) + l * CB;
    __builtin_memcpy(srb + q, &r, CB);
    if (q < 16u) __builtin_memcpy(srb + kStream + q, &r, CB < 16u ? CB : 16u);
  }
  // (round 4: the reads below fetch whole ALIGNED dwords and funnel them -- an LDS instruction whose lanes are not aligned to its
  // width is served one lane per cycle, 64 cycles for a wavefront instead of 2..8: tools/ubench/lds_unaligned.hip)
  __device__ __forceinline__ uint32_t sr_ld32(uint32_t p) const {
    const uint32_t* q = (const uint32_t*)__builtin_assume_aligned(srb + (p & (kStream - 1u) & ~3u), 4);
    return __builtin_amdgcn_alignbyte(q[1], q[0], p & 3u);
  }
  __device__ __forceinline__ uint64_t sr_ld64(uint32_t p) const {
    const uint32_t* q = (const uint32_t*)__builtin_assume_aligned(srb + (p & (kStream - 1u) & ~3u), 4);
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], s = p & 3u;
    return (uint64_t)__builtin_amdgcn_alignbyte(d1, d0, s) | ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, s) << 32);
  }
  __device__ __forceinline__ Chunk<LB / 4> sr_step(uint32_t p) const {   // this lane's LB bytes of the 64 stream bytes at p
    const uint32_t* q = (const uint32_t*)__builtin_assume_aligned(srb + ((p + l * LB) & (kStream - 1u) & ~3u), 4);
    uint32_t d[LB / 4u + 1u];
#pragma unroll
    for (uint32_t k = 0; k <= LB / 4u; k++) d[k] = q[k];
    Chunk<LB / 4> v;
#pragma unroll
    for (uint32_t k = 0; k < LB / 4u; k++) v.w[k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], p & 3u);
    return v;
  }
  __device__ __forceinline__ Chunk<LB / 4> step_load(const uint8_t* m) const {   // this lane's LB bytes of the 64 bytes at m
    Chunk<LB / 4> v;
    const vecLB t = LZ4HIP_MATCH_LOAD((const vecLB*)(m + l * LB));
    __builtin_memcpy(&v, &t, LB);
    return v;
  }
  __device__ __forceinline__ void step_store(uint8_t* d, const Chunk<LB / 4>& v) const { store_out(d + l * LB, v); }
  __device__ __forceinline__ uint32_t lane_bytes() const { return l * LB; }   // this lane's offset inside a step
  // the same, but lanes whose LB bytes lie at or beyond `len` read from `idle` (a step that is cached) instead
  __device__ __forceinline__ Chunk<LB / 4> step_load_upto(const uint8_t* m, uint32_t len, const uint8_t* idle) const {
    return step_load(l * LB < len ? m : idle);
  }

  // ---- backend of the ring loop (lz4_decode_ring.h): the block's window of the compressed stream AND its recent output in LDS.
  // Layout of a block's kRingLds bytes: [stream ring kRs + max(16, LB)][LB pad | output ring KW | 2 LB tail].
  //  * stream ring: indexed by stream position, refilled in aligned 64-byte steps (one chunk per lane), first 16 bytes mirrored
  //    behind the end so that no read wraps (per-lane masking: a lane's chunk wraps by itself);
  //  * output ring: index of output position p = (p + dbase) & (KW - 1), dbase = dst address & 63, so that 64-byte aligned steps
  //    of MEMORY are aligned steps of the ring (the flusher's reads and stores are aligned, a line never straddles the end).
  //    A chunk written at ring index x is stored at t - LB with t = (x + LB) & (KW - 1), and once more KW bytes further on when
  //    t < 2 LB: a chunk that straddles the end lands in [x - KW, ..) (the pad + the ring's first bytes) AND in [x, KW + LB) (the
  //    ring's last bytes + the tail); a chunk inside the first LB bytes is repeated in the tail.  So a read of LB bytes at any
  //    index finds them contiguous, and there is no wrap branch anywhere. ----
#ifndef LZ4HIP_RING_KSTREAM
#define LZ4HIP_RING_KSTREAM 256
#endif
  static constexpr uint32_t kRs = LZ4HIP_RING_KSTREAM;      // stream bytes per block (power of two, >= 256)
  static constexpr uint32_t kRing = KW ? (uint32_t)KW : 512u;
  static constexpr uint32_t kRsTail = LB < 16u ? 16u : LB;   // stream bytes repeated behind the ring's end (a lane reads LB + 4 from a masked index)
  static constexpr uint32_t WB = GL == 1 ? LB + 4u : LB;     // bytes a lane stores at once (GL == 1: the window of five aligned dwords around its 16 bytes)
  static constexpr uint32_t kPad = GL == 1 ? 32u : LB;           // bytes in front of the ring's index 0 (a straddling store's lower copy; 16-byte aligned ring)
  static constexpr uint32_t kRingLds0 = (kRs + kRsTail + kPad + kRing + 2u * WB + 15u) & ~15u;
  static constexpr uint32_t kRingLds = kRingLds0 + (((kRingLds0 >> 4) & 1u) ? 0u : 16u);   // an odd number of 16-byte units: the blocks of a wavefront start in different banks
  uint8_t* rsb = nullptr;   // stream ring
  uint8_t* rgb = nullptr;   // output ring, index 0 (behind the pad)
  uint32_t dbase = 0;
  __device__ __forceinline__ void ring_begin(uint8_t* lds, const uint8_t* dst) {
    rsb = lds; rgb = lds + kRs + kRsTail + kPad; dbase = (uint32_t)(uintptr_t)dst & 63u;
  }
  __device__ __forceinline__ static constexpr uint32_t ring_bytes() { return kRing; }
  __device__ __forceinline__ static constexpr uint32_t ring_step() { return GL * LB; }                       // bytes of a step: 64, or 16 with a lane per block
  __device__ __forceinline__ static constexpr uint32_t ring_piece() { return GL == 1 ? 16u : GL * LB - 4u; }   // bytes of literals / of match a trip emits at most
  __device__ __forceinline__ static constexpr uint32_t ring_stream() { return kRs; }
  __device__ __forceinline__ uint32_t ring_dbase() const { return dbase; }
  // Every LDS access below is NATURALLY ALIGNED, and what the bytes' real position asks for is done in registers.  Measured
  // (tools/ubench/lds_unaligned.hip, lds_masked.hip; profiles/r04_lds_alignment.txt): an LDS instruction whose lanes are not aligned to
  // the access width (b32: 4, b64: 8, b128: 16) is served ONE LANE PER CYCLE -- 64 LDS cycles for a wavefront's unaligned read or
  // write against 2..8 for an aligned one, ~45 with twelve wavefronts per CU contending; with sixteen wavefronts that alone is more
  // than a thousand cycles per trip and instruction.  So: reads fetch whole dwords (LB / 4 + 1 of them from the dword below the
  // position) and funnel them with v_alignbyte; writes store the step as aligned dwords by lanes 1.. of the group (each lane's chunk
  // moved up by the position's byte offset: one dword from the lane below, v_perm for the rest) and only lane 0 -- whose chunk would
  // need the ring's old bytes in front of the position -- stores its LB bytes where they belong (4 .. 16 unaligned lanes per
  // wavefront: 4 .. 16 cycles).  A step written at position w covers [w, w + 64 - (w & 3)): pieces are at most 60 bytes long.
  __device__ __forceinline__ static const uint32_t* dwp(const uint8_t* p) { return (const uint32_t*)__builtin_assume_aligned(p, 4); }
  __device__ __forceinline__ static uint32_t* dwp(uint8_t* p) { return (uint32_t*)__builtin_assume_aligned(p, 4); }
  // LB bytes at byte offset s (0..3) of the LB / 4 + 1 dwords d[]
  __device__ __forceinline__ static LChunk funnel(const uint32_t* d, uint32_t s) {
    LChunk v;
#pragma unroll
    for (uint32_t k = 0; k < LB / 4u; k++) v.w[k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], s);
    return v;
  }
  // stream side
  __device__ __forceinline__ LChunk rs_fetch(const uint8_t* src, uint32_t pos) const {   // pos: multiple of 64; [pos, pos + 64) is readable
    LChunk r;
    __builtin_memcpy(&r, src + pos + l * LB, LB);
    return r;
  }
  __device__ __forceinline__ void rs_put(uint32_t pos, const LChunk& r) {   // (aligned: pos is a multiple of 64, the lane's chunk of LB)
    const uint32_t q = (pos & (kRs - 1u)) + l * LB;
    __builtin_memcpy(__builtin_assume_aligned(rsb + q, LB < 16u ? LB : 16u), &r, LB);
    if (q < kRsTail) __builtin_memcpy(__builtin_assume_aligned(rsb + kRs + q, LB < 16u ? LB : 16u), &r, LB);
  }
  __device__ __forceinline__ uint64_t rs_ld64(uint32_t p) const {   // the 8 stream bytes at p (every lane of the group the same)
    const uint32_t* q = dwp(rsb + (p & (kRs - 1u) & ~3u));
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], s = p & 3u;
    return (uint64_t)__builtin_amdgcn_alignbyte(d1, d0, s) | ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, s) << 32);
  }
  __device__ __forceinline__ LChunk rs_step(uint32_t p) const {     // this lane's LB bytes of the 64 stream bytes at p
    const uint32_t* q = dwp(rsb + ((p + l * LB) & (kRs - 1u) & ~3u));
    uint32_t d[LB / 4u + 1u];
#pragma unroll
    for (uint32_t k = 0; k <= LB / 4u; k++) d[k] = q[k];
    return funnel(d, p & 3u);
  }
  // output side
  __device__ __forceinline__ LChunk rg_read(uint32_t pos) const {   // this lane's LB bytes of the 64 output bytes at pos
    const uint32_t x = pos + dbase + l * LB;
    const uint32_t* q = dwp(rgb + (x & (kRing - 1u) & ~3u));
    uint32_t d[LB / 4u + 1u];
#pragma unroll
    for (uint32_t k = 0; k <= LB / 4u; k++) d[k] = q[k];
    return funnel(d, x & 3u);
  }
  __device__ __forceinline__ LChunk rg_read_al(uint32_t pos) const {   // the same for an aligned step (the flusher's)
    LChunk v;
    __builtin_memcpy(&v, __builtin_assume_aligned(rgb + ((pos + dbase + l * LB) & (kRing - 1u)), LB < 16u ? LB : 16u), LB);
    return v;
  }
  // chunk c (LB bytes, any alignment or dword alignment -- the caller knows) to ring index x: at t - LB with t = (x + LB) & (KW - 1),
  // and once more KW bytes on when t < 2 LB (the mirror rule above)
  __device__ __forceinline__ void rg_put(uint32_t x, const LChunk& c, bool aligned) {
    const uint32_t t = (x + LB) & (kRing - 1u);
    uint8_t* a = (rgb - LB) + t;
    if (aligned) {
      __builtin_memcpy(__builtin_assume_aligned(a, 4), &c, LB);
#ifndef LZ4HIP_RING_PROBE_NOMIRROR   /* developer timing probe, NOT bit-exact: what the mirror stores' tests cost */
      if (t < 2u * LB) __builtin_memcpy(__builtin_assume_aligned(a + kRing, 4), &c, LB);
#endif
    } else {
      __builtin_memcpy(a, &c, LB);
#ifndef LZ4HIP_RING_PROBE_NOMIRROR
      if (t < 2u * LB) __builtin_memcpy(a + kRing, &c, LB);
#endif
    }
  }
  // GL == 1: the lane keeps the five aligned dwords it stored last (cw, at ring position cbase).  Output is written in order, so the
  // bytes in front of the next position are always in that window: the next store is again five ALIGNED dwords, its first one made
  // of the window's bytes below the position and the new bytes -- no lane of the wavefront ever stores unaligned.
  uint32_t cw[5] = {0, 0, 0, 0, 0};
  uint32_t cbase = 0;
  // the 64-byte step v (lane j holds bytes [j LB, j LB + LB)) to output position pos: covers [pos, pos + 64 - s), s = ring index & 3
  // (GL == 1: the lane's 16 bytes, all of them)
  __device__ __forceinline__ void rg_write(uint32_t pos, const LChunk& v) {
    const uint32_t w = pos + dbase, s = w & 3u;
    const uint32_t sel = 0x03020100u + (4u - s) * 0x01010101u;   // v_perm selector: bytes [4 - s, 8 - s) of {hi, lo}
    if constexpr (GL == 1) {
      const uint32_t k = ((w & ~3u) - cbase) >> 2;               // the window dword that holds the bytes below the position (0..4)
      const uint32_t c = k == 0u ? cw[0] : k == 1u ? cw[1] : k == 2u ? cw[2] : k == 3u ? cw[3] : cw[4];
      // first dword: bytes [0, s) of c, then bytes [0, 4 - s) of the new data
      const uint32_t sel0 = 0x03020100u + ((0x04040404u - s * 0x01010101u) & (0xFFFFFFFFu << (8u * s)));
      cw[0] = __builtin_amdgcn_perm(v.w[0], c, sel0);
      cw[1] = __builtin_amdgcn_perm(v.w[1], v.w[0], sel);
      cw[2] = __builtin_amdgcn_perm(v.w[2], v.w[1], sel);
      cw[3] = __builtin_amdgcn_perm(v.w[3], v.w[2], sel);
      cw[4] = __builtin_amdgcn_perm(v.w[3], v.w[3], sel);        // (its upper bytes are bytes that are written again)
      cbase = w & ~3u;
      const uint32_t t = (cbase + WB) & (kRing - 1u);
      uint32_t* a = dwp((rgb - WB) + t);
      a[0] = cw[0]; a[1] = cw[1]; a[2] = cw[2]; a[3] = cw[3]; a[4] = cw[4];
      if (t < 2u * WB) { uint32_t* b = dwp((rgb - WB) + t + kRing); b[0] = cw[0]; b[1] = cw[1]; b[2] = cw[2]; b[3] = cw[3]; b[4] = cw[4]; }
    } else {
      // lanes 1..: the aligned chunk at (w & ~3) + l LB holds the step's bytes [l LB - s, l LB - s + LB): this lane's dwords moved
      // up by s bytes, the lowest bytes from the top dword of the lane below; lane 0: its bytes where they belong (the only
      // unaligned lanes: 64 / GL per wavefront).  ONE index computation and ONE mirror test serve both kinds of lanes (a probe
      // build without the mirror stores -- ten instructions and two conditional stores fewer per trip -- ran 10 % faster).
      const uint32_t below = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.w[LB / 4u - 1u], 0x111 /* row_shr:1 */, 0xF, 0xF, false);
      LChunk m;
#pragma unroll
      for (uint32_t k = 0; k < LB / 4u; k++) m.w[k] = __builtin_amdgcn_perm(v.w[k], k ? v.w[k - 1u] : below, sel);
      const bool l0 = l == 0u;
      const uint32_t t = ((l0 ? w : (w & ~3u) + l * LB) + LB) & (kRing - 1u);
      uint8_t* a = (rgb - LB) + t;
      if (l0) __builtin_memcpy(a, &v, LB); else __builtin_memcpy(__builtin_assume_aligned(a, 4), &m, LB);
#ifndef LZ4HIP_RING_PROBE_NOMIRROR   /* developer timing probe, NOT bit-exact */
      if (__builtin_expect(t < 2u * LB, 0)) {
        if (l0) __builtin_memcpy(a + kRing, &v, LB); else __builtin_memcpy(__builtin_assume_aligned(a + kRing, 4), &m, LB);
      }
#endif
    }
  }
  // (re-)seed: the aligned step at pos (from memory) is the ring's first content
  __device__ __forceinline__ void rg_seed(uint32_t pos, const LChunk& v) {
    if constexpr (GL == 1) { cbase = (pos + dbase) & ~3u; cw[0] = cw[1] = cw[2] = cw[3] = cw[4] = 0u; }
    rg_write(pos, v);
  }
  __device__ __forceinline__ static bool first_active() { return __builtin_amdgcn_mbcnt_hi(__builtin_amdgcn_read_exec_hi(), __builtin_amdgcn_mbcnt_lo(__builtin_amdgcn_read_exec_lo(), 0u)) == 0u; }
  __device__ __forceinline__ static bool any(bool x) { return __builtin_amdgcn_ballot_w64(x) != 0ull; }   // over the wavefront's active lanes
  // every vector memory operation issued so far is waited for HERE (the compiler's wait-count pass sees the instruction: a loop
  // entered behind it has nothing outstanding on its entry edge, so the waits inside count only the loop's own operations)
  __device__ __forceinline__ static void settle(uint32_t&) { __builtin_amdgcn_s_waitcnt(0x0F70); }
  __device__ __forceinline__ static LChunk pick(bool first, const LChunk& a, const LChunk& b) {
    LChunk r;
#pragma unroll
    for (uint32_t k = 0; k < LB / 4u; k++) r.w[k] = first ? a.w[k] : b.w[k];
    return r;
  }

  // dst[op+i] = dst[op-offset+i] for i in [0,len), byte-forward (overlap replicates the pattern)
  __device__ __forceinline__ void copy_match(uint8_t* dst, uint32_t op, uint32_t offset, uint32_t len, bool wild) const {
    uint8_t* d = dst + op;
    const uint8_t* m = d - offset;
    if (wild && offset >= 4u * GL) {
      // one step stores [d+i0, d+i0+4*GL) and reads [m+i0, m+i0+4*GL) <= d+i0: only bytes stored by
      // EARLIER instructions of this wave (or earlier sequences) are read
      for (uint32_t i = l * 4u; i < len; i += 4u * GL) {
        uint32_t v;
        __builtin_memcpy(&v, m + i, 4);
        __builtin_memcpy(d + i, &v, 4);
      }
    } else if (offset == 0) {
      for (uint32_t i = l; i < len; i += GL) d[i] = 0;
    } else {
      // replicate: byte i comes from m[i mod offset], all of which precede the match
      uint32_t r = l < offset ? l : l % offset;
      const uint32_t stp = GL < offset ? (uint32_t)GL : (uint32_t)GL % offset;
      for (uint32_t i = l; i < len; i += GL) {
        d[i] = m[r];
        r += stp;
        if (r >= offset) r -= offset;
      }
    }
  }
};


// ---- backend of the wave loop (lz4_decode_wave.h): ONE WAVEFRONT PER BLOCK.  All 64 lanes move 4 bytes each (a step is 256 bytes, a
// piece up to 252); the block's window of the compressed stream (KS bytes) and its recent output (KW bytes) live in LDS.
// Layout of a wavefront's kWaveLds bytes: [stream ring KS | 16 tail][16 pad | output ring KW | 32 tail].
//  * stream ring: indexed by stream position, refilled in 256-byte steps (a dword per lane), first 16 bytes mirrored behind the end so
//    that the two aligned dwords around any index are contiguous;
//  * output ring: index of output position p = (p + dbase) & (KW - 1), dbase = dst address & 255 -- 256-byte aligned steps of memory
//    are aligned steps of the ring.  A dword written at ring index x is stored at t - 4 with t = (x + 4) & (KW - 1), and once more KW
//    bytes further on when t < 20 (the mirror rule of the ring loop's backend above, for the ring's first 16 bytes): reads of up to 16 bytes never wrap.
// A piece written at ring index w: lane 0 holds its bytes [0, 4) and stores them at w (the only unaligned lane), lane l >= 1 holds the
// bytes [4 l - s, 4 l - s + 4), s = w & 3, and stores them at the aligned index (w & ~3) + 4 l.  The getters deliver a piece in that
// shape for a given s: two aligned dwords around each lane's source index, funnelled (v_alignbyte).
template <int KW, int KS>
struct BlockWaveDev : GroupDev<64, 0> {
  typedef GroupDev<64, 0> Base;
  typedef typename Base::LChunk LChunk;   // one dword
  static_assert(Base::LB == 4u, "a lane of the wave loop moves one dword");
  // (A trip discovers ONE 256-byte window of the stream.  Two windows per trip were built and measured twice -- profiles/r05_wave_notes.txt
  // sections 2 and 4: 21.8 instead of 9.9 sequences per trip on BASELINE configs[2]'s blocks and never faster, 2048 x 4 MiB 26.9 -> 29.1 ms
  // -- and taken out of the source when the discovery was cut down to what the walk needs.)
  static constexpr uint32_t kWaveLds = (uint32_t)KS + 16u + 16u + (uint32_t)KW + 32u;   // (32 tail: the first 16 ring bytes mirrored, plus a dword's reach)
  uint8_t* wsb = nullptr;   // stream ring
  uint8_t* wrb = nullptr;   // output ring, index 0
  uint32_t wdb = 0;
  uint32_t l4;              // 4 * lane
  __device__ __forceinline__ BlockWaveDev() : Base(), l4((threadIdx.x & 63u) * 4u) {}
  __device__ __forceinline__ void wv_begin(uint8_t* lds, const uint8_t* dst) { wsb = lds; wrb = lds + KS + 32; wdb = (uint32_t)(uintptr_t)dst & 255u; }
  __device__ __forceinline__ static constexpr uint32_t wv_ring() { return (uint32_t)KW; }
  __device__ __forceinline__ static constexpr uint32_t wv_stream() { return (uint32_t)KS; }
  __device__ __forceinline__ uint32_t wv_dbase() const { return wdb; }
  __device__ __forceinline__ static uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
  // a piece on its way to ring index w: this lane's dword and its byte offset inside the piece (lane 0: 0; lane l: 4 l - (w & 3))
  struct WPiece { uint32_t v, dl; };
  // (4 l - s saturated at zero is lane 0's 0: ONE instruction, v_sub_u32 with the clamp bit)
  __device__ __forceinline__ uint32_t delta(uint32_t w) const { return __builtin_elementwise_sub_sat(l4, w & 3u); }
  // stream side (shadow the ring loop's: this ring has KS bytes)
  __device__ __forceinline__ LChunk rs_fetch(const uint8_t* src, uint32_t pos) const {   // pos: multiple of 256; [pos, pos + 256) is readable
    LChunk r;
    __builtin_memcpy(&r, src + pos + l4, 4);
    return r;
  }
  // the stream's LAST, partial step (pos: a multiple of 256 with pos < iend < pos + 256): the bytes below iend, zeros behind them -- nothing at
  // or behind src + iend is read (the trio loop's scanner: with it the whole stream is in the ring and the loop runs up to the block's last
  // 306 bytes instead of leaving the last 512 .. 767 to the slower loops behind it)
  __device__ __forceinline__ LChunk rs_fetch_upto(const uint8_t* src, uint32_t pos, uint32_t iend) const {
    LChunk r;
    r.w[0] = 0u;
    const uint32_t a = pos + l4;
    if (a + 4u <= iend) __builtin_memcpy(&r, src + a, 4);
    else {
#pragma unroll
      for (uint32_t k = 0; k < 3u; k++) if (a + k < iend) r.w[0] |= (uint32_t)src[a + k] << (8u * k);
    }
    return r;
  }
  __device__ __forceinline__ void rs_put(uint32_t pos, const LChunk& r) {
    const uint32_t q = (pos & ((uint32_t)KS - 1u)) + l4;
    *Base::dwp(wsb + q) = r.w[0];
    if (q < 16u) *Base::dwp(wsb + KS + q) = r.w[0];
  }
  __device__ __forceinline__ uint64_t rs_ld64(uint32_t p) const {   // the 8 stream bytes at p (every lane the same)
    const uint32_t* q = Base::dwp(wsb + (p & ((uint32_t)KS - 1u) & ~3u));
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], s = p & 3u;
    return (uint64_t)__builtin_amdgcn_alignbyte(d1, d0, s) | ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, s) << 32);
  }
  // getters: the piece that starts at source index sp, shaped for ring index w
  __device__ __forceinline__ WPiece wv_get_stream(uint32_t sp, uint32_t w) const {
    WPiece r;
    r.dl = delta(w);
    const uint32_t a = sp + r.dl;
    const uint32_t* q = Base::dwp(wsb + (a & ((uint32_t)KS - 1u) & ~3u));
    r.v = __builtin_amdgcn_alignbyte(q[1], q[0], a & 3u);
    return r;
  }
  __device__ __forceinline__ WPiece wv_get_ring(uint32_t sw, uint32_t w) const {   // sw: ring coordinates of the source
    WPiece r;
    r.dl = delta(w);
    const uint32_t a = sw + r.dl;
    const uint32_t* q = Base::dwp(wrb + (a & ((uint32_t)KW - 1u) & ~3u));
    r.v = __builtin_amdgcn_alignbyte(q[1], q[0], a & 3u);
    return r;
  }
  __device__ __forceinline__ WPiece wv_get_mem(const uint8_t* m, uint32_t w) const {   // [m, m + 256) is readable
    WPiece r;
    r.dl = delta(w);
    const typename Base::vecLB t = LZ4HIP_MATCH_LOAD((const typename Base::vecLB*)(m + r.dl));
    __builtin_memcpy(&r.v, &t, 4);
    return r;
  }
  // (lane 0's ring index is w, lane l's (w & ~3) + 4 l: both are w + dl.  One store instruction serves the aligned lanes and the one
  // that is not -- the compiler emits ds_write_b32 for either.  Measured, gpurun_out/r05b: lane 0 in a store instruction of its own
  // -- the ring loop's rule for its 4..16 group leaders -- is 14 % SLOWER here (2048 x 4 MiB 52.4 -> 61.6 ms): one misaligned lane
  // costs the LDS a cycle, the exec juggling around a second store costs the wavefront a dozen instructions)
  __device__ __forceinline__ void wv_put(uint32_t w, const WPiece& c) {
    const uint32_t t = (w + 4u + c.dl) & ((uint32_t)KW - 1u);
    uint8_t* a = (wrb - 4) + t;
    __builtin_memcpy(a, &c.v, 4);
    const uint32_t u = (w + 4u) & ((uint32_t)KW - 1u);               // wave-uniform: does any lane of this piece lie at the ring's ends?
    if (__builtin_expect((u < 20u) | (u > (uint32_t)KW - 264u), 0)) {
      asm volatile("; a piece at the ring's ends" ::: "memory");       // (keeps this a scalar branch: flattened into a predicate it cost five instructions per piece)
      if (t < 20u) __builtin_memcpy(a + KW, &c.v, 4);                  // (the ring's first 16 bytes are kept behind its end as well: the parallel trips read 16 bytes at any index)
    }
  }
  __device__ __forceinline__ LChunk wv_read_al(uint32_t fw) const {   // fw: a multiple of 256 in ring coordinates
    LChunk v;
    v.w[0] = *Base::dwp(wrb + ((fw + l4) & ((uint32_t)KW - 1u)));
    return v;
  }
#ifdef LZ4HIP_RING_DBG
  __device__ __forceinline__ void wv_stats(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e, uint32_t f, uint32_t h) const {
    if (this->l == 0u) { atomicAdd(&g_ring_stat[0], a); atomicAdd(&g_ring_stat[1], b); atomicAdd(&g_ring_stat[2], c); atomicAdd(&g_ring_stat[3], d);
                         atomicAdd(&g_ring_stat[4], e); atomicAdd(&g_ring_stat[5], f); atomicAdd(&g_ring_stat[6], h); atomicAdd(&g_ring_stat[7], 1ull); }
  }
#endif
  // ---- lane-parallel side (the parallel trips of lz4_decode_wave.h): per-lane values are plain scalars (SIMT) ----
  typedef uint32_t VU;
  typedef bool VB;
  __device__ __forceinline__ VU vlane() const { return this->l; }
  __device__ __forceinline__ static VU vsel(bool c, VU a, VU b) { return c ? a : b; }
  __device__ __forceinline__ static uint64_t vballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }
  __device__ __forceinline__ static bool vlanes(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }   // per-lane predicate from a wave-uniform lane mask (the SGPR pair IS the mask: no vector instruction)
  __device__ __forceinline__ static uint32_t vreadlane(VU v, uint32_t i) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)i); }
  __device__ __forceinline__ static VU vwritelane(VU v, uint32_t s, uint32_t i) {
    s = (uint32_t)__builtin_amdgcn_readfirstlane((int)s);   // (wave-uniform values the compiler happens to hold in a vector register reach the asm as one: folds away when they are scalar already)
    i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(s), "s"(i) : "m0");
    return v;
  }
  // The walk of a trip (lz4_decode_wave.h): nx holds, a byte per window position (four per lane), where the sequence that starts
  // there is followed by the next one (255: not in this window).  From position 0: the k-th start goes to lane T + k of posv, T counts
  // them (at most 64 lanes).  By hand: the compiler's loop is 15 instructions a hop, six of them the two-part loop condition; this
  // one is 12 -- a window of text is ~58 hops
  __device__ __forceinline__ static void vwalk(VU nx, VU& posv, uint32_t& T) {
    uint32_t s = 0u, t1, t2;
    T = (uint32_t)__builtin_amdgcn_readfirstlane((int)T);
    // Four hops at a time without a test in between: position 255's "next" is 255 (nothing that starts at 248 .. 255 is followed by a
    // start inside the window), so hops past the last start stay there and write 255; the count is then the first lane that holds
    // 255.  While fewer than 61 lanes are taken; the last ones hop one at a time with both tests.  8 instructions a hop + 4 per four.
#define LZ4HIP_WALK_HOP \
        "  s_mov_b32 m0, %[T]\n" \
        "  s_lshr_b32 %[t1], %[s], 2\n" \
        "  v_writelane_b32 %[pv], %[s], m0\n" \
        "  s_lshl_b32 %[t2], %[s], 3\n" \
        "  v_readlane_b32 %[s], %[nx], %[t1]\n" \
        "  s_add_u32 %[T], %[T], 1\n" \
        "  s_lshr_b32 %[s], %[s], %[t2]\n" \
        "  s_and_b32 %[s], %[s], 0xff\n"
    asm volatile(
        "  s_cmp_gt_u32 %[T], 60\n"
        "  s_cbranch_scc1 L_walkchk_%=\n"
        "L_walk4_%=:\n"
        LZ4HIP_WALK_HOP LZ4HIP_WALK_HOP LZ4HIP_WALK_HOP LZ4HIP_WALK_HOP
        "  s_cmpk_eq_u32 %[s], 0xff\n"
        "  s_cbranch_scc1 L_walkend_%=\n"
        "  s_cmp_le_u32 %[T], 60\n"
        "  s_cbranch_scc1 L_walk4_%=\n"
        "L_walkchk_%=:\n"
        "  s_cmp_lt_u32 %[T], 64\n"
        "  s_cbranch_scc0 L_walkend_%=\n"
        "L_walk1_%=:\n"
        LZ4HIP_WALK_HOP
        "  s_cmp_lt_u32 %[T], 64\n"
        "  s_cselect_b32 %[t1], %[s], 0xff\n"
        "  s_cmpk_lg_u32 %[t1], 0xff\n"
        "  s_cbranch_scc1 L_walk1_%=\n"
        "L_walkend_%=:\n"
        "  v_cmp_eq_u32_e32 vcc, 0xff, %[pv]\n"
        "  s_ff1_i32_b64 %[t1], vcc\n"
        "  s_min_u32 %[T], %[T], %[t1]\n"
        : [pv] "+v"(posv), [T] "+s"(T), [s] "+s"(s), [t1] "=&s"(t1), [t2] "=&s"(t2)
        : [nx] "v"(nx)
        : "m0", "scc", "vcc");
#undef LZ4HIP_WALK_HOP
  }
  // The same walk by POINTER DOUBLING, for windows full of sequences (text: ~58 starts in 256 bytes, i.e. ~460 scalar instructions of the
  // walk above = a third of a text trip): J_0 = nx (the successor of every window position, a byte each, four per lane); J_k = J_(k-1) o
  // J_(k-1) is the 2^k-th successor -- four lane gathers per level, five levels; the start of rank r is J_5^(b5) .. J_0^(b0) (0) with b the
  // bits of r -- one gather per level for all 64 ranks at once, lane r its own.  Position 255 is its own successor, so ranks past the last
  // start read 255 and T is the first such lane, as above.  ~190 vector instructions whatever T: the caller takes it when the window before
  // this one held 24 starts or more.  Same posv, same T as vwalk -- the CPU suite's backend computes both and compares.
  __device__ __forceinline__ static uint32_t vgather8(VU tab, VU q) {   // byte q (0 .. 255) of a 256-byte table held four bytes per lane
    const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(q & ~3u), (int)tab);
    return (t >> ((q & 3u) * 8u)) & 255u;
  }
  __device__ __forceinline__ static void vwalk_par(VU nx, VU lane, VU& posv, uint32_t& T) {
    VU J = nx;
    VU pos = (lane & 1u) ? vgather8(J, VU(0u)) : VU(0u);
#pragma unroll
    for (uint32_t k = 1; k < 6u; k++) {
      const VU n0 = vgather8(J, J & 255u), n1 = vgather8(J, (J >> 8) & 255u), n2 = vgather8(J, (J >> 16) & 255u), n3 = vgather8(J, J >> 24);
      J = n0 | (n1 << 8) | (n2 << 16) | (n3 << 24);
      const VU nxt = vgather8(J, pos);
      pos = ((lane >> k) & 1u) ? nxt : pos;
    }
    posv = pos;
    const uint64_t endm = __builtin_amdgcn_ballot_w64(pos == 255u);
    T = endm ? (uint32_t)__builtin_ctzll(endm) : 64u;
  }
  __device__ __forceinline__ static VU vshfl(VU v, VU srcl) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(srcl << 2), (int)v); }
  __device__ __forceinline__ static VU vexcl_scan(VU a) {   // exclusive prefix sum across the wavefront: row_shr DPP adds + row_bcast15/31
    int x = (int)a;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return (uint32_t)x - a;
  }
  __device__ __forceinline__ static VU vmin(VU a, VU b) { return a < b ? a : b; }
  __device__ __forceinline__ static VU valignbyte(VU hi, VU lo, VU sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }   // bytes [sh, sh + 4) of {hi, lo}, sh = 0 .. 3
  // the trip's window: stream bytes [ip + 4 l, ip + 4 l + 8) of lane l, from three ALIGNED dwords (ip is wave-uniform)
  __device__ __forceinline__ void vs_win(uint32_t ip, VU& lo, VU& hi) const {
    const uint32_t* q = Base::dwp(wsb + (((ip & ~3u) + l4) & ((uint32_t)KS - 1u)));
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], s = ip & 3u;
    lo = __builtin_amdgcn_alignbyte(d1, d0, s);
    hi = __builtin_amdgcn_alignbyte(d2, d1, s);
  }
  // the 4 stream bytes at a per-lane position: the two aligned dwords around it, funnelled
  __device__ __forceinline__ VU vs_ld32(VU p) const {
    const uint32_t* q = Base::dwp(wsb + (p & ((uint32_t)KS - 1u) & ~3u));
    return __builtin_amdgcn_alignbyte(q[1], q[0], p & 3u);
  }
  // n bytes (1 .. 16) to ring coordinates w: where they belong, and once more KW bytes on when they touch the ring's first 16 bytes
  // or reach past its end (stored at t - 16, t = (index + 16) & (KW - 1): a store that straddles the end lands in the pad + the
  // ring's first bytes AND in its last bytes + the tail).  T: an unaligned 16 / 8 / 4 / 2 / 1-byte type
  typedef uint32_t u4a __attribute__((ext_vector_type(4), aligned(1)));
  typedef uint32_t u2a __attribute__((ext_vector_type(2), aligned(1)));
  typedef uint32_t u1a __attribute__((aligned(1)));
  typedef uint16_t h1a __attribute__((aligned(1)));
  template <class T> __device__ __forceinline__ void vput(VU w, const T& v, bool m) {
    const uint32_t t = (w + 16u) & ((uint32_t)KW - 1u);
    uint8_t* a = (wrb - 16) + t;
    if (m) *(T*)a = v;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(m & (t < 32u)) != 0ull, 0)) {
      if (m & (t < 32u)) *(T*)(a + KW) = v;
    }
  }
  // A load from the block's flushed output (a FAR match source) that the compiler does not see as a memory operation, waited for
  // on the spot.  Why: the copies sit inside the trip's pass and round loops, and with a vector memory load anywhere in a loop the
  // compiler puts s_waitcnt vmcnt(0) at the loop's head -- where it waits for the stream refill requested at the top of the trip
  // and for the flusher's stores of the trip before (stores count in vmcnt on this part).  Measured with a build that had no far
  // loads in the loops at all (gpurun_out/r05y2): 2048 x 4 MiB 27.0 -> 22.6 ms, 4096: 37.2 -> 32.6.
  template <class T> __device__ __forceinline__ static T vld_mem(const uint8_t* p) {
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    typedef uint32_t u2v __attribute__((ext_vector_type(2)));
    T r;
    if constexpr (sizeof(T) == 16) { u4v t; asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(t) : "v"(p) : "memory"); __builtin_memcpy(&r, &t, 16); }
    else if constexpr (sizeof(T) == 8) { u2v t; asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(t) : "v"(p) : "memory"); __builtin_memcpy(&r, &t, 8); }
    else if constexpr (sizeof(T) == 4) { uint32_t t; asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(t) : "v"(p) : "memory"); __builtin_memcpy(&r, &t, 4); }
    else if constexpr (sizeof(T) == 2) { uint32_t t; asm volatile("global_load_ushort %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(t) : "v"(p) : "memory"); r = (T)t; }
    else { uint32_t t; asm volatile("global_load_ubyte %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(t) : "v"(p) : "memory"); r = (T)t; }
    return r;
  }
  // EXACT lane-per-run copy, the slow form (a destination at the ring's ends, a run of more than 64 bytes): len bytes (lanes with go)
  // from the stream ring, the output ring (source in ring coordinates) or memory to ring coordinates dw, 16 bytes at a time while 16 are
  // left, then 8 / 4 / 2 / 1, every step with its mirror store.  Nothing outside [dw, dw + len) is written; sources and destinations of
  // different lanes do not overlap (the caller's dependency rule)
  template <class T, bool MEM = false> __device__ __forceinline__ void vstep(const uint8_t* sb, uint32_t sm, VU dw, VU sp, VU c, bool m) {
    T v = T();
    if constexpr (MEM) { if (m) v = vld_mem<T>(sb + (sp + c)); }
    else { if (m) v = *(const T*)(sb + ((sp + c) & sm)); }
    vput<T>(dw + c, v, m);
  }
  // SRC 0: the stream ring, 1: the output ring (sp in ring coordinates), 2: memory (sb + sp is the source; [sb + sp, + len) is readable)
  template <int SRC> __device__ __forceinline__ void vcopy(VU dw, const uint8_t* sb0, VU sp, VU len, bool go) {
    const uint8_t* sb = SRC == 0 ? wsb : SRC == 1 ? wrb : sb0;
    const uint32_t sm = SRC == 0 ? (uint32_t)KS - 1u : SRC == 1 ? (uint32_t)KW - 1u : 0xFFFFFFFFu;
    for (uint32_t c = 0u;; c += 16u) {
      const bool m = go && (c + 16u <= len);
      if (__builtin_amdgcn_ballot_w64(m) == 0ull) break;
      vstep<u4a, SRC == 2>(sb, sm, dw, sp, c, m);
    }
    uint32_t c = len & ~15u;
    bool m = go && ((len & 8u) != 0u);
    vstep<u2a, SRC == 2>(sb, sm, dw, sp, c, m);
    c += m ? 8u : 0u;
    m = go && ((len & 4u) != 0u);
    vstep<u1a, SRC == 2>(sb, sm, dw, sp, c, m);
    c += m ? 4u : 0u;
    m = go && ((len & 2u) != 0u);
    vstep<h1a, SRC == 2>(sb, sm, dw, sp, c, m);
    c += m ? 2u : 0u;
    m = go && ((len & 1u) != 0u);
    vstep<uint8_t, SRC == 2>(sb, sm, dw, sp, c, m);
  }
  // The usual round -- every active run <= 64 bytes, no destination at the ring's ends -- issues ALL its reads (4 x 16 bytes and an
  // 8 / 4 / 2 / 1 cascade per run, only the active lanes: an unaligned LDS access costs a cycle per lane) in front of ONE wait and
  // then the predicated stores; the step-by-step form above waits for the LDS once per step (14 round trips per trip, 43 % of the
  // wavefront's cycles when it was the only form: profiles/r05_wave_notes.txt)
  struct Run16 { u4a a0, a1, a2, a3; u2a b; uint32_t c; uint32_t d; uint32_t e; };
  // NA: the 16-byte pieces a round reads and stores -- 4: runs of up to 64 bytes; 1: every run of the round is shorter than 32 bytes; 0: shorter than
  // 16 (text: most rounds).  An unaligned LDS access costs a cycle per active lane whether its bytes are wanted or not
  template <int NA, bool PRED = false>
  __device__ __forceinline__ static Run16 vrun_load(const uint8_t* sb, uint32_t sm, VU sp, VU len) {
    Run16 r;
    if (PRED) {   // the 16-byte pieces read only by the lanes whose run has them (as the stores are): an unaligned LDS access costs a cycle per ACTIVE lane.
      // Three more mask set-ups per round: it pays where the LDS is busy -- the one-wavefront kernels at 8-16 wavefronts per CU (4096 x 4 MiB +2.4 %,
      // text +2 %) -- and costs where a lone wavefront waits for its own instructions (the trio's copier: one block 0.153 -> 0.156 ms); gpurun_out/r06ap
      r.a0 = u4a(); r.a1 = u4a(); r.a2 = u4a(); r.a3 = u4a();
      if (NA >= 1) if (len >= 16u) r.a0 = *(const u4a*)(sb + (sp & sm));
      if (NA >= 2) if (len >= 32u) r.a1 = *(const u4a*)(sb + ((sp + 16u) & sm));
      if (NA >= 3) if (len >= 48u) r.a2 = *(const u4a*)(sb + ((sp + 32u) & sm));
      if (NA >= 4) if (len >= 64u) r.a3 = *(const u4a*)(sb + ((sp + 48u) & sm));
    } else {
      if (NA >= 1) r.a0 = *(const u4a*)(sb + (sp & sm));
      if (NA >= 2) r.a1 = *(const u4a*)(sb + ((sp + 16u) & sm));
      if (NA >= 3) r.a2 = *(const u4a*)(sb + ((sp + 32u) & sm));
      if (NA >= 4) r.a3 = *(const u4a*)(sb + ((sp + 48u) & sm));
    }
    uint32_t c = NA >= 1 ? len & ~15u : 0u;
#if LZ4HIP_PRED_CASCADE   /* the 8- and 4-byte pieces too: nothing on long runs, text 214 -> 224 GB/s (its runs are 2 .. 6 bytes: the 8-byte piece is rarely wanted); gpurun_out/r06ar */
    if (PRED) {
      r.b = u2a(); r.c = 0u;
      if (len & 8u) r.b = *(const u2a*)(sb + ((sp + c) & sm));
      c += len & 8u;
      if (len & 4u) r.c = *(const u1a*)(sb + ((sp + c) & sm));
      c += len & 4u;
      r.d = *(const h1a*)(sb + ((sp + c) & sm));
      c += len & 2u;
      r.e = *(sb + ((sp + c) & sm));
      return r;
    }
#endif
    r.b = *(const u2a*)(sb + ((sp + c) & sm));
    c += len & 8u;
    r.c = *(const u1a*)(sb + ((sp + c) & sm));
    c += len & 4u;
    r.d = *(const h1a*)(sb + ((sp + c) & sm));
    c += len & 2u;
    r.e = *(sb + ((sp + c) & sm));
    return r;
  }
  // the same run from the block's flushed output (a far source): eight loads in flight together, one wait, none of it visible to
  // the compiler as a memory operation (see vld_mem)
  __device__ __forceinline__ static Run16 vrun_load_mem(const uint8_t* p, VU len) {
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    typedef uint32_t u2v __attribute__((ext_vector_type(2)));
    u4v a0, a1, a2, a3; u2v b; uint32_t c, d, e;
    uint32_t o = len & ~15u;
    const uint8_t* pb = p + o;
    o += len & 8u;
    const uint8_t* pc = p + o;
    o += len & 4u;
    const uint8_t* pd = p + o;
    o += len & 2u;
    const uint8_t* pe = p + o;
    asm volatile(
        "global_load_dwordx4 %0, %8, off\n\t"
        "global_load_dwordx4 %1, %8, off offset:16\n\t"
        "global_load_dwordx4 %2, %8, off offset:32\n\t"
        "global_load_dwordx4 %3, %8, off offset:48\n\t"
        "global_load_dwordx2 %4, %9, off\n\t"
        "global_load_dword %5, %10, off\n\t"
        "global_load_ushort %6, %11, off\n\t"
        "global_load_ubyte %7, %12, off\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e)
        : "v"(p), "v"(pb), "v"(pc), "v"(pd), "v"(pe)
        : "memory");
    Run16 r;
    __builtin_memcpy(&r.a0, &a0, 16); __builtin_memcpy(&r.a1, &a1, 16); __builtin_memcpy(&r.a2, &a2, 16); __builtin_memcpy(&r.a3, &a3, 16);
    __builtin_memcpy(&r.b, &b, 8);
    r.c = c; r.d = d; r.e = e;
    return r;
  }
  template <int NA>
  __device__ __forceinline__ void vrun_store(VU dw, VU len, const Run16& r) {   // (no destination at the ring's ends: no mirror stores)
    uint8_t* a = wrb + (dw & ((uint32_t)KW - 1u));   // (contiguous: the caller checked that [dw, dw + len) does not reach the end)
    if (NA >= 1) if (len >= 16u) *(u4a*)a = r.a0;
    if (NA >= 2) if (len >= 32u) *(u4a*)(a + 16) = r.a1;
    if (NA >= 3) if (len >= 48u) *(u4a*)(a + 32) = r.a2;
    if (NA >= 4) if (len >= 64u) *(u4a*)(a + 48) = r.a3;
    uint32_t c = NA >= 1 ? len & ~15u : 0u;
    if (len & 8u) *(u2a*)(a + c) = r.b;
    c += len & 8u;
    if (len & 4u) *(u1a*)(a + c) = r.c;
    c += len & 4u;
    if (len & 2u) *(h1a*)(a + c) = (uint16_t)r.d;
    c += len & 2u;
    if (len & 1u) a[c] = (uint8_t)r.e;
  }
  // One RUN per lane (lz4_decode_wave.h: even lanes the literals of a sequence, odd lanes its match): len bytes to ring coordinates
  // dw, from the stream ring at stream position sp (from_stream) or from the output ring at ring coordinates sp.
  // The lanes of the round, the far ones and the "odd" ones come as wave-uniform LANE MASKS (gom; farm, oddm: computed once per pass
  // by the caller): the round's tests are scalar ANDs, its predicates SGPR pairs -- a ballot of a bool the compiler holds as a mask
  // costs a v_cndmask + v_cmp each time.
  // far: the match source of the lane is not in the ring: it is mem[mpos, mpos + len) (the block's flushed output; 80 bytes from mpos
  // on are readable).  The far lanes of a round have their loads in flight together, behind the ring reads of the others
  __device__ __forceinline__ uint64_t vodd_mask(VU dw, VU len) const {   // lanes whose run cannot take the usual round: longer than 64 bytes, or a destination at the ring's ends
    const uint32_t x = dw & ((uint32_t)KW - 1u);
    return __builtin_amdgcn_ballot_w64(len > 64u) | __builtin_amdgcn_ballot_w64(x < 16u) | __builtin_amdgcn_ballot_w64(x + len + 16u > (uint32_t)KW);   // (a ballot per comparison: see lz4_decode_wave.h)
  }
  template <int NA, bool PRED>
  __device__ __forceinline__ void vround(VU dw, bool from_stream, VU sp, VU len, uint64_t gom, const uint8_t* mem, VU mpos, uint64_t gfm) {
    if (vlanes(gom)) {
      Run16 r = vrun_load<NA, PRED>(from_stream ? wsb : wrb, from_stream ? (uint32_t)KS - 1u : (uint32_t)KW - 1u, sp, len);
      if (__builtin_expect(gfm != 0ull, 0)) {
        // (the rare branches take their operands through an empty asm: what is computed from them -- 64-bit addresses here, ring
        // masks and mirror tests below -- is computed IN the branch; the compiler otherwise hoists ~55 instructions of it in front
        // of the round loop, where every pass pays for them)
        VU mp2 = mpos, ln2 = len;
        asm volatile("" : "+v"(mp2), "+v"(ln2));
        if (vlanes(gfm)) r = vrun_load_mem(mem + mp2, ln2);
      }
      vrun_store<NA>(dw, len, r);
    }
  }
  __device__ __forceinline__ static uint32_t vrun_tier(VU len, uint64_t actm) {   // the short form the runs of the lanes actm allow: 0 / 1, or 4 = none
    if ((__builtin_amdgcn_ballot_w64(len >= 32u) & actm) != 0ull) return 4u;
    return (__builtin_amdgcn_ballot_w64(len >= 16u) & actm) != 0ull ? 1u : 0u;
  }
  template <bool PRED = false>
  __device__ __forceinline__ void vcopy_run(VU dw, bool from_stream, VU sp, VU len, uint64_t gom, const uint8_t* mem, VU mpos, uint64_t farm, uint64_t oddm,
                                            uint32_t tier = 4u) {
    const uint64_t gfm = gom & farm;
    if (__builtin_expect((oddm & gom) == 0ull, 1)) {
      // the form of the round: 4 = pieces for runs of up to 64 bytes; 1 / 0 = every run of the PASS is shorter than 32 / 16 bytes (vrun_tier, once per
      // pass; only the wave loop's SHORT instance passes anything but 4: lz4_decode_wave.h)
      if (LZ4HIP_RUN_TIERS && tier != 4u) {
        VU dw2 = dw, sp2 = sp, ln2 = len;
#if LZ4HIP_TIER_GUARD
        asm volatile("" : "+v"(dw2), "+v"(sp2), "+v"(ln2));
#endif
        if (tier == 0u) vround<0, PRED>(dw2, from_stream, sp2, ln2, gom, mem, mpos, gfm);
        else vround<1, PRED>(dw2, from_stream, sp2, ln2, gom, mem, mpos, gfm);
      } else {
        vround<4, PRED>(dw, from_stream, sp, len, gom, mem, mpos, gfm);
      }
    } else {
      VU dw2 = dw, sp2 = sp, ln2 = len, mp2 = mpos;
      asm volatile("" : "+v"(dw2), "+v"(sp2), "+v"(ln2), "+v"(mp2));
      const bool go = vlanes(gom), far = vlanes(farm);
      vcopy<0>(dw2, nullptr, sp2, ln2, go && from_stream);
      vcopy<1>(dw2, nullptr, sp2, ln2, go && !from_stream && !far);
      if (gfm != 0ull) vcopy<2>(dw2, mem, mp2, ln2, go && far);
    }
  }
  // ---- the pair loop's mailbox (lz4_decode_pair.h: a PARSER and a COPIER wavefront per block): behind the wavefront's rings,
  // kMailSlots messages of 64 lanes x 8 bytes and 16 control words.  Ordering between the two wavefronts rests on the LDS itself --
  // it executes a wavefront's instructions in order, each as a whole -- so a post is a store behind the data it publishes and a peek
  // a load in front of the data it guards; the empty asm statements hold the COMPILER to that order (volatile alone orders only
  // volatile accesses).  No fence: a workgroup-scope release would make the parser wait for its stream refill -- a load from
  // memory that is meant to be in flight for a whole trip -- at every message ----
  static constexpr uint32_t kMailSlots = 3u, kMailSlotBytes = 512u, kMailBytes = kMailSlots * kMailSlotBytes + 64u;
  static constexpr uint32_t kPairLds = kWaveLds + kMailBytes;
  // (the mailbox is addressed as LDS -- address space 3 -- explicitly: through the generic pointer the volatile accesses came out as
  // flat_load / flat_store with system-scope cache bits)
  typedef __attribute__((address_space(3))) uint8_t* lds8p;
  typedef __attribute__((address_space(3))) volatile uint32_t* lds32vp;
  typedef uint32_t u2v8 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) u2v8* lds64p;
  lds8p pmb = nullptr;
  __device__ __forceinline__ void pm_begin(uint8_t* lds) { pmb = (lds8p)(lds + kWaveLds); }
  __device__ __forceinline__ void wv_begin_db(uint8_t* lds, uint32_t db) { wsb = lds; wrb = lds + KS + 32; wdb = db; }   // (the parser: it has no destination pointer, only its low byte)
  __device__ __forceinline__ uint32_t pm_peek(uint32_t i) const {
    asm volatile("" ::: "memory");
    const uint32_t v = *(lds32vp)(pmb + kMailSlots * kMailSlotBytes + 4u * i);
    asm volatile("" ::: "memory");
    return uni(v);
  }
  __device__ __forceinline__ void pm_peek2(uint32_t i, uint32_t& a, uint32_t& b) const {   // control words i (even) and i + 1 in one round trip
    asm volatile("" ::: "memory");
    const u2v8 t = *(__attribute__((address_space(3))) volatile u2v8*)(pmb + kMailSlots * kMailSlotBytes + 4u * i);
    asm volatile("" ::: "memory");
    a = uni(t.x); b = uni(t.y);
  }
  __device__ __forceinline__ void pm_post(uint32_t i, uint32_t v) {
    asm volatile("" ::: "memory");
    if (this->l == 0u) *(lds32vp)(pmb + kMailSlots * kMailSlotBytes + 4u * i) = v;
    asm volatile("" ::: "memory");
  }
  __device__ __forceinline__ void pm_put(uint32_t slot, uint32_t w0, uint32_t w1) {
    u2v8 t; t.x = w0; t.y = w1;
    *(lds64p)(pmb + slot * kMailSlotBytes + 2u * l4) = t;
  }
  // a control word and a message slot in ONE round trip: the slot is read BEHIND the word (the LDS executes the two reads in order), so if
  // the word says the slot is published the data are the message; if not, they are discarded (the copier of the trio loop: HEAD + its next slot)
  __device__ __forceinline__ uint32_t pm_peek_get(uint32_t i, uint32_t slot, uint32_t& w0, uint32_t& w1) const {
    asm volatile("" ::: "memory");
    const uint32_t v = *(lds32vp)(pmb + kMailSlots * kMailSlotBytes + 4u * i);
    asm volatile("" ::: "memory");
    const u2v8 t = *(lds64p)(pmb + slot * kMailSlotBytes + 2u * l4);
    asm volatile("" ::: "memory");
    w0 = t.x; w1 = t.y;
    return uni(v);
  }
  __device__ __forceinline__ void pm_get(uint32_t slot, uint32_t& w0, uint32_t& w1) const {
    const u2v8 t = *(lds64p)(pmb + slot * kMailSlotBytes + 2u * l4);
    w0 = t.x; w1 = t.y;
  }
  // the trio loop's second queue (lz4_decode_trio.h: scanner -> planner), behind the mailbox: kScanSlots entries of 64 positions (a dword
  // per lane) + {count, window position, next window, flags}
  static constexpr uint32_t kScanSlots = 3u, kScanBytes = 288u;
  static constexpr uint32_t kTrioLds = kPairLds + kScanSlots * kScanBytes;
  typedef __attribute__((address_space(3))) uint32_t* lds32p;
  typedef uint32_t u4v16 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) u4v16* lds128p;
  __device__ __forceinline__ void sq_put(uint32_t slot, uint32_t posv, uint32_t T, uint32_t wip, uint32_t nextw, uint32_t flags) {
    lds8p e = pmb + kMailBytes + slot * kScanBytes;
    *(lds32p)(e + l4) = posv;
    const uint32_t hv = this->l == 0u ? T : this->l == 1u ? wip : this->l == 2u ? nextw : flags;
    if (this->l < 4u) *(lds32p)(e + 256u + l4) = hv;
  }
  // (a control word and the entry behind it in one round trip: pm_peek_get's rule)
  __device__ __forceinline__ uint32_t sq_peek_get(uint32_t i, uint32_t slot, uint32_t& posv, uint32_t& T, uint32_t& wip, uint32_t& nextw, uint32_t& flags) const {
    asm volatile("" ::: "memory");
    const uint32_t v = *(lds32vp)(pmb + kMailSlots * kMailSlotBytes + 4u * i);
    asm volatile("" ::: "memory");
    lds8p e = pmb + kMailBytes + slot * kScanBytes;
    posv = *(lds32p)(e + l4);
    const u4v16 h = *(lds128p)(e + 256u);
    asm volatile("" ::: "memory");
    T = uni(h.x); wip = uni(h.y); nextw = uni(h.z); flags = uni(h.w);
    return uni(v);
  }
  __device__ __forceinline__ void sq_get(uint32_t slot, uint32_t& posv, uint32_t& T, uint32_t& wip, uint32_t& nextw, uint32_t& flags) const {
    lds8p e = pmb + kMailBytes + slot * kScanBytes;
    posv = *(lds32p)(e + l4);
    const u4v16 h = *(lds128p)(e + 256u);
    T = uni(h.x); wip = uni(h.y); nextw = uni(h.z); flags = uni(h.w);
  }
  // (why: developer builds with -DLZ4HIP_RING_DBG count the naps by reason -- which stage of the trio loop waits for which: tools/wave_stats.py)
  __device__ __forceinline__ static void pm_nap(uint32_t why = 7u) {                  // a message is a few hundred cycles away
#ifdef LZ4HIP_RING_DBG
    if ((threadIdx.x & 63u) == 0u) atomicAdd(&g_ring_stat[why & 7u], 1ull);
#else
    (void)why;
#endif
    __builtin_amdgcn_s_sleep(1);
  }
  __device__ __forceinline__ static void pm_idle() { __builtin_amdgcn_s_sleep(8); }   // between entries: the copier is in decode_block's exact code, or between blocks
  __device__ __forceinline__ static const uint8_t* pm_ptr(uint32_t lo, uint32_t hi) {   // (through address space 1: a pointer rebuilt from integers is otherwise a FLAT pointer)
    typedef __attribute__((address_space(1))) const uint8_t* G;
    return (const uint8_t*)(G)(((uint64_t)hi << 32) | lo);
  }
  // the step at ring coordinates fw (a multiple of 256) to memory: its bytes inside [lo, hi) (ring coordinates), nothing else
  __device__ __forceinline__ void wv_store(uint8_t* dst, uint32_t fw, const LChunk& c, uint32_t lo, uint32_t hi) const {
    uint8_t* p = dst + (intptr_t)(int32_t)(fw + l4 - wdb);
    if (__builtin_expect((fw >= lo) & (fw + 256u <= hi), 1)) {
      *(uint32_t*)__builtin_assume_aligned(p, 4) = c.w[0];
    } else {
      const uint32_t a = fw + l4;
      if ((a >= lo) & (a + 4u <= hi)) *(uint32_t*)__builtin_assume_aligned(p, 4) = c.w[0];
      else {
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) if ((a + k >= lo) & (a + k < hi)) p[k] = (uint8_t)(c.w[0] >> (8u * k));
      }
    }
  }
};

}  // namespace lz4hip

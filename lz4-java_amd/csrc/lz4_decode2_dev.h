// lz4_decode2_dev.h -- the two-pass LZ4 block decoder for big batches (gfx950).
//
// Replaces, for the "HIP" family, LZ4_decompress_safe / LZ4_decompress_fast (/root/reference/src/jni/net_jpountz_lz4_LZ4JNI.c:216,
// :169); same return codes and bytes as lz4_decode_core.h (which stays the decoder of small batches and the fallback).
//
// Why two passes.  A block decodes as a serial token walk plus copies.  The one-pass kernels keep every block of the batch in
// flight to hide the walk's latency, and then no cache can hold the blocks' 64 KiB windows: every match source is a random
// 128-byte line from HBM (round 1: 26.4 GB of traffic for 6.4 GB of algorithmic bytes).  Here
//   pass 1 (decode2_parse_kernel)  walks the tokens of EVERY block at once -- the exact acceptance checks and error codes of
//          lz4_decode_core.h, through a backend whose "copies" record 16-byte sequence descriptors {literal source, literal
//          length, match length, offset} into chunks drawn from an arena -- and moves no data;
//   pass 2 (decode2_copy_kernel)   gives each block a 256-thread workgroup whose LDS holds the block's 64 KiB output window:
//          descriptors are taken 256 at a time (one per thread), output positions come from a workgroup prefix sum, all literal
//          runs are copied at once (memory -> window), then the matches (window -> window) in dependency rounds -- a match is
//          ready when no unfinished match writes into the 16-byte granules its source covers (a dirty bitmap in LDS), the
//          first unfinished match is always ready -- and finished output leaves the window as whole 128-byte lines.
// HBM traffic = compressed bytes in + output out + 2 x 16 bytes per sequence; match sources never leave the chip.
// Blocks the arena has no room for, and blocks whose matches mostly depend on their neighbours (text), are left to the one-pass
// kernel (flag in Blk2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "lz4_decode_core.h"

namespace lz4hip {

constexpr uint32_t D2_CHUNK = 512u;          // arena chunk: 512 x 16 bytes; slot 0 = {next chunk, -, -, -}, slots 1..511 = descriptors
constexpr uint32_t D2_PER_CHUNK = D2_CHUNK - 1u;
constexpr uint32_t D2_NONE = 0xFFFFFFFFu;

struct Blk2 {            // per block, written by pass 1
  uint32_t first;        // first chunk (D2_NONE: nothing to copy)
  uint32_t ndesc;        // descriptors
  uint32_t flags;        // 1 = left to the one-pass kernel (arena full / dependency-heavy); 0 = pass 2 decodes it
  uint32_t pad;
};

// ---- pass 1 backend: the group interface of lz4_decode_core.h, recording instead of copying ----------------------------
template <int GL>
struct ParseGrp {
  uint32_t l;
  const uint8_t* src_base;
  uint4* arena;
  uint32_t* cursor;        // arena chunk cursor (device word)
  uint32_t cap_chunks;
  uint32_t first = D2_NONE, cur = D2_NONE, nd = 0, in_chunk = 0;
  uint32_t pend_src = 0, pend_lit = 0;
  bool have_pend = false, overflow = false;
  uint32_t near = 0;       // matches that start within 512 bytes of their source: candidates for dependency chains in pass 2

  __device__ __forceinline__ ParseGrp(const uint8_t* s, uint4* a, uint32_t* c, uint32_t cap) : l(threadIdx.x & (GL - 1)), src_base(s), arena(a), cursor(c), cap_chunks(cap) {}

  __device__ __forceinline__ static uint32_t ld8(const uint8_t* p) { return *p; }
  __device__ __forceinline__ static uint32_t ld16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
  __device__ __forceinline__ static uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
  __device__ __forceinline__ static uint64_t ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

  __device__ __forceinline__ void emit(uint32_t lit_src, uint32_t lit, uint32_t ml, uint32_t off) {
    if (overflow) return;
    if (cur == D2_NONE || in_chunk == D2_PER_CHUNK) {
      uint32_t c = 0;
      if (l == 0) c = atomicAdd(cursor, 1u);
      c = __shfl(c, (int)(threadIdx.x & 63u & ~(GL - 1)), 64);   // lane 0 of the group
      if (c >= cap_chunks) { overflow = true; return; }
      if (l == 0) {
        arena[(size_t)c * D2_CHUNK] = make_uint4(D2_NONE, 0u, 0u, 0u);
        if (cur != D2_NONE) ((uint32_t*)&arena[(size_t)cur * D2_CHUNK])[0] = c;
      }
      if (cur == D2_NONE) first = c;
      cur = c;
      in_chunk = 0;
    }
    if (l == 0) arena[(size_t)cur * D2_CHUNK + 1u + in_chunk] = make_uint4(lit_src, lit, ml, off);
    in_chunk++;
    nd++;
  }
  __device__ __forceinline__ void lits(const uint8_t* s, uint32_t len) {
    if (have_pend && pend_lit) emit(pend_src, pend_lit, 0u, 0u);   // (two literal copies in a row do not happen on valid paths; harmless)
    pend_src = (uint32_t)(s - src_base);
    pend_lit = len;
    have_pend = true;
  }
  __device__ __forceinline__ void match(uint32_t offset, uint32_t len) {
    emit(have_pend ? pend_src : 0u, have_pend ? pend_lit : 0u, len, offset);
    have_pend = false;
    pend_lit = 0;
    if (offset < 512u) near++;
  }
  __device__ __forceinline__ void finish() {
    if (have_pend && pend_lit) emit(pend_src, pend_lit, 0u, 0u);
    have_pend = false;
  }

  __device__ __forceinline__ void copy_lits(uint8_t*, const uint8_t* s, uint32_t len, bool) { lits(s, len); }
  __device__ __forceinline__ void copy_lits_wide(uint8_t*, const uint8_t* s, uint32_t len) { lits(s, len); }
  __device__ __forceinline__ void copy_match(uint8_t*, uint32_t, uint32_t offset, uint32_t len, bool) { match(offset, len); }
  __device__ __forceinline__ void copy_match_wide(uint8_t*, uint32_t, uint32_t offset, uint32_t len) { match(offset, len); }

  // (members the pipelined / staged loops of decode_block name; those loops are compiled out: PIPE = STAGE = false)
  struct SeqRegs {};
  static constexpr uint32_t kStage = 16u;
  uint32_t fl = 0;
  __device__ __forceinline__ static constexpr uint32_t step() { return 64u; }
  __device__ __forceinline__ static constexpr uint32_t slack() { return 16u; }
  __device__ __forceinline__ void seq_load(SeqRegs&, const uint8_t*, uint32_t, const uint8_t*, uint32_t) {}
  __device__ __forceinline__ void seq_store(const SeqRegs&, uint8_t*, uint32_t, uint32_t) {}
  __device__ __forceinline__ void st_begin(uint8_t*, uint32_t) {}
  __device__ __forceinline__ void st_flush_lines(uint8_t*, uint32_t) {}
  __device__ __forceinline__ void st_flush_all(uint8_t*, uint32_t) {}
  __device__ __forceinline__ void st_lits(uint8_t*, uint32_t, const uint8_t*, uint32_t) {}
  __device__ __forceinline__ void st_match(uint8_t*, uint32_t, uint32_t, uint32_t) {}
};

// pass 1: GL lanes per block (all lanes walk the tokens redundantly -- uniform loads --, lane 0 records)
// near_pct: a block whose share of near matches (offset < 512) reaches near_pct % is left to the one-pass kernel (0 = never)
template <int GL, bool SAFE>
__global__ __launch_bounds__(256) void decode2_parse_kernel(BatchArgs a, Blk2* meta, uint4* arena, uint32_t* cursor, uint32_t cap_chunks, uint32_t near_pct) {
  const uint32_t gid = (blockIdx.x * 256u + threadIdx.x) / GL;
  if (gid >= a.n) return;
  const uint8_t* src = a.src + a.src_off[gid];
  ParseGrp<GL> g(src, arena, cursor, cap_chunks);
  const int r = decode_block<ParseGrp<GL>, SAFE, false, false>(g, src, a.src_len[gid], nullptr, a.dst_cap[gid], nullptr);
  g.finish();
  if (g.l == 0) {
    Blk2 m;
    m.first = g.first;
    m.ndesc = r >= 0 ? g.nd : 0u;
    m.flags = (g.overflow || (near_pct && g.nd >= 64u && (uint64_t)g.near * 100u >= (uint64_t)g.nd * near_pct)) ? 1u : 0u;
    if (r < 0) m.flags = 0u;          // malformed: the error code stands, nothing is copied
    m.pad = 0;
    meta[gid] = m;
    if (!m.flags) a.out[gid] = r;     // (flagged blocks: the one-pass kernel writes the result)
  }
}

// ---- pass 2 -----------------------------------------------------------------------------------------------------------
constexpr uint32_t D2_NT = 256u;             // threads per block-workgroup
constexpr uint32_t D2_SPAN = 24576u;         // output bytes one batch may cover (dirty bitmap = D2_SPAN / 16 bits)
constexpr uint32_t D2_BIGLEN = 256u;         // literal runs / matches longer than this are copied by the whole workgroup
constexpr uint32_t D2_WIN = 65536u;

struct D2Shared {
  uint8_t win[D2_WIN];                       // output window: byte p of the block at win[(p + bias) & 0xFFFF]
  uint32_t ms[D2_NT], me[D2_NT];             // this batch: output interval [ms, me) of every sequence's match (empty: ms == me)
  uint32_t gmap[D2_SPAN / 64u + 2u];         // gmap[j]: first sequence whose match ends above byte 64 j of the batch
  uint32_t done[D2_NT / 32u];                // bit t: the match of sequence t is in the window
  uint32_t wsum[D2_NT / 64u];                // per-wave totals of the prefix sum
  uint32_t firstun[D2_NT / 64u];
  uint32_t misc[4];
};

__device__ __forceinline__ uint32_t d2_wave_excl_scan(uint32_t a, uint32_t& total) {
  int x = (int)a;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
  total = (uint32_t)__builtin_amdgcn_readlane(x, 63);
  return (uint32_t)x - a;
}

// One thread copies n <= 64 bytes to the window at index dp (dp + 64 <= 64 KiB: no wrap inside); written so that the lanes of a
// wavefront share ONE predicated region: rd(o) returns the source dword at byte offset o.  Whole dwords first -- the trip count is
// the longest lane's; a lane past its own length re-reads and re-writes its last dword --, then, if n is not a multiple of 4, one
// more dword that ENDS at byte n (it overlaps the previous one: no byte loop, nothing is read or written past n).  Lanes with
// n < 4 do nothing here (the caller moves their bytes).  All reads of the copy are requested before its first store.
template <int ND, class RD>
__device__ __forceinline__ void d2_copy_tier(uint8_t* win, uint32_t dp, uint32_t n, RD rd) {
  if (n >= 4u) {
    uint32_t v[ND];
    const uint32_t last = 4u * ((n >> 2) - 1u), to = n - 4u;
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)ND; k++) v[k] = rd(4u * k < last ? 4u * k : last);
    const uint32_t vt = rd(to);
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)ND; k++) __builtin_memcpy(win + dp + (4u * k < last ? 4u * k : last), &v[k], 4);
    __builtin_memcpy(win + dp + to, &vt, 4);
  }
}
template <class RD>
__device__ __forceinline__ void d2_copy64(uint8_t* win, uint32_t dp, uint32_t n, RD rd) {
  // (straight-line tiers chosen per wavefront: a lane past its own length re-reads and re-writes its last dword)
  if (__ballot(n > 32u)) d2_copy_tier<16>(win, dp, n, rd);
  else if (__ballot(n > 16u)) d2_copy_tier<8>(win, dp, n, rd);
  else d2_copy_tier<4>(win, dp, n, rd);
}

// byte-exact copy memory -> window by one thread, 64 bytes per step
__device__ __forceinline__ void d2_lits_thread(uint8_t* win, uint32_t wpos, const uint8_t* s, uint32_t len) {
  uint32_t i = 0;
  while (__ballot(i < len)) {
    const uint32_t n = i < len ? (len - i < 64u ? len - i : 64u) : 0u;
    const uint32_t dp = (wpos + i) & (D2_WIN - 1u);
    const bool wrap = dp + 64u > D2_WIN;
    if (__ballot(n && wrap)) {   // (the window wraps inside this step: bytes)
      if (wrap) for (uint32_t q = 0; q < n; q++) win[(wpos + i + q) & (D2_WIN - 1u)] = s[i + q];
    }
    const uint32_t m = wrap ? 0u : n;
    d2_copy64(win, dp, m, [&](uint32_t o) { uint32_t v; __builtin_memcpy(&v, s + i + o, 4); return v; });
    if (m && m < 4u) for (uint32_t q = 0; q < m; q++) win[dp + q] = s[i + q];
    i += 64u;
  }
}

// window -> window, byte-forward semantics (dst[i] = dst[i - off]), by one thread; the source lies in the window
__device__ __forceinline__ void d2_match_thread(uint8_t* win, uint32_t wdst, uint32_t off, uint32_t len, bool active) {
  // steps of `st` bytes, each read completely before it is written: valid while st <= off
  const uint32_t st = !active ? 64u : (off >= 64u ? 64u : (off >= 16u ? 16u : 0u));
  const uint32_t wsrc = wdst - off;
  uint32_t i = 0;
  while (__ballot(active && i < len)) {
    const bool on = active && i < len;
    if (on && st == 0u) {   // offsets below 16 (and offset 0 = zero fill): bytes
      if (off == 0u) for (uint32_t q = 0; q < len; q++) win[(wdst + q) & (D2_WIN - 1u)] = 0;
      else for (uint32_t q = 0; q < len; q++) win[(wdst + q) & (D2_WIN - 1u)] = win[(wsrc + q) & (D2_WIN - 1u)];
      i = len;
    }
    const bool step = on && st != 0u;
    const uint32_t n = step ? (len - i < st ? len - i : st) : 0u;
#ifndef LZ4HIP_D2_EXP
#define LZ4HIP_D2_EXP 0
#endif
    uint32_t dp = (wdst + i) & (D2_WIN - 1u), sp = (wsrc + i) & (D2_WIN - 1u);
    if (LZ4HIP_D2_EXP == 2) { dp &= ~3u; sp &= ~3u; }   // (timing experiment: aligned accesses, wrong bytes)
    const bool wrap = dp + 64u > D2_WIN || sp + 64u > D2_WIN;
    if (__ballot(n && wrap)) {
      if (n && wrap) for (uint32_t q = 0; q < n; q++) win[(wdst + i + q) & (D2_WIN - 1u)] = win[(wsrc + i + q) & (D2_WIN - 1u)];
    }
    const uint32_t m = (wrap || LZ4HIP_D2_EXP == 1) ? 0u : n;   // (experiment 1: no copy at all)
    d2_copy64(win, dp, m, [&](uint32_t o) { uint32_t v; __builtin_memcpy(&v, win + sp + o, 4); return v; });
    if (m && m < 4u) for (uint32_t q = 0; q < m; q++) win[dp + q] = win[sp + q];
    if (step) i += st;
  }
}

// lowest t with pred (D2_NONE if none); all threads must call; two barriers
__device__ __forceinline__ uint32_t d2_block_first(D2Shared& sh, bool pred, uint32_t t) {
  const uint64_t m = __ballot(pred);
  if ((t & 63u) == 0u) sh.firstun[t >> 6] = m ? (t & ~63u) + (uint32_t)__builtin_ctzll(m) : D2_NONE;
  __syncthreads();
  uint32_t f = D2_NONE;
#pragma unroll
  for (uint32_t k = 0; k < D2_NT / 64u; k++) f = sh.firstun[k] < f ? sh.firstun[k] : f;
  __syncthreads();
  return f;
}

// one block per workgroup
__global__ __launch_bounds__(D2_NT) void decode2_copy_kernel(BatchArgs a, const Blk2* meta, const uint4* arena) {
  __shared__ __attribute__((aligned(128))) D2Shared sh;
  const uint32_t b = blockIdx.x, t = threadIdx.x, lane = t & 63u, wv = t >> 6;
  const Blk2 m = meta[b];
  if (m.flags || m.ndesc == 0u || m.first == D2_NONE) return;
  const uint8_t* src = a.src + a.src_off[b];
  uint8_t* dst = a.dst + a.dst_off[b];
  const uint32_t bias = (uint32_t)((uintptr_t)dst & 127u);   // window index = (p + bias) mod 64 KiB: lines of the window are lines of memory
  uint32_t chunk = m.first, in_chunk = 0, remaining = m.ndesc;
  uint32_t opos = 0;   // output position at the start of the batch
  uint32_t fl = 0;     // output below fl is in memory
  auto W = [&](uint32_t p) -> uint8_t& { return sh.win[(p + bias) & (D2_WIN - 1u)]; };

  // whole lines [fl, upto) -> memory (last = true: everything up to `upto`); all threads, no barrier inside
  auto flush = [&](uint32_t upto, bool last) {
    const uint32_t fend = last ? upto : (uint32_t)((((uintptr_t)dst + upto) & ~(uintptr_t)127u) - (uintptr_t)dst);
    if ((int32_t)(fend - fl) <= 0) return;
    uint32_t p = fl;
    const uint32_t mis = (uint32_t)(((uintptr_t)dst + p) & 15u);
    if (mis) {   // head: up to the first 16-byte boundary of memory (start of a block whose destination is not aligned)
      uint32_t hb = 16u - mis;
      if (hb > fend - p) hb = fend - p;
      if (t < hb) dst[p + t] = W(p + t);
      p += hb;
    }
    const uint32_t nb = (fend - p) & ~15u;
    for (uint32_t o = t * 16u; o < nb; o += D2_NT * 16u) {
      const uint4 v = *(const uint4*)&sh.win[(p + o + bias) & (D2_WIN - 1u)];   // (16-byte aligned in the window: bias mirrors memory)
      *(uint4*)(dst + p + o) = v;
    }
    p += nb;
    if (t < fend - p) dst[p + t] = W(p + t);   // tail (< 16 bytes: last flush only)
    fl = fend;
  };
  // all threads: out[s .. s+len) = the bytes `off` before them (byte-forward semantics), len <= 16384.  Everything below s is
  // final.  wlo = lowest block position the window still holds while [s, s+len) is written; older bytes come back from memory,
  // where this workgroup put them (L2-served load).  No barrier inside.
  auto coop_match = [&](uint32_t s, uint32_t off, uint32_t len, uint32_t lo_min) {
    if (off == 0u) { for (uint32_t i = t; i < len; i += D2_NT) W(s + i) = 0; return; }
    uint32_t wlo = s + len > D2_WIN ? s + len - D2_WIN : 0u;
    if (wlo < lo_min) wlo = lo_min;
    for (uint32_t i = t; i < len; i += D2_NT) {
      const uint32_t q = off >= len ? s + i - off : s - off + (i % off);   // (periodic: every byte has a copy within `off` of s)
      W(s + i) = q >= wlo ? (uint8_t)W(q) : (uint8_t)__builtin_nontemporal_load(dst + q);
    }
  };

#ifdef LZ4HIP_D2_DEBUG
  uint64_t T[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = __builtin_readcyclecounter(), tstart = t0;
  uint32_t nbatch = 0, niter = 0, nprod = 0;
#define D2_T(i) do { const uint64_t t1_ = __builtin_readcyclecounter(); T[i] += t1_ - t0; t0 = t1_; } while (0)
#else
#define D2_T(i) do { } while (0)
#endif
  while (remaining) {
    uint32_t cnt = remaining < D2_NT ? remaining : D2_NT;
    if (cnt > D2_PER_CHUNK - in_chunk) cnt = D2_PER_CHUNK - in_chunk;
    const size_t dbase = (size_t)chunk * D2_CHUNK + 1u + in_chunk;
    uint4 d = make_uint4(0u, 0u, 0u, 0u);
    if (t < cnt) d = arena[dbase + t];
    // ---- output positions: workgroup prefix sum of lit + ml ----
    uint32_t wtot;
    uint32_t ex = d2_wave_excl_scan(d.y + d.z, wtot);
    if (lane == 0) sh.wsum[wv] = wtot;
    __syncthreads();
    {
      uint32_t add = 0;
#pragma unroll
      for (uint32_t k = 0; k < D2_NT / 64u; k++) if (k < wv) add += sh.wsum[k];
      ex += add + opos;
    }
    const uint32_t end = ex + d.y + d.z;
    // ---- cut the batch where it would span more than D2_SPAN bytes (positions grow with t: the fitting sequences are a prefix) ----
    const uint32_t take = (uint32_t)__syncthreads_count(t < cnt && end - opos <= D2_SPAN);
    D2_T(0);
    if (take == 0u) {
      // ---- one sequence larger than a batch: the whole workgroup copies it in slices of <= 16 KB, flushing as it goes ----
      const uint4 d0 = arena[dbase];
      uint32_t p = opos;
      for (uint32_t done = 0; done < d0.y;) {
        const uint32_t piece = d0.y - done < 16384u ? d0.y - done : 16384u;
        for (uint32_t i = t; i < piece; i += D2_NT) W(p + i) = src[d0.x + done + i];
        __syncthreads();
        p += piece;
        done += piece;
        flush(p, false);
        __syncthreads();
      }
      const uint32_t ms = p;
      for (uint32_t done = 0; done < d0.z;) {
        const uint32_t piece = d0.z - done < 16384u ? d0.z - done : 16384u;
        // (a slice of a long match is itself a match with the same offset: everything below it is final)
        coop_match(ms + done, d0.w, piece, 0u);
        __syncthreads();
        p += piece;
        done += piece;
        flush(p, false);
        __syncthreads();
      }
      opos = p;
      in_chunk += 1u;
      remaining -= 1u;
    } else {
      const bool mine = t < take;
      const uint32_t lit = mine ? d.y : 0u, ml = mine ? d.z : 0u, off = d.w;
      const uint32_t lstart = ex, mstart = mine ? ex + lit : 0xFFFFFFFFu;
      if (t == take - 1u) sh.misc[0] = end;
      sh.ms[t] = mstart;
      sh.me[t] = mine ? mstart + ml : 0xFFFFFFFFu;
      if (t < D2_NT / 32u) sh.done[t] = 0u;
      __syncthreads();
      const uint32_t batch_end = sh.misc[0];
      const uint32_t ring_lo = batch_end > D2_WIN ? batch_end - D2_WIN : 0u;   // block positions below this have left the window
      if (mine && ml == 0u) atomicOr(&sh.done[t >> 5], 1u << (t & 31u));       // (literal-only sequence: nothing to wait for)
      // gmap: sequence t owns the 64-byte marks in [end of the previous match, end of its own match)
      if (mine) {
        const uint32_t pe = t ? sh.me[t - 1u] : opos, ce = t == take - 1u ? batch_end + 64u : mstart + ml;
        for (uint32_t j = (pe - opos + 63u) >> 6; (j << 6) + opos < ce && j < D2_SPAN / 64u + 2u; j++) sh.gmap[j] = t;
      }
      // ---- literals: memory -> window.  Runs of more than D2_BIGLEN bytes are copied by the owner's whole wavefront ----
      d2_lits_thread(sh.win, lstart + bias, src + d.x, (lit <= D2_BIGLEN) ? lit : 0u);
      for (uint64_t bm = __ballot(lit > D2_BIGLEN); bm; bm &= bm - 1u) {
        const int k = __builtin_ctzll(bm);
        const uint32_t ls = __shfl(lstart, k, 64), ln = __shfl(lit, k, 64), sx = __shfl(d.x, k, 64);
        for (uint32_t i = lane; i < ln; i += 64u) W(ls + i) = src[sx + i];
      }
      __syncthreads();
      D2_T(1);
      // ---- matches: window -> window.  Dependencies: the earlier matches whose output [ms, me) meets this match's source
      // [ss, se) -- sequences ka..kb, found through gmap.  Every wavefront loops over its own 64 sequences: a match whose
      // dependencies are all marked done is copied and marked; no workgroup barrier, progress is guaranteed because the lowest
      // unfinished match of the workgroup never waits.  (A block's dependency DAG is ~100 levels deep on the bench data: what
      // counts is the latency of one level, not the work.) ----
      bool todo = ml != 0u;
      uint32_t ka = 1u, kb = 0u;   // empty range: no dependency inside the batch
      const uint32_t ss = mstart - off;
      const bool far = todo && off != 0u && ss < ring_lo;           // source has left the window: comes back from memory (blocks > 64 KiB)
      if (todo && off != 0u && !far) {
        const uint32_t se = ss + ml < mstart ? ss + ml : mstart;    // (the bytes actually read lie below the match)
        if (se > opos) {
          uint32_t k = sh.gmap[(ss > opos ? ss - opos : 0u) >> 6];
          while (k < t && sh.me[k] <= ss) k++;
          ka = k;
          uint32_t k2 = k;
          while (k2 + 1u < t && sh.ms[k2 + 1u] < se) k2++;
          kb = (k < t && sh.ms[k] < se) ? k2 : (ka = 1u, 0u);
        }
      }
      D2_T(2);
      for (;;) {
#ifdef LZ4HIP_D2_DEBUG
        niter++;
#endif
        if (!__ballot(todo)) break;
        bool ready = false;
        if (todo) {
          ready = true;
          if (ka <= kb) {
            for (uint32_t w0 = ka >> 5; w0 <= (kb >> 5); w0++) {
              const uint32_t lo = w0 == (ka >> 5) ? (ka & 31u) : 0u, hi = w0 == (kb >> 5) ? (kb & 31u) : 31u;
              const uint32_t need = (hi == 31u ? 0xFFFFFFFFu : ((1u << (hi + 1u)) - 1u)) & ~((1u << lo) - 1u);
              if ((__hip_atomic_load(&sh.done[w0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & need) != need) ready = false;
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (!__ballot(ready)) { __builtin_amdgcn_s_sleep(2); D2_T(6); continue; }
        D2_T(6);
#ifdef LZ4HIP_D2_DEBUG
        nprod++;
#endif
        // long matches: the whole wavefront, lowest first
        const uint64_t bigm = __ballot(ready && ml > D2_BIGLEN);
        if (bigm) {
          const int k = __builtin_ctzll(bigm);
          const uint32_t cm = __shfl(mstart, k, 64), cl = __shfl(ml, k, 64), co = __shfl(off, k, 64);
          if (co == 0u) { for (uint32_t i = lane; i < cl; i += 64u) W(cm + i) = 0; }
          else {
            for (uint32_t i = lane; i < cl; i += 64u) {
              const uint32_t q = co >= cl ? cm + i - co : cm - co + (i % co);   // (periodic: every byte has a copy within `co` of cm)
              W(cm + i) = q >= ring_lo ? (uint8_t)W(q) : (uint8_t)__builtin_nontemporal_load(dst + q);
            }
          }
        }
        if (ready && ml <= D2_BIGLEN) {
          if (far) { for (uint32_t i = 0; i < ml; i++) { const uint32_t q = off >= ml ? ss + i : ss + (i % off); W(mstart + i) = q >= ring_lo ? (uint8_t)W(q) : (uint8_t)__builtin_nontemporal_load(dst + q); } }
        }
        d2_match_thread(sh.win, mstart + bias, off, ml, ready && ml <= D2_BIGLEN && !far);
        D2_T(7);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        const bool fin = ready && (ml <= D2_BIGLEN || (int)lane == __builtin_ctzll(bigm | (1ull << 63)) && bigm);
        if (fin) { atomicOr(&sh.done[t >> 5], 1u << (t & 31u)); todo = false; }
      }
      D2_T(3);
      __syncthreads();
      D2_T(4);
#ifdef LZ4HIP_D2_DEBUG
      nbatch++;
#endif
      opos = batch_end;
      in_chunk += take;
      remaining -= take;
      flush(opos, remaining == 0u);
      __syncthreads();
      D2_T(5);
    }
    if (in_chunk == D2_PER_CHUNK && remaining) {
      chunk = ((const uint32_t*)&arena[(size_t)chunk * D2_CHUNK])[0];
      in_chunk = 0;
    }
  }
  flush(opos, true);   // (whatever a final over-sized sequence left below a line boundary)
#ifdef LZ4HIP_D2_DEBUG
  if (b == 7u && (t & 63u) == 0u) printf("d2 block %u wave %u: total %llu cycles, batches %u, iters %u productive %u poll %llu copy %llu | load+scan %llu lits %llu deps %llu matchloop %llu barrier %llu flush %llu\n", b, t >> 6,
      (unsigned long long)(__builtin_readcyclecounter() - tstart), nbatch, niter, nprod, (unsigned long long)T[6], (unsigned long long)T[7], (unsigned long long)T[0], (unsigned long long)T[1], (unsigned long long)T[2], (unsigned long long)T[3], (unsigned long long)T[4], (unsigned long long)T[5]);
#endif
}

}  // namespace lz4hip

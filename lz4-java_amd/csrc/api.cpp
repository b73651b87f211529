// api.cpp -- the C ABI of liblz4hip.so (include/lz4hip.h): device context, staging for the
// host-pointer batch API, contiguous-range sharding over the initialised devices, and the
// device-pointer entry points bench.py times.  There is deliberately NO CPU code path here:
// without a HIP device every compute entry point returns LZ4HIP_E_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <limits.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <algorithm>
#include <vector>
#include <string>
#include "../../include/lz4hip.h"
#include "kernels.h"

namespace {

thread_local std::string g_err;
std::mutex g_mu;
std::vector<int> g_devs;  // HIP device ordinals the engine is initialised on
bool g_inited = false;

int fail(int status, const char* what, hipError_t e = hipSuccess) {
  char buf[512];
  if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  else snprintf(buf, sizeof buf, "%s", what);
  g_err = buf;
  return status;
}

#define HIPCHK(call)                                                   \
  do {                                                                 \
    hipError_t e_ = (call);                                            \
    if (e_ != hipSuccess) return fail(LZ4HIP_E_HIP, #call, e_);        \
  } while (0)

int ensure_init() {
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_inited) return g_devs.empty() ? LZ4HIP_E_NO_DEVICE : LZ4HIP_OK;
  }
  return lz4hip_init(nullptr, 0);
}

// the HIP ordinal of engine device index `device`
int ordinal(int device, int* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (device < 0 || device >= (int)g_devs.size()) return LZ4HIP_E_ARG;
  *out = g_devs[device];
  return LZ4HIP_OK;
}

struct DeviceGuard {  // callers (e.g. PyTorch) own the thread's current device: restore it
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int ord) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = (prev == ord) || hipSetDevice(ord) == hipSuccess;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

enum Op { OP_COMPRESS_FAST, OP_DECODE_SAFE, OP_DECODE_FAST, OP_COMPRESS_HC };
// tuning knobs (lz4hip_set_option): atomics, so that a caller changing one while other threads launch is a race on the VALUE chosen,
// never undefined behaviour; every launch reads each knob once
std::atomic<int> g_decode_lanes{0};   // "decode_lanes"; 0 = kernel default
std::atomic<int> g_decode_stage{-1};  // "decode_stage": 1 = LDS output staging in the plain loop
std::atomic<int> g_decode_pipe{-1};   // "decode_pipe": 3 = ring loop, 2 = deep loop, 1/0 = pipelined interior loop on/off, -1 = kernel default
std::atomic<int> g_decode_ring{0};    // "decode_ring": bytes of the ring loop's output ring (0 = default for the lane count)

// lz4hip_set_option "compress_core": 5 = adaptive two-pass (default): the lean core with a writer wavefront per chain finishes
// the blocks of long sequences, the window-parallel core the others; 3 = lean core only; 1 = window-parallel core only (the
// forced modes exist so that the suite can put every input through each kernel).  "compress_switch" = bytes per sequence below
// which a block counts as dense and goes to the window-parallel core (probe: sequences 32..95 of the block)
std::atomic<int> g_compress_core{5};
std::atomic<int> g_compress_switch{16};   // (tools/switch_probe.py: 20 sent word-like data of 9 bytes per sequence to the slower core; 12 loses text)

// liblz4's level handling: < 1 -> 9, > 12 -> 12; 10..12 are the optimal parser (lz4-java levels 10..17)
int hc_level(int level, int* out) {
  if (level < 1) level = 9;
  if (level > 12) level = 12;
  *out = level;
  return LZ4HIP_OK;
}

// HC on device pointers.  The chain-delta workspace is indexed by source offset, so its size depends on the batch's source span
// (max of src_off + src_len), which lives in device memory: this entry learns it with ONE synchronisation of the caller's stream
// (stated in lz4hip.h); lz4hip_compress_hc_batch_dev_ws takes span and workspace from the caller and never synchronises.
int dev_hc(const lz4hip::BatchArgs& a, int level, hipStream_t st) {
  uint64_t* d_span = nullptr;
  uint64_t span = 0;
  HIPCHK(hipMallocAsync((void**)&d_span, 8, st));
  hipError_t he = hipMemsetAsync(d_span, 0, 8, st);
  int e = 0;
  if (he == hipSuccess) e = lz4hip::launch_hc_span(a.src_off, a.src_len, a.n, d_span, st);
  if (he == hipSuccess && e == 0) he = hipMemcpyAsync(&span, d_span, 8, hipMemcpyDeviceToHost, st);
  if (he == hipSuccess && e == 0) he = hipStreamSynchronize(st);
  (void)hipFreeAsync(d_span, st);   // (on every path: round 1 leaked it when a call above failed)
  if (e) return fail(LZ4HIP_E_HIP, "kernel launch", (hipError_t)e);
  if (he != hipSuccess) return fail(LZ4HIP_E_HIP, "HC span query", he);
  void* ws = nullptr;
  if (hipMallocAsync(&ws, lz4hip::hc_ws_bytes(span, a.n, level), st) != hipSuccess) return fail(LZ4HIP_E_NOMEM, "HC workspace allocation failed");
  e = lz4hip::launch_compress_hc(a, level, ws, span, st);
  (void)hipFreeAsync(ws, st);
  if (e) return fail(LZ4HIP_E_HIP, "kernel launch", (hipError_t)e);
  return LZ4HIP_OK;
}

// number of CUs of the current device (cached per ordinal)
uint32_t cu_count() {
  static std::mutex mu;
  static std::vector<int> cache(64, 0);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  std::lock_guard<std::mutex> lk(mu);
  if (!cache[dev]) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cache[dev] = v;
  }
  return (uint32_t)cache[dev];
}

// fast compress: the cores of kernels.h, selected by "compress_core"
int launch_fast(const lz4hip::BatchArgs& a, hipStream_t st) {
  // scratch: three queue words + the routed-block list of the adaptive scheme
  uint32_t* scratch = nullptr;
  const uint32_t cus = cu_count();
  const int core = g_compress_core.load(std::memory_order_relaxed);
  const size_t mail_words = core != 1 ? lz4hip::compress_fast_v2w_scratch_words(cus) : 0u;   // rings of the finder/writer pairs
  hipError_t e = hipMallocAsync((void**)&scratch, (3 + (size_t)a.n + mail_words) * sizeof(uint32_t), st);
  if (e != hipSuccess) return (int)e;
  const uint32_t dense64 = 64u * (uint32_t)g_compress_switch.load(std::memory_order_relaxed);
  int le;
  switch (core) {
    case 1: le = lz4hip::launch_compress_fast_ms(a, scratch, nullptr, true, cus, st); break;
    case 3: le = lz4hip::launch_compress_fast_v2w(a, scratch, nullptr, 0u, cus, scratch + 3 + a.n, st); break;
    default:   // 5: adaptive
      le = lz4hip::launch_compress_fast_v2w(a, scratch, scratch + 3, dense64, cus, scratch + 3 + a.n, st);
      if (le == 0) le = lz4hip::launch_compress_fast_ms(a, scratch, scratch + 3, false, cus, st);
      break;
  }
  (void)hipFreeAsync(scratch, st);
  return le;
}

// the decode knobs are process-wide and set one at a time: a combination no kernel exists for is said here, by name (round-5 advisor: it
// used to surface as a bare "kernel launch: invalid value")
const char* decode_knobs_error(int lanes, int pipe, int ring) {
  const bool wave_ring = ring == 0 || ring == 8192 || ring == 16384 || ring == 32768 || ring == 65536;
  if (pipe == 4 || pipe == 5) return wave_ring ? nullptr : "decode_pipe 4 / 5 (the wave loops) take decode_ring 0, 8192, 16384, 32768 or 65536";
  if (pipe == 7) return (wave_ring && ring != 8192) ? nullptr : "decode_pipe 7 (the pair loop) takes decode_ring 0, 16384, 32768 or 65536";
  if (pipe == 8) return wave_ring ? nullptr : "decode_pipe 8 (the trio loop) takes decode_ring 0, 8192, 16384, 32768 or 65536";
  if (pipe == 3) {
    const int gl = lanes == 0 ? 4 : lanes;
    const int kw = ring ? ring : (gl == 1 ? 256 : gl == 4 ? 512 : 4096);
    const bool ok = (gl == 1 && (kw == 256 || kw == 512)) || (gl == 4 && (kw == 512 || kw == 1024 || kw == 2048)) ||
                    (gl == 8 && (kw == 512 || kw == 1024 || kw == 2048 || kw == 4096)) || (gl == 16 && (kw == 2048 || kw == 4096));
    return ok ? nullptr : "decode_pipe 3 (the ring loop) takes decode_lanes 1 (decode_ring 256 / 512), 4 (512 .. 2048), 8 (512 .. 4096) or 16 (2048 / 4096)";
  }
  if (lanes == 1) return "decode_lanes 1 exists for decode_pipe 3 (the ring loop) only";
  return nullptr;
}

// The device-side route's word: one of kRouteWords words of a per-device buffer, handed out round robin.  (It was a stream-ordered
// allocation per launch: hipMallocAsync + hipFreeAsync in front of and behind EVERY routed launch -- tens of microseconds on the headline's
// 6 ms decode, measured with the route's kernels in profiles/r06l_bench_kernel_trace.txt.)  A word is only meaningful between a launch's
// route kernel and its candidate kernels, which follow it in stream order: a slot would have to come round again -- 256 routed launches
// later -- while they are still in flight to be disturbed.
constexpr uint32_t kRouteWords = 256u;
uint32_t* g_route_ring[64];
std::atomic<uint32_t> g_route_next[64];
std::mutex g_route_mu;
uint32_t* route_word() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return nullptr;
  uint32_t* ring = __atomic_load_n(&g_route_ring[d], __ATOMIC_ACQUIRE);
  if (!ring) {
    std::lock_guard<std::mutex> lk(g_route_mu);
    ring = g_route_ring[d];
    if (!ring) {
      if (hipMalloc((void**)&ring, kRouteWords * sizeof(uint32_t)) != hipSuccess) return nullptr;
      __atomic_store_n(&g_route_ring[d], ring, __ATOMIC_RELEASE);
    }
  }
  return ring + (g_route_next[d].fetch_add(1u, std::memory_order_relaxed) % kRouteWords);
}
void route_release(int d) {   // lz4hip_shutdown (the device is current)
  std::lock_guard<std::mutex> lk(g_route_mu);
  if (d >= 0 && d < 64 && g_route_ring[d]) { (void)hipFree(g_route_ring[d]); g_route_ring[d] = nullptr; }
}

int launch_decode(const lz4hip::BatchArgs& a, bool safe, hipStream_t st) {
  if (const char* why = decode_knobs_error(g_decode_lanes.load(), g_decode_pipe.load(), g_decode_ring.load())) return fail(LZ4HIP_E_ARG, why);
  uint32_t* route = a.n > 16u * cu_count() ? route_word() : nullptr;   // (nullptr: no device-side route, the lane-group default of the batch size)
  return lz4hip::launch_decompress(a, safe, g_decode_lanes.load(), g_decode_pipe.load(), g_decode_stage.load(), g_decode_ring.load(), st, route);
}

int launch_op(Op op, const lz4hip::BatchArgs& a, hipStream_t st) {
  int e = 0;
  switch (op) {
    case OP_COMPRESS_FAST: e = launch_fast(a, st); break;
    case OP_DECODE_SAFE: e = launch_decode(a, true, st); break;
    case OP_DECODE_FAST: e = launch_decode(a, false, st); break;
    case OP_COMPRESS_HC: return fail(LZ4HIP_E_ARG, "internal: HC goes through dev_hc");
  }
  if (e == LZ4HIP_E_ARG) return e;        // (decode_knobs_error: the message is set)
  if (e != 0) return fail(LZ4HIP_E_HIP, "kernel launch", (hipError_t)e);
  return LZ4HIP_OK;
}

int dev_batch(Op op, const uint8_t* src, const uint64_t* src_off, const int32_t* src_len, uint8_t* dst,
              const uint64_t* dst_off, const int32_t* dst_cap, int32_t* out, uint32_t n, int device, void* stream, int level = 0) {
  int rc = ensure_init();
  if (rc) return fail(rc, "no HIP device: liblz4hip has no CPU fallback");
  if (n == 0) return LZ4HIP_OK;
  if (!src || !src_off || !src_len || !dst || !dst_off || !dst_cap || !out) return fail(LZ4HIP_E_ARG, "null pointer argument");
  int ord;
  if (ordinal(device, &ord)) return fail(LZ4HIP_E_ARG, "bad device index");
  DeviceGuard g(ord);
  if (!g.ok) return fail(LZ4HIP_E_HIP, "hipSetDevice failed");
  lz4hip::BatchArgs a{src, src_off, src_len, dst, dst_off, dst_cap, out, n};
  if (op == OP_COMPRESS_HC) return dev_hc(a, level, (hipStream_t)stream);
  return launch_op(op, a, (hipStream_t)stream);
}

// ---- host-pointer path ---------------------------------------------------------------------------
// ---- staging of the host-pointer batch API --------------------------------------------------------------------------
// A host batch is cut into chunks (64 MiB of source for the codec calls, CHUNK_SRC for the hashes).  Each chunk is packed into a
// PINNED staging buffer (user memory is pageable: hipMemcpyAsync from it would be staged by the runtime at a fraction of the link
// rate), copied to the device, processed, and its destination slots are copied back into a second pinned buffer, from which the
// bytes each block actually produced go to the caller's slots.  Several buffer sets per device rotate, so the CPU packs chunk c+1
// and unpacks chunk c-1 while the GPU works on chunk c.  Buffers, streams and events are created once per device and kept
// (lz4hip_shutdown releases them) -- the per-call hipMalloc/hipFree/stream churn of the first version cost more than the
// work for single blocks.
constexpr size_t CHUNK_SRC = 128u << 20;   // source bytes per chunk (hash path)
inline int env_int(const char* name, int dflt, int lo, int hi) { const char* v = getenv(name); if (!v) return dflt; int x = atoi(v); return x < lo ? lo : x > hi ? hi : x; }
constexpr uint32_t CHUNK_BLOCKS = 1u << 20;
constexpr int COPY_THREADS = 16;           // host threads packing / unpacking a chunk

struct GrowBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool pinned = false;
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    release();
    size_t want = n + n / 4 + 4096;
    hipError_t e = pinned ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; return e; }
    cap = want;
    return hipSuccess;
  }
  void release() {
    if (p) { if (pinned) (void)hipHostFree(p); else (void)hipFree(p); }
    p = nullptr; cap = 0;
  }
};

struct ChunkSlot {
  hipStream_t st = nullptr;                       // kernels (hash path: everything)
  hipStream_t st_in = nullptr, st_out = nullptr;  // host_shard: the H2D and D2H copies run on streams that never see a kernel
  hipStream_t q_in = nullptr, q_out = nullptr;    // the copy streams of the chunk in flight (= st for a call that is one chunk)
  hipEvent_t done = nullptr, ev_in = nullptr, ev_k = nullptr;
  GrowBuf h_src, h_dst, h_meta, d_src, d_dst, d_meta, d_ws, d_pack, d_poff;
  bool packed = false;                // compress ops: only the useful bytes of the slots come back (device-side packing)
  uint32_t i0 = 0, i1 = 0;            // blocks of the chunk in flight
  size_t src_bytes = 0, dst_bytes = 0;
  std::vector<uint64_t> so, dof;      // packed offsets of the chunk's blocks
  ChunkSlot() { h_src.pinned = h_dst.pinned = h_meta.pinned = true; }
};

// One host batch in flight uses a SlotPair (two buffer sets that alternate).  Pairs are pooled per device: a caller takes a free
// pair (or creates one, up to MAX_PAIRS; beyond that it waits for one), works WITHOUT any lock held -- N callers make progress on N
// pairs, each on its own streams -- and hands the pair back.  (Round 1 held one per-device mutex for the whole batch: every
// concurrent caller of the host API ran one at a time.)
struct SlotPair {
  static constexpr int SETS = 6;   // (host_shard uses g_host_sets of them, the hash path alternates between the first two)
  ChunkSlot slot[SETS];
  hipError_t init() {
    for (auto& s : slot) {
      hipError_t e;
      for (hipStream_t* q : {&s.st, &s.st_in, &s.st_out})
        if ((e = hipStreamCreateWithFlags(q, hipStreamNonBlocking)) != hipSuccess) return e;
      for (hipEvent_t* v : {&s.done, &s.ev_in, &s.ev_k})
        if ((e = hipEventCreateWithFlags(v, hipEventDisableTiming)) != hipSuccess) return e;
    }
    return hipSuccess;
  }
  void release() {
    for (auto& s : slot) {
      for (hipStream_t* q : {&s.st, &s.st_in, &s.st_out}) { if (*q) (void)hipStreamDestroy(*q); *q = nullptr; }
      for (hipEvent_t* v : {&s.done, &s.ev_in, &s.ev_k}) { if (*v) (void)hipEventDestroy(*v); *v = nullptr; }
      for (GrowBuf* g : {&s.h_src, &s.h_dst, &s.h_meta, &s.d_src, &s.d_dst, &s.d_meta, &s.d_ws, &s.d_pack, &s.d_poff}) g->release();
    }
  }
  // keeps the staging of the first pairs (the steady state of big batches), drops big staging of the pairs beyond (bursts of callers)
  void trim(size_t keep) {
    for (auto& s : slot)
      for (GrowBuf* g : {&s.h_src, &s.h_dst, &s.d_src, &s.d_dst, &s.d_ws, &s.d_pack})
        if (g->cap > keep) g->release();
  }
};
struct DevCtx {
  static constexpr size_t MAX_PAIRS = 16, KEEP_PAIRS = 2;
  std::mutex mu;                 // guards the pool only (never held while a batch runs)
  std::condition_variable cv;
  std::vector<SlotPair*> all, idle;
  SlotPair* acquire(hipError_t* err) {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      if (!idle.empty()) { SlotPair* p = idle.back(); idle.pop_back(); return p; }
      if (all.size() < MAX_PAIRS) {
        SlotPair* p = new (std::nothrow) SlotPair();
        if (!p) { *err = hipErrorOutOfMemory; return nullptr; }
        all.push_back(p);
        lk.unlock();
        if ((*err = p->init()) != hipSuccess) { lk.lock(); p->release(); all.erase(std::find(all.begin(), all.end(), p)); delete p; return nullptr; }
        return p;
      }
      cv.wait(lk);
    }
  }
  void give_back(SlotPair* p) {
    // (trimming frees pinned and device memory -- hipFree / hipHostFree synchronise the device -- so it happens BEFORE the pool
    // lock is taken: the pair still belongs to this caller, and nobody's acquire / give_back waits behind it)
    bool extra;
    { std::lock_guard<std::mutex> lk(mu); extra = std::find(all.begin(), all.end(), p) - all.begin() >= (ptrdiff_t)KEEP_PAIRS; }
    if (extra) p->trim(8u << 20);
    { std::lock_guard<std::mutex> lk(mu); idle.push_back(p); }
    cv.notify_one();
  }
  void release() {   // lz4hip_shutdown: nothing is in flight
    std::lock_guard<std::mutex> lk(mu);
    for (SlotPair* p : all) { p->release(); delete p; }
    all.clear(); idle.clear();
  }
};
DevCtx g_ctx[64];

template <class F>
void par_blocks(uint32_t i0, uint32_t i1, size_t bytes, F f) {  // f(i) for i in [i0, i1), on several threads when it is worth it
  const uint32_t n = i1 - i0;
  static const int copy_threads = env_int("LZ4HIP_HOST_COPY_THREADS", COPY_THREADS, 1, 128);
  const int T = (bytes < (4u << 20) || n < 2) ? 1 : (int)std::min<uint32_t>((uint32_t)copy_threads, n);
  if (T == 1) { for (uint32_t i = i0; i < i1; i++) f(i); return; }
  // (a helper that cannot be had -- thread or memory exhaustion -- is no error: its share is done right here; nothing is thrown
  // with threads running, which would end the process from a finisher thread)
  std::thread th[128];
  int made = 0;
  for (int t = 0; t < T; t++) {
    const uint32_t a = i0 + (uint32_t)((uint64_t)n * t / T), b = i0 + (uint32_t)((uint64_t)n * (t + 1) / T);
    bool started = false;
    if (t + 1 < T) {   // (the last share is the caller's own)
      try { th[made] = std::thread([=] { for (uint32_t i = a; i < b; i++) f(i); }); made++; started = true; } catch (...) {}
    }
    if (!started) for (uint32_t i = a; i < b; i++) f(i);
  }
  for (int t = 0; t < made; t++) th[t].join();
}

// one device's share [b0, b1) of a host batch.  Three stages run side by side on LZ4HIP_HOST_SETS rotating buffer sets (default 6,
// each a pinned pair of 64 MiB of source + the destination capacity of its chunk, allocated on first use and kept in the per-device
// pool -- up to ~1 GiB of pinned host memory per cached pair at the defaults; LZ4HIP_HOST_SETS=2 / LZ4HIP_HOST_CHUNK_MB lower it):
// the calling thread PACKS chunk
// c + 1 into pinned memory and enqueues it, the GPU works on chunk c (its H2D, kernels and D2H on the set's own stream), and a
// finisher thread per chunk waits for chunk c - 1 and hands its bytes to the caller's slots.  (Rounds 1-2 did pack, the
// synchronous D2H of the packed bytes and the unpack one after the other on the calling thread: 44 ms per GiB, twice what the
// link needs -- tools/host_path_probe.py.)
int host_shard(Op op, int level, int ord, const uint8_t* src, const uint64_t* src_off, const int32_t* src_len, uint8_t* dst,
               const uint64_t* dst_off, const int32_t* dst_cap, int32_t* out, uint32_t b0, uint32_t b1, std::string* err) {
  auto bad_to = [](std::string* where, const char* what, hipError_t e) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    *where = buf;
    return e == hipErrorOutOfMemory ? (int)LZ4HIP_E_NOMEM : (int)LZ4HIP_E_HIP;
  };
  auto bad = [&](const char* what, hipError_t e) { return bad_to(err, what, e); };
  if (b1 == b0) return LZ4HIP_OK;
  if (ord < 0 || ord >= 64) { *err = "device ordinal out of range"; return LZ4HIP_E_ARG; }
  hipError_t e;
  if ((e = hipSetDevice(ord)) != hipSuccess) return bad("hipSetDevice", e);
  DevCtx& cx = g_ctx[ord];
  SlotPair* pair = cx.acquire(&e);
  if (!pair) return bad("stream/event creation", e);
  struct Return { DevCtx& c; SlotPair* p; ~Return() { c.give_back(p); } } give_back_on_exit{cx, pair};
  auto slen_of = [&](uint32_t i) -> size_t { return src_len[i] > 0 ? (size_t)src_len[i] : 0; };
  auto dcap_of = [&](uint32_t i) -> size_t { return dst_cap[i] > 0 ? (size_t)dst_cap[i] : 0; };
  const bool prof = getenv("LZ4HIP_HOST_PROF") != nullptr;   // developer diagnostics: where the time of the stages goes
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  std::atomic<double> t_wait{0}, t_d2h{0}, t_unpack{0};
  auto add = [](std::atomic<double>& a, double v) { double o = a.load(); while (!a.compare_exchange_weak(o, o + v)) {} };

  // hand the results of the chunk in `s` to the caller (waits for it): runs on the chunk's finisher thread
  struct Fin { std::thread th; int rc = LZ4HIP_OK; std::string err; bool live = false; } fin[SlotPair::SETS];
  struct JoinAll {   // no finisher outlives this call, whichever way it is left (an exception of the packing code included)
    Fin* f;
    ~JoinAll() { for (int t = 0; t < SlotPair::SETS; t++) if (f[t].th.joinable()) f[t].th.join(); }
  } join_all{fin};
  auto finish = [&](ChunkSlot& s, Fin& f) -> int {
    if (s.i1 == s.i0) return LZ4HIP_OK;
    hipError_t fe;
    if ((fe = hipSetDevice(ord)) != hipSuccess) return bad_to(&f.err, "hipSetDevice", fe);
    double c0 = now();
    if ((fe = hipEventSynchronize(s.done)) != hipSuccess) return bad_to(&f.err, "hipEventSynchronize", fe);
    add(t_wait, now() - c0); c0 = now();
    const uint32_t nb = s.i1 - s.i0;
    const int32_t* hout = (const int32_t*)((const uint8_t*)s.h_meta.p + (size_t)nb * 24u);
    memcpy(out + s.i0, hout, (size_t)nb * 4u);
    const uint8_t* hd = (const uint8_t*)s.h_dst.p;
    const uint32_t i0 = s.i0;
    if (s.packed) {   // the sizes are here: fetch exactly the useful bytes (already packed on the device), then hand them out
      uint64_t total = 0;
      for (uint32_t t = 0; t < nb; t++) { s.dof[t] = total; total += hout[t] > 0 ? (uint64_t)hout[t] : 0ull; }
      if (s.q_out != s.st && (fe = hipStreamWaitEvent(s.q_out, s.ev_k, 0)) != hipSuccess) return bad_to(&f.err, "hipStreamWaitEvent", fe);   // (the pack kernel)
      if (total && (fe = hipMemcpyAsync(s.h_dst.p, s.d_pack.p, (size_t)total, hipMemcpyDeviceToHost, s.q_out)) != hipSuccess) return bad_to(&f.err, "D2H packed", fe);
      if ((fe = hipStreamSynchronize(s.q_out)) != hipSuccess) return bad_to(&f.err, "hipStreamSynchronize", fe);
      s.dst_bytes = (size_t)total;
    }
    add(t_d2h, now() - c0); c0 = now();
    par_blocks(s.i0, s.i1, s.dst_bytes, [=, &s](uint32_t i) {
      int64_t produced;   // (bytes past a result stay untouched in the caller's slot)
      if (op == OP_DECODE_FAST) produced = out[i] > 0 ? dst_cap[i] : 0;
      else produced = out[i] > 0 ? out[i] : 0;
      if (produced > 0) memcpy(dst + dst_off[i], hd + s.dof[i - i0], (size_t)produced);
    });
    add(t_unpack, now() - c0);
    return LZ4HIP_OK;
  };
  // the buffer set is free again once its finisher is through
  auto reap = [&](int k) -> int {
    Fin& f = fin[k];
    if (!f.live) return LZ4HIP_OK;
    f.th.join();
    f.live = false;
    pair->slot[k].i0 = pair->slot[k].i1 = 0;
    if (f.rc != LZ4HIP_OK) *err = f.err;
    return f.rc;
  };
  auto start_finisher = [&](int k) {
    Fin& f = fin[k];
    ChunkSlot& s = pair->slot[k];
    f.rc = LZ4HIP_OK;
    try {
      f.th = std::thread([&finish, &s, &f] {
        try { f.rc = finish(s, f); }
        catch (...) { f.rc = LZ4HIP_E_NOMEM; f.err = "host staging: out of memory or threads while handing a chunk back"; }
      });
      f.live = true;
    } catch (...) {   // no thread to be had: this chunk is finished on the calling thread
      f.rc = finish(s, f);
      f.live = false;
      s.i0 = s.i1 = 0;
    }
    return f.live ? LZ4HIP_OK : (f.rc != LZ4HIP_OK ? (*err = f.err, f.rc) : LZ4HIP_OK);
  };

  const int SETS = env_int("LZ4HIP_HOST_SETS", 6, 2, SlotPair::SETS);
  const size_t chunk_src = (size_t)env_int("LZ4HIP_HOST_CHUNK_MB", 64, 1, 1024) << 20;
  int rc = LZ4HIP_OK;
  uint32_t i = b0;
  int k = 0;
  double t_reap = 0, t_pack = 0, t_enq = 0, t_alloc = 0;
  const double t_begin = now();
  while (i < b1 && rc == LZ4HIP_OK) {
    ChunkSlot& s = pair->slot[k];
    double t0 = now();
    if ((rc = reap(k)) != LZ4HIP_OK) break;   // the chunk that used this buffer set SETS rounds ago
    t_reap += now() - t0; t0 = now();
    // the next chunk: blocks [i, j)
    uint32_t j = i;
    size_t sb = 0, db = 0;
    while (j < b1 && j - i < CHUNK_BLOCKS) {
      const size_t a = (slen_of(j) + 15u) & ~(size_t)15u, c = (dcap_of(j) + 15u) & ~(size_t)15u;
      if (j > i && sb + a > chunk_src) break;
      sb += a; db += c; j++;
    }
    const uint32_t nb = j - i;
    s.so.resize(nb); s.dof.resize(nb);
    { size_t so = 0, dofs = 0;
      for (uint32_t t = 0; t < nb; t++) { s.so[t] = so; s.dof[t] = dofs; so += (slen_of(i + t) + 15u) & ~(size_t)15u; dofs += (dcap_of(i + t) + 15u) & ~(size_t)15u; } }
    const size_t meta = (size_t)nb * 28u;   // so[nb] u64 | dof[nb] u64 | src_len[nb] | dst_cap[nb] | out[nb]
    if ((e = s.h_src.reserve(sb + 64)) != hipSuccess || (e = s.h_dst.reserve(db + 64)) != hipSuccess || (e = s.h_meta.reserve(meta)) != hipSuccess ||
        (e = s.d_src.reserve(sb + 64)) != hipSuccess || (e = s.d_dst.reserve(db + 64)) != hipSuccess || (e = s.d_meta.reserve(meta)) != hipSuccess ||
        (op == OP_COMPRESS_HC && (e = s.d_ws.reserve(lz4hip::hc_ws_bytes(sb, nb, level))) != hipSuccess) ||
        ((op == OP_COMPRESS_FAST || op == OP_COMPRESS_HC) && ((e = s.d_pack.reserve(db + 64)) != hipSuccess || (e = s.d_poff.reserve((size_t)nb * 8u)) != hipSuccess))) {
      rc = bad("staging allocation", e);
      break;
    }
    t_alloc += now() - t0; t0 = now();
    uint8_t* hs = (uint8_t*)s.h_src.p;
    { const uint32_t base = i;
      par_blocks(i, j, sb, [=, &s](uint32_t t) { if (src_len[t] > 0) memcpy(hs + s.so[t - base], src + src_off[t], (size_t)src_len[t]); }); }
    uint8_t* hm = (uint8_t*)s.h_meta.p;
    memcpy(hm, s.so.data(), (size_t)nb * 8u);
    memcpy(hm + (size_t)nb * 8u, s.dof.data(), (size_t)nb * 8u);
    memcpy(hm + (size_t)nb * 16u, src_len + i, (size_t)nb * 4u);
    memcpy(hm + (size_t)nb * 20u, dst_cap + i, (size_t)nb * 4u);
    t_pack += now() - t0; t0 = now();
    uint8_t* dm = (uint8_t*)s.d_meta.p;
    // a call that is ONE chunk (single blocks, small batches) has nothing to overlap: one stream, no finisher thread
    const bool single = (i == b0 && j == b1);
    s.q_in = single ? s.st : s.st_in;
    s.q_out = single ? s.st : s.st_out;
    if (sb && (e = hipMemcpyAsync(s.d_src.p, hs, sb, hipMemcpyHostToDevice, s.q_in)) != hipSuccess) { rc = bad("H2D src", e); break; }
    if ((e = hipMemcpyAsync(dm, hm, (size_t)nb * 24u, hipMemcpyHostToDevice, s.q_in)) != hipSuccess) { rc = bad("H2D meta", e); break; }
    if (!single && ((e = hipEventRecord(s.ev_in, s.q_in)) != hipSuccess || (e = hipStreamWaitEvent(s.st, s.ev_in, 0)) != hipSuccess)) { rc = bad("H2D event", e); break; }
    lz4hip::BatchArgs a{(const uint8_t*)s.d_src.p, (const uint64_t*)dm, (const int32_t*)(dm + (size_t)nb * 16u), (uint8_t*)s.d_dst.p,
                        (const uint64_t*)(dm + (size_t)nb * 8u), (const int32_t*)(dm + (size_t)nb * 20u), (int32_t*)(dm + (size_t)nb * 24u), nb};
    int le = 0;
    switch (op) {
      case OP_COMPRESS_FAST: le = launch_fast(a, s.st); break;
      case OP_DECODE_SAFE: le = launch_decode(a, true, s.st); break;
      case OP_DECODE_FAST: le = launch_decode(a, false, s.st); break;
      case OP_COMPRESS_HC: le = lz4hip::launch_compress_hc(a, level, s.d_ws.p, sb, s.st); break;
    }
    if (le == LZ4HIP_E_ARG) { *err = lz4hip_last_error(); rc = le; break; }   // (a decode knob combination without a kernel: launch_decode has said which, on this thread)
    if (le) { rc = bad("kernel launch", (hipError_t)le); break; }
    s.packed = (op == OP_COMPRESS_FAST || op == OP_COMPRESS_HC);
    // the sizes travel first; compress ops: the finisher fetches exactly the packed bytes once it has them
    if (!single && ((e = hipEventRecord(s.ev_k, s.st)) != hipSuccess || (e = hipStreamWaitEvent(s.q_out, s.ev_k, 0)) != hipSuccess)) { rc = bad("kernel event", e); break; }
    if ((e = hipMemcpyAsync(hm + (size_t)nb * 24u, dm + (size_t)nb * 24u, (size_t)nb * 4u, hipMemcpyDeviceToHost, s.q_out)) != hipSuccess) { rc = bad("D2H out", e); break; }
    if (s.packed) {
      if ((e = hipEventRecord(s.done, s.q_out)) != hipSuccess) { rc = bad("hipEventRecord", e); break; }   // sizes on the host
      if ((le = lz4hip::launch_pack(a, (uint64_t*)s.d_poff.p, (uint8_t*)s.d_pack.p, s.st)) != 0) { rc = bad("kernel launch", (hipError_t)le); break; }
      if (!single && (e = hipEventRecord(s.ev_k, s.st)) != hipSuccess) { rc = bad("hipEventRecord", e); break; }   // packed bytes ready (the finisher waits for it)
    } else {
      if (db && (e = hipMemcpyAsync(s.h_dst.p, s.d_dst.p, db, hipMemcpyDeviceToHost, s.q_out)) != hipSuccess) { rc = bad("D2H dst", e); break; }
      if ((e = hipEventRecord(s.done, s.q_out)) != hipSuccess) { rc = bad("hipEventRecord", e); break; }
    }
    s.i0 = i; s.i1 = j; s.src_bytes = sb; s.dst_bytes = db;
    if (single) {
      Fin& f = fin[k];
      if ((rc = finish(s, f)) != LZ4HIP_OK) *err = f.err;
      s.i0 = s.i1 = 0;
    } else {
      rc = start_finisher(k);
    }
    i = j;
    k = (k + 1) % SETS;
    t_enq += now() - t0;
  }
  const double t_loop = now() - t_begin;
  // drain (also after an error: nothing may stay in flight on the cached buffers, no finisher may outlive this call)
  for (int t = 0; t < SETS; t++) {
    const int kk = (k + t) % SETS;
    std::string keep = *err;
    const int r = reap(kk);
    if (rc == LZ4HIP_OK) rc = r; else *err = keep;   // (the first error is the one reported)
    ChunkSlot& s = pair->slot[kk];
    if (rc != LZ4HIP_OK) { (void)hipStreamSynchronize(s.st_in); (void)hipStreamSynchronize(s.st); (void)hipStreamSynchronize(s.st_out); s.i0 = s.i1 = 0; }
  }
  if (prof)
    fprintf(stderr, "[lz4hip host_shard op %d, %u blocks] total %.1f ms: loop %.1f (reap %.1f | alloc %.1f | pack %.1f | enqueue %.1f), drain %.1f; finishers: wait %.1f + d2h %.1f + unpack %.1f\n",
            (int)op, b1 - b0, 1e3 * (now() - t_begin), 1e3 * t_loop, 1e3 * t_reap, 1e3 * t_alloc, 1e3 * t_pack, 1e3 * t_enq,
            1e3 * (now() - t_begin - t_loop), 1e3 * t_wait.load(), 1e3 * t_d2h.load(), 1e3 * t_unpack.load());
  return rc;
}

int host_batch(Op op, const uint8_t* src, const uint64_t* src_off, const int32_t* src_len, uint8_t* dst,
               const uint64_t* dst_off, const int32_t* dst_cap, int32_t* out, uint32_t n, int level = 0) {
  int rc = ensure_init();
  if (rc) return fail(rc, "no HIP device: liblz4hip has no CPU fallback");
  if (n == 0) return LZ4HIP_OK;
  if (!src || !src_off || !src_len || !dst || !dst_off || !dst_cap || !out) return fail(LZ4HIP_E_ARG, "null pointer argument");
  std::vector<int> devs;
  { std::lock_guard<std::mutex> lk(g_mu); devs = g_devs; }
  int prev = -1;
  (void)hipGetDevice(&prev);
  // contiguous block ranges per device; small batches stay on one device
  const uint32_t D = (uint32_t)std::min<size_t>(devs.size(), std::max<uint32_t>(1u, n / 64u));
  std::vector<int> rcs(D, 0);
  std::vector<std::string> errs(D);
  if (D == 1) {
    rcs[0] = host_shard(op, level, devs[0], src, src_off, src_len, dst, dst_off, dst_cap, out, 0, n, &errs[0]);
  } else {
    std::vector<std::thread> th;
    for (uint32_t d = 0; d < D; d++) {
      const uint32_t b0 = (uint32_t)((uint64_t)n * d / D), b1 = (uint32_t)((uint64_t)n * (d + 1) / D);
      th.emplace_back([&, d, b0, b1] { rcs[d] = host_shard(op, level, devs[d], src, src_off, src_len, dst, dst_off, dst_cap, out, b0, b1, &errs[d]); });
    }
    for (auto& t : th) t.join();
  }
  if (prev >= 0) (void)hipSetDevice(prev);
  for (uint32_t d = 0; d < D; d++)
    if (rcs[d]) return fail(rcs[d], errs[d].c_str());
  return LZ4HIP_OK;
}

// hashes of one device's share [b0, b1) of a host batch: the same pinned, double-buffered staging as host_shard (chunks of
// <= CHUNK_SRC bytes packed by several host threads while the previous chunk is on the GPU), from the same pool of buffer pairs
template <class T>
int xxh_shard(bool is64, int ord, const uint8_t* buf, const uint64_t* off, const int32_t* len, uint64_t seed, T* out, uint32_t b0, uint32_t b1, std::string* err) {
  auto bad = [&](const char* what, hipError_t e) {
    char m[512];
    snprintf(m, sizeof m, "%s: %s", what, hipGetErrorString(e));
    *err = m;
    return e == hipErrorOutOfMemory ? (int)LZ4HIP_E_NOMEM : (int)LZ4HIP_E_HIP;
  };
  if (b1 == b0) return LZ4HIP_OK;
  hipError_t e;
  if ((e = hipSetDevice(ord)) != hipSuccess) return bad("hipSetDevice", e);
  DevCtx& cx = g_ctx[ord];
  SlotPair* pair = cx.acquire(&e);
  if (!pair) return bad("stream/event creation", e);
  struct Return { DevCtx& c; SlotPair* p; ~Return() { c.give_back(p); } } give_back_on_exit{cx, pair};
  auto len_of = [&](uint32_t i) -> size_t { return len[i] > 0 ? (size_t)len[i] : 0; };
  auto finish = [&](ChunkSlot& s) -> int {
    if (s.i1 == s.i0) return LZ4HIP_OK;
    if ((e = hipEventSynchronize(s.done)) != hipSuccess) return bad("hipEventSynchronize", e);
    const uint32_t nb = s.i1 - s.i0;
    memcpy(out + s.i0, (const uint8_t*)s.h_meta.p + (((size_t)nb * 12u + 7u) & ~(size_t)7u), (size_t)nb * sizeof(T));
    s.i0 = s.i1 = 0;
    return LZ4HIP_OK;
  };
  int rc = LZ4HIP_OK;
  uint32_t i = b0;
  int k = 0;
  while (i < b1 && rc == LZ4HIP_OK) {
    ChunkSlot& s = pair->slot[k];
    if ((rc = finish(s)) != LZ4HIP_OK) break;
    uint32_t j = i;
    size_t sb = 0;
    while (j < b1 && j - i < CHUNK_BLOCKS) {
      const size_t a = (len_of(j) + 15u) & ~(size_t)15u;
      if (j > i && sb + a > CHUNK_SRC) break;
      sb += a; j++;
    }
    const uint32_t nb = j - i;
    s.so.resize(nb);
    { size_t so = 0; for (uint32_t t = 0; t < nb; t++) { s.so[t] = so; so += (len_of(i + t) + 15u) & ~(size_t)15u; } }
    const size_t hoff = ((size_t)nb * 12u + 7u) & ~(size_t)7u;   // off[nb] u64 | len[nb] i32 | (pad to 8) | hash[nb]
    const size_t meta = hoff + (size_t)nb * sizeof(T);
    if ((e = s.h_src.reserve(sb + 64)) != hipSuccess || (e = s.h_meta.reserve(meta)) != hipSuccess || (e = s.d_src.reserve(sb + 64)) != hipSuccess ||
        (e = s.d_meta.reserve(meta)) != hipSuccess) { rc = bad("staging allocation", e); break; }
    uint8_t* hs = (uint8_t*)s.h_src.p;
    { const uint32_t base = i;
      par_blocks(i, j, sb, [=, &s](uint32_t t) { if (len[t] > 0) memcpy(hs + s.so[t - base], buf + off[t], (size_t)len[t]); }); }
    uint8_t* hm = (uint8_t*)s.h_meta.p;
    memcpy(hm, s.so.data(), (size_t)nb * 8u);
    memcpy(hm + (size_t)nb * 8u, len + i, (size_t)nb * 4u);
    uint8_t* dm = (uint8_t*)s.d_meta.p;
    if (sb && (e = hipMemcpyAsync(s.d_src.p, hs, sb, hipMemcpyHostToDevice, s.st)) != hipSuccess) { rc = bad("H2D src", e); break; }
    if ((e = hipMemcpyAsync(dm, hm, (size_t)nb * 12u, hipMemcpyHostToDevice, s.st)) != hipSuccess) { rc = bad("H2D meta", e); break; }
    const int le = is64 ? lz4hip::launch_xxh64((const uint8_t*)s.d_src.p, (const uint64_t*)dm, (const int32_t*)(dm + (size_t)nb * 8u), seed, (uint64_t*)(dm + hoff), nb, s.st)
                        : lz4hip::launch_xxh32((const uint8_t*)s.d_src.p, (const uint64_t*)dm, (const int32_t*)(dm + (size_t)nb * 8u), (uint32_t)seed, (uint32_t*)(dm + hoff), nb, s.st);
    if (le) { rc = bad("kernel launch", (hipError_t)le); break; }
    if ((e = hipMemcpyAsync(hm + hoff, dm + hoff, (size_t)nb * sizeof(T), hipMemcpyDeviceToHost, s.st)) != hipSuccess) { rc = bad("D2H hashes", e); break; }
    if ((e = hipEventRecord(s.done, s.st)) != hipSuccess) { rc = bad("hipEventRecord", e); break; }
    s.i0 = i; s.i1 = j;
    i = j;
    k ^= 1;
  }
  for (int t = 0; t < 2; t++) {
    ChunkSlot& s = pair->slot[(k + t) & 1];
    if (rc == LZ4HIP_OK) rc = finish(s);
    else { (void)hipStreamSynchronize(s.st); s.i0 = s.i1 = 0; }
  }
  return rc;
}

template <class T>
int host_xxh(bool is64, const uint8_t* buf, const uint64_t* off, const int32_t* len, uint64_t seed, T* out, uint32_t n) {
  int rc = ensure_init();
  if (rc) return fail(rc, "no HIP device: liblz4hip has no CPU fallback");
  if (n == 0) return LZ4HIP_OK;
  if (!buf || !off || !len || !out) return fail(LZ4HIP_E_ARG, "null pointer argument");
  std::vector<int> devs;
  { std::lock_guard<std::mutex> lk(g_mu); devs = g_devs; }
  int prev = -1;
  (void)hipGetDevice(&prev);
  // contiguous ranges per device, like the LZ4 batches; small batches stay on one device
  const uint32_t D = (uint32_t)std::min<size_t>(devs.size(), std::max<uint32_t>(1u, n / 1024u));
  std::vector<int> rcs(D, 0);
  std::vector<std::string> errs(D);
  if (D == 1) {
    rcs[0] = xxh_shard<T>(is64, devs[0], buf, off, len, seed, out, 0, n, &errs[0]);
  } else {
    std::vector<std::thread> th;
    for (uint32_t d = 0; d < D; d++) {
      const uint32_t b0 = (uint32_t)((uint64_t)n * d / D), b1 = (uint32_t)((uint64_t)n * (d + 1) / D);
      th.emplace_back([&, d, b0, b1] { rcs[d] = xxh_shard<T>(is64, devs[d], buf, off, len, seed, out, b0, b1, &errs[d]); });
    }
    for (auto& t : th) t.join();
  }
  if (prev >= 0) (void)hipSetDevice(prev);
  for (uint32_t d = 0; d < D; d++)
    if (rcs[d]) return fail(rcs[d], errs[d].c_str());
  return LZ4HIP_OK;
}

// ---- single-block calls: coalesced ---------------------------------------------------------------------------------------
// The reference's API shape is one block per call from many threads (LZ4Compressor.compress, LZ4Compressor.java:59,82).  A GPU
// launch per block would serialise those threads on launch + PCIe latency, so concurrent single-block calls of the same kind are
// combined: a caller queues its request; whoever finds no batch in progress becomes the leader, takes EVERYTHING queued so far and
// runs it as one host batch; requests that arrive meanwhile form the next batch (led by one of their own callers).  A lone caller
// pays no waiting window; N concurrent callers share one launch.
struct Req {
  const uint8_t* src; int32_t len; uint8_t* dst; int32_t cap;
  int32_t out = 0; int rc = 0; bool done = false; std::string err;
};
struct Combiner {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<Req*> q;
  bool leader = false;
};
Combiner g_comb[4][13];   // [op][HC level]

// one host batch for all of `batch`; returns its rc (every request gets its out[]); may throw (allocation of the index vectors,
// thread creation inside host_batch)
int run_batch_once(Op op, int level, const std::vector<Req*>& batch) {
  static uint8_t dummy_in = 0, dummy_out = 0;
  const uint32_t n = (uint32_t)batch.size();
  const uint8_t* sbase = nullptr; uint8_t* dbase = nullptr;
  for (Req* r : batch) {
    if (!r->src) r->src = &dummy_in;
    if (!r->dst) r->dst = &dummy_out;
    if (!sbase || r->src < sbase) sbase = r->src;
    if (!dbase || r->dst < dbase) dbase = r->dst;
  }
  std::vector<uint64_t> so(n), dof(n);
  std::vector<int32_t> sl(n), dc(n), out(n, 0);
  for (uint32_t i = 0; i < n; i++) { so[i] = (uint64_t)(batch[i]->src - sbase); dof[i] = (uint64_t)(batch[i]->dst - dbase); sl[i] = batch[i]->len; dc[i] = batch[i]->cap; }
  const int rc = host_batch(op, sbase, so.data(), sl.data(), dbase, dof.data(), dc.data(), out.data(), n, level);
  for (uint32_t i = 0; i < n; i++) { batch[i]->rc = rc; batch[i]->out = out[i]; if (rc) batch[i]->err = g_err; }
  return rc;
}

// Combined callers do not share a fate: when the shared batch fails as a whole (staging could not be allocated because of ONE
// caller's huge block, a HIP error) every request is run again on its own, so only the caller whose request is the cause sees
// the failure.  Exceptions (bad_alloc of the index vectors, system_error from thread creation) become LZ4HIP_E_NOMEM for the
// requests concerned; nothing propagates into the leader's bookkeeping.
void run_combined(Op op, int level, std::vector<Req*>& batch) noexcept {
  int rc;
  try { rc = run_batch_once(op, level, batch); }
  catch (...) { rc = LZ4HIP_E_NOMEM; for (Req* r : batch) { r->rc = rc; r->out = 0; r->err = "out of memory while combining single-block calls"; } }
  if (rc == 0 || batch.size() < 2) return;
  for (Req* r : batch) {
    std::vector<Req*> one;
    try { one.push_back(r); (void)run_batch_once(op, level, one); }
    catch (...) { r->rc = LZ4HIP_E_NOMEM; r->out = 0; r->err = "out of memory"; }
  }
}

int single(Op op, const uint8_t* src, int src_len, uint8_t* dst, int dst_cap, int level = 0) {
  Req r{src, src_len, dst, dst_cap};
  Combiner& c = g_comb[(int)op][level < 0 || level > 12 ? 0 : level];
  {
    std::unique_lock<std::mutex> lk(c.mu);
    try { c.q.push_back(&r); }
    catch (...) { return fail(LZ4HIP_E_NOMEM, "out of memory"); }
    while (!r.done) {
      if (!c.leader) {
        c.leader = true;
        std::vector<Req*> batch;
        batch.swap(c.q);          // (contains r: only a leader takes requests out of the queue; swap does not throw)
        lk.unlock();
        run_combined(op, level, batch);   // noexcept: the leader always comes back to hand the results out
        lk.lock();
        for (Req* x : batch) x->done = true;
        c.leader = false;
        c.cv.notify_all();
      } else {
        c.cv.wait(lk);
      }
    }
  }
  if (r.rc) { g_err = r.err; return LZ4HIP_LIB_ERROR(r.rc); }
  return r.out;
}

}  // namespace

// ---- device-side container assembly (kernels.hip launch_container_blocks) ------------------------------------------------------
namespace {
struct ContainerPlan { size_t cws, hcws, qwords, total; uint32_t n; };
// the largest CU count of any visible device: what a workspace must be sized for when the device is not known yet
uint32_t max_cu_count() {
  static std::atomic<uint32_t> cached{0};
  uint32_t v = cached.load(std::memory_order_relaxed);
  if (v) return v;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  for (int d = 0; d < n; d++) {
    int c = 0;
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess && (uint32_t)c > v) v = (uint32_t)c;
  }
  if (!v) v = 256;
  cached.store(v, std::memory_order_relaxed);
  return v;
}
// cus: the CU count the launch will use (cu_count() under the DeviceGuard of the launch device) or max_cu_count() for a query
ContainerPlan container_plan(uint64_t n_bytes, uint32_t block_size, int hc_lv, uint32_t cus) {
  ContainerPlan p{};
  p.n = (uint32_t)((n_bytes + block_size - 1u) / block_size);
  p.cws = (lz4hip::container_ws_bytes(n_bytes, block_size) + 255u) & ~(size_t)255u;
  p.hcws = hc_lv > 0 ? ((lz4hip::hc_ws_bytes(n_bytes, p.n, hc_lv) + 255u) & ~(size_t)255u) : 0u;
  p.qwords = 3u + (size_t)p.n + lz4hip::compress_fast_v2w_scratch_words(cus);
  p.total = p.cws + p.hcws + p.qwords * sizeof(uint32_t) + 256u;
  return p;
}
int container_args(int kind, uint64_t n_bytes, uint32_t block_size, int level, int* hc_lv) {
  *hc_lv = 0;
  if (kind != 0 && kind != 1) return fail(LZ4HIP_E_ARG, "container kind must be 0 (LZ4 Frame blocks) or 1 (LZ4Block blocks)");
  if (block_size < 64u || block_size > (32u << 20)) return fail(LZ4HIP_E_ARG, "block size must be 64 .. 32 MiB");   // LZ4BlockOutputStream.java:50-51
  if (n_bytes / block_size >= 0x7FFFFFFFull) return fail(LZ4HIP_E_ARG, "too many blocks");
  if (level != 0 && hc_level(level, hc_lv)) return fail(LZ4HIP_E_UNSUPPORTED, "unsupported compression level");
  return 0;
}
}  // namespace

extern "C" {
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif
size_t lz4hip_container_workspace_bytes(uint64_t n_bytes, uint32_t block_size, int level) {
  int lv = 0;
  if (ensure_init() || block_size < 64u || (level != 0 && hc_level(level, &lv))) return 0;
  return container_plan(n_bytes, block_size, lv, max_cu_count()).total;   // (enough for whichever device the launch names)
}
int lz4hip_container_blocks_dev(int kind, int flags, int level, const uint8_t* src, uint64_t n_bytes, uint32_t block_size, uint8_t* dst, uint64_t dst_cap,
                                uint64_t* total_dev, void* ws, size_t ws_bytes, int device, void* stream) {
  int rc = ensure_init();
  if (rc) return fail(rc, "no HIP device: liblz4hip has no CPU fallback");
  int lv;
  if ((rc = container_args(kind, n_bytes, block_size, level, &lv)) != 0) return rc;
  if ((n_bytes && !src) || !dst || !total_dev || !ws) return fail(LZ4HIP_E_ARG, "null pointer argument");
  int ord;
  if (ordinal(device, &ord)) return fail(LZ4HIP_E_ARG, "bad device index");
  DeviceGuard g(ord);   // (before the plan: its ring words are a function of the LAUNCH device's CU count)
  const ContainerPlan p = container_plan(n_bytes, block_size, lv, cu_count());
  if (ws_bytes < p.total) return fail(LZ4HIP_E_ARG, "container workspace too small (lz4hip_container_workspace_bytes)");
  uint8_t* w = (uint8_t*)(((uintptr_t)ws + 255u) & ~(uintptr_t)255u);
  const int e = lz4hip::launch_container_blocks(kind, flags & 1, lv, src, n_bytes, block_size, dst, dst_cap, (unsigned long long*)total_dev, w,
                                                p.hcws ? w + p.cws : nullptr, (uint32_t*)(w + p.cws + p.hcws),
                                                64u * (uint32_t)g_compress_switch.load(std::memory_order_relaxed), cu_count(),
                                                g_compress_core.load(std::memory_order_relaxed), stream);
  if (e) return fail(LZ4HIP_E_HIP, "kernel launch", (hipError_t)e);
  return LZ4HIP_OK;
}
int lz4hip_container_blocks(int kind, int flags, int level, const uint8_t* src, uint64_t n_bytes, uint32_t block_size, uint8_t* dst, uint64_t dst_cap,
                            uint64_t* out_bytes) {
  int rc = ensure_init();
  if (rc) return fail(rc, "no HIP device: liblz4hip has no CPU fallback");
  int lv;
  if ((rc = container_args(kind, n_bytes, block_size, level, &lv)) != 0) return rc;
  if ((n_bytes && !src) || !dst || !out_bytes) return fail(LZ4HIP_E_ARG, "null pointer argument");
  *out_bytes = 0;
  if (n_bytes == 0) return LZ4HIP_OK;
  int ord;
  if (ordinal(0, &ord)) return fail(LZ4HIP_E_NO_DEVICE, "no device");
  DeviceGuard g(ord);
  const ContainerPlan p = container_plan(n_bytes, block_size, lv, cu_count());
  const uint64_t worst = n_bytes + (uint64_t)p.n * (kind == 0 ? 8u : 21u);
  hipStream_t st = nullptr;
  uint8_t *d_src = nullptr, *d_dst = nullptr, *d_ws = nullptr;
  uint64_t* d_total = nullptr;
  hipError_t he = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  auto done = [&](int r) {
    if (d_src) (void)hipFreeAsync(d_src, st);
    if (d_dst) (void)hipFreeAsync(d_dst, st);
    if (d_ws) (void)hipFreeAsync(d_ws, st);
    if (d_total) (void)hipFreeAsync(d_total, st);
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    return r;
  };
  if (he != hipSuccess) return fail(LZ4HIP_E_HIP, "hipStreamCreate", he);
  if ((he = hipMallocAsync((void**)&d_src, n_bytes, st)) != hipSuccess || (he = hipMallocAsync((void**)&d_dst, worst, st)) != hipSuccess ||
      (he = hipMallocAsync((void**)&d_ws, p.total, st)) != hipSuccess || (he = hipMallocAsync((void**)&d_total, 8, st)) != hipSuccess)
    return done(fail(he == hipErrorOutOfMemory ? LZ4HIP_E_NOMEM : LZ4HIP_E_HIP, "device allocation", he));
  if ((he = hipMemcpyAsync(d_src, src, n_bytes, hipMemcpyHostToDevice, st)) != hipSuccess) return done(fail(LZ4HIP_E_HIP, "H2D", he));
  uint8_t* w = (uint8_t*)(((uintptr_t)d_ws + 255u) & ~(uintptr_t)255u);
  const int e = lz4hip::launch_container_blocks(kind, flags & 1, lv, d_src, n_bytes, block_size, d_dst, worst, (unsigned long long*)d_total, w,
                                                p.hcws ? w + p.cws : nullptr, (uint32_t*)(w + p.cws + p.hcws),
                                                64u * (uint32_t)g_compress_switch.load(std::memory_order_relaxed), cu_count(),
                                                g_compress_core.load(std::memory_order_relaxed), st);
  if (e) return done(fail(LZ4HIP_E_HIP, "kernel launch", (hipError_t)e));
  uint64_t total = 0;
  if ((he = hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, st)) != hipSuccess || (he = hipStreamSynchronize(st)) != hipSuccess)
    return done(fail(LZ4HIP_E_HIP, "container size", he));
  if (total > dst_cap) return done(fail(LZ4HIP_E_ARG, "destination too small for the container blocks"));
  if (total && (he = hipMemcpyAsync(dst, d_dst, total, hipMemcpyDeviceToHost, st)) != hipSuccess) return done(fail(LZ4HIP_E_HIP, "D2H", he));
  if ((he = hipStreamSynchronize(st)) != hipSuccess) return done(fail(LZ4HIP_E_HIP, "hipStreamSynchronize", he));
  *out_bytes = total;
  return done(LZ4HIP_OK);
}
size_t lz4hip_container_decode_workspace_bytes(uint32_t n_max) { return lz4hip::container_read_ws_bytes(n_max) + 256u; }
int lz4hip_container_decode_dev(int kind, int flags, const uint8_t* body, uint64_t body_bytes, uint32_t max_block, uint8_t* dst, uint64_t slot_bytes,
                                uint32_t n_max, int32_t* sizes_dev, uint64_t* info_dev, void* ws, size_t ws_bytes, int device, void* stream) {
  int rc = ensure_init();
  if (rc) return fail(rc, "no HIP device: liblz4hip has no CPU fallback");
  if (kind != 0 && kind != 1) return fail(LZ4HIP_E_ARG, "container kind must be 0 (LZ4 Frame blocks) or 1 (LZ4Block blocks)");
  if ((body_bytes && !body) || !dst || !sizes_dev || !info_dev || !ws) return fail(LZ4HIP_E_ARG, "null pointer argument");
  if (n_max == 0 || n_max > 0x7FFFFFFFu) return fail(LZ4HIP_E_ARG, "n_max must be 1 .. 2^31 - 1");
  if (slot_bytes == 0 || slot_bytes > 0x7FFFFFFFull || (kind == 0 && (max_block == 0 || slot_bytes < max_block)))
    return fail(LZ4HIP_E_ARG, "slot_bytes must be 1 .. 2^31 - 1 and, for frame blocks, at least the frame's block maximum size");
  if (ws_bytes < lz4hip_container_decode_workspace_bytes(n_max)) return fail(LZ4HIP_E_ARG, "workspace too small (lz4hip_container_decode_workspace_bytes)");
  int ord;
  if (ordinal(device, &ord)) return fail(LZ4HIP_E_ARG, "bad device index");
  DeviceGuard g(ord);
  uint8_t* w = (uint8_t*)(((uintptr_t)ws + 255u) & ~(uintptr_t)255u);
  const int e = lz4hip::launch_container_read(kind, flags & 1, body, body_bytes, max_block, dst, slot_bytes, n_max, sizes_dev,
                                              (unsigned long long*)info_dev, w, stream);
  if (e) return fail(LZ4HIP_E_HIP, "kernel launch", (hipError_t)e);
  return LZ4HIP_OK;
}
// Host-side walk of the size words / headers of a container body IN HOST MEMORY (no device work, no decoding): how many whole blocks
// of the first n_max the body holds, how many bytes their decoded forms can need at most (frame: max_block per compressed block, the
// stored size of a raw one; LZ4Block: the header's original length) and
// the largest of those bounds (incl. the block the walk stopped at, when its header is readable).  It follows the device walk's
// positions (container_walk_kernel) and stops where that stops.  What it is for (round-4 advisor): lz4hip_container_decode sized its
// device slots -- and its callers their destination -- by max_block x n_max whatever the body held: 256 MiB for a 100-byte frame,
// gigabytes for a 21-byte LZ4Block header with level nibble 15.
static void container_prewalk(int kind, int flags, const uint8_t* body, uint64_t body_bytes, uint32_t max_block, uint32_t n_max,
                              uint32_t* n_blocks, uint64_t* dst_bytes, uint64_t* slot_max) {
  auto rd32 = [](const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); };
  uint64_t p = 0, total = 0, smax = 0;
  uint32_t k = 0;
  while (k < n_max && p < body_bytes) {
    uint64_t need, bound;
    if (kind == 0) {
      if (p + 4u > body_bytes) break;
      const uint32_t word = rd32(body + p), size = word & 0x7FFFFFFFu;
      if (size == 0u || size > max_block) break;
      bound = (word & 0x80000000u) ? size : max_block;   // (a compressed block decodes into max_block bytes of capacity, as in the reference: a smaller one could change which check a damaged stream fails first)
      need = 4ull + size + ((flags & 1) ? 4u : 0u);
    } else {
      if (p + 21u > body_bytes) break;
      const uint8_t* h = body + p;
      if (memcmp(h, "LZ4Block", 8) != 0) break;
      const uint32_t method = h[8] & 0xF0u, level = 10u + (h[8] & 0x0Fu);
      const int32_t clen = (int32_t)rd32(h + 9), olen = (int32_t)rd32(h + 13);
      if ((method != 0x10u && method != 0x20u) || olen > (int32_t)(1u << level) || olen <= 0 || clen <= 0 || (method == 0x10u && olen != clen)) break;
      // (round-5 advisor) the device walk stops at a block bigger than the caller's max_block (CR_BLOCK_TOO_BIG: the readers take the
      // host walk, which allocates one block at a time like LZ4BlockInputStream.java:236-253), at a block whose payload is cut short
      // (CR_TRUNCATED) and at a header that announces more than its compressed bytes can decode to -- clen bytes of LZ4 are at most
      // 255 x clen bytes (CR_CORRUPT: the reference's decompressor would fail on it) -- in that order, and none of the three sizes a
      // slot.  Without these rules 256 headers {level nibble 15, olen 32 MiB, clen 1} -- 5 KB of input -- sized 8 GiB of device slots
      // and 8 GiB of destination before block 0 was found corrupt
      if ((uint64_t)olen > max_block) break;
      bound = (uint64_t)olen;
      need = 21ull + (uint64_t)clen;
      if (p + need > body_bytes) break;
      if (method == 0x20u && bound > 255ull * (uint64_t)clen + 64u) break;
    }
    if (bound > smax) smax = bound;             // (frames: also for a block whose payload is cut short: the device walk looks at it)
    if (p + need > body_bytes) break;
    total += bound; p += need; k++;
  }
  *n_blocks = k; *dst_bytes = total; *slot_max = smax;
}
int lz4hip_container_decode_bound(int kind, int flags, const uint8_t* body, uint64_t body_bytes, uint32_t max_block, uint32_t n_max,
                                  uint32_t* n_blocks, uint64_t* dst_bytes) {
  if (kind != 0 && kind != 1) return fail(LZ4HIP_E_ARG, "container kind must be 0 (LZ4 Frame blocks) or 1 (LZ4Block blocks)");
  if ((body_bytes && !body) || !n_blocks || !dst_bytes) return fail(LZ4HIP_E_ARG, "null pointer argument");
  if (n_max == 0 || n_max > 0x7FFFFFFFu || max_block == 0 || max_block > 0x7FFFFFFFu) return fail(LZ4HIP_E_ARG, "n_max and max_block must be 1 .. 2^31 - 1");
  uint64_t smax = 0;
  container_prewalk(kind, flags, body, body_bytes, max_block, n_max, n_blocks, dst_bytes, &smax);
  return LZ4HIP_OK;
}
// host pointers: H2D of the container bytes, the device path above, D2H of the delivered blocks back to back into dst
int lz4hip_container_decode(int kind, int flags, const uint8_t* body, uint64_t body_bytes, uint32_t max_block, uint32_t n_max, uint8_t* dst,
                            uint64_t dst_cap, int32_t* sizes, uint64_t* info) {
  int rc = ensure_init();
  if (rc) return fail(rc, "no HIP device: liblz4hip has no CPU fallback");
  if (kind != 0 && kind != 1) return fail(LZ4HIP_E_ARG, "container kind must be 0 (LZ4 Frame blocks) or 1 (LZ4Block blocks)");
  if ((body_bytes && !body) || !sizes || !info || (dst_cap && !dst)) return fail(LZ4HIP_E_ARG, "null pointer argument");
  if (n_max == 0 || n_max > 0x7FFFFFFFu || max_block == 0 || max_block > 0x7FFFFFFFu) return fail(LZ4HIP_E_ARG, "n_max and max_block must be 1 .. 2^31 - 1");
  for (int i = 0; i < 5; i++) info[i] = 0;
  int ord;
  if (ordinal(0, &ord)) return fail(LZ4HIP_E_NO_DEVICE, "no device");
  DeviceGuard g(ord);
  // slots and their number by what the body holds, not by what the caller allows: one slot more than the whole blocks present (the
  // device walk finds out why the run ends there), each as big as the largest decoded block can be
  uint64_t slot_need;
  {
    uint32_t nb = 0; uint64_t db = 0, smax = 0;
    container_prewalk(kind, flags, body, body_bytes, max_block, n_max, &nb, &db, &smax);
    if (nb + 1u < n_max) n_max = nb + 1u;
    slot_need = smax ? smax : 1u;
  }
  const size_t wsb = lz4hip_container_decode_workspace_bytes(n_max);
  const uint64_t slot = (slot_need + 63u) & ~63ull;
  hipStream_t st = nullptr;
  uint8_t *d_body = nullptr, *d_dst = nullptr, *d_ws = nullptr;
  int32_t* d_sizes = nullptr;
  uint64_t* d_info = nullptr;
  hipError_t he = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  auto done = [&](int r) {
    if (d_body) (void)hipFreeAsync(d_body, st);
    if (d_dst) (void)hipFreeAsync(d_dst, st);
    if (d_ws) (void)hipFreeAsync(d_ws, st);
    if (d_sizes) (void)hipFreeAsync(d_sizes, st);
    if (d_info) (void)hipFreeAsync(d_info, st);
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    return r;
  };
  if (he != hipSuccess) return fail(LZ4HIP_E_HIP, "hipStreamCreate", he);
  if ((he = hipMallocAsync((void**)&d_body, body_bytes ? body_bytes : 1, st)) != hipSuccess || (he = hipMallocAsync((void**)&d_dst, slot * n_max, st)) != hipSuccess ||
      (he = hipMallocAsync((void**)&d_ws, wsb, st)) != hipSuccess || (he = hipMallocAsync((void**)&d_sizes, sizeof(int32_t) * (size_t)n_max, st)) != hipSuccess ||
      (he = hipMallocAsync((void**)&d_info, 5 * sizeof(uint64_t), st)) != hipSuccess)
    return done(fail(he == hipErrorOutOfMemory ? LZ4HIP_E_NOMEM : LZ4HIP_E_HIP, "device allocation", he));
  if (body_bytes && (he = hipMemcpyAsync(d_body, body, body_bytes, hipMemcpyHostToDevice, st)) != hipSuccess) return done(fail(LZ4HIP_E_HIP, "H2D", he));
  uint8_t* w = (uint8_t*)(((uintptr_t)d_ws + 255u) & ~(uintptr_t)255u);
  const int e = lz4hip::launch_container_read(kind, flags & 1, d_body, body_bytes, max_block, d_dst, slot, n_max, d_sizes, (unsigned long long*)d_info, w, st);
  if (e) return done(fail(LZ4HIP_E_HIP, "kernel launch", (hipError_t)e));
  if ((he = hipMemcpyAsync(info, d_info, 5 * sizeof(uint64_t), hipMemcpyDeviceToHost, st)) != hipSuccess ||
      (he = hipMemcpyAsync(sizes, d_sizes, sizeof(int32_t) * (size_t)n_max, hipMemcpyDeviceToHost, st)) != hipSuccess ||
      (he = hipStreamSynchronize(st)) != hipSuccess)
    return done(fail(LZ4HIP_E_HIP, "container result", he));
  const uint64_t n_ok = info[0];
  if (info[3] > dst_cap) return done(fail(LZ4HIP_E_ARG, "destination too small for the decoded blocks"));
  // runs of full slots go back in one copy each (the usual case: every block but the last is a whole block)
  uint64_t o = 0;
  for (uint64_t k = 0; k < n_ok;) {
    uint64_t j = k;
    while (j < n_ok && (uint64_t)sizes[j] == slot) j++;
    if (j > k) {
      if ((he = hipMemcpyAsync(dst + o, d_dst + k * slot, (j - k) * slot, hipMemcpyDeviceToHost, st)) != hipSuccess) return done(fail(LZ4HIP_E_HIP, "D2H", he));
      o += (j - k) * slot; k = j;
    } else {
      const uint64_t sz = sizes[k] > 0 ? (uint64_t)sizes[k] : 0;
      if (sz && (he = hipMemcpyAsync(dst + o, d_dst + k * slot, sz, hipMemcpyDeviceToHost, st)) != hipSuccess) return done(fail(LZ4HIP_E_HIP, "D2H", he));
      o += sz; k++;
    }
  }
  if ((he = hipStreamSynchronize(st)) != hipSuccess) return done(fail(LZ4HIP_E_HIP, "hipStreamSynchronize", he));
  return done(LZ4HIP_OK);
}
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
}  // extern "C"

// ---- streaming xxhash (StreamingXXHash32JNI / StreamingXXHash64JNI: XXHashJNI.c:89-145, :199-255) ------------------------
struct lz4hip_xxh_stream {
  bool is64;
  int ord;                 // HIP ordinal the record lives on (the engine's first device, or the device of update_dev)
  void* rec = nullptr;     // device record (kernels.h)
  void* stage = nullptr;   // device staging for host-pointer updates
  size_t stage_cap = 0;
  uint64_t seed;
  bool pending_reset = true;   // the record is (re)initialised by the next launch
  // ordering of launches on different streams: every launch records `ev`; the next launch's stream waits for it on the device, the
  // digest and the reuse of the staging buffer wait for it on the host.  (`launched` and not "last stream != NULL": the null stream
  // is a stream too -- a host update() followed by update_dev() on a non-blocking stream raced on the record in round 1.)
  hipEvent_t ev = nullptr;
  hipStream_t last = nullptr;
  bool launched = false;
  std::mutex mu;               // the reference's methods are `synchronized`
};
namespace {
constexpr size_t XXH_STAGE_MAX = 64u << 20;
int xxh_stream_create(bool is64, uint64_t seed, lz4hip_xxh_stream** out) {
  int rc = ensure_init();
  if (rc) return fail(rc, "no HIP device: liblz4hip has no CPU fallback");
  if (!out) return fail(LZ4HIP_E_ARG, "null pointer argument");
  int ord;
  if (ordinal(0, &ord)) return fail(LZ4HIP_E_NO_DEVICE, "no device");
  DeviceGuard g(ord);
  auto* st = new (std::nothrow) lz4hip_xxh_stream();
  if (!st) return fail(LZ4HIP_E_NOMEM, "out of memory");
  st->is64 = is64;
  st->ord = ord;
  st->seed = seed;
  if (hipMalloc(&st->rec, lz4hip::xxh_stream_rec_bytes(is64)) != hipSuccess) {
    delete st;
    return fail(LZ4HIP_E_NOMEM, "hipMalloc failed");
  }
  *out = st;
  return LZ4HIP_OK;
}
// absorbs a DEVICE buffer (len may be 0: only applies a pending reset)
int xxh_stream_launch(lz4hip_xxh_stream* st, const uint8_t* dbuf, uint32_t len, hipStream_t stream) {
  if (!st->ev) HIPCHK(hipEventCreateWithFlags(&st->ev, hipEventDisableTiming));
  if (st->launched && st->last != stream) HIPCHK(hipStreamWaitEvent(stream, st->ev, 0));   // updates of one hash are ordered
  const int reset = st->pending_reset ? 1 : 0;
  int e = st->is64 ? lz4hip::launch_xxh64_stream(st->rec, dbuf, len, reset, st->seed, stream)
                   : lz4hip::launch_xxh32_stream(st->rec, dbuf, len, reset, (uint32_t)st->seed, stream);
  if (e) return fail(LZ4HIP_E_HIP, "kernel launch", (hipError_t)e);
  st->pending_reset = false;
  st->last = stream;
  st->launched = true;
  HIPCHK(hipEventRecord(st->ev, stream));
  return LZ4HIP_OK;
}
template <class T>
int xxh_stream_digest(lz4hip_xxh_stream* st, bool is64, T* out) {
  if (!st || !out || st->is64 != is64) return fail(LZ4HIP_E_ARG, "bad stream handle");
  std::lock_guard<std::mutex> lk(st->mu);
  DeviceGuard g(st->ord);
  if (st->pending_reset) {  // nothing absorbed since create/reset: the digest of the empty stream
    int rc = xxh_stream_launch(st, (const uint8_t*)st->rec, 0, nullptr);
    if (rc) return rc;
  }
  if (st->launched) HIPCHK(hipEventSynchronize(st->ev));
  HIPCHK(hipMemcpy(out, (const uint8_t*)st->rec + lz4hip::xxh_stream_digest_offset(is64), sizeof(T), hipMemcpyDeviceToHost));
  return LZ4HIP_OK;
}
}  // namespace

extern "C" {

int lz4hip_version(void) { return LZ4HIP_VERSION; }
const char* lz4hip_last_error(void) { return g_err.c_str(); }

int lz4hip_init(const int* device_ids, int n_devices) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_inited) return g_devs.empty() ? LZ4HIP_E_NO_DEVICE : LZ4HIP_OK;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_inited = true;  // remember: there is nothing to run on (and nothing else to fall back to)
    g_devs.clear();
    return fail(LZ4HIP_E_NO_DEVICE, "hipGetDeviceCount found no device", e);
  }
  std::vector<int> devs;
  if (device_ids && n_devices > 0) {
    for (int i = 0; i < n_devices; i++) {
      if (device_ids[i] < 0 || device_ids[i] >= count) return fail(LZ4HIP_E_ARG, "device id out of range");
      devs.push_back(device_ids[i]);
    }
  } else {
    for (int i = 0; i < count; i++) devs.push_back(i);
  }
  g_devs = devs;
  g_inited = true;
  return LZ4HIP_OK;
}

void lz4hip_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  int prev = -1;
  (void)hipGetDevice(&prev);
  for (int ord : g_devs) {
    if (ord < 0 || ord >= 64) continue;
    if (hipSetDevice(ord) == hipSuccess) { g_ctx[ord].release(); route_release(ord); }
  }
  if (prev >= 0) (void)hipSetDevice(prev);
  g_devs.clear();
  g_inited = false;
}

int lz4hip_device_count(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  return (int)g_devs.size();
}

#ifdef LZ4HIP_RING_DBG
}  // extern "C"
namespace lz4hip { int ring_stats_fetch(unsigned long long* out8); }
extern "C" {
__attribute__((visibility("default"))) int lz4hip_dbg_ring_stats(unsigned long long* out8) { return lz4hip::ring_stats_fetch(out8); }
#endif
int lz4hip_last_decode_route(int device, uint32_t* out6) {
  int rc = ensure_init();
  if (rc) return fail(rc, "no HIP device: liblz4hip has no CPU fallback");
  if (!out6) return fail(LZ4HIP_E_ARG, "null pointer argument");
  int ord;
  if (ordinal(device, &ord)) return fail(LZ4HIP_E_ARG, "bad device index");
  DeviceGuard g(ord);
  const int e = lz4hip::last_decode_route(out6);
  if (e) return fail(LZ4HIP_E_HIP, "hipMemcpyFromSymbol", (hipError_t)e);
  return LZ4HIP_OK;
}
int lz4hip_set_option(const char* name, int value) {
  if (name && strcmp(name, "decode_stage") == 0) {
    if (value < -1 || value > 1) return fail(LZ4HIP_E_ARG, "decode_stage must be -1, 0 or 1");
    g_decode_stage = value;
    return LZ4HIP_OK;
  }
  if (name && strcmp(name, "decode_pipe") == 0) {
    if (value < -1 || value > 8 || value == 6) return fail(LZ4HIP_E_ARG, "decode_pipe must be -1 .. 5, 7 or 8");
    g_decode_pipe = value;
    return LZ4HIP_OK;
  }
  if (name && strcmp(name, "decode_route_short") == 0) {   // average output bytes per sampled sequence up to which a batch of more than 16 blocks per CU goes to the wave kernel; 0: never
    if (value < 0 || value > 255) return fail(LZ4HIP_E_ARG, "decode_route_short must be 0 .. 255");
    lz4hip::set_route_short(value);
    return LZ4HIP_OK;
  }
  if (name && strcmp(name, "decode_ring") == 0) {
    if (value != 0 && value != 256 && value != 512 && value != 1024 && value != 2048 && value != 4096 && value != 8192 && value != 16384 && value != 32768 && value != 65536)
      return fail(LZ4HIP_E_ARG, "decode_ring must be 0, 256 .. 4096 (ring loop) or 8192 .. 65536 (wave loop)");
    g_decode_ring = value;
    return LZ4HIP_OK;
  }
  if (name && strcmp(name, "decode_lanes") == 0) {
    if (value != 0 && value != 1 && value != 4 && value != 8 && value != 16 && value != 32 && value != 64) return fail(LZ4HIP_E_ARG, "decode_lanes must be 0,1 (ring loop only),4,8,16,32,64");
    g_decode_lanes = value;
    return LZ4HIP_OK;
  }
  if (name && strcmp(name, "compress_core") == 0) {
    if (value != 1 && value != 3 && value != 5) return fail(LZ4HIP_E_ARG, "compress_core must be 1, 3 or 5");
    g_compress_core = value;
    return LZ4HIP_OK;
  }
  if (name && strcmp(name, "compress_pack") == 0) {
    if (value != 0 && value != 1) return fail(LZ4HIP_E_ARG, "compress_pack must be 0 or 1");
    lz4hip::set_compress_pack(value);
    return LZ4HIP_OK;
  }
  if (name && strcmp(name, "compress_switch") == 0) {
    if (value < 0 || value > 1024) return fail(LZ4HIP_E_ARG, "compress_switch must be 0..1024");
    g_compress_switch = value;
    return LZ4HIP_OK;
  }
  return fail(LZ4HIP_E_ARG, "unknown option");
}

int lz4hip_compress_bound(int n) {
  if (n < 0 || (uint32_t)n > 0x7E000000u) return 0;
  return n + n / 255 + 16;
}

// ---- host-pointer batch API ----
int lz4hip_compress_fast_batch(const uint8_t* src, const uint64_t* src_off, const int32_t* src_len, uint8_t* dst,
                               const uint64_t* dst_off, const int32_t* dst_cap, int32_t* out_len, uint32_t n) {
  return host_batch(OP_COMPRESS_FAST, src, src_off, src_len, dst, dst_off, dst_cap, out_len, n);
}
int lz4hip_decompress_safe_batch(const uint8_t* src, const uint64_t* src_off, const int32_t* src_len, uint8_t* dst,
                                 const uint64_t* dst_off, const int32_t* dst_cap, int32_t* out_len, uint32_t n) {
  return host_batch(OP_DECODE_SAFE, src, src_off, src_len, dst, dst_off, dst_cap, out_len, n);
}
int lz4hip_decompress_fast_batch(const uint8_t* src, const uint64_t* src_off, const int32_t* src_cap, uint8_t* dst,
                                 const uint64_t* dst_off, const int32_t* dst_len, int32_t* out_consumed, uint32_t n) {
  return host_batch(OP_DECODE_FAST, src, src_off, src_cap, dst, dst_off, dst_len, out_consumed, n);
}
int lz4hip_compress_hc_batch(const uint8_t* src, const uint64_t* src_off, const int32_t* src_len, uint8_t* dst,
                             const uint64_t* dst_off, const int32_t* dst_cap, int32_t* out_len, uint32_t n, int level) {
  int lv;
  if (hc_level(level, &lv)) return LZ4HIP_E_UNSUPPORTED;
  return host_batch(OP_COMPRESS_HC, src, src_off, src_len, dst, dst_off, dst_cap, out_len, n, lv);
}
int lz4hip_xxh32_batch(const uint8_t* buf, const uint64_t* off, const int32_t* len, uint32_t seed, uint32_t* out, uint32_t n) {
  return host_xxh<uint32_t>(false, buf, off, len, seed, out, n);
}
int lz4hip_xxh64_batch(const uint8_t* buf, const uint64_t* off, const int32_t* len, uint64_t seed, uint64_t* out, uint32_t n) {
  return host_xxh<uint64_t>(true, buf, off, len, seed, out, n);
}

// ---- device-pointer batch API ----
int lz4hip_compress_fast_batch_dev(const uint8_t* src, const uint64_t* src_off, const int32_t* src_len, uint8_t* dst,
                                   const uint64_t* dst_off, const int32_t* dst_cap, int32_t* out_len, uint32_t n, int device, void* stream) {
  return dev_batch(OP_COMPRESS_FAST, src, src_off, src_len, dst, dst_off, dst_cap, out_len, n, device, stream);
}
int lz4hip_decompress_safe_batch_dev(const uint8_t* src, const uint64_t* src_off, const int32_t* src_len, uint8_t* dst,
                                     const uint64_t* dst_off, const int32_t* dst_cap, int32_t* out_len, uint32_t n, int device, void* stream) {
  return dev_batch(OP_DECODE_SAFE, src, src_off, src_len, dst, dst_off, dst_cap, out_len, n, device, stream);
}
int lz4hip_decompress_fast_batch_dev(const uint8_t* src, const uint64_t* src_off, const int32_t* src_cap, uint8_t* dst,
                                     const uint64_t* dst_off, const int32_t* dst_len, int32_t* out_consumed, uint32_t n, int device, void* stream) {
  return dev_batch(OP_DECODE_FAST, src, src_off, src_cap, dst, dst_off, dst_len, out_consumed, n, device, stream);
}
size_t lz4hip_hc_workspace_bytes(uint64_t src_span, uint32_t n_blocks, int level) {
  int lv;
  if (hc_level(level, &lv)) return 0;
  return lz4hip::hc_ws_bytes(src_span, n_blocks, lv);
}
int lz4hip_compress_hc_batch_dev_ws(const uint8_t* src, const uint64_t* src_off, const int32_t* src_len, uint8_t* dst, const uint64_t* dst_off,
                                    const int32_t* dst_cap, int32_t* out_len, uint32_t n, int level, int device, void* stream,
                                    uint64_t src_span, void* ws, size_t ws_bytes) {
  int lv;
  if (hc_level(level, &lv)) return LZ4HIP_E_UNSUPPORTED;
  int rc = ensure_init();
  if (rc) return fail(rc, "no HIP device: liblz4hip has no CPU fallback");
  if (n == 0) return LZ4HIP_OK;
  if (!src || !src_off || !src_len || !dst || !dst_off || !dst_cap || !out_len || !ws) return fail(LZ4HIP_E_ARG, "null pointer argument");
  if (ws_bytes < lz4hip::hc_ws_bytes(src_span, n, lv)) return fail(LZ4HIP_E_ARG, "HC workspace too small for the source span");
  int ord;
  if (ordinal(device, &ord)) return fail(LZ4HIP_E_ARG, "bad device index");
  DeviceGuard g(ord);
  if (!g.ok) return fail(LZ4HIP_E_HIP, "hipSetDevice failed");
  lz4hip::BatchArgs a{src, src_off, src_len, dst, dst_off, dst_cap, out_len, n};
  const int e = lz4hip::launch_compress_hc(a, lv, ws, src_span, (hipStream_t)stream);
  return e ? fail(LZ4HIP_E_HIP, "kernel launch", (hipError_t)e) : LZ4HIP_OK;
}
int lz4hip_compress_hc_batch_dev(const uint8_t* src, const uint64_t* src_off, const int32_t* src_len, uint8_t* dst,
                                 const uint64_t* dst_off, const int32_t* dst_cap, int32_t* out_len, uint32_t n, int level, int device, void* stream) {
  int lv;
  if (hc_level(level, &lv)) return LZ4HIP_E_UNSUPPORTED;
  return dev_batch(OP_COMPRESS_HC, src, src_off, src_len, dst, dst_off, dst_cap, out_len, n, device, stream, lv);
}
int lz4hip_xxh32_batch_dev(const uint8_t* buf, const uint64_t* off, const int32_t* len, uint32_t seed, uint32_t* out, uint32_t n, int device, void* stream) {
  int rc = ensure_init();
  if (rc) return fail(rc, "no HIP device: liblz4hip has no CPU fallback");
  if (n == 0) return LZ4HIP_OK;
  if (!buf || !off || !len || !out) return fail(LZ4HIP_E_ARG, "null pointer argument");
  int ord;
  if (ordinal(device, &ord)) return fail(LZ4HIP_E_ARG, "bad device index");
  DeviceGuard g(ord);
  int e = lz4hip::launch_xxh32(buf, off, len, seed, out, n, stream);
  return e ? fail(LZ4HIP_E_HIP, "kernel launch", (hipError_t)e) : LZ4HIP_OK;
}
int lz4hip_xxh64_batch_dev(const uint8_t* buf, const uint64_t* off, const int32_t* len, uint64_t seed, uint64_t* out, uint32_t n, int device, void* stream) {
  int rc = ensure_init();
  if (rc) return fail(rc, "no HIP device: liblz4hip has no CPU fallback");
  if (n == 0) return LZ4HIP_OK;
  if (!buf || !off || !len || !out) return fail(LZ4HIP_E_ARG, "null pointer argument");
  int ord;
  if (ordinal(device, &ord)) return fail(LZ4HIP_E_ARG, "bad device index");
  DeviceGuard g(ord);
  int e = lz4hip::launch_xxh64(buf, off, len, seed, out, n, stream);
  return e ? fail(LZ4HIP_E_HIP, "kernel launch", (hipError_t)e) : LZ4HIP_OK;
}

#ifdef LZ4HIP_DEV_TOOLS
// developer diagnostics (see include/lz4hip.h)
int lz4hip_dbg_compress_fast_profile_dev(const uint8_t* src, const uint64_t* src_off, const int32_t* src_len, uint8_t* dst,
                                         const uint64_t* dst_off, const int32_t* dst_cap, int32_t* out_len, uint32_t n, uint64_t* prof,
                                         int device, void* stream) {
  int rc = ensure_init();
  if (rc) return fail(rc, "no HIP device: liblz4hip has no CPU fallback");
  if (n == 0) return LZ4HIP_OK;
  if (!src || !src_off || !src_len || !dst || !dst_off || !dst_cap || !out_len || !prof) return fail(LZ4HIP_E_ARG, "null pointer argument");
  int ord;
  if (ordinal(device, &ord)) return fail(LZ4HIP_E_ARG, "bad device index");
  DeviceGuard g(ord);
  lz4hip::BatchArgs a{src, src_off, src_len, dst, dst_off, dst_cap, out_len, n};
  int e = lz4hip::launch_compress_fast_prof(a, prof, g_compress_core.load() == 1 ? 1 : 3, stream);
  return e ? fail(LZ4HIP_E_HIP, "kernel launch", (hipError_t)e) : LZ4HIP_OK;
}
#endif  // LZ4HIP_DEV_TOOLS

// ---- single-block convenience ----
int lz4hip_compress_fast(const uint8_t* src, int src_len, uint8_t* dst, int dst_cap) { return single(OP_COMPRESS_FAST, src, src_len, dst, dst_cap); }
int lz4hip_compress_hc(const uint8_t* src, int src_len, uint8_t* dst, int dst_cap, int level) {
  int lv;
  if (hc_level(level, &lv)) return LZ4HIP_LIB_ERROR(LZ4HIP_E_UNSUPPORTED);
  return single(OP_COMPRESS_HC, src, src_len, dst, dst_cap, lv);
}
int lz4hip_decompress_safe(const uint8_t* src, int src_len, uint8_t* dst, int dst_cap) { return single(OP_DECODE_SAFE, src, src_len, dst, dst_cap); }
int lz4hip_decompress_fast(const uint8_t* src, int src_cap, uint8_t* dst, int dst_len) { return single(OP_DECODE_FAST, src, src_cap, dst, dst_len); }
int lz4hip_xxh32(const uint8_t* buf, int len, uint32_t seed, uint32_t* out) {
  const uint64_t zero = 0;
  int32_t l = len;
  uint8_t dummy = 0;
  return lz4hip_xxh32_batch(buf ? buf : &dummy, &zero, &l, seed, out, 1);
}
int lz4hip_xxh64(const uint8_t* buf, int len, uint64_t seed, uint64_t* out) {
  const uint64_t zero = 0;
  int32_t l = len;
  uint8_t dummy = 0;
  return lz4hip_xxh64_batch(buf ? buf : &dummy, &zero, &l, seed, out, 1);
}

// ---- streaming xxhash: thin wrappers over the helpers above the extern "C" block ----
int lz4hip_xxh32_stream_create(uint32_t seed, lz4hip_xxh_stream** out) { return xxh_stream_create(false, seed, out); }
int lz4hip_xxh64_stream_create(uint64_t seed, lz4hip_xxh_stream** out) { return xxh_stream_create(true, seed, out); }
int lz4hip_xxh_stream_reset(lz4hip_xxh_stream* st, uint64_t seed) {
  if (!st) return fail(LZ4HIP_E_ARG, "bad stream handle");
  std::lock_guard<std::mutex> lk(st->mu);
  st->seed = st->is64 ? seed : (uint64_t)(uint32_t)seed;
  st->pending_reset = true;
  return LZ4HIP_OK;
}
int lz4hip_xxh_stream_update(lz4hip_xxh_stream* st, const uint8_t* buf, int len) {
  if (!st || len < 0 || (!buf && len)) return fail(LZ4HIP_E_ARG, "bad argument");
  if (len == 0) return LZ4HIP_OK;
  std::lock_guard<std::mutex> lk(st->mu);
  DeviceGuard g(st->ord);
  size_t done = 0;
  while (done < (size_t)len) {
    const size_t part = std::min((size_t)len - done, XXH_STAGE_MAX);
    if (st->stage_cap < part) {
      if (st->launched) HIPCHK(hipEventSynchronize(st->ev));
      if (st->stage) (void)hipFree(st->stage);
      st->stage = nullptr;
      st->stage_cap = 0;
      size_t cap = 1u << 16;
      while (cap < part) cap <<= 1;
      if (hipMalloc(&st->stage, cap) != hipSuccess) return fail(LZ4HIP_E_NOMEM, "hipMalloc failed");
      st->stage_cap = cap;
    }
    if (st->launched) HIPCHK(hipEventSynchronize(st->ev));  // the staging buffer is reused
    HIPCHK(hipMemcpy(st->stage, buf + done, part, hipMemcpyHostToDevice));
    int rc = xxh_stream_launch(st, (const uint8_t*)st->stage, (uint32_t)part, nullptr);
    if (rc) return rc;
    done += part;
  }
  return LZ4HIP_OK;
}
int lz4hip_xxh_stream_update_dev(lz4hip_xxh_stream* st, const uint8_t* dbuf, int len, void* stream) {
  if (!st || len < 0 || (!dbuf && len)) return fail(LZ4HIP_E_ARG, "bad argument");
  if (len == 0) return LZ4HIP_OK;
  std::lock_guard<std::mutex> lk(st->mu);
  DeviceGuard g(st->ord);
  return xxh_stream_launch(st, dbuf, (uint32_t)len, (hipStream_t)stream);
}
int lz4hip_xxh32_stream_digest(lz4hip_xxh_stream* st, uint32_t* out) { return xxh_stream_digest<uint32_t>(st, false, out); }
int lz4hip_xxh64_stream_digest(lz4hip_xxh_stream* st, uint64_t* out) { return xxh_stream_digest<uint64_t>(st, true, out); }
void lz4hip_xxh_stream_free(lz4hip_xxh_stream* st) {
  if (!st) return;
  {
    std::lock_guard<std::mutex> lk(st->mu);
    DeviceGuard g(st->ord);
    if (st->launched) (void)hipEventSynchronize(st->ev);
    if (st->ev) (void)hipEventDestroy(st->ev);
    if (st->rec) (void)hipFree(st->rec);
    if (st->stage) (void)hipFree(st->stage);
  }
  delete st;
}

int lz4hip_gen_blocks_dev(uint8_t* dst, uint64_t stride, int32_t block_len, uint64_t seed, uint64_t first_idx, uint32_t litmax,
                          uint32_t win, uint32_t n_blocks, int device, void* stream) {
  int rc = ensure_init();
  if (rc) return fail(rc, "no HIP device: liblz4hip has no CPU fallback");
  if (!dst || block_len < 0 || litmax == 0 || win == 0) return fail(LZ4HIP_E_ARG, "bad argument");
  int ord;
  if (ordinal(device, &ord)) return fail(LZ4HIP_E_ARG, "bad device index");
  DeviceGuard g(ord);
  int e = lz4hip::launch_gen_blocks(dst, stride, block_len, seed, first_idx, litmax, win, n_blocks, stream);
  return e ? fail(LZ4HIP_E_HIP, "kernel launch", (hipError_t)e) : LZ4HIP_OK;
}

}  // extern "C"

/*
 * lz4hip.h -- C ABI of liblz4hip.so, the MI355X (gfx950) LZ4 block engine that sits behind
 * lz4-java's LZ4Factory as the fourth ("HIP") implementation family.
 *
 * Every entry point replaces one native call of the reference's JNI shim, but processes a BATCH
 * of independent blocks per HIP launch (the reference makes one liblz4 call per block):
 *
 *   lz4hip_compress_fast*     <- Java_net_jpountz_lz4_LZ4JNI_LZ4_1compress_1limitedOutput
 *                                -> LZ4_compress_default   (src/jni/net_jpountz_lz4_LZ4JNI.c:46-86, call :75)
 *   lz4hip_decompress_safe*   <- ..._LZ4_1decompress_1safe -> LZ4_decompress_safe (LZ4JNI.c:187-227, call :216)
 *   lz4hip_decompress_fast*   <- ..._LZ4_1decompress_1fast -> LZ4_decompress_fast (LZ4JNI.c:140-180, call :169)
 *   lz4hip_compress_hc*       <- ..._LZ4_1compressHC       -> LZ4_compress_HC     (LZ4JNI.c:93-133,  call :122)
 *   lz4hip_compress_bound     <- ..._LZ4_1compressBound    -> LZ4_compressBound   (LZ4JNI.c:234-239)
 *   lz4hip_xxh32* / xxh64*    <- Java_net_jpountz_xxhash_XXHashJNI_XXH32 / XXH64
 *                                (src/jni/net_jpountz_xxhash_XXHashJNI.c:42-59 / :152-169, calls :54 / :164)
 *
 * Return conventions deliberately equal liblz4's so the Java error mapping of the JNI family
 * (LZ4JNICompressor.java:39-41, LZ4JNISafeDecompressor.java:39-41, LZ4JNIFastDecompressor.java:40-42)
 * carries over unchanged:
 *   compress   : out_len[i] > 0 compressed size, 0 = dst_cap[i] too small
 *   decompress_safe : out_len[i] >= 0 decoded size, < 0 = -(input position) - 1
 *   decompress_fast : out_len[i] > 0 bytes consumed from src, < 0 = -(input position) - 1
 * Results are bit-exact against liblz4 1.9.3 (what LZ4Factory.nativeInstance() loads on linux/amd64).
 *
 * All functions return LZ4HIP_OK (0) or a negative lz4hip_status describing a LIBRARY failure
 * (no device, HIP error, bad argument); per-block codec results go to the out arrays.  There is
 * NO CPU fallback: without a usable gfx950 device every compute entry point fails with
 * LZ4HIP_E_NO_DEVICE.  All entry points are re-entrant and thread-safe; the library never keeps a
 * caller pointer past the call.
 */
#ifndef LZ4HIP_H
#define LZ4HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library itself is built with -fvisibility=hidden */
#endif

#define LZ4HIP_VERSION 100 /* 0.1.0 */

typedef enum lz4hip_status {
  LZ4HIP_OK = 0,
  LZ4HIP_E_NO_DEVICE = -1,  /* no HIP device / not initialised and auto-init failed */
  LZ4HIP_E_HIP = -2,        /* a HIP runtime call failed (see lz4hip_last_error) */
  LZ4HIP_E_ARG = -3,        /* null pointer, negative length, bad device index, ... */
  LZ4HIP_E_NOMEM = -4,      /* device or pinned-host allocation failed */
  LZ4HIP_E_UNSUPPORTED = -5 /* e.g. HC level outside what this build implements */
} lz4hip_status;

/* ---- lifecycle ----------------------------------------------------------------------------- */
/* device_ids == NULL or n_devices <= 0: use every visible device.  Idempotent, thread-safe.      */
int lz4hip_init(const int* device_ids, int n_devices);
void lz4hip_shutdown(void);
int lz4hip_device_count(void);            /* devices the engine is initialised on (0 if none)   */
const char* lz4hip_last_error(void);      /* thread-local description of the last failure       */
int lz4hip_version(void);
/* tuning knobs (not part of the reference API; process-wide atomics read once per launch; every setting produces the same bytes).
 * Which decoder a launch of n blocks gets with every knob at its default (CU = compute units of the device, 256 on an MI355X):
 *   n <= 5 CU    the trio loop (lz4_decode_trio.h): THREE wavefronts per block -- scanner, planner, copier
 *   n <= 16 CU   the parallel wave loop (lz4_decode_wave.h): a wavefront per block, several sequences of it per trip
 *   more         chosen ON THE DEVICE from a sample of the batch (decode_route_kernel): 16384 .. 40959 blocks averaging >= 512 KiB compressed ->
 *                the ring loop (lz4_decode_ring.h); sequences of "decode_route_short" or fewer output bytes on average with near match
 *                sources (text) -> the wave loop; n <= 32 CU and streams that are not mostly literals -> the wave loop; otherwise the deep
 *                loop (lz4_decode_deep.h) below 40960 blocks and, from there on, for near match sources, else the 4-lane staged loop
 * "decode_pipe" = -1 (the table above) or one loop for every launch: 8 = trio loop, 7 = pair loop (two wavefronts per block: a parser and a
 *   copier), 5 = parallel wave loop, 4 = wave loop with one
 *   sequence per trip, 3 = ring loop, 2 = deep loop, 1 = two-trip pipelined loop, 0 = plain loop;
 * "decode_ring" = bytes of the output ring in LDS: 0 (by batch size) / 8192 / 16384 / 32768 / 65536 with decode_pipe 4 / 5,
 *   and 8, 0 / 16384 / 32768 / 65536 with 7, 0 / 256 / 512 / 1024 / 2048 / 4096 with 3 (which ones: by "decode_lanes");
 * "decode_lanes" = lanes of a wavefront sharing one block in the lane-group loops (0 = by batch size, 4 / 8 / 16 / 32 / 64; 1 with the ring
 *   loop: a lane per block); a combination no kernel exists for makes the next decode call fail with LZ4HIP_E_ARG and a message that names it;
 * "decode_stage" = 1 / 0 / -1 (by batch size): the plain loop writes through an LDS staging buffer so that output reaches memory as
 *   whole 128-byte lines (faster when the batch is bandwidth-bound);
 * "decode_route_short" = 0 .. 255 (default 8): average output bytes per sequence up to which the device-side route sends a batch (of near match
 *   sources) to the wave loop; 0 = never;
 * "compress_core" = 5 (default: adaptive two-pass -- blocks of long sequences are finished by the lean core (lz4_fast_v2_core.h), whose
 *   parked hits a partner wavefront writes out, blocks of short sequences by the window-parallel core), 3 (lean core only) or 1
 *   (window-parallel core only); "compress_switch" = routing threshold of the adaptive scheme in bytes per sequence (default 16);
 * "compress_pack" = 1 (default) / 0: blocks of 65547 bytes .. 4 MiB are compressed with 32-bit table entries on ten match-finder chains
 *   per CU instead of five (0: every block on the five-chain kernel).                                                           */
int lz4hip_set_option(const char* name, int value);
/* diagnostic: what the device-side route of a device's last routed decode launch decided (device = index as in the _dev calls) -- out6
 * (8 words) = { route (0 lane-group default of the batch size, 1 ring loop, 2 wave loop, 3 deep loop instead of the staged one), hops
 * counted by the sampler, stream bytes it walked, average compressed size of 64 blocks, sampled match offsets within 6 KB, sampled
 * sequences, their output bytes, 0 }; synchronises the device                                                                  */
int lz4hip_last_decode_route(int device, uint32_t* out6);

/* == LZ4_compressBound (LZ4JNI.c:237): n + n/255 + 16, 0 if n < 0 or n > 0x7E000000            */
int lz4hip_compress_bound(int n);

/* ---- host-pointer batch API ----------------------------------------------------------------
 * Block i occupies src[src_off[i] .. +src_len[i]) and owns the slot dst[dst_off[i] .. +dst_cap[i]).
 * The library stages H2D/D2H itself and shards contiguous block ranges over the initialised
 * devices; slots must not overlap.                                        */
int lz4hip_compress_fast_batch(const uint8_t* src, const uint64_t* src_off, const int32_t* src_len,
                               uint8_t* dst, const uint64_t* dst_off, const int32_t* dst_cap,
                               int32_t* out_len, uint32_t n_blocks);
int lz4hip_compress_hc_batch(const uint8_t* src, const uint64_t* src_off, const int32_t* src_len,
                             uint8_t* dst, const uint64_t* dst_off, const int32_t* dst_cap,
                             int32_t* out_len, uint32_t n_blocks, int level);
int lz4hip_decompress_safe_batch(const uint8_t* src, const uint64_t* src_off, const int32_t* src_len,
                                 uint8_t* dst, const uint64_t* dst_off, const int32_t* dst_cap,
                                 int32_t* out_len, uint32_t n_blocks);
/* FAST DECODER CONTRACT.  src_cap[i] bounds how far block i may be read (the Java array/buffer length past srcOff):
 * unlike liblz4's LZ4_decompress_fast (which trusts the stream and reads wherever it points) this implementation never reads
 * beyond src_cap[i] bytes of the source slot and never before the destination slot.
 *   - valid stream whose decoded size is dst_len[i]: out_consumed[i] = the compressed length, dst = the decoded bytes --
 *     identical to LZ4_decompress_fast (LZ4JNI.c:169);
 *   - anything else (truncated / corrupted / random bytes, wrong dst_len): liblz4's own decode loop with ONE change -- a read
 *     that would leave the source slot, or an offset that would reach before the destination, is the error at that point:
 *     out_consumed[i] = -(input position) - 1.  Streams liblz4 happens to accept while reading garbage past the slot are
 *     therefore rejected here; streams liblz4 rejects are rejected with the same code when the rejection does not depend on
 *     bytes outside the slot.
 * The behaviour is pinned by 600 vectors in tests/golden/fast_decode_contract.json (generated from oracle/lz4_oracle.c
 * lz4o_decompress_fast_bounded; its valid cases are cross-checked against the reference library) and tested on the CPU
 * (tests/test_oracle.py) and on the GPU (tests/test_gpu_scale.py).                                                        */
int lz4hip_decompress_fast_batch(const uint8_t* src, const uint64_t* src_off, const int32_t* src_cap,
                                 uint8_t* dst, const uint64_t* dst_off, const int32_t* dst_len,
                                 int32_t* out_consumed, uint32_t n_blocks);
int lz4hip_xxh32_batch(const uint8_t* buf, const uint64_t* off, const int32_t* len, uint32_t seed,
                       uint32_t* out, uint32_t n);
int lz4hip_xxh64_batch(const uint8_t* buf, const uint64_t* off, const int32_t* len, uint64_t seed,
                       uint64_t* out, uint32_t n);

/* ---- device-pointer batch API ----------------------------------------------------------------
 * Same contracts, every pointer is a DEVICE pointer on `device` (an index into the initialised
 * device list), work is enqueued on `stream` (a hipStream_t, NULL = the null stream) and the call
 * returns without synchronising: no PCIe traffic, this is what bench.py times.                  */
int lz4hip_compress_fast_batch_dev(const uint8_t* src, const uint64_t* src_off, const int32_t* src_len,
                                   uint8_t* dst, const uint64_t* dst_off, const int32_t* dst_cap,
                                   int32_t* out_len, uint32_t n_blocks, int device, void* stream);
/* HC: levels follow liblz4 (< 1 -> 9, > 12 -> 12): 1..9 = hash-chain strategy with lazy evaluation, 10..12 = optimal
 * parser (lz4-java levels 10..17).  Levels 10..12 are FUNCTIONAL ONLY: byte-identical output, but the optimal parser's table
 * walk is wave-uniform scalar work (about 1.0 / 0.7 GB/s per GPU at levels 10 / 12 -- no faster than the reference on the host's
 * cores); levels 1..9 are the ones to use for throughput (level 9: ~12 GB/s per GPU on 1 MiB blocks).  The call sizes an internal
 * u16 workspace (2 bytes per source byte, stream-ordered allocation) and therefore synchronises `stream`
 * once before it enqueues the two kernels.                                                          */
int lz4hip_compress_hc_batch_dev(const uint8_t* src, const uint64_t* src_off, const int32_t* src_len,
                                 uint8_t* dst, const uint64_t* dst_off, const int32_t* dst_cap,
                                 int32_t* out_len, uint32_t n_blocks, int level, int device, void* stream);
/* lz4hip_compress_hc_batch_dev sizes its chain workspace from the batch's source span, which it has to read back from the device:
 * it synchronises `stream` ONCE per call (every other *_dev entry never synchronises).  The _ws form is fully asynchronous: the
 * caller states src_span = max over blocks of src_off + src_len (any upper bound, e.g. the size of the source buffer) and passes a
 * device workspace of at least lz4hip_hc_workspace_bytes(src_span, n_blocks, level) bytes that stays valid until the work is done.
 * A block whose range reaches past src_span is not compressed: its out_len is 0 and nothing is written past the workspace.        */
size_t lz4hip_hc_workspace_bytes(uint64_t src_span, uint32_t n_blocks, int level);
int lz4hip_compress_hc_batch_dev_ws(const uint8_t* src, const uint64_t* src_off, const int32_t* src_len,
                                    uint8_t* dst, const uint64_t* dst_off, const int32_t* dst_cap,
                                    int32_t* out_len, uint32_t n_blocks, int level, int device, void* stream,
                                    uint64_t src_span, void* ws, size_t ws_bytes);
int lz4hip_decompress_safe_batch_dev(const uint8_t* src, const uint64_t* src_off, const int32_t* src_len,
                                     uint8_t* dst, const uint64_t* dst_off, const int32_t* dst_cap,
                                     int32_t* out_len, uint32_t n_blocks, int device, void* stream);
int lz4hip_decompress_fast_batch_dev(const uint8_t* src, const uint64_t* src_off, const int32_t* src_cap,
                                     uint8_t* dst, const uint64_t* dst_off, const int32_t* dst_len,
                                     int32_t* out_consumed, uint32_t n_blocks, int device, void* stream);
int lz4hip_xxh32_batch_dev(const uint8_t* buf, const uint64_t* off, const int32_t* len, uint32_t seed,
                           uint32_t* out, uint32_t n, int device, void* stream);
int lz4hip_xxh64_batch_dev(const uint8_t* buf, const uint64_t* off, const int32_t* len, uint64_t seed,
                           uint64_t* out, uint32_t n, int device, void* stream);

/* ---- single-block convenience (== batch of 1; what the Java single-call path and the
 * LZ4Factory constructor self-test, LZ4Factory.java:204-220, go through).  The return value is the
 * liblz4 return value of the corresponding function.  A LIBRARY failure (no device, HIP error) is
 * reported as INT32_MIN + (-status), which no codec result can equal: test LZ4HIP_IS_LIB_ERROR().  */
#define LZ4HIP_LIB_ERROR(status) ((int)(INT32_MIN + (-(status))))
#define LZ4HIP_IS_LIB_ERROR(ret) ((ret) < (int)(INT32_MIN + 64))
int lz4hip_compress_fast(const uint8_t* src, int src_len, uint8_t* dst, int dst_cap);
int lz4hip_compress_hc(const uint8_t* src, int src_len, uint8_t* dst, int dst_cap, int level);
int lz4hip_decompress_safe(const uint8_t* src, int src_len, uint8_t* dst, int dst_cap);
int lz4hip_decompress_fast(const uint8_t* src, int src_cap, uint8_t* dst, int dst_len);
int lz4hip_xxh32(const uint8_t* buf, int len, uint32_t seed, uint32_t* out);
int lz4hip_xxh64(const uint8_t* buf, int len, uint64_t seed, uint64_t* out);

/* ---- streaming xxhash ---------------------------------------------------------------------------
 * What StreamingXXHash32JNI / StreamingXXHash64JNI bind (XXHashJNI.java: XXH32_init / _update / _digest / _free,
 * net_jpountz_xxhash_XXHashJNI.c:89-145; the XXH64 set :199-255).  A stream's state -- accumulators, the bytes of
 * the incomplete stripe, the length so far and the digest of everything so far -- is a small record in DEVICE
 * memory; every update is one kernel launch that continues from it (one wavefront: XXH is a serial chain per
 * stream), so device-resident data (decoded blocks, say) is hashed where it lies.  digest == the one-shot hash
 * of the concatenation of all updates since create / reset.  Calls on one handle are serialised (the
 * reference's methods are `synchronized`).  Handles must be freed before lz4hip_shutdown().                   */
typedef struct lz4hip_xxh_stream lz4hip_xxh_stream;
int lz4hip_xxh32_stream_create(uint32_t seed, lz4hip_xxh_stream** out);  /* XXH32_init, XXHashJNI.c:90-101 */
int lz4hip_xxh64_stream_create(uint64_t seed, lz4hip_xxh_stream** out);  /* XXH64_init, XXHashJNI.c:200-211 */
int lz4hip_xxh_stream_reset(lz4hip_xxh_stream* st, uint64_t seed);       /* StreamingXXHash32JNI.reset()    */
int lz4hip_xxh_stream_update(lz4hip_xxh_stream* st, const uint8_t* buf, int len);   /* host pointer; XXHashJNI.c:108-121 */
/* device pointer (on the engine's first device), enqueued on `stream`, returns without synchronising */
int lz4hip_xxh_stream_update_dev(lz4hip_xxh_stream* st, const uint8_t* dbuf, int len, void* stream);
int lz4hip_xxh32_stream_digest(lz4hip_xxh_stream* st, uint32_t* out);    /* XXH32_digest, XXHashJNI.c:128-133 */
int lz4hip_xxh64_stream_digest(lz4hip_xxh_stream* st, uint64_t* out);    /* XXH64_digest, XXHashJNI.c:238-243 */
void lz4hip_xxh_stream_free(lz4hip_xxh_stream* st);                      /* XXH32_free / XXH64_free          */

/* ---- container blocks assembled on the device ---------------------------
 * The data blocks of an LZ4 Frame (kind 0: what LZ4FrameOutputStream.writeBlock emits per block,
 * /root/reference/src/java/net/jpountz/lz4/LZ4FrameOutputStream.java:199-235 -- 4-byte size word with the "stored uncompressed"
 * bit, payload, and with flags & 1 the XXH32 (seed 0) of the stored payload) or of lz4-java's LZ4Block container (kind 1:
 * LZ4BlockOutputStream.flushBufferedData, LZ4BlockOutputStream.java:203-227 -- 21-byte header {"LZ4Block", method | level,
 * compressed length, original length, XXH32(original, seed 0x9747b28c) & 0x0FFFFFFF}, payload) for src[0, n_bytes) cut into
 * block_size pieces.  Compression, raw-fallback decision, size scan, headers, payload compaction and checksums all run on the
 * device behind one another; frame header, end mark and content checksum stay with the caller (a few bytes, independent of the
 * data).  level 0 = LZ4_compress_default, 1..17 = lz4-java HC levels.  Bytes are identical to the reference writers' for the
 * same input, block size and flags (tests/test_gpu_streams.py: the `lz4` CLI reads and reproduces them).
 *   lz4hip_container_blocks      host pointers (H2D of the data, D2H of the finished blocks only); *out_bytes = bytes written
 *   lz4hip_container_blocks_dev  device pointers, asynchronous on `stream`; *total_dev (device) = bytes written -- if it exceeds
 *                                dst_cap nothing beyond dst_cap was written and the result is unusable; ws = device scratch of
 *                                lz4hip_container_workspace_bytes(n_bytes, block_size, level) bytes                              */
size_t lz4hip_container_workspace_bytes(uint64_t n_bytes, uint32_t block_size, int level);
int lz4hip_container_blocks(int kind, int flags, int level, const uint8_t* src, uint64_t n_bytes, uint32_t block_size,
                            uint8_t* dst, uint64_t dst_cap, uint64_t* out_bytes);
int lz4hip_container_blocks_dev(int kind, int flags, int level, const uint8_t* src, uint64_t n_bytes, uint32_t block_size,
                                uint8_t* dst, uint64_t dst_cap, uint64_t* total_dev, void* ws, size_t ws_bytes, int device, void* stream);

/* Device-side container READ path (round 4; replaces the per-block loops of LZ4FrameInputStream.readBlock,
 * src/java/net/jpountz/lz4/LZ4FrameInputStream.java:258-322, and LZ4BlockInputStream.refill, LZ4BlockInputStream.java:191-264, for a
 * whole run of blocks that lies in DEVICE memory): the size words (kind 0: LZ4 Frame body, starting at the first block's size word;
 * flags & 1 = a block checksum follows every payload; max_block = the frame's block maximum size) or the 21-byte headers (kind 1:
 * lz4-java's "LZ4Block" stream) of body[0, body_bytes) are walked on the device, frame block checksums verified where the payloads
 * lie, compressed blocks decoded (LZ4_decompress_safe into max_block bytes / LZ4_decompress_fast into the header's original length,
 * whose return value must be the header's compressed length), raw blocks copied, LZ4Block checksums verified on the decoded bytes --
 * all asynchronous on `stream`.  Block k decodes to dst + k * slot_bytes (slot_bytes >= max_block for frames; n_max slots);
 * sizes_dev[k] = its decoded size; info_dev[0..4] = { blocks delivered, bytes of body consumed by them (+ the end mark / empty
 * block when it was reached), stop reason, decoded bytes of the delivered blocks, liblz4's code when a decode failed }.
 * Stop reasons follow the readers' own order of checks per block; the first block that fails ends the run:
 *   0 end mark (frame) / empty block (LZ4Block) reached     1 body ended at a block boundary     7 all n_max slots used, more follows
 *   2 body ended inside a block ("Stream ended prematurely")   3 frame: "Block size %s exceeded max"  (or a slot too small;
 *     LZ4Block: an original length above slot_bytes or -- when max_block is not 0 -- above max_block: not a stream error)
 *   4 frame: block checksum mismatch     5 frame: the block does not decode (LZ4Exception)     6 LZ4Block: "Stream is corrupted"
 * ws = device scratch of lz4hip_container_decode_workspace_bytes(n_max) bytes.                                               */
size_t lz4hip_container_decode_workspace_bytes(uint32_t n_max);
int lz4hip_container_decode_dev(int kind, int flags, const uint8_t* body, uint64_t body_bytes, uint32_t max_block, uint8_t* dst,
                                uint64_t slot_bytes, uint32_t n_max, int32_t* sizes_dev, uint64_t* info_dev, void* ws, size_t ws_bytes,
                                int device, void* stream);
/* the same with host pointers: H2D of the container bytes, the device path, D2H of the delivered blocks BACK TO BACK into dst
 * (max_block: frame block maximum / an upper bound of the LZ4Block original lengths, e.g. 1 << (10 + level nibble)); sizes[n_max],
 * info[5] as above */
int lz4hip_container_decode(int kind, int flags, const uint8_t* body, uint64_t body_bytes, uint32_t max_block, uint32_t n_max,
                            uint8_t* dst, uint64_t dst_cap, int32_t* sizes, uint64_t* info);
/* what a caller of lz4hip_container_decode must provide: a host-side walk of the size words / headers (no device work): *n_blocks =
 * whole blocks of the first n_max that body holds, *dst_bytes = the most their decoded forms can need (frame: max_block per
 * compressed block, the stored size of a raw one; LZ4Block: the headers' original lengths) -- so that a 100-byte frame asks for one
 * block's worth of destination, not n_max x max_block (the readers of LZ4FrameInputStream.java:258-322 / LZ4BlockInputStream.java:
 * 191-264 allocate one block at a time).  The walk stops where the device walk stops, and nothing it stops at sizes anything: an
 * LZ4Block header whose original length exceeds max_block (stop reason 3: the readers take their host path), a payload cut short
 * (2), a header that announces more than 255 x compressedLen + 64 bytes -- more than LZ4 can decode from that many (6, "Stream is
 * corrupted", said by the walk itself).  lz4hip_container_decode sizes its own device slots the same way: a few KB of hostile
 * headers cannot make it allocate gigabytes.                                                                                   */
int lz4hip_container_decode_bound(int kind, int flags, const uint8_t* body, uint64_t body_bytes, uint32_t max_block, uint32_t n_max,
                                  uint32_t* n_blocks, uint64_t* dst_bytes);

/* ---- workload helper (not part of the reference API) ------------------------------------------
 * Fills n_blocks slots of `block_len` bytes at dst + i*stride with the SURVEY.md App. F synthetic
 * blocks idx = first_idx + i (deterministic, seed-addressed).  Device pointer, async on stream.
 * bench.py uses it so BASELINE.json's 4 GiB batch never crosses PCIe.                           */
int lz4hip_gen_blocks_dev(uint8_t* dst, uint64_t stride, int32_t block_len, uint64_t seed, uint64_t first_idx,
                          uint32_t litmax, uint32_t win, uint32_t n_blocks, int device, void* stream);

#ifdef LZ4HIP_DEV_TOOLS
/* ---- developer diagnostics (not part of the reference API) --------------------------------------
 * lz4hip_compress_fast_batch_dev with per-phase shader-clock accounting: prof[b*12 .. b*12+11] =
 * {steps, collision steps, fingerprint false positives, sequences, cycles of phases 0..7} of block b
 * (phases are documented in csrc/lz4_fast_core.h).  Output bytes are identical to the product kernel.
 * Only in developer builds of the library (tools/build_variant.sh dev -DLZ4HIP_DEV_TOOLS); the release library does not export it. */
int lz4hip_dbg_compress_fast_profile_dev(const uint8_t* src, const uint64_t* src_off, const int32_t* src_len,
                                         uint8_t* dst, const uint64_t* dst_off, const int32_t* dst_cap,
                                         int32_t* out_len, uint32_t n_blocks, uint64_t* prof, int device, void* stream);
#endif

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* LZ4HIP_H */

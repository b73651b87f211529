// kernels.h -- host-callable launchers of the gfx950 kernels (kernels.hip).  All pointers are device
// pointers; launches are asynchronous on `stream`.  Return value: hipError_t as int.
#pragma once
#include <stdint.h>
#include <stddef.h>

namespace lz4hip {

struct BatchArgs {
  const uint8_t* src; const uint64_t* src_off; const int32_t* src_len;
  uint8_t* dst; const uint64_t* dst_off; const int32_t* dst_cap;
  int32_t* out; uint32_t n;
};

// Fast compress, two kernels (same bytes):
//   launch_compress_fast_v2w lean finder loop (lz4_fast_v2_core.h; common step hand-scheduled, lz4_fast_v2_asm.h) with a writer
//                            wavefront per finder: bare hits parked in lanes, handed over 64 at a time: the default
//   launch_compress_fast_ms  every sequence of a 64-position window per step (lz4_fast_ms_core.h): best when sequences are short
// Both fill each CU with one workgroup of 5 finder wavefronts (5 x 32 KB tables = the CU's whole LDS) that draw blocks from a queue.
// q = three device uint32_t (queue words), n_cus = compute units of the device.
// Adaptive two-pass use: launch_compress_fast_v2w(q, routed = u32[n] device scratch, dense64) finishes the blocks with long
// sequences and lists the others in routed[] (sequences 32..95 cover fewer than dense64 bytes); launch_compress_fast_ms(q, routed,
// first = false) then does exactly those.  routed == nullptr: the kernel does every block (first = true zeroes q).
// `mail`: compress_fast_v2w_scratch_words(n_cus) words of device scratch (the finder/writer rings)
size_t compress_fast_v2w_scratch_words(uint32_t n_cus);
int launch_compress_fast_v2w(const BatchArgs& a, uint32_t* q, uint32_t* routed, uint32_t dense64, uint32_t n_cus, uint32_t* mail, void* stream);
// 1 (default): blocks of 65547 bytes .. 4 MiB run on ten pairs per CU with compact table entries (compress_fast_v2wp_cu_kernel); 0: five pairs for all
void set_compress_pack(int v);
int launch_compress_fast_ms(const BatchArgs& a, uint32_t* q, const uint32_t* routed, bool first, uint32_t n_cus, void* stream);
#ifdef LZ4HIP_DEV_TOOLS
int launch_compress_fast_prof(const BatchArgs& a, uint64_t* prof, int core, void* stream);  // developer build only (tools/build_variant.sh dev -DLZ4HIP_DEV_TOOLS)
#endif
// HC levels 1..12: `ws` = device workspace of hc_ws_bytes(span, n, level) bytes, span = max(src_off+src_len) (launch_hc_span)
int launch_hc_span(const uint64_t* src_off, const int32_t* src_len, uint32_t n, uint64_t* out_dev, void* stream);
size_t hc_ws_bytes(uint64_t span, uint32_t n_blocks, int level);
int launch_compress_hc(const BatchArgs& a, int level, void* ws, uint64_t span, void* stream);
// after a compress launch: moves the out[i] > 0 useful bytes of every slot to pack + sum(out[0..i)); poff: u64[n] scratch
int launch_pack(const BatchArgs& a, uint64_t* poff, uint8_t* pack, void* stream);
// Device-side container assembly (kernels.hip): the data blocks of an LZ4 Frame (kind 0; block_checksum: XXH32 of each stored
// payload) or of lz4-java's LZ4Block container (kind 1) for src[0, n_bytes) cut into block_size pieces -> dst, *total = bytes
// written (if it exceeds dst_cap nothing past dst_cap was written and the caller must not use the result).  ws: device scratch of
// container_ws_bytes(); hc_level > 0: LZ4 HC at that level (hc_ws = hc_ws_bytes(n_bytes, n, level)), else the fast compressor
// (q_scratch = 3 + n + compress_fast_v2w_scratch_words(n_cus) words; core as "compress_core").
size_t container_ws_bytes(uint64_t n_bytes, uint32_t block_size);
int launch_container_blocks(int kind, int block_checksum, int hc_level, const uint8_t* src, uint64_t n_bytes, uint32_t block_size, uint8_t* dst, uint64_t dst_cap,
                            unsigned long long* total, void* ws, void* hc_ws, uint32_t* q_scratch, uint32_t dense64, uint32_t n_cus, int core, void* stream);
// Device-side container READ path (kernels.hip): the data blocks of an LZ4 Frame body (kind 0; block_checksum: a XXH32 word behind
// each payload) or of an LZ4Block stream (kind 1) in body[0, body_bytes) are walked, verified and decoded; block k decodes to
// dst + k * slot_bytes, sizes[k] = its decoded size, info[0..4] = {blocks delivered, body bytes consumed, stop reason, decoded bytes,
// liblz4 code of a failed decode}; ws = container_read_ws_bytes(n_max) bytes of device scratch
size_t container_read_ws_bytes(uint32_t n_max);
int launch_container_read(int kind, int block_checksum, const uint8_t* body, uint64_t body_bytes, uint32_t max_block, uint8_t* dst, uint64_t slot_bytes,
                          uint32_t n_max, int32_t* sizes, unsigned long long* info, void* ws, void* stream);
// lanes_per_block: lanes of a wavefront that share one block in the decoder (4..64); 0 = default
// pipe: 1 = pipelined interior loop (lz4_decode_core.h PIPE), 0 = plain, -1 = default for the batch size
// stage: 1 = the plain interior loop writes through LDS staging (whole-line output), 0 / -1 = off
// pipe 3: the ring loop (lz4_decode_ring.h); ring = bytes of its output ring (512 / 1024 / 2048 / 4096; 0 = default for the lanes)
// route_word: one device uint32_t of scratch (or nullptr): with every knob at its default, batches of more than 16 blocks per CU are routed
// on the device (decode_route_kernel): by the blocks' compressed sizes between the deep loop and the ring loop (12288 .. 40959 blocks), by
// the streams' sequence density between the lane-group loops and the wave kernel (set_route_short: average output bytes per sequence up to which a batch is text-like, 0 = never) and between the staged and the deep loop (40960 blocks and more: near sources)
void set_route_short(int bytes_per_sequence);
int last_decode_route(uint32_t* out6);   // diagnostic (8 words): {route, sampled hops, sampled stream bytes, average compressed size, offsets within 6 KB, sequences looked at, their output bytes, 0} of the current device's last routed launch (synchronises)
int launch_decompress(const BatchArgs& a, bool safe, int lanes_per_block, int pipe, int stage, int ring, void* stream, uint32_t* route_word = nullptr);
int launch_xxh32(const uint8_t* buf, const uint64_t* off, const int32_t* len, uint32_t seed, uint32_t* out, uint32_t n, void* stream);
int launch_xxh64(const uint8_t* buf, const uint64_t* off, const int32_t* len, uint64_t seed, uint64_t* out, uint32_t n, void* stream);
// streaming xxhash: `rec` = device record of xxh_stream_rec_bytes() bytes (the digest so far sits at xxh_stream_digest_offset());
// reset != 0 restarts the stream with `seed` before absorbing data[0..len)
int launch_xxh32_stream(void* rec, const uint8_t* data, uint32_t len, int reset, uint32_t seed, void* stream);
int launch_xxh64_stream(void* rec, const uint8_t* data, uint32_t len, int reset, uint64_t seed, void* stream);
size_t xxh_stream_rec_bytes(bool is64);
size_t xxh_stream_digest_offset(bool is64);
int launch_gen_blocks(uint8_t* dst, uint64_t stride, int32_t block_len, uint64_t seed, uint64_t first_idx,
                      uint32_t litmax, uint32_t win, uint32_t n_blocks, void* stream);

}  // namespace lz4hip

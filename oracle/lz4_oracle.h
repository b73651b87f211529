/*
 * oracle/lz4_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the algorithms lz4-java's nativeInstance() reaches through
 * its JNI shim.  The arithmetic itself lives in the third-party module github.com/lz4/lz4,
 * pinned at v1.9.3 by the un-vendored submodule `src/lz4` of the reference
 * (/root/reference/.gitmodules:1-3, CHANGES.md:5); the reference call sites are
 *   src/jni/net_jpountz_lz4_LZ4JNI.c:75   LZ4_compress_default  -> lz4o_compress_fast
 *   src/jni/net_jpountz_lz4_LZ4JNI.c:122  LZ4_compress_HC       -> lz4o_compress_hc
 *   src/jni/net_jpountz_lz4_LZ4JNI.c:169  LZ4_decompress_fast   -> lz4o_decompress_fast
 *   src/jni/net_jpountz_lz4_LZ4JNI.c:216  LZ4_decompress_safe   -> lz4o_decompress_safe
 *   src/jni/net_jpountz_lz4_LZ4JNI.c:237  LZ4_compressBound     -> lz4o_compress_bound
 *   src/jni/net_jpountz_xxhash_XXHashJNI.c:54,164  XXH32 / XXH64 -> lz4o_xxh32 / lz4o_xxh64
 *
 * Parity is PINNED: tests/test_oracle_*.py check every function here byte-for-byte against
 * (1) the golden table of SURVEY.md App. E (tests/golden/), and (2) the reference's own
 * prebuilt JNI library (oracle/_ref/liblz4-java.so, see oracle/Makefile) on Calgary, synthetic
 * and fuzzed inputs.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code.
 * The product (lz4-java_amd/, liblz4hip.so) never links, loads or calls it.
 */
#ifndef LZ4_ORACLE_H
#define LZ4_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int lz4o_compress_bound(int n);
int lz4o_compress_fast(const uint8_t* src, int n, uint8_t* dst, int cap);
int lz4o_compress_hc(const uint8_t* src, int n, uint8_t* dst, int cap, int level);
int lz4o_decompress_safe(const uint8_t* src, int src_len, uint8_t* dst, int cap);
int lz4o_decompress_fast(const uint8_t* src, uint8_t* dst, int dst_len);
/* memory-safe variant of decompress_fast: never reads at or beyond src+src_cap */
int lz4o_decompress_fast_bounded(const uint8_t* src, int src_cap, uint8_t* dst, int dst_len);
uint32_t lz4o_xxh32(const uint8_t* p, int64_t len, uint32_t seed);
uint64_t lz4o_xxh64(const uint8_t* p, int64_t len, uint64_t seed);
/* SURVEY.md App. F synthetic block generator (defined by the survey, not by the reference) */
void lz4o_gen_block(uint8_t* out, int64_t n, uint64_t seed, uint64_t idx, uint32_t litmax, uint32_t win);

#ifdef __cplusplus
}
#endif
#endif

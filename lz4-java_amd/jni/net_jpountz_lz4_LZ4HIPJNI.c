/*
 * JNI shim between net.jpountz.lz4.LZ4HIPJNI / net.jpountz.xxhash.XXHashHIPJNI and liblz4hip
 * (include/lz4hip.h).  Counterpart of the reference's src/jni/net_jpountz_lz4_LZ4JNI.c and
 * src/jni/net_jpountz_xxhash_XXHashJNI.c, with two deliberate differences:
 *   * a Java heap array is pinned with GetPrimitiveArrayCritical only long enough to memcpy the
 *     block into / out of a native staging buffer -- the GC lock is never held across a GPU launch
 *     (the reference holds it across the liblz4 call, LZ4JNI.c:54-82);
 *   * `in` is released when `out` cannot be pinned (the reference leaks it, LZ4JNI.c:59-73).
 * Build (needs a JDK, which the build image lacks -- see INTEGRATION.md):
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
 *       lz4-java_amd/jni/net_jpountz_lz4_LZ4HIPJNI.c -Llz4-java_amd -llz4hip -o liblz4hip-java.so
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "lz4hip.h"

static jclass OutOfMemoryError;
static jclass RuntimeException;

JNIEXPORT void JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_init(JNIEnv* env, jclass cls) {
  (void)cls;
  jclass local = (*env)->FindClass(env, "java/lang/OutOfMemoryError");
  OutOfMemoryError = (jclass)(*env)->NewGlobalRef(env, local);
  local = (*env)->FindClass(env, "java/lang/RuntimeException");
  RuntimeException = (jclass)(*env)->NewGlobalRef(env, local);
  (void)lz4hip_init(NULL, 0); /* a failure surfaces on the first codec call as a library error */
}

static void throw_OOM(JNIEnv* env) { (*env)->ThrowNew(env, OutOfMemoryError, "Out of memory"); }
/* a liblz4hip failure (no device, HIP error, bad handle) in an entry point whose Java signature has no error channel: the
 * reference's XXHashJNI cannot fail there; returning hash 0 or a stale digest silently would let a checksum comparison pass or
 * fail wrongly, so the caller gets an unchecked exception carrying lz4hip_last_error() */
static void throw_lib(JNIEnv* env) { (*env)->ThrowNew(env, RuntimeException, lz4hip_last_error()); }

/* A (array | direct buffer, offset, length) argument made addressable for the native call. */
typedef struct {
  uint8_t* p;     /* address of byte 0 of the region */
  uint8_t* heap;  /* staging copy to free, or NULL for a direct buffer */
} region_t;

static int region_in(JNIEnv* env, jbyteArray arr, jobject buf, jint off, jint len, int copy_in, region_t* r) {
  r->heap = NULL;
  if (arr == NULL) {
    uint8_t* base = (uint8_t*)(*env)->GetDirectBufferAddress(env, buf);
    if (base == NULL) return -1;
    r->p = base + off;
    return 0;
  }
  r->heap = (uint8_t*)malloc(len > 0 ? (size_t)len : 1);
  if (r->heap == NULL) return -1;
  r->p = r->heap;
  if (copy_in && len > 0) {
    uint8_t* a = (uint8_t*)(*env)->GetPrimitiveArrayCritical(env, arr, 0);
    if (a == NULL) { free(r->heap); r->heap = NULL; return -1; }
    memcpy(r->heap, a + off, (size_t)len);
    (*env)->ReleasePrimitiveArrayCritical(env, arr, a, JNI_ABORT);
  }
  return 0;
}

/* copies `n` produced bytes back into the Java array (if staged) and frees the staging copy */
static int region_out(JNIEnv* env, jbyteArray arr, jint off, jint n, region_t* r) {
  int rc = 0;
  if (r->heap != NULL) {
    if (arr != NULL && n > 0) {
      uint8_t* a = (uint8_t*)(*env)->GetPrimitiveArrayCritical(env, arr, 0);
      if (a == NULL) rc = -1;
      else { memcpy(a + off, r->heap, (size_t)n); (*env)->ReleasePrimitiveArrayCritical(env, arr, a, 0); }
    }
    free(r->heap);
    r->heap = NULL;
  }
  return rc;
}

typedef int (*codec_fn)(const uint8_t*, int, uint8_t*, int);

static jint run_single(JNIEnv* env, int op, int level, jbyteArray srcArray, jobject srcBuffer, jint srcOff, jint srcLen,
                       jbyteArray destArray, jobject destBuffer, jint destOff, jint destLen) {
  region_t in, out;
  if (region_in(env, srcArray, srcBuffer, srcOff, srcLen, 1, &in) != 0) { throw_OOM(env); return 0; }
  if (region_in(env, destArray, destBuffer, destOff, destLen, 0, &out) != 0) {
    region_out(env, NULL, 0, 0, &in); /* release `in` too */
    throw_OOM(env);
    return 0;
  }
  int result;
  switch (op) {
    case 0: result = lz4hip_compress_fast(in.p, srcLen, out.p, destLen); break;
    case 1: result = lz4hip_decompress_safe(in.p, srcLen, out.p, destLen); break;
    case 2: result = lz4hip_decompress_fast(in.p, srcLen /* readable capacity */, out.p, destLen); break;
    default: result = lz4hip_compress_hc(in.p, srcLen, out.p, destLen, level); break;
  }
  region_out(env, NULL, 0, 0, &in);
  jint produced = 0;
  if (!LZ4HIP_IS_LIB_ERROR(result)) {
    if (op == 2) produced = result > 0 ? destLen : 0;      /* fast decompress fills destLen bytes */
    else produced = result > 0 ? result : 0;
  }
  if (region_out(env, destArray, destOff, produced, &out) != 0) { throw_OOM(env); return 0; }
  return result;
}

JNIEXPORT jint JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1compress_1fast(JNIEnv* env, jclass cls, jbyteArray srcArray, jobject srcBuffer,
    jint srcOff, jint srcLen, jbyteArray destArray, jobject destBuffer, jint destOff, jint maxDestLen) {
  (void)cls;
  return run_single(env, 0, 0, srcArray, srcBuffer, srcOff, srcLen, destArray, destBuffer, destOff, maxDestLen);
}

JNIEXPORT jint JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1compressHC(JNIEnv* env, jclass cls, jbyteArray srcArray, jobject srcBuffer,
    jint srcOff, jint srcLen, jbyteArray destArray, jobject destBuffer, jint destOff, jint maxDestLen, jint level) {
  (void)cls;
  return run_single(env, 3, level, srcArray, srcBuffer, srcOff, srcLen, destArray, destBuffer, destOff, maxDestLen);
}

JNIEXPORT jint JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1decompress_1safe(JNIEnv* env, jclass cls, jbyteArray srcArray, jobject srcBuffer,
    jint srcOff, jint srcLen, jbyteArray destArray, jobject destBuffer, jint destOff, jint maxDestLen) {
  (void)cls;
  return run_single(env, 1, 0, srcArray, srcBuffer, srcOff, srcLen, destArray, destBuffer, destOff, maxDestLen);
}

JNIEXPORT jint JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1decompress_1fast(JNIEnv* env, jclass cls, jbyteArray srcArray, jobject srcBuffer,
    jint srcOff, jint srcCap, jbyteArray destArray, jobject destBuffer, jint destOff, jint destLen) {
  (void)cls;
  return run_single(env, 2, 0, srcArray, srcBuffer, srcOff, srcCap, destArray, destBuffer, destOff, destLen);
}

JNIEXPORT jint JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1compressBound(JNIEnv* env, jclass cls, jint len) {
  (void)env; (void)cls;
  return lz4hip_compress_bound(len);
}

JNIEXPORT jstring JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_lastError(JNIEnv* env, jclass cls) {
  (void)cls;
  return (*env)->NewStringUTF(env, lz4hip_last_error());
}

/* many blocks, direct buffers: nothing is copied on the host side, liblz4hip stages H2D/D2H itself */
JNIEXPORT jint JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1batch(JNIEnv* env, jclass cls, jint op, jint level, jobject src, jlongArray srcOff,
    jintArray srcLen, jobject dest, jlongArray destOff, jintArray destCap, jintArray outLen, jint n) {
  (void)cls;
  const uint8_t* s = (const uint8_t*)(*env)->GetDirectBufferAddress(env, src);
  uint8_t* d = (uint8_t*)(*env)->GetDirectBufferAddress(env, dest);
  if (s == NULL || d == NULL) return LZ4HIP_E_ARG;
  jlong* so = (*env)->GetLongArrayElements(env, srcOff, NULL);
  jint* sl = (*env)->GetIntArrayElements(env, srcLen, NULL);
  jlong* dof = (*env)->GetLongArrayElements(env, destOff, NULL);
  jint* dc = (*env)->GetIntArrayElements(env, destCap, NULL);
  jint* ol = (*env)->GetIntArrayElements(env, outLen, NULL);
  jint rc = LZ4HIP_E_NOMEM;
  if (so && sl && dof && dc && ol) {
    switch (op) {
      case 0: rc = lz4hip_compress_fast_batch(s, (const uint64_t*)so, (const int32_t*)sl, d, (const uint64_t*)dof, (const int32_t*)dc, (int32_t*)ol, (uint32_t)n); break;
      case 1: rc = lz4hip_decompress_safe_batch(s, (const uint64_t*)so, (const int32_t*)sl, d, (const uint64_t*)dof, (const int32_t*)dc, (int32_t*)ol, (uint32_t)n); break;
      case 2: rc = lz4hip_decompress_fast_batch(s, (const uint64_t*)so, (const int32_t*)sl, d, (const uint64_t*)dof, (const int32_t*)dc, (int32_t*)ol, (uint32_t)n); break;
      default: rc = lz4hip_compress_hc_batch(s, (const uint64_t*)so, (const int32_t*)sl, d, (const uint64_t*)dof, (const int32_t*)dc, (int32_t*)ol, (uint32_t)n, level); break;
    }
  }
  if (so) (*env)->ReleaseLongArrayElements(env, srcOff, so, JNI_ABORT);
  if (sl) (*env)->ReleaseIntArrayElements(env, srcLen, sl, JNI_ABORT);
  if (dof) (*env)->ReleaseLongArrayElements(env, destOff, dof, JNI_ABORT);
  if (dc) (*env)->ReleaseIntArrayElements(env, destCap, dc, JNI_ABORT);
  if (ol) (*env)->ReleaseIntArrayElements(env, outLen, ol, 0);
  return rc;
}

/* the data blocks of an LZ4 Frame (kind 0) / of lz4-java's LZ4Block container (kind 1) for src[srcOff, srcOff + len) cut into
 * blockSize pieces, assembled on the device (include/lz4hip.h lz4hip_container_blocks): what LZ4FrameOutputStream.writeBlock /
 * LZ4BlockOutputStream.flushBufferedData emit block by block, in one call.  Direct buffers; returns the bytes written at
 * dest[destOff ..), or the (negative) lz4hip_status */
JNIEXPORT jlong JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1containerBlocks(JNIEnv* env, jclass cls, jint kind, jint flags, jint level, jobject src,
    jlong srcOff, jlong len, jint blockSize, jobject dest, jlong destOff, jlong destCap) {
  (void)cls;
  const uint8_t* s = (const uint8_t*)(*env)->GetDirectBufferAddress(env, src);
  uint8_t* d = (uint8_t*)(*env)->GetDirectBufferAddress(env, dest);
  if (s == NULL || d == NULL || srcOff < 0 || len < 0 || destOff < 0 || destCap < 0 || blockSize <= 0) return (jlong)LZ4HIP_E_ARG;
  uint64_t out = 0;
  const int rc = lz4hip_container_blocks(kind, flags, level, s + srcOff, (uint64_t)len, (uint32_t)blockSize, d + destOff, (uint64_t)destCap, &out);
  return rc == 0 ? (jlong)out : (jlong)rc;   /* (status codes are negative) */
}

/* the READ side (include/lz4hip.h lz4hip_container_decode): the data blocks of an LZ4 Frame body (kind 0: from the first block's size
 * word on; flags & 1 = block checksums; maxBlock = the frame's block maximum) / of an LZ4Block stream (kind 1) in
 * src[srcOff, srcOff + len) are walked, verified and decoded on the device -- LZ4FrameInputStream.readBlock (LZ4FrameInputStream.java:
 * 258-322) / LZ4BlockInputStream.refill (LZ4BlockInputStream.java:191-264) for a run of up to nMax blocks in one call.  The decoded
 * blocks land back to back at dest[destOff ..); sizes[k] = decoded size of block k; info[0..4] = { blocks delivered, bytes of src
 * consumed, stop reason, decoded bytes, liblz4's code of a failed decode }.  Direct buffers; returns 0 or the (negative)
 * lz4hip_status.  LZ4HIPBatch.containerDecodeBound tells how much of dest a call can need. */
JNIEXPORT jint JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1containerDecode(JNIEnv* env, jclass cls, jint kind, jint flags, jobject src, jlong srcOff,
    jlong len, jint maxBlock, jint nMax, jobject dest, jlong destOff, jlong destCap, jintArray sizes, jlongArray info) {
  (void)cls;
  const uint8_t* s = (const uint8_t*)(*env)->GetDirectBufferAddress(env, src);
  uint8_t* d = (uint8_t*)(*env)->GetDirectBufferAddress(env, dest);
  if (s == NULL || d == NULL || srcOff < 0 || len < 0 || destOff < 0 || destCap < 0 || maxBlock <= 0 || nMax <= 0) return LZ4HIP_E_ARG;
  if (sizes == NULL || info == NULL) return LZ4HIP_E_ARG;   /* (GetArrayLength of a null reference crashes a real JVM: round-5 advisor) */
  if ((*env)->GetArrayLength(env, sizes) < nMax || (*env)->GetArrayLength(env, info) < 5) return LZ4HIP_E_ARG;
  jint* sz = (*env)->GetIntArrayElements(env, sizes, NULL);
  jlong* inf = (*env)->GetLongArrayElements(env, info, NULL);
  jint rc = LZ4HIP_E_NOMEM;
  if (sz && inf) rc = lz4hip_container_decode(kind, flags, s + srcOff, (uint64_t)len, (uint32_t)maxBlock, (uint32_t)nMax, d + destOff, (uint64_t)destCap,
                                              (int32_t*)sz, (uint64_t*)inf);
  if (sz) (*env)->ReleaseIntArrayElements(env, sizes, sz, 0);
  if (inf) (*env)->ReleaseLongArrayElements(env, info, inf, 0);
  return rc;
}
/* what a containerDecode call can need of dest (a host-side walk of the headers, no device work): returns the bytes, or the (negative)
 * lz4hip_status; blocks[0] = whole blocks of the first nMax that src holds */
JNIEXPORT jlong JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1containerDecodeBound(JNIEnv* env, jclass cls, jint kind, jint flags, jobject src, jlong srcOff,
    jlong len, jint maxBlock, jint nMax, jintArray blocks) {
  (void)cls;
  const uint8_t* s = (const uint8_t*)(*env)->GetDirectBufferAddress(env, src);
  if (s == NULL || srcOff < 0 || len < 0 || blocks == NULL || (*env)->GetArrayLength(env, blocks) < 1) return (jlong)LZ4HIP_E_ARG;
  uint32_t nb = 0;
  uint64_t need = 0;
  const int rc = lz4hip_container_decode_bound(kind, flags, s + srcOff, (uint64_t)len, (uint32_t)maxBlock, (uint32_t)nMax, &nb, &need);
  if (rc != 0) return (jlong)rc;
  jint* b = (*env)->GetIntArrayElements(env, blocks, NULL);
  if (b == NULL) return (jlong)LZ4HIP_E_NOMEM;
  b[0] = (jint)nb;
  (*env)->ReleaseIntArrayElements(env, blocks, b, 0);
  return (jlong)need;
}

/* ---- xxhash (XXHashJNI.c:42-82, :152-192 counterparts) ---- */
static int hash_region(JNIEnv* env, jbyteArray arr, jobject buf, jint off, jint len, int is64, uint64_t seed, uint64_t* out) {
  region_t in;
  if (region_in(env, arr, buf, off, len, 1, &in) != 0) { throw_OOM(env); return -1; }
  int rc;
  if (is64) rc = lz4hip_xxh64(in.p, len, seed, out);
  else { uint32_t h = 0; rc = lz4hip_xxh32(in.p, len, (uint32_t)seed, &h); *out = h; }
  region_out(env, NULL, 0, 0, &in);
  if (rc != 0) throw_lib(env);
  return rc;
}

JNIEXPORT jint JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32(JNIEnv* env, jclass cls, jbyteArray buf, jint off, jint len, jint seed) {
  (void)cls; uint64_t h = 0; hash_region(env, buf, NULL, off, len, 0, (uint32_t)seed, &h); return (jint)(uint32_t)h;
}
JNIEXPORT jint JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32BB(JNIEnv* env, jclass cls, jobject buf, jint off, jint len, jint seed) {
  (void)cls; uint64_t h = 0; hash_region(env, NULL, buf, off, len, 0, (uint32_t)seed, &h); return (jint)(uint32_t)h;
}
JNIEXPORT jlong JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64(JNIEnv* env, jclass cls, jbyteArray buf, jint off, jint len, jlong seed) {
  (void)cls; uint64_t h = 0; hash_region(env, buf, NULL, off, len, 1, (uint64_t)seed, &h); return (jlong)h;
}
JNIEXPORT jlong JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64BB(JNIEnv* env, jclass cls, jobject buf, jint off, jint len, jlong seed) {
  (void)cls; uint64_t h = 0; hash_region(env, NULL, buf, off, len, 1, (uint64_t)seed, &h); return (jlong)h;
}
JNIEXPORT jint JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32Batch(JNIEnv* env, jclass cls, jobject input, jlongArray off, jintArray len, jint seed,
    jintArray out32, jint n) {
  (void)cls;
  const uint8_t* s = (const uint8_t*)(*env)->GetDirectBufferAddress(env, input);
  if (s == NULL) return LZ4HIP_E_ARG;
  jlong* o = (*env)->GetLongArrayElements(env, off, NULL);
  jint* l = (*env)->GetIntArrayElements(env, len, NULL);
  jint* h = (*env)->GetIntArrayElements(env, out32, NULL);
  jint rc = (o && l && h) ? lz4hip_xxh32_batch(s, (const uint64_t*)o, (const int32_t*)l, (uint32_t)seed, (uint32_t*)h, (uint32_t)n) : LZ4HIP_E_NOMEM;
  if (o) (*env)->ReleaseLongArrayElements(env, off, o, JNI_ABORT);
  if (l) (*env)->ReleaseIntArrayElements(env, len, l, JNI_ABORT);
  if (h) (*env)->ReleaseIntArrayElements(env, out32, h, 0);
  return rc;
}
JNIEXPORT jint JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64Batch(JNIEnv* env, jclass cls, jobject input, jlongArray off, jintArray len, jlong seed,
    jlongArray out64, jint n) {
  (void)cls;
  const uint8_t* s = (const uint8_t*)(*env)->GetDirectBufferAddress(env, input);
  if (s == NULL) return LZ4HIP_E_ARG;
  jlong* o = (*env)->GetLongArrayElements(env, off, NULL);
  jint* l = (*env)->GetIntArrayElements(env, len, NULL);
  jlong* h = (*env)->GetLongArrayElements(env, out64, NULL);
  jint rc = (o && l && h) ? lz4hip_xxh64_batch(s, (const uint64_t*)o, (const int32_t*)l, (uint64_t)seed, (uint64_t*)h, (uint32_t)n) : LZ4HIP_E_NOMEM;
  if (o) (*env)->ReleaseLongArrayElements(env, off, o, JNI_ABORT);
  if (l) (*env)->ReleaseIntArrayElements(env, len, l, JNI_ABORT);
  if (h) (*env)->ReleaseLongArrayElements(env, out64, h, 0);
  return rc;
}

/* ---- streaming xxhash (XXHashJNI.c:89-145, :199-255 counterparts): the jlong is a lz4hip_xxh_stream* ---- */
static jlong stream_init(JNIEnv* env, int is64, uint64_t seed) {
  lz4hip_xxh_stream* st = NULL;
  const int rc = is64 ? lz4hip_xxh64_stream_create(seed, &st) : lz4hip_xxh32_stream_create((uint32_t)seed, &st);
  if (rc != 0) { throw_OOM(env); return 0; }   /* same surface as the reference: XXHashJNI.c:94-98 */
  return (jlong)(intptr_t)st;
}
static void stream_update(JNIEnv* env, jlong state, jbyteArray src, jint off, jint len) {
  region_t in;   /* staged copy: the GC lock is not held across the launch */
  if (region_in(env, src, NULL, off, len, 1, &in) != 0) { throw_OOM(env); return; }
  const int rc = lz4hip_xxh_stream_update((lz4hip_xxh_stream*)(intptr_t)state, in.p, len);
  region_out(env, NULL, 0, 0, &in);
  if (rc != 0) throw_lib(env);
}
JNIEXPORT jlong JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32_1init(JNIEnv* env, jclass cls, jint seed) { (void)cls; return stream_init(env, 0, (uint32_t)seed); }
JNIEXPORT jlong JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64_1init(JNIEnv* env, jclass cls, jlong seed) { (void)cls; return stream_init(env, 1, (uint64_t)seed); }
JNIEXPORT void JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32_1reset(JNIEnv* env, jclass cls, jlong state, jint seed) {
  (void)cls; if (lz4hip_xxh_stream_reset((lz4hip_xxh_stream*)(intptr_t)state, (uint32_t)seed) != 0) throw_lib(env);
}
JNIEXPORT void JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64_1reset(JNIEnv* env, jclass cls, jlong state, jlong seed) {
  (void)cls; if (lz4hip_xxh_stream_reset((lz4hip_xxh_stream*)(intptr_t)state, (uint64_t)seed) != 0) throw_lib(env);
}
JNIEXPORT void JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32_1update(JNIEnv* env, jclass cls, jlong state, jbyteArray src, jint off, jint len) {
  (void)cls; stream_update(env, state, src, off, len);
}
JNIEXPORT void JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64_1update(JNIEnv* env, jclass cls, jlong state, jbyteArray src, jint off, jint len) {
  (void)cls; stream_update(env, state, src, off, len);
}
JNIEXPORT jint JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32_1digest(JNIEnv* env, jclass cls, jlong state) {
  (void)cls; uint32_t h = 0; if (lz4hip_xxh32_stream_digest((lz4hip_xxh_stream*)(intptr_t)state, &h) != 0) throw_lib(env); return (jint)h;
}
JNIEXPORT jlong JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64_1digest(JNIEnv* env, jclass cls, jlong state) {
  (void)cls; uint64_t h = 0; if (lz4hip_xxh64_stream_digest((lz4hip_xxh_stream*)(intptr_t)state, &h) != 0) throw_lib(env); return (jlong)h;
}
JNIEXPORT void JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH_1free(JNIEnv* env, jclass cls, jlong state) {
  (void)env; (void)cls; lz4hip_xxh_stream_free((lz4hip_xxh_stream*)(intptr_t)state);
}

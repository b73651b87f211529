/* tests/jni_stub/fake_jni.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Executes the JNI shim (lz4-java_amd/jni/net_jpountz_lz4_LZ4HIPJNI.c) without a JVM: a fake JNIEnv function table in
 * which a Java byte[] / int[] / long[] is a malloc'd buffer with pin accounting and a direct ByteBuffer is a pointer.  The shim is
 * compiled against tests/jni_stub/jni.h into this program and linked with liblz4hip.so, so every Java_net_jpountz_* entry point
 * the Java classes declare runs against the real C ABI -- including the paths the reference gets wrong or cannot have (the
 * `out` array that cannot be pinned: LZ4JNI.c:59-73 leaks `in`; a library failure inside an xxhash call).
 *
 *   fake_jni            on a GPU box: the full scenario list, prints "fake_jni: N checks ok"
 *   fake_jni --no-gpu   anywhere: only what must hold without a device (every compute call fails LOUDLY: exception or error code)
 * The shim's malloc/free are counted (-Dmalloc=t_malloc -Dfree=t_free on its translation unit).
 */
#include <jni.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lz4hip.h"

/* ---- fake objects ---- */
typedef struct {
  int kind;          /* 1 = byte[], 2 = int[], 3 = long[], 4 = direct ByteBuffer, 5 = heap ByteBuffer (no direct address) */
  uint8_t* data;
  size_t bytes;
  int pins;          /* outstanding Get*Critical / Get*ArrayElements */
  int refuse_pin;    /* GetPrimitiveArrayCritical returns NULL (a VM that cannot pin) */
} fobj;

static long g_alloc = 0;       /* outstanding shim allocations */
void* t_malloc(size_t n) { g_alloc++; return malloc(n); }
void t_free(void* p) { if (p) g_alloc--; free(p); }

static const char* g_exc_class = NULL;
static char g_exc_msg[512];
static int g_checks = 0;

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "fake_jni: CHECK failed at line %d: %s (pending exception: %s \"%s\")\n", __LINE__, #c, \
    g_exc_class ? g_exc_class : "none", g_exc_msg); exit(1); } g_checks++; } while (0)

static jclass f_FindClass(JNIEnv* e, const char* name) { (void)e; return (jclass)strdup(name); }
static jint f_ThrowNew(JNIEnv* e, jclass c, const char* msg) { (void)e; g_exc_class = (const char*)c; snprintf(g_exc_msg, sizeof g_exc_msg, "%s", msg ? msg : ""); return 0; }
static jobject f_NewGlobalRef(JNIEnv* e, jobject o) { (void)e; return o; }
static void* f_GetCritical(JNIEnv* e, jarray a, jboolean* isCopy) {
  (void)e; fobj* o = (fobj*)a;
  if (isCopy) *isCopy = 0;
  if (o->refuse_pin) return NULL;
  o->pins++;
  return o->data;
}
static void f_ReleaseCritical(JNIEnv* e, jarray a, void* p, jint mode) { (void)e; (void)mode; fobj* o = (fobj*)a; if (p != o->data) { fprintf(stderr, "release of a foreign pointer\n"); exit(1); } o->pins--; }
static void* f_GetDirect(JNIEnv* e, jobject b) { (void)e; fobj* o = (fobj*)b; return o->kind == 4 ? o->data : NULL; }
static jstring f_NewStringUTF(JNIEnv* e, const char* s) { (void)e; return (jstring)strdup(s ? s : ""); }
static jlong* f_GetLongs(JNIEnv* e, jlongArray a, jboolean* c) { (void)e; if (c) *c = 0; ((fobj*)a)->pins++; return (jlong*)((fobj*)a)->data; }
static jint* f_GetInts(JNIEnv* e, jintArray a, jboolean* c) { (void)e; if (c) *c = 0; ((fobj*)a)->pins++; return (jint*)((fobj*)a)->data; }
static void f_RelLongs(JNIEnv* e, jlongArray a, jlong* p, jint m) { (void)e; (void)p; (void)m; ((fobj*)a)->pins--; }
static void f_RelInts(JNIEnv* e, jintArray a, jint* p, jint m) { (void)e; (void)p; (void)m; ((fobj*)a)->pins--; }

static jint f_ArrayLength(JNIEnv* e, jarray a) { (void)e; const fobj* o = (const fobj*)a; return (jint)(o->bytes / (o->kind == 3 ? 8u : o->kind == 2 ? 4u : 1u)); }

static const struct JNINativeInterface_ g_table = {f_FindClass, f_ThrowNew, f_NewGlobalRef, f_GetCritical, f_ReleaseCritical, f_GetDirect,
                                                   f_NewStringUTF, f_GetLongs, f_GetInts, f_RelLongs, f_RelInts, f_ArrayLength};
static JNIEnv g_env = &g_table;

static fobj* mk(int kind, size_t bytes) { fobj* o = calloc(1, sizeof *o); o->kind = kind; o->bytes = bytes; o->data = calloc(bytes ? bytes : 1, 1); return o; }
static int no_exc(void) { return g_exc_class == NULL; }
static void clear_exc(void) { g_exc_class = NULL; g_exc_msg[0] = 0; }

/* ---- the shim's entry points (same translation unit names as the Java natives) ---- */
JNIEXPORT void JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_init(JNIEnv*, jclass);
JNIEXPORT jint JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1compress_1fast(JNIEnv*, jclass, jbyteArray, jobject, jint, jint, jbyteArray, jobject, jint, jint);
JNIEXPORT jint JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1compressHC(JNIEnv*, jclass, jbyteArray, jobject, jint, jint, jbyteArray, jobject, jint, jint, jint);
JNIEXPORT jint JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1decompress_1safe(JNIEnv*, jclass, jbyteArray, jobject, jint, jint, jbyteArray, jobject, jint, jint);
JNIEXPORT jint JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1decompress_1fast(JNIEnv*, jclass, jbyteArray, jobject, jint, jint, jbyteArray, jobject, jint, jint);
JNIEXPORT jint JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1compressBound(JNIEnv*, jclass, jint);
JNIEXPORT jstring JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_lastError(JNIEnv*, jclass);
JNIEXPORT jint JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1batch(JNIEnv*, jclass, jint, jint, jobject, jlongArray, jintArray, jobject, jlongArray, jintArray, jintArray, jint);
JNIEXPORT jlong JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1containerBlocks(JNIEnv*, jclass, jint, jint, jint, jobject, jlong, jlong, jint, jobject, jlong, jlong);
JNIEXPORT jint JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1containerDecode(JNIEnv*, jclass, jint, jint, jobject, jlong, jlong, jint, jint, jobject, jlong, jlong, jintArray, jlongArray);
JNIEXPORT jlong JNICALL Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1containerDecodeBound(JNIEnv*, jclass, jint, jint, jobject, jlong, jlong, jint, jint, jintArray);
JNIEXPORT jint JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32(JNIEnv*, jclass, jbyteArray, jint, jint, jint);
JNIEXPORT jint JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32BB(JNIEnv*, jclass, jobject, jint, jint, jint);
JNIEXPORT jlong JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64(JNIEnv*, jclass, jbyteArray, jint, jint, jlong);
JNIEXPORT jlong JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64BB(JNIEnv*, jclass, jobject, jint, jint, jlong);
JNIEXPORT jint JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32Batch(JNIEnv*, jclass, jobject, jlongArray, jintArray, jint, jintArray, jint);
JNIEXPORT jint JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64Batch(JNIEnv*, jclass, jobject, jlongArray, jintArray, jlong, jlongArray, jint);
JNIEXPORT jlong JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32_1init(JNIEnv*, jclass, jint);
JNIEXPORT jlong JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64_1init(JNIEnv*, jclass, jlong);
JNIEXPORT void JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32_1reset(JNIEnv*, jclass, jlong, jint);
JNIEXPORT void JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32_1update(JNIEnv*, jclass, jlong, jbyteArray, jint, jint);
JNIEXPORT void JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64_1update(JNIEnv*, jclass, jlong, jbyteArray, jint, jint);
JNIEXPORT jint JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32_1digest(JNIEnv*, jclass, jlong);
JNIEXPORT jlong JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64_1digest(JNIEnv*, jclass, jlong);
JNIEXPORT void JNICALL Java_net_jpountz_xxhash_XXHashHIPJNI_XXH_1free(JNIEnv*, jclass, jlong);

static const uint8_t SELFTEST_IN[] = "abcd      abcdefghij";                    /* LZ4Factory.java:205 */
static const uint8_t SELFTEST_OUT[] = {0x51, 0x61, 0x62, 0x63, 0x64, 0x20, 0x01, 0x00, 0xa0, 0x61, 0x62, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a};
static const uint8_t README_IN[] = "12345345234572";                            /* README.md:54 */

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_state >> 33); }

int main(int argc, char** argv) {
  const int gpu = !(argc > 1 && strcmp(argv[1], "--no-gpu") == 0);
  JNIEnv* env = &g_env;
  Java_net_jpountz_lz4_LZ4HIPJNI_init(env, NULL);
  CHECK(no_exc());
  CHECK(Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1compressBound(env, NULL, 65536) == 65809);
  CHECK(Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1compressBound(env, NULL, 0x7E000001) == 0);

  fobj* src = mk(1, 64); fobj* dst = mk(1, 128); fobj* back = mk(1, 64);
  memcpy(src->data + 7, SELFTEST_IN, 20);
  if (!gpu) {
    /* no device: every compute call reports a LIBRARY error (never a fake result), hashes throw, nothing leaks or stays pinned */
    jint r = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1compress_1fast(env, NULL, (jbyteArray)src, NULL, 7, 20, (jbyteArray)dst, NULL, 3, 100);
    CHECK(LZ4HIP_IS_LIB_ERROR(r) && no_exc() && g_alloc == 0 && src->pins == 0 && dst->pins == 0);
    const char* msg = (const char*)Java_net_jpountz_lz4_LZ4HIPJNI_lastError(env, NULL);
    CHECK(msg && strlen(msg) > 0);
    (void)Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32(env, NULL, (jbyteArray)src, 7, 14, 0);
    CHECK(g_exc_class && strcmp(g_exc_class, "java/lang/RuntimeException") == 0 && g_alloc == 0 && src->pins == 0);
    clear_exc();
    (void)Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64(env, NULL, (jbyteArray)src, 7, 14, 0);
    CHECK(g_exc_class && strcmp(g_exc_class, "java/lang/RuntimeException") == 0);
    clear_exc();
    jlong st = Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32_1init(env, NULL, 0);
    CHECK(st == 0 && g_exc_class != NULL);   /* creation fails loudly too */
    printf("fake_jni: %d checks ok (no device: every compute call failed loudly)\n", g_checks);
    return 0;
  }

  /* ---- 1. LZ4Factory's constructor self-test through byte[] arguments with offsets (LZ4Factory.java:176-220) ---- */
  jint r = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1compress_1fast(env, NULL, (jbyteArray)src, NULL, 7, 20, (jbyteArray)dst, NULL, 3, 100);
  CHECK(no_exc() && r == (jint)sizeof SELFTEST_OUT && memcmp(dst->data + 3, SELFTEST_OUT, sizeof SELFTEST_OUT) == 0);
  CHECK(dst->data[2] == 0 && dst->data[3 + r] == 0);                 /* nothing outside [destOff, destOff + r) */
  r = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1decompress_1safe(env, NULL, (jbyteArray)dst, NULL, 3, (jint)sizeof SELFTEST_OUT, (jbyteArray)back, NULL, 5, 40);
  CHECK(no_exc() && r == 20 && memcmp(back->data + 5, SELFTEST_IN, 20) == 0);
  memset(back->data, 0, 64);
  r = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1decompress_1fast(env, NULL, (jbyteArray)dst, NULL, 3, 125, (jbyteArray)back, NULL, 5, 20);
  CHECK(no_exc() && r == (jint)sizeof SELFTEST_OUT && memcmp(back->data + 5, SELFTEST_IN, 20) == 0 && back->data[25] == 0);
  r = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1compressHC(env, NULL, (jbyteArray)src, NULL, 7, 20, (jbyteArray)dst, NULL, 3, 100, 9);
  CHECK(no_exc() && r == (jint)sizeof SELFTEST_OUT && memcmp(dst->data + 3, SELFTEST_OUT, sizeof SELFTEST_OUT) == 0);   /* fast == HC9 here */
  CHECK(src->pins == 0 && dst->pins == 0 && back->pins == 0 && g_alloc == 0);
  /* dest too small: liblz4's 0, no exception from the shim (the Java class turns it into LZ4Exception) */
  r = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1compress_1fast(env, NULL, (jbyteArray)src, NULL, 7, 20, (jbyteArray)dst, NULL, 3, 10);
  CHECK(no_exc() && r == 0 && g_alloc == 0);
  /* malformed input: negative code passes through */
  { static const uint8_t bad[] = {0x60, 42, 43, 44, 45, 46, 47, 5, 0};   /* LZ4Test.java:366 */
    memcpy(dst->data, bad, sizeof bad);
    r = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1decompress_1safe(env, NULL, (jbyteArray)dst, NULL, 0, (jint)sizeof bad, (jbyteArray)back, NULL, 0, 20);
    CHECK(no_exc() && r == -2); }

  /* ---- 2. direct ByteBuffers (no staging copy), and the mixed cases of AbstractLZ4Test.java:66-116 ---- */
  fobj* dsrc = mk(4, 70000); fobj* ddst = mk(4, 80000); fobj* dback = mk(4, 70000);
  for (size_t i = 0; i < 70000; i++) dsrc->data[i] = (uint8_t)((i % 700) < 300 ? rnd() : (i * 7) >> 3);
  r = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1compress_1fast(env, NULL, NULL, (jobject)dsrc, 11, 65536, NULL, (jobject)ddst, 13, 70000);
  CHECK(no_exc() && r > 0 && r < 65536 && g_alloc == 0);
  const jint clen = r;
  r = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1decompress_1safe(env, NULL, NULL, (jobject)ddst, 13, clen, NULL, (jobject)dback, 17, 65536);
  CHECK(no_exc() && r == 65536 && memcmp(dback->data + 17, dsrc->data + 11, 65536) == 0);
  { fobj* harr = mk(1, 70000);   /* direct source -> heap destination */
    r = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1decompress_1fast(env, NULL, NULL, (jobject)ddst, 13, 70000 - 13, (jbyteArray)harr, NULL, 1, 65536);
    CHECK(no_exc() && r == clen && memcmp(harr->data + 1, dsrc->data + 11, 65536) == 0 && harr->pins == 0 && g_alloc == 0); }
  /* a heap ByteBuffer passed where a direct one is required has no address: OutOfMemoryError-class failure, as in the reference */
  { fobj* hb = mk(5, 64);
    r = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1compress_1fast(env, NULL, NULL, (jobject)hb, 0, 20, (jbyteArray)dst, NULL, 0, 100);
    CHECK(g_exc_class && strcmp(g_exc_class, "java/lang/OutOfMemoryError") == 0 && g_alloc == 0);
    clear_exc(); }

  /* ---- 3. the path the reference leaks: `out` cannot be pinned -> `in` is released, OOM is thrown ---- */
  { fobj* nopin = mk(1, 128); nopin->refuse_pin = 1;
    r = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1compress_1fast(env, NULL, (jbyteArray)src, NULL, 7, 20, (jbyteArray)nopin, NULL, 0, 100);
    CHECK(g_exc_class && strcmp(g_exc_class, "java/lang/OutOfMemoryError") == 0);
    CHECK(g_alloc == 0 && src->pins == 0 && nopin->pins == 0);     /* nothing leaked, nothing left pinned */
    clear_exc();
    src->refuse_pin = 1;                                            /* and `in` itself */
    r = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1compress_1fast(env, NULL, (jbyteArray)src, NULL, 7, 20, (jbyteArray)dst, NULL, 0, 100);
    CHECK(g_exc_class && strcmp(g_exc_class, "java/lang/OutOfMemoryError") == 0 && g_alloc == 0 && dst->pins == 0);
    clear_exc();
    src->refuse_pin = 0; }

  /* ---- 4. LZ4HIP_batch: many blocks, direct buffers, long[] / int[] descriptors ---- */
  { const int n = 8, blk = 8192;
    fobj* so = mk(3, 8 * n); fobj* sl = mk(2, 4 * n); fobj* dof = mk(3, 8 * n); fobj* dc = mk(2, 4 * n); fobj* ol = mk(2, 4 * n);
    const int bound = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1compressBound(env, NULL, blk);
    for (int i = 0; i < n; i++) { ((jlong*)so->data)[i] = 100 + (jlong)i * blk; ((jint*)sl->data)[i] = blk; ((jlong*)dof->data)[i] = (jlong)i * bound; ((jint*)dc->data)[i] = bound; }
    r = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1batch(env, NULL, 0, 0, (jobject)dsrc, (jlongArray)so, (jintArray)sl, (jobject)ddst, (jlongArray)dof, (jintArray)dc, (jintArray)ol, n);
    CHECK(no_exc() && r == 0 && so->pins == 0 && sl->pins == 0 && dof->pins == 0 && dc->pins == 0 && ol->pins == 0);
    fobj* cl = mk(2, 4 * n); fobj* bo = mk(3, 8 * n); fobj* bl = mk(2, 4 * n); fobj* res = mk(2, 4 * n);
    for (int i = 0; i < n; i++) { CHECK(((jint*)ol->data)[i] > 0); ((jint*)cl->data)[i] = ((jint*)ol->data)[i]; ((jlong*)bo->data)[i] = (jlong)i * blk; ((jint*)bl->data)[i] = blk; }
    r = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1batch(env, NULL, 1, 0, (jobject)ddst, (jlongArray)dof, (jintArray)cl, (jobject)dback, (jlongArray)bo, (jintArray)bl, (jintArray)res, n);
    CHECK(no_exc() && r == 0);
    for (int i = 0; i < n; i++) CHECK(((jint*)res->data)[i] == blk);
    CHECK(memcmp(dback->data, dsrc->data + 100, (size_t)n * blk) == 0);
    /* HC through the same entry (op 3) decodes back as well */
    r = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1batch(env, NULL, 3, 9, (jobject)dsrc, (jlongArray)so, (jintArray)sl, (jobject)ddst, (jlongArray)dof, (jintArray)dc, (jintArray)ol, n);
    CHECK(no_exc() && r == 0);
    for (int i = 0; i < n; i++) ((jint*)cl->data)[i] = ((jint*)ol->data)[i];
    memset(dback->data, 0, (size_t)n * blk);
    r = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1batch(env, NULL, 2, 0, (jobject)ddst, (jlongArray)dof, (jintArray)dc, (jobject)dback, (jlongArray)bo, (jintArray)bl, (jintArray)res, n);
    CHECK(no_exc() && r == 0 && memcmp(dback->data, dsrc->data + 100, (size_t)n * blk) == 0);
    for (int i = 0; i < n; i++) CHECK(((jint*)res->data)[i] == ((jint*)cl->data)[i]);
    /* container blocks assembled on the device (LZ4HIPBatch.containerBlocks): frame blocks of the same n blocks = n x {size word,
       payload}; every payload is the compressed block the batch call produced, the size words say so */
    r = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1batch(env, NULL, 0, 0, (jobject)dsrc, (jlongArray)so, (jintArray)sl, (jobject)ddst, (jlongArray)dof, (jintArray)dc, (jintArray)ol, n);
    CHECK(no_exc() && r == 0);
    { const jlong got = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1containerBlocks(env, NULL, 0, 0, 0, (jobject)dsrc, 100, (jlong)n * blk, blk, (jobject)dback, 0, (jlong)dback->bytes);
      CHECK(no_exc() && got > 0);
      size_t p = 0;
      for (int i = 0; i < n; i++) {
        const uint32_t w = (uint32_t)dback->data[p] | ((uint32_t)dback->data[p + 1] << 8) | ((uint32_t)dback->data[p + 2] << 16) | ((uint32_t)dback->data[p + 3] << 24);
        const jint c = ((jint*)ol->data)[i];
        CHECK(w == (uint32_t)c && memcmp(dback->data + p + 4, ddst->data + (size_t)i * bound, (size_t)c) == 0);
        p += 4 + (size_t)c;
      }
      CHECK((jlong)p == got);
      CHECK(Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1containerBlocks(env, NULL, 0, 0, 0, (jobject)dsrc, 100, (jlong)n * blk, blk, (jobject)dback, 0, 10) < 0);
      /* ... and the READ side (LZ4HIPBatch.containerDecode): the frame body just assembled, walked and decoded on the device: every
         block back, the whole body consumed, stop reason 1 (the body ended at a block boundary); then the same body with an end mark
         behind it (reason 0), cut short inside the last block (reason 2, n - 1 blocks), and with a damaged payload (reason 5) */
      fobj* body = mk(4, (size_t)got + 16);
      memcpy(body->data, dback->data, (size_t)got);
      fobj* szs = mk(2, sizeof(jint) * 64); fobj* inf = mk(3, sizeof(jlong) * 5); fobj* nbk = mk(2, sizeof(jint));
      const jlong need = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1containerDecodeBound(env, NULL, 0, 0, (jobject)body, 0, got, blk, 64, (jintArray)nbk);
      CHECK(no_exc() && need == (jlong)n * blk && ((jint*)nbk->data)[0] == n);
      fobj* dec = mk(4, (size_t)need + 64);
      jint rc2 = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1containerDecode(env, NULL, 0, 0, (jobject)body, 0, got, blk, 64, (jobject)dec, 32, need, (jintArray)szs, (jlongArray)inf);
      CHECK(no_exc() && rc2 == 0);
      { const jlong* I = (const jlong*)inf->data;
        CHECK(I[0] == n && I[1] == got && I[2] == 1 && I[3] == (jlong)n * blk);
        for (int i = 0; i < n; i++) CHECK(((jint*)szs->data)[i] == blk);
        CHECK(memcmp(dec->data + 32, dsrc->data + 100, (size_t)n * blk) == 0); }
      memset(body->data + got, 0, 4);                                   /* the end mark */
      rc2 = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1containerDecode(env, NULL, 0, 0, (jobject)body, 0, got + 4, blk, 64, (jobject)dec, 32, need, (jintArray)szs, (jlongArray)inf);
      CHECK(rc2 == 0 && ((jlong*)inf->data)[0] == n && ((jlong*)inf->data)[1] == got + 4 && ((jlong*)inf->data)[2] == 0);
      rc2 = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1containerDecode(env, NULL, 0, 0, (jobject)body, 0, got - 7, blk, 64, (jobject)dec, 32, need, (jintArray)szs, (jlongArray)inf);
      CHECK(rc2 == 0 && ((jlong*)inf->data)[0] == n - 1 && ((jlong*)inf->data)[2] == 2);
      memset(body->data + 4, 0xFF, 12);                                 /* block 0's payload: a literal-length run that cannot fit the block */
      rc2 = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1containerDecode(env, NULL, 0, 0, (jobject)body, 0, got, blk, 64, (jobject)dec, 32, need, (jintArray)szs, (jlongArray)inf);
      CHECK(rc2 == 0 && ((jlong*)inf->data)[0] == 0 && ((jlong*)inf->data)[1] == 0 && ((jlong*)inf->data)[2] == 5 && ((jlong*)inf->data)[4] < 0);   /* nothing delivered, "the block does not decode", liblz4's negative code */
      CHECK(Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1containerDecode(env, NULL, 0, 0, (jobject)body, 0, got, blk, 64, (jobject)dec, 32, 10, (jintArray)szs, (jlongArray)inf) != 0 ||
            ((jlong*)inf->data)[0] == 0);                               /* a destination that is too small is an error, not an overrun */
      /* null array references (round-5 advisor: GetArrayLength of null crashes a real JVM -- and this fake one): an argument error, no call into the env */
      CHECK(Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1containerDecodeBound(env, NULL, 0, 0, (jobject)body, 0, got, blk, 64, NULL) < 0);
      CHECK(Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1containerDecode(env, NULL, 0, 0, (jobject)body, 0, got, blk, 64, (jobject)dec, 32, need, NULL, (jlongArray)inf) != 0);
      CHECK(Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1containerDecode(env, NULL, 0, 0, (jobject)body, 0, got, blk, 64, (jobject)dec, 32, need, (jintArray)szs, NULL) != 0);
      CHECK(no_exc() && szs->pins == 0 && inf->pins == 0 && nbk->pins == 0);
      free(body->data); free(body); free(szs->data); free(szs); free(inf->data); free(inf); free(nbk->data); free(nbk); free(dec->data); free(dec); } }

  /* ---- 5. xxhash: one-shot (heap + direct), batch, streaming; known answers of SURVEY App. D ---- */
  memcpy(src->data + 7, README_IN, 14);
  CHECK((uint32_t)Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32(env, NULL, (jbyteArray)src, 7, 14, 0) == 0xeccb33ceu);
  CHECK((uint32_t)Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32(env, NULL, (jbyteArray)src, 7, 14, (jint)0x9747b28c) == 0x1e34488cu);
  CHECK((uint64_t)Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64(env, NULL, (jbyteArray)src, 7, 14, 0) == 0xf46bd83bde991b30ull);
  CHECK((uint64_t)Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64(env, NULL, (jbyteArray)src, 7, 14, (jlong)0x9747b28c) == 0xea6b9bde2112e286ull);
  CHECK((uint32_t)Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32(env, NULL, (jbyteArray)src, 7, 0, 0) == 0x02cc5d05u);
  CHECK((uint64_t)Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64(env, NULL, (jbyteArray)src, 7, 0, 0) == 0xef46db3751d8e999ull);
  memcpy(dsrc->data + 33, README_IN, 14);
  CHECK((uint32_t)Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32BB(env, NULL, (jobject)dsrc, 33, 14, 0) == 0xeccb33ceu);
  CHECK((uint64_t)Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64BB(env, NULL, (jobject)dsrc, 33, 14, 0) == 0xf46bd83bde991b30ull);
  CHECK(no_exc() && g_alloc == 0 && src->pins == 0);
  { const int n = 5;
    fobj* off = mk(3, 8 * n); fobj* len = mk(2, 4 * n); fobj* o32 = mk(2, 4 * n); fobj* o64 = mk(3, 8 * n);
    for (int i = 0; i < n; i++) { ((jlong*)off->data)[i] = 33; ((jint*)len->data)[i] = i == 4 ? 0 : 14; }
    CHECK(Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32Batch(env, NULL, (jobject)dsrc, (jlongArray)off, (jintArray)len, 0, (jintArray)o32, n) == 0);
    CHECK(Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64Batch(env, NULL, (jobject)dsrc, (jlongArray)off, (jintArray)len, 0, (jlongArray)o64, n) == 0);
    CHECK(((uint32_t*)o32->data)[0] == 0xeccb33ceu && ((uint32_t*)o32->data)[4] == 0x02cc5d05u && ((uint64_t*)o64->data)[3] == 0xf46bd83bde991b30ull);
    CHECK(off->pins == 0 && len->pins == 0 && o32->pins == 0 && o64->pins == 0); }
  { jlong s32 = Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32_1init(env, NULL, 0), s64 = Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64_1init(env, NULL, 0);
    CHECK(no_exc() && s32 && s64);
    Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32_1update(env, NULL, s32, (jbyteArray)src, 7, 5);
    Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32_1update(env, NULL, s32, (jbyteArray)src, 12, 9);
    Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64_1update(env, NULL, s64, (jbyteArray)src, 7, 1);
    Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64_1update(env, NULL, s64, (jbyteArray)src, 8, 13);
    CHECK((uint32_t)Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32_1digest(env, NULL, s32) == 0xeccb33ceu);
    CHECK((uint64_t)Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64_1digest(env, NULL, s64) == 0xf46bd83bde991b30ull);
    Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32_1reset(env, NULL, s32, (jint)0x9747b28c);
    Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32_1update(env, NULL, s32, (jbyteArray)src, 7, 14);
    CHECK((uint32_t)Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32_1digest(env, NULL, s32) == 0x1e34488cu && no_exc() && g_alloc == 0);
    Java_net_jpountz_xxhash_XXHashHIPJNI_XXH_1free(env, NULL, s32);
    Java_net_jpountz_xxhash_XXHashHIPJNI_XXH_1free(env, NULL, s64); }
  /* a library failure inside a hash call is an exception, not a stale digest (a null stream handle: liblz4hip rejects it) */
  Java_net_jpountz_xxhash_XXHashHIPJNI_XXH32_1update(env, NULL, 0, (jbyteArray)src, 7, 5);
  CHECK(g_exc_class && strcmp(g_exc_class, "java/lang/RuntimeException") == 0 && g_alloc == 0 && src->pins == 0);
  clear_exc();
  (void)Java_net_jpountz_xxhash_XXHashHIPJNI_XXH64_1digest(env, NULL, 0);
  CHECK(g_exc_class && strcmp(g_exc_class, "java/lang/RuntimeException") == 0);
  clear_exc();

  /* ---- 6. a few hundred random round trips through byte[] with random offsets: pins and allocations stay balanced ---- */
  for (int it = 0; it < 200; it++) {
    const int n = (int)(rnd() % 5000u), so_ = (int)(rnd() % 31u), do_ = (int)(rnd() % 17u);
    fobj* a = mk(1, (size_t)n + 40); fobj* c = mk(1, (size_t)n + n / 200 + 64); fobj* b = mk(1, (size_t)n + 40);
    for (int i = 0; i < n; i++) a->data[so_ + i] = (uint8_t)((i % 97) < 40 ? rnd() : i / 3);
    const jint cap = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1compressBound(env, NULL, n);
    jint cl = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1compress_1fast(env, NULL, (jbyteArray)a, NULL, so_, n, (jbyteArray)c, NULL, do_, cap <= (jint)c->bytes - do_ ? cap : (jint)c->bytes - do_);
    CHECK(no_exc() && cl > 0);
    jint dl = Java_net_jpountz_lz4_LZ4HIPJNI_LZ4HIP_1decompress_1safe(env, NULL, (jbyteArray)c, NULL, do_, cl, (jbyteArray)b, NULL, so_, n);
    CHECK(no_exc() && dl == n && memcmp(a->data + so_, b->data + so_, (size_t)n) == 0 && a->pins == 0 && b->pins == 0 && c->pins == 0 && g_alloc == 0);
    free(a->data); free(a); free(b->data); free(b); free(c->data); free(c);
  }
  printf("fake_jni: %d checks ok\n", g_checks);
  return 0;
}

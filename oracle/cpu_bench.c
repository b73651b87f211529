/*
 * oracle/cpu_bench.c -- TEST/BENCH INFRASTRUCTURE ONLY: the CPU baseline that bench.py reports next
 * to the GPU number ("cpu_baseline").  Times, on THIS box's host cores, exactly the functions the
 * reference's JNI shim calls (LZ4_compress_default, LZ4_decompress_safe, LZ4_decompress_fast:
 * /root/reference/src/jni/net_jpountz_lz4_LZ4JNI.c:75,216,169) on the same synthetic blocks the GPU
 * processes.  kind "reference" = dlopen of the reference's own liblz4 1.9.3
 * (oracle/_ref/liblz4-java.so); kind "port" = the C restatement in lz4_oracle.c.  No JVM exists in
 * this image, so JNI call overhead (array pinning + one call, sub-microsecond vs >= 25 us of work per
 * 64 KiB block) is not included.
 *
 *   cpu_bench <reference|port> <lib.so> <liblz4oracle.so> <n_blocks> <block_size> <threads> <reps> <first_idx> <litmax> <win>
 * prints one JSON line.  With LZ4_HC_LEVEL=<1..9> in the environment the compress leg is LZ4_compress_HC at that level
 * (LZ4JNI.c:122; row a4) instead of LZ4_compress_default ("reference" kind only).  With XXH_MODE=1 the buffers are hashed
 * instead: XXH32 and XXH64 (seed 0x9747b28c) of every block (src/jni/net_jpountz_xxhash_XXHashJNI.c:54,164; row a6), and the
 * line reports xxh32_GBps / xxh64_GBps.  With CPU_BENCH_FILE=<path> the blocks are slices of that file (offset (i * 7919) mod
 * (length - block size): the real-text leg of bench.py) instead of generated ones.  Every rate is reported twice: best of <reps> and (..._median) the median over them.
 *
 * Timing discipline (round 4): a PERSISTENT pool of <threads> workers, each pinned to one of the CPUs the process may run on,
 * created once before anything is timed; every buffer is first touched by the worker that owns its blocks (generation, warm-up
 * passes); a timed repetition starts and ends at a barrier the main thread takes part in and consists of as many whole passes over
 * the sample as it takes to last CPU_BENCH_MIN_MS (default 300) milliseconds, so a repetition never measures thread start-up or a
 * single 10 ms burst.  Inside a repetition a worker walks its own contiguous share of the blocks pass after pass and, when it has
 * finished all of them, takes blocks from the shares of workers that are behind (one atomic ticket per block), so one descheduled
 * worker does not set the time.  The line reports "passes" (per repetition) and "min_ms".
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int (*compress_fn)(const char*, char*, int, int);
typedef int (*compress_hc_fn)(const char*, char*, int, int, int);
typedef int (*dsafe_fn)(const char*, char*, int, int);
typedef int (*dfast_fn)(const char*, char*, int);
typedef int (*p_compress_fn)(const uint8_t*, int, uint8_t*, int);
typedef int (*p_dsafe_fn)(const uint8_t*, int, uint8_t*, int);
typedef int (*p_dfast_fn)(const uint8_t*, uint8_t*, int);
typedef void (*gen_fn)(uint8_t*, int64_t, uint64_t, uint64_t, uint32_t, uint32_t);

static int is_ref, hc_level;
static compress_hc_fn r_hc;
static compress_fn r_c; static dsafe_fn r_ds; static dfast_fn r_df;
static p_compress_fn p_c; static p_dsafe_fn p_ds; static p_dfast_fn p_df;
static gen_fn gen;
static int n_blocks, block_size, n_threads, bound;
static uint64_t first_idx; static uint32_t litmax, win;
static uint8_t *src, *comp, *back; static int* clen;
typedef uint32_t (*xxh32_fn)(const void*, size_t, uint32_t);
typedef uint64_t (*xxh64_fn)(const void*, size_t, uint64_t);
static xxh32_fn x32; static xxh64_fn x64;
static uint64_t* hsum;
static int phase;  /* 0 gen, 1 compress, 2 dsafe, 3 dfast, 4 xxh32, 5 xxh64 */
static volatile int bad = 0;
static uint8_t* file_buf; static long file_len;   /* CPU_BENCH_FILE: block i = 64 KiB-class slice of that file at offset (i * 7919) % (len - block) */

static int cmp_d(const void* a, const void* b) { double x = *(const double*)a, y = *(const double*)b; return x < y ? -1 : x > y; }
static double median(double* v, int n) { qsort(v, (size_t)n, sizeof(double), cmp_d); return n & 1 ? v[n / 2] : 0.5 * (v[n / 2 - 1] + v[n / 2]); }
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static void do_block(int ph, long t, int i) {
  uint8_t* s = src + (size_t)i * block_size; uint8_t* c = comp ? comp + (size_t)i * bound : NULL; uint8_t* d = back ? back + (size_t)i * block_size : NULL;
  int r;
  switch (ph) {
    case 0: if (file_buf) memcpy(s, file_buf + ((uint64_t)(first_idx + i) * 7919u) % (uint64_t)(file_len - block_size), (size_t)block_size);
            else gen(s, block_size, 0x4C5A3447ull, first_idx + (uint64_t)i, litmax, win);
            break;
    case 1: r = hc_level ? r_hc((const char*)s, (char*)c, block_size, bound, hc_level)
                         : (is_ref ? r_c((const char*)s, (char*)c, block_size, bound) : p_c(s, block_size, c, bound));
            clen[i] = r; if (r <= 0) bad = 1; break;
    case 2: r = is_ref ? r_ds((const char*)c, (char*)d, clen[i], block_size) : p_ds(c, clen[i], d, block_size); if (r != block_size) bad = 1; break;
    case 3: r = is_ref ? r_df((const char*)c, (char*)d, block_size) : p_df(c, d, block_size); if (r != clen[i]) bad = 1; break;
    case 4: hsum[t] += x32(s, (size_t)block_size, 0x9747b28cu); break;
    case 5: hsum[t] += x64(s, (size_t)block_size, 0x9747b28cull); break;
  }
}

/* the pool: workers wait at `gate`, run the repetition main has set up (phase, passes), meet main again at `gate` */
static pthread_barrier_t gate;
static int n_passes, quit;
static struct Share { _Atomic long next; long total; int b0, len; char pad[40]; } share[256];   /* tickets: pass * len + (block - b0) */
static int cpus[1024], n_cpus;
/* one worker per block at a time: with stealing, a late ticket of pass p and a ticket of pass p + 1 can name the same block, and two
   decoders writing one output (wild copies run past the match end before the next sequence overwrites them) would read each other's
   scratch bytes */
static _Atomic unsigned char* busy;

static void run_share(int ph, long t, int v) {   /* blocks of worker v's share, taken ticket by ticket */
  struct Share* s = &share[v];
  if (!s->len) return;
  for (;;) {
    long k = atomic_fetch_add_explicit(&s->next, 1, memory_order_relaxed);
    if (k >= s->total) return;
    const int b = s->b0 + (int)(k % s->len);
    while (atomic_exchange_explicit(&busy[b], 1, memory_order_acquire)) __builtin_ia32_pause();
    do_block(ph, t, b);
    atomic_store_explicit(&busy[b], 0, memory_order_release);
  }
}
static void* worker(void* arg) {
  long t = (long)arg;
  if (n_cpus) { cpu_set_t set; CPU_ZERO(&set); CPU_SET(cpus[t % n_cpus], &set); (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set); }
  for (;;) {
    pthread_barrier_wait(&gate);
    if (quit) return NULL;
    run_share(phase, t, (int)t);
    /* phase 0 (generation = first touch) stays with the owner; the timed phases help whoever is behind */
    if (phase) for (int d = 1; d < n_threads; d++) run_share(phase, t, (int)((t + d) % n_threads));
    pthread_barrier_wait(&gate);
  }
}
static pthread_t pool[256];
static void pool_start(void) {
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0)
    for (int c = 0; c < CPU_SETSIZE && n_cpus < 1024; c++) if (CPU_ISSET(c, &set)) cpus[n_cpus++] = c;
  busy = calloc((size_t)n_blocks, 1);
  if (!busy) { fprintf(stderr, "malloc\n"); exit(5); }
  pthread_barrier_init(&gate, NULL, (unsigned)n_threads + 1u);
  for (long t = 0; t < n_threads; t++) pthread_create(&pool[t], NULL, worker, (void*)t);
}
static void pool_stop(void) {
  quit = 1;
  pthread_barrier_wait(&gate);
  for (int t = 0; t < n_threads; t++) pthread_join(pool[t], NULL);
}
/* one repetition: `passes` whole passes of phase ph over the sample, seconds PER PASS */
static double run_phase_n(int ph, int passes) {
  phase = ph; n_passes = passes;
  for (int t = 0; t < n_threads; t++) {
    int b0 = (int)((long long)n_blocks * t / n_threads), b1 = (int)((long long)n_blocks * (t + 1) / n_threads);
    share[t].b0 = b0; share[t].len = b1 - b0; share[t].total = (long)(b1 - b0) * passes;
    atomic_store_explicit(&share[t].next, 0, memory_order_relaxed);
  }
  double t0 = now();
  pthread_barrier_wait(&gate);
  pthread_barrier_wait(&gate);
  return (now() - t0) / passes;
}
static double run_phase(int ph) { return run_phase_n(ph, 1); }
static double min_ms = 300.0;
static int passes_for(double one_pass_s) {   /* whole passes a repetition needs to last min_ms */
  int k = (int)(min_ms * 1e-3 / (one_pass_s > 1e-6 ? one_pass_s : 1e-6)) + 1;
  return k < 1 ? 1 : (k > 100000 ? 100000 : k);
}

int main(int argc, char** argv) {
  if (argc < 10) { fprintf(stderr, "usage: cpu_bench <reference|port> <lib.so> <port.so> n_blocks block_size threads reps first_idx litmax win\n"); return 2; }
  is_ref = strcmp(argv[1], "reference") == 0;
  void* lib = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
  void* plib = dlopen(argv[3], RTLD_NOW | RTLD_LOCAL);
  if (!lib || !plib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
  n_blocks = atoi(argv[4]); block_size = atoi(argv[5]); n_threads = atoi(argv[6]);
  int reps = atoi(argv[7]); first_idx = strtoull(argv[8], 0, 10); litmax = (uint32_t)atoi(argv[9]); win = (uint32_t)atoi(argv[10]);
  if (n_threads > 256) n_threads = 256;
  gen = (gen_fn)dlsym(plib, "lz4o_gen_block");
  if (is_ref) {
    r_c = (compress_fn)dlsym(lib, "LZ4_compress_default"); r_ds = (dsafe_fn)dlsym(lib, "LZ4_decompress_safe"); r_df = (dfast_fn)dlsym(lib, "LZ4_decompress_fast");
    if (!r_c || !r_ds || !r_df) { fprintf(stderr, "missing LZ4_* symbols\n"); return 4; }
    if (getenv("LZ4_HC_LEVEL")) {
      hc_level = atoi(getenv("LZ4_HC_LEVEL"));
      r_hc = (compress_hc_fn)dlsym(lib, "LZ4_compress_HC");
      if (!r_hc || hc_level < 1 || hc_level > 12) { fprintf(stderr, "LZ4_compress_HC unavailable / bad level\n"); return 4; }
    }
  } else {
    p_c = (p_compress_fn)dlsym(lib, "lz4o_compress_fast"); p_ds = (p_dsafe_fn)dlsym(lib, "lz4o_decompress_safe"); p_df = (p_dfast_fn)dlsym(lib, "lz4o_decompress_fast");
    if (!p_c || !p_ds || !p_df) { fprintf(stderr, "missing lz4o_* symbols\n"); return 4; }
  }
  bound = block_size + block_size / 255 + 16;
  if (n_threads < 1) n_threads = 1;
  if (getenv("CPU_BENCH_MIN_MS")) min_ms = atof(getenv("CPU_BENCH_MIN_MS"));
  if (getenv("CPU_BENCH_FILE")) {
    FILE* f = fopen(getenv("CPU_BENCH_FILE"), "rb");
    if (!f) { fprintf(stderr, "cannot open CPU_BENCH_FILE\n"); return 6; }
    fseek(f, 0, SEEK_END); file_len = ftell(f); fseek(f, 0, SEEK_SET);
    file_buf = malloc((size_t)file_len);
    if (!file_buf || file_len <= block_size || fread(file_buf, 1, (size_t)file_len, f) != (size_t)file_len) { fprintf(stderr, "CPU_BENCH_FILE too short\n"); return 6; }
    fclose(f);
  }
  if (getenv("XXH_MODE")) {
    x32 = (xxh32_fn)dlsym(lib, is_ref ? "XXH32" : "lz4o_xxh32_raw"); x64 = (xxh64_fn)dlsym(lib, is_ref ? "XXH64" : "lz4o_xxh64_raw");
    if (!x32 || !x64) { fprintf(stderr, "missing XXH32/XXH64 symbols\n"); return 4; }
    src = malloc((size_t)n_blocks * block_size); hsum = calloc(256, sizeof(uint64_t));
    if (!src || !hsum) { fprintf(stderr, "malloc\n"); return 5; }
    pool_start();
    run_phase(0);
    double b32 = 1e30, b64 = 1e30, t32[64], t64[64];
    if (reps > 64) reps = 64;
    run_phase(4);
    const int k32 = passes_for(run_phase(4)), k64 = passes_for(run_phase(5));
    memset(hsum, 0, 256 * sizeof(uint64_t));
    run_phase(4); run_phase(5);   /* the check value: every hash exactly once */
    uint64_t chk = 0; for (int i = 0; i < 256; i++) chk += hsum[i];
    for (int r = 0; r < reps; r++) { double t = run_phase_n(4, k32); t32[r] = t; if (t < b32) b32 = t; t = run_phase_n(5, k64); t64[r] = t; if (t < b64) b64 = t; }
    pool_stop();
    double bytes = (double)n_blocks * block_size;
    printf("{\"kind\": \"%s\", \"threads\": %d, \"n_blocks\": %d, \"block_size\": %d, \"check\": %llu, \"xxh32_GBps\": %.4f, \"xxh64_GBps\": %.4f, "
           "\"xxh32_GBps_median\": %.4f, \"xxh64_GBps_median\": %.4f, \"passes\": [%d, %d], \"min_ms\": %.0f, \"pinned_cpus\": %d}\n",
           argv[1], n_threads, n_blocks, block_size, (unsigned long long)chk, bytes / b32 / 1e9, bytes / b64 / 1e9,
           bytes / median(t32, reps) / 1e9, bytes / median(t64, reps) / 1e9, k32, k64, min_ms, n_cpus);
    return 0;
  }
  src = malloc((size_t)n_blocks * block_size); comp = malloc((size_t)n_blocks * bound); back = malloc((size_t)n_blocks * block_size);
  clen = malloc(sizeof(int) * n_blocks);
  if (!src || !comp || !back || !clen) { fprintf(stderr, "malloc\n"); return 5; }
  pool_start();
  run_phase(0);
  double best[4] = {0, 1e30, 1e30, 1e30}, all[4][64];
  if (reps > 64) reps = 64;
  run_phase(1); run_phase(2); run_phase(3);  /* warm-up = first touch of comp / back by the owners */
  int k[4] = {1, 1, 1, 1};
  for (int ph = 1; ph <= 3; ph++) k[ph] = passes_for(run_phase(ph));
  for (int r = 0; r < reps; r++)
    for (int ph = 1; ph <= 3; ph++) { double t = run_phase_n(ph, k[ph]); all[ph][r] = t; if (t < best[ph]) best[ph] = t; }
  pool_stop();
  double med[4] = {0, median(all[1], reps), median(all[2], reps), median(all[3], reps)};
  if (memcmp(src, back, (size_t)n_blocks * block_size) != 0) bad = 1;
  long long csum = 0; for (int i = 0; i < n_blocks; i++) csum += clen[i];
  double bytes = (double)n_blocks * block_size;
  if (hc_level) printf("{\"hc_level\": %d, ", hc_level); else printf("{");
  printf("\"kind\": \"%s\", \"threads\": %d, \"n_blocks\": %d, \"block_size\": %d, \"ratio\": %.4f, \"ok\": %s, "
         "\"compress_GBps\": %.4f, \"decompress_safe_GBps\": %.4f, \"decompress_fast_GBps\": %.4f, \"roundtrip_GBps\": %.4f, "
         "\"compress_GBps_median\": %.4f, \"decompress_safe_GBps_median\": %.4f, \"decompress_fast_GBps_median\": %.4f, \"roundtrip_GBps_median\": %.4f, "
         "\"passes\": [%d, %d, %d], \"min_ms\": %.0f, \"pinned_cpus\": %d}\n",
         argv[1], n_threads, n_blocks, block_size, bytes / (double)csum, bad ? "false" : "true",
         bytes / best[1] / 1e9, bytes / best[2] / 1e9, bytes / best[3] / 1e9, bytes / (best[1] + best[2]) / 1e9,
         bytes / med[1] / 1e9, bytes / med[2] / 1e9, bytes / med[3] / 1e9, bytes / (med[1] + med[2]) / 1e9, k[1], k[2], k[3], min_ms, n_cpus);
  return bad ? 1 : 0;
}

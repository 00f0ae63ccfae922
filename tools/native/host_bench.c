/* ns per call of the reference-named functions on small host arrays: this library against the
 * reference itself (oracle/_ref/libstreamvbyte_ref.so, built from /root/reference by oracle/Makefile),
 * both loaded with dlopen so that their identical symbol names do not collide.
 *     host_bench <libstreamvbyte.so.2> [<libstreamvbyte_ref.so>]
 * With a GPU in the process the library probes what kind of memory every pointer is before it takes the
 * host codec; "probe off" repeats the measurement with svbgpu_set_option("host_pointers_only", 1). */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef size_t (*enc_fn)(const uint32_t*, uint32_t, uint8_t*);
typedef size_t (*dec_fn)(const uint8_t*, uint32_t*, uint32_t);
typedef size_t (*denc_fn)(const uint32_t*, uint32_t, uint8_t*, uint32_t);
typedef size_t (*ddec_fn)(const uint8_t*, uint32_t*, uint32_t, uint32_t);
typedef int (*opt_fn)(const char*, int);

static double now(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}
struct lib {
  enc_fn enc, enc0124;
  dec_fn dec, dec0124;
  denc_fn denc;
  ddec_fn ddec;
  opt_fn opt;
};
static int load(const char* path, struct lib* L) {
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); return 0; }
  L->enc = (enc_fn)dlsym(h, "streamvbyte_encode");
  L->dec = (dec_fn)dlsym(h, "streamvbyte_decode");
  L->enc0124 = (enc_fn)dlsym(h, "streamvbyte_encode_0124");
  L->dec0124 = (dec_fn)dlsym(h, "streamvbyte_decode_0124");
  L->denc = (denc_fn)dlsym(h, "streamvbyte_delta_encode");
  L->ddec = (ddec_fn)dlsym(h, "streamvbyte_delta_decode");
  L->opt = (opt_fn)dlsym(h, "svbgpu_set_option");
  return L->enc && L->dec && L->enc0124 && L->dec0124 && L->denc && L->ddec;
}
/* best of 5 runs of about 20 ms each, ns per call */
#define TIME(stmt, ns_out)                                                   \
  do {                                                                       \
    unsigned reps = 1;                                                       \
    for (;;) {                                                               \
      double t0 = now();                                                     \
      for (unsigned r = 0; r < reps; ++r) { stmt; }                          \
      double dt = now() - t0;                                                \
      if (dt > 0.004) { reps = (unsigned)(reps * 0.02 / dt) + 1; break; }    \
      reps *= 4;                                                             \
    }                                                                        \
    double best = 1e30;                                                      \
    for (int k = 0; k < 5; ++k) {                                            \
      double t0 = now();                                                     \
      for (unsigned r = 0; r < reps; ++r) { stmt; }                          \
      double dt = (now() - t0) / reps;                                       \
      if (dt < best) best = dt;                                              \
    }                                                                        \
    ns_out = best * 1e9;                                                     \
  } while (0)

static volatile size_t sink;
static void run(const char* name, struct lib* L, const uint32_t* v, uint32_t n, uint8_t* buf, uint32_t* back) {
  double e, d, e0, d0, de, dd;
  size_t s = L->enc(v, n, buf);
  TIME(sink = L->enc(v, n, buf), e);
  TIME(sink = L->dec(buf, back, n), d);
  if (s != sink || memcmp(v, back, 4 * (size_t)n)) { printf("MISMATCH 1234 n=%u\n", n); exit(1); }
  L->enc0124(v, n, buf);
  TIME(sink = L->enc0124(v, n, buf), e0);
  TIME(sink = L->dec0124(buf, back, n), d0);
  if (memcmp(v, back, 4 * (size_t)n)) { printf("MISMATCH 0124 n=%u\n", n); exit(1); }
  L->denc(v, n, buf, 7);
  TIME(sink = L->denc(v, n, buf, 7), de);
  TIME(sink = L->ddec(buf, back, n, 7), dd);
  if (memcmp(v, back, 4 * (size_t)n)) { printf("MISMATCH delta n=%u\n", n); exit(1); }
  printf("%-22s n=%6u  ns/call: encode %9.1f decode %9.1f | 0124 %9.1f %9.1f | delta %9.1f %9.1f   (Gint/s enc %.2f dec %.2f)\n",
         name, n, e, d, e0, d0, de, dd, n / e, n / d);
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: host_bench lib [reflib]\n"); return 2; }
  struct lib A, R;
  if (!load(argv[1], &A)) return 1;
  const int have_ref = argc > 2 && load(argv[2], &R);
  const uint32_t sizes[] = {4, 16, 64, 1024, 65535};
  const uint32_t N = 65535;
  uint32_t* v = malloc(4 * (size_t)N + 64);
  uint32_t* back = malloc(4 * (size_t)N + 64);
  uint8_t* buf = malloc(5 * (size_t)N + 64);
  srand(1);
  for (uint32_t k = 0; k < N; ++k) v[k] = (uint32_t)rand() >> (31u & (uint32_t)rand()); /* the shape of reference tests/perf.c:52-53 */
  for (unsigned i = 0; i < sizeof sizes / sizeof *sizes; ++i) {
    const uint32_t n = sizes[i];
    if (have_ref) run("reference", &R, v, n, buf, back);
    run("this library", &A, v, n, buf, back);
    if (A.opt && A.opt("host_pointers_only", 1) == 0) {
      run("this library, probe off", &A, v, n, buf, back);
      A.opt("host_pointers_only", 0);
    }
  }
  return 0;
}

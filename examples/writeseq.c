/* Write a sequence of integers to a file and read it back: the job of the
 * reference's tests/writeseq.c, which has to store the count next to the stream by
 * hand.  Frames (streamvbyte_frame.h) carry count and length themselves.  Build:
 *   gcc -O2 -Iinclude examples/writeseq.c -Lstreamvbyte_amd -l:libstreamvbyte.so.2 \
 *       -Wl,-rpath,$PWD/streamvbyte_amd -o writeseq
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "streamvbyte_frame.h"
#include "streamvbyte_gpu.h"

int main(int argc, char **argv) {
  const char *path = argc > 1 ? argv[1] : "/tmp/svb_writeseq.bin";
  const size_t n = 50000;
  const uint32_t frame_values = 16384;
  uint32_t *data = malloc(n * sizeof(uint32_t));
  for (size_t k = 0; k < n; ++k) data[k] = (uint32_t)(k * k);  /* sorted: delta coding pays */

  size_t cap = svbframe_max_bytes_many(n, frame_values), written = 0;
  uint8_t *frames = malloc(cap);
  int rc = svbframe_encode_many(data, n, frame_values, SVBGPU_1234, SVBGPU_DELTA, frames, cap, &written);
  if (rc) { fprintf(stderr, "encode: %d\n", rc); return 1; }
  FILE *f = fopen(path, "wb");
  if (!f || fwrite(frames, 1, written, f) != written || fclose(f)) { perror(path); return 1; }

  /* a reader that knows nothing but the file */
  f = fopen(path, "rb");
  if (!f) { perror(path); return 1; }
  fseek(f, 0, SEEK_END);
  const size_t bytes = (size_t)ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t *in = malloc(bytes);
  if (fread(in, 1, bytes, f) != bytes) { perror(path); return 1; }
  fclose(f);
  size_t nframes = 0, nvalues = 0;
  rc = svbframe_scan(in, bytes, &nframes, &nvalues);
  if (rc) { fprintf(stderr, "scan: %d\n", rc); return 1; }
  uint32_t *back = malloc(nvalues * sizeof(uint32_t));
  rc = svbframe_decode_many(in, bytes, back, nvalues, &nvalues);
  if (rc) { fprintf(stderr, "decode: %d\n", rc); return 1; }
  const int ok = nvalues == n && memcmp(back, data, n * sizeof(uint32_t)) == 0;
  printf("%zu integers -> %zu bytes in %zu frames -> %s\n", n, bytes, nframes, ok ? "read back ok" : "MISMATCH");
  remove(path);
  return ok ? 0 : 1;
}

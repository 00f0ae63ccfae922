/* A plain C caller of the reference API: compiled with gcc against include/ and
 * linked against libstreamvbyte.so.2, with no knowledge of HIP.  Exit 0 = ok. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "streamvbyte.h"
#include "streamvbyte_frame.h"
#include "streamvbyte_gpu.h"
#include "streamvbyte_zigzag.h"
#include "streamvbytedelta.h"

int main(void) {
  const uint32_t n = 100003;
  uint32_t *in = malloc(n * sizeof(uint32_t)), *back = malloc(n * sizeof(uint32_t));
  uint8_t *buf = malloc(streamvbyte_max_compressedbytes(n));
  uint32_t x = 12345;
  for (uint32_t i = 0; i < n; i++) {
    x = x * 1664525u + 1013904223u;
    in[i] = x >> (x & 31u);
  }
  size_t a = streamvbyte_encode(in, n, buf);
  if (a != streamvbyte_compressedbytes(in, n)) return 1;
  if (!streamvbyte_validate_stream(buf, a, n)) return 2;
  if (streamvbyte_decode(buf, back, n) != a || memcmp(in, back, n * 4)) return 3;
  a = streamvbyte_encode_0124(in, n, buf);
  if (a != streamvbyte_compressedbytes_0124(in, n)) return 4;
  if (!streamvbyte_validate_stream_0124(buf, a, n)) return 5;
  if (streamvbyte_decode_0124(buf, back, n) != a || memcmp(in, back, n * 4)) return 6;
  a = streamvbyte_delta_encode(in, n, buf, 77);
  if (streamvbyte_delta_decode(buf, back, n, 77) != a || memcmp(in, back, n * 4)) return 7;
  zigzag_encode((const int32_t *)in, back, n);
  zigzag_decode(back, (int32_t *)back, n);
  if (memcmp(in, back, n * 4)) return 8;
  zigzag_delta_encode((const int32_t *)in, back, n, -3);
  uint32_t *tmp = malloc(n * 4);
  zigzag_delta_decode(back, (int32_t *)tmp, n, -3);
  if (memcmp(in, tmp, n * 4)) return 9;
  /* frames: pageable host buffers, three frames, delta coded */
  size_t cap = svbframe_max_bytes_many(n, 40000), written = 0, nvalues = 0;
  uint8_t *frames = malloc(cap);
  if (svbframe_encode_many(in, n, 40000, SVBGPU_1234, SVBGPU_DELTA, frames, cap, &written) != SVBFRAME_OK) return 10;
  svbframe_info info;
  if (svbframe_peek(frames, written, &info) != SVBFRAME_OK || info.count != 40000 || info.flags != SVBGPU_DELTA) return 11;
  if (svbframe_decode_many(frames, written, NULL, 0, &nvalues) != SVBFRAME_OK || nvalues != n) return 12;
  memset(tmp, 0, n * 4);
  if (svbframe_decode_many(frames, written, tmp, n, &nvalues) != SVBFRAME_OK || memcmp(in, tmp, n * 4)) return 13;
  frames[3] ^= 0x40; /* magic */
  if (svbframe_decode_many(frames, written, tmp, n, &nvalues) != SVBFRAME_EBADFRAME) return 14;
  /* list frames: the same values as 1000 lists of growing length (the first one empty) */
  enum { NLISTS = 1000 };
  uint64_t start[NLISTS + 1], start_back[NLISTS + 1];
  for (uint32_t i = 0; i <= NLISTS; i++) start[i] = (uint64_t)n * i / NLISTS * i / NLISTS;
  size_t lcap = svbframe_lists_max_bytes(n, NLISTS, 30000), nlists = 0;
  uint8_t *lframes = malloc(lcap);
  if (svbframe_encode_lists(in, start, NLISTS, 30000, SVBGPU_0124, 0, lframes, lcap, &written) != SVBFRAME_OK) return 15;
  if (svbframe_decode_lists(lframes, written, NULL, 0, NULL, 0, &nlists, &nvalues) != SVBFRAME_OK || nlists != NLISTS ||
      nvalues != n)
    return 16;
  memset(tmp, 0, n * 4);
  if (svbframe_decode_lists(lframes, written, tmp, n, start_back, NLISTS, &nlists, &nvalues) != SVBFRAME_OK ||
      memcmp(in, tmp, n * 4) || memcmp(start, start_back, sizeof(start)))
    return 17;
  printf("drop-in round trips ok (%zu bytes)\n", a);
  return 0;
}

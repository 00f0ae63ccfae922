/* Smallest use of the reference API (what examples/example.c of lemire/streamvbyte
 * does): compress an array, decompress it, compare.  Build:
 *   gcc -O2 -Iinclude examples/example.c -Lstreamvbyte_amd -l:libstreamvbyte.so.2 \
 *       -Wl,-rpath,$PWD/streamvbyte_amd -o example
 */
#include <stdio.h>
#include <stdlib.h>

#include "streamvbyte.h"

int main(void) {
  const uint32_t n = 5000;
  uint32_t *data = malloc(n * sizeof(uint32_t));
  uint32_t *back = malloc(n * sizeof(uint32_t));
  uint8_t *compressed = malloc(streamvbyte_max_compressedbytes(n));
  for (uint32_t k = 0; k < n; ++k) data[k] = 100 + k * 7;
  const size_t size = streamvbyte_encode(data, n, compressed);
  const size_t used = streamvbyte_decode(compressed, back, n);
  int ok = used == size;
  for (uint32_t k = 0; k < n && ok; ++k) ok = back[k] == data[k];
  printf("Compressed %u integers down to %zu bytes: %s\n", n, size, ok ? "round trip ok" : "MISMATCH");
  free(data);
  free(back);
  free(compressed);
  return ok ? 0 : 1;
}

/*
 * streamvbytedelta.h -- differential StreamVByte (1234 container).
 * Link-compatible with reference include/streamvbytedelta.h.
 */
#ifndef INCLUDE_STREAMVBYTEDELTA_H_
#define INCLUDE_STREAMVBYTEDELTA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Encode in[i]-in[i-1] (uint32 wrap-around, in[-1] = prev).  Returns bytes
 * written; `out` must hold streamvbyte_max_compressedbytes(length).
 * (replaces reference include/streamvbytedelta.h:19) */
size_t streamvbyte_delta_encode(const uint32_t *in, uint32_t length, uint8_t *out, uint32_t prev);

/* Decode and rebuild values as a wrap-around running sum seeded with prev.
 * Returns bytes consumed.
 * (replaces reference include/streamvbytedelta.h:29) */
size_t streamvbyte_delta_decode(const uint8_t *in, uint32_t *out, uint32_t length, uint32_t prev);

#ifdef __cplusplus
}
#endif

#endif /* INCLUDE_STREAMVBYTEDELTA_H_ */

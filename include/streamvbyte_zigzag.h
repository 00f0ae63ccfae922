/*
 * streamvbyte_zigzag.h -- signed <-> unsigned zigzag maps.
 * Link-compatible with reference include/streamvbyte_zigzag.h.
 */
#ifndef INCLUDE_STREAMVBYTE_ZIGZAG_H_
#define INCLUDE_STREAMVBYTE_ZIGZAG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* out[i] = zigzag(in[i]).  (replaces reference include/streamvbyte_zigzag.h:14) */
void zigzag_encode(const int32_t *in, uint32_t *out, size_t N);

/* out[i] = zigzag(in[i] - in[i-1]), in[-1] = prev, wrap-around.
 * (replaces reference include/streamvbyte_zigzag.h:19) */
void zigzag_delta_encode(const int32_t *in, uint32_t *out, size_t N, int32_t prev);

/* out[i] = unzigzag(in[i]).  (replaces reference include/streamvbyte_zigzag.h:24) */
void zigzag_decode(const uint32_t *in, int32_t *out, size_t N);

/* out[i] = prev + sum_{j<=i} unzigzag(in[j]), wrap-around.
 * (replaces reference include/streamvbyte_zigzag.h:29) */
void zigzag_delta_decode(const uint32_t *in, int32_t *out, size_t N, int32_t prev);

#ifdef __cplusplus
}
#endif

#endif /* INCLUDE_STREAMVBYTE_ZIGZAG_H_ */

/*
 * svb_oracle.c -- CPU restatement of the StreamVByte stream format.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is the checker the GPU library is
 * compared against; it is never linked into, loaded by or called from the
 * product library (libstreamvbyte.so.2).  Only tests/, the smoke check and the
 * cpu_baseline leg of bench.py may load it.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks every function here
 * against the committed golden vectors in tests/golden/ (the reference's own
 * unit-test fixtures plus outputs of the reference library compiled from
 * /root/reference by oracle/Makefile), and, when oracle/_ref/ is present,
 * against that compiled reference directly on seeded random inputs.
 *
 * Written from the format rules, one plain scalar loop per operation; no
 * lookup tables, no SIMD.  Each function cites the reference lines whose
 * observable behaviour it restates (paths relative to /root/reference).
 *
 * Variants:  SVBO_1234 -> code c means c+1 data bytes
 *            SVBO_0124 -> code c means (1<<c)>>1 data bytes (0,1,2,4)
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define SVBO_1234 0
#define SVBO_0124 1

/* Width code of one value.
 * 1234: src/streamvbyte_encode.c:14-37 (thresholds 2^8, 2^16, 2^24).
 * 0124: src/streamvbyte_0124_encode.c:12-32 (0 -> code 0; anything >= 2^16 -> code 3). */
static inline uint32_t svbo_code(int variant, uint32_t v) {
    if (variant == SVBO_1234)
        return (uint32_t)(v > 0xFFu) + (uint32_t)(v > 0xFFFFu) + (uint32_t)(v > 0xFFFFFFu);
    return (uint32_t)(v != 0) + (uint32_t)(v > 0xFFu) + (uint32_t)(v > 0xFFFFu);
}

/* Data bytes stored for a code.
 * 1234: src/streamvbyte_decode.c:22-36.  0124: src/streamvbyte_0124_decode.c:210. */
static inline uint32_t svbo_len(int variant, uint32_t code) {
    return variant == SVBO_1234 ? code + 1u : ((1u << code) >> 1);
}

/* include/streamvbyte.h:33-39 */
size_t svbo_max_compressedbytes(uint32_t n) {
    return (size_t)((n + 3u) / 4u) + (size_t)n * 4u + 16u;
}

/* Exact stream size: control bytes + sum of widths.
 * src/streamvbyte_encode.c:69-108. */
size_t svbo_compressedbytes(int variant, const uint32_t *in, uint32_t n) {
    size_t total = ((size_t)n + 3u) / 4u;
    for (uint32_t i = 0; i < n; i++)
        total += svbo_len(variant, svbo_code(variant, in[i]));
    return total;
}

/* Encoder.  Layout: keyLen=(n+3)/4 control bytes, then the data bytes.
 * Control byte g holds the codes of values 4g..4g+3, value i in bits
 * 2(i&3)..2(i&3)+1; unused slots of the last control byte stay 0 and add no
 * data bytes.  delta!=0 stores in[i]-in[i-1] (uint32 wrap), in[-1]=prev.
 * src/streamvbyte_encode.c:39-62,113-137; src/streamvbyte_0124_encode.c:34-57,96-113;
 * src/streamvbytedelta_encode.c:39-76.
 * Writes exactly the returned number of bytes (the reference may scribble
 * past that inside max_compressedbytes; those bytes are not part of the stream). */
size_t svbo_encode(int variant, const uint32_t *in, uint32_t n, uint8_t *out,
                   int delta, uint32_t prev) {
    if (n == 0) return 0;
    size_t keylen = ((size_t)n + 3u) / 4u;
    uint8_t *data = out + keylen;
    memset(out, 0, keylen);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t v = in[i];
        if (delta) { uint32_t d = v - prev; prev = v; v = d; }
        uint32_t c = svbo_code(variant, v);
        uint32_t len = svbo_len(variant, c);
        out[i >> 2] |= (uint8_t)(c << (2u * (i & 3u)));
        for (uint32_t b = 0; b < len; b++)
            *data++ = (uint8_t)(v >> (8u * b));
    }
    return (size_t)(data - out);
}

/* Decoder.  Returns bytes consumed (= keyLen + data bytes).  delta!=0 adds a
 * running uint32 wrap-around sum seeded with prev.
 * src/streamvbyte_decode.c:18-87; src/streamvbyte_0124_decode.c:38-79,165-184;
 * src/streamvbytedelta_decode.c:34-74. */
size_t svbo_decode(int variant, const uint8_t *in, uint32_t *out, uint32_t n,
                   int delta, uint32_t prev) {
    if (n == 0) return 0;
    size_t keylen = ((size_t)n + 3u) / 4u;
    const uint8_t *data = in + keylen;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t c = (in[i >> 2] >> (2u * (i & 3u))) & 3u;
        uint32_t len = svbo_len(variant, c);
        uint32_t v = 0;
        for (uint32_t b = 0; b < len; b++)
            v |= (uint32_t)(*data++) << (8u * b);
        if (delta) { v += prev; prev = v; }
        out[i] = v;
    }
    return (size_t)(data - in);
}

/* Stream validation: true iff the control bytes of an n-value stream account
 * for exactly inlen - keyLen data bytes.  Empty rules first.
 * src/streamvbyte_decode.c:89-136; src/streamvbyte_0124_decode.c:186-229.
 * (The reference dereferences one control byte past the keys when n%4==0;
 * that read is not reproduced, only the result.) */
int svbo_validate(int variant, const uint8_t *in, size_t inlen, uint32_t n) {
    if (inlen == 0 || n == 0) return inlen == (size_t)n;
    uint64_t keylen = ((uint64_t)n + 3u) / 4u;
    if (keylen > inlen) return 0;
    uint64_t need = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t c = (in[i >> 2] >> (2u * (i & 3u))) & 3u;
        need += svbo_len(variant, c);
    }
    return need == (uint64_t)inlen - keylen;
}

/* Zigzag maps.  src/streamvbyte_zigzag.c:3-11,20-29 */
void svbo_zigzag_encode(const int32_t *in, uint32_t *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        uint32_t v = (uint32_t)in[i];
        out[i] = (v << 1) ^ (0u - (v >> 31));
    }
}

void svbo_zigzag_decode(const uint32_t *in, int32_t *out, size_t n) {
    for (size_t i = 0; i < n; i++)
        out[i] = (int32_t)((in[i] >> 1) ^ (0u - (in[i] & 1u)));
}

/* Zigzag of the wrap-around difference to the previous input.
 * src/streamvbyte_zigzag.c:13-18 */
void svbo_zigzag_delta_encode(const int32_t *in, uint32_t *out, size_t n, int32_t prev) {
    uint32_t p = (uint32_t)prev;
    for (size_t i = 0; i < n; i++) {
        uint32_t cur = (uint32_t)in[i];
        uint32_t d = cur - p;
        out[i] = (d << 1) ^ (0u - (d >> 31));
        p = cur;
    }
}

/* Running wrap-around sum of zigzag-decoded values.
 * src/streamvbyte_zigzag.c:32-38 */
void svbo_zigzag_delta_decode(const uint32_t *in, int32_t *out, size_t n, int32_t prev) {
    uint32_t p = (uint32_t)prev;
    for (size_t i = 0; i < n; i++) {
        p += (in[i] >> 1) ^ (0u - (in[i] & 1u));
        out[i] = (int32_t)p;
    }
}

/* ---- 64-bit values, the "1248" variant ----------------------------------------
 * No counterpart in the reference (its API is uint32_t only, include/streamvbyte.h:23):
 * PARITY UNPINNED for these two functions -- there is no reference output to pin them to.
 * They restate the format rule that include/streamvbyte_gpu.h gives for svbgpu_encode64 /
 * svbgpu_decode64: the reference's container (control bytes, then data bytes; four 2-bit
 * codes per control byte, value i in bits 2(i&3)..; unused slots 0 and without data) with
 * code c meaning 1 << c little-endian data bytes (1, 2, 4, 8), the smallest that hold the
 * value. */
static inline uint32_t svbo_code64(uint64_t v) {
    return (uint32_t)(v > 0xFFull) + (uint32_t)(v > 0xFFFFull) + (uint32_t)(v > 0xFFFFFFFFull);
}

size_t svbo_encode64(const uint64_t *in, uint32_t n, uint8_t *out) {
    if (n == 0) return 0;
    size_t keylen = ((size_t)n + 3u) / 4u;
    uint8_t *data = out + keylen;
    memset(out, 0, keylen);
    for (uint32_t i = 0; i < n; i++) {
        uint64_t v = in[i];
        uint32_t c = svbo_code64(v);
        out[i >> 2] |= (uint8_t)(c << (2u * (i & 3u)));
        for (uint32_t b = 0; b < (1u << c); b++)
            *data++ = (uint8_t)(v >> (8u * b));
    }
    return (size_t)(data - out);
}

size_t svbo_decode64(const uint8_t *in, uint64_t *out, uint32_t n) {
    if (n == 0) return 0;
    size_t keylen = ((size_t)n + 3u) / 4u;
    const uint8_t *data = in + keylen;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t c = (in[i >> 2] >> (2u * (i & 3u))) & 3u;
        uint64_t v = 0;
        for (uint32_t b = 0; b < (1u << c); b++)
            v |= (uint64_t)(*data++) << (8u * b);
        out[i] = v;
    }
    return (size_t)(data - in);
}

/*
 * streamvbyte.h -- StreamVByte codec, MI355X (gfx950) implementation.
 *
 * Source- and link-compatible with the header of the same name in
 * lemire/streamvbyte (reference: include/streamvbyte.h).  Every function here
 * keeps the reference's signature, return value and byte-exact stream format;
 * the work is done by HIP kernels in libstreamvbyte.so.2.
 *
 * Pointers may be host pointers or device pointers from hipMalloc.
 *   - Host pointers: calls of 65 536 values or more are staged through device
 *     memory and run on the GPU; smaller ones run on the calling thread (AVX-512,
 *     SSE4.1 / SSSE3 or scalar code inside the library, same bytes either way) because two PCIe
 *     crossings cost more than the work.  The threshold is the option
 *     "host_cpu_cutoff" of svbgpu_set_option (streamvbyte_gpu.h); 0 sends every
 *     call to the GPU.  Without a usable gfx950 device every call runs on the
 *     calling thread.  A HIP failure in the middle of a GPU call prints a message
 *     and abort()s: this API has no error channel.
 *   - Device pointers are used in place, no copies, always on the GPU.  The
 *     kernels run on a stream of the library's own; before they start the
 *     library waits for everything the caller has queued on the device
 *     (hipDeviceSynchronize on entry), and the call returns when the result is
 *     in memory, so a caller needs no synchronisation of its own on either side.
 *     Callers that want asynchronous calls on their own stream use
 *     streamvbyte_gpu.h.
 *   - Threads: any thread may call any function at any time on buffers of its
 *     own, as with the reference.  Calls that run on the calling thread are as
 *     independent as the reference's.  Calls that go to the GPU take turns: by
 *     default one host-pointer call at a time per device runs inside the library
 *     (a mutex; option "host_serialize").  That is a precaution, not a known
 *     necessity -- an abort seen once with eight threads in such calls was never
 *     explained or seen again (profiles/r3_thread_stress.txt in the source tree
 *     records the runs with the mutex off) -- and costs nothing measurable: such
 *     calls are bound by the PCIe link they share anyway.
 *   - What kind a pointer is, is asked of the HIP runtime per call (about 0.1 us
 *     per pointer; never for memory on the caller's own stack, never in a process
 *     without a device).  A program that only passes host memory can switch the
 *     question off: SVB_HOST_POINTERS_ONLY=1 in the environment, or option
 *     "host_pointers_only".
 *
 * Stream layout for `length` values: (length+3)/4 control bytes, then the data
 * bytes.  `length` itself is not stored.
 */
#ifndef INCLUDE_STREAMVBYTE_H_
#define INCLUDE_STREAMVBYTE_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Decoders may read, never use, this many bytes past the end of a stream;
 * encoders need this much slack after the worst-case stream.
 * (reference: include/streamvbyte.h:12) */
#define STREAMVBYTE_PADDING 16

/* Encode `length` uint32 from `in` into `out` using 1,2,3 or 4 data bytes per
 * value.  Returns bytes written.  `out` needs no alignment and must hold
 * streamvbyte_max_compressedbytes(length) bytes.
 * (replaces reference include/streamvbyte.h:23) */
size_t streamvbyte_encode(const uint32_t *in, uint32_t length, uint8_t *out);

/* As streamvbyte_encode with 0,1,2 or 4 data bytes per value.
 * (replaces reference include/streamvbyte.h:27) */
size_t streamvbyte_encode_0124(const uint32_t *in, uint32_t length, uint8_t *out);

/* Worst-case stream size including STREAMVBYTE_PADDING.
 * (the reference's bound, include/streamvbyte.h:33-39 there: control bytes +
 * four data bytes per value + padding) */
static inline size_t streamvbyte_max_compressedbytes(const uint32_t length) {
  return ((size_t)length + 3) / 4 + (size_t)length * sizeof(uint32_t) + STREAMVBYTE_PADDING;
}

/* Exact stream size streamvbyte_encode would produce (padding not included).
 * (replaces reference include/streamvbyte.h:47) */
size_t streamvbyte_compressedbytes(const uint32_t *in, uint32_t length);

/* Exact stream size streamvbyte_encode_0124 would produce.
 * (replaces reference include/streamvbyte.h:55) */
size_t streamvbyte_compressedbytes_0124(const uint32_t *in, uint32_t length);

/* Decode `length` values from `in` into `out`; returns bytes consumed.  `in`
 * needs no alignment and must be readable STREAMVBYTE_PADDING bytes past the
 * stream.  Writes exactly length*4 bytes.
 * (replaces reference include/streamvbyte.h:64) */
size_t streamvbyte_decode(const uint8_t *in, uint32_t *out, uint32_t length);

/* Decoder for streamvbyte_encode_0124 streams.
 * (replaces reference include/streamvbyte.h:68) */
size_t streamvbyte_decode_0124(const uint8_t *in, uint32_t *out, uint32_t length);

/* True iff the control bytes of an `outLength`-value stream account for exactly
 * `inLength` bytes.  Reads control bytes only.
 * (replaces reference include/streamvbyte.h:75) */
bool streamvbyte_validate_stream(const uint8_t *in, size_t inLength, uint32_t outLength);

/* Same for 0124 streams.  (replaces reference include/streamvbyte.h:79) */
bool streamvbyte_validate_stream_0124(const uint8_t *in, size_t inLength, uint32_t outLength);

#ifdef __cplusplus
}
#endif

#endif /* INCLUDE_STREAMVBYTE_H_ */

/*
 * streamvbyte_frame.h -- self-describing frames around StreamVByte streams and
 * a pipelined host-memory codec over sequences of frames.
 *
 * The reference format deliberately stores neither the value count nor the
 * stream length (README.md:200-205, include/streamvbyte.h:16-17 there): every
 * caller that writes streams to a file or a socket has to add them (its own
 * tests/writeseq.c:19-46 does so ad hoc).  A frame adds exactly that, and a
 * long array is cut into frames so that host-to-device copies, kernels and
 * device-to-host copies of neighbouring frames overlap.  There is no
 * counterpart in the reference API; the payload of every frame is a stream the
 * reference decodes (streamvbyte_decode / _0124 / delta_decode with the stored
 * count and prev).
 *
 * Frame layout, little endian, frames start at multiples of 16 bytes:
 *     0  u32  magic  'S' 'V' 'B' 'F'
 *     4  u8   version (1)
 *     5  u8   variant   SVBGPU_1234 | SVBGPU_0124
 *     6  u8   flags     SVBGPU_DELTA | SVBGPU_ZIGZAG
 *     7  u8   0
 *     8  u32  count     values in the frame
 *    12  u32  prev      seed of the differential coding (0 without DELTA)
 *    16  u64  stream_bytes
 *    24  u64  0
 *    32  the stream, then zero bytes up to the next multiple of 16
 *
 * All pointers are HOST pointers.  Page-locked buffers (svbframe_alloc_pinned,
 * hipHostMalloc, hipHostRegister) give overlapped copies at PCIe speed;
 * pageable buffers work too, copied through page-locked chunks of the library's
 * own (it never page-locks caller memory: pages are shared with other threads'
 * buffers).  Functions return SVBFRAME_OK, a negative SVBFRAME_E* code, or a
 * negative SVBGPU_E* code from streamvbyte_gpu.h; nothing is decoded on the CPU.
 */
#ifndef INCLUDE_STREAMVBYTE_FRAME_H_
#define INCLUDE_STREAMVBYTE_FRAME_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVBFRAME_HEADER_BYTES 32u

#define SVBFRAME_OK 0
#define SVBFRAME_ETRUNCATED (-11) /* buffer ends inside a frame                    */
#define SVBFRAME_EBADFRAME (-12)  /* magic, version, variant, flags or sizes wrong */
#define SVBFRAME_ECAPACITY (-13)  /* output buffer too small                       */
#define SVBFRAME_EINVALID (-14)   /* a payload fails streamvbyte_validate_stream   */

typedef struct svbframe_info {
  uint32_t count;
  uint32_t prev;
  uint8_t variant;
  uint8_t flags;
  uint64_t stream_bytes; /* payload */
  uint64_t frame_bytes;  /* header + payload + padding: offset of the next frame */
} svbframe_info;

/* Worst-case size of one frame of `count` values. */
size_t svbframe_max_bytes(uint32_t count);

/* Worst-case size of the frames svbframe_encode_many writes for n values. */
size_t svbframe_max_bytes_many(size_t n, uint32_t frame_values);

/* Parse the header at `frame` (`avail` readable bytes).  Host only. */
int svbframe_peek(const uint8_t *frame, size_t avail, svbframe_info *info);

/* Walk consecutive frames: number of frames and of values.  Host only. */
int svbframe_scan(const uint8_t *frames, size_t bytes, size_t *n_frames, size_t *n_values);

/* Encode n values as consecutive frames of at most frame_values values each
 * (the last one may be shorter).  With SVBGPU_DELTA the first frame is seeded
 * with 0 and every later one with the value before its first, so a frame
 * decodes on its own.  *written receives the bytes written. */
int svbframe_encode_many(const uint32_t *in, size_t n, uint32_t frame_values, int variant,
                         unsigned flags, uint8_t *out, size_t out_capacity, size_t *written);

/* Decode consecutive frames into out[0 .. *n_values).  Every payload is
 * validated against its header on the device before it is decoded; a frame
 * that fails stops the call with SVBFRAME_EINVALID.  With out == NULL only
 * *n_values is computed. */
int svbframe_decode_many(const uint8_t *frames, size_t bytes, uint32_t *out, size_t out_capacity,
                         size_t *n_values);

/* Page-locked host memory for the buffers above. */
void *svbframe_alloc_pinned(size_t bytes);
void svbframe_free_pinned(void *p);

#ifdef __cplusplus
}
#endif

#endif /* INCLUDE_STREAMVBYTE_FRAME_H_ */

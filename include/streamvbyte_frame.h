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

/*
 * List frames: many independent lists (posting lists, one stream each) per
 * frame, with a directory, for callers who would otherwise write one tiny
 * frame per list.  List i holds values[list_start[i] .. list_start[i+1]) and is
 * coded as a stream of its own (with SVBGPU_DELTA seeded with 0), exactly what
 * the reference's streamvbyte_encode / _0124 / delta_encode writes for it.
 *
 * Layout, little endian, every part starts at a multiple of 16 bytes:
 *     0  u32  magic  'S' 'V' 'B' 'L'
 *     4  u8   version (1)
 *     5  u8   variant
 *     6  u8   flags
 *     7  u8   0
 *     8  u32  n_lists
 *    12  u32  0
 *    16  u64  n_values      sum of the counts
 *    24  u64  payload_bytes sum of the sizes
 *    32  u32  count[n_lists]   values per list, zero padded to 16 bytes
 *        u32  size[n_lists]    stream bytes per list, zero padded to 16 bytes
 *        the streams, back to back in list order, zero padded to 16 bytes
 * A list holds at most SVBFRAME_LIST_MAX_VALUES values, a frame at most
 * SVBFRAME_MAX_LISTS lists.
 */
#define SVBFRAME_LIST_MAX_VALUES (1u << 29)
#define SVBFRAME_MAX_LISTS (1u << 20)

/* Worst-case size of what svbframe_encode_lists writes. */
size_t svbframe_lists_max_bytes(size_t n_values, size_t n_lists, uint32_t frame_values);

/* Encode n_lists lists as consecutive list frames.  A frame takes whole lists
 * while it holds at most frame_values values (a longer list gets a frame of its
 * own).  list_start has n_lists + 1 non-decreasing entries, list_start[0] is
 * the index of the first value in `values`. */
int svbframe_encode_lists(const uint32_t *values, const uint64_t *list_start, size_t n_lists,
                          uint32_t frame_values, int variant, unsigned flags, uint8_t *out,
                          size_t out_capacity, size_t *written);

/* Walk consecutive list frames (headers only): frames, lists, values.  Host only. */
int svbframe_scan_lists(const uint8_t *frames, size_t bytes, size_t *n_frames, size_t *n_lists,
                        size_t *n_values);

/* Decode consecutive list frames: values[0 .. *n_values) and list_start[0 ..
 * *n_lists] (list_start[0] = 0).  values_capacity counts values, lists_capacity
 * lists (list_start has room for lists_capacity + 1 entries).  The host checks
 * every directory, the device every list against its directory entry before
 * anything is decoded (SVBFRAME_EBADFRAME / SVBFRAME_EINVALID).  With values ==
 * NULL only *n_lists and *n_values are computed. */
int svbframe_decode_lists(const uint8_t *frames, size_t bytes, uint32_t *values,
                          size_t values_capacity, uint64_t *list_start, size_t lists_capacity,
                          size_t *n_lists, size_t *n_values);

/* Page-locked host memory for the buffers above. */
void *svbframe_alloc_pinned(size_t bytes);
void svbframe_free_pinned(void *p);

#ifdef __cplusplus
}
#endif

#endif /* INCLUDE_STREAMVBYTE_FRAME_H_ */

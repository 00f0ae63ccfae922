/*
 * streamvbyte_gpu.h -- device-pointer entry points of libstreamvbyte.so.2.
 *
 * These are the calls a GPU-resident caller binds instead of the host-pointer
 * functions of streamvbyte.h.  Plain C ABI: raw device pointers, sizes, an
 * opaque stream handle (a hipStream_t passed as void*, NULL = default stream).
 * All work is enqueued on `stream` and returns without synchronising; results
 * that the host-pointer API returns by value (bytes written / consumed, sizes,
 * validity) are written to device words the caller supplies.
 *
 * Every function returns SVBGPU_OK or a negative SVBGPU_E* code and never
 * falls back to a CPU implementation.
 *
 * Buffer rules are the reference's (include/streamvbyte.h:12-63 there):
 * encoder output holds streamvbyte_max_compressedbytes(count) bytes, decoder
 * input is readable 16 bytes past the stream, byte pointers need no alignment,
 * uint32 pointers are 4-byte aligned.
 */
#ifndef INCLUDE_STREAMVBYTE_GPU_H_
#define INCLUDE_STREAMVBYTE_GPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVBGPU_1234 0 /* code c -> c+1 data bytes        (streamvbyte_encode)      */
#define SVBGPU_0124 1 /* code c -> (1<<c)>>1 data bytes   (streamvbyte_encode_0124) */

#define SVBGPU_1248 2 /* 64-bit values, code c -> 1 << c data bytes: svbgpu_encode64 /
                         svbgpu_decode64 and svbgpu_validate only                    */

#define SVBGPU_DELTA 1u  /* differential coding, uint32 wrap-around, seeded with prev */
#define SVBGPU_ZIGZAG 2u /* encode: values are int32, zigzag-mapped before coding;
                            decode: un-zigzag after decoding (fuses zigzag_encode/
                            zigzag_decode into the codec pass)                      */
/* Both variants take every combination of the flags.  DELTA | ZIGZAG codes
 * zigzag(in[i] - in[i-1]), i.e. zigzag_delta_encode (reference
 * src/streamvbyte_zigzag.c:13-18) followed by the encoder, in one pass; the
 * reference itself offers delta only for 1234 (include/streamvbytedelta.h). */

#define SVBGPU_OK 0
#define SVBGPU_EINVAL (-1)     /* bad argument                                   */
#define SVBGPU_ENODEVICE (-2)  /* no usable gfx950 device                        */
#define SVBGPU_EWORKSPACE (-3) /* workspace missing or too small                 */
#define SVBGPU_EHIP (-4)       /* a HIP runtime call failed                      */
#define SVBGPU_ESCAN (-5)      /* device scan did not converge (reported by
                                  svbgpu_check_status)                           */

#define SVBGPU_ETOOBIG (-6)    /* count beyond SVBGPU_STATIC_MAX_COUNT
                                  (svbgpu_decode_static only)                    */

const char *svbgpu_strerror(int code);

/* 1 when the current HIP device is a gfx950 the library can run on. */
int svbgpu_device_ok(void);
/* 1 if this build of the library has the named diagnostic feature compiled in, else 0.
 * "lookback_stress": the build in which svbgpu_set_option("debug_ablate", 64) delays a random quarter
 * of the look-back's publications by tens of microseconds (libstreamvbyte_stress.so.2; results stay
 * exact, speed does not). */
int svbgpu_build_has(const char *feature);

/* Tuning knobs.  Each is an atomic word; every API call reads all of them once
 * on entry and works from that snapshot, so a change never reaches a call that
 * is already running and may be made from any thread.
 *   "encode_pipeline" = 0|1|2|3 3 (default): the encoder is chosen by the stream's
 *                               size (short streams and plain input of 2^25..2^27
 *                               values the software-pipelined kernel, in which a
 *                               tile's loads and look-back overlap the previous
 *                               tile's stores; the rest one tile at a time, a
 *                               workgroup per tile); 0: every encode one tile at a
 *                               time; 1: plain and zigzag encodes pipelined,
 *                               differential ones not; 2: every encode pipelined
 *   "encode_fresh" = 0|1        tile-at-a-time encoder: 1 (default) a workgroup per
 *                               tile, 0 persistent workgroups that loop over tiles
 *   "encode_waves" = 4|8|16     tile-at-a-time encoder when "encode_pipeline" is not
 *                               3: waves per workgroup (a tile is 4096 values a wave)
 *   "encode_ticket_early" = -1|0|1  pipelined encoder: 1 asks for a tile's place in
 *                               the stream one store phase before the tile is
 *                               loaded, 0 when it is loaded; -1 (default): 1 for
 *                               0124 (+4 %), 0 for 1234 (1 measured slower)
 *   "nt_threshold_mib" = n      calls that write at least n MiB (default 128, half
 *                               the 256 MiB Infinity Cache) store non-temporally;
 *                               calls that read at least n MiB and touch at least
 *                               2.5 n MiB load non-temporally (such a stream only
 *                               sweeps through the cache and evicts what the next
 *                               call would have found there); -1: never, 0: always.
 *                               Batched calls apply the rule on the device to the
 *                               extent of the value array between the first and the
 *                               last directory entry (the host does not see their
 *                               sizes)
 *   "nt_force" = -1 | 0..3      -1 (default): by size as above; else bit 0 = loads,
 *                               bit 1 = stores non-temporal whatever the size
 *   "encode_tile" = 1 | 2 | 4   wave tiles (1024 values) each wavefront of the
 *                               one-tile-at-a-time encoder carries through the
 *                               look-back (default 4)
 *   "encode64_waves" = 4|8|16   wavefronts (one wave tile each) per workgroup of
 *                               the 64-bit encoder (default 8)
 *   "scan_waves" = 4 | 8 | 16   wavefronts per workgroup of the running sum
 *                               behind svbgpu_zigzag_delta_decode and of the
 *                               single-pass differential decoder (default 16:
 *                               a tile of 65 536 values per look-back; streams
 *                               below 3 * 2^22 values are decoded with 8)
 *   "scan_fresh" = 0|1          running sum: 0 (default) persistent workgroups that
 *                               loop over tiles, 1 a workgroup per tile (slower)
 *   "decode_shortcut" = 0|1     1 (default): the plain decoder copies wave tiles
 *                               (1024 values) whose keys are all zero straight
 *                               from their data bytes; 0: one code path for all
 *   "delta_decode_lookback" = 0|1  1 (default): differential decode in a single
 *                               pass with look-back; 0: two streaming passes
 *                               (value sums first), no waiting between
 *                               workgroups
 *   "decode_grid_per_cu" = k    decoders: 0 (default) a workgroup per 4096
 *                               values, handed out in order by the hardware;
 *                               k > 0: a persistent grid of k workgroups per CU
 *                               with the wave tiles strided over it (slower:
 *                               kept for comparison)
 *   "batch_chunk" = k           batched calls: -1 (default) every wavefront takes
 *                               k = 16 (decode) or 8 (encode) consecutive streams,
 *                               short ones packed a group of four values per lane
 *                               across stream borders, long ones a wave tile at a
 *                               time; k in 1..64 sets the number; 0: the older
 *                               pair of launches (a 16-lane row per short stream,
 *                               a wavefront per long one)
 *   "batch_lists_per_wave" = k  only with "batch_chunk" 0: a workgroup per 4k
 *                               streams (default 8); 0: one grid that fills the
 *                               device, streams strided over it
 *   "host_pointers_only" = 1    the caller promises that the reference-named
 *                               functions only ever get host memory: they then
 *                               skip asking the HIP runtime what kind of pointer
 *                               each argument is (about 0.1 us per pointer, more
 *                               than a hundred values take to code; arguments on
 *                               the calling thread's stack are never asked about);
 *                               default 0.  SVB_HOST_POINTERS_ONLY=1 in the
 *                               environment does the same for a program that
 *                               cannot be changed
 *   "host_cpu_cutoff" = n       host-pointer calls of the reference-named
 *                               functions with fewer than n values (default
 *                               65 536) run on the calling thread (SSE4.1 / SSSE3
 *                               code in the library) instead of crossing PCIe;
 *                               0: always the GPU.  Never applies to device
 *                               pointers.
 *   "host_serialize" = 0|1      1 (default): host-pointer and frame calls on one
 *                               device take turns (a mutex per device); 0: they
 *                               run concurrently, each on its own context
 *   "host_stage_pageable" = 0|1 1 (default): pageable caller memory moves through
 *                               the library's page-locked chunks; 0: it is handed
 *                               to hipMemcpyAsync directly
 *   "zigzag_piece" = n          values per device scan of the host-pointer
 *                               zigzag_delta_decode (default 2^31; longer arrays
 *                               go in pieces)
 *   "debug_ablate" = bits       timing-only diagnostics, results become wrong
 * Apart from "debug_ablate", results never depend on a knob, only speed. */
int svbgpu_set_option(const char *name, int value);

/* Device scratch any single-stream call below needs for `count` values.
 * The workspace must be 256-byte aligned; its contents are undefined between
 * calls, and two calls that share a workspace must be ordered on one stream. */
size_t svbgpu_workspace_bytes(uint32_t count);

/* Optional, once per workspace (no counterpart in the reference, which needs no scratch): zeroes the
 * workspace on `stream` and remembers its address.  Calls that chain tiles through look-back words
 * (svbgpu_encode, svbgpu_quantize_encode, svbgpu_encode64, svbgpu_zigzag_delta_decode) and
 * svbgpu_validate then skip the clearing launch they otherwise start with (4-5 us a call, 5 % of an
 * encode of 2^26 values): every such call marks the words it writes with a tag of its own, so what
 * earlier calls left behind reads as empty.  Rules: the workspace must not be written by anything
 * else between calls; call svbgpu_workspace_release before freeing it (an address that comes back from
 * the allocator with other contents must not still be remembered); after a call that ended in
 * SVBGPU_ESCAN initialise again.  Calls recorded into a HIP graph keep the clearing launch (a replay
 * would repeat the tag).  Without it everything works as before. */
int svbgpu_workspace_init(void *d_workspace, size_t workspace_bytes, void *stream);
void svbgpu_workspace_release(void *d_workspace);

/* After a stream synchronisation: SVBGPU_OK, or SVBGPU_ESCAN if a kernel that
 * used this workspace gave up waiting on another workgroup (never observed in
 * a correct run; bounded so that a bug cannot hang the device). */
int svbgpu_check_status(const void *d_workspace, void *stream);

/* Device-pointer form of streamvbyte_decode / _decode_0124 /
 * streamvbyte_delta_decode (reference include/streamvbyte.h:64,68,
 * include/streamvbytedelta.h:29).  d_consumed (optional) receives the bytes
 * consumed = the reference's return value. */
int svbgpu_decode(int variant, unsigned flags, const uint8_t *d_in, uint32_t *d_out,
                  uint32_t count, uint32_t prev, void *d_workspace, size_t workspace_bytes,
                  uint64_t *d_consumed, void *stream);

/* The two halves of svbgpu_decode, for callers that decode one stream more
 * than once (cached posting lists) or want the phases on different streams:
 * _plan reads only the control bytes and leaves the data-offset directory of
 * the stream in the workspace; _run decodes with that directory.  The
 * workspace must not be used by another call in between. */
int svbgpu_decode_plan(int variant, const uint8_t *d_in, uint32_t count, void *d_workspace,
                       size_t workspace_bytes, void *stream);
int svbgpu_decode_run(int variant, unsigned flags, const uint8_t *d_in, uint32_t *d_out,
                      uint32_t count, uint32_t prev, void *d_workspace, size_t workspace_bytes,
                      uint64_t *d_consumed, void *stream);

/* svbgpu_decode without a workspace argument, for callers that cannot
 * allocate one: code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  No counterpart in the reference (its decoder needs no
 * scratch at all).  Same kernels, same results and same d_consumed as
 * svbgpu_decode; the data-offset directory and the look-back words live in a
 * static device array inside the library (about 2.9 MB per GPU, held until the
 * process ends), so the call allocates nothing -- neither device nor host
 * memory -- and only enqueues kernels on `stream`.
 *   - count > SVBGPU_STATIC_MAX_COUNT is refused with SVBGPU_ETOOBIG (nothing
 *     is enqueued); use svbgpu_decode with a workspace of your own.
 *   - Every call on one GPU shares that GPU's array: calls must be ordered
 *     (same stream, or events between streams); two that overlap race on the
 *     directory and give wrong values.  Different GPUs never interfere.
 *   - svbgpu_check_status does not apply (there is no workspace pointer to
 *     give it); the look-back's bounded spin still guarantees termination, and
 *     svbgpu_static_status reports whether one gave up.
 *   - The first call on a device looks the array's address up
 *     (hipGetSymbolAddress); make that call before starting a capture. */
#define SVBGPU_STATIC_MAX_COUNT (1u << 28)
int svbgpu_decode_static(int variant, unsigned flags, const uint8_t *d_in, uint32_t *d_out,
                         uint32_t count, uint32_t prev, uint64_t *d_consumed, void *stream);
/* The error word of this GPU's static scratch, for callers of svbgpu_decode_static (no counterpart
 * in the reference; like that call it allocates nothing and only enqueues a kernel): *d_error
 * (a device word) becomes 0, or non-zero if a bounded spin of the last svbgpu_decode_static call
 * queued before it on `stream` gave up (its output is then wrong: decode again with
 * svbgpu_set_option("delta_decode_lookback", 0), which waits on nothing).  Read the word after
 * synchronising the stream. */
int svbgpu_static_status(uint32_t *d_error, void *stream);

/* Device-pointer form of streamvbyte_encode / _encode_0124 /
 * streamvbyte_delta_encode (reference include/streamvbyte.h:23,27,
 * include/streamvbytedelta.h:19).  d_written (optional) receives the bytes
 * written = the reference's return value.  Bytes of d_out past that count and
 * inside streamvbyte_max_compressedbytes(count) are unspecified. */
int svbgpu_encode(int variant, unsigned flags, const uint32_t *d_in, uint32_t count,
                  uint8_t *d_out, uint32_t prev, void *d_workspace, size_t workspace_bytes,
                  uint64_t *d_written, void *stream);

/* Quantised floats (gradient compression: quantise -> zigzag -> encode, and back).
 * No counterpart in the reference, whose callers run these steps as separate
 * passes over memory around zigzag_encode / streamvbyte_encode_0124
 * (src/streamvbyte_zigzag.c:3-11, src/streamvbyte_0124_encode.c:96-113); here
 * they ride on the codec's own load and store.
 *   svbgpu_quantize_encode:    q[i] = (int32) nearest-even(d_in[i] * scale), then
 *                              exactly svbgpu_encode(variant, SVBGPU_ZIGZAG, q, ...)
 *   svbgpu_decode_dequantize:  q = svbgpu_decode(variant, SVBGPU_ZIGZAG, ...), then
 *                              d_out[i] = (float)q[i] * scale, or d_out[i] += ... when
 *                              accumulate != 0 (summing several workers' chunks)
 * Products beyond the int32 range saturate. */
int svbgpu_quantize_encode(int variant, const float *d_in, uint32_t count, float scale,
                           uint8_t *d_out, void *d_workspace, size_t workspace_bytes,
                           uint64_t *d_written, void *stream);
int svbgpu_decode_dequantize(int variant, const uint8_t *d_in, float *d_out, uint32_t count,
                             float scale, int accumulate, void *d_workspace,
                             size_t workspace_bytes, uint64_t *d_consumed, void *stream);

/* Device-pointer form of streamvbyte_compressedbytes / _0124 (reference
 * include/streamvbyte.h:47,55); with SVBGPU_DELTA / SVBGPU_ZIGZAG the size the
 * matching svbgpu_encode call would produce.  *d_bytes receives the size. */
int svbgpu_compressedbytes(int variant, unsigned flags, const uint32_t *d_in, uint32_t count,
                           uint32_t prev, uint64_t *d_bytes, void *stream);

/* Device-pointer form of streamvbyte_validate_stream / _0124 (reference
 * include/streamvbyte.h:75,79).  *d_ok receives 1 or 0.  Reads only the
 * (count+3)/4 control bytes, and none when they do not fit in in_length.
 * Needs 256 bytes of workspace. */
int svbgpu_validate(int variant, const uint8_t *d_in, uint64_t in_length, uint32_t count,
                    uint32_t *d_ok, void *d_workspace, size_t workspace_bytes, void *stream);

/* Device-pointer forms of the zigzag maps (reference
 * include/streamvbyte_zigzag.h:14,19,24,29).  The delta decode needs a
 * workspace of svbgpu_workspace_bytes(min(n, UINT32_MAX)). */
int svbgpu_zigzag_encode(const int32_t *d_in, uint32_t *d_out, size_t n, void *stream);
int svbgpu_zigzag_decode(const uint32_t *d_in, int32_t *d_out, size_t n, void *stream);
int svbgpu_zigzag_delta_encode(const int32_t *d_in, uint32_t *d_out, size_t n, int32_t prev,
                               void *stream);
int svbgpu_zigzag_delta_decode(const uint32_t *d_in, int32_t *d_out, size_t n, int32_t prev,
                               void *d_workspace, size_t workspace_bytes, void *stream);

/*
 * Differential decode fused with an intersection.  The stream at d_in holds
 * `count` values coded as streamvbyte_delta_encode / svbgpu_encode with
 * SVBGPU_DELTA writes them (reference include/streamvbytedelta.h:19,29);
 * d_sorted[0 .. n_sorted) is ascending.  d_out receives, in stream order,
 * every decoded value that occurs in d_sorted (needs room for `count` values:
 * a stream may repeat a value), *d_n_out their number, *d_consumed (optional)
 * the stream's bytes.  The decoded array itself is never written.  No
 * counterpart in the reference, whose callers decode and then intersect.
 * Meant for ascending streams (posting lists); any stream gives the right
 * answer, but one whose values jump about makes every wavefront walk much of
 * d_sorted.  Workspace as for svbgpu_decode.
 */
int svbgpu_decode_intersect(int variant, const uint8_t *d_in, uint32_t count, uint32_t prev,
                            const uint32_t *d_sorted, uint32_t n_sorted, uint32_t *d_out,
                            uint64_t *d_n_out, void *d_workspace, size_t workspace_bytes,
                            uint64_t *d_consumed, void *stream);

/*
 * Two delta-coded streams intersected: d_out receives, in A's order, every value of
 * stream A (count_a values, seed prev_a, coded as streamvbyte_delta_encode writes them,
 * reference include/streamvbytedelta.h:19) that occurs in stream B; *d_n_out their
 * number.  Neither stream is written out decoded: B's per-1024-value running sums serve
 * as a skip list, and only the parts of B that a piece of A can meet are decoded, into
 * on-chip memory.  B must ascend without wrapping past 2^32 (a posting list); A may be
 * any stream (one that ascends is fastest).  d_out needs room for count_a values.  Both
 * streams readable STREAMVBYTE_PADDING bytes past their end.  Workspace:
 * svbgpu_intersect_workspace_bytes(count_a, count_b), 256-byte aligned.  No counterpart
 * in the reference, whose callers decode both lists and then intersect.
 */
size_t svbgpu_intersect_workspace_bytes(uint32_t count_a, uint32_t count_b);
int svbgpu_intersect_streams(int variant_a, const uint8_t *d_a, uint32_t count_a, uint32_t prev_a,
                             int variant_b, const uint8_t *d_b, uint32_t count_b, uint32_t prev_b,
                             uint32_t *d_out, uint64_t *d_n_out, void *d_workspace,
                             size_t workspace_bytes, void *stream);

/*
 * Many pairs of delta-coded lists intersected in one launch (a query batch against an inverted
 * index).  Pair i: list A_i = d_counts_a[i] values coded at d_bytes_a + d_byte_off_a[i], seed
 * d_prevs_a[i] (0 for all when d_prevs_a is NULL); list B_i likewise.  The values of A_i that occur in
 * B_i go, in A_i's order, to d_out + d_out_off[i] (room for d_counts_a[i] values), their number to
 * d_n_out[i].  Every B_i must ascend without wrapping past 2^32; an A_i that steps back is answered
 * correctly but makes its wavefront walk B_i again.  Both byte buffers readable STREAMVBYTE_PADDING
 * bytes past their last stream.  One wavefront per pair: meant for lists of up to some tens of
 * thousands of values (svbgpu_intersect_streams is the call for two long lists).  No workspace.  No
 * counterpart in the reference.
 */
int svbgpu_intersect_batch(int variant, const uint8_t *d_bytes_a, const uint64_t *d_byte_off_a,
                           const uint32_t *d_counts_a, const uint32_t *d_prevs_a,
                           const uint8_t *d_bytes_b, const uint64_t *d_byte_off_b,
                           const uint32_t *d_counts_b, const uint32_t *d_prevs_b, uint32_t npairs,
                           uint32_t *d_out, const uint64_t *d_out_off, uint32_t *d_n_out, void *stream);

/*
 * 64-bit values ("1248").  No counterpart in the reference, whose values are
 * uint32_t (include/streamvbyte.h:23 there); the container is the reference's:
 * (count + 3) / 4 control bytes, then the data bytes; four 2-bit codes per
 * control byte, value i of a group in bits 2i..2i+1, unused slots of the last
 * control byte 0 and without data.  Code c stands for 1 << c little-endian
 * data bytes (1, 2, 4, 8) -- the smallest of these that holds the value.
 * Encoder output holds svbgpu_max_compressedbytes64(count) bytes; decoder input
 * is readable SVBGPU_PADDING64 bytes past the stream.  Workspace as for the
 * 32-bit calls (svbgpu_workspace_bytes(count)).  svbgpu_validate(SVBGPU_1248, ..)
 * checks such a stream's control bytes against its length.
 */
#define SVBGPU_PADDING64 32
static inline size_t svbgpu_max_compressedbytes64(uint32_t count) {
  return ((size_t)count + 3) / 4 + 8 * (size_t)count + SVBGPU_PADDING64;
}
/* flags: SVBGPU_DELTA codes value - previous value (mod 2^64, seed `prev`) and decodes by running
 * sums; SVBGPU_ZIGZAG maps through (v << 1) ^ (0 - (v >> 63)) before coding and back after decoding;
 * both together code the zigzag of the difference.  The encoder is one pass over the values (tile
 * offsets by decoupled look-back, as for 32-bit values); the differential decoder reads the stream
 * twice (value sums per 4096 values first). */
int svbgpu_encode64(unsigned flags, const uint64_t *d_in, uint32_t count, uint8_t *d_out, uint64_t prev,
                    void *d_workspace, size_t workspace_bytes, uint64_t *d_written, void *stream);
int svbgpu_decode64(unsigned flags, const uint8_t *d_in, uint64_t *d_out, uint32_t count, uint64_t prev,
                    void *d_workspace, size_t workspace_bytes, uint64_t *d_consumed, void *stream);
/* Batched 64-bit calls: `nstreams` independent 1248 streams in one launch, arguments as for
 * svbgpu_encode_batch / svbgpu_decode_batch below (offsets in elements resp. bytes, d_prev may be NULL,
 * d_sizes[s] = bytes written / consumed; optional for decode).  For encode, d_byte_off[s] must leave
 * svbgpu_max_compressedbytes64(d_counts[s]) - SVBGPU_PADDING64 bytes for stream s; for decode, the last
 * stream of the buffer must be followed by SVBGPU_PADDING64 readable bytes (a stream inside the buffer is
 * followed by its neighbour, which serves).  A wavefront per stream, a wave tile (1024 values) at a
 * time: made for many streams of 10^3..10^5 values; streams of a few values leave most lanes idle.
 * No counterpart in the reference. */
int svbgpu_encode64_batch(unsigned flags, const uint64_t *d_values, const uint64_t *d_value_off, uint8_t *d_bytes,
                          const uint64_t *d_byte_off, const uint32_t *d_counts, const uint64_t *d_prev,
                          uint32_t nstreams, uint64_t *d_sizes, void *stream);
int svbgpu_decode64_batch(unsigned flags, const uint8_t *d_bytes, const uint64_t *d_byte_off, uint64_t *d_values,
                          const uint64_t *d_value_off, const uint32_t *d_counts, const uint64_t *d_prev,
                          uint32_t nstreams, uint64_t *d_sizes, void *stream);

/*
 * Batched entry points: `nstreams` independent streams in one launch (posting
 * lists, gradient chunks).  No counterpart in the reference, whose callers loop
 * over streamvbyte_encode/decode one stream at a time (e.g. reference
 * tests/unit.c:99-173).  Stream s has d_counts[s] values, seed d_prev[s]
 * (d_prev may be NULL = all zero; ignored without SVBGPU_DELTA), its integers
 * at d_values + d_value_off[s] (offset in elements) and its bytes at
 * d_bytes + d_byte_off[s].  For encode, d_byte_off[s] must leave
 * streamvbyte_max_compressedbytes(d_counts[s]) bytes for stream s and
 * d_sizes[s] receives the bytes written; for decode d_sizes[s] (optional)
 * receives the bytes consumed.
 * A wavefront takes a run of consecutive streams: those of up to 256 values
 * are packed a group of four values per lane across stream borders, longer
 * ones go a wave tile (1024 values) at a time.  That suits streams of up to
 * some 10^5 values; give longer ones to the single-stream calls above, which
 * spread one stream over the whole device.  d_value_off may be anything:
 * the decoder stores the whole tiles of a long stream as 64-byte-aligned
 * blocks wherever its values start; streams that start on such a boundary
 * themselves (offsets that are multiples of 16) still decode about 2 % faster.
 */
int svbgpu_encode_batch(int variant, unsigned flags, const uint32_t *d_values,
                        const uint64_t *d_value_off, uint8_t *d_bytes, const uint64_t *d_byte_off,
                        const uint32_t *d_counts, const uint32_t *d_prev, uint32_t nstreams,
                        uint64_t *d_sizes, void *stream);
int svbgpu_decode_batch(int variant, unsigned flags, const uint8_t *d_bytes,
                        const uint64_t *d_byte_off, uint32_t *d_values, const uint64_t *d_value_off,
                        const uint32_t *d_counts, const uint32_t *d_prev, uint32_t nstreams,
                        uint64_t *d_sizes, void *stream);

/* streamvbyte_compressedbytes / _0124 (reference include/streamvbyte.h:47,55) for every stream of a
 * batch: d_sizes[s] = the exact number of bytes svbgpu_encode_batch with the same arguments writes for
 * stream s, so that a caller can place the streams back to back before encoding them (an exclusive
 * scan of d_sizes gives d_byte_off).  Reads 4 bytes per value, writes nothing else. */
int svbgpu_compressedbytes_batch(int variant, unsigned flags, const uint32_t *d_values,
                                 const uint64_t *d_value_off, const uint32_t *d_counts,
                                 const uint32_t *d_prev, uint32_t nstreams, uint64_t *d_sizes,
                                 void *stream);

/* streamvbyte_validate_stream / _0124 (reference include/streamvbyte.h:75,79)
 * for every stream of a batch: d_ok[s] = 1 when the first d_counts[s] codes of
 * the stream at d_bytes + d_byte_off[s] promise exactly d_lengths[s] bytes of
 * keys and data together, else 0 (an empty stream is valid only with length 0,
 * as in the reference).  Reads nothing beyond a stream's key bytes. */
int svbgpu_validate_batch(int variant, const uint8_t *d_bytes, const uint64_t *d_byte_off,
                          const uint64_t *d_lengths, const uint32_t *d_counts, uint32_t nstreams,
                          uint32_t *d_ok, void *stream);

#ifdef __cplusplus
}
#endif

#endif /* INCLUDE_STREAMVBYTE_GPU_H_ */

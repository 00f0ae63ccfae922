/*
 * StreamVByte for MI355X -- batches of streams over the GPUs of one node.
 *
 * No counterpart in the reference, which is single-threaded and codes one stream per call
 * (reference src/streamvbyte_encode.c:113-137, src/streamvbyte_decode.c:64-87).  Streams are
 * independent, so a batch shards at stream granularity: every device codes a contiguous range of
 * whole streams with the batched kernels of streamvbyte_gpu.h, and nothing of the payload ever
 * crosses between devices.  The one exchange is the per-stream compressed sizes after an encode
 * (8 bytes per stream), all-gathered over RCCL so that every device can lay out the job's byte
 * directory.  A single stream is never split across devices.
 *
 * One process drives all devices (one HIP stream and one RCCL communicator per device, made by
 * ncclCommInitAll).  RCCL (librccl.so.1) is loaded when the first context is created, not with this
 * library.  For one process per GPU use the svbgpu_* calls per rank and all-gather the sizes with
 * the rank's own communicator (streamvbyte_amd/sharding.py does that over torch.distributed).
 */
#ifndef INCLUDE_STREAMVBYTE_MULTI_H_
#define INCLUDE_STREAMVBYTE_MULTI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVBMULTI_MAX_DEVICES 64
#define SVBMULTI_ERCCL (-7) /* an RCCL call failed or librccl.so.1 could not be loaded: svbmulti_last_error() */

/* Cut streams 0..nstreams-1 into `world` contiguous ranges of near-equal total value count:
 * shard r owns streams cut[r] .. cut[r+1]-1 (cut has world + 1 entries, cut[0] = 0,
 * cut[world] = nstreams).  Balances integers, not streams: lists differ in length by orders of
 * magnitude.  Plain host code, no device needed. */
int svbgpu_partition_streams(const uint32_t *counts, uint64_t nstreams, int world, uint64_t *cut);

typedef struct svbmulti svbmulti_t;

/* devices: HIP device ordinals, ndev of them (1 .. SVBMULTI_MAX_DEVICES), each a usable gfx950.
 * Creates a stream per device and the RCCL communicators.  SVBGPU_ENODEVICE / SVBGPU_EHIP /
 * SVBMULTI_ERCCL on failure (*out stays NULL). */
int svbmulti_create(const int *devices, int ndev, svbmulti_t **out);
void svbmulti_destroy(svbmulti_t *m);
int svbmulti_ndev(const svbmulti_t *m);
/* The hipStream_t that device i's work is queued on, for ordering a caller's own work. */
void *svbmulti_stream(svbmulti_t *m, int i);
const char *svbmulti_last_error(void);

/* What device i works on: pointers into THAT device's memory, meaning as in svbgpu_encode_batch /
 * svbgpu_decode_batch, for its nstreams streams (shard i holds the streams that follow shard i-1's
 * in the job's order). */
typedef struct {
  const uint32_t *d_values;   /* decode: written (cast away const) */
  const uint64_t *d_value_off;
  uint8_t *d_bytes;
  const uint64_t *d_byte_off;
  const uint32_t *d_counts;
  const uint32_t *d_prev;     /* may be NULL */
  uint32_t nstreams;
  /* encode, out: bytes written per stream for ALL streams of the job in job order (sum of the
   * shards' nstreams entries), on this device; */
  uint64_t *d_sizes_all;
  /* encode, out, optional: their exclusive prefix sum and the job's total in one more entry
   * (sum + 1 entries): where each stream starts in a packed concatenation. */
  uint64_t *d_offsets_all;
} svbmulti_shard_t;

/* Encodes every shard on its device, all-gathers the sizes (ncclAllGather, 8 bytes per stream) and
 * fills d_sizes_all / d_offsets_all on every device.  Everything is queued on the devices' streams;
 * returns without synchronising.  shards: svbmulti_ndev(m) entries. */
int svbmulti_encode_batch(svbmulti_t *m, int variant, unsigned flags, const svbmulti_shard_t *shards);
/* Decodes every shard on its device (no exchange). */
int svbmulti_decode_batch(svbmulti_t *m, int variant, unsigned flags, const svbmulti_shard_t *shards);
/* Waits for every device's stream. */
int svbmulti_synchronize(svbmulti_t *m);

#ifdef __cplusplus
}
#endif

#endif /* INCLUDE_STREAMVBYTE_MULTI_H_ */

// Kernels around the batched codec for list frames (include/streamvbyte_frame.h):
// validation of every list of a frame against its directory entry, the scan of
// the lists' sizes and the move of the streams into their packed places.
#pragma once
#include "svb_device.h"

namespace svb {

// Every list s (counts[s] values, sizes[s] bytes at bytes + byte_off[s], sizes[s]
// >= its key bytes, checked by the host) must hold exactly the data bytes its
// keys promise: the rule of the reference's streamvbyte_validate_stream
// (src/streamvbyte_decode.c:113-135, src/streamvbyte_0124_decode.c:206-228),
// list by list.  *ok must be 1 before the launch and is cleared by any list that
// fails.  One wavefront per list; reads 0.25 B per value.
// With PER_LIST the verdicts go to ok[s] (1 or 0, nothing need be preset) and a
// size below the key bytes counts as a failure instead of being excluded by the
// caller.
template <int V, class SizeT, bool PER_LIST>
__global__ __launch_bounds__(kBlock) void k_validate_lists(const u8* __restrict__ bytes,
                                                           const u64* __restrict__ byte_off,
                                                           const u32* __restrict__ counts,
                                                           const SizeT* __restrict__ sizes, u32 n_lists,
                                                           u32* __restrict__ ok) {
  const u32 lane = lane_id();
  const u32 nwaves = gridDim.x * kWaves;
  for (u32 s = blockIdx.x * kWaves + wave_id(); s < n_lists; s += nwaves) {
    const u32 count = counts[s];
    const u32 keylen = (count + 3u) >> 2;
    const u8* in = bytes + byte_off[s];
    if (PER_LIST && (u64)sizes[s] < keylen) {  // not even the keys fit: do not read them
      if (lane == 0) ok[s] = 0u;
      continue;
    }
    u64 acc = 0;
    for (u32 koff = 4u * lane; koff < keylen; koff += 256u) {
      u32 w = 0;
      if (koff + 4u <= keylen) {
        w = ld_u32(in + koff);
      } else {  // the last key bytes: what follows them is data
        for (u32 b = koff; b < keylen; ++b) w |= (u32)in[b] << (8u * (b - koff));
      }
      const u32 nv = min(16u, count - 4u * koff);
      acc += word_len<V>(mask_codes(w, nv), nv);
    }
    acc = wave_sum64(acc);
    const bool good = acc + keylen == (u64)sizes[s];
    if (PER_LIST) {
      if (lane == 0) ok[s] = good ? 1u : 0u;
    } else if (lane == 0 && !good) {
      *ok = 0u;
    }
  }
}

// Inclusive prefix sum of a 64-bit value over the wavefront, as three DPP scans
// of 26-bit pieces (64 lanes x 2^26 stays below 2^32).
__device__ __forceinline__ u64 wave_scan_incl64(u64 x) {
  const u32 a = (u32)x & 0x3FFFFFFu, b = (u32)(x >> 26) & 0x3FFFFFFu, c = (u32)(x >> 52);
  return (u64)wave_scan_incl(a, 0) + ((u64)wave_scan_incl(b, 0) << 26) + ((u64)wave_scan_incl(c, 0) << 52);
}

// packed_off[s] = sizes[0] + ... + sizes[s-1], sizes32[s] = sizes[s], *total = the
// sum.  One workgroup walks the array kBlock entries at a time (a frame holds at
// most 2^20 lists).
__global__ __launch_bounds__(kBlock) void k_scan_sizes(const u64* __restrict__ sizes, u32 n_lists,
                                                       u64* __restrict__ packed_off,
                                                       u32* __restrict__ sizes32, u64* __restrict__ total) {
  __shared__ u64 s_wave[kWaves];
  const u32 lane = lane_id(), wave = wave_id();
  u64 carry = 0;
  for (u32 base = 0; base < n_lists; base += kBlock) {
    const u32 i = base + threadIdx.x;
    const u64 x = i < n_lists ? sizes[i] : 0;
    const u64 incl = wave_scan_incl64(x);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    u64 before = carry, all = 0;
#pragma unroll
    for (u32 w = 0; w < (u32)kWaves; ++w) {
      const u64 t = s_wave[w];
      if (w < wave) before += t;
      all += t;
    }
    if (i < n_lists) {
      packed_off[i] = before + incl - x;
      sizes32[i] = (u32)x;
    }
    carry += all;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

// Move list s (sizes[s] bytes) from src + src_off[s] to dst + dst_off[s].  One
// wavefront per list.
__global__ __launch_bounds__(kBlock) void k_pack_lists(const u8* __restrict__ src,
                                                       const u64* __restrict__ src_off,
                                                       u8* __restrict__ dst, const u64* __restrict__ dst_off,
                                                       const u64* __restrict__ sizes, u32 n_lists) {
  const u32 lane = lane_id();
  const u32 nwaves = gridDim.x * kWaves;
  for (u32 s = blockIdx.x * kWaves + wave_id(); s < n_lists; s += nwaves) {
    const u8* a = src + src_off[s];
    u8* b = dst + dst_off[s];
    const u64 n = sizes[s], nd = n >> 2;
    for (u64 i = lane; i < nd; i += 64) st_u32(b + 4 * i, ld_u32(a + 4 * i));
    const u64 t = (nd << 2) + lane;
    if (t < n) b[t] = a[t];
  }
}

}  // namespace svb

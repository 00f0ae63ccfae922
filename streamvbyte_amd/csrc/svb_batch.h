// Batched streams: one wavefront per independent stream (posting list, gradient
// chunk), looping over that stream's wave tiles with the data offset and the
// delta carry kept in registers.  Streams need no inter-wave communication, so
// a batch is embarrassingly parallel across waves, workgroups and devices.
//
// The reference has no batched call; its callers loop over the single-stream
// functions (e.g. tests/unit.c:99-173).  Per-stream results are identical to
// those calls.
#pragma once
#include "svb_decode.h"
#include "svb_encode.h"

namespace svb {

template <int V, bool DELTA, bool ZZ>
__global__ __launch_bounds__(kBlock) void k_decode_batch(
    const u8* __restrict__ bytes, const u64* __restrict__ byte_off, u32* __restrict__ values,
    const u64* __restrict__ value_off, const u32* __restrict__ counts,
    const u32* __restrict__ prevs, u32 nstreams, u64* __restrict__ sizes) {
  __shared__ WaveLds s_lds[kWaves];
  const int lane = lane_id(), wave = wave_id();
  WaveLds& S = s_lds[wave];
  const u32 stride = gridDim.x * kWaves;
  for (u32 s = blockIdx.x * kWaves + wave; s < nstreams; s += stride) {
    const u32 count = counts[s];
    const u8* in = bytes + byte_off[s];
    u32* out = values + value_off[s];
    const u64 keylen = ((u64)count + 3) >> 2;
    const u32 n_wt = (u32)((keylen + kWtKeys - 1) / kWtKeys);
    u64 doff = 0;
    u32 run = (DELTA && prevs) ? prevs[s] : 0u;
    for (u32 wt = 0; wt < n_wt; ++wt) {
      const u64 koff = (u64)wt * kWtKeys + 4 * lane;
      const u32 w = koff < keylen ? ld_u32(in + koff) : 0u;
      const u64 int0 = (u64)wt * kWtInts;
      const u32 nints = (u32)min((u64)kWtInts, (u64)count - int0);
      uint4 v[4];
      doff += decode_wavetile<V, ZZ>(w, nints, in + keylen + doff, lane, S, v);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint4 c = v[j];
        if (DELTA) {
          c.y += c.x; c.z += c.y; c.w += c.z;
          const u32 incl = wave_scan_incl(c.w, lane);
          const u32 ex = incl - c.w + run;
          run += wave_last(incl);
          c.x += ex; c.y += ex; c.z += ex; c.w += ex;
        }
        store_group(out, int0 + 4u * (j * 64 + lane), count, c);
      }
    }
    if (sizes && lane == 0) sizes[s] = count ? keylen + doff : 0;
  }
}

template <int V, bool DELTA, bool ZZ>
__global__ __launch_bounds__(kBlock) void k_encode_batch(
    const u32* __restrict__ values, const u64* __restrict__ value_off, u8* __restrict__ bytes,
    const u64* __restrict__ byte_off, const u32* __restrict__ counts,
    const u32* __restrict__ prevs, u32 nstreams, u64* __restrict__ sizes) {
  __shared__ EncWaveLds s_lds[kWaves];
  const int lane = lane_id(), wave = wave_id();
  EncWaveLds& S = s_lds[wave];
  const u32 stride = gridDim.x * kWaves;
  for (u32 s = blockIdx.x * kWaves + wave; s < nstreams; s += stride) {
    const u32 count = counts[s];
    const u32* in = values + value_off[s];
    u8* out = bytes + byte_off[s];
    const u64 keylen = ((u64)count + 3) >> 2;
    const u32 n_wt = (u32)((keylen + kWtKeys - 1) / kWtKeys);
    u64 doff = 0;
    u32 carry = (DELTA && prevs) ? prevs[s] : 0u;
    for (u32 wt = 0; wt < n_wt; ++wt) {
      const u64 int0 = (u64)wt * kWtInts;
      const u32 nints = (u32)min((u64)kWtInts, (u64)count - int0);
      uint4 x[4];
      u32 g[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = load_group(in, int0 + 4u * (j * 64 + lane), count);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint4 c = x[j];
        if (DELTA) {
          const u32 p = wave_shift_up1(c.w, carry);
          carry = wave_last(c.w);
          c = make_uint4(c.x - p, c.y - c.x, c.z - c.y, c.w - c.z);
        }
        if (ZZ) c = make_uint4(zz_enc(c.x), zz_enc(c.y), zz_enc(c.z), zz_enc(c.w));
        const u64 first = int0 + 4u * (j * 64 + lane);
        if (DELTA || ZZ) zero_absent(c, first, count);
        x[j] = c;
        g[j] = classify_group<V>(c, first < count ? (u32)min((u64)4, (u64)count - first) : 0u);
      }
      doff += encode_wavetile<V, false>(x, g, nints, lane, S, out + keylen + doff, out + (int0 >> 2));
    }
    if (lane == 0) sizes[s] = count ? keylen + doff : 0;
  }
}

template <int V, bool D, bool Z>
void launch_decode_batch(unsigned grid, const uint8_t* d_bytes, const uint64_t* d_byte_off,
                         uint32_t* d_values, const uint64_t* d_value_off, const uint32_t* d_counts,
                         const uint32_t* d_prev, uint32_t nstreams, uint64_t* d_sizes,
                         hipStream_t st) {
  hipLaunchKernelGGL((k_decode_batch<V, D, Z>), dim3(grid), dim3(kBlock), 0, st, d_bytes, d_byte_off,
                     d_values, d_value_off, d_counts, d_prev, nstreams, d_sizes);
}
template <int V, bool D, bool Z>
void launch_encode_batch(unsigned grid, const uint32_t* d_values, const uint64_t* d_value_off,
                         uint8_t* d_bytes, const uint64_t* d_byte_off, const uint32_t* d_counts,
                         const uint32_t* d_prev, uint32_t nstreams, uint64_t* d_sizes,
                         hipStream_t st) {
  hipLaunchKernelGGL((k_encode_batch<V, D, Z>), dim3(grid), dim3(kBlock), 0, st, d_values, d_value_off,
                     d_bytes, d_byte_off, d_counts, d_prev, nstreams, d_sizes);
}

}  // namespace svb

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

#include <algorithm>

namespace svb {

template <int V, bool DELTA, bool ZZ>
__global__ __launch_bounds__(kBlock) void k_decode_batch(
    const u8* __restrict__ bytes, const u64* __restrict__ byte_off, u32* __restrict__ values,
    const u64* __restrict__ value_off, const u32* __restrict__ counts,
    const u32* __restrict__ prevs, u32 nstreams, u64* __restrict__ sizes, u32 min_count) {
  __shared__ WaveLds s_lds[kWaves];
  const int lane = lane_id(), wave = wave_id();
  WaveLds& S = s_lds[wave];
  const u32 stride = gridDim.x * kWaves;
  for (u32 s = blockIdx.x * kWaves + wave; s < nstreams; s += stride) {
    const u32 count = counts[s];
    if (count < min_count) continue;  // short lists belong to k_decode_batch_rows
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

// Short lists: one list per ROW of 16 lanes (four lists per wavefront at a time),
// one group per lane and step, no LDS at all.  A lane reads its key byte, a row
// scan (DPP row_shr) of the group lengths gives its data offset, the group's
// <= 16 data bytes come in with one 16-byte load at their own byte address
// (neighbouring lanes hit the same cache lines; the over-read stays inside the
// stream's 16-byte padding) and the four values are cut out of that register
// quad.  Handles lists of 1..kRowMax values; longer ones loop in steps of 64.
constexpr u32 kRowMax = 256;

// value of lane 15 of the caller's row
__device__ __forceinline__ u32 row_last(u32 x, int lane) { return __shfl(x, (lane & 48) | 15, 64); }

template <int V, bool DELTA, bool ZZ>
__global__ __launch_bounds__(kBlock) void k_decode_batch_rows(
    const u8* __restrict__ bytes, const u64* __restrict__ byte_off, u32* __restrict__ values,
    const u64* __restrict__ value_off, const u32* __restrict__ counts,
    const u32* __restrict__ prevs, u32 nstreams, u64* __restrict__ sizes) {
  const int lane = lane_id();
  const int lr = lane & 15;
  const u32 nrows = gridDim.x * (kBlock / 16);
  for (u32 s = blockIdx.x * (kBlock / 16) + (threadIdx.x >> 4); s < nstreams; s += nrows) {
    const u32 count = counts[s];
    if (count == 0) {
      if (sizes && lr == 0) sizes[s] = 0;
      continue;
    }
    if (count > kRowMax) continue;  // k_decode_batch takes the long lists
    const u8* in = bytes + byte_off[s];
    u32* out = values + value_off[s];
    const u32 keylen = (count + 3) >> 2;
    const u8* data = in + keylen;
    u32 doff = 0;
    u32 run = (DELTA && prevs) ? prevs[s] : 0u;
    for (u32 g0 = 0; g0 < keylen; g0 += 16) {
      const u32 g = g0 + lr;
      const bool have = g < keylen;
      u32 key = have ? in[g] : 0u;
      const u32 nv = have ? min(4u, count - 4u * g) : 0u;
      key &= nv >= 4u ? 0xFFu : ((1u << (2u * nv)) - 1u);
      const u32 L = group_len<V>(key, nv);
      const u32 incl = row_scan_incl(L);
      const u32 off = doff + incl - L;
      doff += row_last(incl, lane);
      uint4 d = make_uint4(0, 0, 0, 0);
      if (have) d = ld_u128(data + off);
      uint4 c = values_of_group<V, ZZ>(d, key, nv);
      if (DELTA) {
        c.y += c.x; c.z += c.y; c.w += c.z;
        const u32 si = row_scan_incl(c.w);
        const u32 ex = si - c.w + run;
        run += row_last(si, lane);
        c.x += ex; c.y += ex; c.z += ex; c.w += ex;
      }
      if (have) store_group(out, 4u * g, count, c);
    }
    if (sizes && lr == 0) sizes[s] = (u64)keylen + doff;
  }
}

// Short lists, encoder: one list per row of 16 lanes, one group per lane and
// step, no LDS.  A row scan of the group lengths places every group; its bytes
// are concatenated in registers and stored at their own byte address as whole
// dwords plus at most three single bytes, exactly the group's length, so
// neighbouring lanes never touch each other's bytes.
template <int V, bool DELTA, bool ZZ>
__global__ __launch_bounds__(kBlock) void k_encode_batch_rows(
    const u32* __restrict__ values, const u64* __restrict__ value_off, u8* __restrict__ bytes,
    const u64* __restrict__ byte_off, const u32* __restrict__ counts,
    const u32* __restrict__ prevs, u32 nstreams, u64* __restrict__ sizes) {
  const int lane = lane_id();
  const int lr = lane & 15;
  const u32 nrows = gridDim.x * (kBlock / 16);
  for (u32 s = blockIdx.x * (kBlock / 16) + (threadIdx.x >> 4); s < nstreams; s += nrows) {
    const u32 count = counts[s];
    if (count == 0) {
      if (lr == 0) sizes[s] = 0;
      continue;
    }
    if (count > kRowMax) continue;  // k_encode_batch takes the long lists
    const u32* in = values + value_off[s];
    u8* out = bytes + byte_off[s];
    const u32 keylen = (count + 3) >> 2;
    u8* data = out + keylen;
    u32 doff = 0;
    u32 carry = (DELTA && prevs) ? prevs[s] : 0u;
    for (u32 g0 = 0; g0 < keylen; g0 += 16) {
      const u32 g = g0 + lr;
      const bool have = g < keylen;
      const u32 nv = have ? min(4u, count - 4u * g) : 0u;
      uint4 c = load_group(in, 4ull * g, have ? count : 0u);
      if (DELTA) {
        // value before this group's first: last value of the lane to the left, of the previous step for lane 0
        const u32 last = nv == 4u ? c.w : (nv == 3u ? c.z : (nv == 2u ? c.y : c.x));
        u32 p = (u32)__builtin_amdgcn_update_dpp(0, (int)c.w, 0x111, 0xF, 0xF, true);  // row_shr:1
        if (lr == 0) p = carry;
        carry = row_last(last, lane);  // only read again when the row was full, i.e. nv == 4 in lane 15
        c = make_uint4(c.x - p, c.y - c.x, c.z - c.y, c.w - c.z);
      }
      if (ZZ) c = make_uint4(zz_enc(c.x), zz_enc(c.y), zz_enc(c.z), zz_enc(c.w));
      if (DELTA || ZZ) zero_absent(c, 4ull * g, have ? count : 0u);
      const u32 gd = classify_group<V, false>(c, nv);
      const u32 L = gi_total(gd);
      const u32 incl = row_scan_incl(L);
      u8* q = data + doff + (incl - L);
      doff += row_last(incl, lane);
      if (have) {
        out[g] = (u8)gi_key(gd);
        const uint4 d = pack_group<V>(c, gd);
        if (L >= 4u) st_u32(q, d.x);
        if (L >= 8u) st_u32(q + 4, d.y);
        if (L >= 12u) st_u32(q + 8, d.z);
        if (L >= 16u) st_u32(q + 12, d.w);
        const u32 full = L & ~3u, rest = L & 3u;
        const u32 t = full == 0u ? d.x : (full == 4u ? d.y : (full == 8u ? d.z : d.w));
        if (rest >= 1u) q[full] = (u8)t;
        if (rest >= 2u) q[full + 1] = (u8)(t >> 8);
        if (rest >= 3u) q[full + 2] = (u8)(t >> 16);
      }
    }
    if (lr == 0) sizes[s] = (u64)keylen + doff;
  }
}

template <int V, bool DELTA, bool ZZ>
__global__ __launch_bounds__(kBlock) void k_encode_batch(
    const u32* __restrict__ values, const u64* __restrict__ value_off, u8* __restrict__ bytes,
    const u64* __restrict__ byte_off, const u32* __restrict__ counts,
    const u32* __restrict__ prevs, u32 nstreams, u64* __restrict__ sizes, u32 min_count) {
  __shared__ EncWaveLds s_lds[kWaves];
  const int lane = lane_id(), wave = wave_id();
  EncWaveLds& S = s_lds[wave];
  const u32 stride = gridDim.x * kWaves;
  for (u32 s = blockIdx.x * kWaves + wave; s < nstreams; s += stride) {
    const u32 count = counts[s];
    if (count < min_count) continue;  // short lists belong to k_encode_batch_rows
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
        g[j] = classify_group<V, false>(c, first < count ? (u32)min((u64)4, (u64)count - first) : 0u);
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
  // short lists, four to a wavefront; then the long ones, one wavefront each
  const unsigned rows_grid = (unsigned)std::min<unsigned long long>(grid, ((unsigned long long)nstreams + 15) / 16);
  hipLaunchKernelGGL((k_decode_batch_rows<V, D, Z>), dim3(rows_grid ? rows_grid : 1), dim3(kBlock), 0, st,
                     d_bytes, d_byte_off, d_values, d_value_off, d_counts, d_prev, nstreams, d_sizes);
  hipLaunchKernelGGL((k_decode_batch<V, D, Z>), dim3(grid), dim3(kBlock), 0, st, d_bytes, d_byte_off,
                     d_values, d_value_off, d_counts, d_prev, nstreams, d_sizes, kRowMax + 1);
}
template <int V, bool D, bool Z>
void launch_encode_batch(unsigned grid, const uint32_t* d_values, const uint64_t* d_value_off,
                         uint8_t* d_bytes, const uint64_t* d_byte_off, const uint32_t* d_counts,
                         const uint32_t* d_prev, uint32_t nstreams, uint64_t* d_sizes,
                         hipStream_t st) {
  const unsigned rows_grid = (unsigned)std::min<unsigned long long>(grid, ((unsigned long long)nstreams + 15) / 16);
  hipLaunchKernelGGL((k_encode_batch_rows<V, D, Z>), dim3(rows_grid ? rows_grid : 1), dim3(kBlock), 0, st,
                     d_values, d_value_off, d_bytes, d_byte_off, d_counts, d_prev, nstreams, d_sizes);
  hipLaunchKernelGGL((k_encode_batch<V, D, Z>), dim3(grid), dim3(kBlock), 0, st, d_values, d_value_off,
                     d_bytes, d_byte_off, d_counts, d_prev, nstreams, d_sizes, kRowMax + 1);
}

}  // namespace svb

// 64-bit values: the "1248" variant (code c -> 1 << c data bytes).  The reference's
// container with wider values; the reference itself stops at uint32_t
// (include/streamvbyte.h:23 there), so there is nothing to port and nothing to be
// bit-compatible with except this library's own format rule (include/streamvbyte_gpu.h).
//
// Both directions work a group (four values, 4..32 data bytes) per lane and step, 64
// groups a step, four steps a wave tile, with a wave scan of the group lengths for the
// places; a wave tile's own place comes from the same two-level directory as the 32-bit
// decoders (Dir):
//   decode:  k_key_sums<V1248> over the key block, then k_decode64;
//   encode:  k_sizes64 over the VALUES (the sizes are not known before), then k_encode64.
// The encoder therefore reads its input twice (16 + 0.25 + dbar bytes per value instead of
// 8 + 0.25 + dbar): a first, plain implementation -- the look-back encoder of the 32-bit
// path is the model for a single-pass one.
#pragma once
#include "svb_decode.h"

namespace svb {

// width code of a 64-bit value: 0..3 for 1, 2, 4, 8 bytes
__device__ __forceinline__ u32 code64(u64 v) {
  return (u32)(v > 0xFFull) + (u32)(v > 0xFFFFull) + (u32)(v > 0xFFFFFFFFull);
}
// data bytes of the first nv values of a key
__device__ __forceinline__ u32 group_len64(u32 key, u32 nv) {
  u32 L = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) L += (u32)k < nv ? 1u << ((key >> (2 * k)) & 3u) : 0u;
  return L;
}

// ---------------------------------------------------------------------------
// k_sizes64: the encoder's directory.  One workgroup of 1024 threads per mega tile;
// a round takes 16 wave tiles, one per wave (8 loads of 16 bytes per lane), sums the
// widths, and wave 0 lays the 16 sums behind the mega tile's running offset.
template <int NT = 0>
__global__ __launch_bounds__(1024) void k_sizes64(const u64* __restrict__ in, u32 count, u32* __restrict__ wt_rel,
                                                  u32* __restrict__ mega_sum, u32 wt_per_mega) {
  __shared__ u32 s_sum[2][16];
  const int lane = lane_id(), wave = wave_id();
  const u64 n_wt = ((u64)count + kWtInts - 1) / kWtInts;
  const u64 wt0 = (u64)blockIdx.x * wt_per_mega, wt1 = min(n_wt, wt0 + wt_per_mega);
  u32 base = 0;
  int buf = 0;
  for (u64 wtb = wt0; wtb < wt1; wtb += 16) {
    const u64 wt = wtb + wave;
    u32 s = 0;
    if (wt < wt1) {
      const u64 first = wt * kWtInts;
#pragma unroll
      for (int i = 0; i < 8; ++i) {  // values first + 2 * (64 i + lane), + 1
        const u64 idx = first + 2ull * (64 * i + lane);
        u64 a = 0, b = 0;
        if (idx + 1 < count) {
          const uint4 p = ld_stream16<(NT & kNtLoads) != 0>(in + idx);
          a = ((u64)p.y << 32) | p.x; b = ((u64)p.w << 32) | p.z;
          s += (1u << code64(a)) + (1u << code64(b));
        } else if (idx < count) {
          s += 1u << code64(in[idx]);
        }
      }
    }
    s = wave_sum(s);
    if (lane == 0) s_sum[buf][wave] = s;
    __syncthreads();
    if (wave == 0) {
      const u32 x = lane < 16 ? s_sum[buf][lane] : 0u;
      const u32 incl = wave_scan_incl(x, lane);
      if (lane < 16 && wtb + lane < wt1) wt_rel[wtb + lane] = base + incl - x;
      base += wave_last(incl);
    }
    buf ^= 1;
  }
  if (threadIdx.x == 0) mega_sum[blockIdx.x] = base;
}

// ---------------------------------------------------------------------------
// k_encode64: a wave per wave tile.  Lane l of step s holds group 64 s + l: two 16-byte
// loads, four width codes, a wave scan of the group lengths, then every value stored at
// its own byte address in exactly its own length (1, 2, 4 or 8 bytes).
template <int NT = 0>
__global__ __launch_bounds__(kBlock) void k_encode64(const u64* __restrict__ in, u32 count, u8* __restrict__ out,
                                                     Dir dir, u64* __restrict__ written) {
  const int lane = lane_id(), wave = wave_id();
  const u64 keylen = ((u64)count + 3) >> 2;
  const u32 n_wt = (u32)((keylen + kWtKeys - 1) / kWtKeys);
  const u32 wt = blockIdx.x * kWaves + wave;
  if (wt >= n_wt) return;
  u64 doff = dir_offset(dir, wt, lane);
  u8* data = out + keylen;
#pragma unroll 1
  for (int step = 0; step < 4; ++step) {
    const u64 g = (u64)wt * kWtKeys + 64 * step + lane;  // group = key index
    const u64 first = 4 * g;
    const u32 nv = first < count ? (u32)min((u64)4, (u64)count - first) : 0u;
    u64 v[4] = {0, 0, 0, 0};
    if (nv == 4u) {
      const uint4 p = ld_stream16<(NT & kNtLoads) != 0>(in + first), q = ld_stream16<(NT & kNtLoads) != 0>(in + first + 2);
      v[0] = ((u64)p.y << 32) | p.x; v[1] = ((u64)p.w << 32) | p.z;
      v[2] = ((u64)q.y << 32) | q.x; v[3] = ((u64)q.w << 32) | q.z;
    } else {
      for (u32 k = 0; k < nv; ++k) v[k] = in[first + k];
    }
    u32 key = 0, len[4], L = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32 c = (u32)k < nv ? code64(v[k]) : 0u;
      key |= c << (2 * k);
      len[k] = (u32)k < nv ? 1u << c : 0u;
      L += len[k];
    }
    const u32 incl = wave_scan_incl(L, lane);
    u8* p = data + doff + (incl - L);
    doff += wave_last(incl);
    if (nv) {
      out[g] = (u8)key;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u64 x = v[k];
        if (len[k] == 8u) __builtin_memcpy(p, &x, 8);
        else if (len[k] == 4u) { const u32 y = (u32)x; __builtin_memcpy(p, &y, 4); }
        else if (len[k] == 2u) { const unsigned short y = (unsigned short)x; __builtin_memcpy(p, &y, 2); }
        else if (len[k] == 1u) *p = (u8)x;
        p += len[k];
      }
    }
  }
  if (wt == n_wt - 1 && lane == 0 && written) *written = keylen + doff;
}

// ---------------------------------------------------------------------------
// k_decode64: a wave per wave tile.  Lane l of step s holds group 64 s + l: its key byte,
// a wave scan of the group lengths, two 16-byte loads at the group's own byte address (they
// may run up to 31 bytes past the group: SVBGPU_PADDING64), four values cut out of the 32
// bytes, two 16-byte stores.
__device__ __forceinline__ u64 bytes64_at(const u32 (&d)[8], u32 o, u32 len) {
  // the 8 bytes at byte offset o (0..24) of the 32-byte window d, then the low `len` of them
  const u32 i = o >> 2;
  u32 a = 0, b = 0, c = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (i == (u32)j) a = d[j];
    if (i + 1 == (u32)j) b = d[j];
    if (i + 2 == (u32)j) c = d[j];
  }
  const u32 lo = __builtin_amdgcn_alignbyte(b, a, o), hi = __builtin_amdgcn_alignbyte(c, b, o);
  const u64 x = ((u64)hi << 32) | lo;
  return len >= 8u ? x : x & ((1ull << (8u * len)) - 1ull);
}
template <int NT = 0>
__global__ __launch_bounds__(kBlock) void k_decode64(const u8* __restrict__ in, u64* __restrict__ out, u32 count,
                                                     Dir dir, u64* __restrict__ consumed) {
  const int lane = lane_id(), wave = wave_id();
  const u64 keylen = ((u64)count + 3) >> 2;
  const u32 n_wt = (u32)((keylen + kWtKeys - 1) / kWtKeys);
  const u32 wt = blockIdx.x * kWaves + wave;
  if (wt >= n_wt) return;
  u64 doff = dir_offset(dir, wt, lane);
  const u8* data = in + keylen;
#pragma unroll 1
  for (int step = 0; step < 4; ++step) {
    const u64 g = (u64)wt * kWtKeys + 64 * step + lane;
    const u64 first = 4 * g;
    const u32 nv = first < count ? (u32)min((u64)4, (u64)count - first) : 0u;
    const u32 key = nv ? in[g] : 0u;
    const u32 L = group_len64(key, nv);
    const u32 incl = wave_scan_incl(L, lane);
    const u8* p = data + doff + (incl - L);
    doff += wave_last(incl);
    if (nv) {
      uint4 lo4 = ld_u128(p), hi4 = ld_u128(p + 16);
      const u32 d[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
      u64 v[4];
      u32 o = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32 len = (u32)k < nv ? 1u << ((key >> (2 * k)) & 3u) : 0u;
        v[k] = len ? bytes64_at(d, o, len) : 0ull;
        o += len;
      }
      if (nv == 4u) {
        st_stream16<(NT & kNtStores) != 0>(out + first, make_uint4((u32)v[0], (u32)(v[0] >> 32), (u32)v[1], (u32)(v[1] >> 32)));
        st_stream16<(NT & kNtStores) != 0>(out + first + 2, make_uint4((u32)v[2], (u32)(v[2] >> 32), (u32)v[3], (u32)(v[3] >> 32)));
      } else {
        for (u32 k = 0; k < nv; ++k) out[first + k] = v[k];
      }
    }
  }
  if (wt == n_wt - 1 && lane == 0 && consumed) *consumed = keylen + doff;
}

}  // namespace svb

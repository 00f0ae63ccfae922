// Encode path: classify, pack keys, scan lengths, scatter data bytes via LDS.
//
// Replaces the serial encoder loops of the reference
// (src/streamvbyte_x64_encode.c:15-29,55-97; scalar src/streamvbyte_encode.c:14-62;
// 0124 src/streamvbyte_0124_encode.c:12-57,61-113; delta
// src/streamvbytedelta_x64_encode.c:13-75, src/streamvbytedelta_encode.c:39-76).
//
// One workgroup = one tile of 16 wave tiles (16384 values); wave w owns wave
// tiles 4w..4w+3.  Pass 1 loads the 64 values of every lane into registers and
// sums their widths; the tile's byte count is published and the tile's data
// offset obtained by decoupled look-back; pass 2 scatters the data bytes of one
// wave tile at a time into a wave-private LDS image laid out on the 16-byte
// grid of the destination, then copies it out with aligned 16-byte stores
// (byte stores for the ragged head and tail, which neighbouring waves share).
// HBM traffic per value: 4 (in) + 0.25 (keys) + dbar (data).
#pragma once
#include "svb_device.h"
#include "svb_lookback.h"

namespace svb {

struct __align__(16) EncWaveLds {
  u8 data[kWtDataMax + 32];  // image of one wave tile's data bytes (+16 alignment shift)
  u8 keys[kWtKeys];
};

// Load the 4 values of group `g` (global value index first = 4*g) with bounds.
__device__ __forceinline__ uint4 load_group(const u32* __restrict__ in, u64 first, u32 count) {
  uint4 x = make_uint4(0, 0, 0, 0);
  if (first + 4 <= count) {
    __builtin_memcpy(&x, in + first, 16);
  } else if (first < count) {
    x.x = in[first];
    if (first + 1 < count) x.y = in[first + 1];
    if (first + 2 < count) x.z = in[first + 2];
  }
  return x;
}

// Key byte and packed per-value lengths (4 x 8 bit) of one group with `nv`
// valid values (absent values: code 0, no bytes).
template <int V>
__device__ __forceinline__ void classify_group(const uint4& x, u32 nv, u32& key, u32& lens) {
  const u32 v[4] = {x.x, x.y, x.z, x.w};
  key = 0;
  lens = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool ok = (u32)k < nv;
    const u32 c = ok ? code_of<V>(v[k]) : 0u;
    const u32 l = ok ? len_of<V>(c) : 0u;
    key |= c << (2 * k);
    lens |= l << (8 * k);
  }
}

// One wave tile, values -> bytes.  x[j] holds group j*64+lane (already delta /
// zigzag transformed), `nints` of the tile's 1024 values exist.  Scatters the
// data bytes into the wave's LDS image, copies them to dst[0..total) and the
// (nints+3)/4 keys to kdst.  Returns total.
template <int V>
__device__ __forceinline__ u32 encode_wavetile(const uint4 (&x)[4], u32 nints, int lane,
                                               EncWaveLds& S, u8* dst, u8* kdst) {
  u32 key[4], lens[4];
  u64 packed = 0;  // 4 x 16-bit group lengths, one per j
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32 first = 4u * (j * 64 + lane);
    const u32 nv = first < nints ? min(4u, nints - first) : 0u;
    classify_group<V>(x[j], nv, key[j], lens[j]);
    packed |= (u64)((lens[j] * 0x01010101u) >> 24) << (16 * j);
  }
  // one 64-bit wave scan gives, per j, the inclusive prefix over lanes
  u64 incl = packed;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u64 y = __shfl_up(incl, d, 64);
    if (lane >= d) incl += y;
  }
  const u64 tots = __shfl(incl, 63, 64);
  const u64 excl = incl - packed;
  const u32 a = (u32)((uintptr_t)dst & 15u);  // LDS image byte i+a <-> dst[i]
  u32 jbase = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u32 p = a + jbase + (u32)((excl >> (16 * j)) & 0xFFFFu);
    jbase += (u32)((tots >> (16 * j)) & 0xFFFFu);
    const u32 v[4] = {x[j].x, x[j].y, x[j].z, x[j].w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32 l = (lens[j] >> (8 * k)) & 0xFFu;
      if (l > 0) S.data[p] = (u8)v[k];
      if (l > 1) S.data[p + 1] = (u8)(v[k] >> 8);
      if (l > 2) S.data[p + 2] = (u8)(v[k] >> 16);
      if (l > 3) S.data[p + 3] = (u8)(v[k] >> 24);
      p += l;
    }
    S.keys[j * 64 + lane] = (u8)key[j];
  }
  const u32 total = jbase;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // data: aligned interior chunks, ragged head / tail by bytes
  {
    const u32 end = a + total;                 // image bytes [a, end)
    const u32 c_lo = (a + 15u) >> 4;           // first chunk lying fully inside
    const u32 c_hi = end >> 4;                 // one past the last such chunk
    u8* g0 = dst - a;                          // 16-byte aligned
    uint4* g16 = reinterpret_cast<uint4*>(g0);
    const uint4* l16 = reinterpret_cast<const uint4*>(S.data);
    for (u32 c = c_lo + lane; c < c_hi; c += 64) g16[c] = l16[c];
    const u32 hend = min(16u * c_lo, end);     // head bytes [a, hend)
    const u32 tstart = max(16u * c_hi, hend);  // tail bytes [tstart, end)
    if (lane < 16) {
      const u32 i = a + lane;
      if (i < hend) g0[i] = S.data[i];
    } else if (lane < 32) {
      const u32 i = tstart + (lane - 16);
      if (i < end) g0[i] = S.data[i];
    }
  }
  // keys
  {
    const u32 nkeys = (nints + 3u) >> 2;
    if ((((uintptr_t)kdst) & 3u) == 0 && 4u * lane + 4u <= nkeys) {
      *reinterpret_cast<u32*>(kdst + 4 * lane) = reinterpret_cast<const u32*>(S.keys)[lane];
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (4u * lane + b < nkeys) kdst[4 * lane + b] = S.keys[4 * lane + b];
    }
  }
  __builtin_amdgcn_wave_barrier();
  return total;
}

template <int V, bool DELTA, bool ZZ>
__global__ __launch_bounds__(kBlock, 4) void k_encode(const u32* __restrict__ in, u32 count,
                                                      u8* __restrict__ out, u32 prev,
                                                      WsHeader* __restrict__ hdr,
                                                      u64* __restrict__ status,
                                                      u64* __restrict__ written) {
  __shared__ EncWaveLds s_lds[kWaves];
  __shared__ u32 s_wave_tot[kWaves];
  __shared__ u64 s_prefix;
  __shared__ u32 s_tile;

  const int lane = lane_id(), wave = wave_id();
  if (threadIdx.x == 0) s_tile = atomicAdd(&hdr->ticket, 1u);
  __syncthreads();
  const u32 tile = s_tile;
  const u64 keylen = ((u64)count + 3) >> 2;
  const u64 wave_first = ((u64)tile * kLbWt + (u64)wave * 4) * kWtInts;  // first value of this wave

  // ---- pass 1: load, transform, widths ------------------------------------
  uint4 x[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      x[q][j] = load_group(in, wave_first + (u64)q * kWtInts + 4u * (j * 64 + lane), count);

  if (DELTA) {
    // previous value of the wave's first element, then neighbours by lane shuffles
    u32 carry = prev;
    if (wave_first > 0 && wave_first <= count) carry = in[wave_first - 1];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint4 c = x[q][j];
        u32 p = __shfl_up(c.w, 1, 64);
        if (lane == 0) p = carry;
        carry = __shfl(c.w, 63, 64);
        x[q][j] = make_uint4(c.x - p, c.y - c.x, c.z - c.y, c.w - c.z);
      }
  }
  if (ZZ) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint4& c = x[q][j];
        c = make_uint4(zz_enc(c.x), zz_enc(c.y), zz_enc(c.z), zz_enc(c.w));
      }
  }

  u32 lane_tot = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u64 first = wave_first + (u64)q * kWtInts + 4u * (j * 64 + lane);
      const u32 nv = first < count ? (u32)min((u64)4, (u64)count - first) : 0u;
      u32 key, lens;
      classify_group<V>(x[q][j], nv, key, lens);
      lane_tot += (lens * 0x01010101u) >> 24;
    }
  const u32 wave_tot = wave_sum(lane_tot);
  if (lane == 0) s_wave_tot[wave] = wave_tot;
  __syncthreads();

  // ---- tile offset by decoupled look-back ----------------------------------
  if (wave == 0) {
    const u32 tile_tot = s_wave_tot[0] + s_wave_tot[1] + s_wave_tot[2] + s_wave_tot[3];
    u64 excl = 0;
    if (tile == 0) {
      if (lane == 0) lb_publish(status, 0, kLbPfx, tile_tot);
    } else {
      if (lane == 0) lb_publish(status, tile, kLbAgg, tile_tot);
      excl = lb_exclusive_prefix(status, tile, lane, &hdr->error);
      if (lane == 0) lb_publish(status, tile, kLbPfx, excl + tile_tot);
    }
    if (lane == 0) {
      s_prefix = excl;
      const u64 last_tile = keylen ? (keylen - 1) / ((u64)kLbWt * kWtKeys) : 0;
      if (tile == last_tile && written) *written = keylen + excl + tile_tot;
    }
  }
  __syncthreads();
  u64 doff = s_prefix;
  for (int w = 0; w < wave; ++w) doff += s_wave_tot[w];

  // ---- pass 2: per wave tile, scan -> scatter into LDS -> store ----------------
  EncWaveLds& S = s_lds[wave];
  u8* data_out = out + keylen;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const u64 wt_first = wave_first + (u64)q * kWtInts;
    if (wt_first < count) {
      const u32 nints = (u32)min((u64)kWtInts, (u64)count - wt_first);
      doff += encode_wavetile<V>(x[q], nints, lane, S, data_out + doff, out + (wt_first >> 2));
    }
  }
}

}  // namespace svb

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
//   encode:  k_encode64_lb, one pass: the tile offsets come from the same decoupled look-back as
//            the 32-bit encoder's, the bytes are assembled in an LDS image.
#pragma once
#include "svb_decode.h"
#include "svb_encode.h"
#include "svb_lists.h"

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
// k_encode64_lb: the single-pass encoder.  The 32-bit encoder's tile cycle (k_encode in
// svb_encode.h) on 64-bit values: persistent workgroups draw tiles of NW wave tiles from the
// ticket; a wave loads its wave tile once (lane l of step s holds group 64 s + l: two 16-byte
// loads), classifies, the tile's byte count goes through the two-level look-back, and the bytes
// are assembled in a wave-private LDS image laid out on the 16-byte grid of the destination and
// copied out with aligned 16-byte stores.  8 + 0.25 + dbar bytes per value.
struct __align__(16) EncWaveLds64 {
  u8 data[2 * kWtDataMax + 64];  // image of one wave tile's data bytes (+16 alignment shift, +36 store overhang)
  u8 keys[kWtKeys];
};
typedef unsigned __int128 u128;
__device__ __forceinline__ u64 zz_enc64(u64 v) { return (v << 1) ^ (0ull - (v >> 63)); }
__device__ __forceinline__ u64 zz_dec64(u64 v) { return (v >> 1) ^ (0ull - (v & 1ull)); }
// The data bytes of a group, value 0 first, as eight dwords (values are below 256^len, absent ones zero).
__device__ __forceinline__ void pack_group64(const u64 (&x)[4], const u32 (&len)[4], u32 (&d)[8]) {
  const u128 lo = (u128)x[0] | ((u128)x[1] << (8u * len[0]));  // len[0] <= 8
  const u128 hi = (u128)x[2] | ((u128)x[3] << (8u * len[2]));
  const u32 s = len[0] + len[1];                                // 0..16 bytes taken by values 0 and 1
  const u128 low = s >= 16u ? lo : (lo | (hi << (8u * s)));
  const u128 high = s >= 16u ? hi : (s == 0u ? (u128)0 : (hi >> (128u - 8u * s)));
  d[0] = (u32)low; d[1] = (u32)(low >> 32); d[2] = (u32)(low >> 64); d[3] = (u32)(low >> 96);
  d[4] = (u32)high; d[5] = (u32)(high >> 32); d[6] = (u32)(high >> 64); d[7] = (u32)(high >> 96);
}
// First half of a wave tile's encoding: loads the 1024 values that start at in[wt_first] (in[] holds `count`
// values; lane l of step s gets group 64 s + l, absent values zero), applies DELTA (the value before the wave
// tile's first is in[wt_first - 1], or `prev` at the start) and ZZ, and classifies.  key4 / lens4: the lane's four
// keys and four group lengths (0..32), one byte per step.  Returns the wave tile's data bytes.
template <bool DELTA, bool ZZ, int NT>
__device__ __forceinline__ u32 classify_wavetile64(const u64* __restrict__ in, u64 count, u64 wt_first, u64 prev, int lane,
                                                   u64 (&v)[4][4], u32& key4, u32& lens4) {
  key4 = 0;
  lens4 = 0;
#pragma unroll
  for (int step = 0; step < 4; ++step) {
    const u64 first = wt_first + 4ull * (64 * step + lane);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[step][k] = 0;
    if (first + 4 <= count) {
      const uint4 p = ld_stream16<(NT & kNtLoads) != 0>(in + first), q = ld_stream16<(NT & kNtLoads) != 0>(in + first + 2);
      v[step][0] = ((u64)p.y << 32) | p.x; v[step][1] = ((u64)p.w << 32) | p.z;
      v[step][2] = ((u64)q.y << 32) | q.x; v[step][3] = ((u64)q.w << 32) | q.z;
    } else {
      for (u32 k = 0; k < 4 && first + k < count; ++k) v[step][k] = in[first + k];
    }
  }
  if (DELTA) {  // value - the value before it (wrap-around); the one before the wave tile's first comes from memory
    u64 carry = prev;
    if (wt_first > 0 && wt_first <= count) carry = in[wt_first - 1];
#pragma unroll
    for (int step = 0; step < 4; ++step) {
      const u64 last = v[step][3];
      const u32 pl = wave_shift_up1((u32)last, (u32)carry), ph = wave_shift_up1((u32)(last >> 32), (u32)(carry >> 32));
      carry = ((u64)wave_last((u32)(last >> 32)) << 32) | wave_last((u32)last);
      const u64 before = ((u64)ph << 32) | pl;
      v[step][3] -= v[step][2]; v[step][2] -= v[step][1]; v[step][1] -= v[step][0]; v[step][0] -= before;
    }
  }
#pragma unroll
  for (int step = 0; step < 4; ++step) {
    const u64 first = wt_first + 4ull * (64 * step + lane);
    const u32 nv = first < count ? (u32)min((u64)4, count - first) : 0u;
    u32 key = 0, L = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      u64 x = v[step][k];
      if (ZZ) x = zz_enc64(x);
      if ((u32)k >= nv) x = 0;
      v[step][k] = x;
      const u32 c = (u32)k < nv ? code64(x) : 0u;
      key |= c << (2 * k);
      L += (u32)k < nv ? 1u << c : 0u;
    }
    key4 |= key << (8 * step);
    lens4 |= L << (8 * step);
  }
  const u32 lane_tot = (lens4 & 0xFFu) + ((lens4 >> 8) & 0xFFu) + ((lens4 >> 16) & 0xFFu) + (lens4 >> 24);
  return wave_sum(lane_tot);
}
// Second half: the classified wave tile (index wt of its stream, values from wt_first on) goes to the stream at
// `out` -- its keys to out + 256 wt, its data bytes to out + keylen + doff -- through the wave's LDS image.
template <int NT>
__device__ __forceinline__ void emit_wavetile64(EncWaveLds64& S, u8* __restrict__ out, u64 keylen, u64 count, u64 wt,
                                                u64 wt_first, u64 doff, int lane, u64 (&v)[4][4], u32 key4, u32 lens4) {
  // ---- places of the groups: scans over the lanes, one 16-bit field per step --------------
  const u64 packed = (u64)(lens4 & 0xFFu) | ((u64)((lens4 >> 8) & 0xFFu) << 16) | ((u64)((lens4 >> 16) & 0xFFu) << 32) |
                     ((u64)(lens4 >> 24) << 48);
  const u32 inc_lo = wave_scan_incl((u32)packed, lane), inc_hi = wave_scan_incl((u32)(packed >> 32), lane);
  const u64 incl = ((u64)inc_hi << 32) | inc_lo;
  const u64 tots = ((u64)wave_last(inc_hi) << 32) | wave_last(inc_lo);
  const u64 excl = incl - packed;
  u8* dst = out + keylen + doff;
  const u32 a = (u32)((uintptr_t)dst & 15u);  // LDS image byte i + a <-> dst[i]
  const u32 t0 = (u32)(tots & 0xFFFFu), t1 = (u32)((tots >> 16) & 0xFFFFu), t2 = (u32)((tots >> 32) & 0xFFFFu),
            t3 = (u32)(tots >> 48);
  const u32 total = t0 + t1 + t2 + t3;
  const u32 pj[4] = {a + (u32)(excl & 0xFFFFu), a + t0 + (u32)((excl >> 16) & 0xFFFFu),
                     a + t0 + t1 + (u32)((excl >> 32) & 0xFFFFu), a + t0 + t1 + t2 + (u32)(excl >> 48)};
  // Aligned dword stores as in the 32-bit 1234 image (encode_wavetile): steps last first and
  // dword index by dword index, so that on a border dword the earlier group's tail lands after the
  // later group's dword 0; the later group's first three bytes are written again afterwards.  Whole
  // groups have at least four bytes; the one group that may have fewer is the stream's last.
  u32 head[4];
#pragma unroll
  for (int step = 3; step >= 0; --step) {
    const u32 L = (lens4 >> (8 * step)) & 0xFFu, key = (key4 >> (8 * step)) & 0xFFu;
    S.keys[step * 64 + lane] = (u8)key;
    u32 len[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) len[k] = 0;
    {  // lengths of the present values: the first nv codes count, the others are zero values of length 0
      const u64 first = wt_first + 4ull * (64 * step + lane);
      const u32 nv = first < count ? (u32)min((u64)4, count - first) : 0u;
#pragma unroll
      for (int k = 0; k < 4; ++k) len[k] = (u32)k < nv ? 1u << ((key >> (2 * k)) & 3u) : 0u;
    }
    u32 d[8];
    pack_group64(v[step], len, d);
    const u32 p = pj[step];
    const u32 sel = shift_selector(p);
    u32* q = reinterpret_cast<u32*>(S.data + (p & ~3u));
    const u32 e = L ? L + (p & 3u) : 0u;  // dword I is written iff e > 4 I
    if (e > 0u) q[0] = __builtin_amdgcn_perm(d[0], 0u, sel);
#pragma unroll
    for (int I = 1; I < 8; ++I)
      if (e > 4u * I) q[I] = __builtin_amdgcn_perm(d[I], d[I - 1], sel);
    if (e > 32u) q[8] = __builtin_amdgcn_perm(0u, d[7], sel);
    head[step] = d[0];
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int step = 0; step < 4; ++step)
    if ((lens4 >> (8 * step)) & 0xFFu) {
      const u32 p = pj[step];
      S.data[p] = (u8)head[step];
      S.data[p + 1] = (u8)(head[step] >> 8);
      S.data[p + 2] = (u8)(head[step] >> 16);
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // data: aligned interior chunks, ragged head / tail by bytes
  {
    const u32 end = a + total;
    const u32 c_lo = (a + 15u) >> 4, c_hi = end >> 4;
    u8* g0 = dst - a;
    uint4* g16 = reinterpret_cast<uint4*>(g0);
    const uint4* l16 = reinterpret_cast<const uint4*>(S.data);
    for (u32 c = c_lo + lane; c < c_hi; c += 64) st_stream16<(NT & kNtStores) != 0>(g16 + c, l16[c]);
    const u32 hend = min(16u * c_lo, end);
    const u32 tstart = max(16u * c_hi, hend);
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
    const u64 k0 = wt * kWtKeys;
    const u32 nkeys = (u32)min((u64)kWtKeys, keylen - k0);
    u8* kdst = out + k0;
    if ((((uintptr_t)kdst) & 3u) == 0 && 4u * lane + 4u <= nkeys) {
      *reinterpret_cast<u32*>(kdst + 4 * lane) = reinterpret_cast<const u32*>(S.keys)[lane];
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (4u * lane + b < nkeys) kdst[4 * lane + b] = S.keys[4 * lane + b];
    }
  }
  __builtin_amdgcn_wave_barrier();
}

template <bool DELTA, bool ZZ, int NW, int NT>
__global__ __launch_bounds__(NW * 64, 4) void k_encode64_lb(const u64* __restrict__ in, u32 count, u32 n_tiles,
                                                           u8* __restrict__ out, u64 prev, WsHeader* __restrict__ hdr,
                                                           LbWords lb, u64* __restrict__ written) {
  __shared__ EncWaveLds64 s_lds[NW];
  __shared__ u32 s_wave_tot[NW];
  __shared__ u64 s_prefix;
  __shared__ u32 s_tile;
  const int lane = lane_id(), wave = wave_id();
  const u64 keylen = ((u64)count + 3) >> 2;
  EncWaveLds64& S = s_lds[wave];
  for (;;) {
    if (threadIdx.x == 0) s_tile = atomicAdd(&hdr->ticket, 1u);
    __syncthreads();
    const u32 tile = __builtin_amdgcn_readfirstlane(s_tile);
    if (tile >= n_tiles) break;
    const u64 wt = (u64)tile * NW + wave;
    const u64 wt_first = wt * kWtInts;
    u64 v[4][4];
    u32 key4, lens4;
    const u32 wave_tot = classify_wavetile64<DELTA, ZZ, NT>(in, count, wt_first, prev, lane, v, key4, lens4);
    if (lane == 0) s_wave_tot[wave] = wave_tot;
    __syncthreads();

    // ---- tile offset ---------------------------------------------------------
    if (wave == 0) {
      u32 tile_tot = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) tile_tot += s_wave_tot[w];
      const u64 excl = lb2_exclusive_prefix(lb, tile, tile_tot, lane, &hdr->error);
      if (lane == 0) {
        s_prefix = excl;
        if (tile == 0u) reset_other_ticket(hdr);
        if (tile == n_tiles - 1 && written) *written = keylen + excl + tile_tot;
      }
    }
    __syncthreads();
    u64 doff = s_prefix;
    for (int w = 0; w < wave; ++w) doff += s_wave_tot[w];
    if (wt_first < count) emit_wavetile64<NT>(S, out, keylen, count, wt, wt_first, doff, lane, v, key4, lens4);
    if (gridDim.x >= n_tiles) break;  // launched with a workgroup per tile: every ticket has its drawer
  }
}

// ---------------------------------------------------------------------------
// Batched 64-bit calls: a wavefront per stream, its wave tiles one after the other, the data offset and
// the differential carry in registers (no look-back, no directory).  Stream s: counts[s] values at
// values + value_off[s], bytes at bytes + byte_off[s]; sizes[s] = bytes written / consumed.
#ifndef SVB_BATCH64_WAVES
#define SVB_BATCH64_WAVES 1
#endif
// Streams (= wavefronts) per workgroup.  One: the waves share nothing, and a slot goes back to the dispatcher the
// moment its stream ends (posting lists of 1..4096 values, four per workgroup: encode 1.49 ms, one: 1.03 ms).
constexpr int kBatch64Waves = SVB_BATCH64_WAVES;
template <bool DELTA, bool ZZ>
__global__ __launch_bounds__(64 * kBatch64Waves) void k_encode64_batch(const u64* __restrict__ values, const u64* __restrict__ value_off,
                                                           u8* __restrict__ bytes, const u64* __restrict__ byte_off,
                                                           const u32* __restrict__ counts, const u64* __restrict__ prevs,
                                                           u32 nstreams, u64* __restrict__ sizes) {
  __shared__ EncWaveLds64 s_lds[kBatch64Waves];
  const int lane = lane_id(), wave = wave_id();
  const u32 s = blockIdx.x * kBatch64Waves + wave;
  if (s >= nstreams) return;
  const u64 count = counts[s];
  const u64* in = values + value_off[s];
  u8* out = bytes + byte_off[s];
  const u64 prev = (DELTA && prevs) ? prevs[s] : 0ull;
  const u64 keylen = (count + 3) >> 2;
  u64 doff = 0;
  for (u64 wt = 0; wt * kWtInts < count; ++wt) {
    u64 v[4][4];
    u32 key4, lens4;
    const u32 tot = classify_wavetile64<DELTA, ZZ, 0>(in, count, wt * kWtInts, prev, lane, v, key4, lens4);
    emit_wavetile64<0>(s_lds[wave], out, keylen, count, wt, wt * kWtInts, doff, lane, v, key4, lens4);
    doff += tot;
  }
  if (lane == 0 && sizes) sizes[s] = keylen + doff;
}

// ---------------------------------------------------------------------------
// k_decode64: a wave per wave tile.  Lane l of step s holds group 64 s + l: its key byte and a
// wave scan of the group lengths give the group's place; the wave tile's data bytes (up to 8 KiB)
// come into a wave-private LDS image with aligned 16-byte loads, and every value is then read at
// its own byte position there (three dwords and two byte-aligns); two 16-byte stores per group.
// Nothing is read more than 15 bytes past the wave tile's data.  (A first form loaded 32 bytes per
// group straight from memory at the group's own address and picked the values out of that window
// with select chains: 0.46 of the HBM peak against 0.60 for this one, 2^27 values of mixed widths.)
// Decodes the 1024 values of wave tile wt into v[step][k] (absent values zero); doff = data
// offset of the wave tile on entry, of the next one on return.
struct __align__(16) DecWaveLds64 {
  u8 data[2 * kWtDataMax + 48];  // +16 alignment shift, +16 partial chunk, +slack for the 12-byte over-read of a value
};
template <bool ZZ, bool NTL>
__device__ __forceinline__ void decode_wavetile64_lds(const u8* __restrict__ in, const u8* __restrict__ data, u64& doff, u32 wt,
                                                      u32 count, int lane, DecWaveLds64& S, u64 (&v)[4][4]) {
  u32 key[4], nvs[4];
  u64 packed = 0;  // the four group lengths (0..32), one 16-bit field per step
#pragma unroll
  for (int step = 0; step < 4; ++step) {
    const u64 g = (u64)wt * kWtKeys + 64 * step + lane;
    const u64 first = 4 * g;
    nvs[step] = first < count ? (u32)min((u64)4, (u64)count - first) : 0u;
    key[step] = nvs[step] ? in[g] : 0u;
    packed |= (u64)group_len64(key[step], nvs[step]) << (16 * step);
  }
  const u32 inc_lo = wave_scan_incl((u32)packed, lane), inc_hi = wave_scan_incl((u32)(packed >> 32), lane);
  const u64 incl = ((u64)inc_hi << 32) | inc_lo;
  const u64 tots = ((u64)wave_last(inc_hi) << 32) | wave_last(inc_lo);
  const u64 excl = incl - packed;
  const u32 t0 = (u32)(tots & 0xFFFFu), t1 = (u32)((tots >> 16) & 0xFFFFu), t2 = (u32)((tots >> 32) & 0xFFFFu),
            t3 = (u32)(tots >> 48);
  const u32 total = t0 + t1 + t2 + t3;
  const u8* dp = data + doff;
  doff += total;
  // ---- stage the span: every load before the first LDS write ----
  const u32 shift = (u32)((uintptr_t)dp & 15u);
  {
    const uint4* src = reinterpret_cast<const uint4*>(dp - shift);
    const u32 nch = max((shift + total + 15u) >> 4, 1u);  // <= 514
    const u32 last = nch - 1u;
    uint4 r[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      r[i] = make_uint4(0, 0, 0, 0);
      if (64u * i < nch) r[i] = ld_stream16<NTL>(src + min((u32)lane + 64u * i, last));
    }
    uint4* d = reinterpret_cast<uint4*>(S.data);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const u32 c = lane + 64u * i;
      if (c < nch) d[c] = r[i];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const u32 pj[4] = {shift + (u32)(excl & 0xFFFFu), shift + t0 + (u32)((excl >> 16) & 0xFFFFu),
                     shift + t0 + t1 + (u32)((excl >> 32) & 0xFFFFu), shift + t0 + t1 + t2 + (u32)(excl >> 48)};
#pragma unroll
  for (int step = 0; step < 4; ++step) {
    u32 p = pj[step];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32 len = (u32)k < nvs[step] ? 1u << ((key[step] >> (2 * k)) & 3u) : 0u;
      const u32* q = reinterpret_cast<const u32*>(S.data + (p & ~3u));
      const u32 a = q[0], b = q[1], c = q[2];
      const u32 lo = __builtin_amdgcn_alignbyte(b, a, p), hi = __builtin_amdgcn_alignbyte(c, b, p);
      u64 x = ((u64)hi << 32) | lo;
      x = len >= 8u ? x : x & ((1ull << (8u * len)) - 1ull);
      if (ZZ) x = len ? zz_dec64(x) : 0ull;
      v[step][k] = x;
      p += len;
    }
  }
  __builtin_amdgcn_wave_barrier();
}
// Differential decode, first pass: the sum (mod 2^64) of the values of each workgroup's four wave tiles.
template <bool ZZ>
__global__ __launch_bounds__(kBlock) void k_value_sums64(const u8* __restrict__ in, u32 count, Dir dir,
                                                         u64* __restrict__ wg_sum) {
  __shared__ u64 s_tot[kWaves];
  __shared__ DecWaveLds64 s_lds[kWaves];
  const int lane = lane_id(), wave = wave_id();
  const u64 keylen = ((u64)count + 3) >> 2;
  const u32 n_wt = (u32)((keylen + kWtKeys - 1) / kWtKeys);
  const u32 wt = blockIdx.x * kWaves + wave;
  u64 tot = 0;
  if (wt < n_wt) {
    u64 doff = dir_offset(dir, wt, lane);
    u64 v[4][4];
    decode_wavetile64_lds<ZZ, false>(in, in + keylen, doff, wt, count, lane, s_lds[wave], v);
#pragma unroll
    for (int step = 0; step < 4; ++step) tot += v[step][0] + v[step][1] + v[step][2] + v[step][3];
    tot = wave_sum64(tot);
  }
  if (lane == 0) s_tot[wave] = tot;
  __syncthreads();
  if (threadIdx.x == 0) wg_sum[blockIdx.x] = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
}
// blk[b] = sum of x[4096 b .. 4096 b + 4095] (the block sums k_multi_offsets starts from).
__global__ __launch_bounds__(1024) void k_block_sums64(const u64* __restrict__ x, u64 n, u64* __restrict__ blk) {
  __shared__ u64 s_wave[16];
  const u64 g0 = ((u64)blockIdx.x * 1024 + threadIdx.x) * 4;
  u64 sum = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) sum += g0 + k < n ? x[g0 + k] : 0ull;
  const u64 w = wave_sum64(sum);
  if (lane_id() == 0) s_wave[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 t = 0;
    for (int i = 0; i < 16; ++i) t += s_wave[i];
    blk[blockIdx.x] = t;
  }
}
// DELTA: wg_base[blockIdx.x] = sum of the values of all earlier workgroups (k_value_sums64, scanned);
// the values become running sums seeded with prev.
template <bool DELTA, bool ZZ, int NT = 0>
__global__ __launch_bounds__(kBlock) void k_decode64(const u8* __restrict__ in, u64* __restrict__ out, u32 count,
                                                     Dir dir, u64* __restrict__ consumed, u64 prev,
                                                     const u64* __restrict__ wg_base) {
  __shared__ u64 s_tot[kWaves];
  __shared__ DecWaveLds64 s_lds[kWaves];
  const int lane = lane_id(), wave = wave_id();
  const u64 keylen = ((u64)count + 3) >> 2;
  const u32 n_wt = (u32)((keylen + kWtKeys - 1) / kWtKeys);
  const u32 wt = blockIdx.x * kWaves + wave;
  if (!DELTA && wt >= n_wt) return;
  u64 v[4][4];
  u64 doff = 0;
  if (wt < n_wt) {
    doff = dir_offset(dir, wt, lane);
    decode_wavetile64_lds<ZZ, (NT & kNtLoads) != 0>(in, in + keylen, doff, wt, count, lane, s_lds[wave], v);
  }
  if (DELTA) {
    u64 carry = 0;  // sum of the steps before, inside this wave tile
    if (wt < n_wt) {
#pragma unroll
      for (int step = 0; step < 4; ++step) {
        v[step][1] += v[step][0]; v[step][2] += v[step][1]; v[step][3] += v[step][2];
        const u64 incl = wave_scan_incl64(v[step][3]);
        const u64 before = carry + incl - v[step][3];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[step][k] += before;
        carry += ((u64)wave_last((u32)(incl >> 32)) << 32) | wave_last((u32)incl);
      }
    }
    if (lane == 0) s_tot[wave] = carry;
    __syncthreads();
    if (wt >= n_wt) return;
    u64 base = prev + wg_base[blockIdx.x];
    for (int w = 0; w < wave; ++w) base += s_tot[w];
#pragma unroll
    for (int step = 0; step < 4; ++step)
#pragma unroll
      for (int k = 0; k < 4; ++k) v[step][k] += base;
  }
#pragma unroll
  for (int step = 0; step < 4; ++step) {
    const u64 first = 4 * ((u64)wt * kWtKeys + 64 * step + lane);
    if (first + 4 <= count) {
      st_stream16<(NT & kNtStores) != 0>(out + first, make_uint4((u32)v[step][0], (u32)(v[step][0] >> 32), (u32)v[step][1], (u32)(v[step][1] >> 32)));
      st_stream16<(NT & kNtStores) != 0>(out + first + 2, make_uint4((u32)v[step][2], (u32)(v[step][2] >> 32), (u32)v[step][3], (u32)(v[step][3] >> 32)));
    } else {
      for (u32 k = 0; k < 4 && first + k < count; ++k) out[first + k] = v[step][k];
    }
  }
  if (wt == n_wt - 1 && lane == 0 && consumed) *consumed = keylen + doff;
}

// The bytes of stream s may be followed directly by the next stream's: the aligned 16-byte loads that stage a wave
// tile reach up to 15 bytes past it, so only the LAST stream of the buffer needs readable bytes behind it (the
// interface asks for SVBGPU_PADDING64).
template <bool DELTA, bool ZZ>
__global__ __launch_bounds__(64 * kBatch64Waves) void k_decode64_batch(const u8* __restrict__ bytes, const u64* __restrict__ byte_off,
                                                           u64* __restrict__ values, const u64* __restrict__ value_off,
                                                           const u32* __restrict__ counts, const u64* __restrict__ prevs,
                                                           u32 nstreams, u64* __restrict__ sizes) {
  __shared__ DecWaveLds64 s_lds[kBatch64Waves];
  const int lane = lane_id(), wave = wave_id();
  const u32 s = blockIdx.x * kBatch64Waves + wave;
  if (s >= nstreams) return;
  const u32 count = counts[s];
  const u8* in = bytes + byte_off[s];
  u64* out = values + value_off[s];
  const u64 keylen = ((u64)count + 3) >> 2;
  const u32 n_wt = (u32)((keylen + kWtKeys - 1) / kWtKeys);
  u64 doff = 0;
  u64 base = (DELTA && prevs) ? prevs[s] : 0ull;  // the value before the wave tile's first
  for (u32 wt = 0; wt < n_wt; ++wt) {
    u64 v[4][4];
    decode_wavetile64_lds<ZZ, false>(in, in + keylen, doff, wt, count, lane, s_lds[wave], v);
    if (DELTA) {
#pragma unroll
      for (int step = 0; step < 4; ++step) {
        v[step][1] += v[step][0]; v[step][2] += v[step][1]; v[step][3] += v[step][2];
        const u64 incl = wave_scan_incl64(v[step][3]);
        const u64 before = base + incl - v[step][3];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[step][k] += before;
        base += ((u64)wave_last((u32)(incl >> 32)) << 32) | wave_last((u32)incl);
      }
    }
#pragma unroll
    for (int step = 0; step < 4; ++step) {
      const u64 first = 4 * ((u64)wt * kWtKeys + 64 * step + lane);
      if (first + 4 <= count) {
        st_stream16<false>(out + first, make_uint4((u32)v[step][0], (u32)(v[step][0] >> 32), (u32)v[step][1], (u32)(v[step][1] >> 32)));
        st_stream16<false>(out + first + 2, make_uint4((u32)v[step][2], (u32)(v[step][2] >> 32), (u32)v[step][3], (u32)(v[step][3] >> 32)));
      } else {
        for (u32 k = 0; k < 4 && first + k < count; ++k) out[first + k] = v[step][k];
      }
    }
  }
  if (lane == 0 && sizes) sizes[s] = keylen + doff;
}

}  // namespace svb

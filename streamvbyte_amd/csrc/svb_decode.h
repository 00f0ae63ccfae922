// Decode path: offset directory (key sums + scan) and the decode kernel.
//
// Replaces the serial `dataPtr += lengthTable[key]` chain of the reference
// (src/streamvbyte_x64_decode.c:10-26,39-112; scalar src/streamvbyte_decode.c:18-60;
// 0124 twin src/streamvbyte_0124_decode.c:17-27,38-79,83-157) with prefix sums.
#pragma once
#include <type_traits>
#include "svb_device.h"

namespace svb {

// ---------------------------------------------------------------------------
// k_key_sums: the offset directory (see Dir) in one pass, one workgroup of 1024
// threads per mega tile.  A round takes 16 super tiles, one per wave: a wave reads
// its super tile (16 KiB of keys = 64 wave tiles) with sixteen 16-byte loads per
// lane, all in flight at once and consumed in the order they land; in load i every
// row of 16 lanes covers one wave tile (number 4 i + row).  A lane's 16 keys of a
// load cost three population counts per dword (dir_counts); the 16 per-load sums
// of a lane are then added up across the row AND dealt out one per lane in the
// same 15 DPP steps (row_reduce_scatter16), one trip through the wave's own LDS
// words puts the 64 tile sums in tile order, and a wave scan places the tiles
// inside the super tile.  The waves meet once per round, to learn what the super
// tiles before theirs add up to.  When the mega tile has another round, every load
// register is refilled right after its last use, so the memory pipe stays full
// across the barrier (which waits for LDS only).  Reads 0.25 B per value, writes
// 4 B per 1024 values; no scan kernel behind it, no atomics, nothing to clear.
constexpr int kDirLoads = 16;  // 16-byte loads per lane and super tile

// v[i], i = 0..15: a value per lane.  Returns, in lane p of every row of 16 lanes, the sum over the
// row of v[dir_dealt(p)]: each halving step keeps one half of the values, hands the other half to the
// partner lane and adds what the partner hands over (15 DPP adds instead of 16 x 4).
__device__ __forceinline__ u32 dir_dealt(u32 p) {  // which v a lane ends up with
  return 8u * (p >> 3) + 4u * ((p >> 2) & 1u) + 2u * (p & 1u) + ((p >> 1) & 1u);
}
template <int CTRL>
__device__ __forceinline__ u32 dpp_get(u32 x) {
  return (u32)__builtin_amdgcn_mov_dpp((int)x, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ u32 row_reduce_scatter16(const u32 (&v)[16], int lane) {
  const bool h1 = lane & 8, h2 = lane & 4, h3 = lane & 1, h4 = lane & 2;
  u32 k8[8], k4[4], k2[2];
#pragma unroll
  for (int j = 0; j < 8; ++j) k8[j] = (h1 ? v[j + 8] : v[j]) + dpp_get<0x140>(h1 ? v[j] : v[j + 8]);  // row_mirror
#pragma unroll
  for (int j = 0; j < 4; ++j) k4[j] = (h2 ? k8[j + 4] : k8[j]) + dpp_get<0x141>(h2 ? k8[j] : k8[j + 4]);  // row_half_mirror
#pragma unroll
  for (int j = 0; j < 2; ++j) k2[j] = (h3 ? k4[j + 2] : k4[j]) + dpp_get<0xB1>(h3 ? k4[j] : k4[j + 2]);  // quad_perm [1,0,3,2]
  return (h4 ? k2[1] : k2[0]) + dpp_get<0x4E>(h4 ? k2[0] : k2[1]);  // quad_perm [2,3,0,1]
}

// clear / clear_dwords: words this launch also sets to zero (the look-back words of a differential decode
// that follows on the same stream; a fill launch of their own costs that call 4-5 us).
template <int V>
__global__ __launch_bounds__(1024) void k_key_sums(const u8* __restrict__ in, u32 count,
                                                   u32* __restrict__ wt_rel, u32* __restrict__ mega_sum,
                                                   u32 supers_per_mega, u32* __restrict__ clear, u32 clear_dwords) {
  constexpr int kW = 16;  // waves per workgroup
  for (u32 i = blockIdx.x * 1024u + threadIdx.x; i < clear_dwords; i += gridDim.x * 1024u) clear[i] = 0u;
  __shared__ u32 s_turn[kW][kSuperWt];
  __shared__ u32 s_tot[2][kW];
  const u64 keylen = ((u64)count + 3) >> 2;
  const u64 n_wt = (keylen + kWtKeys - 1) / kWtKeys;
  const u64 n_super = (n_wt + kSuperWt - 1) / kSuperWt;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // in a scalar register: branches on it are scalar
  const u64 sb0 = (u64)blockIdx.x * supers_per_mega, sb1 = min(n_super, sb0 + supers_per_mega);
  constexpr u64 kSuperKeys = (u64)kSuperWt * kWtKeys;  // 16 KiB
  // Loads are unconditional, at a clamped address (what a lane past the end reads is masked out by its
  // count of valid values): a load under a divergent branch makes the compiler wait for ALL outstanding
  // loads at every use.  May overrun the keys by <16 B: data or padding.  A wave without a super tile in
  // this round re-reads the mega tile's last one (cache hits) rather than another workgroup's keys.
  // (All of it scalar but the lane's 32-bit offset inside the super tile.)
  const u64 koff_last = (keylen - 1) & ~(u64)15;
  auto load = [&](u64 sb, int i) -> uint4 {
    const u64 base = min(sb, sb1 - 1) * kSuperKeys;                     // <= koff_last: that super tile exists
    const u32 room = (u32)min(koff_last - base, (u64)kSuperKeys);        // largest offset that may be read
    return ld_u128(in + base + min((u32)(1024 * i + 16 * lane), room));
  };
  uint4 k[kDirLoads];
  if (sb0 + wave < sb1) {  // scalar: a short stream leaves most waves of its few workgroups without work
#pragma unroll
    for (int i = 0; i < kDirLoads; ++i) k[i] = load(sb0 + wave, i);
  }
  u32 carry = 0;  // data bytes of the mega tile before the current round
  int buf = 0;
  // One round; MORE: the mega tile has another one, whose keys are requested as registers fall free.
  auto round = [&](u64 sbb, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    const u64 sb = sbb + wave;
    u32 v[kDirLoads];
    if (sb >= sb1) {  // scalar: no super tile for this wave in this round (and then in no later one)
#pragma unroll
      for (int i = 0; i < kDirLoads; ++i) v[i] = 0;
    } else if ((sb + 1) * (kSuperKeys * 4) <= (u64)count) {  // scalar: the wave's super tile lacks no value
#pragma unroll
      for (int i = 0; i < kDirLoads; ++i) {
        v[i] = dir_counts<V>(k[i]);
        if (MORE) k[i] = load(sb + kW, i);
      }
    } else {
      // values of the stream from the super tile's first on, at most a super tile's worth (scalar)
      const u64 sfirst = sb * (kSuperKeys * 4);
      const int have = sfirst < count ? (int)min((u64)count - sfirst, kSuperKeys * 4) : 0;
#pragma unroll
      for (int i = 0; i < kDirLoads; ++i) {
        const u32 nv = (u32)min(max(have - (4096 * i + 64 * lane), 0), 64);  // of this lane's 16 keys
        const u32 w[4] = {k[i].x, k[i].y, k[i].z, k[i].w};
        u32 s = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const u32 nvd = nv > 16u * d ? min(nv - 16u * d, 16u) : 0u;
          s += word_len<V>(mask_codes(w[d], nvd), nvd);
        }
        v[i] = s;
        if (MORE) k[i] = load(sb + kW, i);
      }
    }
    const u32 mine = row_reduce_scatter16(v, lane);
    // lane l holds wave tile 4 * dir_dealt(l % 16) + l / 16 of the super tile; turn that into tile l
    s_turn[wave][4u * dir_dealt(lane & 15) + (lane >> 4)] = mine;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const u32 x = s_turn[wave][lane];
    const u32 incl = wave_scan_incl(x, lane);
    if (lane == 63) s_tot[buf][wave] = incl;
    lds_barrier();  // not __syncthreads(): the next round's loads stay in flight
    u32 before = carry, all = 0;
#pragma unroll
    for (int w = 0; w < kW; ++w) {
      const u32 tw = s_tot[buf][w];
      before += w < wave ? tw : 0u;
      all += tw;
    }
    carry += all;
    const u64 wt = sb * kSuperWt + lane;
    if (sb < sb1 && wt < n_wt) wt_rel[wt] = before + incl - x;
    buf ^= 1;  // the buffer written two rounds on is free: the next round's barrier lies in between
  };
  for (u64 sbb = sb0; sbb < sb1; sbb += kW) {
    if (sbb + kW < sb1) round(sbb, std::true_type{});
    else round(sbb, std::false_type{});
  }
  if (threadIdx.x == 0) mega_sum[blockIdx.x] = carry;
}

// ---------------------------------------------------------------------------
// Per-wave LDS: 256 entries (data offset | key<<16) + the wave tile's data span.
// (Measured: 16-bit offsets and key bytes in arrays of their own bring the wave's
// share under 5 KiB and an eighth workgroup onto the CU, but the two narrow LDS
// reads per group instead of one dword cost k_decode 8 %.)
struct __align__(16) WaveLds {
  u32 ent[kWtKeys];
  u8 data[kWtDataMax + 64];  // +16 alignment slack, +16 partial chunk, +slack for 8-byte over-read; the batched decoder also parks a tile's 1024 values plus 16 in it
};

// Blocked phase: lane holds keys 4l..4l+3 in w.  Writes the 4 entries of those
// keys and returns the wave tile's data byte count.
// A/B build only (-DSVB_LENGTH_TABLE_LDS, tools/ab_length_table.sh): the reference's 256-entry
// length table (src/streamvbyte_shuffle_tables_decode.h:5-22, _0124_decode.h:2-19 there) resident in
// LDS, one byte per key, filled by k_decode's 256 threads at kernel start.  The product evaluates the
// same table arithmetically, four keys per instruction (key_lens4); the measurement that decided it is
// recorded with the kernel's description.
template <int V>
__device__ __forceinline__ u8* lds_len_table() {
  __shared__ u8 tab[256];
  return tab;
}
template <int V>
__device__ __forceinline__ u32 key_lens4_table(u32 w_masked, u32 nv) {
  const u8* tab = lds_len_table<V>();
  u32 L = (u32)tab[w_masked & 0xFFu] | ((u32)tab[(w_masked >> 8) & 0xFFu] << 8) |
          ((u32)tab[(w_masked >> 16) & 0xFFu] << 16) | ((u32)tab[w_masked >> 24] << 24);
  // the table counts four values per key; a 1234 key with absent values counts one byte each too many
  if (V == V1234) L -= 0x04040404u - valid_per_key4(nv);
  return L;
}

template <int V, bool TABLE = false>
__device__ __forceinline__ u32 scan_keys_to_entries(u32 w, u32 nv, int lane, u32* ent) {
  w = mask_codes(w, nv);
  const u32 L = TABLE ? key_lens4_table<V>(w, nv) : key_lens4<V>(w, nv);  // 4 packed lengths
  const u32 tot = (L * 0x01010101u) >> 24;            // their sum (<= 64)
  const u32 P = L * 0x01010100u;                      // packed exclusive prefix inside the lane
  const u32 incl = wave_scan_incl(tot, lane);
  const u32 ex = incl - tot;
  uint4 e;
  e.x = (ex) | ((w & 0xFFu) << 16);
  e.y = (ex + ((P >> 8) & 0xFFu)) | (((w >> 8) & 0xFFu) << 16);
  e.z = (ex + ((P >> 16) & 0xFFu)) | (((w >> 16) & 0xFFu) << 16);
  e.w = (ex + (P >> 24)) | ((w >> 24) << 16);
  *reinterpret_cast<uint4*>(ent + 4 * lane) = e;
  return wave_last(incl);
}

// Stage the contiguous data span [dp, dp+total) into LDS with aligned 16-byte
// loads.  Returns the offset of dp's first byte inside the LDS image.
template <bool NTL = false>
__device__ __forceinline__ u32 stage_span(const u8* dp, u32 total, int lane, u8* sdata) {
  const u32 shift = (u32)((uintptr_t)dp & 15u);
  const uint4* src = reinterpret_cast<const uint4*>(dp - shift);
  const u32 nch = (shift + total + 15u) >> 4;  // <= 257
  // Issue every load before the first LDS write so that up to five are in
  // flight.  Lanes past the end re-read the last chunk (same cache line) so
  // that the loads stay unconditional within a wave-uniform branch.
  uint4 r[5];
  const u32 last = nch - 1u;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    r[i] = make_uint4(0, 0, 0, 0);
    if (64u * i < nch) r[i] = ld_stream16<NTL>(src + min((u32)lane + 64u * i, last));
  }
  uint4* d = reinterpret_cast<uint4*>(sdata);
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const u32 c = lane + 64u * i;
    if (c < nch) d[c] = r[i];
  }
  return shift;
}

// Striped phase: decode group g = j*64 + lane from the LDS image.
template <int V>
__device__ __forceinline__ uint4 extract_group(u32 entry, u32 shift, const u8* sdata) {
  const u32 key = entry >> 16;
  u32 p = (entry & 0xFFFFu) + shift;
  u32 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const u32 c = (key >> (2 * k)) & 3u;
#ifdef SVB_UNALIGNED_LDS
    v[k] = lds_u32(sdata + p) & mask_of<V>(c);
#else
    v[k] = lds_u32_aligned(sdata, p) & mask_of<V>(c);
#endif
    p += len_of<V>(c);
  }
  return make_uint4(v[0], v[1], v[2], v[3]);
}

// One wave tile, keys -> values: lane ends up with the (up to) 4 groups
// j*64+lane, j = 0..3, of the wave tile whose keys it holds blocked in `w`.
// Values past `nints` are returned as zero.  Returns the tile's data bytes.
template <int V, bool ZZ, bool SHORTCUT = true, bool TABLE = false, bool NTL = false>
__device__ __forceinline__ u32 decode_wavetile(u32 w, u32 nints, const u8* dp, int lane, WaveLds& S,
                                               uint4 (&v)[4]) {
  // A full wave tile whose 256 keys are all zero (every value below 256; dense
  // posting-list gaps are the common case, and the reference has the same
  // shortcut, src/streamvbytedelta_x64_decode.c:87-100): 1234 data bytes ARE the
  // values, group j*64+lane is the dword at byte 4*(j*64+lane); 0124 has no data
  // bytes at all.  No scan, no LDS.
  if (SHORTCUT && nints == kWtInts && __ballot(w != 0u) == 0ull) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint4 r = make_uint4(0, 0, 0, 0);
      if (V == V1234) {
        const u32 d = ld_u32(dp + 4u * (j * 64 + lane));
        r = make_uint4(d & 0xFFu, (d >> 8) & 0xFFu, (d >> 16) & 0xFFu, d >> 24);
        if (ZZ) r = make_uint4(zz_dec(r.x), zz_dec(r.y), zz_dec(r.z), zz_dec(r.w));
      }
      v[j] = r;
    }
    return V == V1234 ? (u32)kWtInts : 0u;
  }
  const u32 nv = nints > 16u * lane ? min(nints - 16u * lane, 16u) : 0u;
  const u32 total = scan_keys_to_entries<V, TABLE>(w, nv, lane, S.ent);
  const u32 shift = stage_span<NTL>(dp, total, lane, S.data);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32 first = 4u * (j * 64 + lane);
    uint4 r = make_uint4(0, 0, 0, 0);
    if (first < nints) {
      r = extract_group<V>(S.ent[j * 64 + lane], shift, S.data);
      if (ZZ) r = make_uint4(zz_dec(r.x), zz_dec(r.y), zz_dec(r.z), zz_dec(r.w));
      const u32 left = nints - first;
      if (left < 4) { r.w = 0; if (left < 3) r.z = 0; if (left < 2) r.y = 0; }
    }
    v[j] = r;
  }
  __builtin_amdgcn_wave_barrier();
  return total;
}

// decode_wavetile in two halves, for kernels whose waves take several wave tiles: the data loads of ALL of
// them are sent before the first is used (a wave tile at a time, a wave has one tile's span in flight per
// round trip).  span_issue requests the span [dp, dp + total) -- total from wavetile_data_bytes -- into K
// registers per lane, decode_wavetile_landed scans the keys, drops the registers into the wave's LDS image
// and extracts.  Spans of more than K * 64 chunks of 16 bytes do not fit: kSpanFits says so, and the caller
// decodes such a wave tile with decode_wavetile.
template <int V>
__device__ __forceinline__ u32 wavetile_data_bytes(u32 w, u32 nints, int lane) {
  const u32 nv = nints > 16u * lane ? min(nints - 16u * lane, 16u) : 0u;
  return wave_sum(word_len<V>(mask_codes(w, nv), nv));
}
template <int K>
struct SpanRegs {
  uint4 r[K];
  u32 shift, nch;
};
template <int K>
__device__ __forceinline__ bool span_fits(const u8* dp, u32 total) {
  return (((u32)((uintptr_t)dp & 15u) + total + 15u) >> 4) <= 64u * K;
}
template <int K, bool NTL>
__device__ __forceinline__ void span_issue(const u8* dp, u32 total, int lane, SpanRegs<K>& R) {
  R.shift = (u32)((uintptr_t)dp & 15u);
  const uint4* src = reinterpret_cast<const uint4*>(dp - R.shift);
  R.nch = (R.shift + total + 15u) >> 4;
  const u32 last = R.nch - 1u;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    R.r[i] = make_uint4(0, 0, 0, 0);
    if (64u * i < R.nch) R.r[i] = ld_stream16<NTL>(src + min((u32)lane + 64u * i, last));
  }
}
template <int V, bool ZZ, int K>
__device__ __forceinline__ u32 decode_wavetile_landed(u32 w, u32 nints, const SpanRegs<K>& R, int lane, WaveLds& S,
                                                      uint4 (&v)[4]) {
  const u32 nv = nints > 16u * lane ? min(nints - 16u * lane, 16u) : 0u;
  const u32 total = scan_keys_to_entries<V, false>(w, nv, lane, S.ent);
  uint4* d = reinterpret_cast<uint4*>(S.data);
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const u32 c = lane + 64u * i;
    if (c < R.nch) d[c] = R.r[i];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32 first = 4u * (j * 64 + lane);
    uint4 r = make_uint4(0, 0, 0, 0);
    if (first < nints) {
      r = extract_group<V>(S.ent[j * 64 + lane], R.shift, S.data);
      if (ZZ) r = make_uint4(zz_dec(r.x), zz_dec(r.y), zz_dec(r.z), zz_dec(r.w));
      const u32 left = nints - first;
      if (left < 4) { r.w = 0; if (left < 3) r.z = 0; if (left < 2) r.y = 0; }
    }
    v[j] = r;
  }
  __builtin_amdgcn_wave_barrier();
  return total;
}

// ---- a group straight from registers (no LDS) ----------------------------------
// bytes [o, o+4) of the 16-byte little-endian value d (o = 0..12), zero beyond
__device__ __forceinline__ u32 bytes_at(const uint4& d, u32 o) {
  const u32 i = o >> 2;
  const u32 lo = i == 0 ? d.x : (i == 1 ? d.y : (i == 2 ? d.z : d.w));
  const u32 hi = i == 0 ? d.y : (i == 1 ? d.z : (i == 2 ? d.w : 0u));
  return __builtin_amdgcn_alignbyte(hi, lo, o);
}
// data bytes of the group coded by `key`, whose first nv (0..4) values exist
template <int V>
__device__ __forceinline__ u32 group_len(u32 key, u32 nv) {
  u32 L = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) L += (u32)k < nv ? len_of<V>((key >> (2 * k)) & 3u) : 0u;
  return L;
}
// The four values of a group whose data bytes start at byte 0 of d (the 16 bytes
// loaded at the group's own byte address); absent values are zero.
template <int V, bool ZZ>
__device__ __forceinline__ uint4 values_of_group(const uint4& d, u32 key, u32 nv) {
  u32 l[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) l[k] = (u32)k < nv ? len_of<V>((key >> (2 * k)) & 3u) : 0u;
  uint4 c;
  c.x = bytes_at(d, 0) & (l[0] ? mask_of<V>(key & 3u) : 0u);
  c.y = bytes_at(d, l[0]) & (l[1] ? mask_of<V>((key >> 2) & 3u) : 0u);
  c.z = bytes_at(d, l[0] + l[1]) & (l[2] ? mask_of<V>((key >> 4) & 3u) : 0u);
  c.w = bytes_at(d, l[0] + l[1] + l[2]) & (l[3] ? mask_of<V>((key >> 6) & 3u) : 0u);
  if (ZZ) {
    c = make_uint4(zz_dec(c.x), zz_dec(c.y), zz_dec(c.z), zz_dec(c.w));
    if (nv < 4u) { c.w = 0; if (nv < 3u) c.z = 0; if (nv < 2u) c.y = 0; if (nv < 1u) c.x = 0; }
  }
  return c;
}

// Store group `first..first+3` of a stream of `count` values.
template <bool NTS = false>
__device__ __forceinline__ void store_group(u32* out, u64 first, u64 count, const uint4& v) {
  if (first + 4 <= count) {
    st_stream16<NTS>(out + first, v);
  } else if (first < count) {
    out[first] = v.x;
    if (first + 1 < count) out[first + 1] = v.y;
    if (first + 2 < count) out[first + 2] = v.z;
  }
}

// ---------------------------------------------------------------------------
// k_decode: every wave takes wave tile blockIdx.x * kWaves + wave; the launch
// normally has a workgroup per 4 wave tiles (fastest: the workgroups in flight
// cover one advancing stretch of the stream).  On a smaller grid a wave goes on
// with a grid stride, the next tile's keys and offset prefetched.  No
// inter-workgroup dependency: data offsets come from the directory built by
// k_key_sums.  HBM traffic per value: 0.25 (keys) + dbar (data)
// + 4 (out).
// OUT: 0 = the integers; 1 = float32 (int32)value * scale (dequantise fused into the
// store); 2 = the same added to what out already holds (dequantise-accumulate).
// scale travels as the bits of `aux`.  NT: non-temporal stream loads (bit 0) / value stores (bit 1).
template <int V, bool ZZ, bool SHORTCUT, int OUT = 0, int NT = 0>
__global__ __launch_bounds__(kBlock) void k_decode(const u8* __restrict__ in,
                                                   u32* __restrict__ out, u32 count,
                                                   Dir dir, u64* __restrict__ consumed, u32 aux) {
  __shared__ WaveLds s_lds[kWaves];
  const int lane = lane_id(), wave = wave_id();
  WaveLds& S = s_lds[wave];
#ifdef SVB_LENGTH_TABLE_LDS
  constexpr bool kTable = true;
  {
    const u32 key = threadIdx.x;  // kBlock == 256: one table entry per thread
    u32 len = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) len += len_of<V>((key >> (2 * k)) & 3u);
    lds_len_table<V>()[key] = (u8)len;
    __syncthreads();
  }
#else
  constexpr bool kTable = false;
#endif
  const u64 keylen = ((u64)count + 3) >> 2;
  const u32 n_wt = (u32)((keylen + kWtKeys - 1) / kWtKeys);
  const u32 stride = gridDim.x * kWaves;
  u32 wt = blockIdx.x * kWaves + wave;
  if (wt >= n_wt) return;
  const u8* data_base = in + keylen;

  auto load_keys = [&](u32 t) -> u32 {
    const u64 koff = (u64)t * kWtKeys + 4 * lane;
    return koff < keylen ? ld_u32(in + koff) : 0u;
  };
  u32 w_next = load_keys(wt);
  u64 off_next = dir_offset(dir, wt, lane);

  while (true) {
    const u32 w = w_next;
    const u64 doff = off_next;
    const u32 wt_n = wt + stride;
    if (wt_n < n_wt) {
      w_next = load_keys(wt_n);
      off_next = dir_offset(dir, wt_n, lane);
    }
    const u64 int0 = (u64)wt * kWtInts;
    const u32 nints = (u32)min((u64)kWtInts, (u64)count - int0);
    uint4 v[4];
    const u32 total = decode_wavetile<V, ZZ, SHORTCUT, kTable, (NT & kNtLoads) != 0>(w, nints, data_base + doff, lane, S, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u64 first = int0 + 4u * (j * 64 + lane);
      uint4 r = v[j];
      if (OUT != 0) {
        const float scale = __uint_as_float(aux);
        // __fmul_rn / __fadd_rn: product and sum are rounded separately, never fused
        float f[4] = {__fmul_rn((float)(int)r.x, scale), __fmul_rn((float)(int)r.y, scale),
                      __fmul_rn((float)(int)r.z, scale), __fmul_rn((float)(int)r.w, scale)};
        if (OUT == 2) {
          const float* acc = reinterpret_cast<const float*>(out);
          if (first + 4 <= count) {
            float4 a;
            __builtin_memcpy(&a, acc + first, 16);
            f[0] = __fadd_rn(f[0], a.x); f[1] = __fadd_rn(f[1], a.y); f[2] = __fadd_rn(f[2], a.z); f[3] = __fadd_rn(f[3], a.w);
          } else {
#pragma unroll
            for (int k = 0; k < 3; ++k)
              if (first + k < count) f[k] = __fadd_rn(f[k], acc[first + k]);
          }
        }
        r = make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
      }
      store_group<(NT & kNtStores) != 0>(out, first, count, r);
    }
    if (wt == n_wt - 1 && lane == 0 && consumed) *consumed = keylen + doff + total;
    if (wt_n >= n_wt) break;
    wt = wt_n;
  }
}

}  // namespace svb

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


// ===========================================================================
// Chunked batch kernels: a wavefront takes K consecutive streams (K <= 64), lane
// i holds the directory entry of stream i of the chunk (three coalesced loads
// for up to 64 streams instead of dependent scalar loads per stream), and the
// chunk is worked off in two parts:
//
//  * FLAT: all streams of up to kFlatMax values together, one GROUP per lane
//    and step whatever stream it belongs to.  The groups of the chunk's short
//    streams are numbered consecutively ("slots"); a step covers 64 slots.  A
//    64-bit mask of the slots where a stream starts gives every lane its stream
//    (population count below the lane), a SEGMENTED scan over the lanes (key =
//    the stream) turns group lengths into data offsets and, for differential
//    coding, group sums into running values; a stream cut by a step's end is
//    carried into the next step.  Lanes are full whatever the length mix:
//    streams of 3 values and of 200 values cost the same per group.
//  * LONG: streams beyond kFlatMax values one after the other, the whole
//    wavefront on one wave tile at a time as in the single-stream kernels, the
//    next wave tile's keys requested before the current one is decoded.
//
// No workspace, no pass over the whole directory per length class, one launch.
// Per-stream results equal the single-stream calls (reference
// src/streamvbyte_decode.c:64-87, src/streamvbyte_encode.c:113-137 list by list).
constexpr u32 kFlatMax = 256;  // values; such a stream has at most 64 groups
// Wavefronts per workgroup of the chunked kernels (they share nothing).  Measured on the posting-list mix,
// same box: one wave per workgroup, which hands a slot back the moment its chunk ends, decodes 1 % faster
// and encodes 1 % slower than four -- the hardware backfills well enough either way.
#ifndef SVB_BATCH_WAVES
#define SVB_BATCH_WAVES 4
#endif
constexpr int kBatchWaves = SVB_BATCH_WAVES;
// 1: the long part of the chunked decoder stores the whole wave tiles of a stream whose values do not start on a
// 64-byte boundary as aligned quarters, by way of the LDS image (see there).  Same box, alternating libraries, on
// 2^20 posting lists laid back to back: 0.508-0.510 -> 0.519-0.523 of peak; the long lists alone 0.521 -> 0.543.
// The stream's last, partial tile taken the same way added nothing and is not built.
#ifndef SVB_BATCH_REALIGN
#define SVB_BATCH_REALIGN 1
#endif

// Directory entry of a short stream, compacted to the front of the wave's LDS.
struct __align__(16) FlatEnt {
  u64 boff, voff;
  u32 cnt, first_slot, prev, sid;
};
static_assert(sizeof(FlatEnt) == 32 && 64 * sizeof(FlatEnt) <= kWtDataMax, "entries alias the wave's data image");

// What a lane needs to know about its slot in one step of the flat part.
struct FlatSlot {
  FlatEnt e;    // the slot's stream
  u32 idx;      // index of that stream among the chunk's short streams (the segment key)
  u32 gi;       // group index inside the stream
  bool have;    // the slot exists
};
// Compacts the short streams' entries into `ent` and returns the number of slots.
__device__ __forceinline__ u32 flat_setup(bool is_short, u32 cnt, u64 boff, u64 voff, u32 prev, u32 sid, int lane,
                                          FlatEnt* ent, u32& n_short, u32& my_first) {
  const u32 ng = is_short ? (cnt + 3u) >> 2 : 0u;
  const u32 incl = wave_scan_incl(ng, lane);
  const u64 smask = __ballot(is_short);
  const u32 rank = (u32)__popcll(smask & ((1ull << lane) - 1ull));
  if (is_short) {
    FlatEnt e;
    e.boff = boff; e.voff = voff; e.cnt = cnt; e.first_slot = incl - ng; e.prev = prev; e.sid = sid;
    ent[rank] = e;
  }
  n_short = (u32)__popcll(smask);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  my_first = (u32)lane < n_short ? ent[lane].first_slot : 0xFFFFFFFFu;
  return wave_last(incl);
}
__device__ __forceinline__ FlatSlot flat_slot(u32 base, u32 n_slots, u32 n_short, u32 my_first, int lane,
                                              const FlatEnt* ent) {
  // bit h of the mask: a stream starts at slot base + h
  const u32 h = my_first - base;
  const bool head_here = my_first != 0xFFFFFFFFu && my_first >= base && h < 64u;
  const u32 m_lo = wave_or(head_here && h < 32u ? 1u << h : 0u);
  const u32 m_hi = wave_or(head_here && h >= 32u ? 1u << (h - 32u) : 0u);
  const u64 heads = ((u64)m_hi << 32) | m_lo;
  const u32 before = (u32)__popcll(__ballot(my_first < base));  // streams that started in earlier steps
  FlatSlot s;
  s.have = base + lane < n_slots;
  const u64 upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
  u32 idx = before + (u32)__popcll(heads & upto) - 1u;
  s.idx = min(idx, n_short - 1u);  // slots past the end borrow the last stream's entry and do nothing
  s.e = ent[s.idx];
  s.gi = base + lane - s.e.first_slot;
  return s;
}

// Cache policy of a batched call, chosen on the DEVICE: the directory lives in device memory, so the host
// does not know how many bytes a batch moves.  Every wavefront takes the extent of the value array from
// the first and the last directory entry (three uniform loads that hit in cache) and applies the rule the
// single-stream calls apply on the host: non-temporal stores from `nt_bytes` written, loads as well when
// the input is that long and the footprint passes 2.5 x nt_bytes; the streams count with the least size
// they can have.  It is a policy only: a directory that is not laid out in order gets a wrong guess and
// the same results.  nt_force >= 0 (options nt_force, or 0 when the policy is off) skips the guess.
template <int V>
__device__ __forceinline__ int batch_nt_policy(int nt_force, u64 nt_bytes, const u64* __restrict__ value_off,
                                               const u32* __restrict__ counts, u32 nstreams, bool encode) {
  if (nt_force >= 0) return nt_force;
  const u64 v0 = value_off[0], v1 = value_off[nstreams - 1] + counts[nstreams - 1];
  const u64 nvals = v1 > v0 ? v1 - v0 : v0 - v1;
  const u64 vbytes = 4 * nvals, sbytes = ((nvals + 3) >> 2) + (V == V1234 ? nvals : 0);
  const u64 ld = encode ? vbytes : sbytes, st = encode ? sbytes : vbytes;
  return (ld >= nt_bytes && 2 * (ld + st) >= 5 * nt_bytes ? kNtLoads : 0) | (st >= nt_bytes ? kNtStores : 0);
}

// Blocks [0, n_long_blocks) work off the LONG streams, K_long of the batch's streams per wavefront;
// the blocks behind them the FLAT part, 64 streams per wavefront (few streams per wavefront keep the
// long-running wavefronts similar in length and the launch's tail short; many keep the flat steps full).
// NT: bit 0 the streams' data bytes of the long part, bit 1 the values (see ld_stream16 / st_stream16).
template <int V, bool DELTA, bool ZZ, int NT>
__device__ __forceinline__ void decode_batch2_body(
    const u8* __restrict__ bytes, const u64* __restrict__ byte_off, u32* __restrict__ values,
    const u64* __restrict__ value_off, const u32* __restrict__ counts,
    const u32* __restrict__ prevs, u32 nstreams, u64* __restrict__ sizes, u32 K_long, u32 n_long_blocks,
    WaveLds& S) {
  const int lane = lane_id(), wave = wave_id();
  const bool flat_role = blockIdx.x >= n_long_blocks;
  const u32 K = flat_role ? 64u : K_long;
  const u64 s0 = ((u64)(flat_role ? blockIdx.x - n_long_blocks : blockIdx.x) * kBatchWaves + wave) * K;
  if (s0 >= nstreams) return;  // wave-uniform; no workgroup barrier below
  const bool valid = (u32)lane < K && s0 + lane < nstreams;
  const u32 sid = (u32)(s0 + lane);
  u32 cnt = 0, prev = 0;
  u64 boff = 0, voff = 0;
  if (valid) {
    cnt = counts[sid];
    boff = byte_off[sid];
    voff = value_off[sid];
    if (DELTA && prevs) prev = prevs[sid];
  }

  if (flat_role) {
    if (valid && cnt == 0 && sizes) sizes[sid] = 0;
    const bool is_short = valid && cnt > 0 && cnt <= kFlatMax;
    FlatEnt* ent = reinterpret_cast<FlatEnt*>(S.data);
    u32 n_short, my_first;
    const u32 n_slots = flat_setup(is_short, cnt, boff, voff, prev, sid, lane, ent, n_short, my_first);
    u32 carry_idx = 0xFFFFFFFFu, carry_doff = 0, carry_val = 0;  // wave-uniform: the stream cut by the last step's end
    if (n_slots == 0) return;
    // A step's slots and key bytes depend on the directory alone, so they are worked out and requested a
    // step ahead (one unconditional byte load at one place; lanes without a slot read their borrowed
    // stream's first key): a step then waits for its data bytes only, not for its keys and then its data.
    auto key_address = [&](const FlatSlot& t) -> const u8* { return bytes + t.e.boff + (t.have ? t.gi : 0u); };
    FlatSlot f_next = flat_slot(0, n_slots, n_short, my_first, lane, ent);
    u32 key_next = *key_address(f_next);
    for (u32 base = 0; base < n_slots; base += 64) {
      const FlatSlot f = f_next;
      const u32 key_raw = key_next;
      f_next = flat_slot(base + 64, n_slots, n_short, my_first, lane, ent);  // past the end: no lane has a slot
      key_next = *key_address(f_next);
      const u8* in = bytes + f.e.boff;
      const u32 keylen = (f.e.cnt + 3u) >> 2;
      u32 key = f.have ? key_raw : 0u;
      const u32 nv = f.have ? min(4u, f.e.cnt - 4u * f.gi) : 0u;
      key &= nv >= 4u ? 0xFFu : ((1u << (2u * nv)) - 1u);
      const u32 L = group_len<V>(key, nv);
      const u32 incl = seg_add_scan(f.idx, L) + (f.idx == carry_idx ? carry_doff : 0u);
      uint4 d = make_uint4(0, 0, 0, 0);
      if (f.have) d = ld_u128(in + keylen + (incl - L));
      uint4 c = values_of_group<V, ZZ>(d, key, nv);
      if (DELTA) {
        c.y += c.x; c.z += c.y; c.w += c.z;
        const u32 si = seg_add_scan(f.idx, c.w) + (f.idx == carry_idx ? carry_val : f.e.prev);
        const u32 ex = si - c.w;
        c.x += ex; c.y += ex; c.z += ex; c.w += ex;
      }
      if (f.have) {
        store_group<(NT & kNtStores) != 0>(values + f.e.voff, 4ull * f.gi, f.e.cnt, c);
        if (sizes && f.gi == keylen - 1u) sizes[f.e.sid] = (u64)keylen + incl;
      }
      carry_idx = (u32)__builtin_amdgcn_readlane((int)f.idx, 63);
      carry_doff = (u32)__builtin_amdgcn_readlane((int)incl, 63);
      if (DELTA) carry_val = (u32)__builtin_amdgcn_readlane((int)c.w, 63);
    }
    return;
  }

  // ---- long streams, one after the other, the whole wavefront on a wave tile at a time ----
  // One loop over (stream, wave tile) items with ONE key load per round, issued at one place: the keys
  // of the item after the current one, which is the stream's next wave tile or the first wave tile of the
  // chunk's next long stream, travel while the current item is decoded and stored.  The load is
  // unconditional (lanes past the keys re-read the last key dword; the tile decoder masks what lies past
  // the tile's values anyway), so nothing between the load and its use a round later needs its result.
  u64 todo = __ballot(valid && cnt > kFlatMax);
  if (!todo) return;
  auto of_lane = [&](u64 x, int l) -> u64 {
    return ((u64)(u32)__builtin_amdgcn_readlane((int)(x >> 32), l) << 32) | (u32)__builtin_amdgcn_readlane((int)x, l);
  };
  auto load_keys = [&](const u8* kaddr, u32 klim) -> u32 {  // klim >= 1 keys from kaddr on
    return ld_u32(kaddr + min(4u * (u32)lane, (klim - 1u) & ~3u));  // may overrun the keys by < 4 B: data or padding
  };
  int src = (int)__builtin_ctzll(todo);
  todo &= todo - 1ull;
  u32 count = (u32)__builtin_amdgcn_readlane((int)cnt, src);
  const u8* in = bytes + of_lane(boff, src);
  u32* out = values + of_lane(voff, src);
  u32 run = (u32)__builtin_amdgcn_readlane((int)prev, src);
  u32 keylen = (u32)(((u64)count + 3) >> 2);  // count > kFlatMax: at least 65 keys
  u32 wt = 0;
  u64 doff = 0;
  u32 w_raw = load_keys(in, keylen);
#if SVB_BATCH_REALIGN
  bool have_carry = false;  // wave-uniform: cr holds values of the current stream that are not stored yet
  u32 cr = 0;
#endif
  for (;;) {
    const u32 w = w_raw;
    const u32 keys_left = keylen - wt * (u32)kWtKeys;  // from this wave tile on
    const bool last_tile = keys_left <= (u32)kWtKeys;
    int nsrc = -1;
    {
      const u8* kaddr = in;  // nothing follows: a harmless re-read of this stream's first keys
      u32 klim = keylen;
      if (!last_tile) {
        kaddr = in + (u64)(wt + 1u) * kWtKeys;
        klim = keys_left - (u32)kWtKeys;
      } else if (todo) {
        nsrc = (int)__builtin_ctzll(todo);
        kaddr = bytes + of_lane(boff, nsrc);
        klim = (u32)(((u64)(u32)__builtin_amdgcn_readlane((int)cnt, nsrc) + 3) >> 2);
      }
      w_raw = load_keys(kaddr, klim);
    }
    const u64 int0 = (u64)wt * kWtInts;
    uint4 v[4];
    auto integrate = [&](uint4 c) -> uint4 {  // differences -> values, `run` carried along
      if (DELTA) {
        c.y += c.x; c.z += c.y; c.w += c.z;
        const u32 incl = wave_scan_incl(c.w, lane);
        const u32 ex = incl - c.w + run;
        run += wave_last(incl);
        c.x += ex; c.y += ex; c.z += ex; c.w += ex;
      }
      return c;
    };
    if (int0 + kWtInts <= count) {
      // a whole wave tile (wave-uniform): the length is a constant, so every bounds check of the
      // tile decoder and of the stores folds away
      doff += decode_wavetile<V, ZZ, true, false, (NT & kNtLoads) != 0>(w, (u32)kWtInts, in + keylen + doff, lane, S, v);
#if SVB_BATCH_REALIGN
      const u32 a = (u32)(((uintptr_t)out >> 2) & 15u);  // dwords from the 64-byte boundary below the stream's first value
      if (a != 0u) {
        // The stream's values do not start on a 64-byte boundary: a quarter's 1 KiB would straddle one in every
        // four lanes.  The tile's values take a trip through the (now free) LDS image, displaced by `a` dwords,
        // and leave as four ALIGNED quarters; the `a` values that hang over wait in a register for the next tile.
        uint4 c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = integrate(v[j]);
        u32* B = reinterpret_cast<u32*>(S.data);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          u32* q = B + a + 256u * j + 4u * (u32)lane;
          q[0] = c[j].x; q[1] = c[j].y; q[2] = c[j].z; q[3] = c[j].w;
        }
        if (have_carry && (u32)lane < a) B[lane] = cr;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        u32* o = out + int0 - a;  // 64-byte aligned; below the stream's first value in its first tile
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const u32 p = 256u * k + 4u * (u32)lane;
          const uint4 q = *reinterpret_cast<const uint4*>(B + p);
          if (k == 0 && !have_carry && p < a) {  // the stream's first tile: positions below a lie before the stream
            if (p + 1u >= a) o[p + 1u] = q.y;
            if (p + 2u >= a) o[p + 2u] = q.z;
            if (p + 3u >= a) o[p + 3u] = q.w;
          } else {
            st_stream16<(NT & kNtStores) != 0>(o + p, q);
          }
        }
        cr = lane < 16 ? B[1024 + lane] : 0u;
        __builtin_amdgcn_wave_barrier();
        have_carry = true;
        if (last_tile && (u32)lane < a) out[int0 + kWtInts - a + (u32)lane] = cr;  // the stream ends with this tile
      } else
#endif
      {
      u32* o = out + int0;
#pragma unroll
      for (int j = 0; j < 4; ++j) st_stream16<(NT & kNtStores) != 0>(o + 4u * (u32)(j * 64 + lane), integrate(v[j]));
      }
    } else {
      const u32 nints = (u32)((u64)count - int0);
#if SVB_BATCH_REALIGN
      const u32 a = (u32)(((uintptr_t)out >> 2) & 15u);
      if (have_carry && (u32)lane < a) out[int0 - a + (u32)lane] = cr;  // what hung over the last whole tile
#endif
      doff += decode_wavetile<V, ZZ, true, false, (NT & kNtLoads) != 0>(w, nints, in + keylen + doff, lane, S, v);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (256u * j >= nints) break;  // wave-uniform: the stream's last wave tile ends before this quarter
        store_group<(NT & kNtStores) != 0>(out, int0 + 4u * (j * 64 + lane), count, integrate(v[j]));
      }
    }
    if (!last_tile) {
      ++wt;
      continue;
    }
    if (sizes && lane == 0) sizes[s0 + src] = (u64)keylen + doff;
    if (nsrc < 0) break;
    src = nsrc;
    todo &= todo - 1ull;
    count = (u32)__builtin_amdgcn_readlane((int)cnt, src);
    in = bytes + of_lane(boff, src);
    out = values + of_lane(voff, src);
    run = (u32)__builtin_amdgcn_readlane((int)prev, src);
    keylen = (u32)(((u64)count + 3) >> 2);
    wt = 0;
    doff = 0;
#if SVB_BATCH_REALIGN
    have_carry = false;
#endif
  }
}

template <int V, bool DELTA, bool ZZ>
__global__ __launch_bounds__(64 * kBatchWaves, 7) void k_decode_batch2(
    const u8* __restrict__ bytes, const u64* __restrict__ byte_off, u32* __restrict__ values,
    const u64* __restrict__ value_off, const u32* __restrict__ counts,
    const u32* __restrict__ prevs, u32 nstreams, u64* __restrict__ sizes, u32 K_long, u32 n_long_blocks,
    int nt_force, u64 nt_bytes) {
  __shared__ WaveLds s_lds[kBatchWaves];
  WaveLds& S = s_lds[wave_id()];
  const int nt = batch_nt_policy<V>(nt_force, nt_bytes, value_off, counts, nstreams, false);
  if (nt == 3)
    decode_batch2_body<V, DELTA, ZZ, 3>(bytes, byte_off, values, value_off, counts, prevs, nstreams, sizes, K_long, n_long_blocks, S);
  else if (nt == 2)
    decode_batch2_body<V, DELTA, ZZ, 2>(bytes, byte_off, values, value_off, counts, prevs, nstreams, sizes, K_long, n_long_blocks, S);
  else
    decode_batch2_body<V, DELTA, ZZ, 0>(bytes, byte_off, values, value_off, counts, prevs, nstreams, sizes, K_long, n_long_blocks, S);
}

// NT: bit 0 the values, bit 1 the streams' bytes of the long part.
template <int V, bool DELTA, bool ZZ, int NT>
__device__ __forceinline__ void encode_batch2_body(
    const u32* __restrict__ values, const u64* __restrict__ value_off, u8* __restrict__ bytes,
    const u64* __restrict__ byte_off, const u32* __restrict__ counts,
    const u32* __restrict__ prevs, u32 nstreams, u64* __restrict__ sizes, u32 K_long, u32 n_long_blocks,
    EncWaveLds& S) {
  const int lane = lane_id(), wave = wave_id();
  const bool flat_role = blockIdx.x >= n_long_blocks;  // as in k_decode_batch2
  const u32 K = flat_role ? 64u : K_long;
  const u64 s0 = ((u64)(flat_role ? blockIdx.x - n_long_blocks : blockIdx.x) * kBatchWaves + wave) * K;
  if (s0 >= nstreams) return;  // wave-uniform; no workgroup barrier below
  const bool valid = (u32)lane < K && s0 + lane < nstreams;
  const u32 sid = (u32)(s0 + lane);
  u32 cnt = 0, prev = 0;
  u64 boff = 0, voff = 0;
  if (valid) {
    cnt = counts[sid];
    boff = byte_off[sid];
    voff = value_off[sid];
    if (DELTA && prevs) prev = prevs[sid];
  }
  const bool is_short = valid && cnt > 0 && cnt <= kFlatMax;
  const bool is_long = valid && cnt > kFlatMax;

  if (flat_role) {
  if (valid && cnt == 0) sizes[sid] = 0;
  FlatEnt* ent = reinterpret_cast<FlatEnt*>(S.data);
  u32 n_short, my_first;
  const u32 n_slots = flat_setup(is_short, cnt, boff, voff, prev, sid, lane, ent, n_short, my_first);
  u32 carry_idx = 0xFFFFFFFFu, carry_doff = 0;  // wave-uniform: the stream cut by the last step's end
  for (u32 base = 0; base < n_slots; base += 64) {
    const FlatSlot f = flat_slot(base, n_slots, n_short, my_first, lane, ent);
    const u32* in = values + f.e.voff;
    u8* out = bytes + f.e.boff;
    const u32 keylen = (f.e.cnt + 3u) >> 2;
    const u32 lim = f.have ? f.e.cnt : 0u;
    const u32 nv = f.have ? min(4u, f.e.cnt - 4u * f.gi) : 0u;
    uint4 c = load_group<(NT & kNtLoads) != 0>(in, 4ull * f.gi, lim);
    if (DELTA) {
      u32 p = f.e.prev;  // the value before the group's first: the stream's seed or the stream's own value
      if (f.have && f.gi) p = in[4u * f.gi - 1u];
      c = make_uint4(c.x - p, c.y - c.x, c.z - c.y, c.w - c.z);
    }
    if (ZZ) c = make_uint4(zz_enc(c.x), zz_enc(c.y), zz_enc(c.z), zz_enc(c.w));
    if (DELTA || ZZ) zero_absent(c, 4ull * f.gi, lim);
    const u32 gd = classify_group<V, false>(c, nv);
    const u32 L = gi_total(gd);
    const u32 incl = seg_add_scan(f.idx, L) + (f.idx == carry_idx ? carry_doff : 0u);
    if (f.have) {
      u8* q = out + keylen + (incl - L);
      out[f.gi] = (u8)gi_key(gd);
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
      if (f.gi == keylen - 1u) sizes[f.e.sid] = (u64)keylen + incl;
    }
    carry_idx = (u32)__builtin_amdgcn_readlane((int)f.idx, 63);
    carry_doff = (u32)__builtin_amdgcn_readlane((int)incl, 63);
  }
  return;
  }

  // ---- long streams, one after the other, the whole wavefront on a wave tile at a time ----
  u64 todo = __ballot(is_long);
  while (todo) {
    const int src = (int)__builtin_ctzll(todo);
    todo &= todo - 1ull;
    const u32 count = (u32)__builtin_amdgcn_readlane((int)cnt, src);
    const u64 lboff = ((u64)(u32)__builtin_amdgcn_readlane((int)(boff >> 32), src) << 32) |
                      (u32)__builtin_amdgcn_readlane((int)boff, src);
    const u64 lvoff = ((u64)(u32)__builtin_amdgcn_readlane((int)(voff >> 32), src) << 32) |
                      (u32)__builtin_amdgcn_readlane((int)voff, src);
    u32 carry = (u32)__builtin_amdgcn_readlane((int)prev, src);
    const u32* in = values + lvoff;
    u8* out = bytes + lboff;
    const u64 keylen = ((u64)count + 3) >> 2;
    const u32 n_wt = (u32)((keylen + kWtKeys - 1) / kWtKeys);
    u64 doff = 0;
    for (u32 wt = 0; wt < n_wt; ++wt) {
      const u64 int0 = (u64)wt * kWtInts;
      uint4 x[4];
      u32 g[4];
      auto differences = [&](uint4 c) -> uint4 {  // values -> what gets coded, `carry` carried along
        if (DELTA) {
          const u32 p = wave_shift_up1(c.w, carry);
          carry = wave_last(c.w);
          c = make_uint4(c.x - p, c.y - c.x, c.z - c.y, c.w - c.z);
        }
        if (ZZ) c = make_uint4(zz_enc(c.x), zz_enc(c.y), zz_enc(c.z), zz_enc(c.w));
        return c;
      };
      if (int0 + kWtInts <= count) {
        // a whole wave tile (wave-uniform): unconditional loads, and the tile encoder of the
        // single-stream kernels for full tiles (keys first, lengths from the four keys of a lane at once)
        const u32* src = in + int0;
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = ld_stream16<(NT & kNtLoads) != 0>(src + 4u * (u32)(j * 64 + lane));
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = differences(x[j]);
        u32 k4 = classify_key<V>(x[0]) | (classify_key<V>(x[1]) << 8) | (classify_key<V>(x[2]) << 16) |
                 (classify_key<V>(x[3]) << 24);
        asm volatile("" : "+v"(k4));  // the four keys stay in one register
        const u32 lens4 = key_lens4<V>(k4, 16u);
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = ((k4 >> (8 * j)) & 0xFFu) | (((lens4 >> (8 * j)) & 0xFFu) << 24);
        doff += encode_wavetile<V, true, (NT & kNtStores) != 0>(x, g, (u32)kWtInts, lane, S, out + keylen + doff, out + (int0 >> 2));
      } else {
        const u32 nints = (u32)((u64)count - int0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          x[j] = make_uint4(0, 0, 0, 0);
          if (256u * j < nints)  // wave-uniform: quarters past the stream's end load and classify nothing
            x[j] = load_group<(NT & kNtLoads) != 0>(in, int0 + 4u * (j * 64 + lane), count);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          g[j] = 0u;  // no key bits, no bytes
          if (256u * j < nints) {
            uint4 c = differences(x[j]);
            const u64 first = int0 + 4u * (j * 64 + lane);
            if (DELTA || ZZ) zero_absent(c, first, count);
            x[j] = c;
            g[j] = classify_group<V, false>(c, first < count ? (u32)min((u64)4, (u64)count - first) : 0u);
          }
        }
        doff += encode_wavetile<V, false, (NT & kNtStores) != 0>(x, g, nints, lane, S, out + keylen + doff, out + (int0 >> 2));
      }
    }
    if (lane == 0) sizes[s0 + src] = keylen + doff;
  }
}

template <int V, bool DELTA, bool ZZ>
__global__ __launch_bounds__(64 * kBatchWaves) void k_encode_batch2(
    const u32* __restrict__ values, const u64* __restrict__ value_off, u8* __restrict__ bytes,
    const u64* __restrict__ byte_off, const u32* __restrict__ counts,
    const u32* __restrict__ prevs, u32 nstreams, u64* __restrict__ sizes, u32 K_long, u32 n_long_blocks,
    int nt_force, u64 nt_bytes) {
  __shared__ EncWaveLds s_lds[kBatchWaves];
  EncWaveLds& S = s_lds[wave_id()];
  // both or neither: loads alone measured slower than the default policy on a batch of posting lists
  if (batch_nt_policy<V>(nt_force, nt_bytes, value_off, counts, nstreams, true) == 3)
    encode_batch2_body<V, DELTA, ZZ, 3>(values, value_off, bytes, byte_off, counts, prevs, nstreams, sizes, K_long, n_long_blocks, S);
  else
    encode_batch2_body<V, DELTA, ZZ, 0>(values, value_off, bytes, byte_off, counts, prevs, nstreams, sizes, K_long, n_long_blocks, S);
}

template <int V, bool D, bool Z>
void launch_decode_batch(unsigned grid, unsigned chunk, const uint8_t* d_bytes, const uint64_t* d_byte_off,
                         uint32_t* d_values, const uint64_t* d_value_off, const uint32_t* d_counts,
                         const uint32_t* d_prev, uint32_t nstreams, uint64_t* d_sizes,
                         int nt_force, unsigned long long nt_bytes, hipStream_t st) {
  if (chunk) {  // chunked kernel: long streams `chunk` of the batch's streams per wavefront, then the flat part
    const unsigned long long long_blocks = (((unsigned long long)nstreams + chunk - 1) / chunk + kBatchWaves - 1) / kBatchWaves;
    const unsigned long long flat_blocks = (((unsigned long long)nstreams + 63) / 64 + kBatchWaves - 1) / kBatchWaves;
    hipLaunchKernelGGL((k_decode_batch2<V, D, Z>), dim3((unsigned)(long_blocks + flat_blocks)), dim3(64 * kBatchWaves), 0, st,
                       d_bytes, d_byte_off, d_values, d_value_off, d_counts, d_prev, nstreams, d_sizes, chunk,
                       (unsigned)long_blocks, nt_force, (u64)nt_bytes);
    return;
  }
  // short lists, four to a wavefront; then the long ones, one wavefront each
  const unsigned rows_grid = (unsigned)std::min<unsigned long long>(grid, ((unsigned long long)nstreams + 15) / 16);
  hipLaunchKernelGGL((k_decode_batch_rows<V, D, Z>), dim3(rows_grid ? rows_grid : 1), dim3(kBlock), 0, st,
                     d_bytes, d_byte_off, d_values, d_value_off, d_counts, d_prev, nstreams, d_sizes);
  hipLaunchKernelGGL((k_decode_batch<V, D, Z>), dim3(grid), dim3(kBlock), 0, st, d_bytes, d_byte_off,
                     d_values, d_value_off, d_counts, d_prev, nstreams, d_sizes, kRowMax + 1);
}
template <int V, bool D, bool Z>
void launch_encode_batch(unsigned grid, unsigned chunk, const uint32_t* d_values, const uint64_t* d_value_off,
                         uint8_t* d_bytes, const uint64_t* d_byte_off, const uint32_t* d_counts,
                         const uint32_t* d_prev, uint32_t nstreams, uint64_t* d_sizes,
                         int nt_force, unsigned long long nt_bytes, hipStream_t st) {
  if (chunk) {
    const unsigned long long long_blocks = (((unsigned long long)nstreams + chunk - 1) / chunk + kBatchWaves - 1) / kBatchWaves;
    const unsigned long long flat_blocks = (((unsigned long long)nstreams + 63) / 64 + kBatchWaves - 1) / kBatchWaves;
    hipLaunchKernelGGL((k_encode_batch2<V, D, Z>), dim3((unsigned)(long_blocks + flat_blocks)), dim3(64 * kBatchWaves), 0, st,
                       d_values, d_value_off, d_bytes, d_byte_off, d_counts, d_prev, nstreams, d_sizes, chunk,
                       (unsigned)long_blocks, nt_force, (u64)nt_bytes);
    return;
  }
  const unsigned rows_grid = (unsigned)std::min<unsigned long long>(grid, ((unsigned long long)nstreams + 15) / 16);
  hipLaunchKernelGGL((k_encode_batch_rows<V, D, Z>), dim3(rows_grid ? rows_grid : 1), dim3(kBlock), 0, st,
                     d_values, d_value_off, d_bytes, d_byte_off, d_counts, d_prev, nstreams, d_sizes);
  hipLaunchKernelGGL((k_encode_batch<V, D, Z>), dim3(grid), dim3(kBlock), 0, st, d_values, d_value_off,
                     d_bytes, d_byte_off, d_counts, d_prev, nstreams, d_sizes, kRowMax + 1);
}

}  // namespace svb

// Differential decode and plain running sums: decode fused with a uint32
// wrap-around inclusive scan seeded with the caller's prev.
//
// Replaces the second serial chain of the reference delta decoder
// (src/streamvbytedelta_x64_decode.c:32-65,69-175; scalar
// src/streamvbytedelta_decode.c:34-57) and the loop of zigzag_delta_decode
// (src/streamvbyte_zigzag.c:32-38).
//
// One workgroup = one tile of 16 wave tiles (16384 values), wave w owns wave
// tiles 4w..4w+3 and keeps its 64 decoded values per lane in registers while
// the tile's sum travels through the decoupled look-back; data offsets come
// from the same directory the plain decoder uses.
#pragma once
#include "svb_decode.h"
#include "svb_lookback.h"

namespace svb {

// Turn the 4 Q uint4 of a wave (Q wave tiles x 4 striped groups) into running
// sums relative to the wave's first value.  Returns the wave's total.
template <int Q = 4>
__device__ __forceinline__ u32 wave_running_sums(uint4 (&v)[Q][4], int lane) {
  u32 run = 0;
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint4& c = v[q][j];
      c.y += c.x;
      c.z += c.y;
      c.w += c.z;
      const u32 incl = wave_scan_incl(c.w, lane);
      const u32 ex = incl - c.w + run;
      run += wave_last(incl);
      c.x += ex; c.y += ex; c.z += ex; c.w += ex;
      __builtin_amdgcn_sched_barrier(0);  // one group at a time: interleaved scans cost ~40 VGPRs
    }
  return run;
}

// Workgroup-level part shared by the delta decoder and the plain scan kernel:
// exchange wave totals, publish the tile total, look back, return the value to
// add to every element of this wave.
template <int NW = kWaves>
__device__ __forceinline__ u32 tile_value_prefix(u32 wave_tot, u32 tile, u32 seed, int lane,
                                                 int wave, WsHeader* hdr, const LbWords& lb,
                                                 u32* s_wave_tot, u32* s_prefix) {
  if (lane == 0) s_wave_tot[wave] = wave_tot;
  __syncthreads();
  if (wave == 0) {
    u32 tile_tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) tile_tot += s_wave_tot[w];
    // four tile words per lane: the 255 tiles before this one are read directly, which covers nearly every
    // tile still in flight; with one word per lane most of them came through group words, a hop later
    // (measured: dense postings +1 ... 3 %, the plain running sum +2 %)
    const u32 excl = seed + (u32)lb2_exclusive_prefix<4>(lb, tile, tile_tot, lane, &hdr->error);
    if (lane == 0) *s_prefix = excl;
  }
  __syncthreads();
  u32 add = *s_prefix;
  for (int w = 0; w < wave; ++w) add += s_wave_tot[w];
  return add;
}

// STAMPS: diagnostic build with wall-clock stamps per tile (the product instantiation carries none of it).
template <int V, bool ZZ, int NW = kWaves, int NT = 0, bool STAMPS = false>
__global__ __launch_bounds__(NW * 64, NW >= 8 ? 4 : 3) void k_delta_decode(
    const u8* __restrict__ in, u32* __restrict__ out, u32 count, u32 prev,
    Dir dir, WsHeader* __restrict__ hdr, LbWords lb, u64* __restrict__ consumed, u64* __restrict__ stamps_arg) {
  u64* const stamps = STAMPS ? stamps_arg : nullptr;
  __shared__ WaveLds s_lds[NW];
  __shared__ u32 s_wave_tot[NW];
  __shared__ u32 s_prefix;
  __shared__ u32 s_tile;
  const int lane = lane_id(), wave = wave_id();
  const u64 t_start = stamps ? __builtin_amdgcn_s_memrealtime() : 0;  // diagnostic build only: 100 MHz wall clock
  if (threadIdx.x == 0) {
    const u32 t = atomicAdd(&hdr->ticket, 1u);
    // a workgroup per tile, a ticket each: the last one drawn leaves the counter as the next call expects it
    if (t == gridDim.x - 1u) __hip_atomic_store(&hdr->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_tile = t;
  }
  __syncthreads();
  const u32 tile = s_tile;
  // stamps of a tile: workgroup start, ticket in hand, keys and offsets in, decoded, prefix known, stored
  auto stamp = [&](int i) {
    if (stamps && threadIdx.x == 64 * (NW - 1)) stamps[8 * (u64)tile + i] = __builtin_amdgcn_s_memrealtime();
  };
  if (stamps && threadIdx.x == 64 * (NW - 1)) stamps[8 * (u64)tile + 0] = t_start;
  stamp(1);
  WaveLds& S = s_lds[wave];
  const u64 keylen = ((u64)count + 3) >> 2;
  const u32 n_wt = (u32)((keylen + kWtKeys - 1) / kWtKeys);
  const u8* data_base = in + keylen;
  const u32 wt0 = tile * (NW * 4) + wave * 4;  // NW waves of 4 wave tiles: NW * 4096 values per look-back tile

  uint4 v[4][4];
  // keys and offsets of the wave's 4 wave tiles up front
  u32 w[4];
  u64 doff[4];
  DirWalk walk;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const u32 wt = wt0 + q;
    w[q] = 0;
    doff[q] = 0;
    if (wt < n_wt) {
      const u64 koff = (u64)wt * kWtKeys + 4 * lane;
      if (koff < keylen) w[q] = ld_u32(in + koff);
      doff[q] = walk.offset(dir, wt, lane);
    }
  }
  // Full wave tiles whose 256 keys are all zero (dense posting-list gaps: every value below 256) need no
  // scan and no LDS -- group j*64+lane is the dword at byte 4*(j*64+lane) of the tile's data (see
  // decode_wavetile).  Their loads are all sent before any is used: taken one wave tile at a time, as
  // decode_wavetile would, a wave has 1 KiB in flight per round trip and the whole CU some 20 KiB, and
  // the kernel is bound by exactly that.
  if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(2); }
  bool direct[4];  // wave-uniform
  // The other wave tiles' data spans are requested now as well, all four before the first is used, where a
  // span fits kEarly registers per lane (up to 3 KiB: under three data bytes per value); the tile's byte count
  // comes from its keys.  One wave tile at a time, a wave had 1-2 KiB in flight per round trip.
  constexpr int kEarly = 3;
#ifndef SVB_DELTA_WINDOW16
#define SVB_DELTA_WINDOW16 0
#endif
  // wave tiles whose spans are in flight together: all four in workgroups of up to 8 waves.  The 16 waves of the
  // large workgroup move in step, and anything requested ahead only deepens the burst they send together
  // (measured at 2^28 values with 1.6 data bytes each: window 0 / 2 / 3 / 4 = 0.538 / 0.534 / 0.526 / 0.516 of
  // the HBM peak; 8 waves with all four ahead 0.567, without 0.512), so there a wave tile's span is requested
  // when its turn comes.
  constexpr int kWindow = NW >= 16 ? SVB_DELTA_WINDOW16 : 4;
  bool early[4];
  u32 nints[4], tot[4];
  SpanRegs<kEarly> span[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const u32 wt = wt0 + q;
    nints[q] = wt < n_wt ? (u32)min((u64)kWtInts, (u64)count - (u64)wt * kWtInts) : 0u;
    direct[q] = nints[q] == (u32)kWtInts && __ballot(w[q] != 0u) == 0ull;
    early[q] = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[q][j] = make_uint4(0, 0, 0, 0);
    if (V == V1234 && direct[q]) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[q][j].x = ld_u32(data_base + doff[q] + 4u * (j * 64 + lane));
    } else if (!direct[q] && nints[q]) {
      tot[q] = wavetile_data_bytes<V>(w[q], nints[q], lane);
      early[q] = kWindow > 0 && span_fits<kEarly>(data_base + doff[q], tot[q]);
      if (early[q] && q < kWindow) span_issue<kEarly, (NT & kNtLoads) != 0>(data_base + doff[q], tot[q], lane, span[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const u32 wt = wt0 + q;
    if (direct[q]) {
      if (V == V1234) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32 d = v[q][j].x;
          uint4 r = make_uint4(d & 0xFFu, (d >> 8) & 0xFFu, (d >> 16) & 0xFFu, d >> 24);
          if (ZZ) r = make_uint4(zz_dec(r.x), zz_dec(r.y), zz_dec(r.z), zz_dec(r.w));
          v[q][j] = r;
        }
      }
      if (wt == n_wt - 1 && lane == 0 && consumed) *consumed = keylen + doff[q] + (V == V1234 ? (u32)kWtInts : 0u);
    } else if (nints[q]) {
      const u32 total = early[q] ? decode_wavetile_landed<V, ZZ, kEarly>(w[q], nints[q], span[q], lane, S, v[q])
                                 : decode_wavetile<V, ZZ, false, false, (NT & kNtLoads) != 0>(w[q], nints[q], data_base + doff[q], lane, S, v[q]);
      if (wt == n_wt - 1 && lane == 0 && consumed) *consumed = keylen + doff[q] + total;
    }
    if (kWindow > 0 && q + kWindow < 4 && early[(q + kWindow) & 3])
      span_issue<kEarly, (NT & kNtLoads) != 0>(data_base + doff[(q + kWindow) & 3], tot[(q + kWindow) & 3], lane, span[(q + kWindow) & 3]);
  }
  const u32 wave_tot = wave_running_sums(v, lane);
  stamp(3);
  const u32 add = tile_value_prefix<NW>(wave_tot, tile, prev, lane, wave, hdr, lb, s_wave_tot,
                                        &s_prefix);
  stamp(4);
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint4 r = v[q][j];
      r.x += add; r.y += add; r.z += add; r.w += add;
      store_group<(NT & kNtStores) != 0>(out, ((u64)(wt0 + q)) * kWtInts + 4u * (j * 64 + lane), count, r);
    }
  stamp(5);
}

// ---------------------------------------------------------------------------
// Two-pass differential decode, no inter-workgroup waiting at all:
//   k_value_sums     decode every wave tile once, keep only the sum of its values
//   k_scan_u32_*     exclusive wrap-around scan of the 1024-value sums
//   k_delta_decode2  decode again, add the wave tile's base, running sums, store
// The stream is read twice (2*(0.25+dbar) + 4 bytes per value instead of
// 0.25+dbar+4), which a streaming kernel at two thirds of HBM peak still turns
// into less time than the look-back version's per-tile hand-off wait.
template <int V, bool ZZ>
__global__ __launch_bounds__(kBlock) void k_value_sums(const u8* __restrict__ in, u32 count, Dir dir,
                                                       u32* __restrict__ wt_vsum) {
  __shared__ WaveLds s_lds[kWaves];
  const int lane = lane_id(), wave = wave_id();
  WaveLds& S = s_lds[wave];
  const u64 keylen = ((u64)count + 3) >> 2;
  const u32 n_wt = (u32)((keylen + kWtKeys - 1) / kWtKeys);
  const u32 stride = gridDim.x * kWaves;
  const u8* data_base = in + keylen;
  DirWalk walk;
  for (u32 wt = blockIdx.x * kWaves + wave; wt < n_wt; wt += stride) {
    const u64 koff = (u64)wt * kWtKeys + 4 * lane;
    const u32 w = koff < keylen ? ld_u32(in + koff) : 0u;
    const u64 doff = walk.offset(dir, wt, lane);
    const u64 int0 = (u64)wt * kWtInts;
    const u32 nints = (u32)min((u64)kWtInts, (u64)count - int0);
    uint4 v[4];
    decode_wavetile<V, ZZ>(w, nints, data_base + doff, lane, S, v);
    u32 s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += v[j].x + v[j].y + v[j].z + v[j].w;
    s = wave_sum(s);
    if (lane == 0) wt_vsum[wt] = s;
  }
}

// In-place exclusive scan of `n` u32 in blocks of 1024 (4 per thread); block
// totals to blk_tot[].  Second kernel: exclusive scan of the block totals by one
// workgroup, seeded.  Wrap-around arithmetic throughout.
__global__ __launch_bounds__(kBlock) void k_scan_u32_blocks(u32* __restrict__ a, u32 n,
                                                            u32* __restrict__ blk_tot) {
  __shared__ u32 s_w[kWaves];
  const int lane = lane_id(), wave = wave_id();
  const u32 i0 = blockIdx.x * 1024u + threadIdx.x * 4u;
  u32 x[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) x[k] = i0 + k < n ? a[i0 + k] : 0u;
  const u32 t = x[0] + x[1] + x[2] + x[3];
  const u32 incl = wave_scan_incl(t, lane);
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  u32 base = incl - t;
  for (int w = 0; w < wave; ++w) base += s_w[w];
  u32 run = base;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (i0 + k < n) a[i0 + k] = run;
    run += x[k];
  }
  if (threadIdx.x == kBlock - 1) blk_tot[blockIdx.x] = run;
}
__global__ __launch_bounds__(1024) void k_scan_u32_top(u32* __restrict__ blk, u32 n, u32 seed) {
  __shared__ u32 s_w[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const u32 per = (n + 1023) / 1024;
  const u32 lo = min(n, t * per), hi = min(n, lo + per);
  u32 mine = 0;
  for (u32 i = lo; i < hi; ++i) mine += blk[i];
  const u32 incl = wave_scan_incl(mine, lane);
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  u32 run = seed + incl - mine;
  for (int w = 0; w < wave; ++w) run += s_w[w];
  for (u32 i = lo; i < hi; ++i) {
    const u32 x = blk[i];
    blk[i] = run;
    run += x;
  }
}

template <int V, bool ZZ>
__global__ __launch_bounds__(kBlock) void k_delta_decode2(const u8* __restrict__ in,
                                                          u32* __restrict__ out, u32 count, Dir dir,
                                                          const u32* __restrict__ wt_vpre,
                                                          const u32* __restrict__ blk_pre,
                                                          u64* __restrict__ consumed) {
  __shared__ WaveLds s_lds[kWaves];
  const int lane = lane_id(), wave = wave_id();
  WaveLds& S = s_lds[wave];
  const u64 keylen = ((u64)count + 3) >> 2;
  const u32 n_wt = (u32)((keylen + kWtKeys - 1) / kWtKeys);
  const u32 stride = gridDim.x * kWaves;
  const u8* data_base = in + keylen;
  DirWalk walk;
  for (u32 wt = blockIdx.x * kWaves + wave; wt < n_wt; wt += stride) {
    const u64 koff = (u64)wt * kWtKeys + 4 * lane;
    const u32 w = koff < keylen ? ld_u32(in + koff) : 0u;
    const u64 doff = walk.offset(dir, wt, lane);
    u32 run = wt_vpre[wt] + blk_pre[wt >> 10];  // sum of everything before this wave tile, incl. prev
    const u64 int0 = (u64)wt * kWtInts;
    const u32 nints = (u32)min((u64)kWtInts, (u64)count - int0);
    uint4 v[4];
    const u32 total = decode_wavetile<V, ZZ>(w, nints, data_base + doff, lane, S, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint4 c = v[j];
      c.y += c.x; c.z += c.y; c.w += c.z;
      const u32 incl = wave_scan_incl(c.w, lane);
      const u32 ex = incl - c.w + run;
      run += wave_last(incl);
      c.x += ex; c.y += ex; c.z += ex; c.w += ex;
      store_group(out, int0 + 4u * (j * 64 + lane), count, c);
    }
    if (wt == n_wt - 1 && lane == 0 && consumed) *consumed = keylen + doff + total;
  }
}

// Plain inclusive running sum out[i] = seed + sum_{j<=i} f(in[j]), f = identity
// or un-zigzag (zigzag_delta_decode).  8 B per value.
// NW waves per workgroup, each with 4 wave tiles: a tile of the look-back holds NW * 4096 values.
template <bool ZZ, int NW = kWaves, int NT = 0>
__global__ __launch_bounds__(NW * 64, 4) void k_running_sum(const u32* __restrict__ in,
                                                           u32* __restrict__ out, u32 count,
                                                           u32 seed, u32 n_tiles,
                                                           WsHeader* __restrict__ hdr, LbWords lb) {
  __shared__ u32 s_wave_tot[NW];
  __shared__ u32 s_prefix;
  __shared__ u32 s_tile;
  const int lane = lane_id(), wave = wave_id();
  // tiles are drawn from the ticket: by persistent workgroups until none is left, or one by each workgroup of a
  // launch that has a workgroup per tile
  for (;;) {
  if (threadIdx.x == 0) s_tile = atomicAdd(&hdr->ticket, 1u);
  __syncthreads();
  const u32 tile = __builtin_amdgcn_readfirstlane(s_tile);
  if (tile >= n_tiles) break;
  if (tile == 0u && threadIdx.x == 0) reset_other_ticket(hdr);
  const u64 wave_first = ((u64)tile * (NW * 4) + (u64)wave * 4) * kWtInts;
  uint4 v[4][4];
  const bool whole = wave_first + 4u * kWtInts <= count;  // wave-uniform: all 4096 values of this wave exist
  if (whole) {
    const uint4* src = reinterpret_cast<const uint4*>(in + wave_first);  // uniform base + lane offset
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint4 r = ld_stream16<(NT & kNtLoads) != 0>(src + ((u32)lane + (u32)(q * 256 + j * 64)));
        if (ZZ) r = make_uint4(zz_dec(r.x), zz_dec(r.y), zz_dec(r.z), zz_dec(r.w));
        v[q][j] = r;
      }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u64 first = wave_first + (u64)q * kWtInts + 4u * (j * 64 + lane);
        uint4 r = make_uint4(0, 0, 0, 0);
        if (first + 4 <= count) {
          __builtin_memcpy(&r, in + first, 16);
          if (ZZ) r = make_uint4(zz_dec(r.x), zz_dec(r.y), zz_dec(r.z), zz_dec(r.w));
        } else if (first < count) {
          r.x = in[first];
          if (first + 1 < count) r.y = in[first + 1];
          if (first + 2 < count) r.z = in[first + 2];
          if (ZZ) {
            r.x = zz_dec(r.x);
            r.y = first + 1 < count ? zz_dec(r.y) : 0u;
            r.z = first + 2 < count ? zz_dec(r.z) : 0u;
          }
        }
        v[q][j] = r;
      }
  }
  const u32 wave_tot = wave_running_sums(v, lane);
  const u32 add = tile_value_prefix<NW>(wave_tot, tile, seed, lane, wave, hdr, lb, s_wave_tot,
                                        &s_prefix);
  if (whole) {
    uint4* dst = reinterpret_cast<uint4*>(out + wave_first);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint4 r = v[q][j];
        r.x += add; r.y += add; r.z += add; r.w += add;
        st_stream16<(NT & kNtStores) != 0>(dst + ((u32)lane + (u32)(q * 256 + j * 64)), r);
      }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint4 r = v[q][j];
        r.x += add; r.y += add; r.z += add; r.w += add;
        store_group(out, wave_first + (u64)q * kWtInts + 4u * (j * 64 + lane), count, r);
      }
  }
  if (gridDim.x >= n_tiles) break;  // launched with a workgroup per tile: every ticket has its drawer
  }
}

}  // namespace svb

// Decoupled look-back: single-pass prefix sums across workgroups.
//
// Each tile owns one 64-bit status word per channel:
//     [63:62] 0 = nothing yet, 1 = AGGREGATE of this tile, 2 = inclusive PREFIX
//     [61:0]  the value
// A word is written by ONE 8-byte agent-scope store and read by 8-byte
// agent-scope loads, so flag and value travel together and no fence is needed
// (the value is the whole payload).  Tiles draw their index from an atomic
// ticket, so every predecessor of a running tile has itself started and will
// publish without waiting on anything later: the chain cannot deadlock whatever
// the dispatch order.  Every spin is bounded; on give-up the error word of the
// workspace header is set and the kernel still terminates.
#pragma once
#include "svb_device.h"

namespace svb {

constexpr u64 kLbAgg = 1ull << 62;
constexpr u64 kLbPfx = 2ull << 62;
constexpr u64 kLbVal = (1ull << 62) - 1;
constexpr u32 kLbSpinLimit = 1u << 20;  // polls before giving up

__device__ __forceinline__ void lb_publish(u64* status, u32 tile, u64 flag, u64 val) {
  __hip_atomic_store(status + tile, flag | (val & kLbVal), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// Sum of the aggregates of all tiles before `tile` (tile > 0), seeded by the
// nearest published inclusive prefix.  Must be called by all 64 lanes of one
// wavefront; returns the same value in every lane.
template <int kLbWin = 4>  // status words polled per lane and round (window = 64 * kLbWin tiles)
__device__ __forceinline__ u64 lb_exclusive_prefix(const u64* status, u32 tile, int lane,
                                                   u32* err, u32* rounds_out = nullptr) {
  u64 mine = 0;                          // this lane's share of the sum, reduced once at the end
  long long base = (long long)tile - 1;  // nearest predecessor not yet summed
  u32 spins = 0, rounds = 0;
  while (true) {
    ++rounds;
    u64 s[kLbWin];
#pragma unroll
    for (int i = 0; i < kLbWin; ++i) {
      const long long idx = base - lane - 64 * i;
      s[i] = kLbPfx;  // before tile 0: an empty prefix ends the chain
      if (idx >= 0)
        s[i] = __hip_atomic_load(status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    bool done = false, stall = false;
#pragma unroll
    for (int i = 0; i < kLbWin; ++i) {
      if (!done && !stall) {
        const u32 flag = (u32)(s[i] >> 62);
        const u64 pfx = __ballot(flag == 2u);
        const u64 inv = __ballot(flag == 0u);
        const int first = pfx ? (int)__builtin_ctzll(pfx) : 64;       // nearest prefix lane
        const u64 need = first >= 63 ? ~0ull : ((2ull << first) - 1);  // lanes 0..first
        if (inv & need) {
          stall = true;  // a word we need is not published yet
        } else {
          if (lane <= first) mine += s[i] & kLbVal;
          if (first < 64) done = true;
          else base -= 64;
        }
      }
    }
    if (done) break;
    if (stall) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > kLbSpinLimit) {
        if (lane == 0) atomicOr(err, 1u);
        break;
      }
    }
  }
  const u64 prefix = wave_sum64(mine);
  if (rounds_out) *rounds_out = rounds;
  return prefix;
}

// ---- two-level look-back -----------------------------------------------------
// Polling is what a look-back costs: every agent-scope load of a status word
// goes to memory, and a window of 256 words re-read by every waiting tile
// slows the whole device down measurably, while the waiting itself is cheap
// (so are not: atomic adds into shared words, which were tried for the group
// aggregates and made every poll of those words several times slower).
// Here a tile polls ONE 8-byte word per lane and round:
//   lanes  0..15  the words of the earlier tiles of its own GROUP of 16 tiles
//                 (one 128-byte line),
//   lanes 16..63  the words of the 48 groups before its own.
// The last tile of a group (its leader) publishes the group's aggregate as soon
// as the 15 other tile words are in, and the group's inclusive prefix once its
// own look-back over the group words has ended; nobody else writes group words.
// A tile's exclusive prefix = tile words before it in its group + group
// aggregates back to the nearest group prefix.  Only the lanes whose word was
// not yet published poll again, about once a microsecond.
struct LbWords {
  u64* tile;   // one word per tile: kLbAgg | value
  u64* group;  // one word per 16 tiles: kLbAgg | aggregate, later kLbPfx | inclusive prefix
};
constexpr int kLbGroupLanes = 48;

__device__ __forceinline__ u64 lb2_exclusive_prefix(const LbWords& L, u32 tile, u64 my_agg, int lane,
                                                    u32* err, u32* rounds_out = nullptr) {
  const u32 k = tile >> 4, r = tile & 15u;
  if (lane == 0) __hip_atomic_store(L.tile + tile, kLbAgg | my_agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool tile_lane = lane < 16;
  const bool leader = r == 15u;
  u64 w = 0, mine = 0;
  long long gbase = (long long)k - 1;  // nearest group not yet summed
  bool tile_done = false, group_done = false;
  u32 spins = 0, rounds = 0;
  while (true) {
    ++rounds;
    if (tile_lane) {
      if (!tile_done && (u32)lane < r && (w >> 62) == 0)
        w = __hip_atomic_load(L.tile + (tile - r) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (!group_done) {
      const long long idx = gbase - (lane - 16);
      if (idx < 0) w = kLbPfx;  // before group 0: an empty prefix ends the chain
      else if ((w >> 62) == 0)
        w = __hip_atomic_load(L.group + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const u32 flag = (u32)(w >> 62);
    if (!tile_done) {
      if (!__ballot(tile_lane && (u32)lane < r && flag == 0u)) {
        tile_done = true;
        const u64 v = (tile_lane && (u32)lane < r) ? (w & kLbVal) : 0ull;
        mine += v;
        if (leader) {  // the group is complete: let later groups count it
          const u64 gagg = wave_sum64(v) + my_agg;
          if (lane == 0)
            __hip_atomic_store(L.group + k, (k ? kLbAgg : kLbPfx) | gagg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (!group_done) {
      const u64 pfx = __ballot(!tile_lane && flag == 2u);
      const u64 inv = __ballot(!tile_lane && flag == 0u);
      const int first = pfx ? (int)__builtin_ctzll(pfx) : 64;  // nearest group prefix
      const u64 need = (first >= 63 ? ~0ull : ((2ull << first) - 1)) & ~0xFFFFull;  // lanes 16..first
      if (!(inv & need)) {
        if (!tile_lane && lane <= first) mine += w & kLbVal;
        if (first < 64) {
          group_done = true;
        } else {  // 48 aggregates and no prefix yet: go further back
          gbase -= kLbGroupLanes;
          if (!tile_lane) w = 0;
        }
      }
    }
    if (tile_done && group_done) break;
    __builtin_amdgcn_s_sleep(40);
    if (++spins > kLbSpinLimit) {
      if (lane == 0) atomicOr(err, 1u);
      break;
    }
  }
  const u64 prefix = wave_sum64(mine);
  if (leader && k && lane == 0)
    __hip_atomic_store(L.group + k, kLbPfx | (prefix + my_agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (rounds_out) *rounds_out = rounds;
  return prefix;
}

}  // namespace svb

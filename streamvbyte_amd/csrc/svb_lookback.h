// Decoupled look-back: single-pass prefix sums across workgroups, two levels.
//
// Tiles draw their index from an atomic ticket, so every predecessor of a
// running tile has itself started and will publish without waiting on anything
// later: the chain cannot deadlock whatever the dispatch order.  A status word is
//     [63:62] 0 = nothing yet, 1 = AGGREGATE, 2 = inclusive PREFIX
//     [61:0]  the value
// written by ONE 8-byte agent-scope store and read by 8-byte agent-scope loads,
// so flag and value travel together and no fence is needed (the value is the
// whole payload).  Every spin is bounded; on give-up the error word of the
// workspace header is set and the kernel still terminates.
//
// Polling is what a look-back costs: every agent-scope load of a status word
// goes to memory, and a window of 256 words re-read by every waiting tile
// (the classic one-level scheme, tried first) slows the whole device down
// measurably, while the waiting itself is cheap.  Atomic adds into shared
// words are not cheap either: tried for the group aggregates, they made every
// poll of those words several times slower.
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
#pragma once
#include "svb_device.h"

namespace svb {

constexpr u64 kLbAgg = 1ull << 62;
constexpr u64 kLbPfx = 2ull << 62;
constexpr u64 kLbVal = (1ull << 62) - 1;
constexpr u32 kLbSpinLimit = 1u << 20;  // polls before giving up

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
  u32 spins = 0, rounds = 0, tile_round = 0, group_round = 0;  // the *_round values feed diagnostics only
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
        tile_round = rounds;
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
          group_round = rounds;
        } else {  // 48 aggregates and no prefix yet: go further back
          gbase -= kLbGroupLanes;
          if (!tile_lane) w = 0;
        }
      }
    }
    if (tile_done && group_done) break;
    // every later tile waits for what a leader publishes: leaders poll eagerly
    if (leader) __builtin_amdgcn_s_sleep(4);
    else __builtin_amdgcn_s_sleep(40);
    if (++spins > kLbSpinLimit) {
      if (lane == 0) atomicOr(err, 1u);
      break;
    }
  }
  const u64 prefix = wave_sum64(mine);
  if (leader && k && lane == 0)
    __hip_atomic_store(L.group + k, kLbPfx | (prefix + my_agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (rounds_out) *rounds_out = rounds | (tile_round << 8) | (group_round << 16);
  return prefix;
}

}  // namespace svb

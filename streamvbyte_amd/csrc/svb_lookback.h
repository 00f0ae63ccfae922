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

}  // namespace svb

// Decoupled look-back: single-pass prefix sums across workgroups, two levels.
//
// Tiles draw their index from an atomic ticket, so every predecessor of a
// running tile has itself started and will publish without waiting on anything
// later: the chain cannot deadlock whatever the dispatch order.  A status word is
//     [63:62] 0 = nothing yet, 1 = AGGREGATE, 2 = inclusive PREFIX
//     [61:35] the call's tag (see LbWords::tag2): a word whose tag is not the running call's reads as
//             "nothing yet", so a workspace whose words were zero ONCE never needs clearing again
//     [34:0]  the value (a stream of 64-bit values holds less than 2^35 data bytes)
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
// Here a tile polls at most kT + 1 eight-byte words per lane and round:
//   kT tile words: the earlier tiles of its own GROUP of 16 tiles and all tiles
//                  of the 4 kT - 1 groups before it (a window of < 64 kT tiles),
//   1 group word:  the 64 groups before that window.
// The last tile of a group (its leader) publishes the group's aggregate as soon
// as the 15 other tile words are in, and the group's inclusive prefix once its
// own look-back has ended; nobody else writes group words.  Group words are read
// only for groups at least 4 kT groups back, whose leaders have long published.
// A tile's exclusive prefix = tile words of its window + group aggregates back
// to the nearest group prefix.  Only the lanes whose word was not yet published
// poll again, about once a microsecond.
#pragma once
#include "svb_device.h"

namespace svb {

// The header holds two ticket counters 8 bytes apart; a call works on the one its header pointer leads
// to (WsHeader::ticket1), and the workgroup that handles its tile 0 zeroes the OTHER one: that is the
// counter of the next call on this workspace, and nobody uses it now (the call before, which did, has
// ended).  So a workspace that was zero once needs no clearing launch for its tickets.  (Kept out of
// the ticket draw itself: the pipelined encoder has no scalar register to spare there.)
__device__ __forceinline__ void reset_other_ticket(WsHeader* hdr) {
  __hip_atomic_store(reinterpret_cast<u32*>(reinterpret_cast<uintptr_t>(&hdr->ticket) ^ 8u), 0u, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
// The error word of the workspace whose header (or header + 8, see draw_ticket) `p` points into.
__device__ __forceinline__ u32* ws_error_word(u32* p) {
  return reinterpret_cast<u32*>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)255);
}
constexpr u64 kLbAgg = 1ull << 62;
constexpr u64 kLbPfx = 2ull << 62;
constexpr u64 kLbVal = (1ull << 35) - 1;
#ifdef SVB_LB_STRESS
constexpr u32 kLbTags = 1u << 4;   // stress build: the tags run out every 16 calls, so that tests see the wrap
#else
constexpr u32 kLbTags = 1u << 27;  // tags 1 .. kLbTags - 1
#endif
constexpr u32 kLbSpinLimit = 1u << 20;  // polls before giving up

struct LbWords {
  u64* tile;   // one word per tile: kLbAgg | tag | value
  // one word per 16 tiles: kLbAgg | tag | aggregate, later kLbPfx | tag | inclusive prefix.  Kept as a
  // distance from `tile` in words: one scalar register instead of two in kernels that have none to spare.
  u32 group_off;
  __device__ __forceinline__ u64* group_words() const { return tile + group_off; }
  // The running call's tag, shifted to its place in the high dword of a word (bits 29..3 there).  0 for
  // calls that clear the words first; 1 .. 2^27 - 1, counted per workspace by the host, for calls on
  // an initialised workspace (svbgpu_workspace_init), which skip the clearing launch.
  u32 tag2 = 0;
  // flag of a word as this call sees it: 0 unless the word carries this call's tag
  __device__ __forceinline__ u32 flag(u64 w) const {
    const u32 hi = (u32)(w >> 32);
    return ((hi ^ tag2) & 0x3FFFFFF8u) ? 0u : hi >> 30;
  }
  // value: a byte count below 2^35, or a sum that its users take modulo 2^32 (running sums of values may
  // exceed the field: only the low bits are kept)
  __device__ __forceinline__ u64 word(u64 flag_bits, u64 value) const {
    return flag_bits | ((u64)tag2 << 32) | (value & ((1ull << 35) - 1));
  }
};
// kT = tile words polled per lane: the window of tiles read one by one covers
// the tile's own group and the 4 kT - 1 groups before it.
// The two halves of a look-back, for kernels that do other work between
// publishing a tile's aggregate and needing its prefix (by then the
// predecessors' words have usually arrived and one poll round answers):
//   lb2_publish   the tile's aggregate becomes visible to later tiles
//   lb2_resolve   poll until the exclusive prefix of the tile is known
// lb2_exclusive_prefix is one after the other.
// Stress build of the look-back (-DSVB_LB_STRESS, libstreamvbyte_stress.so.2; results stay exact).
// With svbgpu_set_option("debug_ablate", 64) the first word of the workspace header's second 128-byte
// line -- err[32], zero otherwise -- makes a pseudo-random quarter of the tiles sleep for tens of
// microseconds right before they publish (salt 0), before a leader publishes its group's aggregate (1)
// and before it publishes the group's prefix (2), so that words arrive late and out of order however
// evenly the hardware happens to run.  The protocol must give the same bytes under any such timing; the
// parity tests run once against that build.  Not in the product build: even as a never-taken branch
// on a cached scalar load the hook cost the encoder 2.5-3 % (same box, alternating builds), and placed
// in the ticket's cache line, which the whole device hammers with atomics, it tripled its time.
constexpr int kLbStressWord = 32;
__device__ __forceinline__ void lb_stress_delay(const u32* err, u32 tile, u32 salt) {
#ifdef SVB_LB_STRESS
  if (!err || !ws_error_word(const_cast<u32*>(err))[kLbStressWord]) return;  // wave-uniform
  const u32 h = (tile * 2654435761u + salt * 40503u) >> 13;
  if (h & 3u) return;
  const int naps = 2 + (int)((h >> 2) & 7u);  // about 8 .. 35 us
  for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(127);
#endif
}
__device__ __forceinline__ void lb2_publish(const LbWords& L, u32 tile, u64 my_agg, int lane, const u32* err = nullptr) {
  lb_stress_delay(err, tile, 0);
  if (lane == 0) __hip_atomic_store(L.tile + tile, L.word(kLbAgg, my_agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// What the first poll round of a look-back read (lb2_begin): words not yet
// published are 0 and are polled again by lb2_resolve.
template <int kT>
struct LbState {
  u64 tw[kT];
  u64 gw;
};
// Sends the first round of polls of `tile`'s look-back and returns without
// looking at the answers, so that they travel while the caller does other work.
template <int kT = 2>
__device__ __forceinline__ LbState<kT> lb2_begin(const LbWords& L, u32 tile, int lane) {
  constexpr u32 kDirect = 4 * kT - 1;
  const u32 k = tile >> 4;
  const u32 g0 = k > kDirect ? k - kDirect : 0u;
  const int t0 = 16 * (int)g0;
  LbState<kT> st;
#pragma unroll
  for (int i = 0; i < kT; ++i) {
    const int idx = (int)tile - 1 - lane - 64 * i;
    st.tw[i] = idx >= t0 ? __hip_atomic_load(L.tile + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  }
  const int gidx = (int)g0 - 1 - lane;
  st.gw = gidx >= 0 ? __hip_atomic_load(L.group_words() + gidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  return st;
}
template <int kT = 2>
__device__ __forceinline__ u64 lb2_resolve(const LbWords& L, u32 tile, u64 my_agg, int lane, u32* err,
                                           u32* rounds_out = nullptr, const LbState<kT>* first = nullptr) {
  constexpr u32 kDirect = 4 * kT - 1;
  const u32 k = tile >> 4;
  const u32 g0 = k > kDirect ? k - kDirect : 0u;  // first group of the tile window
  const int t0 = 16 * (int)g0;  // tile and group indices fit 32 bits (at most 2^20 tiles)
  const bool leader = (tile & 15u) == 15u;
  u64 tw[kT], gw = 0, mine = 0;
#pragma unroll
  for (int i = 0; i < kT; ++i) tw[i] = first ? first->tw[i] : 0ull;
  if (first) gw = first->gw;
  int gbase = (int)g0 - 1;  // nearest group not yet summed
  bool tile_done = false, group_done = false, agg_out = false;
  u32 spins = 0, rounds = 0, tile_round = 0, group_round = 0;  // the *_round values feed diagnostics only
  while (true) {
    ++rounds;
    if (!tile_done) {
#pragma unroll
      for (int i = 0; i < kT; ++i) {
        const int idx = (int)tile - 1 - lane - 64 * i;
        if (idx >= t0 && L.flag(tw[i]) == 0)
          tw[i] = __hip_atomic_load(L.tile + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (!group_done) {
      const int idx = gbase - lane;
      if (idx < 0) gw = L.word(kLbPfx, 0);  // before group 0: an empty prefix ends the chain
      else if (L.flag(gw) == 0)
        gw = __hip_atomic_load(L.group_words() + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (leader && !agg_out && !__ballot(lane < 15 && L.flag(tw[0]) == 0)) {
      // the group is complete: let later groups count it
      agg_out = true;
      const u64 gagg = wave_sum64(lane < 15 ? (tw[0] & kLbVal) : 0ull) + my_agg;
      lb_stress_delay(err, tile, 1);
      if (lane == 0)
        __hip_atomic_store(L.group_words() + k, L.word(k ? kLbAgg : kLbPfx, gagg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!tile_done) {
      bool missing = false;
#pragma unroll
      for (int i = 0; i < kT; ++i)
        missing |= ((int)tile - 1 - lane - 64 * i >= t0) && L.flag(tw[i]) == 0;
      if (!__ballot(missing)) {
        tile_done = true;
        tile_round = rounds;
#pragma unroll
        for (int i = 0; i < kT; ++i)
          if ((int)tile - 1 - lane - 64 * i >= t0) mine += tw[i] & kLbVal;
      }
    }
    if (!group_done) {
      const u32 flag = L.flag(gw);
      const u64 pfx = __ballot(flag == 2u);
      const u64 inv = __ballot(flag == 0u);
      const int first = pfx ? (int)__builtin_ctzll(pfx) : 64;       // nearest group prefix
      const u64 need = first >= 63 ? ~0ull : ((2ull << first) - 1);  // lanes 0..first
      if (!(inv & need)) {
        if (lane <= first) mine += gw & kLbVal;
        if (first < 64) {
          group_done = true;
          group_round = rounds;
        } else {  // 64 aggregates and no prefix yet: go further back
          gbase -= 64;
          gw = 0;
        }
      }
    }
    if (tile_done && group_done) break;
    // every later tile waits for what a leader publishes: leaders poll eagerly
    if (leader) __builtin_amdgcn_s_sleep(4);
    else __builtin_amdgcn_s_sleep(40);
    if (++spins > kLbSpinLimit) {
      if (lane == 0) atomicOr(ws_error_word(err), 1u);
      break;
    }
  }
  const u64 prefix = wave_sum64(mine);
  if (leader && k) lb_stress_delay(err, tile, 2);
  if (leader && k && lane == 0)
    __hip_atomic_store(L.group_words() + k, L.word(kLbPfx, prefix + my_agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (rounds_out) *rounds_out = rounds | (tile_round << 8) | (group_round << 16);
  return prefix;
}
template <int kT = 2>
__device__ __forceinline__ u64 lb2_exclusive_prefix(const LbWords& L, u32 tile, u64 my_agg, int lane,
                                                    u32* err, u32* rounds_out = nullptr) {
  lb2_publish(L, tile, my_agg, lane, err);
  return lb2_resolve<kT>(L, tile, my_agg, lane, err, rounds_out);
}

}  // namespace svb

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

constexpr u64 kLbAgg = 1ull << 62;
constexpr u64 kLbPfx = 2ull << 62;
constexpr u64 kLbVal = (1ull << 62) - 1;
constexpr u32 kLbSpinLimit = 1u << 20;  // polls before giving up

struct LbWords {
  u64* tile;   // one word per tile: kLbAgg | value
  u64* group;  // one word per 16 tiles: kLbAgg | aggregate, later kLbPfx | inclusive prefix
  __device__ __forceinline__ u64* tw(int t) const { return tile + t; }
};
// kT = tile words polled per lane: the window of tiles read one by one covers
// the tile's own group and the 4 kT - 1 groups before it.
// The two halves of a look-back, for kernels that do other work between
// publishing a tile's aggregate and needing its prefix (by then the
// predecessors' words have usually arrived and one poll round answers):
//   lb2_publish   the tile's aggregate becomes visible to later tiles
//   lb2_resolve   poll until the exclusive prefix of the tile is known
// lb2_exclusive_prefix is one after the other.
__device__ __forceinline__ void lb2_publish(const LbWords& L, u32 tile, u64 my_agg, int lane) {
  if (lane == 0) __hip_atomic_store(L.tw((int)tile), kLbAgg | my_agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    st.tw[i] = idx >= t0 ? __hip_atomic_load(L.tw(idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  }
  const int gidx = (int)g0 - 1 - lane;
  st.gw = gidx >= 0 ? __hip_atomic_load(L.group + gidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
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
        if (idx >= t0 && (tw[i] >> 62) == 0)
          tw[i] = __hip_atomic_load(L.tw(idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (!group_done) {
      const int idx = gbase - lane;
      if (idx < 0) gw = kLbPfx;  // before group 0: an empty prefix ends the chain
      else if ((gw >> 62) == 0)
        gw = __hip_atomic_load(L.group + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (leader && !agg_out && !__ballot(lane < 15 && (tw[0] >> 62) == 0)) {
      // the group is complete: let later groups count it
      agg_out = true;
      const u64 gagg = wave_sum64(lane < 15 ? (tw[0] & kLbVal) : 0ull) + my_agg;
      if (lane == 0)
        __hip_atomic_store(L.group + k, (k ? kLbAgg : kLbPfx) | gagg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!tile_done) {
      bool missing = false;
#pragma unroll
      for (int i = 0; i < kT; ++i)
        missing |= ((int)tile - 1 - lane - 64 * i >= t0) && (tw[i] >> 62) == 0;
      if (!__ballot(missing)) {
        tile_done = true;
        tile_round = rounds;
#pragma unroll
        for (int i = 0; i < kT; ++i)
          if ((int)tile - 1 - lane - 64 * i >= t0) mine += tw[i] & kLbVal;
      }
    }
    if (!group_done) {
      const u32 flag = (u32)(gw >> 62);
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
template <int kT = 2>
__device__ __forceinline__ u64 lb2_exclusive_prefix(const LbWords& L, u32 tile, u64 my_agg, int lane,
                                                    u32* err, u32* rounds_out = nullptr) {
  lb2_publish(L, tile, my_agg, lane);
  return lb2_resolve<kT>(L, tile, my_agg, lane, err, rounds_out);
}

// ---- the look-back of the short-tile kernels (thousands of tiles in flight) ---------------------
// A tile word is 4 bytes (flag | value), so that ONE 16-byte load per lane reads the words of 256
// consecutive tiles: with over 100 tiles starting per microsecond and 2-4 us from a word's store to
// its arrival in a reader's registers, a tile must read several hundred predecessors directly --
// nobody can have summed them up for it yet.  Older tiles are read through the group words above.
constexpr u32 kLb32Flag = 0x80000000u;

struct Lb4Words {
  u32* tile;   // one word per tile: kLb32Flag | byte count of the tile
  u64* group;  // as LbWords::group
};
// What the first round of reads returned: kC 16-byte chunks of tile words per lane (the tile's own
// group and the 16 kC - 1 groups before it) and one group word.
template <int kC>
struct Lb4Polls {
  uint4 tw[kC];
  u64 gw;
};
__device__ __forceinline__ uint4 lb4_load16(const Lb4Words& L, int chunk) {
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(L.tile, 0, 0x7FFFFFFF, 0x00020000);
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, 16 * chunk, 0, 16);  // aux 16 = sc1: agent-coherent
  return make_uint4(v[0], v[1], v[2], v[3]);
}
// Publishes the tile's byte count and sends the first round of reads without looking at the answers.
template <int kC>
__device__ __forceinline__ Lb4Polls<kC> lb4_begin(const Lb4Words& L, u32 tile, u32 my_agg, int lane) {
  if (lane == 0) __hip_atomic_store(L.tile + tile, kLb32Flag | my_agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  constexpr u32 kDirect = 16 * kC - 1;
  const u32 k = tile >> 4;
  const u32 g0 = k > kDirect ? k - kDirect : 0u;
  Lb4Polls<kC> P;
#pragma unroll
  for (int i = 0; i < kC; ++i) {
    const int c = (int)(tile >> 2) - lane - 64 * i;
    P.tw[i] = lb4_load16(L, max(c, 0));
  }
  const int gidx = (int)g0 - 1 - lane;
  P.gw = gidx >= 0 ? __hip_atomic_load(L.group + gidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  return P;
}
// Exclusive prefix of `tile`.  After a round that left words missing ONE lane watches ONE word, the
// nearest missing predecessor, and the others are read again only when that one has arrived: with
// thousands of tiles in flight, every waiting tile re-reading all its missing words floods exactly
// the lines the awaited stores go to.
template <int kC>
__device__ __forceinline__ u64 lb4_resolve(const Lb4Words& L, u32 tile, u32 my_agg, int lane, Lb4Polls<kC>& P, u32* err,
                                           u32* rounds_out = nullptr) {
  constexpr u32 kDirect = 16 * kC - 1;
  const u32 k = tile >> 4;
  const u32 g0 = k > kDirect ? k - kDirect : 0u;
  const int c0 = 4 * (int)g0;  // first chunk of the window
  const bool leader = (tile & 15u) == 15u;
  u64 gw = P.gw, gsum = 0;
  u32 tsum = 0;
  int gbase = (int)g0 - 1;
  bool tile_done = false, group_done = false, agg_out = false, first = true;
  u32 spins = 0, rounds = 0, watches = 0;
  // bit j of the result: word j of lane's chunk i is needed and not yet published
  auto missing_of = [&](int i) -> u32 {
    const int c = (int)(tile >> 2) - lane - 64 * i;
    if (c < c0) return 0u;
    const u32 w[4] = {P.tw[i].x, P.tw[i].y, P.tw[i].z, P.tw[i].w};
    u32 m = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((u32)(4 * c + j) < tile && !(w[j] & kLb32Flag)) m |= 1u << j;
    return m;
  };
  auto sum_of = [&](int i) -> u32 {
    const int c = (int)(tile >> 2) - lane - 64 * i;
    if (c < c0) return 0u;
    const u32 w[4] = {P.tw[i].x, P.tw[i].y, P.tw[i].z, P.tw[i].w};
    u32 s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((u32)(4 * c + j) < tile) s += w[j] & ~kLb32Flag;
    return s;
  };
  while (true) {
    ++rounds;
    if (!first) {  // read again what was missing
      if (!tile_done) {
#pragma unroll
        for (int i = 0; i < kC; ++i)
          if (missing_of(i)) P.tw[i] = lb4_load16(L, (int)(tile >> 2) - lane - 64 * i);
      }
      if (!group_done) {
        const int idx = gbase - lane;
        if (idx >= 0 && (gw >> 62) == 0) gw = __hip_atomic_load(L.group + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    first = false;
    if (!group_done && gbase - lane < 0) gw = kLbPfx;  // before group 0: an empty prefix ends the chain
    // leader: the 15 other tiles of its group are words of lanes 0..3, chunk 0
    if (leader && !agg_out && !__ballot(lane < 4 && missing_of(0))) {
      agg_out = true;
      const u64 gagg = (u64)wave_sum(lane < 4 ? sum_of(0) : 0u) + my_agg;
      if (lane == 0)
        __hip_atomic_store(L.group + k, (k ? kLbAgg : kLbPfx) | gagg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const void* watch = nullptr;  // wave-uniform: the nearest word still missing
    bool watch32 = false;
    if (!tile_done) {
#pragma unroll
      for (int i = kC - 1; i >= 0; --i) {
        const u32 m = missing_of(i);
        const u64 b = __ballot(m != 0u);
        if (b) {  // nearest: lowest lane, highest word of its chunk
          const int l = (int)__builtin_ctzll(b);
          const u32 ml = (u32)__builtin_amdgcn_readlane((int)m, l);
          watch = L.tile + (4 * ((int)(tile >> 2) - l - 64 * i) + (31 - __builtin_clz(ml)));
          watch32 = true;
        }
      }
      if (!watch) {
        tile_done = true;
#pragma unroll
        for (int i = 0; i < kC; ++i) tsum += sum_of(i);
      }
    }
    if (!group_done) {
      const u32 flag = (u32)(gw >> 62);
      const u64 pfx = __ballot(flag == 2u);
      const u64 inv = __ballot(flag == 0u);
      const int firstp = pfx ? (int)__builtin_ctzll(pfx) : 64;
      const u64 need = firstp >= 63 ? ~0ull : ((2ull << firstp) - 1);
      if (!(inv & need)) {
        if (lane <= firstp) gsum += gw & kLbVal;
        if (firstp < 64) group_done = true;
        else {  // 64 aggregates and no prefix yet: go further back
          gbase -= 64;
          gw = 0;
          if (!watch) continue;
        }
      } else if (!watch) {
        watch = L.group + (gbase - (int)__builtin_ctzll(inv & need));
      }
    }
    if (tile_done && group_done) break;
    // The words missing from the first round are mostly those of tiles that started within a
    // microsecond of this one: they are read again at once; the quiet wait starts after that.
    if (rounds < 2) continue;
    ++watches;
    for (;;) {
      if (leader) __builtin_amdgcn_s_sleep(8);
      else __builtin_amdgcn_s_sleep(24);
      u32 got = 0;
      if (lane == 0) {
        if (watch32) got = __hip_atomic_load(static_cast<const u32*>(watch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 31;
        else got = (u32)(__hip_atomic_load(static_cast<const u64*>(watch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 62);
      }
      if (__builtin_amdgcn_readfirstlane(got)) break;
      if (++spins > kLbSpinLimit) break;
    }
    if (spins > kLbSpinLimit) {
      if (lane == 0) atomicOr(err, 1u);
      break;
    }
  }
  const u64 prefix = wave_sum64(gsum) + wave_sum(tsum);
  if (leader && k && lane == 0)
    __hip_atomic_store(L.group + k, kLbPfx | (prefix + my_agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (rounds_out) *rounds_out = rounds | (watches << 8) | (min(spins, 0xFFFFu) << 16);
  return prefix;
}


}  // namespace svb

// Differential decode fused with an intersection: the values of a delta-coded
// stream that also occur in a sorted array, written in stream order, without the
// decoded array ever reaching memory.
//
// The reference stops at the decoded array (streamvbyte_delta_decode,
// src/streamvbytedelta_decode.c:63-74); intersecting posting lists is what its
// users do with it next (README.md:30-39 there).
//
// Structure: the two-pass differential decoder's pre-pass gives every wave tile
// the sum of all values before it (k_value_sums + scans, svb_delta.h).  One
// workgroup then takes a tile of 16 wave tiles in ticket order; each wave decodes
// its 4096 values into registers, finds the part of the sorted array that lies
// between its smallest and largest value with two 64-way searches, walks that
// part through LDS 1024 entries at a time and looks every value up by binary
// search in LDS.  The number of hits travels through the decoupled look-back,
// which gives every tile its place in the output.
#pragma once
#include "svb_delta.h"

namespace svb {

// Maximum over the wavefront, wave-uniform (the DPP pattern of wave_scan_incl;
// the zero shifted in is the identity of max on unsigned values).
__device__ __forceinline__ u32 wave_max(u32 x) {
  x = max(x, dpp_zero<0x111, 0xF>(x));
  x = max(x, dpp_zero<0x112, 0xF>(x));
  x = max(x, dpp_zero<0x114, 0xF>(x));
  x = max(x, dpp_zero<0x118, 0xF>(x));
  x = max(x, dpp_zero<0x142, 0xA>(x));
  x = max(x, dpp_zero<0x143, 0xC>(x));
  return wave_last(x);
}

// First index i of the ascending array b[0..n) with b[i] >= key (GT false) or
// b[i] > key (GT true), n if there is none.  key and the result are
// wave-uniform; every step probes 64 places, so 2^32 entries take six steps.
template <bool GT>
__device__ __forceinline__ u32 wave_bound(const u32* __restrict__ b, u32 n, u32 key, int lane) {
  u32 lo = 0, hi = n;
  while (lo < hi) {
    const u32 step = (hi - lo + 63u) >> 6;
    const u64 idx = (u64)lo + (u64)lane * step;
    bool ok = true;  // places past hi count as "large enough"
    if (idx < hi) {
      const u32 x = b[idx];
      ok = GT ? x > key : x >= key;
    }
    const u64 m = __ballot(ok);
    const u32 first = m ? (u32)__builtin_ctzll(m) : 64u;  // the probes before it fail, it and all after it pass
    if (first == 0) {
      hi = lo;
    } else {
      const u64 nh = (u64)lo + (u64)first * step;
      lo = lo + (first - 1) * step + 1;
      hi = (u32)min((u64)hi, nh);
    }
  }
  return lo;
}

// bound[w] = first index i with b[i] >= the value that precedes wave w's first
// (wave w takes wave tiles 4w..4w+3; wt_vpre + blk_pre as in k_delta_intersect).
// Of an ascending stream every value of wave w lies between the values that
// precede waves w and w+1, so b[bound[w] .. bound[w+1]] is all it can meet.
// One thread per wave; neighbouring threads search for neighbouring keys.
__global__ __launch_bounds__(kBlock) void k_wave_bounds(const u32* __restrict__ wt_vpre,
                                                        const u32* __restrict__ blk_pre, u32 n_waves,
                                                        const u32* __restrict__ b, u32 n_b,
                                                        u32* __restrict__ bound) {
  const u32 w = blockIdx.x * kBlock + threadIdx.x;
  if (w >= n_waves) return;
  const u32 key = wt_vpre[4 * w] + blk_pre[(4 * w) >> 10];
  u32 lo = 0, hi = n_b;
  while (lo < hi) {
    const u32 mid = lo + ((hi - lo) >> 1);
    if (b[mid] < key) lo = mid + 1; else hi = mid;
  }
  bound[w] = lo;
}

// Which of the four values of c (those flagged in `valid`) occur in the sorted
// LDS array buf[0..len), len <= 1024, entries from len to 1023 hold 0xFFFFFFFF?
// first / last = buf[0] / buf[len-1]; top = half the smallest power of two >= len
// (wave-uniform: the searches only look at the first 2*top entries).  Returns a
// 4-bit mask.
__device__ __forceinline__ u32 lds_find4(const u32* buf, u32 len, u32 top, u32 first, u32 last, const uint4& c,
                                         u32 valid) {
  const bool r0 = (valid & 1u) && c.x >= first && c.x <= last;
  const bool r1 = (valid & 2u) && c.y >= first && c.y <= last;
  const bool r2 = (valid & 4u) && c.z >= first && c.z <= last;
  const bool r3 = (valid & 8u) && c.w >= first && c.w <= last;
  if (__ballot(r0 || r1 || r2 || r3) == 0ull) return 0u;
  u32 p0 = 0, p1 = 0, p2 = 0, p3 = 0;
  for (u32 step = top; step; step >>= 1) {  // four independent lower-bound searches
    const u32 a0 = buf[p0 + step - 1], a1 = buf[p1 + step - 1], a2 = buf[p2 + step - 1], a3 = buf[p3 + step - 1];
    p0 += a0 < c.x ? step : 0u;
    p1 += a1 < c.y ? step : 0u;
    p2 += a2 < c.z ? step : 0u;
    p3 += a3 < c.w ? step : 0u;
  }
  u32 m = 0;
  m |= (r0 && p0 < len && buf[p0] == c.x) ? 1u : 0u;
  m |= (r1 && p1 < len && buf[p1] == c.y) ? 2u : 0u;
  m |= (r2 && p2 < len && buf[p2] == c.z) ? 4u : 0u;
  m |= (r3 && p3 < len && buf[p3] == c.w) ? 8u : 0u;
  return m;
}

// The same question answered from a bitmap: bit x - first of bits[] is set for
// every entry x of the sorted array between first and last.
__device__ __forceinline__ u32 lds_bits4(const u32* bits, u32 first, u32 last, const uint4& c, u32 valid) {
  const bool r0 = (valid & 1u) && c.x >= first && c.x <= last;
  const bool r1 = (valid & 2u) && c.y >= first && c.y <= last;
  const bool r2 = (valid & 4u) && c.z >= first && c.z <= last;
  const bool r3 = (valid & 8u) && c.w >= first && c.w <= last;
  if (__ballot(r0 || r1 || r2 || r3) == 0ull) return 0u;
  const u32 d0 = r0 ? c.x - first : 0u, d1 = r1 ? c.y - first : 0u, d2 = r2 ? c.z - first : 0u,
            d3 = r3 ? c.w - first : 0u;
  const u32 w0 = bits[d0 >> 5], w1 = bits[d1 >> 5], w2 = bits[d2 >> 5], w3 = bits[d3 >> 5];
  u32 m = 0;
  m |= (r0 && ((w0 >> (d0 & 31u)) & 1u)) ? 1u : 0u;
  m |= (r1 && ((w1 >> (d1 & 31u)) & 1u)) ? 2u : 0u;
  m |= (r2 && ((w2 >> (d2 & 31u)) & 1u)) ? 4u : 0u;
  m |= (r3 && ((w3 >> (d3 & 31u)) & 1u)) ? 8u : 0u;
  return m;
}

// 4-bit mask of the values of group j*64+lane that exist in a wave tile of nints values.
__device__ __forceinline__ u32 valid4(u32 nints, int j, int lane) {
  const u32 first = 4u * (j * 64 + lane);
  const u32 left = nints > first ? nints - first : 0u;
  return left >= 4u ? 15u : ((1u << left) - 1u);
}

// out[0 .. *n_out) = the values a_i of the stream (a_i = prev + d_0 + ... + d_i,
// wrap-around) with a_i in b[0..n_b), in stream order.  b ascending.  wt_vpre +
// blk_pre: sum of everything before a wave tile, prev included (made by
// k_value_sums, k_scan_u32_blocks, k_scan_u32_top).  One workgroup per tile of
// 16 wave tiles, header and look-back words zero before the launch.
// Meant for ascending streams (posting lists): a wave whose values are not
// ascending still gets the right answer but may walk much of b.
template <int V>
__global__ __launch_bounds__(kBlock, 4) void k_delta_intersect(
    const u8* __restrict__ in, u32 count, Dir dir, const u32* __restrict__ wt_vpre, const u32* __restrict__ blk_pre,
    const u32* __restrict__ b, u32 n_b, const u32* __restrict__ bound, u32* __restrict__ out,
    u64* __restrict__ n_out, u32 n_tiles, WsHeader* __restrict__ hdr, LbWords lb, u64* __restrict__ consumed) {
  __shared__ WaveLds s_lds[kWaves];
  __shared__ u32 s_wave_tot[kWaves];
  __shared__ u64 s_prefix;
  __shared__ u32 s_tile;
  const int lane = lane_id(), wave = wave_id();
  if (threadIdx.x == 0) {
    const u32 t = atomicAdd(&hdr->ticket, 1u);
    // a workgroup per tile, a ticket each: the last one drawn leaves the counter as the next call expects it
    if (t == n_tiles - 1u) __hip_atomic_store(&hdr->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_tile = t;
  }
  __syncthreads();
  const u32 tile = __builtin_amdgcn_readfirstlane(s_tile);
  WaveLds& S = s_lds[wave];
  const u64 keylen = ((u64)count + 3) >> 2;
  const u32 n_wt = (u32)((keylen + kWtKeys - 1) / kWtKeys);
  const u8* data_base = in + keylen;
  const u32 wt0 = tile * kLbWt + wave * 4;

  // ---- decode the wave's 4 wave tiles and turn them into values ----
  uint4 v[4][4];
  u32 nints[4];
  // keys, data offsets and preceding sums of the wave's 4 wave tiles up front
  u32 w[4], vpre[4];
  u64 doff[4];
  DirWalk walk;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const u32 wt = wt0 + q;
    w[q] = 0;
    doff[q] = 0;
    vpre[q] = 0;
    nints[q] = 0;
    if (wt < n_wt) {
      const u64 koff = (u64)wt * kWtKeys + 4 * lane;
      if (koff < keylen) w[q] = ld_u32(in + koff);
      doff[q] = walk.offset(dir, wt, lane);
      vpre[q] = wt_vpre[wt] + blk_pre[wt >> 10];
      nints[q] = (u32)min((u64)kWtInts, (u64)count - (u64)wt * kWtInts);
    }
  }
  const u32 before = vpre[0];  // the value preceding the wave's first
  u32 last_value = before;     // the wave's last value
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[q][j] = make_uint4(0, 0, 0, 0);
    if (nints[q]) {
      const u32 total = decode_wavetile<V, false>(w[q], nints[q], data_base + doff[q], lane, S, v[q]);
      if (wt0 + q == n_wt - 1 && lane == 0 && consumed) *consumed = keylen + doff[q] + total;
      u32 run = vpre[q];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint4& c = v[q][j];
        c.y += c.x; c.z += c.y; c.w += c.z;
        const u32 incl = wave_scan_incl(c.w, lane);
        const u32 ex = incl - c.w + run;
        run += wave_last(incl);
        c.x += ex; c.y += ex; c.z += ex; c.w += ex;
      }
      last_value = run;  // absent values are zero deltas
    }
  }

  // ---- the span of the wave's values, and the part of b inside it ----
  u32 vmax = 0, vmin_inv = 0;  // min as the max of the complements
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 m = valid4(nints[q], j, lane);
      const uint4& c = v[q][j];
      if (m & 1u) { vmax = max(vmax, c.x); vmin_inv = max(vmin_inv, ~c.x); }
      if (m & 2u) { vmax = max(vmax, c.y); vmin_inv = max(vmin_inv, ~c.y); }
      if (m & 4u) { vmax = max(vmax, c.z); vmin_inv = max(vmin_inv, ~c.z); }
      if (m & 8u) { vmax = max(vmax, c.w); vmin_inv = max(vmin_inv, ~c.w); }
    }
  vmax = wave_max(vmax);
  const u32 vmin = ~wave_max(vmin_inv);
  u32 lo = 0, hi = 0;
  if (nints[0] && n_b) {  // the wave has values (its first wave tile does) and b has entries
    const u32 w = wt0 >> 2;
    if (vmin >= before && vmax <= last_value && wt0 + 4 < n_wt) {
      // ascending so far as it matters, and not the last wave: the pre-pass knows the places
      lo = bound[w];
      hi = min(n_b, bound[w + 1] + 1u);  // bound[w+1] is where last_value would be
    } else {
      lo = wave_bound<false>(b, n_b, vmin, lane);
      hi = wave_bound<true>(b, n_b, vmax, lane);
    }
  }

  // ---- walk b[lo, hi) through LDS a chunk at a time, look every value up ----
  // Entries that lie close together (last - first < 32768: the usual case when b
  // is about as dense as the stream or denser) become a bitmap, one LDS read per
  // value; otherwise the entries themselves go to LDS and are searched.
  u32* buf = reinterpret_cast<u32*>(S.data);
  static_assert(sizeof(S.data) >= 4096, "the wave's LDS image holds 1024 entries of the sorted array");
  u64 hits = 0;  // bit q*16 + j*4 + k
  for (u64 pos = lo; pos < hi;) {
    // up to 4096 entries at a time if they fit the bitmap, else 1024
    u32 len = (u32)min((u64)4096, (u64)hi - pos);
    const u32 first = b[pos];  // wave-uniform
    u32 last = b[pos + len - 1];
    if (last - first >= 32768u && len > 1024u) {
      len = 1024u;
      last = b[pos + len - 1];
    }
    if (last - first < 32768u) {
      const u32 nw = ((last - first) >> 5) + 1u;
      for (u32 e = lane; e < nw; e += 64) buf[e] = 0u;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (u32 e = lane; e < len; e += 64) {
        const u32 d = b[pos + e] - first;
        atomicOr(&buf[d >> 5], 1u << (d & 31u));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          hits |= (u64)lds_bits4(buf, first, last, v[q][j], valid4(nints[q], j, lane)) << (q * 16 + j * 4);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const u32 e = i * 64 + lane;
        buf[e] = e < len ? b[pos + e] : 0xFFFFFFFFu;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const u32 top = len > 1u ? 1u << (31 - __builtin_clz(len - 1u)) : 0u;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          hits |= (u64)lds_find4(buf, len, top, first, last, v[q][j], valid4(nints[q], j, lane)) << (q * 16 + j * 4);
    }
    __builtin_amdgcn_wave_barrier();
    pos += len;
  }

  // ---- the tile's place in the output ----
  const u32 wave_tot = wave_sum((u32)__popcll(hits));
  if (lane == 0) s_wave_tot[wave] = wave_tot;
  __syncthreads();
  if (wave == 0) {
    const u32 tile_tot = s_wave_tot[0] + s_wave_tot[1] + s_wave_tot[2] + s_wave_tot[3];
    const u64 excl = lb2_exclusive_prefix<1>(lb, tile, tile_tot, lane, &hdr->error);
    if (lane == 0) {
      s_prefix = excl;
      if (tile == n_tiles - 1) *n_out = excl + tile_tot;
    }
  }
  __syncthreads();
  u64 base = s_prefix;
  for (int w = 0; w < wave; ++w) base += s_wave_tot[w];
  if (wave_tot == 0) return;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 m = (u32)(hits >> (q * 16 + j * 4)) & 15u;
      const u32 c = (u32)__popc(m);
      const u32 incl = wave_scan_incl(c, lane);
      u32* p = out + base + (incl - c);
      base += wave_last(incl);
      const uint4& r = v[q][j];
      if (m & 1u) *p++ = r.x;
      if (m & 2u) *p++ = r.y;
      if (m & 4u) *p++ = r.z;
      if (m & 8u) *p++ = r.w;
    }
}

// ---------------------------------------------------------------------------
// Two delta-coded streams intersected without either being written out decoded.
//
// Both streams get the pre-pass of the two-pass differential decoder, which leaves
// for every wave tile the value that precedes it (vpre[wt] + blk[wt >> 10]): for an
// ascending stream B that is a sorted skip list with one entry per 1024 values.  A
// wavefront decodes 2048 values of stream A into registers, finds with two 64-way
// searches in B's skip list the wave tiles of B that can hold any of them, decodes
// those one at a time into its LDS image -- as a bitmap when the tile's values lie
// within 32768 of each other, else as the sorted values themselves -- and looks
// its own values up there.  Hit counts travel through the look-back as in
// k_delta_intersect.  B must ascend without wrapping; A may be anything (a wave
// whose values jump about walks more of B).
constexpr int kSxQ = 2;                 // wave tiles of stream A per wavefront
constexpr int kSxWaves = 4;             // wavefronts per workgroup = per look-back tile (16 measured 3-10 % slower)

// First i in [0, n) with f(i) >= key (GT false) or f(i) > key (GT true), n if none; f ascending.
// key and the result are wave-uniform; 64 probes a step.
template <bool GT, class F>
__device__ __forceinline__ u32 wave_bound_fn(F f, u32 n, u32 key, int lane) {
  u32 lo = 0, hi = n;
  while (lo < hi) {
    const u32 step = (hi - lo + 63u) >> 6;
    const u64 idx = (u64)lo + (u64)lane * step;
    bool ok = true;
    if (idx < hi) {
      const u32 x = f((u32)idx);
      ok = GT ? x > key : x >= key;
    }
    const u64 m = __ballot(ok);
    const u32 first = m ? (u32)__builtin_ctzll(m) : 64u;
    if (first == 0) {
      hi = lo;
    } else {
      const u64 nh = (u64)lo + (u64)first * step;
      lo = lo + (first - 1) * step + 1;
      hi = (u32)min((u64)hi, nh);
    }
  }
  return lo;
}

// Which of the wave's values v[Q][4] (nints[q] of them valid per wave tile, all within [vmin, vmax]) occur among
// the values c[4] of one wave tile of an ascending stream B (absolute values, bn of them valid, t_last the last)?
// Bit q*16 + j*4 + k of the result stands for value k of v[q][j].  buf: 1024 words of the wave's own LDS.
template <int Q>
__device__ __forceinline__ u32 probe_wavetile(const uint4 (&c)[4], u32 bn, u32 t_last, u32 vmin, u32 vmax,
                                              const uint4 (&v)[Q][4], const u32 (&nints)[Q], u32* buf, int lane) {
  static_assert(Q <= 2, "hit bits are kept in 32 bits");
  const u32 t_first = (u32)__builtin_amdgcn_readfirstlane((int)c[0].x);
  if (t_last < vmin || t_first > vmax) return 0u;  // wave-uniform
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // A bitmap of B's values (one LDS read per value of A) where the values to enter lie within 32768 of each
  // other: the whole wave tile when it is that dense, else only its values inside the span of the wave's own.
  u32 hits = 0;  // bit q*16 + j*4 + k
  u32 inr = 0xFFFFu, first = t_first, last = t_last, below = 0;
  const bool whole = t_last - t_first < 32768u;
  if (whole) {
    inr = valid4(bn, 0, lane) | (valid4(bn, 1, lane) << 4) | (valid4(bn, 2, lane) << 8) | (valid4(bn, 3, lane) << 12);
  } else {
    u32 fmax = 0, fmin_inv = 0;
    inr = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 m = valid4(bn, j, lane);
      const u32 x[4] = {c[j].x, c[j].y, c[j].z, c[j].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool valid = (m >> k) & 1u;
        const bool in = valid && x[k] >= vmin && x[k] <= vmax;
        below += (valid && x[k] < vmin) ? 1u : 0u;
        if (in) {
          inr |= 1u << (4 * j + k);
          fmax = max(fmax, x[k]);
          fmin_inv = max(fmin_inv, ~x[k]);
        }
      }
    }
    if (__ballot(inr != 0u) == 0ull) return 0u;  // wave-uniform: nothing of this tile can meet the wave's values
    first = ~wave_max(fmin_inv);
    last = wave_max(fmax);
  }
  if (last - first < 32768u) {
    const u32 span = last - first;  // a value outside it (B not ascending) is dropped, never written out of bounds
    const u32 nw = (span >> 5) + 1u;
    for (u32 e = lane; e < nw; e += 64) buf[e] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 d0 = c[j].x - first, d1 = c[j].y - first, d2 = c[j].z - first, d3 = c[j].w - first;
      if ((inr & (1u << (4 * j))) && d0 <= span) atomicOr(&buf[d0 >> 5], 1u << (d0 & 31u));
      if ((inr & (2u << (4 * j))) && d1 <= span) atomicOr(&buf[d1 >> 5], 1u << (d1 & 31u));
      if ((inr & (4u << (4 * j))) && d2 <= span) atomicOr(&buf[d2 >> 5], 1u << (d2 & 31u));
      if ((inr & (8u << (4 * j))) && d3 <= span) atomicOr(&buf[d3 >> 5], 1u << (d3 & 31u));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        hits |= lds_bits4(buf, first, last, v[q][j], valid4(nints[q], j, lane)) << (q * 16 + j * 4);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 m = valid4(bn, j, lane);
      uint4 x = c[j];
      if (!(m & 1u)) x.x = 0xFFFFFFFFu;
      if (!(m & 2u)) x.y = 0xFFFFFFFFu;
      if (!(m & 4u)) x.z = 0xFFFFFFFFu;
      if (!(m & 8u)) x.w = 0xFFFFFFFFu;
      *reinterpret_cast<uint4*>(buf + 4 * (j * 64 + lane)) = x;  // value i of the wave tile at buf[i]
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // B ascends, so the values inside the span are buf[lo .. lo + len): search a window of the array that
    // holds them -- a power of two of entries, moved down where it would reach past the array's end (what
    // it then takes in below lo is smaller than every candidate and sorted with the rest)
    const u32 lo = wave_sum(below), len = wave_sum((u32)__popc(inr));
    const u32 top = len > 1u ? 1u << (31 - __builtin_clz(len - 1u)) : 0u;
    const u32 win = top ? 2u * top : 1u;
    const u32 wb = min(lo, (u32)kWtInts - win);
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        hits |= lds_find4(buf + wb, win, top, first, last, v[q][j], valid4(nints[q], j, lane)) << (q * 16 + j * 4);
  }
  __builtin_amdgcn_wave_barrier();
  return hits;
}

// range[w] = the wave tiles [x, y) of stream B that wavefront w of k_stream_intersect (wave tiles kSxQ w ..
// of stream A) can meet if A ascends there: from the tile ahead of the first whose preceding value is >= the
// value preceding the wave's first, to the last whose preceding value is <= the wave's last value (= the
// value preceding the next wave's first).  One thread per wavefront; the last wavefront searches for itself.
__global__ __launch_bounds__(kBlock) void k_stream_bounds(const u32* __restrict__ a_vpre, const u32* __restrict__ a_blk,
                                                          u32 a_nwt, const u32* __restrict__ b_vpre,
                                                          const u32* __restrict__ b_blk, u32 b_nwt,
                                                          uint2* __restrict__ range) {
  const u32 w = blockIdx.x * kBlock + threadIdx.x;
  const u32 wt = w * kSxQ;
  if (wt + kSxQ >= a_nwt) return;
  const u32 key_lo = a_vpre[wt] + a_blk[wt >> 10];
  const u32 key_hi = a_vpre[wt + kSxQ] + a_blk[(wt + kSxQ) >> 10];
  u32 lo = 0, hi = b_nwt;
  while (lo < hi) {  // first i with before_b(i) >= key_lo
    const u32 mid = lo + ((hi - lo) >> 1);
    if (b_vpre[mid] + b_blk[mid >> 10] < key_lo) lo = mid + 1; else hi = mid;
  }
  const u32 x = lo ? lo - 1u : 0u;
  hi = b_nwt;  // first i with before_b(i) > key_hi; not below lo when key_hi >= key_lo
  if (key_hi < key_lo) lo = 0;
  while (lo < hi) {
    const u32 mid = lo + ((hi - lo) >> 1);
    if (b_vpre[mid] + b_blk[mid >> 10] <= key_hi) lo = mid + 1; else hi = mid;
  }
  range[w] = make_uint2(x, lo);
}

// Five waves per SIMD (96 VGPRs, two or three dwords spilled): the kernel waits on chains of dependent loads,
// and a fifth wave is worth 8-10 % (2^27 postings against as many: 0.577 -> 0.535 ms); a sixth (80 VGPRs, ten
// dwords spilled) gives it back.
template <int VA, int VB, int NW>
__global__ __launch_bounds__(NW * 64, 5) void k_stream_intersect(
    const u8* __restrict__ a_in, u32 a_count, Dir a_dir, const u32* __restrict__ a_vpre,
    const u32* __restrict__ a_blk, const u8* __restrict__ b_in, u32 b_count, Dir b_dir,
    const u32* __restrict__ b_vpre, const u32* __restrict__ b_blk, const uint2* __restrict__ range,
    u32* __restrict__ out, u64* __restrict__ n_out, u32 n_tiles, WsHeader* __restrict__ hdr, LbWords lb) {
  __shared__ WaveLds s_lds[NW];
  __shared__ u32 s_wave_tot[NW];
  __shared__ u64 s_prefix;
  __shared__ u32 s_tile;
  const int lane = lane_id(), wave = wave_id();
  if (threadIdx.x == 0) {
    const u32 t = atomicAdd(&hdr->ticket, 1u);
    if (t == n_tiles - 1u) __hip_atomic_store(&hdr->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_tile = t;
  }
  __syncthreads();
  const u32 tile = __builtin_amdgcn_readfirstlane(s_tile);
  WaveLds& S = s_lds[wave];
  const u64 a_keylen = ((u64)a_count + 3) >> 2;
  const u32 a_nwt = (u32)((a_keylen + kWtKeys - 1) / kWtKeys);
  const u8* a_data = a_in + a_keylen;
  const u32 wt0 = (tile * NW + wave) * kSxQ;

  const u64 b_keylen = ((u64)b_count + 3) >> 2;
  const u32 b_nwt = (u32)((b_keylen + kWtKeys - 1) / kWtKeys);
  const u8* b_data = b_in + b_keylen;
  // What a wave tile of B needs before it can be decoded, requested a tile ahead of its use.
  struct BTile { u32 w; u64 doff; u32 before; };
  DirWalk bwalk;
  auto fetch = [&](u32 t) {
    BTile x;
    const u64 koff = (u64)t * kWtKeys + 4 * lane;
    x.w = koff < b_keylen ? ld_u32(b_in + koff) : 0u;
    x.doff = bwalk.offset(b_dir, t, lane);
    x.before = b_vpre[t] + b_blk[t >> 10];
    return x;
  };
  // the pre-pass's guess of the wave's share of B (right when A ascends here), and its first tile on the way
  const bool guessed = wt0 + kSxQ < a_nwt;
  uint2 guess = make_uint2(0u, 0u);
  if (guessed) {
    const uint2 g = range[wt0 / kSxQ];
    guess = make_uint2((u32)__builtin_amdgcn_readfirstlane((int)g.x), (u32)__builtin_amdgcn_readfirstlane((int)g.y));
  }
  BTile cur{};
  if (guess.x < guess.y) cur = fetch(guess.x);

  // ---- the wave's values of stream A ----
  uint4 v[kSxQ][4];
  u32 nints[kSxQ];
  u32 before_a = 0, last_a = 0;
  {
    u32 w[kSxQ], vpre[kSxQ];
    u64 doff[kSxQ];
    DirWalk walk;
#pragma unroll
    for (int q = 0; q < kSxQ; ++q) {
      const u32 wt = wt0 + q;
      w[q] = 0; doff[q] = 0; vpre[q] = 0; nints[q] = 0;
      if (wt < a_nwt) {
        const u64 koff = (u64)wt * kWtKeys + 4 * lane;
        if (koff < a_keylen) w[q] = ld_u32(a_in + koff);
        doff[q] = walk.offset(a_dir, wt, lane);
        vpre[q] = a_vpre[wt] + a_blk[wt >> 10];
        nints[q] = (u32)min((u64)kWtInts, (u64)a_count - (u64)wt * kWtInts);
      }
    }
    before_a = last_a = vpre[0];
#pragma unroll
    for (int q = 0; q < kSxQ; ++q) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[q][j] = make_uint4(0, 0, 0, 0);
      if (nints[q]) {
        decode_wavetile<VA, false>(w[q], nints[q], a_data + doff[q], lane, S, v[q]);
        u32 run = vpre[q];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint4& c = v[q][j];
          c.y += c.x; c.z += c.y; c.w += c.z;
          const u32 incl = wave_scan_incl(c.w, lane);
          const u32 ex = incl - c.w + run;
          run += wave_last(incl);
          c.x += ex; c.y += ex; c.z += ex; c.w += ex;
        }
        last_a = run;
      }
    }
  }
  u32 vmax = 0, vmin_inv = 0;
#pragma unroll
  for (int q = 0; q < kSxQ; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 m = valid4(nints[q], j, lane);
      const uint4& c = v[q][j];
      if (m & 1u) { vmax = max(vmax, c.x); vmin_inv = max(vmin_inv, ~c.x); }
      if (m & 2u) { vmax = max(vmax, c.y); vmin_inv = max(vmin_inv, ~c.y); }
      if (m & 4u) { vmax = max(vmax, c.z); vmin_inv = max(vmin_inv, ~c.z); }
      if (m & 8u) { vmax = max(vmax, c.w); vmin_inv = max(vmin_inv, ~c.w); }
    }
  vmax = wave_max(vmax);
  const u32 vmin = ~wave_max(vmin_inv);

  // ---- the wave tiles of stream B that can hold any of them ----
  u32 t_lo = 0, t_hi = 0;  // [t_lo, t_hi)
  if (nints[0] && b_count) {
    if (guessed && vmin >= before_a && vmax <= last_a) {  // ascending as far as it matters
      t_lo = guess.x;
      t_hi = guess.y;
    } else {
      auto before = [&](u32 i) { return b_vpre[i] + b_blk[i >> 10]; };  // the value preceding wave tile i of B
      // wave tile t holds values from before(t) to before(t + 1): the first that can reach vmin is the one
      // ahead of the first whose predecessor is already >= vmin; the last is the last with before(t) <= vmax
      const u32 i_lo = wave_bound_fn<false>(before, b_nwt, vmin, lane);
      t_lo = i_lo ? i_lo - 1u : 0u;
      t_hi = wave_bound_fn<true>(before, b_nwt, vmax, lane);
      if (t_lo < t_hi && !(guess.x < guess.y && guess.x == t_lo)) cur = fetch(t_lo);
    }
  }

  u32* buf = reinterpret_cast<u32*>(S.data);
  static_assert(sizeof(S.data) >= 4096, "the wave's LDS image holds the 1024 values of a wave tile of B");
  u32 hits = 0;  // bit q*16 + j*4 + k
  for (u32 t = t_lo; t < t_hi; ++t) {
    const BTile now = cur;
    if (t + 1 < t_hi) cur = fetch(t + 1);
    const u32 bn = (u32)min((u64)kWtInts, (u64)b_count - (u64)t * kWtInts);
    uint4 c[4];
    decode_wavetile<VB, false>(now.w, bn, b_data + now.doff, lane, S, c);
    u32 run = now.before;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint4& x = c[j];
      x.y += x.x; x.z += x.y; x.w += x.z;
      const u32 incl = wave_scan_incl(x.w, lane);
      const u32 ex = incl - x.w + run;
      run += wave_last(incl);
      x.x += ex; x.y += ex; x.z += ex; x.w += ex;
    }
    hits |= probe_wavetile<kSxQ>(c, bn, run, vmin, vmax, v, nints, buf, lane);  // run: B's last value so far
  }

  // ---- the tile's place in the output ----
  const u32 wave_tot = wave_sum((u32)__popc(hits));
  if (lane == 0) s_wave_tot[wave] = wave_tot;
  __syncthreads();
  if (wave == 0) {
    u32 tile_tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) tile_tot += s_wave_tot[w];
    const u64 excl = lb2_exclusive_prefix<1>(lb, tile, tile_tot, lane, &hdr->error);
    if (lane == 0) {
      s_prefix = excl;
      if (tile == n_tiles - 1) *n_out = excl + tile_tot;
    }
  }
  __syncthreads();
  u64 base = s_prefix;
  for (int w = 0; w < wave; ++w) base += s_wave_tot[w];
  if (wave_tot == 0) return;
#pragma unroll
  for (int q = 0; q < kSxQ; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 m = (hits >> (q * 16 + j * 4)) & 15u;
      const u32 cnt = (u32)__popc(m);
      const u32 incl = wave_scan_incl(cnt, lane);
      u32* p = out + base + (incl - cnt);
      base += wave_last(incl);
      const uint4& r = v[q][j];
      if (m & 1u) *p++ = r.x;
      if (m & 2u) *p++ = r.y;
      if (m & 4u) *p++ = r.z;
      if (m & 8u) *p++ = r.w;
    }
}

// ---------------------------------------------------------------------------
// Many pairs of delta-coded lists intersected in one launch: a wavefront per pair, no directory, no
// look-back (pair i writes its hits to a place of its own).  The wave walks list A 2048 values at a time
// (data offset and running sum carried in registers, as the batched decoder does) and, for each such
// piece, list B one wave tile at a time from the first tile that reached the previous piece, up to the
// first whose preceding value lies beyond the piece.  B must ascend; an A that steps back makes the wave
// start B over.  Results equal svbgpu_intersect_streams pair by pair.
#ifndef SVB_ISB_WAVES
#define SVB_ISB_WAVES 1
#endif
constexpr int kIsbWaves = SVB_ISB_WAVES;  // pairs (= wavefronts) per workgroup
template <int V>
__global__ __launch_bounds__(64 * kIsbWaves) void k_intersect_batch(
    const u8* __restrict__ a_bytes, const u64* __restrict__ a_off, const u32* __restrict__ a_counts,
    const u32* __restrict__ a_prevs, const u8* __restrict__ b_bytes, const u64* __restrict__ b_off,
    const u32* __restrict__ b_counts, const u32* __restrict__ b_prevs, u32 npairs, u32* __restrict__ out,
    const u64* __restrict__ out_off, u32* __restrict__ n_out) {
  __shared__ WaveLds s_lds[kIsbWaves];
  const int lane = lane_id(), wave = wave_id();
  const u32 pair = blockIdx.x * kIsbWaves + wave;
  if (pair >= npairs) return;  // wave-uniform; no workgroup barrier below
  WaveLds& S = s_lds[wave];
  u32* buf = reinterpret_cast<u32*>(S.data);
  const u32 a_count = a_counts[pair], b_count = b_counts[pair];
  const u8* a_in = a_bytes + a_off[pair];
  const u8* b_in = b_bytes + b_off[pair];
  u32* dst = out + out_off[pair];
  const u32 a_keylen = (u32)(((u64)a_count + 3) >> 2), b_keylen = (u32)(((u64)b_count + 3) >> 2);
  const u32 a_nwt = (a_keylen + kWtKeys - 1) / kWtKeys, b_nwt = (b_keylen + kWtKeys - 1) / kWtKeys;
  auto load_keys = [&](const u8* in, u32 keylen, u32 wt) -> u32 {  // the wave tile's keys; past the stream's keys: its last
    const u32 left = keylen - wt * (u32)kWtKeys;                    // >= 1
    return ld_u32(in + (u64)wt * kWtKeys + min(4u * (u32)lane, (left - 1u) & ~3u));
  };
  auto integrate = [&](uint4 (&c)[4], u32& run) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint4& x = c[j];
      x.y += x.x; x.z += x.y; x.w += x.z;
      const u32 incl = wave_scan_incl(x.w, lane);
      const u32 ex = incl - x.w + run;
      run += wave_last(incl);
      x.x += ex; x.y += ex; x.z += ex; x.w += ex;
    }
  };
  // where B's walk starts for the next piece of A: a wave tile, its data offset, the value preceding it
  const u32 b_seed = b_prevs ? b_prevs[pair] : 0u;
  u32 rs_t = 0, rs_before = b_seed;
  u64 rs_doff = 0;
  u32 a_run = a_prevs ? a_prevs[pair] : 0u;
  u64 a_doff = 0;
  u32 total_hits = 0;
  for (u32 wt0 = 0; wt0 < a_nwt; wt0 += kSxQ) {
    uint4 v[kSxQ][4];
    u32 nints[kSxQ];
#pragma unroll
    for (int q = 0; q < kSxQ; ++q) {
      const u32 wt = wt0 + q;
      nints[q] = wt < a_nwt ? min((u32)kWtInts, a_count - wt * (u32)kWtInts) : 0u;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[q][j] = make_uint4(0, 0, 0, 0);
      if (nints[q]) {
        const u32 w = load_keys(a_in, a_keylen, wt);
        a_doff += decode_wavetile<V, false>(w, nints[q], a_in + a_keylen + a_doff, lane, S, v[q]);
        integrate(v[q], a_run);
      }
    }
    u32 vmax = 0, vmin_inv = 0;
#pragma unroll
    for (int q = 0; q < kSxQ; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32 m = valid4(nints[q], j, lane);
        const uint4& c = v[q][j];
        if (m & 1u) { vmax = max(vmax, c.x); vmin_inv = max(vmin_inv, ~c.x); }
        if (m & 2u) { vmax = max(vmax, c.y); vmin_inv = max(vmin_inv, ~c.y); }
        if (m & 4u) { vmax = max(vmax, c.z); vmin_inv = max(vmin_inv, ~c.z); }
        if (m & 8u) { vmax = max(vmax, c.w); vmin_inv = max(vmin_inv, ~c.w); }
      }
    vmax = wave_max(vmax);
    const u32 vmin = ~wave_max(vmin_inv);
    // every tile before rs_t ends at or below rs_before: it can be skipped only if that lies below vmin
    if (rs_t && vmin <= rs_before) { rs_t = 0; rs_before = b_seed; rs_doff = 0; }
    u32 t = rs_t, run = rs_before;
    u64 doff = rs_doff;
    bool kept = false;
    u32 hits = 0;
    while (t < b_nwt && run <= vmax) {  // run = the value preceding tile t: nothing from t on is below it
      const u32 before = run;
      const u64 doff0 = doff;
      const u32 bn = min((u32)kWtInts, b_count - t * (u32)kWtInts);
      uint4 c[4];
      const u32 w = load_keys(b_in, b_keylen, t);
      doff += decode_wavetile<V, false>(w, bn, b_in + b_keylen + doff, lane, S, c);
      integrate(c, run);
      if (run >= vmin) {
        if (!kept) { kept = true; rs_t = t; rs_before = before; rs_doff = doff0; }
        hits |= probe_wavetile<kSxQ>(c, bn, run, vmin, vmax, v, nints, buf, lane);
      }
      ++t;
    }
    if (!kept) { rs_t = t; rs_before = run; rs_doff = doff; }  // everything walked lies below this piece
#pragma unroll
    for (int q = 0; q < kSxQ; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32 m = (hits >> (q * 16 + j * 4)) & 15u;
        const u32 cnt = (u32)__popc(m);
        const u32 incl = wave_scan_incl(cnt, lane);
        u32* p = dst + total_hits + (incl - cnt);
        total_hits += wave_last(incl);
        const uint4& r = v[q][j];
        if (m & 1u) *p++ = r.x;
        if (m & 2u) *p++ = r.y;
        if (m & 4u) *p++ = r.z;
        if (m & 8u) *p++ = r.w;
      }
  }
  if (lane == 0) n_out[pair] = total_hits;
}

}  // namespace svb

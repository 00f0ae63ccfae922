// The streaming encoder: one short-lived workgroup per tile, handed out by the hardware in order.
//
// k_encode_pipe keeps two fat tiles in flight per persistent workgroup (64 values per lane in
// registers, four waves per SIMD, a ticket per tile); what bounds it is not the look-back but the
// phases of a workgroup, which its few resident waves cannot overlap.  Here a tile is NW wave tiles, one
// per wave (sixteen values per lane), a workgroup lives for exactly one tile and the CU holds eight
// waves per SIMD, so that loads, classification, look-back and stores of different workgroups overlap
// the way they do in the decoder.  Tile cycle (reference src/streamvbyte_x64_encode.c:55-97 is the
// serial loop this replaces):
//
//   load -> classify -> publish the tile's byte count and SEND the look-back's reads
//        -> assemble the wave tiles' bytes in LDS (needs no offset)   | the reads travel meanwhile
//        -> collect the look-back -> copy the LDS images out, shifted onto the destination's 16-byte grid.
//
// Tile index = blockIdx.x, no ticket: one same-address atomic per tile caps the whole device at 60-90
// tiles per microsecond, and this kernel starts over 100.  The hardware starts workgroups in index
// order (round-robin over the XCDs), so a tile's predecessors have published or are about to; the
// spins are bounded like every spin in this library.
//
// Look-back words: a tile word is 4 bytes (flag | byte count), so that ONE 16-byte load per lane reads
// the words of 256 consecutive tiles; with over 100 tiles starting per microsecond and 2-4 us from
// a word's store to its arrival in a reader's registers, a tile must read several hundred
// predecessors directly -- nobody can have summed them up for it yet.  Older tiles are read through
// the group words of svb_lookback.h (one per 16 tiles, written by the group's last tile).
#pragma once
#include "svb_encode.h"

namespace svb {

// Values of one wave tile -> transformed values, the wave tile's four key bytes per lane (k4_out)
// and the lane's data bytes.
template <int V, bool DELTA, bool ZZ, bool QUANT>
__device__ __forceinline__ u32 transform_classify(uint4 (&x)[4], const u32* __restrict__ in, u64 wt_first, u32 prev,
                                                  u32& k4_out) {
  if (QUANT) {
    const float scale = __uint_as_float(prev);
    auto q1 = [scale](u32 bits) -> u32 { return (u32)(int)rintf(__fmul_rn(__uint_as_float(bits), scale)); };
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = make_uint4(q1(x[j].x), q1(x[j].y), q1(x[j].z), q1(x[j].w));
  }
  if (DELTA) {
    u32 carry = prev;
    if (wt_first > 0) carry = in[wt_first - 1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint4 c = x[j];
      const u32 pv = wave_shift_up1(c.w, carry);
      carry = wave_last(c.w);
      x[j] = make_uint4(c.x - pv, c.y - c.x, c.z - c.y, c.w - c.z);
    }
  }
  if (ZZ) {
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = make_uint4(zz_enc(x[j].x), zz_enc(x[j].y), zz_enc(x[j].z), zz_enc(x[j].w));
  }
  u32 k4 = classify_key<V>(x[0]) | (classify_key<V>(x[1]) << 8) | (classify_key<V>(x[2]) << 16) |
           (classify_key<V>(x[3]) << 24);
  asm volatile("" : "+v"(k4));  // the four keys stay in ONE register
  k4_out = k4;
  return word_len<V>(k4, 16u);
}

// The data bytes of one full wave tile into the wave's LDS image, byte 0 of the image = first data
// byte of the wave tile (the destination's alignment is not known yet), and its keys into S.keys.
// Same assembly as encode_wavetile.  Not for wave tiles whose keys are all zero (see copyout_wavetile).
template <int V>
__device__ __forceinline__ void assemble_wavetile(const uint4 (&x)[4], u32 k4, int lane, EncWaveLds& S) {
  const u32 lens4 = key_lens4<V>(k4, 16u);
  u32 g[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) g[j] = ((k4 >> (8 * j)) & 0xFFu) | (((lens4 >> (8 * j)) & 0xFFu) << 24);
  u64 packed = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) packed |= (u64)gi_total(g[j]) << (16 * j);
  const u32 inc_lo = wave_scan_incl((u32)packed, lane), inc_hi = wave_scan_incl((u32)(packed >> 32), lane);
  const u64 incl = ((u64)inc_hi << 32) | inc_lo;
  const u64 tots = ((u64)wave_last(inc_hi) << 32) | wave_last(inc_lo);
  const u64 excl = incl - packed;
  const u32 t0 = (u32)(tots & 0xFFFFu), t1 = (u32)((tots >> 16) & 0xFFFFu), t2 = (u32)((tots >> 32) & 0xFFFFu),
            t3 = (u32)(tots >> 48);
  const u32 pj[4] = {(u32)(excl & 0xFFFFu), t0 + (u32)((excl >> 16) & 0xFFFFu), t0 + t1 + (u32)((excl >> 32) & 0xFFFFu),
                     t0 + t1 + t2 + (u32)(excl >> 48)};
#pragma unroll
  for (int j = 0; j < 4; ++j) S.keys[j * 64 + lane] = (u8)gi_key(g[j]);
  if (V == V1234) {
    u32 head[4];
#pragma unroll
    for (int j = 3; j >= 0; --j) {
      const uint4 d = pack_group<V>(x[j], g[j]);
      const GroupSlot s = group_slot<true>(S.data, pj[j], g[j]);
      store_dword<0, true>(s, d);
      store_dword<1, true>(s, d);
      store_dword<2, true>(s, d);
      store_dword<3, true>(s, d);
      store_dword<4, true>(s, d);
      head[j] = d.x;
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) store_head<true>(S.data, head[j], pj[j], g[j]);
  } else {
    const u32 total = t0 + t1 + t2 + t3;
    const u32 nz = (total + 15u + 16u) >> 4;  // the copy-out reads up to 19 bytes past the last data byte
    uint4* l16 = reinterpret_cast<uint4*>(S.data);
    for (u32 c = lane; c < nz; c += 64) l16[c] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      or_into_image(S.data, pj[j], gi_total(g[j]), pack_group<V>(x[j], g[j]));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// The wave's LDS image (total bytes, image byte i <-> dst[i]) and its 256 keys to global memory.
// Whole 16-byte chunks of the destination are read from the image as the five aligned dwords that
// hold them and shifted into place (the image was assembled before dst was known); the ragged head
// and tail, which neighbouring waves share, go byte by byte.
template <bool NTS>
__device__ __forceinline__ void copyout_wavetile(const EncWaveLds& S, u32 total, int lane, u8* dst, u8* kdst) {
  const u32 a = (u32)((uintptr_t)dst & 15u);
  const u32 end = a + total;        // destination bytes [a, end) relative to the aligned base
  const u32 c_lo = (a + 15u) >> 4;  // first chunk lying fully inside
  const u32 c_hi = end >> 4;        // one past the last such chunk
  u8* g0 = dst - a;
  uint4* g16 = reinterpret_cast<uint4*>(g0);
  for (u32 c = c_lo + lane; c < c_hi; c += 64) {
    const u32 off = 16u * c - a;  // image byte of the chunk's first byte
    const u32* q = reinterpret_cast<const u32*>(S.data + (off & ~3u));
    const u32 d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
    uint4 v;
    v.x = __builtin_amdgcn_alignbyte(d1, d0, off);
    v.y = __builtin_amdgcn_alignbyte(d2, d1, off);
    v.z = __builtin_amdgcn_alignbyte(d3, d2, off);
    v.w = __builtin_amdgcn_alignbyte(d4, d3, off);
    st_stream16<NTS>(g16 + c, v);
  }
  const u32 hend = min(16u * c_lo, end);     // head bytes [a, hend)
  const u32 tstart = max(16u * c_hi, hend);  // tail bytes [tstart, end)
  if (lane < 16) {
    const u32 i = a + lane;
    if (i < hend) g0[i] = S.data[i - a];
  } else if (lane < 32) {
    const u32 i = tstart + (lane - 16);
    if (i < end) g0[i] = S.data[i - a];
  }
  if ((((uintptr_t)kdst) & 3u) == 0) {
    *reinterpret_cast<u32*>(kdst + 4 * lane) = reinterpret_cast<const u32*>(S.keys)[lane];
  } else {
#pragma unroll
    for (int b = 0; b < 4; ++b) kdst[4 * lane + b] = S.keys[4 * lane + b];
  }
}

// QW wave tiles per wave (a tile is NW * QW wave tiles): the wait for the look-back is the same
// whatever the tile holds, so more bytes per resident wave make it a smaller share of a tile's life.
// Only the wave's first wave tile is assembled ahead of the look-back (one LDS image per wave); the
// others follow once the first has been copied out.
template <int V, bool DELTA, bool ZZ, bool QUANT, int NW, int NT, int kC, bool STAMPS = false, int QW = 1>
__global__ __launch_bounds__(NW * 64, QW == 1 ? 8 : (NW == 8 ? 6 : 8)) void k_encode_stream(
    const u32* __restrict__ in, u32 count, u32 n_tiles, u8* __restrict__ out, u32 prev,
    WsHeader* __restrict__ hdr, Lb4Words lb, u64* __restrict__ written, u32 dbg, u64* __restrict__ stamps_arg = nullptr) {
  u64* const stamps = STAMPS ? stamps_arg : nullptr;
  __shared__ EncWaveLds s_lds[NW];
  __shared__ u32 s_wt_tot[NW * QW];
  __shared__ u64 s_prefix;

  const int lane = lane_id(), wave = wave_id();
  const u32 tile = blockIdx.x;
  const u64 keylen = ((u64)count + 3) >> 2;
  constexpr u64 kTileInts = (u64)NW * QW * kWtInts;
  const u64 wave_first = (u64)tile * kTileInts + (u64)wave * QW * kWtInts;
  if (stamps && threadIdx.x == 0) stamps[5 * (u64)tile + 0] = __builtin_amdgcn_s_memrealtime();

  uint4 x[QW][4];
  {
    const uint4* src = reinterpret_cast<const uint4*>(in + wave_first);
#pragma unroll
    for (int q = 0; q < QW; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) x[q][j] = ld_stream16<(NT & kNtLoads) != 0>(src + ((u32)lane + (u32)(q * 256 + j * 64)));
  }
  u32 k4[QW];
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    const u32 lane_tot = transform_classify<V, DELTA, ZZ, QUANT>(x[q], in, wave_first + (u64)q * kWtInts, prev, k4[q]);
    const u32 wt_tot = wave_sum(lane_tot);
    if (lane == 0) s_wt_tot[wave * QW + q] = wt_tot;
  }
  lds_barrier();

  // wave 0: publish the tile's byte count, send the look-back's reads
  Lb4Polls<kC> polls;
  u32 tile_tot = 0;
  const bool lookback = wave == 0 && !(dbg & 1u);
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < NW * QW; ++w) tile_tot += s_wt_tot[w];
    if (stamps && lane == 0) stamps[5 * (u64)tile + 1] = __builtin_amdgcn_s_memrealtime();
  }
  if (lookback) polls = lb4_begin<kC>(lb, tile, tile_tot, lane);

  // every wave: the bytes of its first wave tile into its LDS image, while the reads travel.
  // A wave tile whose 256 keys are all zero needs no image: a group's bytes are the low bytes of its
  // four values, one dword at byte 4 * group (the mirror image of the decoders' shortcut).
  EncWaveLds& S = s_lds[wave];
  bool zero_keys = __ballot(k4[0] != 0u) == 0ull;
  if (!zero_keys) assemble_wavetile<V>(x[0], k4[0], lane, S);

  if (wave == 0) {
    u64 excl;
    u32 rounds = 0;
    if (lookback) excl = lb4_resolve<kC>(lb, tile, tile_tot, lane, polls, &hdr->error, &rounds);
    else excl = (u64)tile * (kTileInts * 5 / 2);  // timing experiment: no look-back, plausible (wrong) offsets
    if (stamps && lane == 0) {
      stamps[5 * (u64)tile + 2] = __builtin_amdgcn_s_memrealtime();
      stamps[5 * (u64)tile + 4] = rounds | ((u64)__builtin_amdgcn_s_getreg(20 | (3 << 11)) << 32) |
                                   ((u64)__builtin_amdgcn_s_getreg(4 | (15 << 11)) << 40);  // XCC_ID, HW_ID[15:0]
    }
    if (lane == 0) {
      s_prefix = excl;
      const u64 last_tile = keylen ? (keylen - 1) / ((u64)NW * QW * kWtKeys) : 0;
      if (tile == last_tile && written) *written = keylen + excl + tile_tot;
    }
  }
  lds_barrier();
  u64 doff = s_prefix;
  for (int w = 0; w < wave * QW; ++w) doff += s_wt_tot[w];
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    if (q > 0) {
      zero_keys = __ballot(k4[q] != 0u) == 0ull;
      if (!zero_keys) assemble_wavetile<V>(x[q], k4[q], lane, S);
    }
    u8* dst = out + keylen + doff;
    u8* kdst = out + ((wave_first + (u64)q * kWtInts) >> 2);
    const u32 total = s_wt_tot[wave * QW + q];
    if (zero_keys) {
      if (V == V1234) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          st_u32(dst + 4u * (j * 64 + lane), x[q][j].x | (x[q][j].y << 8) | (x[q][j].z << 16) | (x[q][j].w << 24));
      }
      st_u32(kdst + 4 * lane, 0u);
    } else {
      copyout_wavetile<(NT & kNtStores) != 0>(S, total, lane, dst, kdst);
      __builtin_amdgcn_wave_barrier();
    }
    doff += total;
  }
  if (stamps && threadIdx.x == 0) stamps[5 * (u64)tile + 3] = __builtin_amdgcn_s_memrealtime();
}

}  // namespace svb

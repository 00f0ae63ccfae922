// Encode path: classify, pack keys, scan lengths, assemble data bytes in LDS.
//
// Replaces the serial encoder loops of the reference
// (src/streamvbyte_x64_encode.c:15-29,55-97; scalar src/streamvbyte_encode.c:14-62;
// 0124 src/streamvbyte_0124_encode.c:12-57,61-113; delta
// src/streamvbytedelta_x64_encode.c:13-75, src/streamvbytedelta_encode.c:39-76).
//
// One workgroup = one tile of 4*QW wave tiles; wave w owns QW consecutive wave
// tiles.  Pass 1 loads the wave's values into registers, classifies them once
// (a one-register descriptor per group) and sums their widths; the tile's byte
// count is published and the tile's data offset obtained by decoupled
// look-back; pass 2 assembles the data bytes of one wave tile at a time in a
// wave-private LDS image laid out on the 16-byte grid of the destination and
// copies it out with aligned 16-byte stores (byte stores for the ragged head
// and tail, which neighbouring waves share).
// HBM traffic per value: 4 (in) + 0.25 (keys) + dbar (data).
#pragma once
#include "svb_device.h"
#include "svb_lookback.h"

namespace svb {

struct __align__(16) EncWaveLds {
  u8 data[kWtDataMax + 48];  // image of one wave tile's data bytes (+16 alignment shift, +slack)
  u8 keys[kWtKeys];
};

// Load the 4 values of the group starting at value index `first`, with bounds.
template <bool NTL = false>
__device__ __forceinline__ uint4 load_group(const u32* __restrict__ in, u64 first, u32 count) {
  uint4 x = make_uint4(0, 0, 0, 0);
  if (first + 4 <= count) {
    if (NTL) x = ld_stream16<true>(in + first);
    else __builtin_memcpy(&x, in + first, 16);
  } else if (first < count) {
    x.x = in[first];
    if (first + 1 < count) x.y = in[first + 1];
    if (first + 2 < count) x.z = in[first + 2];
  }
  return x;
}

// v_ffbh_u32: number of leading zero bits, 0xFFFFFFFF for 0.  Raw instruction
// (the clz builtins are undefined at 0 or add a select).
__device__ __forceinline__ u32 ffbh_raw(u32 v) {
  u32 r;
  asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}

// Group descriptor in one register:
//   bits  7..0   key (four 2-bit codes; absent values code 0)
//   bits 28..24  data length of the group (0..16)
__device__ __forceinline__ u32 gi_key(u32 g) { return g & 0xFFu; }
__device__ __forceinline__ u32 gi_total(u32 g) { return g >> 24; }
// data bits (8 * length) of value k of a group of four present values
template <int V>
__device__ __forceinline__ u32 gi_bits(u32 g, int k) {
  const u32 c = (g >> (2 * k)) & 3u;
  return V == V1234 ? 8u * c + 8u : ((8u << c) >> 1) & ~7u;  // 0124: 0, 8, 16, 32
}

// Classify the four values of a group; nv of them exist (FULL: nv == 4, not read).
// Same codes as code_of<V> (reference src/streamvbyte_encode.c:14-37,
// src/streamvbyte_0124_encode.c:12-24), from the number of leading zero bytes
// z = ffbh >> 3 (3 for both 0 and 1..255): 1234 code = 3 - z; 0124 code = 4 - z
// clamped to 3, and 0 for the value 0.
template <int V, bool FULL>
__device__ __forceinline__ u32 classify_group(const uint4& x, u32 nv) {
  const u32 v[4] = {x.x, x.y, x.z, x.w};
  u32 key, tot;
  if (V == V1234) {
    u32 z[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] = (ffbh_raw(v[k]) >> 3) & 3u;
    key = (z[0] | (z[1] << 2) | (z[2] << 4) | (z[3] << 6)) ^ 0xFFu;
    tot = 16u - (z[0] + z[1] + z[2] + z[3]);
  } else {
    key = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // t = leading zero bytes (7 for v = 0): code = clamp(4 - t, 0, 3)
      const int t = (int)((ffbh_raw(v[k]) >> 3) & 7u);
      const u32 c = (u32)min(max(4 - t, 0), 3);
      key |= c << (2 * k);
      tot += len_of<V>(c);
    }
  }
  if (!FULL && nv < 4u) {  // absent values: code 0, no bytes
    key &= (1u << (2u * nv)) - 1u;
    tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if ((u32)k < nv) tot += len_of<V>((key >> (2 * k)) & 3u);
  }
  u32 g = key | (tot << 24);
  // keep the descriptor in ONE register from here on (left alone, the compiler
  // defers the combination and carries the five parts of every group instead)
  asm volatile("" : "+v"(g));
  return g;
}

// Key byte alone of a group of four present values (what classify_group puts in
// bits 7..0).
template <int V>
__device__ __forceinline__ u32 classify_key(const uint4& x) {
  const u32 v[4] = {x.x, x.y, x.z, x.w};
  u32 key = 0;
  if (V == V1234) {
#pragma unroll
    for (int k = 0; k < 4; ++k) key |= ((ffbh_raw(v[k]) >> 3) & 3u) << (2 * k);
    return key ^ 0xFFu;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t = (int)((ffbh_raw(v[k]) >> 3) & 7u);
    key |= (u32)min(max(4 - t, 0), 3) << (2 * k);
  }
  return key;
}

// Concatenate the data bytes of a group: the low len[k] bytes of each value,
// value 0 first, into 128 bits (d.x = first four bytes).  Values must already
// be < 256^len (true for every classified value; absent values are zero, and
// only follow present ones).
template <int V>
__device__ __forceinline__ uint4 pack_group(const uint4& x, u32 g) {
  const u32 b0 = gi_bits<V>(g, 0), b1 = gi_bits<V>(g, 1), b2 = gi_bits<V>(g, 2);
  const u64 lo = (u64)x.x | ((u64)x.y << b0);  // b0 <= 32
  const u64 hi = (u64)x.z | ((u64)x.w << b2);
  const u32 s = b0 + b1;                       // 0..64 bits taken by values 0 and 1 (1234: >= 16)
  const u64 vlo = lo | (s < 64u ? hi << s : 0ull);
  const u64 vhi = (V != V1234 && s == 0u) ? 0ull : (s < 64u ? hi >> (64u - s) : hi);
  return make_uint4((u32)vlo, (u32)(vlo >> 32), (u32)vhi, (u32)(vhi >> 32));
}

// Byte selector that moves a group's bytes up by a = p & 3 bytes: applied to the
// register pair {hi, lo} it picks bytes 4-a .. 7-a.
__device__ __forceinline__ u32 shift_selector(u32 p) {
  return __builtin_amdgcn_alignbyte(0x08070605u, 0x04030201u, ~p);  // (~p) & 3 = 3 - a
}
// Dword `I` (0..4) of the group's bytes once shifted to start at byte (p & 3)
// of an aligned dword; sel = shift_selector(p).
template <int I>
__device__ __forceinline__ u32 shifted_dword(const uint4& d, u32 sel) {
  const u32 lo = I == 0 ? 0u : (I == 1 ? d.x : (I == 2 ? d.y : (I == 3 ? d.z : d.w)));
  const u32 hi = I == 0 ? d.x : (I == 1 ? d.y : (I == 2 ? d.z : (I == 3 ? d.w : 0u)));
  return __builtin_amdgcn_perm(hi, lo, sel);
}

// 0124: OR the group's bytes into the zeroed LDS image at byte offset p.  An
// image dword can be shared by several neighbouring groups (0124 groups can be
// empty) whose bytes are disjoint, so an LDS atomic OR assembles it whatever
// the order.
__device__ __forceinline__ void or_into_image(u8* image, u32 p, u32 L, const uint4& d) {
  const u32 sel = shift_selector(p);
  u32* q = reinterpret_cast<u32*>(image + (p & ~3u));
  const u32 e = L ? L + (p & 3u) : 0u;  // dword I is touched iff e > 4 I
  if (e > 0u) atomicOr(q + 0, shifted_dword<0>(d, sel));
  if (e > 4u) atomicOr(q + 1, shifted_dword<1>(d, sel));
  if (e > 8u) atomicOr(q + 2, shifted_dword<2>(d, sel));
  if (e > 12u) atomicOr(q + 3, shifted_dword<3>(d, sel));
  if (e > 16u) atomicOr(q + 4, shifted_dword<4>(d, sel));
}

// Where and how one group's bytes go into the 1234 image (see encode_wavetile).
struct GroupSlot {
  u32* q;   // aligned dword holding the group's first byte
  u32 sel;  // shift_selector(p)
  u32 e;    // bytes from the start of q[0] to the group's end; dword I is written iff e > 4 I
};
template <bool FULL>
__device__ __forceinline__ GroupSlot group_slot(u8* image, u32 p, u32 g) {
  GroupSlot s;
  s.q = reinterpret_cast<u32*>(image + (p & ~3u));
  s.sel = shift_selector(p);
  const u32 L = gi_total(g);
  s.e = (FULL || L) ? L + (p & 3u) : 0u;
  return s;
}
template <int I, bool FULL>
__device__ __forceinline__ void store_dword(const GroupSlot& s, const uint4& d) {
  const u32 v = shifted_dword<I>(d, s.sel);
  if ((FULL && I == 0) || s.e > 4u * I) s.q[I] = v;
}
template <bool FULL>
__device__ __forceinline__ void store_head(u8* image, u32 first_dword, u32 p, u32 g) {
  if (FULL || gi_total(g)) {
    image[p] = (u8)first_dword;
    image[p + 1] = (u8)(first_dword >> 8);
    image[p + 2] = (u8)(first_dword >> 16);
  }
}

// One wave tile, values -> bytes.  x[j] holds group j*64+lane (already delta /
// zigzag transformed, absent values zero) and g[j] its descriptor; `nints` of
// the tile's 1024 values exist (FULL: all of them).  Assembles the data bytes
// in the wave's LDS image, copies them to dst[0..total) and the keys to kdst.
// Returns total.
template <int V, bool FULL, bool NTS = false>
__device__ __forceinline__ u32 encode_wavetile(const uint4 (&x)[4], const u32 (&g)[4], u32 nints,
                                               int lane, EncWaveLds& S, u8* dst, u8* kdst) {
  // A full wave tile whose 256 keys are all zero (every value below 256 for
  // 1234, every value zero for 0124; the mirror image of the decoder's
  // shortcut): a group's data bytes are the low bytes of its four values, one
  // dword at byte 4 * (j*64+lane).  No scan, no LDS.
  if (FULL && __ballot((gi_key(g[0]) | gi_key(g[1]) | gi_key(g[2]) | gi_key(g[3])) != 0u) == 0ull) {
    if (V == V1234) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        st_u32(dst + 4u * (j * 64 + lane), x[j].x | (x[j].y << 8) | (x[j].z << 16) | (x[j].w << 24));
    }
    st_u32(kdst + 4 * lane, 0u);
    return V == V1234 ? (u32)kWtInts : 0u;
  }
  u64 packed = 0;  // 4 x 16-bit group lengths, one per j
#pragma unroll
  for (int j = 0; j < 4; ++j) packed |= (u64)gi_total(g[j]) << (16 * j);
  // the four 16-bit fields never carry into each other (a field sums to at most
  // 64*16), so two 32-bit scans give, per j, the inclusive prefix over lanes
  const u32 inc_lo = wave_scan_incl((u32)packed, lane), inc_hi = wave_scan_incl((u32)(packed >> 32), lane);
  const u64 incl = ((u64)inc_hi << 32) | inc_lo;
  const u64 tots = ((u64)wave_last(inc_hi) << 32) | wave_last(inc_lo);
  const u64 excl = incl - packed;
  const u32 a = (u32)((uintptr_t)dst & 15u);  // LDS image byte i+a <-> dst[i]
  const u32 t0 = (u32)(tots & 0xFFFFu), t1 = (u32)((tots >> 16) & 0xFFFFu),
            t2 = (u32)((tots >> 32) & 0xFFFFu), t3 = (u32)(tots >> 48);
  const u32 total = t0 + t1 + t2 + t3;
  const u32 p0 = a + (u32)(excl & 0xFFFFu);
  const u32 p1 = a + t0 + (u32)((excl >> 16) & 0xFFFFu);
  const u32 p2 = a + t0 + t1 + (u32)((excl >> 32) & 0xFFFFu);
  const u32 p3 = a + t0 + t1 + t2 + (u32)(excl >> 48);
#pragma unroll
  for (int j = 0; j < 4; ++j) S.keys[j * 64 + lane] = (u8)gi_key(g[j]);
  const u32 pj[4] = {p0, p1, p2, p3};

  if (V == V1234) {
    // Plain aligned dword stores, no atomics.  A dword on the border of two
    // groups is written by both (1234 groups have >= 4 bytes, so never by
    // three): by the later group as its dword 0, by the earlier one as its last
    // dword, index >= 1.  The quarters are stored last first, and inside a
    // quarter dword index by dword index, so the earlier group's store always
    // lands after the later group's dword 0 (LDS executes one wave's stores in
    // program order): the border dword ends up with the earlier group's tail
    // bytes and zeros where the later group's first 1..3 bytes belong.  Those
    // are written again at the end, byte by byte, by every lane for its own
    // groups (the first three bytes of a 1234 group always exist).  Groups
    // without bytes (past the end of the stream) store nothing.  One quarter at
    // a time keeps only one packed group per lane live.
    u32 head[4];
#pragma unroll
    for (int j = 3; j >= 0; --j) {
      const uint4 d = pack_group<V>(x[j], g[j]);
      const GroupSlot s = group_slot<FULL>(S.data, pj[j], g[j]);
      store_dword<0, FULL>(s, d);
      store_dword<1, FULL>(s, d);
      store_dword<2, FULL>(s, d);
      store_dword<3, FULL>(s, d);
      store_dword<4, FULL>(s, d);
      head[j] = d.x;
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) store_head<FULL>(S.data, head[j], pj[j], g[j]);
  } else {
    // zero the image: [0, a+total) rounded up to whole 16-byte chunks
    const u32 nz = (a + total + 15u) >> 4;  // <= 258
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

  // data: aligned interior chunks, ragged head / tail by bytes
  {
    const u32 end = a + total;                 // image bytes [a, end)
    const u32 c_lo = (a + 15u) >> 4;           // first chunk lying fully inside
    const u32 c_hi = end >> 4;                 // one past the last such chunk
    u8* g0 = dst - a;                          // 16-byte aligned
    uint4* g16 = reinterpret_cast<uint4*>(g0);
    const uint4* l16 = reinterpret_cast<const uint4*>(S.data);
    for (u32 c = c_lo + lane; c < c_hi; c += 64) st_stream16<NTS>(g16 + c, l16[c]);
    const u32 hend = min(16u * c_lo, end);     // head bytes [a, hend)
    const u32 tstart = max(16u * c_hi, hend);  // tail bytes [tstart, end)
    if (lane < 16) {
      const u32 i = a + lane;
      if (i < hend) g0[i] = S.data[i];
    } else if (lane < 32) {
      const u32 i = tstart + (lane - 16);
      if (i < end) g0[i] = S.data[i];
    }
  }
  // keys
  {
    const u32 nkeys = FULL ? (u32)kWtKeys : ((nints + 3u) >> 2);
    if ((((uintptr_t)kdst) & 3u) == 0 && 4u * lane + 4u <= nkeys) {
      *reinterpret_cast<u32*>(kdst + 4 * lane) = reinterpret_cast<const u32*>(S.keys)[lane];
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (4u * lane + b < nkeys) kdst[4 * lane + b] = S.keys[4 * lane + b];
    }
  }
  __builtin_amdgcn_wave_barrier();
  return total;
}

// Absent values (past the end of the stream) must be zero before packing.
__device__ __forceinline__ void zero_absent(uint4& c, u64 first, u64 count) {
  if (first + 4 > count) {
    if (first + 3 >= count) c.w = 0;
    if (first + 2 >= count) c.z = 0;
    if (first + 1 >= count) c.y = 0;
    if (first >= count) c.x = 0;
  }
}

#ifndef SVB_ENCODE_LB_KT
#define SVB_ENCODE_LB_KT 2  // tile words polled per lane in k_encode's look-back (A/B at 2^28 values, a workgroup per tile: 1 / 2 / 4 = 0.664 / 0.667 / 0.605 of peak on mixed widths, 0.636 / 0.655 / 0.660 on dense postings)
#endif
// k_encode<.., FULL=true> handles the tiles that lie wholly inside the stream
// (one workgroup each, no bounds logic); the one ragged tile at the end, if any,
// is a second single-workgroup launch of the FULL=false instantiation on the
// same stream, continuing the same ticket and look-back words.
// QW = wave tiles per wave; the workgroup's tile is 4*QW wave tiles.
// QUANT: the input holds float32; every value becomes (int32)rintf(x * scale) first
// (scale travels as the bits of `prev`, which only DELTA uses otherwise) -- the
// quantise step of gradient compression fused into the encoder's load.
// dbg bit 3 (set by the launch code, not a diagnostic): the launch has a workgroup per full tile, each draws
// one ticket and ends after its tile (the dispatcher refills the slot); else persistent workgroups loop
// over tickets.  A run-time flag, so that both forms are the same code.
template <int V, bool DELTA, bool ZZ, int QW, bool FULL, bool QUANT = false, int NW = kWaves, int NT = 0>
__global__ __launch_bounds__(NW * 64, !FULL ? 1 : (NW == 8 || QW > 2 || V != V1234 ? 4 : 5)) void k_encode(
    const u32* __restrict__ in, u32 count, u32 n_tiles, u8* __restrict__ out, u32 prev,
    WsHeader* __restrict__ hdr, LbWords lb, u64* __restrict__ written, u32 dbg,
    u64* __restrict__ stamps) {
  __shared__ EncWaveLds s_lds[NW];
  __shared__ u32 s_wave_tot[NW];
  __shared__ u64 s_prefix;
  __shared__ u32 s_tile;

  const int lane = lane_id(), wave = wave_id();
  // FULL: a workgroup draws tiles from the ticket -- one (dbg bit 3) or until the n_tiles full tiles
  // are taken.  !FULL: the one ragged tile, index n_tiles, one workgroup.
  for (;;) {
  u32 tile = n_tiles;
  if (FULL) {
    if (threadIdx.x == 0) s_tile = atomicAdd(&hdr->ticket, 1u);
    __syncthreads();
    tile = __builtin_amdgcn_readfirstlane(s_tile);
    if (tile >= n_tiles) break;
  }
  // diagnostic build only (stamps != nullptr): wall-clock stamps per tile, 100 MHz
  if (stamps && threadIdx.x == 0) stamps[5 * (u64)tile + 0] = __builtin_amdgcn_s_memrealtime();
  const u64 keylen = ((u64)count + 3) >> 2;
  constexpr u64 kTileInts = (u64)NW * QW * kWtInts;
  const u64 wave_first = (u64)tile * kTileInts + (u64)wave * QW * kWtInts;  // first value of this wave

  // ---- pass 1: load, transform, classify -----------------------------------
  uint4 x[QW][4];
  u32 g[QW][4];  // !FULL: one descriptor per group
  u32 kq[QW];    // FULL: the four key bytes of a wave tile in one register (lengths follow from them)
  if (FULL) {
    const uint4* src = reinterpret_cast<const uint4*>(in + wave_first);  // wave-uniform base + lane offset
#pragma unroll
    for (int q = 0; q < QW; ++q) {
#pragma unroll
      for (int j = 0; j < 4; ++j) x[q][j] = ld_stream16<(NT & kNtLoads) != 0>(src + ((u32)lane + (u32)(q * 256 + j * 64)));
    }
  } else {
#pragma unroll
    for (int q = 0; q < QW; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        x[q][j] = load_group(in, wave_first + (u64)q * kWtInts + 4u * (j * 64 + lane), count);
  }
  if (QUANT) {
    const float scale = __uint_as_float(prev);
    auto q1 = [scale](u32 bits) -> u32 { return (u32)(int)rintf(__fmul_rn(__uint_as_float(bits), scale)); };
#pragma unroll
    for (int q = 0; q < QW; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint4& c = x[q][j];
        c = make_uint4(q1(c.x), q1(c.y), q1(c.z), q1(c.w));
      }
  }
  if (DELTA) {
    // previous value of the wave's first element, then neighbours by lane shuffles
    u32 carry = prev;
    if (wave_first > 0 && wave_first <= count) carry = in[wave_first - 1];
#pragma unroll
    for (int q = 0; q < QW; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint4 c = x[q][j];
        const u32 pv = wave_shift_up1(c.w, carry);
        carry = wave_last(c.w);
        x[q][j] = make_uint4(c.x - pv, c.y - c.x, c.z - c.y, c.w - c.z);
      }
  }
  u32 lane_tot = 0;
#pragma unroll
  for (int q = 0; q < QW; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint4& c = x[q][j];
      if (ZZ) c = make_uint4(zz_enc(c.x), zz_enc(c.y), zz_enc(c.z), zz_enc(c.w));
      u32 nv = 4u;
      if (!FULL) {
        const u64 first = wave_first + (u64)q * kWtInts + 4u * (j * 64 + lane);
        if (DELTA || ZZ) zero_absent(c, first, count);
        nv = first < count ? (u32)min((u64)4, (u64)count - first) : 0u;
      }
      if (!FULL) {
        g[q][j] = classify_group<V, false>(c, nv);
        lane_tot += gi_total(g[q][j]);
      }
    }
  if (FULL) {
#pragma unroll
    for (int q = 0; q < QW; ++q) {
      u32 k4 = classify_key<V>(x[q][0]) | (classify_key<V>(x[q][1]) << 8) | (classify_key<V>(x[q][2]) << 16) |
               (classify_key<V>(x[q][3]) << 24);
      asm volatile("" : "+v"(k4));  // keep the four keys in ONE register from here on
      kq[q] = k4;
      lane_tot += word_len<V>(k4, 16u);
    }
  }
  const u32 wave_tot = wave_sum(lane_tot);
  if (lane == 0) s_wave_tot[wave] = wave_tot;
  __syncthreads();
  if (stamps && threadIdx.x == 0) stamps[5 * (u64)tile + 1] = __builtin_amdgcn_s_memrealtime();

  // ---- tile offset: sum of the byte counts of all earlier tiles ---------------
  if (wave == 0) {
    u32 tile_tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) tile_tot += s_wave_tot[w];
    u64 excl = 0;
    u32 rounds = 0;
    if (!(dbg & 1u)) {
      excl = lb2_exclusive_prefix<SVB_ENCODE_LB_KT>(lb, tile, tile_tot, lane, &hdr->error, &rounds);
    } else {
      excl = (u64)tile * (kTileInts * 5 / 2);  // timing experiment: no look-back, plausible (wrong) offsets
      if (dbg & 4u) {                          // ... after idling about as long as a look-back takes
        const u64 t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < 1100) __builtin_amdgcn_s_sleep(16);
      }
    }
    if (stamps && lane == 0) stamps[5 * (u64)tile + 4] = rounds;
    if (stamps && lane == 0) stamps[5 * (u64)tile + 2] = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      s_prefix = excl;
      const u64 last_tile = keylen ? (keylen - 1) / ((u64)NW * QW * kWtKeys) : 0;
      if (tile == last_tile && written) *written = keylen + excl + tile_tot;
      if (tile == 0u || !FULL) reset_other_ticket(hdr);  // once per call: the full tiles' kernel, and the ragged tile's
    }
  }
  __syncthreads();
  u64 doff = s_prefix;
  for (int w = 0; w < wave; ++w) doff += s_wave_tot[w];

  // ---- pass 2: per wave tile, scan -> assemble in LDS -> store -----------------
  EncWaveLds& S = s_lds[wave];
  u8* data_out = out + keylen;
  if (!(dbg & 2u)) {
  // A rolled loop: the wave tile being emitted is always x[0] / g[0], the rest
  // rotate down by register moves.  This keeps one iteration's temporaries from
  // overlapping the next one's (an unrolled loop costs ~40 more VGPRs and a
  // workgroup per CU).
#pragma unroll 1
  for (int q = 0; q < QW; ++q) {
    const u64 wt_first = wave_first + (u64)q * kWtInts;
    if (FULL) {
      const u32 lens4 = key_lens4<V>(kq[0], 16u);
      u32 gq[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) gq[j] = ((kq[0] >> (8 * j)) & 0xFFu) | (((lens4 >> (8 * j)) & 0xFFu) << 24);
      doff += encode_wavetile<V, true, (NT & kNtStores) != 0>(x[0], gq, kWtInts, lane, S, data_out + doff, out + (wt_first >> 2));
    } else if (wt_first < count) {
      const u32 nints = (u32)min((u64)kWtInts, (u64)count - wt_first);
      doff += encode_wavetile<V, false>(x[0], g[0], nints, lane, S, data_out + doff, out + (wt_first >> 2));
    }
#pragma unroll
    for (int r = 0; r + 1 < QW; ++r) {
      kq[r] = kq[r + 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[r][j] = x[r + 1][j];
        g[r][j] = g[r + 1][j];
      }
    }
  }
  }
  if (stamps && threadIdx.x == 0) stamps[5 * (u64)tile + 3] = __builtin_amdgcn_s_memrealtime();
  if (!FULL || (dbg & 8u)) break;  // dbg bit 3: this launch has a workgroup per tile
  }
}

// ---------------------------------------------------------------------------
// k_encode_pipe: the encoder with the look-back taken off the critical path.
//
// k_encode runs a tile as load -> publish -> WAIT for the prefix -> store, and
// the wait (a few memory round trips under load plus the slowest predecessor)
// is over 40 % of a tile's life while its values sit in registers.  Here a
// workgroup keeps TWO tiles of half the size in flight and loads, classifies
// and publishes the NEXT one before it stores the current one:
//
//     T(b) | L(b) + R(a) | barrier | P(b) S(a) | T(c) | L(c) + R(b) | barrier | P(c) S(b) | ...
//
// (T ticket, L load + classify, P publish the tile's byte count, R resolve the
// prefix, S assemble + store).  A tile is resolved a whole store + load phase
// after it was published, when its predecessors' words have usually arrived:
// wave 0 sends the polls right behind the loads of the next tile, so they
// travel while that tile is classified, and one round answers.  Two barriers
// per tile.  Between T(x) and P(x) the workgroup waits only for tiles with
// smaller tickets (R of its previous tile), so by induction over the ticket
// every ticketed tile is published: the look-back chain cannot deadlock.
// Same registers as k_encode with tiles of twice the size (two tiles of QW wave
// tiles per wave).  Handles only tiles that lie wholly inside the stream; the
// ragged last tile is a launch of k_encode<.., QW, false, .., NW> on the same
// look-back words.
// STAMPS: diagnostic build of the kernel that writes wall-clock stamps per tile (the product instantiation
// carries none of it: the kernel sits at its register limits and every scalar counts).
template <int V, bool DELTA, bool ZZ, int QW, bool QUANT = false, int NW = 8, int NT = 0, bool STAMPS = false>
__global__ __launch_bounds__(NW * 64, 4) void k_encode_pipe(
    const u32* __restrict__ in, u32 count, u32 n_tiles, u8* __restrict__ out, u32 prev,
    WsHeader* __restrict__ hdr, LbWords lb, u64* __restrict__ written, u64* __restrict__ stamps_arg, u32 early) {
  u64* const stamps = STAMPS ? stamps_arg : nullptr;
  __shared__ EncWaveLds s_lds[NW];
  __shared__ u32 s_wave_tot[2][NW];
  __shared__ u64 s_prefix;
  __shared__ u32 s_tile;
  // kT = 4: the tile words of the own group and of the 15 groups before it are read directly; group
  // words are needed only for groups more than 240 tiles back, whose leaders have resolved long ago.
  constexpr int kT = 4;

  const int lane = lane_id(), wave = wave_id();
  const u64 keylen = ((u64)count + 3) >> 2;
  constexpr u64 kTileInts = (u64)NW * QW * kWtInts;
  const u64 last_tile = keylen ? (keylen - 1) / ((u64)NW * QW * kWtKeys) : 0;
  u8* data_out = out + keylen;
  EncWaveLds& S = s_lds[wave];

  // T: the barrier also ends the store phase before it (s_prefix, s_wave_tot and s_tile are free again)
  auto draw = [&]() -> u32 {
    if (threadIdx.x == 0) s_tile = atomicAdd(&hdr->ticket, 1u);
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(s_tile);
  };
  // wave 0: prefix of `tile` (whose byte count is tile_tot) into s_prefix; `first` = polls already sent
  auto resolve = [&](u32 tile, u32 tile_tot, const LbState<kT>* first) {
    u32 rounds = 0;
    const u64 excl = lb2_resolve<kT>(lb, tile, tile_tot, lane, &hdr->error, &rounds, first);
    if (lane == 0) {
      s_prefix = excl;
      if (tile == 0u) reset_other_ticket(hdr);
      if (tile == last_tile && written) *written = keylen + excl + tile_tot;
      if (stamps) {  // diagnostic build only: wall-clock stamps per tile, 100 MHz
        stamps[5 * (u64)tile + 2] = __builtin_amdgcn_s_memrealtime();
        stamps[5 * (u64)tile + 4] = rounds | ((u64)blockIdx.x << 32);
      }
    }
  };
  // L (+ R of the tile before): values of `tile` into x, their keys into kq, the waves' byte counts
  // into s_wave_tot[buf]; wave 0 resolves tile_res meanwhile (0xFFFFFFFF: none).
  auto load = [&](u32 tile, int buf, uint4 (&x)[QW][4], u32 (&kq)[QW], u32 tile_res, u32 tot_res) {
    if (stamps && threadIdx.x == 0) stamps[5 * (u64)tile + 0] = __builtin_amdgcn_s_memrealtime();
    const u64 wave_first = (u64)tile * kTileInts + (u64)wave * QW * kWtInts;
    const uint4* src = reinterpret_cast<const uint4*>(in + wave_first);
#pragma unroll
    for (int q = 0; q < QW; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) x[q][j] = ld_stream16<(NT & kNtLoads) != 0>(src + ((u32)lane + (u32)(q * 256 + j * 64)));
    LbState<kT> polls;
    const bool resolving = wave == 0 && tile_res != 0xFFFFFFFFu;
    if (resolving) polls = lb2_begin<kT>(lb, tile_res, lane);
    if (QUANT) {
      const float scale = __uint_as_float(prev);
      auto q1 = [scale](u32 bits) -> u32 { return (u32)(int)rintf(__fmul_rn(__uint_as_float(bits), scale)); };
#pragma unroll
      for (int q = 0; q < QW; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint4& c = x[q][j];
          c = make_uint4(q1(c.x), q1(c.y), q1(c.z), q1(c.w));
        }
    }
    if (DELTA) {
      u32 carry = prev;
      if (wave_first > 0) carry = in[wave_first - 1];
#pragma unroll
      for (int q = 0; q < QW; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint4 c = x[q][j];
          const u32 pv = wave_shift_up1(c.w, carry);
          carry = wave_last(c.w);
          x[q][j] = make_uint4(c.x - pv, c.y - c.x, c.z - c.y, c.w - c.z);
        }
    }
    u32 lane_tot = 0;
#pragma unroll
    for (int q = 0; q < QW; ++q) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint4& c = x[q][j];
        if (ZZ) c = make_uint4(zz_enc(c.x), zz_enc(c.y), zz_enc(c.z), zz_enc(c.w));
      }
      u32 k4 = classify_key<V>(x[q][0]) | (classify_key<V>(x[q][1]) << 8) | (classify_key<V>(x[q][2]) << 16) |
               (classify_key<V>(x[q][3]) << 24);
      asm volatile("" : "+v"(k4));  // the four keys stay in ONE register
      kq[q] = k4;
      lane_tot += word_len<V>(k4, 16u);
    }
    const u32 wave_tot = wave_sum(lane_tot);
    if (lane == 0) s_wave_tot[buf][wave] = wave_tot;
    if (resolving) resolve(tile_res, tot_res, &polls);
  };
  // P (wave 0): the tile's byte count becomes visible to later tiles.  Returns it (valid in wave 0).
  auto publish = [&](u32 tile, int buf) -> u32 {
    u32 tile_tot = 0;
    if (wave == 0) {
#pragma unroll
      for (int w = 0; w < NW; ++w) tile_tot += s_wave_tot[buf][w];
      lb2_publish(lb, tile, tile_tot, lane, &hdr->error);
      if (stamps && lane == 0) stamps[5 * (u64)tile + 1] = __builtin_amdgcn_s_memrealtime();
    }
    return tile_tot;
  };
  // S: assemble and store the wave tiles of `tile`, whose prefix is in s_prefix.
  auto store = [&](u32 tile, int buf, uint4 (&x)[QW][4], u32 (&kq)[QW]) {
    u64 doff = s_prefix;
    for (int w = 0; w < wave; ++w) doff += s_wave_tot[buf][w];
    const u64 wave_first = (u64)tile * kTileInts + (u64)wave * QW * kWtInts;
#pragma unroll 1
    for (int q = 0; q < QW; ++q) {
      const u64 wt_first = wave_first + (u64)q * kWtInts;
      const u32 lens4 = key_lens4<V>(kq[0], 16u);
      u32 gq[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) gq[j] = ((kq[0] >> (8 * j)) & 0xFFu) | (((lens4 >> (8 * j)) & 0xFFu) << 24);
      doff += encode_wavetile<V, true, (NT & kNtStores) != 0>(x[0], gq, kWtInts, lane, S, data_out + doff, out + (wt_first >> 2));
#pragma unroll
      for (int r = 0; r + 1 < QW; ++r) {
        kq[r] = kq[r + 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[r][j] = x[r + 1][j];
      }
    }
    if (stamps && threadIdx.x == 0) stamps[5 * (u64)tile + 3] = __builtin_amdgcn_s_memrealtime();
  };

  // early: the ticket of the tile after next is requested right after a publish and collected after the
  // store phase that follows, so its round trip (about a microsecond under load) is off the tile loop.
  // The workgroup then holds a ticket through one store phase before it loads that tile; between the
  // draw and the tile's publish it still waits only for smaller tickets.
  auto request = [&]() -> u32 { return threadIdx.x == 0 ? atomicAdd(&hdr->ticket, 1u) : 0u; };
  auto collect = [&](u32 requested) -> u32 {
    if (threadIdx.x == 0) s_tile = requested;
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(s_tile);
  };

  uint4 xa[QW][4], xb[QW][4];
  u32 kqa[QW], kqb[QW];
  u32 tile_a = draw();
  if (tile_a >= n_tiles) return;
  load(tile_a, 0, xa, kqa, 0xFFFFFFFFu, 0u);
  __syncthreads();
  u32 tot_a = publish(tile_a, 0);
  u32 requested = early ? request() : 0u;
  int buf = 0;
  for (;;) {
    const u32 tile_b = early ? collect(requested) : draw();
    const bool more = tile_b < n_tiles;  // workgroup-uniform
    if (more) load(tile_b, buf ^ 1, xb, kqb, tile_a, tot_a);
    else if (wave == 0) resolve(tile_a, tot_a, nullptr);
    __syncthreads();
    u32 tot_b = 0;
    if (more) tot_b = publish(tile_b, buf ^ 1);
    if (more && early) requested = request();
    store(tile_a, buf, xa, kqa);
    if (!more) break;
#pragma unroll
    for (int q = 0; q < QW; ++q) {
      kqa[q] = kqb[q];
#pragma unroll
      for (int j = 0; j < 4; ++j) xa[q][j] = xb[q][j];
    }
    tile_a = tile_b;
    tot_a = tot_b;
    buf ^= 1;
  }
}

}  // namespace svb

// Device-side building blocks shared by every StreamVByte kernel (gfx950 only).
//
// Vocabulary
//   key        one control byte = the 2-bit codes of 4 consecutive values
//   group      the 4 values of one key and their 0..16 data bytes
//   wave tile  256 keys = 1024 values, the unit one 64-lane wavefront handles:
//              lane l owns keys 4l..4l+3 while lengths are scanned ("blocked")
//              and groups l, l+64, l+128, l+192 while bytes are moved
//              ("striped"), so that every global access of the uint32 side is
//              one contiguous 1 KiB per wave instruction.
//   super tile 64 wave tiles, the unit of the two-level offset directory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svb {

constexpr int V1234 = 0;
constexpr int V0124 = 1;
constexpr int V1248 = 2;  // 64-bit values: code c -> 1 << c data bytes (only the key-block kernels know it)

constexpr int kBlock = 256;          // threads per workgroup (4 wavefronts)
constexpr int kWaves = 4;
constexpr int kWtKeys = 256;         // keys per wave tile
constexpr int kWtInts = 1024;        // values per wave tile
constexpr int kSuperWt = 64;         // wave tiles per super tile
constexpr int kWtDataMax = 4096;     // data bytes of a wave tile, worst case

using u8 = uint8_t;
using u32 = uint32_t;
using u64 = uint64_t;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() {
  return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
}

// gfx950 executes dword global and LDS accesses at any byte address.
__device__ __forceinline__ u32 ld_u32(const u8* p) {
  u32 v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
__device__ __forceinline__ void st_u32(u8* p, u32 v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ uint4 ld_u128(const u8* p) {
  uint4 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}
// 16-byte global accesses of data that a call reads or writes exactly once (the values and the
// streams).  NT marks them non-temporal: the line is not kept in L2 / the Infinity Cache, which a long
// stream would only sweep through, evicting (and writing back) what it displaces.  Kernels take the
// choice as a template argument (bit 0: loads, bit 1: stores); the launch code picks it from the size
// of the call -- short streams keep the default policy, their bytes are likely to be used again from
// the cache by whatever runs next.
constexpr int kNtLoads = 1, kNtStores = 2;
// The pointers these take are value and stream addresses of the caller's: uint32 arrays are only
// 4-byte aligned (batched calls start a stream's values anywhere in an array), so the vector type says
// 4 and not 16 -- gfx950 executes a dwordx4 access at any dword address, but a type that promises
// more than the pointer keeps would be undefined behaviour the compiler may build on.
typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
typedef u32x4_t u32x4_a4 __attribute__((aligned(4)));
template <bool NT>
__device__ __forceinline__ uint4 ld_stream16(const void* p) {
  const u32x4_a4* q = reinterpret_cast<const u32x4_a4*>(p);
  const u32x4_t v = NT ? __builtin_nontemporal_load(q) : *q;
  return make_uint4(v.x, v.y, v.z, v.w);
}
template <bool NT>
__device__ __forceinline__ void st_stream16(void* p, const uint4& v) {
  u32x4_a4* q = reinterpret_cast<u32x4_a4*>(p);
  if (NT) __builtin_nontemporal_store(u32x4_t{v.x, v.y, v.z, v.w}, q);
  else *q = u32x4_t{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ u32 lds_u32(const u8* p) {
  u32 v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
// The 4 bytes at byte offset `p` of an LDS image, read as the aligned dword
// pair that holds them and funnel-shifted (a misaligned ds_read_b32 is legal on
// gfx950 but is replayed lane by lane).
__device__ __forceinline__ u32 lds_u32_aligned(const u8* base, u32 p) {
  const u32* q = reinterpret_cast<const u32*>(base + (p & ~3u));
  const u32 lo = q[0], hi = q[1];
  return __builtin_amdgcn_alignbyte(hi, lo, p);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for
// every outstanding global load of the wave (its fence drains vmcnt), which ends
// any software pipelining of loads across the barrier; here loads requested
// before the barrier stay in flight.  Use only where the data handed over at the
// barrier lives in LDS.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- cross-lane primitives on DPP (no LDS traffic) ------------------------------
// row_shr:n moves data n lanes up inside each row of 16 lanes (zero shifted in),
// row_bcast:15 / :31 carry the last lane of a row / of the lower half-wave into
// the following row(s): six DPP adds make a 64-lane inclusive scan.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u32 dpp_zero(u32 x) {
  return (u32)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, true);
}
// Inclusive prefix sum over the 64 lanes of a wavefront.
__device__ __forceinline__ u32 wave_scan_incl(u32 x, int /*lane*/) {
  x += dpp_zero<0x111, 0xF>(x);  // row_shr:1
  x += dpp_zero<0x112, 0xF>(x);  // row_shr:2
  x += dpp_zero<0x114, 0xF>(x);  // row_shr:4
  x += dpp_zero<0x118, 0xF>(x);  // row_shr:8
  x += dpp_zero<0x142, 0xA>(x);  // row_bcast:15 into rows 1 and 3
  x += dpp_zero<0x143, 0xC>(x);  // row_bcast:31 into rows 2 and 3
  return x;
}
// Inclusive prefix sum inside each row of 16 lanes.
__device__ __forceinline__ u32 row_scan_incl(u32 x) {
  x += dpp_zero<0x111, 0xF>(x);
  x += dpp_zero<0x112, 0xF>(x);
  x += dpp_zero<0x114, 0xF>(x);
  x += dpp_zero<0x118, 0xF>(x);
  return x;
}
// Sum over each row of 16 lanes, in every lane of the row (a butterfly: quad_perm, row_half_mirror,
// row_mirror reach the partner lanes).
__device__ __forceinline__ u32 row_sum_all(u32 x) {
  x += (u32)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  x += (u32)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  x += (u32)__builtin_amdgcn_mov_dpp((int)x, 0x141, 0xF, 0xF, false);  // row_half_mirror: lane i <- 7 - i of its half row
  x += (u32)__builtin_amdgcn_mov_dpp((int)x, 0x140, 0xF, 0xF, false);  // row_mirror: lane i <- 15 - i
  return x;
}
// Value of lane 63 (wave-uniform).
__device__ __forceinline__ u32 wave_last(u32 x) { return (u32)__builtin_amdgcn_readlane((int)x, 63); }
// Value of the previous lane; lane 0 receives `first`.
__device__ __forceinline__ u32 wave_shift_up1(u32 x, u32 first) {
  return (u32)__builtin_amdgcn_update_dpp((int)first, (int)x, 0x138, 0xF, 0xF, false);  // wave_shr:1
}
// Value of the next lane; lane 63 receives `last`.
__device__ __forceinline__ u32 wave_shift_down1(u32 x, u32 last) {
  return (u32)__builtin_amdgcn_update_dpp((int)last, (int)x, 0x130, 0xF, 0xF, false);  // wave_shl:1
}
// Segmented inclusive prefix sum over the lanes: lane l ends up with the sum of
// val over the lanes l' <= l that have the same key.  Keys must not decrease
// with the lane.  The six DPP steps of wave_scan_incl; a step contributes only
// where the key it brings along equals the lane's own (keys are sorted, so then
// every lane in between has that key too).  Lanes without a source see val 0.
__device__ __forceinline__ u32 seg_add_scan(u32 key, u32 val) {
#define SVB_SEG_STEP(CTRL, MASK)                                                \
  {                                                                             \
    const u32 k2 = dpp_zero<CTRL, MASK>(key), v2 = dpp_zero<CTRL, MASK>(val);   \
    val += k2 == key ? v2 : 0u;                                                 \
  }
  SVB_SEG_STEP(0x111, 0xF)
  SVB_SEG_STEP(0x112, 0xF)
  SVB_SEG_STEP(0x114, 0xF)
  SVB_SEG_STEP(0x118, 0xF)
  SVB_SEG_STEP(0x142, 0xA)
  SVB_SEG_STEP(0x143, 0xC)
#undef SVB_SEG_STEP
  return val;
}
// OR of x over the wavefront (wave-uniform result).
__device__ __forceinline__ u32 wave_or(u32 x) {
  x |= dpp_zero<0x111, 0xF>(x);
  x |= dpp_zero<0x112, 0xF>(x);
  x |= dpp_zero<0x114, 0xF>(x);
  x |= dpp_zero<0x118, 0xF>(x);
  x |= dpp_zero<0x142, 0xA>(x);
  x |= dpp_zero<0x143, 0xC>(x);
  return wave_last(x);
}
__device__ __forceinline__ u32 wave_sum(u32 x) { return wave_last(wave_scan_incl(x, 0)); }
// 64-bit sum over the wavefront as three 32-bit DPP reductions of 26-bit pieces
// (a chain of twelve ds_bpermute shuffles costs several hundred cycles).
__device__ __forceinline__ u64 wave_sum64(u64 x) {
  const u32 a = (u32)x & 0x3FFFFFFu, b = (u32)(x >> 26) & 0x3FFFFFFu, c = (u32)(x >> 52);
  return (u64)wave_sum(a) + ((u64)wave_sum(b) << 26) + ((u64)wave_sum(c) << 52);
}

// ---- width arithmetic -------------------------------------------------------
// Width code of a value (reference src/streamvbyte_encode.c:14-37 and
// src/streamvbyte_0124_encode.c:12-32).
template <int V>
__device__ __forceinline__ u32 code_of(u32 v) {
  if (V == V1234) return (u32)(v > 0xFFu) + (u32)(v > 0xFFFFu) + (u32)(v > 0xFFFFFFu);
  return (u32)(v != 0u) + (u32)(v > 0xFFu) + (u32)(v > 0xFFFFu);
}
// Data bytes of a code (reference src/streamvbyte_decode.c:22-36,
// src/streamvbyte_0124_decode.c:210).
template <int V>
__device__ __forceinline__ u32 len_of(u32 c) {
  return V == V1234 ? c + 1u : ((1u << c) >> 1);
}
// Mask keeping the low len_of(c) bytes.
template <int V>
__device__ __forceinline__ u32 mask_of(u32 c) {
  if (V == V1234) return 0xFFFFFFFFu >> (24u - 8u * c);
  u32 m = 0xFFFFFFFFu >> ((32u - (8u << c >> 1)) & 31u);  // c=1:>>24  c=2:>>16  c=3:>>0
  return c ? m : 0u;
}

// Per-byte sums of the four 2-bit fields of each byte of w (SWAR).
__device__ __forceinline__ u32 code_sums4(u32 w) {
  u32 s = (w & 0x33333333u) + ((w >> 2) & 0x33333333u);
  return (s + (s >> 4)) & 0x0F0F0F0Fu;
}
// Per-byte count of 2-bit fields equal to 3.
__device__ __forceinline__ u32 code_threes4(u32 w) {
  u32 t = w & (w >> 1) & 0x55555555u;
  t = (t + (t >> 2)) & 0x33333333u;
  return (t + (t >> 4)) & 0x0F0F0F0Fu;
}
// nv = number of valid values among the 16 coded by w (0..16).
// Returns w with the codes of absent values cleared.
__device__ __forceinline__ u32 mask_codes(u32 w, u32 nv) {
  u32 m = nv >= 16u ? 0xFFFFFFFFu : ((1u << (2u * nv)) - 1u);
  return w & m;
}
// Packed per-key count of valid values (each byte 0..4) for a lane whose 16
// values have nv valid ones.
__device__ __forceinline__ u32 valid_per_key4(u32 nv) {
  if (nv >= 16u) return 0x04040404u;
  u32 a = min(nv, 4u);
  u32 b = min(nv - min(nv, 4u), 4u);
  u32 c = min(nv - min(nv, 8u), 4u);
  u32 d = min(nv - min(nv, 12u), 4u);
  return a | (b << 8) | (c << 16) | (d << 24);
}
// Packed data-byte lengths (one byte per key, each 0..16) of the 4 keys in w.
// This is the length table of the reference
// (src/streamvbyte_shuffle_tables_decode.h:5-22, _0124_decode.h:2-19,
// generator utils/shuffle_tables.c:10-38) evaluated arithmetically for four
// keys at once; absent values of a final partial key count for nothing.
template <int V>
__device__ __forceinline__ u32 key_lens4(u32 w_masked, u32 nv) {
  if (V == V1234) return code_sums4(w_masked) + valid_per_key4(nv);
  return code_sums4(w_masked) + code_threes4(w_masked);
}
// Total data bytes of the 16 values coded by w.
template <int V>
__device__ __forceinline__ u32 word_len(u32 w_masked, u32 nv) {
  u32 lo = __popc(w_masked & 0x55555555u), hi = __popc(w_masked & 0xAAAAAAAAu);
  if (V == V1248) {  // n0 + 2 n1 + 4 n2 + 8 n3 = nv + n1 + 3 n2 + 7 n3 with n3 = codes with both bits set
    const u32 n3 = __popc(w_masked & (w_masked >> 1) & 0x55555555u);
    return nv + (lo - n3) + 3u * (hi - n3) + 7u * n3;
  }
  if (V == V1234) return nv + lo + 2u * hi;
  return lo + 2u * hi + __popc(w_masked & (w_masked >> 1) & 0x55555555u);
}

// Data bytes coded by the 16 keys (64 values, all present) in the four dwords of q.
// With a = ones among the code bits, b = ones among the high bits of the codes and
// c = codes with both bits set:  1234: 64 + a + b;  0124: a + b + c;  1248: 64 + a + 2 b + 3 c
// (the same sums as word_len, regrouped so that v_bcnt's free accumulate does the adding).
template <int V>
__device__ __forceinline__ u32 dir_counts(const uint4& q) {
  const u32 w[4] = {q.x, q.y, q.z, q.w};
  u32 a = V == V0124 ? 0u : 64u, b = 0, c = 0;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    a += __popc(w[d]);
    b += __popc(w[d] & 0xAAAAAAAAu);
    if (V != V1234) c += __popc(w[d] & (w[d] >> 1) & 0x55555555u);
  }
  if (V == V1234) return a + b;
  if (V == V0124) return a + b + c;
  return a + 2u * b + 3u * c;
}

__device__ __forceinline__ u32 zz_enc(u32 v) { return (v << 1) ^ (0u - (v >> 31)); }
__device__ __forceinline__ u32 zz_dec(u32 v) { return (v >> 1) ^ (0u - (v & 1u)); }

// ---- workspace layout ---------------------------------------------------------
// One 256-byte header followed by the offset directory and the look-back words.
struct WsHeader {
  u32 error;        // set non-zero when a bounded spin gave up
  u32 ticket;       // dynamic tile counter of the look-back kernels
  u32 unused;       // (where `error` would be for a kernel handed the header 8 bytes up, see ticket1)
  u32 ticket1;      // a second tile counter: calls on an initialised workspace alternate between the two,
                    // each zeroing the one the next call will use (draw_ticket), so none needs a clearing
                    // launch; such a call gets the header pointer 8 bytes up when its turn is ticket1
  u64 accum;        // reduction scratch (validate)
  u32 pad[58];
};
static_assert(sizeof(WsHeader) == 256, "header is one 256-byte block");

struct WsLayout {
  u64 n_wt, n_super, n_tiles;
  size_t off_mega_sum, off_wt_rel, off_status, off_lb_group, off_wt_vsum, off_blk, off_bounds, total;
  u32 wt_per_mega, n_mega;
};
__host__ __device__ constexpr size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
constexpr int kLbWt = 16;  // wave tiles per look-back tile (16384 values)
constexpr int kMegas = 256;  // mega tiles of the offset directory (at most; one workgroup of k_key_sums each)

__host__ __device__ constexpr WsLayout ws_layout(u32 count) {
  WsLayout L{};
  u64 keylen = ((u64)count + 3) >> 2;
  L.n_wt = (keylen + kWtKeys - 1) / kWtKeys;
  L.n_super = (L.n_wt + kSuperWt - 1) / kSuperWt;
  L.n_tiles = (L.n_wt + kLbWt - 1) / kLbWt;
  size_t o = sizeof(WsHeader);
  // the look-back words follow the header so one memset clears both
  L.off_status = o;    o += align256(8 * (L.n_wt / kWaves + 2));  // room for the smallest tile size
  L.off_lb_group = o;  o += align256(8 * (L.n_wt / kWaves / 16 + 2));  // group words of the two-level look-back
  // offset directory: the key block is cut into at most kMegas equal runs of whole super tiles
  const u64 spm = (L.n_super + kMegas - 1) / kMegas;  // super tiles per mega tile (0 for an empty stream)
  L.wt_per_mega = (u32)(spm * kSuperWt);
  L.n_mega = spm ? (u32)((L.n_super + spm - 1) / spm) : 0u;
  L.off_mega_sum = o;  o += align256(4 * kMegas);
  L.off_wt_rel = o;    o += align256(4 * L.n_wt);
  L.off_wt_vsum = o;   o += align256(4 * L.n_wt);            // per-wave-tile value sums (delta decode)
  L.off_blk = o;       o += align256(4 * (L.n_wt / 1024 + 2));
  L.off_bounds = o;    o += align256(4 * (L.n_wt / 4 + 2));  // one place in the sorted array per 4 wave tiles (intersection)
  L.total = o;
  return L;
}

// ---- offset directory ------------------------------------------------------------
// Two levels, built by k_key_sums in ONE pass over the key block with no
// communication between workgroups: the key block is cut into n_mega <= 256 runs
// of wt_per_mega wave tiles; mega_sum[m] = data bytes of run m (at most 64 MiB:
// a stream holds < 2^32 values), wt_rel[wt] = data bytes between the start of
// wave tile wt's run and the wave tile.  The bytes before a run are summed where
// they are needed: 256 sums are one 16-byte load per lane and three DPP
// reductions, cheaper than a scan kernel of their own (a launch costs ~5 us).
struct Dir {
  const u32* mega_sum;  // kMegas entries are readable; those at or beyond n_mega hold anything
  const u32* wt_rel;
  u32 wt_per_mega;
};
// Data bytes before mega tile m (wave-uniform in and out; every lane takes part).
__device__ __forceinline__ u64 dir_mega_base(const Dir& d, u32 m, int lane) {
  const uint4 s = *reinterpret_cast<const uint4*>(d.mega_sum + 4 * lane);
  const u32 i0 = 4u * lane;
  const u64 mine = (u64)(i0 < m ? s.x : 0u) + (i0 + 1u < m ? s.y : 0u) + (i0 + 2u < m ? s.z : 0u) + (i0 + 3u < m ? s.w : 0u);
  return wave_sum64(mine);
}
// Data offset of wave tile wt (wave-uniform in, wave-uniform out; every lane takes part).
__device__ __forceinline__ u64 dir_offset(const Dir& d, u32 wt, int lane) {
  return dir_mega_base(d, wt / d.wt_per_mega, lane) + d.wt_rel[wt];
}
// The same for a wave that walks several wave tiles: the mega tile's base is summed once and
// again only when the walk crosses into another mega tile.
struct DirWalk {
  u32 mega = 0xFFFFFFFFu;
  u64 base = 0;
  __device__ __forceinline__ u64 offset(const Dir& d, u32 wt, int lane) {
    const u32 m = wt / d.wt_per_mega;
    if (m != mega) {  // wave-uniform
      mega = m;
      base = dir_mega_base(d, m, lane);
    }
    return base + d.wt_rel[wt];
  }
};

}  // namespace svb

// Reductions and element-wise maps: exact size, stream validation, zigzag.
#pragma once
#include "svb_device.h"

namespace svb {

// Fill `n` 4-byte words with `value` (n small: header, status words, result
// words).  A kernel of the library's own instead of hipMemsetAsync: memset
// nodes recorded into a HIP graph were seen to write garbage from the second
// replay on (ROCm 7.2), a kernel node replays like any other.
__global__ void k_fill_u32(u32* __restrict__ p, u32 value, u64 n) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) p[i] = value;
}

// *dst = *src (one word, one thread).
__global__ void k_copy_u32(const u32* __restrict__ src, u32* __restrict__ dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = *src;
}

// Sticky error: OR the error word of a workspace header into *sticky (one
// workgroup; ordered on the stream after the kernels whose give-ups it keeps).
__global__ void k_keep_error(const u32* __restrict__ ws_error, u32* __restrict__ sticky) {
  if (threadIdx.x == 0 && *ws_error) *sticky |= *ws_error;
}

// Block-wide sum of a 64-bit value, result valid in thread 0.
__device__ __forceinline__ u64 block_sum64(u64 x, u64* s_tmp) {
  x = wave_sum64(x);
  if (lane_id() == 0) s_tmp[threadIdx.x >> 6] = x;
  __syncthreads();
  u64 r = 0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += s_tmp[w];
  return r;
}

// ---------------------------------------------------------------------------
// k_size: exact stream size = keyLen + sum of widths (reference
// src/streamvbyte_encode.c:69-108, src/streamvbyte_x64_encode.c:33-51).
// With DELTA the widths are those of in[i]-in[i-1]; ZZ zigzag-maps the value
// (or the difference) first.
// *result must be zero before the launch.  Reads 4 B per value.
constexpr u64 kSizeChunk = 64 * kBlock;  // quads (of 4 values) per workgroup: 256 KiB of input per atomic
template <int V, bool DELTA, bool ZZ, int NT = 0>
__global__ __launch_bounds__(kBlock) void k_size(const u32* __restrict__ in, u32 count, u32 prev,
                                                 u64* __restrict__ result) {
  __shared__ u64 s_tmp[kWaves];
  const u64 nquads = (u64)count >> 2;
  u64 acc = 0;
  // workgroup b takes the contiguous quads [b * kSizeChunk, (b + 1) * kSizeChunk) (grid-stride beyond that)
  for (u64 c0 = (u64)blockIdx.x * kSizeChunk; c0 < nquads; c0 += (u64)gridDim.x * kSizeChunk)
  for (u64 q = c0 + threadIdx.x; q < min(nquads, c0 + kSizeChunk); q += kBlock) {
    const uint4 x = ld_stream16<(NT & kNtLoads) != 0>(in + 4 * q);
    u32 v[4] = {x.x, x.y, x.z, x.w};
    if (DELTA) {
      const u32 p = q ? in[4 * q - 1] : prev;
      const u32 d0 = v[0] - p, d1 = v[1] - v[0], d2 = v[2] - v[1], d3 = v[3] - v[2];
      v[0] = d0; v[1] = d1; v[2] = d2; v[3] = d3;
    }
    if (ZZ) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = zz_enc(v[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += len_of<V>(code_of<V>(v[k]));
  }
  // tail values (count % 4), handled by the first threads of block 0
  if (blockIdx.x == 0 && threadIdx.x < (count & 3u)) {
    const u64 i = (nquads << 2) + threadIdx.x;
    u32 v = in[i];
    if (DELTA) v -= i ? in[i - 1] : prev;
    if (ZZ) v = zz_enc(v);
    acc += len_of<V>(code_of<V>(v));
  }
  const u64 tot = block_sum64(acc, s_tmp);
  if (threadIdx.x == 0) {
    const u64 add = tot + (blockIdx.x == 0 ? (((u64)count + 3) >> 2) : 0);
    if (add) atomicAdd(reinterpret_cast<unsigned long long*>(result), (unsigned long long)add);
  }
}

// ---------------------------------------------------------------------------
// k_size_batch: k_size for every stream of a batch -- sizes[s] = the bytes streamvbyte_compressedbytes
// (reference src/streamvbyte_encode.c:91-108) gives for stream s, whose counts[s] values start at
// values + value_off[s].  A wavefront per stream; values come in 16-byte loads (a stream may start at
// any uint32), the last one to three singly.  With DELTA the widths are those of the differences, the
// first against prevs[s] (0 when prevs is null).
template <int V, bool DELTA, bool ZZ>
__global__ __launch_bounds__(kBlock) void k_size_batch(const u32* __restrict__ values, const u64* __restrict__ value_off,
                                                       const u32* __restrict__ counts, const u32* __restrict__ prevs,
                                                       u32 nstreams, u64* __restrict__ sizes) {
  const int lane = lane_id();
  const u32 s = blockIdx.x * kWaves + wave_id();
  if (s >= nstreams) return;
  const u32 count = counts[s];
  const u32* in = values + value_off[s];
  const u32 prev = (DELTA && prevs) ? prevs[s] : 0u;
  const u32 nquads = count >> 2;
  u32 acc = 0;  // at most 4 bytes for each of < 2^32 values, 64 lanes: a lane's share fits 32 bits
  for (u32 q = lane; q < nquads; q += 64) {
    const uint4 x = ld_stream16<false>(in + 4ull * q);
    u32 v[4] = {x.x, x.y, x.z, x.w};
    if (DELTA) {
      const u32 p = q ? in[4ull * q - 1] : prev;
      const u32 d0 = v[0] - p, d1 = v[1] - v[0], d2 = v[2] - v[1], d3 = v[3] - v[2];
      v[0] = d0; v[1] = d1; v[2] = d2; v[3] = d3;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += len_of<V>(code_of<V>(ZZ ? zz_enc(v[k]) : v[k]));
  }
  if ((u32)lane < (count & 3u)) {
    const u64 i = ((u64)nquads << 2) + lane;
    u32 v = in[i];
    if (DELTA) v -= i ? in[i - 1] : prev;
    acc += len_of<V>(code_of<V>(ZZ ? zz_enc(v) : v));
  }
  const u64 tot = wave_sum64((u64)acc);
  if (lane == 0) sizes[s] = tot + (((u64)count + 3) >> 2);
}

// ---------------------------------------------------------------------------
// k_validate_sum: data bytes promised by the first `count` codes of the key
// block (reference src/streamvbyte_decode.c:113-133,
// src/streamvbyte_0124_decode.c:206-226) and, in the workgroup that finishes
// last, the final comparison with inLength (src/streamvbyte_decode.c:134-135).
// *accum must be zero before launch; in_length >= keyLen > 0.
// Reads 0.25 B per value.
// Workgroups of NT threads; a workgroup's round r takes the 16 NT bytes of keys at piece
// blockIdx.x + r * gridDim.x, so that at any moment the grid reads one compact window of
// the key block (contiguous shares per workgroup, 256 KiB apart, all start on the same
// memory channels and crawl).  At most a few hundred workgroups: their one atomic each
// lands on the same word, ~12 ns apiece.  kValAhead 16-byte loads in flight per thread.
constexpr int kValAhead = 6;
template <int V, int NT = 1024>
__global__ __launch_bounds__(NT) void k_validate_sum(const u8* __restrict__ in, u32 count,
                                                     u64 in_length, u64* __restrict__ accum,
                                                     u32* __restrict__ ok) {
  __shared__ u64 s_tmp[16];
  const u64 keylen = ((u64)count + 3) >> 2;
  const u64 nchunks = (keylen + 15) >> 4;  // 16-byte pieces of the key block
  // whole pieces in the streaming loop; the last piece of the key block, when it is partial, apart
  // (byte loads that stay inside the key block: validation may not trust what follows it)
  const u64 whole = keylen >> 4;
  u64 acc = 0;
  auto add_piece = [&](const uint4& p, u64 c) {
    const u32 nv = (u32)min((u64)64, (u64)count - (c << 6));  // values coded by piece c
    const u32 w[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const u32 nvd = nv > 16u * d ? min(nv - 16u * d, 16u) : 0u;
      acc += word_len<V>(mask_codes(w[d], nvd), nvd);
    }
  };
  if (blockIdx.x == 0 && threadIdx.x == 0 && whole < nchunks) {
    const u64 koff = whole << 4;
    u32 w[4] = {0, 0, 0, 0};
    for (u64 b = koff; b < keylen; ++b) w[(b - koff) >> 2] |= (u32)in[b] << (8 * ((b - koff) & 3));
    add_piece(make_uint4(w[0], w[1], w[2], w[3]), whole);
  }
  if (whole) {  // workgroup-uniform
    // Unconditional loads at a clamped piece index (a load under a branch makes the compiler wait for
    // ALL outstanding loads at every use); what lies past the end is never added.
    // A piece index past the end is replaced by the thread's OWN first piece (a cache hit at an address no
    // other thread asks for): sending every such load to the last whole piece makes thousands of
    // wavefronts queue on one cache line, which cost more than reading the whole key block.
    const u64 stride = (u64)gridDim.x * NT;
    const u64 first = (u64)blockIdx.x * NT + threadIdx.x;
    const u64 own = min(first, whole - 1);
    auto load = [&](u64 c) -> uint4 { return ld_u128(in + ((c < whole ? c : own) << 4)); };
    uint4 k[kValAhead];
#pragma unroll
    for (int i = 0; i < kValAhead; ++i) k[i] = load(first + stride * i);
    // unrolled kValAhead times: every load keeps its own register (rotating the registers by moves
    // would wait for the loads being moved)
    u32 quick = 0;  // a thread adds at most 2^26 / 1024 pieces of at most 256 bytes
    for (u64 wb = (u64)blockIdx.x * NT; wb < whole; wb += stride * kValAhead) {  // the workgroup's first piece: scalar
#pragma unroll
      for (int r = 0; r < kValAhead; ++r) {
        const u64 c = wb + stride * r + threadIdx.x;
        // scalar: every piece of the workgroup in this round is a whole piece whose 64 values all exist
        const u64 wg_end = wb + stride * r + NT;
        if (wg_end <= whole && (wg_end << 6) <= (u64)count) quick += dir_counts<V>(k[r]);
        else if (c < whole) add_piece(k[r], c);
        k[r] = load(c + stride * kValAhead);  // behind the last use: the load lands in the register it frees
      }
    }
    acc += quick;
  }
  const u64 tot = block_sum64(acc, s_tmp);
  if (threadIdx.x == 0) {
    // one atomic per workgroup carries its sum (low 48 bits; a stream holds < 2^35
    // data bytes) and counts it in (high 16 bits; the grid has < 2^16 workgroups)
    const u64 old = atomicAdd(reinterpret_cast<unsigned long long*>(accum), (unsigned long long)(tot | (1ull << 48)));
    if ((old >> 48) == gridDim.x - 1) {  // every other workgroup's sum is in
      *ok = ((old & ((1ull << 48) - 1)) + tot) == in_length - keylen;
      *accum = 0;  // nobody adds after the last: the next call finds the accumulator as it must be
    }
  }
}

// ---------------------------------------------------------------------------
// Zigzag maps (reference src/streamvbyte_zigzag.c:3-29).  8 B per value.
// MODE 0: encode, 1: decode, 2: delta encode (out[i] = zz(in[i]-in[i-1])).
template <int MODE, int NT = 0>
__global__ __launch_bounds__(kBlock) void k_zigzag(const u32* __restrict__ in,
                                                   u32* __restrict__ out, u64 n, u32 prev) {
  const u64 nquads = n >> 2;
  for (u64 q = (u64)blockIdx.x * kBlock + threadIdx.x; q < nquads; q += (u64)gridDim.x * kBlock) {
    const uint4 x = ld_stream16<(NT & kNtLoads) != 0>(in + 4 * q);
    uint4 y;
    if (MODE == 0) {
      y = make_uint4(zz_enc(x.x), zz_enc(x.y), zz_enc(x.z), zz_enc(x.w));
    } else if (MODE == 1) {
      y = make_uint4(zz_dec(x.x), zz_dec(x.y), zz_dec(x.z), zz_dec(x.w));
    } else {
      const u32 p = q ? in[4 * q - 1] : prev;
      y = make_uint4(zz_enc(x.x - p), zz_enc(x.y - x.x), zz_enc(x.z - x.y), zz_enc(x.w - x.z));
    }
    st_stream16<(NT & kNtStores) != 0>(out + 4 * q, y);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const u64 i = (nquads << 2) + threadIdx.x;
    const u32 v = in[i];
    u32 r;
    if (MODE == 0) r = zz_enc(v);
    else if (MODE == 1) r = zz_dec(v);
    else r = zz_enc(v - (i ? in[i - 1] : prev));
    out[i] = r;
  }
}

}  // namespace svb

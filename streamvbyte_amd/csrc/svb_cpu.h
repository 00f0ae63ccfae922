// Host-side codec of libstreamvbyte.so.2: what the reference-named functions run
// when the work is too small to be worth a trip over PCIe, or when the process
// has no gfx950 device at all.  Written for this library; it shares no code with
// the test oracle and is never used by the svbgpu_* device-pointer API.
//
// Stream format and return values are those of the reference
// (src/streamvbyte_encode.c:14-62,113-137, src/streamvbyte_decode.c:18-87,
// src/streamvbyte_0124_encode.c:12-57,96-113, src/streamvbyte_0124_decode.c:38-79,165-229,
// src/streamvbytedelta_encode.c:39-76, src/streamvbytedelta_decode.c:34-74,
// src/streamvbyte_zigzag.c:3-38).  Two tiers:
//   * portable scalar loops (any x86-64 / any host), and
//   * SSE4.1 / SSSE3 coders for whole groups of four values (one byte shuffle per
//     group, masks built at first use from the width rule, not stored tables), and
//   * AVX-512 coders for blocks of sixteen values (one byte compress / expand per block),
//     chosen at run time with __builtin_cpu_supports.
// Buffer rules as in the reference: decoders may read up to 16 bytes past the
// stream (STREAMVBYTE_PADDING) and write exactly `count` values; encoders write
// only inside streamvbyte_max_compressedbytes(count).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <mutex>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace svbcpu {

constexpr int k1234 = 0;
constexpr int k0124 = 1;

// ---- width rule ----------------------------------------------------------------
// 1234: code = number of bytes needed - 1; 0124: 0 for zero, 1 / 2 for one / two
// bytes, 3 (four bytes) for anything wider than two bytes.
inline unsigned code_1234(uint32_t v) { return (31u - (unsigned)__builtin_clz(v | 1u)) >> 3; }
inline unsigned code_0124(uint32_t v) {
  if (v == 0) return 0;
  const unsigned bytes = ((31u - (unsigned)__builtin_clz(v)) >> 3) + 1u;  // 1..4
  return bytes >= 3 ? 3u : bytes;
}
inline unsigned len_1234(unsigned code) { return code + 1u; }
inline unsigned len_0124(unsigned code) { return (1u << code) >> 1; }

template <int V>
inline unsigned code_of(uint32_t v) { return V == k1234 ? code_1234(v) : code_0124(v); }
template <int V>
inline unsigned len_of(unsigned code) { return V == k1234 ? len_1234(code) : len_0124(code); }

// ---- scalar encode ----------------------------------------------------------------
// Encodes values [first, count) (first % 4 == 0) behind what keys / data already hold.
template <int V, bool DELTA>
uint8_t* encode_scalar_from(const uint32_t* in, uint32_t first, uint32_t count, uint8_t* keys, uint8_t* data,
                            uint32_t prev) {
  for (uint32_t i = first; i < count;) {
    unsigned key = 0;
    const uint32_t in_group = count - i < 4 ? count - i : 4;
    for (uint32_t k = 0; k < in_group; ++k, ++i) {
      uint32_t v = in[i];
      if (DELTA) {
        const uint32_t cur = v;
        v = cur - prev;  // uint32 wrap-around: the input need not be sorted
        prev = cur;
      }
      const unsigned c = code_of<V>(v);
      key |= c << (2 * k);
      // all four bytes are stored (little-endian host) and the pointer moves on by the value's
      // length: the spare bytes are overwritten by the next value or lie inside the 16 bytes of
      // slack of streamvbyte_max_compressedbytes
      std::memcpy(data, &v, 4);
      data += len_of<V>(c);
    }
    *keys++ = (uint8_t)key;
  }
  return data;
}

template <int V, bool DELTA>
size_t encode(const uint32_t* in, uint32_t count, uint8_t* out, uint32_t prev);

// ---- scalar decode ----------------------------------------------------------------
template <int V>
inline uint32_t read_value(const uint8_t* p, unsigned len) {
  uint32_t v = 0;
  for (unsigned b = 0; b < len; ++b) v |= (uint32_t)p[b] << (8 * b);
  return v;
}
// Decodes values [first, count) given the key pointer / data pointer at `first` (first % 4 == 0).
template <int V, bool DELTA>
const uint8_t* decode_scalar_from(const uint8_t* keys, const uint8_t* data, uint32_t* out, uint32_t first,
                                  uint32_t count, uint32_t& prev) {
  for (uint32_t i = first; i < count;) {
    const unsigned key = *keys++;
    const uint32_t in_group = count - i < 4 ? count - i : 4;
    for (uint32_t k = 0; k < in_group; ++k, ++i) {
      const unsigned len = len_of<V>((key >> (2 * k)) & 3u);
      uint32_t v = read_value<V>(data, len);
      data += len;
      if (DELTA) {
        v += prev;
        prev = v;
      }
      out[i] = v;
    }
  }
  return data;
}

#if defined(__x86_64__)
// ---- SSE4.1 encode of whole groups ----------------------------------------------------
// Four values at a time: widths from three unsigned compares, the key byte by one
// multiply, the data bytes compacted by one byte shuffle whose mask is looked up
// by the key (built at first use from the width rule).  A 16-byte store per group
// stays inside streamvbyte_max_compressedbytes.
struct PackTables {
  alignas(16) uint8_t mask[2][256][16];
  uint8_t len[2][256];
  PackTables() {
    for (int variant = 0; variant < 2; ++variant)
      for (int key = 0; key < 256; ++key) {
        unsigned pos = 0;
        for (unsigned b = 0; b < 16; ++b) mask[variant][key][b] = 0x80;
        for (int k = 0; k < 4; ++k) {
          const unsigned code = (key >> (2 * k)) & 3u;
          const unsigned n = variant == k1234 ? len_1234(code) : len_0124(code);
          for (unsigned b = 0; b < n; ++b) mask[variant][key][pos + b] = (uint8_t)(4 * k + b);
          pos += n;
        }
        len[variant][key] = (uint8_t)pos;
      }
  }
};
inline const PackTables& pack_tables() {
  static const PackTables t;
  return t;
}
inline bool have_sse41() {
  static const bool ok = __builtin_cpu_supports("sse4.1");
  return ok;
}
template <int V, bool DELTA>
__attribute__((target("sse4.1"))) uint32_t encode_groups_sse41(const uint32_t* in, uint32_t count, uint8_t*& keys,
                                                                uint8_t*& data, uint32_t& prev) {
  const PackTables& T = pack_tables();
  const uint32_t groups = count / 4;
  // thresholds: a value at or below t[i] needs no byte beyond the i-th class
  const __m128i t1 = _mm_set1_epi32(V == k1234 ? 0xFF : 0), t2 = _mm_set1_epi32(V == k1234 ? 0xFFFF : 0xFF),
                t3 = _mm_set1_epi32(V == k1234 ? 0xFFFFFF : 0xFFFF);
  const __m128i three = _mm_set1_epi32(3);
  const __m128i gather = _mm_setr_epi8(0, 4, 8, 12, -128, -128, -128, -128, -128, -128, -128, -128, -128, -128, -128, -128);
  __m128i before = _mm_set1_epi32((int)prev);
  uint8_t* d = data;
  uint32_t g = 0;
  // Four groups at a time: their 16 width codes are worked out together and folded into four key
  // bytes with two multiply-adds, so that the only chain from group to group is the data pointer
  // (d += len[key]: a table load and an add) and not the whole classification.
#define SVBCPU_CODES(v) /* 3 - [v <= t1] - [v <= t2] - [v <= t3]  (each compare gives -1 where true) */ \
  _mm_add_epi32(_mm_add_epi32(_mm_add_epi32(three, _mm_cmpeq_epi32(_mm_min_epu32((v), t1), (v))),      \
                              _mm_cmpeq_epi32(_mm_min_epu32((v), t2), (v))),                           \
                _mm_cmpeq_epi32(_mm_min_epu32((v), t3), (v)))
  const __m128i pair = _mm_set1_epi16(0x0401), quad = _mm_set1_epi32(0x00100001);
  const __m128i zero = _mm_setzero_si128(), ones8 = _mm_set1_epi8(1), weights = _mm_set1_epi32(0x08040201);
  // index = set of non-zero bytes; 1234: the highest non-zero byte's number; 0124: that plus one, at most 3
  const __m128i lut = V == k1234 ? _mm_setr_epi8(0, 0, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3)
                                 : _mm_setr_epi8(0, 1, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3);
  for (; g + 4 <= groups; g += 4) {
    __m128i v[4];
    for (int u = 0; u < 4; ++u) {
      v[u] = _mm_loadu_si128(reinterpret_cast<const __m128i*>(in + 4 * (size_t)(g + u)));
      if (DELTA) {
        const __m128i shifted = _mm_alignr_epi8(v[u], before, 12);  // {prev, v0, v1, v2}
        before = v[u];
        v[u] = _mm_sub_epi32(v[u], shifted);
      }
    }
    // A value's width code follows from WHICH of its four bytes are non-zero: give byte k the weight 1 << k,
    // add the weights of the non-zero bytes of every value (0..15) and look the code up, 16 values at a time.
    const __m128i n0 = _mm_andnot_si128(_mm_cmpeq_epi8(v[0], zero), weights), n1 = _mm_andnot_si128(_mm_cmpeq_epi8(v[1], zero), weights),
                  n2 = _mm_andnot_si128(_mm_cmpeq_epi8(v[2], zero), weights), n3 = _mm_andnot_si128(_mm_cmpeq_epi8(v[3], zero), weights);
    const __m128i s01 = _mm_hadd_epi16(_mm_maddubs_epi16(n0, ones8), _mm_maddubs_epi16(n1, ones8)),
                  s23 = _mm_hadd_epi16(_mm_maddubs_epi16(n2, ones8), _mm_maddubs_epi16(n3, ones8));
    const __m128i c16 = _mm_shuffle_epi8(lut, _mm_packus_epi16(s01, s23));  // one code per byte, in order
    const __m128i k32 = _mm_madd_epi16(_mm_maddubs_epi16(c16, pair), quad);  // c0 + 4 c1 + 16 c2 + 64 c3 per group
    const uint32_t k4 = (uint32_t)_mm_cvtsi128_si32(_mm_shuffle_epi8(k32, gather));
    std::memcpy(keys + g, &k4, 4);
    for (int u = 0; u < 4; ++u) {
      const unsigned key = (k4 >> (8 * u)) & 0xFFu;
      _mm_storeu_si128(reinterpret_cast<__m128i*>(d),
                       _mm_shuffle_epi8(v[u], _mm_load_si128(reinterpret_cast<const __m128i*>(T.mask[V][key]))));
      d += T.len[V][key];
    }
  }
  for (; g < groups; ++g) {
    __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(in + 4 * (size_t)g));
    if (DELTA) {
      const __m128i shifted = _mm_alignr_epi8(v, before, 12);  // {prev, v0, v1, v2}
      before = v;
      v = _mm_sub_epi32(v, shifted);
    }
    const __m128i c = SVBCPU_CODES(v);
    const uint32_t packed = (uint32_t)_mm_cvtsi128_si32(_mm_shuffle_epi8(c, gather));  // one code per byte
    const unsigned key = (packed * 0x01041040u) >> 24;  // c0 | c1 << 2 | c2 << 4 | c3 << 6
    keys[g] = (uint8_t)key;
    _mm_storeu_si128(reinterpret_cast<__m128i*>(d),
                     _mm_shuffle_epi8(v, _mm_load_si128(reinterpret_cast<const __m128i*>(T.mask[V][key]))));
    d += T.len[V][key];
  }
#undef SVBCPU_CODES
  if (DELTA && groups) prev = in[4 * (size_t)groups - 1];
  keys += groups;
  data = d;
  return groups * 4;
}

// ---- SSSE3 decode of whole groups ----------------------------------------------------
// For every key byte: the pshufb mask that spreads the group's packed bytes over
// four 32-bit lanes (0x80 = zero fill) and the group's data length.  Built once
// from the width rule.
struct ShuffleTables {
  alignas(16) uint8_t mask[2][256][16];
  uint8_t len[2][256];
  ShuffleTables() {
    for (int variant = 0; variant < 2; ++variant)
      for (int key = 0; key < 256; ++key) {
        unsigned pos = 0;
        for (int k = 0; k < 4; ++k) {
          const unsigned code = (key >> (2 * k)) & 3u;
          const unsigned n = variant == k1234 ? len_1234(code) : len_0124(code);
          for (unsigned b = 0; b < 4; ++b) mask[variant][key][4 * k + b] = b < n ? (uint8_t)(pos + b) : 0x80;
          pos += n;
        }
        len[variant][key] = (uint8_t)pos;
      }
  }
};
inline const ShuffleTables& tables() {
  static const ShuffleTables t;  // thread-safe initialisation (C++11 magic static)
  return t;
}
inline bool have_ssse3() {
  static const bool ok = __builtin_cpu_supports("ssse3");
  return ok;
}

// Decodes every whole group (count / 4 of them): a 16-byte load at the group's
// first data byte stays inside the stream plus its 16 bytes of padding, a
// 16-byte store inside out[0 .. count).  Returns the number of values decoded;
// keys / data / prev advance.
template <int V, bool DELTA>
__attribute__((target("ssse3"))) uint32_t decode_groups_ssse3(const uint8_t*& keys, const uint8_t*& data,
                                                               uint32_t* out, uint32_t count, uint32_t& prev) {
  const ShuffleTables& T = tables();
  const uint32_t groups = count / 4;  // only full groups
  __m128i run = _mm_set1_epi32((int)prev);
  uint32_t g = 0;
  const uint8_t* k = keys;
  const uint8_t* d = data;
  for (; g < groups; ++g) {
    const unsigned key = k[g];
    const __m128i bytes = _mm_loadu_si128(reinterpret_cast<const __m128i*>(d));  // may read past the group: padding
    __m128i v = _mm_shuffle_epi8(bytes, _mm_load_si128(reinterpret_cast<const __m128i*>(T.mask[V][key])));
    d += T.len[V][key];
    if (DELTA) {
      // inclusive prefix sum inside the register, then the carry of the previous group
      v = _mm_add_epi32(v, _mm_slli_si128(v, 4));
      v = _mm_add_epi32(v, _mm_slli_si128(v, 8));
      v = _mm_add_epi32(v, run);
      run = _mm_shuffle_epi32(v, 0xFF);
    }
    _mm_storeu_si128(reinterpret_cast<__m128i*>(out + 4 * (size_t)g), v);
  }
  if (DELTA) prev = (uint32_t)_mm_cvtsi128_si32(run);
  keys = k + groups;
  data = d;
  return groups * 4;
}

// ---- AVX-512 (VBMI2) encode / decode of sixteen values at a time ----------------------------
// Where the CPU has byte compress / expand (vpcompressb, vpexpandb: Ice Lake and later, Zen 4 and
// later) a block of sixteen values is ONE such instruction: the encoder keeps byte b of a value iff
// b < its length (a 64-bit byte mask from one compare) and compresses the 64 bytes; the decoder makes
// the same mask from the block's four key bytes, loads exactly popcount(mask) bytes (masked load:
// nothing past the stream is touched) and expands them.  No
// shuffle tables; the chain from block to block is a popcount and an add.  Same bytes as the scalar
// rule above (reference src/streamvbyte_encode.c:14-62, src/streamvbyte_decode.c:18-60,
// src/streamvbyte_0124_encode.c:12-57, src/streamvbyte_0124_decode.c:38-79), checked against the oracle.
#define SVBCPU_AVX512 __attribute__((target("avx512f,avx512bw,avx512vl,avx512cd,avx512vbmi2,bmi2,popcnt")))
inline bool have_avx512() {
  static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") &&
                         __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512cd") &&
                         __builtin_cpu_supports("avx512vbmi2") && __builtin_cpu_supports("bmi2") &&
                         __builtin_cpu_supports("popcnt");
  return ok;
}
// Encodes the whole blocks of sixteen values of in[0 .. count); returns the number of values done.
// A 64-byte store per block stays inside the 4 bytes per value the output buffer holds for the values
// of this and the following blocks (a block is only taken when sixteen values remain).
template <int V, bool DELTA>
SVBCPU_AVX512 uint32_t encode_blocks_avx512(const uint32_t* in, uint32_t count, uint8_t*& keys, uint8_t*& data,
                                            uint32_t& prev) {
  const uint32_t blocks = count / 16;
  const __m512i byte_idx = _mm512_set1_epi32(0x03020100);
  const __m512i rep = _mm512_broadcast_i32x4(_mm_setr_epi8(0, 0, 0, 0, 4, 4, 4, 4, 8, 8, 8, 8, 12, 12, 12, 12));
  __m512i before = _mm512_set1_epi32((int)prev);
  uint8_t* d = data;
  for (uint32_t b = 0; b < blocks; ++b) {
    __m512i v = _mm512_loadu_si512(in + 16 * (size_t)b);
    if (DELTA) {
      const __m512i shifted = _mm512_alignr_epi32(v, before, 15);  // {prev, v0, .., v14}
      before = v;
      v = _mm512_sub_epi32(v, shifted);
    }
    __m512i code, len;
    if (V == k1234) {  // code = (index of the highest set bit of v | 1) / 8, length = code + 1
      code = _mm512_srli_epi32(_mm512_sub_epi32(_mm512_set1_epi32(31), _mm512_lzcnt_epi32(_mm512_or_si512(v, _mm512_set1_epi32(1)))), 3);
      len = _mm512_add_epi32(code, _mm512_set1_epi32(1));
    } else {  // bytes needed = (39 - lzcnt) / 8 (0 for 0); code = min(bytes, 3); length = 0, 1, 2, 4
      const __m512i nbytes = _mm512_srli_epi32(_mm512_sub_epi32(_mm512_set1_epi32(39), _mm512_lzcnt_epi32(v)), 3);
      code = _mm512_min_epu32(nbytes, _mm512_set1_epi32(3));
      len = _mm512_add_epi32(code, _mm512_srli_epi32(_mm512_add_epi32(code, _mm512_set1_epi32(1)), 2));
    }
    const __mmask64 keep = _mm512_cmplt_epu8_mask(byte_idx, _mm512_shuffle_epi8(len, rep));
    _mm512_storeu_si512(d, _mm512_maskz_compress_epi8(keep, v));
    d += (unsigned)_mm_popcnt_u64(keep);
    // sixteen codes, one per byte, to four key bytes: the low two bits of every byte, gathered
    const __m128i c8 = _mm512_cvtepi32_epi8(code);
    const uint32_t k4 = (uint32_t)_pext_u64((uint64_t)_mm_cvtsi128_si64(c8), 0x0303030303030303ull) |
                        ((uint32_t)_pext_u64((uint64_t)_mm_extract_epi64(c8, 1), 0x0303030303030303ull) << 16);
    std::memcpy(keys + 4 * (size_t)b, &k4, 4);
  }
  if (DELTA && blocks) prev = in[16 * (size_t)blocks - 1];
  keys += 4 * (size_t)blocks;
  data = d;
  return blocks * 16;
}
// Decodes the whole blocks of sixteen values; returns the number of values done.  The byte mask of a
// block comes from its four key bytes in five vector instructions: the sixteen 2-bit codes spread over
// the 32-bit lanes (broadcast + variable shift), their lengths looked up and copied to the four bytes
// of each lane (two byte shuffles), and byte b of a lane is present iff b < length.
template <int V, bool DELTA>
SVBCPU_AVX512 uint32_t decode_blocks_avx512(const uint8_t*& keys, const uint8_t*& data, uint32_t* out, uint32_t count,
                                            uint32_t& prev) {
  const uint32_t blocks = count / 16;
  const __m512i zero = _mm512_setzero_si512(), last = _mm512_set1_epi32(15), three = _mm512_set1_epi32(3);
  const __m512i shifts = _mm512_setr_epi32(0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30);
  const __m512i byte_idx = _mm512_set1_epi32(0x03020100);
  const __m512i rep = _mm512_broadcast_i32x4(_mm_setr_epi8(0, 0, 0, 0, 4, 4, 4, 4, 8, 8, 8, 8, 12, 12, 12, 12));
  const __m512i lens = _mm512_broadcast_i32x4(V == k1234 ? _mm_setr_epi8(1, 2, 3, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
                                                         : _mm_setr_epi8(0, 1, 2, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0));
  __m512i run = _mm512_set1_epi32((int)prev);
  const uint8_t* d = data;
  for (uint32_t b = 0; b < blocks; ++b) {
    uint32_t k4;
    std::memcpy(&k4, keys + 4 * (size_t)b, 4);
    const __m512i code = _mm512_and_si512(_mm512_srlv_epi32(_mm512_set1_epi32((int)k4), shifts), three);
    const __m512i lenrep = _mm512_shuffle_epi8(lens, _mm512_shuffle_epi8(code, rep));
    const __mmask64 m = _mm512_cmplt_epu8_mask(byte_idx, lenrep);
    const unsigned n = (unsigned)_mm_popcnt_u64(m);
    const __mmask64 have = _bzhi_u64(~0ull, n);  // the low n bits (all 64 for n = 64): nothing past the stream is read
    __m512i v = _mm512_maskz_expand_epi8(m, _mm512_maskz_loadu_epi8(have, d));
    d += n;
    if (DELTA) {  // inclusive prefix sum over the sixteen lanes, then the carry of the block before
      v = _mm512_add_epi32(v, _mm512_alignr_epi32(v, zero, 15));
      v = _mm512_add_epi32(v, _mm512_alignr_epi32(v, zero, 14));
      v = _mm512_add_epi32(v, _mm512_alignr_epi32(v, zero, 12));
      v = _mm512_add_epi32(v, _mm512_alignr_epi32(v, zero, 8));
      v = _mm512_add_epi32(v, run);
      run = _mm512_permutexvar_epi32(last, v);
    }
    _mm512_storeu_si512(out + 16 * (size_t)b, v);
  }
  if (DELTA && blocks) prev = out[16 * (size_t)blocks - 1];
  keys += 4 * (size_t)blocks;
  data = d;
  return blocks * 16;
}
#endif

template <int V, bool DELTA>
size_t decode(const uint8_t* in, uint32_t* out, uint32_t count, uint32_t prev) {
  if (count == 0) return 0;
  const size_t keylen = ((size_t)count + 3) / 4;
  const uint8_t* keys = in;
  const uint8_t* data = in + keylen;
  uint32_t done = 0;
#if defined(__x86_64__)
  if (count >= 32 && have_avx512()) done = decode_blocks_avx512<V, DELTA>(keys, data, out, count, prev);
  if (count - done >= 8 && have_ssse3()) done += decode_groups_ssse3<V, DELTA>(keys, data, out + done, count - done, prev);
#endif
  data = decode_scalar_from<V, DELTA>(keys, data, out, done, count, prev);
  return (size_t)(data - in);
}

template <int V, bool DELTA>
size_t encode(const uint32_t* in, uint32_t count, uint8_t* out, uint32_t prev) {
  if (count == 0) return 0;
  const size_t keylen = ((size_t)count + 3) / 4;
  uint8_t* keys = out;
  uint8_t* data = out + keylen;
  uint32_t done = 0;
#if defined(__x86_64__)
  if (count >= 32 && have_avx512()) done = encode_blocks_avx512<V, DELTA>(in, count, keys, data, prev);
  if (count - done >= 8 && have_sse41()) done += encode_groups_sse41<V, DELTA>(in + done, count - done, keys, data, prev);
#endif
  data = encode_scalar_from<V, DELTA>(in, done, count, keys, data, prev);
  return (size_t)(data - out);
}

template <int V>
size_t compressedbytes(const uint32_t* in, uint32_t count) {
  size_t bytes = ((size_t)count + 3) / 4;
  for (uint32_t i = 0; i < count; ++i) bytes += len_of<V>(code_of<V>(in[i]));
  return bytes;
}

// The reference's rule (src/streamvbyte_decode.c:89-136): an empty stream is valid
// only for an empty output; the key block must fit; the lengths the first `count`
// codes promise must add up to exactly the bytes behind the key block.
template <int V>
bool validate(const uint8_t* in, size_t in_length, uint32_t count) {
  if (in_length == 0 || count == 0) return in_length == (size_t)count;
  const size_t keylen = ((size_t)count + 3) / 4;
  if (keylen > in_length) return false;
  uint64_t promised = 0;
  const size_t full = count / 4;
  for (size_t k = 0; k < full; ++k) {
    const unsigned key = in[k];
    for (int j = 0; j < 4; ++j) promised += len_of<V>((key >> (2 * j)) & 3u);
  }
  if (count & 3u) {
    const unsigned key = in[full];
    for (unsigned j = 0; j < (count & 3u); ++j) promised += len_of<V>((key >> (2 * j)) & 3u);
  }
  return promised == (uint64_t)(in_length - keylen);
}

inline uint32_t zz_enc(int32_t v) { return ((uint32_t)v << 1) ^ (uint32_t)(v >> 31); }
inline int32_t zz_dec(uint32_t u) { return (int32_t)((u >> 1) ^ (0u - (u & 1u))); }

inline void zigzag(int mode, const void* in_, void* out_, size_t n, int32_t prev) {
  if (mode == 0) {
    const int32_t* in = static_cast<const int32_t*>(in_);
    uint32_t* out = static_cast<uint32_t*>(out_);
    for (size_t i = 0; i < n; ++i) out[i] = zz_enc(in[i]);
  } else if (mode == 1) {
    const uint32_t* in = static_cast<const uint32_t*>(in_);
    int32_t* out = static_cast<int32_t*>(out_);
    for (size_t i = 0; i < n; ++i) out[i] = zz_dec(in[i]);
  } else if (mode == 2) {
    const int32_t* in = static_cast<const int32_t*>(in_);
    uint32_t* out = static_cast<uint32_t*>(out_);
    for (size_t i = 0; i < n; ++i) {
      const int32_t cur = in[i];
      out[i] = zz_enc((int32_t)((uint32_t)cur - (uint32_t)prev));  // wrap-around difference
      prev = cur;
    }
  } else {
    const uint32_t* in = static_cast<const uint32_t*>(in_);
    int32_t* out = static_cast<int32_t*>(out_);
    for (size_t i = 0; i < n; ++i) {
      prev = (int32_t)((uint32_t)zz_dec(in[i]) + (uint32_t)prev);
      out[i] = prev;
    }
  }
}

// ---- run-time dispatch on variant / delta --------------------------------------------
inline size_t encode_any(int variant, bool delta, const uint32_t* in, uint32_t count, uint8_t* out, uint32_t prev) {
  if (variant == k1234) return delta ? encode<k1234, true>(in, count, out, prev) : encode<k1234, false>(in, count, out, 0);
  return delta ? encode<k0124, true>(in, count, out, prev) : encode<k0124, false>(in, count, out, 0);
}
inline size_t decode_any(int variant, bool delta, const uint8_t* in, uint32_t* out, uint32_t count, uint32_t prev) {
  if (variant == k1234) return delta ? decode<k1234, true>(in, out, count, prev) : decode<k1234, false>(in, out, count, 0);
  return delta ? decode<k0124, true>(in, out, count, prev) : decode<k0124, false>(in, out, count, 0);
}

}  // namespace svbcpu

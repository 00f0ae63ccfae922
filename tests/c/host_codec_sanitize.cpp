// AddressSanitizer + UndefinedBehaviorSanitizer run of the host-side code (CPU build only):
//   * the library's host codec (streamvbyte_amd/csrc/svb_cpu.h), included as it is,
//   * the scalar checker (oracle/svb_oracle.c), compiled into the same program,
// on buffers of EXACTLY the sizes the API promises (input streams readable for
// STREAMVBYTE_PADDING bytes past their end and not one more, encoder outputs of
// streamvbyte_max_compressedbytes, decoder outputs of count values), so that any
// read or write outside the contract faults.  Every result is also compared between
// the two implementations.  Built and run by tests/test_sanitizers.py
// (mirrors the reference's STREAMVBYTE_SANITIZE / _UNDEFINED builds, CMakeLists.txt:63,82-94,149-153).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../streamvbyte_amd/csrc/svb_cpu.h"

extern "C" {
size_t svbo_encode(int variant, const uint32_t* in, uint32_t n, uint8_t* out, int delta, uint32_t prev);
size_t svbo_decode(int variant, const uint8_t* in, uint32_t* out, uint32_t n, int delta, uint32_t prev);
size_t svbo_compressedbytes(int variant, const uint32_t* in, uint32_t n);
int svbo_validate(int variant, const uint8_t* in, size_t inlen, uint32_t n);
void svbo_zigzag_encode(const int32_t* in, uint32_t* out, size_t n);
void svbo_zigzag_decode(const uint32_t* in, int32_t* out, size_t n);
void svbo_zigzag_delta_encode(const int32_t* in, uint32_t* out, size_t n, int32_t prev);
void svbo_zigzag_delta_decode(const uint32_t* in, int32_t* out, size_t n, int32_t prev);
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t next_u64() {  // splitmix64
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
#define CHECK(cond)                                                         \
  do {                                                                      \
    if (!(cond)) {                                                          \
      std::fprintf(stderr, "%s:%d: %s failed (n=%u kind=%d variant=%d delta=%d)\n", __FILE__, __LINE__, #cond, n, kind, variant, delta); \
      return 1;                                                             \
    }                                                                       \
  } while (0)

int main() {
  const uint32_t sizes[] = {0, 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 255, 256, 257, 1000, 4099, 70001};
  for (uint32_t n : sizes)
    for (int kind = 0; kind < 5; ++kind)
      for (int variant = 0; variant < 2; ++variant)
        for (int delta = 0; delta < 2; ++delta) {
          const uint32_t prev = delta ? 0xDEADBEEFu : 0u;
          // heap buffers of exactly n values: a vector of size 0 has no storage at all
          std::vector<uint32_t> in(n), back(n), back2(n);
          for (uint32_t i = 0; i < n; ++i) {
            const uint64_t r = next_u64();
            in[i] = kind == 0 ? (uint32_t)r >> (r >> 59)            // mixed widths
                  : kind == 1 ? (uint32_t)(r & 0xFF) * (uint32_t)((r >> 8) % 3 == 0)  // small and zero
                  : kind == 2 ? 0xFFFFFFFFu
                  : kind == 3 ? 0u
                              : (uint32_t)r;
          }
          const size_t cap = ((size_t)n + 3) / 4 + 4 * (size_t)n + 16;  // streamvbyte_max_compressedbytes
          std::vector<uint8_t> a(cap), b(cap);
          const size_t sa = svbcpu::encode_any(variant, delta, in.data(), n, a.data(), prev);
          const size_t sb = svbo_encode(variant, in.data(), n, b.data(), delta, prev);
          CHECK(sa == sb && std::memcmp(a.data(), b.data(), sa) == 0);
          if (!delta) CHECK((variant ? svbcpu::compressedbytes<1>(in.data(), n) : svbcpu::compressedbytes<0>(in.data(), n)) == sa);
          // the stream in a buffer that ends exactly STREAMVBYTE_PADDING bytes behind it
          std::vector<uint8_t> s(sa + 16);
          std::memcpy(s.data(), a.data(), sa);
          CHECK(svbcpu::decode_any(variant, delta, s.data(), back.data(), n, prev) == sa);
          CHECK(svbo_decode(variant, s.data(), back2.data(), n, delta, prev) == sa);
          CHECK(n == 0 || std::memcmp(back.data(), in.data(), 4 * (size_t)n) == 0);
          CHECK(n == 0 || std::memcmp(back2.data(), in.data(), 4 * (size_t)n) == 0);
          // validation reads the key block only: a buffer of exactly the stream's bytes
          std::vector<uint8_t> exact(a.begin(), a.begin() + sa);
          const bool v1 = variant ? svbcpu::validate<1>(exact.data(), sa, n) : svbcpu::validate<0>(exact.data(), sa, n);
          CHECK(v1 && svbo_validate(variant, exact.data(), sa, n));
          if (sa) {
            const bool v2 = variant ? svbcpu::validate<1>(exact.data(), sa - 1, n) : svbcpu::validate<0>(exact.data(), sa - 1, n);
            CHECK(v2 == (svbo_validate(variant, exact.data(), sa - 1, n) != 0));
          }
          // zigzag maps, all four
          std::vector<int32_t> si(n), so(n), so2(n);
          std::vector<uint32_t> z(n), z2(n);
          for (uint32_t i = 0; i < n; ++i) si[i] = (int32_t)in[i];
          svbcpu::zigzag(0, si.data(), z.data(), n, 0);
          svbo_zigzag_encode(si.data(), z2.data(), n);
          CHECK(n == 0 || std::memcmp(z.data(), z2.data(), 4 * (size_t)n) == 0);
          svbcpu::zigzag(1, z.data(), so.data(), n, 0);
          CHECK(n == 0 || std::memcmp(so.data(), si.data(), 4 * (size_t)n) == 0);
          svbcpu::zigzag(2, si.data(), z.data(), n, -77);
          svbo_zigzag_delta_encode(si.data(), z2.data(), n, -77);
          CHECK(n == 0 || std::memcmp(z.data(), z2.data(), 4 * (size_t)n) == 0);
          svbcpu::zigzag(3, z.data(), so.data(), n, -77);
          svbo_zigzag_delta_decode(z2.data(), so2.data(), n, -77);
          CHECK(n == 0 || (std::memcmp(so.data(), si.data(), 4 * (size_t)n) == 0 && std::memcmp(so2.data(), si.data(), 4 * (size_t)n) == 0));
        }
  std::puts("host codec and oracle: clean under ASan + UBSan");
  return 0;
}

// LDS instruction throughput on gfx950, one number per access shape: cycles of the CU's LDS pipe
// per wave instruction, with 16 waves per CU issuing back to back.  Answers what an encoder's byte
// assembly may cost: plain dword stores, sub-dword stores that share a dword between lanes, and
// atomic ORs with and without lanes on the same dword.
//   hipcc --offload-arch=gfx950 -O3 -o tools/native/lds_ops.bin tools/native/lds_ops.hip && tools/native/lds_ops.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

using u32 = uint32_t;
using u8 = uint8_t;
using u16 = uint16_t;

constexpr int kIters = 2000, kUnroll = 8;

// MODE: 0 b32 store, lane-consecutive dwords      1 atomic OR, lane-consecutive dwords
//       2 atomic OR, 2 lanes per dword            3 atomic OR, 4 lanes per dword
//       4 b8 store, 4 lanes per dword (consecutive bytes)   5 b8 store, 2 lanes per dword
//       6 b8 store, one lane per dword            7 b16 store, 2 lanes per dword
//       8 b16 store, one lane per dword           9 b128 store, lane-consecutive
//      10 atomic OR, 16 lanes per dword          11 b32 store, 2 lanes per dword (same address)
//      12 b8 store, consecutive bytes, every other lane masked off
template <int MODE>
__global__ __launch_bounds__(256) void k_lds(u32* out, u32 seed) {
  __shared__ __attribute__((aligned(16))) u32 buf[4][kUnroll * 64 * 4 + 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32* w = buf[wave];
  // explicit LDS pointers: through generic volatile pointers the stores become flat_store
  typedef __attribute__((address_space(3))) volatile u8* lds8_t;
  typedef __attribute__((address_space(3))) volatile u16* lds16_t;
  typedef __attribute__((address_space(3))) volatile u32* lds32_t;
  typedef u32 v4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) volatile v4* lds128_t;
  lds8_t b8 = (lds8_t)(__attribute__((address_space(3))) u32*)w;
  lds16_t b16 = (lds16_t)b8;
  lds32_t b32 = (lds32_t)b8;
  lds128_t b128 = (lds128_t)b8;
  u32 v = seed + lane;
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int base = u * 64;  // dwords
      if (MODE == 0) b32[base + lane] = v;
      if (MODE == 1) atomicOr(&w[base + lane], v);
      if (MODE == 2) atomicOr(&w[base + (lane >> 1)], v);
      if (MODE == 3) atomicOr(&w[base + (lane >> 2)], v);
      if (MODE == 4) b8[4 * base + lane] = (u8)v;
      if (MODE == 5) b8[4 * base + 2 * lane] = (u8)v;
      if (MODE == 6) b8[4 * base + 4 * lane] = (u8)v;
      if (MODE == 7) b16[2 * base + lane] = (u16)v;
      if (MODE == 8) b16[2 * base + 2 * lane] = (u16)v;
      if (MODE == 9) b128[base + lane] = v4{v, v, v, v};
      if (MODE == 10) atomicOr(&w[base + (lane >> 4)], v);
      if (MODE == 11) b32[base + (lane >> 1)] = v;
      if (MODE == 12) { if (lane & 1) b8[4 * base + lane] = (u8)v; }
      v += 0x01010101u;
    }
  }
  __syncthreads();
  if (seed == 0xFFFFFFFFu) out[blockIdx.x * 256 + threadIdx.x] = w[lane];
}

template <int MODE>
void run(const char* name, u32* d_out) {
  const int wgs = 256 * 4;  // 4 workgroups of 4 waves per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k_lds<MODE><<<wgs, 256>>>(d_out, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_lds<MODE><<<wgs, 256>>>(d_out, 2);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  int clk_khz = 0;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  const double instr_per_cu = (double)kIters * kUnroll * 16;  // 16 waves per CU
  const double ns_per_instr = ms * 1e6 / instr_per_cu;
  printf("%-58s %8.3f ms  %6.2f ns per wave instruction per CU = %5.1f cycles at %d MHz\n", name, ms, ns_per_instr,
         ns_per_instr * clk_khz * 1e-6, clk_khz / 1000);
}

int main() {
  u32* d_out;
  hipMalloc(&d_out, 256 * 4 * 256 * 4);
  run<0>("b32 store, lane-consecutive dwords", d_out);
  run<11>("b32 store, 2 lanes per dword (same address)", d_out);
  run<9>("b128 store, lane-consecutive", d_out);
  run<1>("atomic OR, lane-consecutive dwords", d_out);
  run<2>("atomic OR, 2 lanes per dword", d_out);
  run<3>("atomic OR, 4 lanes per dword", d_out);
  run<10>("atomic OR, 16 lanes per dword", d_out);
  run<4>("b8 store, consecutive bytes (4 lanes per dword)", d_out);
  run<12>("b8 store, consecutive bytes, odd lanes only", d_out);
  run<5>("b8 store, 2 lanes per dword", d_out);
  run<6>("b8 store, one lane per dword", d_out);
  run<7>("b16 store, 2 lanes per dword", d_out);
  run<8>("b16 store, one lane per dword", d_out);
  hipFree(d_out);
  return 0;
}

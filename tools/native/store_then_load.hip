// Does a wave's next load wait for its previous stores?  gfx9 waves count loads and stores in ONE in-order counter
// (vmcnt), so a wave that stores a tile and then loads the next can use the loaded data only once the stores
// before it have been acknowledged.  Each wave copies `iters` consecutive 4 KiB pieces (1 KiB per instruction):
//   mode 0  load piece i, wait, store piece i              (the stores of i precede the loads of i + 1)
//   mode 1  load piece i + 1, store piece i, wait for i + 1 (the loads precede the stores: vmcnt(4) suffices)
//   mode 2  as mode 0, but piece i of wave w is piece i * waves + w of the buffer (modes 0 and 1: each wave its own stretch)
// 7 waves per SIMD, as the batched decoder runs.
//   hipcc --offload-arch=gfx950 -O3 -o tools/native/store_then_load.bin tools/native/store_then_load.hip && tools/native/store_then_load.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool NT>
__global__ __launch_bounds__(256, 7) void k_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int iters) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const u32x4* s = src + wave * iters * 256 + lane;
  u32x4* d = dst + wave * iters * 256 + lane;
  auto ld = [&](const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; };
  auto st = [&](u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; };
  if (MODE == 2) {  // pieces dealt round-robin: all waves advance through memory together
    const size_t nw = (size_t)gridDim.x * 4;
    for (int i = 0; i < iters; ++i) {
      const u32x4* sp = src + ((size_t)i * nw + wave) * 256 + lane;
      u32x4* dp = dst + ((size_t)i * nw + wave) * 256 + lane;
      u32x4 r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = ld(sp + j * 64);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) { r[j].x ^= 1u; st(dp + j * 64, r[j]); }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (MODE == 0) {
    for (int i = 0; i < iters; ++i) {
      u32x4 r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = ld(s + i * 256 + j * 64);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) { r[j].x ^= 1u; st(d + i * 256 + j * 64, r[j]); }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    u32x4 r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = ld(s + j * 64);
    for (int i = 0; i < iters; ++i) {
      u32x4 n[4];
      const int in = i + 1 < iters ? i + 1 : i;
#pragma unroll
      for (int j = 0; j < 4; ++j) n[j] = ld(s + in * 256 + j * 64);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) { r[j].x ^= 1u; st(d + i * 256 + j * 64, r[j]); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = n[j];
    }
  }
}

// mode 3: a fresh workgroup per 4 * per_wave pieces (each wave per_wave consecutive pieces), dispatched in order
// mode 4: persistent workgroups that draw the next 4 consecutive pieces from an atomic ticket
template <int MODE, bool NT>
__global__ __launch_bounds__(256, 7) void k_copy2(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int per_wave,
                                                  unsigned total_groups, unsigned* ticket) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto ld = [&](const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; };
  auto st = [&](u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; };
  auto piece = [&](size_t idx) {
    const u32x4* sp = src + idx * 256 + lane;
    u32x4* dp = dst + idx * 256 + lane;
    u32x4 r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = ld(sp + j * 64);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j].x ^= 1u; st(dp + j * 64, r[j]); }
  };
  if (MODE == 3) {
    for (int i = 0; i < per_wave; ++i) piece(((size_t)blockIdx.x * 4 + wave) * per_wave + i);
  } else {
    __shared__ unsigned s_t;
    for (;;) {
      if (threadIdx.x == 0) s_t = atomicAdd(ticket, 1u);
      __syncthreads();
      const unsigned t = s_t;
      __syncthreads();
      if (t >= total_groups) break;
      piece((size_t)t * 4 + wave);
    }
  }
}

template <int MODE, bool NT>
static void run2(const char* name, const u32x4* src, u32x4* dst, size_t pieces, int per_wave, unsigned* ticket) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const unsigned groups = (unsigned)(pieces / 4);
  const unsigned grid = MODE == 3 ? (unsigned)(pieces / 4 / per_wave) : 256 * 7;
  const int reps = 10;
  float tot = 0;
  for (int r = 0; r < reps + 2; ++r) {
    (void)hipMemsetAsync(ticket, 0, 4, 0);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_copy2<MODE, NT>), dim3(grid), dim3(256), 0, 0, src, dst, per_wave, groups, ticket);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (r >= 2) tot += ms;
  }
  const float ms = tot / reps;
  printf("%-44s %8.4f ms  %7.1f GB/s (read + write)\n", name, ms, 2.0 * pieces * 4096.0 / ms / 1e6);
}

template <int MODE, bool NT>
static void run(const char* name, const u32x4* src, u32x4* dst, int blocks, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_copy<MODE, NT>), dim3(blocks), dim3(256), 0, 0, src, dst, iters);
  (void)hipEventRecord(e0);
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_copy<MODE, NT>), dim3(blocks), dim3(256), 0, 0, src, dst, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double bytes = 2.0 * blocks * 4.0 * iters * 4096.0;
  printf("%-44s %8.4f ms  %7.1f GB/s (read + write)\n", name, ms, bytes / ms / 1e6);
}

int main() {
  const int blocks = 256 * 7;  // 7 workgroups of 4 waves per CU
  for (int iters : {4, 16, 64}) {
    const size_t n16 = (size_t)blocks * 4 * iters * 256;
    u32x4 *src, *dst;
    if (hipMalloc(&src, n16 * 16) != hipSuccess || hipMalloc(&dst, n16 * 16) != hipSuccess) return 1;
    (void)hipMemset(src, 1, n16 * 16);
    printf("pieces per wave: %d (%.0f MiB each way)\n", iters, n16 * 16 / 1048576.0);
    run<0, false>("  store, then load the next", src, dst, blocks, iters);
    run<1, false>("  load the next, then store", src, dst, blocks, iters);
    run<2, false>("  pieces dealt round-robin over the waves", src, dst, blocks, iters);
    run<2, true>("  pieces dealt round-robin (non-temporal)", src, dst, blocks, iters);
    run<0, true>("  store, then load the next (non-temporal)", src, dst, blocks, iters);
    run<1, true>("  load the next, then store (non-temporal)", src, dst, blocks, iters);
    unsigned* ticket;
    if (hipMalloc(&ticket, 256) != hipSuccess) return 1;
    const size_t pieces = n16 / 256;
    run2<3, true>("  fresh workgroups, 1 piece per wave (nt)", src, dst, pieces, 1, ticket);
    run2<3, true>("  fresh workgroups, 4 pieces per wave (nt)", src, dst, pieces, 4, ticket);
    run2<3, false>("  fresh workgroups, 1 piece per wave", src, dst, pieces, 1, ticket);
    run2<4, true>("  persistent, pieces by ticket (nt)", src, dst, pieces, 1, ticket);
    run2<4, false>("  persistent, pieces by ticket", src, dst, pieces, 1, ticket);
    (void)hipFree(ticket);
    (void)hipFree(src); (void)hipFree(dst);
  }
  return 0;
}

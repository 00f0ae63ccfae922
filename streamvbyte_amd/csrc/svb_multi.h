// Batches of streams over the devices of one node (include/streamvbyte_multi.h): a contiguous range
// of whole streams per device through the batched kernels, per-stream sizes all-gathered with RCCL.
// The reference has no counterpart (single-threaded, one stream per call).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <string>

#include "svb_device.h"
#include "svb_lists.h"

namespace svb {

// Where each shard's streams start in the job's order; passed to the kernels by value.
struct ShardStarts {
  u32 n;           // shards
  u32 longest;     // entries per shard in the gathered buffer
  u64 start[65];   // start[r] = first stream of shard r, start[n] = all streams
};

// gathered: n shards x longest entries (shard r's sizes, padded) -> sizes_all in job order, and the
// sum of every 4096 consecutive entries to blk[blockIdx.x].
__global__ __launch_bounds__(1024) void k_multi_compact(const u64* __restrict__ gathered, ShardStarts S,
                                                        u64* __restrict__ sizes_all, u64* __restrict__ blk) {
  __shared__ u64 s_wave[16];
  const u64 n_all = S.start[S.n];
  const u64 g0 = ((u64)blockIdx.x * 1024 + threadIdx.x) * 4;
  u64 sum = 0;
  u32 r = 0;
  while (r + 1 < S.n && S.start[r + 1] <= g0) ++r;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const u64 g = g0 + k;
    if (g < n_all) {
      while (S.start[r + 1] <= g) ++r;  // empty shards are skipped; g < n_all ends the walk
      const u64 x = gathered[(u64)r * S.longest + (g - S.start[r])];
      sizes_all[g] = x;
      sum += x;
    }
  }
  const u64 w = wave_sum64(sum);
  if (lane_id() == 0) s_wave[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 t = 0;
    for (int i = 0; i < 16; ++i) t += s_wave[i];
    blk[blockIdx.x] = t;
  }
}
// offsets_all[g] = sizes_all[0] + .. + sizes_all[g-1], offsets_all[n_all] = the total.  Every
// workgroup sums the block sums before its own (a job of 2^20 streams has 256 of them).
__global__ __launch_bounds__(1024) void k_multi_offsets(const u64* __restrict__ sizes_all, u64 n_all,
                                                        const u64* __restrict__ blk, u64* __restrict__ offsets_all) {
  __shared__ u64 s_wave[16];
  __shared__ u64 s_base;
  u64 mine = 0;
  for (u32 b = threadIdx.x; b < blockIdx.x; b += 1024) mine += blk[b];
  const u64 wb = wave_sum64(mine);
  if (lane_id() == 0) s_wave[threadIdx.x >> 6] = wb;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 t = 0;
    for (int i = 0; i < 16; ++i) t += s_wave[i];
    s_base = t;
  }
  __syncthreads();
  const u64 g0 = ((u64)blockIdx.x * 1024 + threadIdx.x) * 4;
  u64 x[4], tot = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    x[k] = g0 + k < n_all ? sizes_all[g0 + k] : 0;
    tot += x[k];
  }
  const u64 incl = wave_scan_incl64(tot);
  __syncthreads();
  if (lane_id() == 63) s_wave[threadIdx.x >> 6] = incl;
  __syncthreads();
  u64 before = s_base + incl - tot;
  for (u32 w = 0; w < (threadIdx.x >> 6); ++w) before += s_wave[w];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (g0 + k < n_all) offsets_all[g0 + k] = before;
    before += x[k];
    if (g0 + k + 1 == n_all) offsets_all[n_all] = before;
  }
  if (n_all == 0 && blockIdx.x == 0 && threadIdx.x == 0) offsets_all[0] = 0;
}

// librccl.so.1, loaded on first use: the library itself stays light (RCCL is over half a gigabyte).
struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string& err) {
    if (handle) return true;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
      err = std::string("dlopen librccl.so.1: ") + dlerror();
      return false;
    }
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(h, name);
      if (!p) err = std::string("librccl.so.1 lacks ") + name;
      return p;
    };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    if (!CommInitAll || !CommDestroy || !AllGather || !GroupStart || !GroupEnd || !GetErrorString) {
      dlclose(h);
      return false;
    }
    handle = h;
    return true;
  }
};

}  // namespace svb

// C ABI of libstreamvbyte.so.2: the device-pointer entry points of
// include/streamvbyte_gpu.h and the reference-compatible host/device-pointer
// functions of include/streamvbyte.h, streamvbytedelta.h, streamvbyte_zigzag.h.
// Device pointers always run HIP kernels; host pointers do too from host_cpu_cutoff
// values up, smaller host calls run the library's own host codec (svb_cpu.h) on the
// calling thread.  The multi-device layer of include/streamvbyte_multi.h is at the end.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <pthread.h>
#include <string>
#include <vector>

#include "../../include/streamvbyte.h"
#include "../../include/streamvbyte_gpu.h"
#include "../../include/streamvbyte_multi.h"
#include "../../include/streamvbyte_zigzag.h"
#include "../../include/streamvbytedelta.h"
#include "svb_cpu.h"
#include "svb_decode.h"
#include "svb_delta.h"
#include "svb_encode.h"
#include "svb_misc.h"
#include "svb_batch.h"
#include "svb_lists.h"
#include "svb_intersect.h"
#include "svb_wide.h"
#include "svb_multi.h"

using namespace svb;

#define HIP_TRY(expr)                          \
  do {                                         \
    hipError_t e_ = (expr);                    \
    if (e_ != hipSuccess) {                    \
      g_last_hip_error = e_;                   \
      return SVBGPU_EHIP;                      \
    }                                          \
  } while (0)

namespace {

thread_local hipError_t g_last_hip_error = hipSuccess;

struct DevInfo {
  bool probed = false, ok = false;
  int cus = 0;
};
DevInfo g_dev[64];
std::mutex g_dev_mu;

const DevInfo& dev_info() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) {
    static DevInfo none;
    return none;
  }
  std::lock_guard<std::mutex> lk(g_dev_mu);
  DevInfo& I = g_dev[d];
  if (!I.probed) {
    hipDeviceProp_t p;
    I.probed = true;
    if (hipGetDeviceProperties(&p, d) == hipSuccess) {
      I.cus = p.multiProcessorCount;
      I.ok = std::strncmp(p.gcnArchName, "gfx950", 6) == 0;
    }
  }
  return I;
}

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Grid of a streaming kernel whose workgroups loop over `work_blocks` units with a grid stride.
// per_cu > 0: a persistent grid, at most per_cu workgroups per CU.  per_cu <= 0: a workgroup per unit,
// handed out in order by the hardware as earlier ones finish.  The kernels that read and write long
// arrays run 10-20 % faster that way (the workgroups in flight then cover one compact stretch of memory;
// strided over a persistent grid they drift apart); the ones that end in an atomic per workgroup
// (exact size, validation) keep the small grid.
inline unsigned stream_grid(unsigned long long work_blocks, int per_cu) {
  unsigned long long cap = per_cu > 0 ? (unsigned long long)dev_info().cus * per_cu : 0x7FFFFFFFull;
  if (cap == 0) cap = 2048;
  return (unsigned)(work_blocks < cap ? (work_blocks ? work_blocks : 1) : cap);
}

// Runs `stmt` with the constexpr int `NTV` set to `nt` if `nt` is one of the listed values, else to 0:
// one kernel instantiation per listed policy.
#define SVB_NT_CASE(v, NTV, ...) case v: { constexpr int NTV = v; __VA_ARGS__; } break;
#define SVB_NT_SWITCH_023(nt, NTV, ...) \
  switch (nt) { SVB_NT_CASE(2, NTV, __VA_ARGS__) SVB_NT_CASE(3, NTV, __VA_ARGS__) default: { constexpr int NTV = 0; __VA_ARGS__; } }
#define SVB_NT_SWITCH_013(nt, NTV, ...) \
  switch (nt) { SVB_NT_CASE(1, NTV, __VA_ARGS__) SVB_NT_CASE(3, NTV, __VA_ARGS__) default: { constexpr int NTV = 0; __VA_ARGS__; } }
#define SVB_NT_SWITCH_03(nt, NTV, ...) \
  switch (nt) { SVB_NT_CASE(3, NTV, __VA_ARGS__) default: { constexpr int NTV = 0; __VA_ARGS__; } }
#define SVB_NT_SWITCH_01(nt, NTV, ...) \
  switch (nt) { SVB_NT_CASE(1, NTV, __VA_ARGS__) default: { constexpr int NTV = 0; __VA_ARGS__; } }

// Tuning knobs (svbgpu_set_option): process-wide atomics, read ONCE at the top of
// an API call into a plain struct that travels down to the launches, so a call
// never runs on half of a change made by another thread and what a call checks
// afterwards matches what it launched.
struct Options {
  int scan_waves;        // waves per workgroup of k_running_sum (4, 8 or 16) and of k_delta_decode
  int decode_per_cu;     // decoders: workgroups per CU of a persistent grid, 0 = a workgroup per 4 wave tiles
  int decode_shortcut;   // plain decoder: shortcut for wave tiles of all-zero keys
  int delta_lookback;    // 1: single-pass look-back differential decoder, 0: two streaming passes
  int encode_waves;      // waves per workgroup of k_encode when a wave carries 4 wave tiles (4, 8 or 16)
  int encode64_waves;    // waves (= wave tiles) per workgroup of k_encode64_lb (4, 8 or 16)
  int encode_fresh;      // k_encode as a workgroup per tile (1) or as persistent workgroups looping over tickets (0)
  int scan_fresh;        // the same for k_running_sum (measured: 16 waves 0.72 -> 0.67 of peak, 8 waves 0.73; default 0)
  int encode_qw;         // wave tiles per wave in k_encode (1, 2 or 4)
  int batch_lists_per_wave;  // row/wave batched kernels: lists per wave of a launch, 0 = one grid that fills the device
  int batch_chunk;       // batched calls: -1 chunked kernels, streams per wavefront chosen from the batch size;
                         // 1..64 chunked kernels with that many; 0 the row + wave kernel pair
  unsigned debug;        // timing-only ablations (wrong results!)
  int zigzag_piece;      // values per device scan of the host-pointer zigzag_delta_decode
  int encode_pipeline;   // 0: k_encode, 1: k_encode_pipe (two half-size tiles in flight per workgroup) for plain input, 2: always, 3: by size (choose_encoder)
  int encode_ticket_early;  // k_encode_pipe: 1 requests a tile's ticket one store phase before the tile is loaded
  int host_cpu_cutoff;   // host-pointer calls on fewer values than this run on the host codec (svb_cpu.h)
  int host_serialize;    // host-pointer calls: one at a time per device inside the runtime
  int host_stage_pageable;  // host-pointer calls: pageable caller memory moves through page-locked chunks
  int nt_threshold_mib;  // a call's loads / stores are non-temporal when it reads / writes at least this many MiB
                         // (-1: never, 0: always)
  int nt_force;          // -1: by size; 0..3: this policy whatever the size (bit 0 loads, bit 1 stores)
  unsigned lb_tag2 = 0;  // not a knob: the look-back tag of the running call (LbWords::tag2), set by prepare_lookback
};
std::atomic<int> o_scan_waves{16}, o_decode_per_cu{0}, o_decode_shortcut{1}, o_delta_lookback{1}, o_encode_waves{8}, o_encode64_waves{8}, o_encode_fresh{1}, o_scan_fresh{0},
    o_encode_qw{4}, o_encode_pipeline{3}, o_encode_ticket_early{-1}, o_batch_lists_per_wave{8}, o_batch_chunk{-1}, o_debug{0}, o_zigzag_piece{0x7FFFFFFF}, o_host_cpu_cutoff{1 << 16}, o_host_serialize{1},
    o_host_stage_pageable{1}, o_nt_threshold_mib{128}, o_nt_force{-1}, o_host_pointers_only{0};
inline Options options() {
  Options o;
  const auto rl = std::memory_order_relaxed;
  o.scan_waves = o_scan_waves.load(rl);
  o.decode_per_cu = o_decode_per_cu.load(rl);
  o.decode_shortcut = o_decode_shortcut.load(rl);
  o.delta_lookback = o_delta_lookback.load(rl);
  o.encode_waves = o_encode_waves.load(rl);
  o.encode64_waves = o_encode64_waves.load(rl);
  o.encode_fresh = o_encode_fresh.load(rl);
  o.scan_fresh = o_scan_fresh.load(rl);
  o.encode_qw = o_encode_qw.load(rl);
  o.encode_pipeline = o_encode_pipeline.load(rl);
  o.encode_ticket_early = o_encode_ticket_early.load(rl);
  o.batch_lists_per_wave = o_batch_lists_per_wave.load(rl);
  o.batch_chunk = o_batch_chunk.load(rl);
  o.debug = (unsigned)o_debug.load(rl);
  o.zigzag_piece = o_zigzag_piece.load(rl);
  o.host_cpu_cutoff = o_host_cpu_cutoff.load(rl);
  o.host_serialize = o_host_serialize.load(rl);
  o.host_stage_pageable = o_host_stage_pageable.load(rl);
  o.nt_threshold_mib = o_nt_threshold_mib.load(rl);
  o.nt_force = o_nt_force.load(rl);
  return o;
}

// Non-temporal policy of a call that reads `load_bytes` and writes `store_bytes` once each (lower bounds
// where the exact size is not known before the call): see ld_stream16 / st_stream16.
inline int nt_policy(const Options& o, unsigned long long load_bytes, unsigned long long store_bytes) {
  if (o.nt_force >= 0) return o.nt_force;
  if (o.nt_threshold_mib < 0) return 0;
  // Stores: an output of half the Infinity Cache (256 MiB) or more cannot wait there for its consumer.
  // Loads: an input that size is still served from the cache when a call repeats on it and little else
  // is written (exact size at 2^26 values: 0.70 against 0.57 of peak), so they also ask for a footprint
  // beyond the cache.  Measured with tools/nt_sweep.py at 2^24 ... 2^28 values.
  const unsigned long long t = (unsigned long long)o.nt_threshold_mib << 20;
  const bool stores = store_bytes >= t;
  const bool loads = load_bytes >= t && 2 * (load_bytes + store_bytes) >= 5 * t;
  return (loads ? kNtLoads : 0) | (stores ? kNtStores : 0);
}
// Batched calls choose on the device (batch_nt_policy): the host hands over the forced policy or -1 and
// the threshold in bytes.
inline int batch_nt_force(const Options& o) { return o.nt_force >= 0 ? o.nt_force : (o.nt_threshold_mib < 0 ? 0 : -1); }
inline unsigned long long batch_nt_bytes(const Options& o) {
  return o.nt_threshold_mib < 0 ? 0 : (unsigned long long)o.nt_threshold_mib << 20;
}
// Least size of a stream of `count` values: the keys plus one byte per value (1234) or nothing (0124).
inline unsigned long long min_stream_bytes(int v, unsigned long long count) {
  return (count + 3) / 4 + (v == V1234 ? count : 0);
}
// The encoders' form of the rule: their input also loads non-temporally once it is as large as the cache
// itself (2 x the threshold), whatever they write -- an encoder gains nothing from cached input (it is not
// bound by its reads), and 0124 streams are too short to lift the footprint over the general rule's mark
// (2^26 gradients, 256 MiB in and 20 MiB out: 0.422 -> 0.433 of peak; at half that size it loses 3 %).
inline int nt_policy_encode(const Options& o, int v, unsigned long long count) {
  int nt = nt_policy(o, 4ull * count, min_stream_bytes(v, count));
  if (o.nt_force < 0 && o.nt_threshold_mib >= 0 && 4ull * count >= ((unsigned long long)o.nt_threshold_mib << 21))
    nt |= kNtLoads;
  return nt;
}

inline WsHeader* ws_hdr(void* ws) { return reinterpret_cast<WsHeader*>(ws); }
// The header as a call with look-back tag `tag2` gets it: 8 bytes up for odd tags, whose tickets are
// WsHeader::ticket1 (draw_ticket).  Only for kernels that reach the header through draw_ticket, the
// look-back functions and nothing else.
inline WsHeader* ws_hdr_slot(void* ws, unsigned tag2) {
  return reinterpret_cast<WsHeader*>(reinterpret_cast<char*>(ws) + ((tag2 >> 3) & 1u) * 8);
}
inline LbWords lb_words(void* ws, const WsLayout& L, unsigned tag2) {
  return LbWords{reinterpret_cast<u64*>(reinterpret_cast<char*>(ws) + L.off_status), (u32)((L.off_lb_group - L.off_status) / 8), tag2};
}
template <class T>
inline T* ws_at(void* ws, size_t off) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(ws) + off);
}

// Device fill by a kernel (see k_fill_u32): ptr 4-byte aligned, bytes a multiple of 4.
int fill_u32(void* ptr, uint32_t value, size_t bytes, hipStream_t st) {
  const unsigned long long n = bytes / 4;
  const unsigned grid = (unsigned)std::min<unsigned long long>((n + kBlock - 1) / kBlock, 1024ull);
  hipLaunchKernelGGL(k_fill_u32, dim3(grid ? grid : 1), dim3(kBlock), 0, st, static_cast<u32*>(ptr), value, (u64)n);
  HIP_TRY(hipGetLastError());
  return SVBGPU_OK;
}

// Workspaces whose look-back words were zeroed once by svbgpu_workspace_init: calls on them skip the
// clearing launch and mark the words they write with a tag of their own, counted here per workspace.
struct WsReg { u32 tag = 0; };
std::mutex g_ws_mu;
std::vector<std::pair<void*, WsReg>> g_ws_reg;  // a handful of entries: a linear search beats a hash
WsReg* ws_find(void* ws) {
  for (auto& e : g_ws_reg)
    if (e.first == ws) return &e.second;
  return nullptr;
}
bool ws_registered(void* ws) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  return ws_find(ws) != nullptr;
}
// The tag (shifted into place, LbWords::tag2) of a call that may skip the clearing launch, or 0: the
// workspace is not initialised, the stream is being captured into a graph (a replay would reuse the
// tag, so captured calls clear), or the tags have run out (this call clears and the count restarts).
u32 next_lookback_tag(void* ws, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  WsReg* r = ws_find(ws);
  if (!r) return 0;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (cs != hipStreamCaptureStatusNone) return 0;
  if (++r->tag >= kLbTags) { r->tag = 0; return 0; }
  return r->tag << 3;
}

int check_ws(uint32_t count, void* ws, size_t ws_bytes) {
  if (!ws || ((uintptr_t)ws & 255u)) return SVBGPU_EWORKSPACE;
  if (ws_bytes < ws_layout(count).total) return SVBGPU_EWORKSPACE;
  return SVBGPU_OK;
}

// header + look-back words start every call at zero
// Stress build only: "debug_ablate" bit 64 arms lb_stress_delay for the kernels that follow.
int arm_lookback_stress(void* ws, hipStream_t st) {
#ifdef SVB_LB_STRESS
  if (options().debug & 64u) return fill_u32(reinterpret_cast<u32*>(ws) + kLbStressWord, 1u, 4, st);
#endif
  (void)ws; (void)st;
  return SVBGPU_OK;
}
int clear_lookback(uint32_t count, void* ws, hipStream_t st) {
  const WsLayout L = ws_layout(count);
  const int rc = fill_u32(ws, 0u, L.off_mega_sum, st);
  return rc ? rc : arm_lookback_stress(ws, st);
}

// ---- offset directory -------------------------------------------------------
// clear_lb: the launch also zeroes the header and the look-back words (what clear_lookback does), for a
// differential decode that follows on the same stream.
int build_directory(int variant, const uint8_t* d_in, uint32_t count, void* ws, hipStream_t st, bool clear_lb = false) {
  const WsLayout L = ws_layout(count);
  u32* wt_rel = ws_at<u32>(ws, L.off_wt_rel);
  u32* mega_sum = ws_at<u32>(ws, L.off_mega_sum);
  const u32 spm = L.wt_per_mega / kSuperWt;
  u32* clr = clear_lb ? ws_at<u32>(ws, 0) : nullptr;
  const u32 clr_n = clear_lb ? (u32)(L.off_mega_sum / 4) : 0u;
  if (variant == SVBGPU_1234)
    hipLaunchKernelGGL(k_key_sums<V1234>, dim3(L.n_mega), dim3(1024), 0, st, d_in, count, wt_rel, mega_sum, spm, clr, clr_n);
  else if (variant == SVBGPU_1248)
    hipLaunchKernelGGL(k_key_sums<V1248>, dim3(L.n_mega), dim3(1024), 0, st, d_in, count, wt_rel, mega_sum, spm, clr, clr_n);
  else
    hipLaunchKernelGGL(k_key_sums<V0124>, dim3(L.n_mega), dim3(1024), 0, st, d_in, count, wt_rel, mega_sum, spm, clr, clr_n);
  HIP_TRY(hipGetLastError());
  return clear_lb ? arm_lookback_stress(ws, st) : SVBGPU_OK;
}
// What a look-back call does before its kernels: nothing but picking a tag on an initialised
// workspace, else the clearing launch.
int prepare_lookback(uint32_t count, void* ws, hipStream_t st, unsigned* tag2) {
  *tag2 = next_lookback_tag(ws, st);
  if (*tag2) return arm_lookback_stress(ws, st);
  return clear_lookback(count, ws, st);
}
// ... and after them.  A call that cleared (tag 0) on an INITIALISED workspace -- recorded into a graph,
// or the one call in 2^27 on which the tags ran out -- leaves its ticket counter used; a graph replay
// happens without this code knowing, so such a call ends with a launch that puts both counters back
// to zero, which is what the next tagged call counts on.
int finish_lookback(void* ws, hipStream_t st, unsigned tag2) {
  if (tag2 || !ws_registered(ws)) return SVBGPU_OK;
  return fill_u32(&ws_hdr(ws)->ticket, 0u, 12, st);
}
inline Dir ws_dir(void* ws, const WsLayout& L) {
  return Dir{ws_at<u32>(ws, L.off_mega_sum), ws_at<u32>(ws, L.off_wt_rel), L.wt_per_mega};
}

template <int V, bool ZZ>
void launch_decode(const Options& o, const uint8_t* d_in, uint32_t* d_out, uint32_t count, void* ws,
                   uint64_t* d_consumed, hipStream_t st) {
  const WsLayout L = ws_layout(count);
  const unsigned grid = stream_grid((L.n_wt + kWaves - 1) / kWaves, o.decode_per_cu);
  if (o.decode_shortcut)
    SVB_NT_SWITCH_023(nt_policy(o, min_stream_bytes(V, count), 4ull * count), NTV,
                      hipLaunchKernelGGL((k_decode<V, ZZ, true, 0, NTV>), dim3(grid), dim3(kBlock), 0, st, d_in, d_out,
                                         count, ws_dir(ws, L), d_consumed, 0u))
  else
    hipLaunchKernelGGL((k_decode<V, ZZ, false>), dim3(grid), dim3(kBlock), 0, st, d_in, d_out, count,
                       ws_dir(ws, L), d_consumed, 0u);
}
template <int V, bool ZZ>
void launch_delta_decode(const Options& o, const uint8_t* d_in, uint32_t* d_out, uint32_t count, uint32_t prev,
                         void* ws, size_t ws_bytes, uint64_t* d_consumed, hipStream_t st) {
  const WsLayout L = ws_layout(count);
  // diagnostic stamps (8 words per tile) live behind the regular workspace when the caller
  // over-allocated it and asked for them
  u64* stamps = (o.debug & 16u) && ws_bytes >= L.total + 64ull * (L.n_wt / 16 + 2) ? ws_at<u64>(ws, L.total) : nullptr;
  if (stamps && V == V1234 && !ZZ && o.scan_waves >= 8) {  // diagnostic instantiations with the stamps compiled in
    const LbWords lbw = lb_words(ws, L, 0u);
    if (o.scan_waves == 16)
      hipLaunchKernelGGL((k_delta_decode<V1234, false, 16, 3, true>), dim3((unsigned)((L.n_wt + 63) / 64)), dim3(1024), 0, st,
                         d_in, d_out, count, prev, ws_dir(ws, L), ws_hdr(ws), lbw, d_consumed, stamps);
    else
      hipLaunchKernelGGL((k_delta_decode<V1234, false, 8, 3, true>), dim3((unsigned)((L.n_wt + 31) / 32)), dim3(512), 0, st,
                         d_in, d_out, count, prev, ws_dir(ws, L), ws_hdr(ws), lbw, d_consumed, stamps);
  } else if (o.scan_waves == 16 && count >= (3u << 22))  // shorter streams do better with more, smaller tiles (2^23 values: 8 waves 7-8 % ahead, 2^24: 16 waves 12-18 %)
    SVB_NT_SWITCH_023(nt_policy(o, min_stream_bytes(V, count), 4ull * count), NTV,
                      hipLaunchKernelGGL((k_delta_decode<V, ZZ, 16, NTV>), dim3((unsigned)((L.n_wt + 63) / 64)), dim3(1024),
                                         0, st, d_in, d_out, count, prev, ws_dir(ws, L), ws_hdr(ws),
                                         lb_words(ws, L, 0u),
                                         d_consumed, (u64*)nullptr))
  else if (o.scan_waves >= 8)  // 8 waves per workgroup: 32 wave tiles per look-back tile
    SVB_NT_SWITCH_023(nt_policy(o, min_stream_bytes(V, count), 4ull * count), NTV,
                      hipLaunchKernelGGL((k_delta_decode<V, ZZ, 8, NTV>), dim3((unsigned)((L.n_wt + 31) / 32)), dim3(512), 0,
                                         st, d_in, d_out, count, prev, ws_dir(ws, L), ws_hdr(ws),
                                         lb_words(ws, L, 0u),
                                         d_consumed, (u64*)nullptr))
  else
  hipLaunchKernelGGL((k_delta_decode<V, ZZ>), dim3((unsigned)L.n_tiles), dim3(kBlock), 0, st, d_in,
                     d_out, count, prev, ws_dir(ws, L), ws_hdr(ws),
                     lb_words(ws, L, 0u), d_consumed, (u64*)nullptr);
}

template <int V, bool ZZ>
void launch_delta_decode2(const Options& o, const uint8_t* d_in, uint32_t* d_out, uint32_t count, uint32_t prev,
                          void* ws, uint64_t* d_consumed, hipStream_t st) {
  const WsLayout L = ws_layout(count);
  const unsigned grid = stream_grid((L.n_wt + kWaves - 1) / kWaves, o.decode_per_cu);
  const Dir dir = ws_dir(ws, L);
  u32* vsum = ws_at<u32>(ws, L.off_wt_vsum);
  u32* blk = ws_at<u32>(ws, L.off_blk);
  const unsigned nblk = (unsigned)((L.n_wt + 1023) / 1024);
  hipLaunchKernelGGL((k_value_sums<V, ZZ>), dim3(grid), dim3(kBlock), 0, st, d_in, count, dir, vsum);
  hipLaunchKernelGGL(k_scan_u32_blocks, dim3(nblk), dim3(kBlock), 0, st, vsum, (u32)L.n_wt, blk);
  hipLaunchKernelGGL(k_scan_u32_top, dim3(1), dim3(1024), 0, st, blk, nblk, prev);
  hipLaunchKernelGGL((k_delta_decode2<V, ZZ>), dim3(grid), dim3(kBlock), 0, st, d_in, d_out, count, dir, vsum, blk,
                     d_consumed);
}

// decode with the directory already in the workspace
// lb_cleared: the header and look-back words were zeroed by the directory launch (build_directory(.., true))
int run_decode(const Options& o, int variant, unsigned flags, const uint8_t* d_in, uint32_t* d_out, uint32_t count,
               uint32_t prev, void* ws, uint64_t* d_consumed, hipStream_t st, size_t ws_bytes = 0,
               bool lb_cleared = false) {
  const bool zz = flags & SVBGPU_ZIGZAG, delta = flags & SVBGPU_DELTA;
  const bool v0 = variant == SVBGPU_1234;
  if (!delta) {
    if (v0) { if (zz) launch_decode<V1234, true>(o, d_in, d_out, count, ws, d_consumed, st);
              else    launch_decode<V1234, false>(o, d_in, d_out, count, ws, d_consumed, st); }
    else    { if (zz) launch_decode<V0124, true>(o, d_in, d_out, count, ws, d_consumed, st);
              else    launch_decode<V0124, false>(o, d_in, d_out, count, ws, d_consumed, st); }
  } else if (o.delta_lookback) {
    if (!lb_cleared) {
      int rc = clear_lookback(count, ws, st);
      if (rc) return rc;
    }
    if (v0) { if (zz) launch_delta_decode<V1234, true>(o, d_in, d_out, count, prev, ws, ws_bytes, d_consumed, st);
              else    launch_delta_decode<V1234, false>(o, d_in, d_out, count, prev, ws, ws_bytes, d_consumed, st); }
    else    { if (zz) launch_delta_decode<V0124, true>(o, d_in, d_out, count, prev, ws, ws_bytes, d_consumed, st);
              else    launch_delta_decode<V0124, false>(o, d_in, d_out, count, prev, ws, ws_bytes, d_consumed, st); }
  } else {
    if (v0) { if (zz) launch_delta_decode2<V1234, true>(o, d_in, d_out, count, prev, ws, d_consumed, st);
              else    launch_delta_decode2<V1234, false>(o, d_in, d_out, count, prev, ws, d_consumed, st); }
    else    { if (zz) launch_delta_decode2<V0124, true>(o, d_in, d_out, count, prev, ws, d_consumed, st);
              else    launch_delta_decode2<V0124, false>(o, d_in, d_out, count, prev, ws, d_consumed, st); }
  }
  HIP_TRY(hipGetLastError());
  return SVBGPU_OK;
}

// Batched kernels: a wave takes about this many lists, so that a launch has many more workgroups than fit
// the device and the hardware hands them out as earlier ones finish.  Lists differ in length by orders of
// magnitude; lists strided over a grid that just fills the device left some waves with 1.3x the average
// work.  0 = such a persistent grid.  See svbgpu_set_option.
// Streams of the batch per wavefront in the long-stream blocks of the chunked batch kernels: `most` when
// the batch is large enough to give every CU some 32 wavefronts that way (measured best on a million
// posting lists: 16 for the decoder -- 32 before its key loads ran ahead across streams --, 8 for the encoder),
// fewer for smaller batches so that the device still fills.
inline unsigned batch_chunk(const Options& o, uint32_t nstreams, unsigned most) {
  if (o.batch_chunk >= 0) return (unsigned)o.batch_chunk;
  const unsigned long long want_waves = (unsigned long long)(dev_info().cus ? dev_info().cus : 256) * 32;
  const unsigned long long k = nstreams / want_waves;
  return (unsigned)(k < 1 ? 1 : (k > most ? most : k));
}
inline unsigned batch_grid(const Options& o, uint32_t nstreams) {
  const unsigned long long blocks = ((unsigned long long)nstreams + kWaves - 1) / kWaves;
  const unsigned fill = stream_grid(blocks, 8);
  if (o.batch_lists_per_wave <= 0) return fill;
  return (unsigned)std::max<unsigned long long>(fill, (blocks + o.batch_lists_per_wave - 1) / o.batch_lists_per_wave);
}

template <int V, bool D, bool Z, int QW, bool QUANT = false, int NW = kWaves>
void launch_encode_qw(const Options& o, const uint32_t* d_in, uint32_t count, uint8_t* d_out, uint32_t prev, void* ws,
                      size_t ws_bytes, uint64_t* d_written, hipStream_t st) {
  const WsLayout L = ws_layout(count);
  constexpr unsigned long long kTileInts = (unsigned long long)NW * QW * kWtInts;
  const unsigned full_tiles = (unsigned)(count / kTileInts);
  const bool ragged = (unsigned long long)full_tiles * kTileInts < count;
  WsHeader* h = ws_hdr_slot(ws, o.lb_tag2);
  const LbWords status = lb_words(ws, L, o.lb_tag2);
  // diagnostic stamps live behind the regular workspace when the caller over-allocated it
  u64* stamps = (o.debug & 16u) && ws_bytes >= L.total + 40ull * (full_tiles + 2)
                    ? ws_at<u64>(ws, L.total) : nullptr;
  const unsigned grid = stream_grid(full_tiles, NW == 16 ? 1 : NW == 8 ? (QW <= 2 ? 3 : 2) : (QW > 2 || V != V1234 ? 4 : 5));
  if (full_tiles && QW == 4 && o.encode_fresh)  // a workgroup per tile
    SVB_NT_SWITCH_013(nt_policy_encode(o, V, count), NTV,
                      hipLaunchKernelGGL((k_encode<V, D, Z, QW, true, QUANT, NW, (QW == 4 ? NTV : 0)>), dim3(full_tiles), dim3(NW * 64), 0,
                                         st, d_in, count, full_tiles, d_out, prev, h, status, d_written, o.debug | 8u, stamps))
  else if (full_tiles && QW == 4)  // the shapes in regular use take the non-temporal policy of the call
    SVB_NT_SWITCH_013(nt_policy_encode(o, V, count), NTV,
                      hipLaunchKernelGGL((k_encode<V, D, Z, QW, true, QUANT, NW, (QW == 4 ? NTV : 0)>), dim3(grid), dim3(NW * 64), 0,
                                         st, d_in, count, full_tiles, d_out, prev, h, status, d_written, o.debug, stamps))
  else if (full_tiles)  // the smaller tiles: experiments only (option encode_tile)
    hipLaunchKernelGGL((k_encode<V, D, Z, QW, true, QUANT, NW>), dim3(o.encode_fresh ? full_tiles : grid), dim3(NW * 64),
                       0, st, d_in, count, full_tiles, d_out, prev, h, status, d_written, o.debug | (o.encode_fresh ? 8u : 0u), stamps);
  if (ragged)  // the one tile that runs past the end of the stream: same look-back words
    hipLaunchKernelGGL((k_encode<V, D, Z, QW, false, QUANT, NW>), dim3(1), dim3(NW * 64), 0, st, d_in, count, full_tiles,
                       d_out, prev, h, status, d_written, o.debug, stamps);
}
// The pipelined encoder: full tiles of NW waves x QW wave tiles by k_encode_pipe, the ragged last
// tile by k_encode<.., FULL = false> of the same tile size on the same ticket and look-back words.
template <int V, bool D, bool Z, int QW, bool QUANT = false, int NW = 8>
void launch_encode_pipe(const Options& o, const uint32_t* d_in, uint32_t count, uint8_t* d_out, uint32_t prev, void* ws,
                        size_t ws_bytes, uint64_t* d_written, hipStream_t st) {
  const WsLayout L = ws_layout(count);
  constexpr unsigned long long kTileInts = (unsigned long long)NW * QW * kWtInts;
  const unsigned full_tiles = (unsigned)(count / kTileInts);
  const bool ragged = (unsigned long long)full_tiles * kTileInts < count;
  WsHeader* h = ws_hdr_slot(ws, o.lb_tag2);
  const LbWords status = lb_words(ws, L, o.lb_tag2);
  // diagnostic stamps live behind the regular workspace when the caller over-allocated it
  u64* stamps = (o.debug & 16u) && ws_bytes >= L.total + 40ull * (full_tiles + 2) ? ws_at<u64>(ws, L.total) : nullptr;
  // the ticket a phase early pays for 0124 (short store phases), not for 1234
  const u32 early = o.encode_ticket_early < 0 ? (V == V0124 ? 1u : 0u) : (u32)o.encode_ticket_early;
  if (full_tiles && stamps && !QUANT)  // diagnostic instantiation with the stamps compiled in
    hipLaunchKernelGGL((k_encode_pipe<V, D, Z, QW, false, NW, 0, true>), dim3(stream_grid(full_tiles, 2)), dim3(NW * 64), 0,
                       st, d_in, count, full_tiles, d_out, prev, h, status, d_written, stamps, early);
  else if (full_tiles)
    SVB_NT_SWITCH_013(nt_policy_encode(o, V, count), NTV,
                      hipLaunchKernelGGL((k_encode_pipe<V, D, Z, QW, QUANT, NW, NTV>), dim3(stream_grid(full_tiles, 2)),
                                         dim3(NW * 64), 0, st, d_in, count, full_tiles, d_out, prev, h, status, d_written,
                                         (u64*)nullptr, early))
  if (ragged)
    hipLaunchKernelGGL((k_encode<V, D, Z, QW, false, QUANT, NW>), dim3(1), dim3(NW * 64), 0, st, d_in, count, full_tiles,
                       d_out, prev, h, status, d_written, 0u, (u64*)nullptr);
}
// Which encoder for a call.  encode_pipeline 0 / 1 / 2 force k_encode / k_encode_pipe for plain input /
// k_encode_pipe always; 3 (the default) picks by size, from the steady-state sweep of
// tools/encode_shape_sweep.py: short streams do best pipelined (half-size tiles: more of them to spread),
// streams of 2^23 values with the tile-at-a-time kernel, a workgroup per tile; around 2^24 the 16-wave tile
// (one round of tiles over the device); plain input up to 2^27 pipelined again; everything longer, and
// differential input from 2^25 up, tile at a time with 8 waves and a workgroup per tile (2^28 values of mixed
// widths: 0.667 of the HBM peak against 0.639 pipelined; dense postings 0.662 against 0.513).
struct EncoderChoice { bool pipe; int waves; };
inline EncoderChoice choose_encoder(const Options& o, bool delta, uint32_t count) {
  if (o.encode_pipeline != 3) return {o.encode_pipeline == 2 || (o.encode_pipeline == 1 && !delta), o.encode_waves};
  if (count < (1u << 23)) return {true, 8};
  if (count < (1u << 24)) return {false, 8};
  if (count < (1u << 25)) return {false, 16};
  if (!delta && count < (1u << 27)) return {true, 8};
  return {false, 8};
}
template <int V, bool D, bool Z>
void launch_encode(const Options& o, const uint32_t* d_in, uint32_t count, uint8_t* d_out, uint32_t prev, void* ws,
                   size_t ws_bytes, uint64_t* d_written, hipStream_t st) {
  const EncoderChoice c = choose_encoder(o, D, count);
  if (c.pipe && !(o.debug & ~(16u | 64u))) launch_encode_pipe<V, D, Z, 2>(o, d_in, count, d_out, prev, ws, ws_bytes, d_written, st);
  else if (o.encode_qw == 1) launch_encode_qw<V, D, Z, 1>(o, d_in, count, d_out, prev, ws, ws_bytes, d_written, st);
  else if (o.encode_qw == 2 && c.waves == 8) launch_encode_qw<V, D, Z, 2, false, 8>(o, d_in, count, d_out, prev, ws, ws_bytes, d_written, st);
  else if (o.encode_qw == 2) launch_encode_qw<V, D, Z, 2>(o, d_in, count, d_out, prev, ws, ws_bytes, d_written, st);
  else if (c.waves == 16) launch_encode_qw<V, D, Z, 4, false, 16>(o, d_in, count, d_out, prev, ws, ws_bytes, d_written, st);
  else if (c.waves == 8 && (V == V1234 || o.encode_fresh))  // persistent workgroups: 0124 on small values is faster with the smaller tile
    launch_encode_qw<V, D, Z, 4, false, 8>(o, d_in, count, d_out, prev, ws, ws_bytes, d_written, st);
  else launch_encode_qw<V, D, Z, 4>(o, d_in, count, d_out, prev, ws, ws_bytes, d_written, st);
}
template <int V, bool D, bool Z>
void launch_size(const uint32_t* d_in, uint32_t count, uint32_t prev, uint64_t* d_bytes,
                 hipStream_t st) {
  const unsigned grid = stream_grid(((unsigned long long)count / 4 + kSizeChunk - 1) / kSizeChunk, 8);
  SVB_NT_SWITCH_01(nt_policy(options(), 4ull * count, 0), NTV,
                   hipLaunchKernelGGL((k_size<V, D, Z, NTV>), dim3(grid), dim3(kBlock), 0, st, d_in, count, prev, d_bytes))
}

template <int V, bool D, bool Z>
void launch_size_batch(const uint32_t* d_values, const uint64_t* d_value_off, const uint32_t* d_counts,
                       const uint32_t* d_prev, uint32_t nstreams, uint64_t* d_sizes, hipStream_t st) {
  hipLaunchKernelGGL((k_size_batch<V, D, Z>), dim3((nstreams + kWaves - 1) / kWaves), dim3(kBlock), 0, st, d_values,
                     d_value_off, d_counts, d_prev, nstreams, d_sizes);
}
#define DISPATCH_VDZ(fn, variant, flags, ...)                                              \
  do {                                                                                     \
    const bool d_ = (flags)&SVBGPU_DELTA, z_ = (flags)&SVBGPU_ZIGZAG;                      \
    if ((variant) == SVBGPU_1234) {                                                        \
      if (d_) { if (z_) fn<V1234, true, true>(__VA_ARGS__); else fn<V1234, true, false>(__VA_ARGS__); } \
      else    { if (z_) fn<V1234, false, true>(__VA_ARGS__); else fn<V1234, false, false>(__VA_ARGS__); } \
    } else {                                                                               \
      if (d_) { if (z_) fn<V0124, true, true>(__VA_ARGS__); else fn<V0124, true, false>(__VA_ARGS__); } \
      else    { if (z_) fn<V0124, false, true>(__VA_ARGS__); else fn<V0124, false, false>(__VA_ARGS__); } \
    }                                                                                      \
  } while (0)

bool bad_variant(int v) { return v != SVBGPU_1234 && v != SVBGPU_0124; }
bool bad_flags(unsigned f) { return f & ~(SVBGPU_DELTA | SVBGPU_ZIGZAG); }

}  // namespace

// =============================================================================
// Device-pointer API
// =============================================================================
extern "C" {

const char* svbgpu_strerror(int code) {
  switch (code) {
    case SVBGPU_OK: return "ok";
    case SVBGPU_EINVAL: return "invalid argument";
    case SVBGPU_ENODEVICE: return "no usable gfx950 device";
    case SVBGPU_EWORKSPACE: return "workspace missing, misaligned or too small";
    case SVBGPU_EHIP: return hipGetErrorString(g_last_hip_error);
    case SVBGPU_ESCAN: return "device scan gave up waiting on another workgroup";
    case SVBGPU_ETOOBIG: return "count beyond what the library's static scratch is sized for";
    default: return "unknown error";
  }
}

int svbgpu_device_ok(void) { return dev_info().ok ? 1 : 0; }

int svbgpu_build_has(const char* feature) {
  if (!feature) return 0;
#ifdef SVB_LB_STRESS
  if (!std::strcmp(feature, "lookback_stress")) return 1;
#endif
  return 0;
}

int svbgpu_workspace_init(void* ws, size_t ws_bytes, void* stream) {
  if (!ws || ((uintptr_t)ws & 255u) || ws_bytes < sizeof(WsHeader) || (ws_bytes & 3u)) return SVBGPU_EWORKSPACE;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  const int rc = fill_u32(ws, 0u, ws_bytes, S(stream));
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_ws_mu);
  if (WsReg* r = ws_find(ws)) r->tag = 0;
  else g_ws_reg.emplace_back(ws, WsReg{});
  return SVBGPU_OK;
}

void svbgpu_workspace_release(void* ws) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  for (size_t i = 0; i < g_ws_reg.size(); ++i)
    if (g_ws_reg[i].first == ws) {
      g_ws_reg[i] = g_ws_reg.back();
      g_ws_reg.pop_back();
      return;
    }
}

int svbgpu_set_option(const char* name, int value) {
  if (!name) return SVBGPU_EINVAL;
  struct Knob { const char* name; std::atomic<int>* cell; bool (*ok)(int); };
  static const Knob knobs[] = {
      {"debug_ablate", &o_debug, [](int) { return true; }},
      {"zigzag_piece", &o_zigzag_piece, [](int v) { return v >= 1; }},
      {"decode_shortcut", &o_decode_shortcut, [](int v) { return v == 0 || v == 1; }},
      {"delta_decode_lookback", &o_delta_lookback, [](int v) { return v == 0 || v == 1; }},
      {"batch_lists_per_wave", &o_batch_lists_per_wave, [](int v) { return v >= 0; }},
      {"batch_chunk", &o_batch_chunk, [](int v) { return v >= -1 && v <= 64; }},
      {"decode_grid_per_cu", &o_decode_per_cu, [](int v) { return v >= 0; }},
      {"scan_waves", &o_scan_waves, [](int v) { return v == 4 || v == 8 || v == 16; }},
      {"encode_waves", &o_encode_waves, [](int v) { return v == 4 || v == 8 || v == 16; }},
      {"encode64_waves", &o_encode64_waves, [](int v) { return v == 4 || v == 8 || v == 16; }},
      {"encode_fresh", &o_encode_fresh, [](int v) { return v == 0 || v == 1; }},
      {"scan_fresh", &o_scan_fresh, [](int v) { return v == 0 || v == 1; }},
      {"encode_tile", &o_encode_qw, [](int v) { return v == 1 || v == 2 || v == 4; }},
      {"encode_pipeline", &o_encode_pipeline, [](int v) { return v >= 0 && v <= 3; }},
      {"encode_ticket_early", &o_encode_ticket_early, [](int v) { return v >= -1 && v <= 1; }},
      {"nt_threshold_mib", &o_nt_threshold_mib, [](int v) { return v >= -1; }},
      {"nt_force", &o_nt_force, [](int v) { return v >= -1 && v <= 3; }},
      {"host_cpu_cutoff", &o_host_cpu_cutoff, [](int v) { return v >= 0; }},
      {"host_pointers_only", &o_host_pointers_only, [](int v) { return v == 0 || v == 1; }},
      {"host_serialize", &o_host_serialize, [](int v) { return v == 0 || v == 1; }},
      {"host_stage_pageable", &o_host_stage_pageable, [](int v) { return v == 0 || v == 1; }},
  };
  for (const Knob& k : knobs)
    if (!std::strcmp(name, k.name)) {
      if (!k.ok(value)) return SVBGPU_EINVAL;
      k.cell->store(value, std::memory_order_relaxed);
      return SVBGPU_OK;
    }
  return SVBGPU_EINVAL;
}

size_t svbgpu_workspace_bytes(uint32_t count) { return ws_layout(count).total; }

int svbgpu_check_status(const void* d_workspace, void* stream) {
  if (!d_workspace) return SVBGPU_EWORKSPACE;
  u32 err = 0;
  HIP_TRY(hipStreamSynchronize(S(stream)));
  HIP_TRY(hipMemcpy(&err, d_workspace, 4, hipMemcpyDeviceToHost));
  return err ? SVBGPU_ESCAN : SVBGPU_OK;
}

int svbgpu_decode(int variant, unsigned flags, const uint8_t* d_in, uint32_t* d_out, uint32_t count,
                  uint32_t prev, void* ws, size_t ws_bytes, uint64_t* d_consumed, void* stream) {
  if (bad_variant(variant) || bad_flags(flags)) return SVBGPU_EINVAL;
  if (count == 0) {
    if (d_consumed) { int rc0 = fill_u32(d_consumed, 0u, 8, S(stream)); if (rc0) return rc0; }
    return SVBGPU_OK;
  }
  if (!d_in || !d_out) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  int rc = check_ws(count, ws, ws_bytes);
  if (rc) return rc;
  const Options o = options();
  const bool fold_clear = (flags & SVBGPU_DELTA) && o.delta_lookback && count > 0;  // one launch fewer
  rc = build_directory(variant, d_in, count, ws, S(stream), fold_clear);
  if (rc) return rc;
  return run_decode(o, variant, flags, d_in, d_out, count, prev, ws, d_consumed, S(stream), ws_bytes, fold_clear);
}

int svbgpu_decode_plan(int variant, const uint8_t* d_in, uint32_t count, void* ws, size_t ws_bytes,
                       void* stream) {
  if (bad_variant(variant)) return SVBGPU_EINVAL;
  if (count == 0) return SVBGPU_OK;
  if (!d_in) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  int rc = check_ws(count, ws, ws_bytes);
  if (rc) return rc;
  return build_directory(variant, d_in, count, ws, S(stream));
}

int svbgpu_decode_run(int variant, unsigned flags, const uint8_t* d_in, uint32_t* d_out,
                      uint32_t count, uint32_t prev, void* ws, size_t ws_bytes, uint64_t* d_consumed,
                      void* stream) {
  if (bad_variant(variant) || bad_flags(flags)) return SVBGPU_EINVAL;
  if (count == 0) {
    if (d_consumed) { int rc0 = fill_u32(d_consumed, 0u, 8, S(stream)); if (rc0) return rc0; }
    return SVBGPU_OK;
  }
  if (!d_in || !d_out) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  int rc = check_ws(count, ws, ws_bytes);
  if (rc) return rc;
  return run_decode(options(), variant, flags, d_in, d_out, count, prev, ws, d_consumed, S(stream));
}

}  // extern "C"

// ---- svbgpu_decode_static: the decoder on scratch the library itself holds ------
// Callers that cannot allocate (under graph capture, in a serving loop, in a
// harness that forbids it) have no workspace to pass.  The workspace of this
// entry point is a file-scope device array: kStaticScratchBytes on every GPU the
// library runs on, held until the process ends.  In host code the array's name
// is only a stand-in (passing it to a kernel would hand the kernel a host
// address): the device address comes from hipGetSymbolAddress, once per device,
// with that device current -- every GPU has its own copy at its own address.
constexpr uint32_t kStaticMaxCount = SVBGPU_STATIC_MAX_COUNT;
constexpr size_t kStaticScratchBytes = svb::ws_layout(kStaticMaxCount).total;
static_assert(kStaticScratchBytes <= (4u << 20), "the static scratch is meant to stay modest");
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {
void* g_static_ptr[64];  // device address of `scratch` per device ordinal, nullptr = not resolved yet
std::mutex g_static_mu;
// Device address of this device's copy of `scratch`.  Allocates nothing.
int static_scratch(void** out) {
  int d = 0;
  HIP_TRY(hipGetDevice(&d));
  if (d < 0 || d >= 64) return SVBGPU_ENODEVICE;
  std::lock_guard<std::mutex> lk(g_static_mu);
  if (!g_static_ptr[d]) {
    void* ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr || ((uintptr_t)ptr & 255u)) return SVBGPU_EWORKSPACE;
    g_static_ptr[d] = ptr;
  }
  *out = g_static_ptr[d];
  return SVBGPU_OK;
}
}  // namespace

extern "C" {

int svbgpu_decode_static(int variant, unsigned flags, const uint8_t* d_in, uint32_t* d_out, uint32_t count,
                         uint32_t prev, uint64_t* d_consumed, void* stream) {
  if (bad_variant(variant) || bad_flags(flags)) return SVBGPU_EINVAL;
  if (count > kStaticMaxCount) return SVBGPU_ETOOBIG;
  if (count == 0) {
    if (d_consumed) { int rc0 = fill_u32(d_consumed, 0u, 8, S(stream)); if (rc0) return rc0; }
    return SVBGPU_OK;
  }
  if (!d_in || !d_out) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  void* ws = nullptr;  // never the array's name: the address hipGetSymbolAddress gave for this device
  int rc = static_scratch(&ws);
  if (rc) return rc;
  // Same kernels as svbgpu_decode.  Everything they read from the workspace is
  // written first by this very call (directory by build_directory, which also clears the header
  // and look-back words ahead of the look-back decoder), so what
  // the previous call left behind does not matter.
  const Options o = options();
  const bool fold_clear = (flags & SVBGPU_DELTA) && o.delta_lookback;
  rc = build_directory(variant, d_in, count, ws, S(stream), fold_clear);
  if (rc) return rc;
  return run_decode(o, variant, flags, d_in, d_out, count, prev, ws, d_consumed, S(stream), 0, fold_clear);
}

int svbgpu_static_status(uint32_t* d_error, void* stream) {
  if (!d_error) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  void* ws = nullptr;
  const int rc = static_scratch(&ws);
  if (rc) return rc;
  hipLaunchKernelGGL(k_copy_u32, dim3(1), dim3(64), 0, S(stream), &ws_hdr(ws)->error, d_error);
  HIP_TRY(hipGetLastError());
  return SVBGPU_OK;
}

int svbgpu_encode(int variant, unsigned flags, const uint32_t* d_in, uint32_t count, uint8_t* d_out,
                  uint32_t prev, void* ws, size_t ws_bytes, uint64_t* d_written, void* stream) {
  if (bad_variant(variant) || bad_flags(flags)) return SVBGPU_EINVAL;
  if (count == 0) {
    if (d_written) { int rc0 = fill_u32(d_written, 0u, 8, S(stream)); if (rc0) return rc0; }
    return SVBGPU_OK;
  }
  if (!d_in || !d_out) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  int rc = check_ws(count, ws, ws_bytes);
  if (rc) return rc;
  Options o = options();
  rc = prepare_lookback(count, ws, S(stream), &o.lb_tag2);
  if (rc) return rc;
  DISPATCH_VDZ(launch_encode, variant, flags, o, d_in, count, d_out, prev, ws, ws_bytes, d_written, S(stream));
  HIP_TRY(hipGetLastError());
  return finish_lookback(ws, S(stream), o.lb_tag2);
}

int svbgpu_quantize_encode(int variant, const float* d_in, uint32_t count, float scale, uint8_t* d_out,
                           void* ws, size_t ws_bytes, uint64_t* d_written, void* stream) {
  if (bad_variant(variant)) return SVBGPU_EINVAL;
  if (count == 0) {
    if (d_written) { int rc0 = fill_u32(d_written, 0u, 8, S(stream)); if (rc0) return rc0; }
    return SVBGPU_OK;
  }
  if (!d_in || !d_out) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  int rc = check_ws(count, ws, ws_bytes);
  if (rc) return rc;
  Options o = options();
  rc = prepare_lookback(count, ws, S(stream), &o.lb_tag2);
  if (rc) return rc;
  uint32_t bits;
  std::memcpy(&bits, &scale, 4);
  const uint32_t* in = reinterpret_cast<const uint32_t*>(d_in);
  if (choose_encoder(o, false, count).pipe) {
    if (variant == SVBGPU_1234) launch_encode_pipe<V1234, false, true, 2, true>(o, in, count, d_out, bits, ws, ws_bytes, d_written, S(stream));
    else launch_encode_pipe<V0124, false, true, 2, true>(o, in, count, d_out, bits, ws, ws_bytes, d_written, S(stream));
  } else if (variant == SVBGPU_1234)
    launch_encode_qw<V1234, false, true, 4, true, 8>(o, in, count, d_out, bits, ws, ws_bytes, d_written, S(stream));
  else
    launch_encode_qw<V0124, false, true, 4, true, 8>(o, in, count, d_out, bits, ws, ws_bytes, d_written, S(stream));
  HIP_TRY(hipGetLastError());
  return finish_lookback(ws, S(stream), o.lb_tag2);
}

int svbgpu_decode_dequantize(int variant, const uint8_t* d_in, float* d_out, uint32_t count, float scale,
                             int accumulate, void* ws, size_t ws_bytes, uint64_t* d_consumed, void* stream) {
  if (bad_variant(variant)) return SVBGPU_EINVAL;
  if (count == 0) {
    if (d_consumed) { int rc0 = fill_u32(d_consumed, 0u, 8, S(stream)); if (rc0) return rc0; }
    return SVBGPU_OK;
  }
  if (!d_in || !d_out) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  int rc = check_ws(count, ws, ws_bytes);
  if (rc) return rc;
  rc = build_directory(variant, d_in, count, ws, S(stream));
  if (rc) return rc;
  uint32_t bits;
  std::memcpy(&bits, &scale, 4);
  const WsLayout L = ws_layout(count);
  const unsigned grid = stream_grid((L.n_wt + kWaves - 1) / kWaves, options().decode_per_cu);
  u32* out = reinterpret_cast<u32*>(d_out);
  const Dir dir = ws_dir(ws, L);
  hipStream_t st = S(stream);
  // accumulate reads what it then overwrites: those lines are worth keeping until the store, so only
  // the plain dequantising store follows the call's non-temporal policy
  const int nt = accumulate ? 0 : nt_policy(options(), min_stream_bytes(variant == SVBGPU_1234 ? V1234 : V0124, count), 4ull * count);
  if (variant == SVBGPU_1234) {
    if (accumulate) hipLaunchKernelGGL((k_decode<V1234, true, true, 2>), dim3(grid), dim3(kBlock), 0, st, d_in, out, count, dir, d_consumed, bits);
    else SVB_NT_SWITCH_023(nt, NTV, hipLaunchKernelGGL((k_decode<V1234, true, true, 1, NTV>), dim3(grid), dim3(kBlock), 0, st, d_in, out, count, dir, d_consumed, bits))
  } else {
    if (accumulate) hipLaunchKernelGGL((k_decode<V0124, true, true, 2>), dim3(grid), dim3(kBlock), 0, st, d_in, out, count, dir, d_consumed, bits);
    else SVB_NT_SWITCH_023(nt, NTV, hipLaunchKernelGGL((k_decode<V0124, true, true, 1, NTV>), dim3(grid), dim3(kBlock), 0, st, d_in, out, count, dir, d_consumed, bits))
  }
  HIP_TRY(hipGetLastError());
  return SVBGPU_OK;
}

int svbgpu_compressedbytes(int variant, unsigned flags, const uint32_t* d_in, uint32_t count,
                           uint32_t prev, uint64_t* d_bytes, void* stream) {
  if (bad_variant(variant) || bad_flags(flags) || !d_bytes) return SVBGPU_EINVAL;
  { int rc0 = fill_u32(d_bytes, 0u, 8, S(stream)); if (rc0) return rc0; }
  if (count == 0) return SVBGPU_OK;
  if (!d_in) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  DISPATCH_VDZ(launch_size, variant, flags, d_in, count, prev, d_bytes, S(stream));
  HIP_TRY(hipGetLastError());
  return SVBGPU_OK;
}

int svbgpu_validate(int variant, const uint8_t* d_in, uint64_t in_length, uint32_t count,
                    uint32_t* d_ok, void* ws, size_t ws_bytes, void* stream) {
  if ((bad_variant(variant) && variant != SVBGPU_1248) || !d_ok) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  if (!ws || ws_bytes < sizeof(WsHeader) || ((uintptr_t)ws & 255u)) return SVBGPU_EWORKSPACE;
  const u64 keylen = ((u64)count + 3) >> 2;
  if (in_length == 0 || count == 0 || keylen > in_length) {
    // nothing to sum (reference src/streamvbyte_decode.c:91-111): the answer is known here
    const int r = (in_length == 0 || count == 0) ? in_length == (u64)count : 0;
    return fill_u32(d_ok, (uint32_t)r, 4, S(stream));
  }
  if (!d_in) return SVBGPU_EINVAL;
  WsHeader* h = ws_hdr(ws);
  // the kernel leaves the accumulator at zero behind it; only a workspace that was never initialised
  // may hold anything else
  if (!ws_registered(ws)) { int rc0 = fill_u32(&h->accum, 0u, 8, S(stream)); if (rc0) return rc0; }
  // a workgroup of 1024 threads per CU at most, at least 16 KiB of keys (one load per thread) each
  const unsigned grid = stream_grid((keylen + 16383) / 16384, 1);
  if (variant == SVBGPU_1234)
    hipLaunchKernelGGL(k_validate_sum<V1234>, dim3(grid), dim3(1024), 0, S(stream), d_in, count,
                       (u64)in_length, &h->accum, d_ok);
  else if (variant == SVBGPU_1248)
    hipLaunchKernelGGL(k_validate_sum<V1248>, dim3(grid), dim3(1024), 0, S(stream), d_in, count,
                       (u64)in_length, &h->accum, d_ok);
  else
    hipLaunchKernelGGL(k_validate_sum<V0124>, dim3(grid), dim3(1024), 0, S(stream), d_in, count,
                       (u64)in_length, &h->accum, d_ok);
  HIP_TRY(hipGetLastError());
  return SVBGPU_OK;
}

static int zigzag_launch(int mode, const uint32_t* in, uint32_t* out, size_t n, uint32_t prev,
                         void* stream) {
  if (n == 0) return SVBGPU_OK;
  if (!in || !out) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  const unsigned grid = stream_grid(((unsigned long long)n / 4 + kBlock - 1) / kBlock + 1, 0);
  SVB_NT_SWITCH_03(nt_policy(options(), 4ull * n, 4ull * n), NTV, {
    if (mode == 0) hipLaunchKernelGGL((k_zigzag<0, NTV>), dim3(grid), dim3(kBlock), 0, S(stream), in, out, (u64)n, prev);
    else if (mode == 1) hipLaunchKernelGGL((k_zigzag<1, NTV>), dim3(grid), dim3(kBlock), 0, S(stream), in, out, (u64)n, prev);
    else hipLaunchKernelGGL((k_zigzag<2, NTV>), dim3(grid), dim3(kBlock), 0, S(stream), in, out, (u64)n, prev);
  })
  HIP_TRY(hipGetLastError());
  return SVBGPU_OK;
}

int svbgpu_zigzag_encode(const int32_t* d_in, uint32_t* d_out, size_t n, void* stream) {
  return zigzag_launch(0, reinterpret_cast<const uint32_t*>(d_in), d_out, n, 0, stream);
}
int svbgpu_zigzag_decode(const uint32_t* d_in, int32_t* d_out, size_t n, void* stream) {
  return zigzag_launch(1, d_in, reinterpret_cast<uint32_t*>(d_out), n, 0, stream);
}
int svbgpu_zigzag_delta_encode(const int32_t* d_in, uint32_t* d_out, size_t n, int32_t prev,
                               void* stream) {
  return zigzag_launch(2, reinterpret_cast<const uint32_t*>(d_in), d_out, n, (uint32_t)prev, stream);
}
int svbgpu_zigzag_delta_decode(const uint32_t* d_in, int32_t* d_out, size_t n, int32_t prev,
                               void* ws, size_t ws_bytes, void* stream) {
  if (n == 0) return SVBGPU_OK;
  if (!d_in || !d_out || n > 0xFFFFFFFFull) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  const uint32_t count = (uint32_t)n;
  int rc = check_ws(count, ws, ws_bytes);
  if (rc) return rc;
  unsigned tag2 = 0;
  rc = prepare_lookback(count, ws, S(stream), &tag2);
  if (rc) return rc;
  const WsLayout L = ws_layout(count);
  const u32 n_tiles8 = (u32)((L.n_wt + 31) / 32);  // 8 waves of 4 wave tiles per workgroup
  const u32 n_tiles16 = (u32)((L.n_wt + 63) / 64);
  const int scan_waves = options().scan_waves;
  const bool fresh = options().scan_fresh;  // a workgroup per tile instead of persistent workgroups
  if (scan_waves == 16)
    SVB_NT_SWITCH_03(nt_policy(options(), 4ull * count, 4ull * count), NTV,
                     hipLaunchKernelGGL((k_running_sum<true, 16, NTV>), dim3(fresh ? n_tiles16 : stream_grid(n_tiles16, 1)), dim3(1024), 0,
                                        S(stream), d_in, reinterpret_cast<uint32_t*>(d_out), count, (uint32_t)prev,
                                        n_tiles16, ws_hdr_slot(ws, tag2),
                                        lb_words(ws, L, tag2)))
  else if (scan_waves == 8)
    hipLaunchKernelGGL((k_running_sum<true, 8>), dim3(fresh ? n_tiles8 : stream_grid(n_tiles8, 2)), dim3(512), 0, S(stream), d_in,
                       reinterpret_cast<uint32_t*>(d_out), count, (uint32_t)prev, n_tiles8, ws_hdr_slot(ws, tag2),
                       lb_words(ws, L, tag2));
  else
  hipLaunchKernelGGL(k_running_sum<true>, dim3(stream_grid(L.n_tiles, 4)), dim3(kBlock), 0, S(stream), d_in,
                     reinterpret_cast<uint32_t*>(d_out), count, (uint32_t)prev, (u32)L.n_tiles, ws_hdr_slot(ws, tag2),
                     lb_words(ws, L, tag2));
  HIP_TRY(hipGetLastError());
  return finish_lookback(ws, S(stream), tag2);
}

extern "C++" {
template <int V>
void launch_intersect(const uint8_t* d_in, uint32_t count, uint32_t prev, const uint32_t* d_sorted,
                      uint32_t n_sorted, uint32_t* d_out, uint64_t* d_n_out, void* ws, uint64_t* d_consumed,
                      hipStream_t st) {
  const WsLayout L = ws_layout(count);
  const unsigned grid = stream_grid((L.n_wt + kWaves - 1) / kWaves, options().decode_per_cu);
  const Dir dir = ws_dir(ws, L);
  u32* vsum = ws_at<u32>(ws, L.off_wt_vsum);
  u32* blk = ws_at<u32>(ws, L.off_blk);
  const unsigned nblk = (unsigned)((L.n_wt + 1023) / 1024);
  hipLaunchKernelGGL((k_value_sums<V, false>), dim3(grid), dim3(kBlock), 0, st, d_in, count, dir, vsum);
  hipLaunchKernelGGL(k_scan_u32_blocks, dim3(nblk), dim3(kBlock), 0, st, vsum, (u32)L.n_wt, blk);
  hipLaunchKernelGGL(k_scan_u32_top, dim3(1), dim3(1024), 0, st, blk, nblk, prev);
  u32* bound = ws_at<u32>(ws, L.off_bounds);
  const u32 n_waves = (u32)((L.n_wt + 3) / 4);
  if (n_sorted)
    hipLaunchKernelGGL(k_wave_bounds, dim3((n_waves + kBlock - 1) / kBlock), dim3(kBlock), 0, st, vsum, blk, n_waves,
                       d_sorted, n_sorted, bound);
  hipLaunchKernelGGL((k_delta_intersect<V>), dim3((unsigned)L.n_tiles), dim3(kBlock), 0, st, d_in, count, dir, vsum, blk, d_sorted, n_sorted, bound, d_out, d_n_out, (u32)L.n_tiles, ws_hdr(ws),
                     lb_words(ws, L, 0u), d_consumed);
}
}  // extern "C++"

int svbgpu_decode_intersect(int variant, const uint8_t* d_in, uint32_t count, uint32_t prev,
                            const uint32_t* d_sorted, uint32_t n_sorted, uint32_t* d_out, uint64_t* d_n_out,
                            void* ws, size_t ws_bytes, uint64_t* d_consumed, void* stream) {
  if (bad_variant(variant) || !d_n_out || (n_sorted && !d_sorted)) return SVBGPU_EINVAL;
  if (count == 0) {
    int rc0 = fill_u32(d_n_out, 0u, 8, S(stream));
    if (!rc0 && d_consumed) rc0 = fill_u32(d_consumed, 0u, 8, S(stream));
    return rc0;
  }
  if (!d_in || !d_out) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  int rc = check_ws(count, ws, ws_bytes);
  if (rc) return rc;
  rc = clear_lookback(count, ws, S(stream));
  if (rc) return rc;
  rc = build_directory(variant, d_in, count, ws, S(stream));
  if (rc) return rc;
  if (variant == SVBGPU_1234)
    launch_intersect<V1234>(d_in, count, prev, d_sorted, n_sorted, d_out, d_n_out, ws, d_consumed, S(stream));
  else
    launch_intersect<V0124>(d_in, count, prev, d_sorted, n_sorted, d_out, d_n_out, ws, d_consumed, S(stream));
  HIP_TRY(hipGetLastError());
  return SVBGPU_OK;
}

extern "C++" {
// The pre-pass both sides of svbgpu_intersect_streams get: directory, then the value preceding every wave tile.
template <int V>
void launch_value_prefix(const uint8_t* d_in, uint32_t count, uint32_t prev, void* ws, hipStream_t st) {
  const WsLayout L = ws_layout(count);
  const unsigned grid = stream_grid((L.n_wt + kWaves - 1) / kWaves, options().decode_per_cu);
  u32* vsum = ws_at<u32>(ws, L.off_wt_vsum);
  u32* blk = ws_at<u32>(ws, L.off_blk);
  const unsigned nblk = (unsigned)((L.n_wt + 1023) / 1024);
  hipLaunchKernelGGL((k_value_sums<V, false>), dim3(grid), dim3(kBlock), 0, st, d_in, count, ws_dir(ws, L), vsum);
  hipLaunchKernelGGL(k_scan_u32_blocks, dim3(nblk), dim3(kBlock), 0, st, vsum, (u32)L.n_wt, blk);
  hipLaunchKernelGGL(k_scan_u32_top, dim3(1), dim3(1024), 0, st, blk, nblk, prev);
}
}  // extern "C++"

// two workspaces back to back, then one (first, end) pair of wave tiles of B per wavefront of k_stream_intersect
static size_t sx_range_bytes(const WsLayout& La) { return align256(8 * (La.n_wt / kSxQ + 2)); }
size_t svbgpu_intersect_workspace_bytes(uint32_t count_a, uint32_t count_b) {
  const WsLayout La = ws_layout(count_a);
  return La.total + ws_layout(count_b).total + sx_range_bytes(La);
}

int svbgpu_intersect_streams(int variant_a, const uint8_t* d_a, uint32_t count_a, uint32_t prev_a, int variant_b,
                             const uint8_t* d_b, uint32_t count_b, uint32_t prev_b, uint32_t* d_out,
                             uint64_t* d_n_out, void* ws, size_t ws_bytes, void* stream) {
  if (bad_variant(variant_a) || bad_variant(variant_b) || !d_n_out) return SVBGPU_EINVAL;
  if (count_a == 0 || count_b == 0) return fill_u32(d_n_out, 0u, 8, S(stream));
  if (!d_a || !d_b || !d_out) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  const WsLayout La = ws_layout(count_a), Lb = ws_layout(count_b);
  if (!ws || ((uintptr_t)ws & 255u) || ws_bytes < La.total + Lb.total + sx_range_bytes(La)) return SVBGPU_EWORKSPACE;
  void* ws_b = static_cast<char*>(ws) + La.total;  // totals are multiples of 256
  uint2* range = reinterpret_cast<uint2*>(static_cast<char*>(ws_b) + Lb.total);
  hipStream_t st = S(stream);
  int rc = clear_lookback(count_a, ws, st);
  if (rc) return rc;
  rc = build_directory(variant_a, d_a, count_a, ws, st);
  if (rc) return rc;
  rc = build_directory(variant_b, d_b, count_b, ws_b, st);
  if (rc) return rc;
  if (variant_a == SVBGPU_1234) launch_value_prefix<V1234>(d_a, count_a, prev_a, ws, st);
  else launch_value_prefix<V0124>(d_a, count_a, prev_a, ws, st);
  if (variant_b == SVBGPU_1234) launch_value_prefix<V1234>(d_b, count_b, prev_b, ws_b, st);
  else launch_value_prefix<V0124>(d_b, count_b, prev_b, ws_b, st);
  const u32 n_tiles = (u32)((La.n_wt + kSxWaves * kSxQ - 1) / (kSxWaves * kSxQ));
  const u32 n_sx_waves = (u32)((La.n_wt + kSxQ - 1) / kSxQ);
  hipLaunchKernelGGL(k_stream_bounds, dim3((n_sx_waves + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                     ws_at<u32>(ws, La.off_wt_vsum), ws_at<u32>(ws, La.off_blk), (u32)La.n_wt,
                     ws_at<u32>(ws_b, Lb.off_wt_vsum), ws_at<u32>(ws_b, Lb.off_blk), (u32)Lb.n_wt, range);
#define SVB_SX_LAUNCH(VA, VB)                                                                                   \
  hipLaunchKernelGGL((k_stream_intersect<VA, VB, kSxWaves>), dim3(n_tiles), dim3(kSxWaves * 64), 0, st, d_a, count_a, \
                     ws_dir(ws, La), ws_at<u32>(ws, La.off_wt_vsum), ws_at<u32>(ws, La.off_blk), d_b, count_b,  \
                     ws_dir(ws_b, Lb), ws_at<u32>(ws_b, Lb.off_wt_vsum), ws_at<u32>(ws_b, Lb.off_blk), range,   \
                     d_out, d_n_out, n_tiles, ws_hdr(ws), lb_words(ws, La, 0u))
  if (variant_a == SVBGPU_1234) {
    if (variant_b == SVBGPU_1234) SVB_SX_LAUNCH(V1234, V1234); else SVB_SX_LAUNCH(V1234, V0124);
  } else {
    if (variant_b == SVBGPU_1234) SVB_SX_LAUNCH(V0124, V1234); else SVB_SX_LAUNCH(V0124, V0124);
  }
#undef SVB_SX_LAUNCH
  HIP_TRY(hipGetLastError());
  return SVBGPU_OK;
}

int svbgpu_intersect_batch(int variant, const uint8_t* d_bytes_a, const uint64_t* d_byte_off_a,
                           const uint32_t* d_counts_a, const uint32_t* d_prevs_a, const uint8_t* d_bytes_b,
                           const uint64_t* d_byte_off_b, const uint32_t* d_counts_b, const uint32_t* d_prevs_b,
                           uint32_t npairs, uint32_t* d_out, const uint64_t* d_out_off, uint32_t* d_n_out,
                           void* stream) {
  if (bad_variant(variant)) return SVBGPU_EINVAL;
  if (npairs == 0) return SVBGPU_OK;
  if (!d_bytes_a || !d_byte_off_a || !d_counts_a || !d_bytes_b || !d_byte_off_b || !d_counts_b || !d_out ||
      !d_out_off || !d_n_out)
    return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  const unsigned grid = (npairs + kIsbWaves - 1) / kIsbWaves;
  if (variant == SVBGPU_1234)
    hipLaunchKernelGGL(k_intersect_batch<V1234>, dim3(grid), dim3(64 * kIsbWaves), 0, S(stream), d_bytes_a, d_byte_off_a,
                       d_counts_a, d_prevs_a, d_bytes_b, d_byte_off_b, d_counts_b, d_prevs_b, npairs, d_out,
                       d_out_off, d_n_out);
  else
    hipLaunchKernelGGL(k_intersect_batch<V0124>, dim3(grid), dim3(64 * kIsbWaves), 0, S(stream), d_bytes_a, d_byte_off_a,
                       d_counts_a, d_prevs_a, d_bytes_b, d_byte_off_b, d_counts_b, d_prevs_b, npairs, d_out,
                       d_out_off, d_n_out);
  HIP_TRY(hipGetLastError());
  return SVBGPU_OK;
}

int svbgpu_encode64(unsigned flags, const uint64_t* d_in, uint32_t count, uint8_t* d_out, uint64_t prev, void* ws,
                    size_t ws_bytes, uint64_t* d_written, void* stream) {
  if (bad_flags(flags)) return SVBGPU_EINVAL;
  if (count == 0) {
    if (d_written) { int rc0 = fill_u32(d_written, 0u, 8, S(stream)); if (rc0) return rc0; }
    return SVBGPU_OK;
  }
  if (!d_in || !d_out) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  int rc = check_ws(count, ws, ws_bytes);
  if (rc) return rc;
  unsigned tag2 = 0;
  rc = prepare_lookback(count, ws, S(stream), &tag2);
  if (rc) return rc;
  const WsLayout L = ws_layout(count);
  const LbWords lbw = lb_words(ws, L, tag2);
  const Options o = options();
  const bool d = flags & SVBGPU_DELTA, z = flags & SVBGPU_ZIGZAG;
  // NW = wave tiles (= waves) per look-back tile
#define SVB_ENC64_NW(D, Z, NW, PER_CU)                                                                               \
  {                                                                                                                  \
    const unsigned n_tiles = (unsigned)((L.n_wt + NW - 1) / NW);                                                     \
    const unsigned grid = stream_grid(n_tiles, PER_CU);                                                              \
    SVB_NT_SWITCH_013(nt_policy(o, 8ull * count, (count + 3ull) / 4 + count), NTV,                                   \
                      hipLaunchKernelGGL((k_encode64_lb<D, Z, NW, NTV>), dim3(grid), dim3(NW * 64), 0, S(stream),     \
                                         d_in, count, n_tiles, d_out, (u64)prev, ws_hdr_slot(ws, tag2), lbw,         \
                                         d_written))                                                                 \
  }
#define SVB_ENC64(D, Z)                                          \
  {                                                              \
    if (o.encode64_waves == 16) SVB_ENC64_NW(D, Z, 16, 1)        \
    else if (o.encode64_waves == 4) SVB_ENC64_NW(D, Z, 4, 4)     \
    else SVB_ENC64_NW(D, Z, 8, 2)                                \
  }
  if (d) { if (z) SVB_ENC64(true, true) else SVB_ENC64(true, false) }
  else   { if (z) SVB_ENC64(false, true) else SVB_ENC64(false, false) }
#undef SVB_ENC64
#undef SVB_ENC64_NW
  HIP_TRY(hipGetLastError());
  return finish_lookback(ws, S(stream), tag2);
}

int svbgpu_decode64(unsigned flags, const uint8_t* d_in, uint64_t* d_out, uint32_t count, uint64_t prev, void* ws,
                    size_t ws_bytes, uint64_t* d_consumed, void* stream) {
  if (bad_flags(flags)) return SVBGPU_EINVAL;
  if (count == 0) {
    if (d_consumed) { int rc0 = fill_u32(d_consumed, 0u, 8, S(stream)); if (rc0) return rc0; }
    return SVBGPU_OK;
  }
  if (!d_in || !d_out) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  int rc = check_ws(count, ws, ws_bytes);
  if (rc) return rc;
  rc = build_directory(SVBGPU_1248, d_in, count, ws, S(stream));
  if (rc) return rc;
  const WsLayout L = ws_layout(count);
  const unsigned grid = (unsigned)((L.n_wt + kWaves - 1) / kWaves);
  const bool d = flags & SVBGPU_DELTA, z = flags & SVBGPU_ZIGZAG;
  const Dir dir = ws_dir(ws, L);
  // differential: value sums per workgroup of four wave tiles (8 bytes each, in the room of the 32-bit
  // decoder's per-wave-tile sums), scanned in place, then the decoder adds them up
  u64* wg = ws_at<u64>(ws, L.off_wt_vsum);
  if (d) {
    u64* blk = ws_at<u64>(ws, L.off_blk);
    const unsigned nblk = (grid + 4095) / 4096;
    if (z) hipLaunchKernelGGL(k_value_sums64<true>, dim3(grid), dim3(kBlock), 0, S(stream), d_in, count, dir, wg);
    else hipLaunchKernelGGL(k_value_sums64<false>, dim3(grid), dim3(kBlock), 0, S(stream), d_in, count, dir, wg);
    hipLaunchKernelGGL(k_block_sums64, dim3(nblk), dim3(1024), 0, S(stream), wg, (u64)grid, blk);
    hipLaunchKernelGGL(k_multi_offsets, dim3(nblk), dim3(1024), 0, S(stream), wg, (u64)grid, blk, wg);
  }
#define SVB_DEC64(D, Z)                                                                                          \
  SVB_NT_SWITCH_023(nt_policy(options(), (count + 3ull) / 4 + count, 8ull * count), NTV,                         \
                    hipLaunchKernelGGL((k_decode64<D, Z, NTV>), dim3(grid), dim3(kBlock), 0, S(stream), d_in, d_out, \
                                       count, dir, d_consumed, (u64)prev, wg))
  if (d) { if (z) SVB_DEC64(true, true) else SVB_DEC64(true, false) }
  else   { if (z) SVB_DEC64(false, true) else SVB_DEC64(false, false) }
#undef SVB_DEC64
  HIP_TRY(hipGetLastError());
  return SVBGPU_OK;
}

int svbgpu_validate_batch(int variant, const uint8_t* d_bytes, const uint64_t* d_byte_off,
                          const uint64_t* d_lengths, const uint32_t* d_counts, uint32_t nstreams,
                          uint32_t* d_ok, void* stream) {
  if (bad_variant(variant)) return SVBGPU_EINVAL;
  if (nstreams == 0) return SVBGPU_OK;
  if (!d_bytes || !d_byte_off || !d_lengths || !d_counts || !d_ok) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  const unsigned grid = batch_grid(options(), nstreams);
  if (variant == SVBGPU_1234)
    hipLaunchKernelGGL((k_validate_lists<V1234, u64, true>), dim3(grid), dim3(kBlock), 0, S(stream), d_bytes, d_byte_off,
                       d_counts, d_lengths, nstreams, d_ok);
  else
    hipLaunchKernelGGL((k_validate_lists<V0124, u64, true>), dim3(grid), dim3(kBlock), 0, S(stream), d_bytes, d_byte_off,
                       d_counts, d_lengths, nstreams, d_ok);
  HIP_TRY(hipGetLastError());
  return SVBGPU_OK;
}

int svbgpu_compressedbytes_batch(int variant, unsigned flags, const uint32_t* d_values, const uint64_t* d_value_off,
                                 const uint32_t* d_counts, const uint32_t* d_prev, uint32_t nstreams,
                                 uint64_t* d_sizes, void* stream) {
  if (bad_variant(variant) || bad_flags(flags)) return SVBGPU_EINVAL;
  if (nstreams == 0) return SVBGPU_OK;
  if (!d_values || !d_value_off || !d_counts || !d_sizes) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  DISPATCH_VDZ(launch_size_batch, variant, flags, d_values, d_value_off, d_counts, d_prev, nstreams, d_sizes, S(stream));
  HIP_TRY(hipGetLastError());
  return SVBGPU_OK;
}

int svbgpu_encode_batch(int variant, unsigned flags, const uint32_t* d_values,
                        const uint64_t* d_value_off, uint8_t* d_bytes, const uint64_t* d_byte_off,
                        const uint32_t* d_counts, const uint32_t* d_prev, uint32_t nstreams,
                        uint64_t* d_sizes, void* stream) {
  if (bad_variant(variant) || bad_flags(flags)) return SVBGPU_EINVAL;
  if (nstreams == 0) return SVBGPU_OK;
  if (!d_values || !d_value_off || !d_bytes || !d_byte_off || !d_counts || !d_sizes) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  const Options o = options();
  const unsigned grid = batch_grid(o, nstreams);
  DISPATCH_VDZ(launch_encode_batch, variant, flags, grid, batch_chunk(o, nstreams, 8), d_values, d_value_off, d_bytes, d_byte_off,
               d_counts, d_prev, nstreams, d_sizes, batch_nt_force(o), batch_nt_bytes(o), S(stream));
  HIP_TRY(hipGetLastError());
  return SVBGPU_OK;
}

int svbgpu_decode_batch(int variant, unsigned flags, const uint8_t* d_bytes, const uint64_t* d_byte_off,
                        uint32_t* d_values, const uint64_t* d_value_off, const uint32_t* d_counts,
                        const uint32_t* d_prev, uint32_t nstreams, uint64_t* d_sizes, void* stream) {
  if (bad_variant(variant) || bad_flags(flags)) return SVBGPU_EINVAL;
  if (nstreams == 0) return SVBGPU_OK;
  if (!d_bytes || !d_byte_off || !d_values || !d_value_off || !d_counts) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  const Options o = options();
  const unsigned grid = batch_grid(o, nstreams);
  DISPATCH_VDZ(launch_decode_batch, variant, flags, grid, batch_chunk(o, nstreams, 16), d_bytes, d_byte_off, d_values, d_value_off,
               d_counts, d_prev, nstreams, d_sizes, batch_nt_force(o), batch_nt_bytes(o), S(stream));
  HIP_TRY(hipGetLastError());
  return SVBGPU_OK;
}

int svbgpu_encode64_batch(unsigned flags, const uint64_t* d_values, const uint64_t* d_value_off, uint8_t* d_bytes,
                          const uint64_t* d_byte_off, const uint32_t* d_counts, const uint64_t* d_prev,
                          uint32_t nstreams, uint64_t* d_sizes, void* stream) {
  if (bad_flags(flags)) return SVBGPU_EINVAL;
  if (nstreams == 0) return SVBGPU_OK;
  if (!d_values || !d_value_off || !d_bytes || !d_byte_off || !d_counts || !d_sizes) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  const unsigned grid = (nstreams + kBatch64Waves - 1) / kBatch64Waves;
  const bool d = flags & SVBGPU_DELTA, z = flags & SVBGPU_ZIGZAG;
#define SVB_ENC64B(D, Z)                                                                                          \
  hipLaunchKernelGGL((k_encode64_batch<D, Z>), dim3(grid), dim3(64 * kBatch64Waves), 0, S(stream), d_values, d_value_off, d_bytes, \
                     d_byte_off, d_counts, d_prev, nstreams, d_sizes)
  if (d) { if (z) SVB_ENC64B(true, true); else SVB_ENC64B(true, false); }
  else   { if (z) SVB_ENC64B(false, true); else SVB_ENC64B(false, false); }
#undef SVB_ENC64B
  HIP_TRY(hipGetLastError());
  return SVBGPU_OK;
}

int svbgpu_decode64_batch(unsigned flags, const uint8_t* d_bytes, const uint64_t* d_byte_off, uint64_t* d_values,
                          const uint64_t* d_value_off, const uint32_t* d_counts, const uint64_t* d_prev,
                          uint32_t nstreams, uint64_t* d_sizes, void* stream) {
  if (bad_flags(flags)) return SVBGPU_EINVAL;
  if (nstreams == 0) return SVBGPU_OK;
  if (!d_bytes || !d_byte_off || !d_values || !d_value_off || !d_counts) return SVBGPU_EINVAL;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  const unsigned grid = (nstreams + kBatch64Waves - 1) / kBatch64Waves;
  const bool d = flags & SVBGPU_DELTA, z = flags & SVBGPU_ZIGZAG;
#define SVB_DEC64B(D, Z)                                                                                          \
  hipLaunchKernelGGL((k_decode64_batch<D, Z>), dim3(grid), dim3(64 * kBatch64Waves), 0, S(stream), d_bytes, d_byte_off, d_values, \
                     d_value_off, d_counts, d_prev, nstreams, d_sizes)
  if (d) { if (z) SVB_DEC64B(true, true); else SVB_DEC64B(true, false); }
  else   { if (z) SVB_DEC64B(false, true); else SVB_DEC64B(false, false); }
#undef SVB_DEC64B
  HIP_TRY(hipGetLastError());
  return SVBGPU_OK;
}

}  // extern "C"

// =============================================================================
// Reference-compatible API (host or device pointers)
// =============================================================================
namespace {

[[noreturn]] void die(const char* what, int rc) {
  std::fprintf(stderr, "libstreamvbyte (gfx950): %s failed: %s\n", what, svbgpu_strerror(rc));
  std::abort();
}
#define HIP_OR_DIE(expr)                                                              \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      std::fprintf(stderr, "libstreamvbyte (gfx950): %s: %s\n", #expr, hipGetErrorString(e_)); \
      std::abort();                                                                   \
    }                                                                                 \
  } while (0)

// Per-call scratch, recycled through a pool so concurrent callers never share
// a stream or a buffer (the reference API is re-entrant; this keeps it so).
struct Ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  void* ws = nullptr;      size_t ws_cap = 0;
  void* a = nullptr;       size_t a_cap = 0;   // staged input
  void* b = nullptr;       size_t b_cap = 0;   // staged output
  void* p = nullptr;       size_t p_cap = 0;   // list frames: the packed streams
  void* dir = nullptr;     size_t dir_cap = 0; // list frames: the directory on the device
  void* hdir = nullptr;    size_t hdir_cap = 0;  // ... and its page-locked host image
  uint64_t* d_res = nullptr;                    // 64 bytes of device results
  uint64_t* h_res = nullptr;                    // pinned mirror
  void* h_sums = nullptr;                       // pinned, 1 KiB: the directory's mega tile sums
  uint8_t* stage[2] = {nullptr, nullptr};       // page-locked staging chunks for pageable caller memory
  hipEvent_t stage_ev[2] = {nullptr, nullptr};  // the chunk's last copy has ended
};
constexpr size_t kStageChunk = 4u << 20;
std::mutex g_pool_mu;
std::vector<Ctx*> g_pool;

Ctx* ctx_acquire() {
  int d = 0;
  HIP_OR_DIE(hipGetDevice(&d));
  if (!dev_info().ok) die("device probe", SVBGPU_ENODEVICE);
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size(); ++i)
      if (g_pool[i]->device == d) {
        Ctx* c = g_pool[i];
        g_pool.erase(g_pool.begin() + i);
        return c;
      }
  }
  Ctx* c = new Ctx;
  c->device = d;
  HIP_OR_DIE(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIP_OR_DIE(hipMalloc(&c->d_res, 64));
  HIP_OR_DIE(hipHostMalloc(&c->h_res, 64, hipHostMallocDefault));
  HIP_OR_DIE(hipHostMalloc(&c->h_sums, 4 * kMegas, hipHostMallocDefault));
  for (int k = 0; k < 2; ++k) {
    HIP_OR_DIE(hipHostMalloc(&c->stage[k], kStageChunk, hipHostMallocDefault));
    HIP_OR_DIE(hipEventCreateWithFlags(&c->stage_ev[k], hipEventDisableTiming));
  }
  return c;
}
void ctx_release(Ctx* c) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  g_pool.push_back(c);
}
// Host-pointer calls are re-entrant: each has its own context (stream, buffers,
// staging chunks).  With the "host_serialize" option on, one such call at a
// time per DEVICE is inside the HIP runtime (a mutex per device ordinal; calls
// on different devices never wait for each other).
std::mutex g_host_api_mu[64];
struct HostSerial {
  std::mutex* mu = nullptr;
  explicit HostSerial(const Options& o) {
    int d = 0;
    if (o.host_serialize && hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) {
      mu = &g_host_api_mu[d];
      mu->lock();
    }
  }
  ~HostSerial() { if (mu) mu->unlock(); }
  HostSerial(const HostSerial&) = delete;
  HostSerial& operator=(const HostSerial&) = delete;
};
struct CtxGuard {
  Options o;
  HostSerial serial;
  Ctx* c;
  CtxGuard() : o(options()), serial(o), c(ctx_acquire()) {}
  ~CtxGuard() { ctx_release(c); }
};

// The reference-named functions take device pointers in place, but work on the
// context's own stream, which is ordered against nothing else: whatever the
// caller still has in flight on the device (the kernel producing the input, a
// memset of the output on any stream, the null stream included) must have ended
// before that stream touches the buffers.  The calls are synchronous anyway
// (they return a size), so a device-wide wait on entry costs nothing extra.
void order_after_callers_work(std::initializer_list<const void*> ptrs);

void grow(void*& p, size_t& cap, size_t need) {
  if (need <= cap) return;
  if (p) HIP_OR_DIE(hipFree(p));
  size_t n = need + need / 4 + 4096;
  HIP_OR_DIE(hipMalloc(&p, n));
  cap = n;
}

void grow_pinned(void*& p, size_t& cap, size_t need) {
  if (need <= cap) return;
  if (p) HIP_OR_DIE(hipHostFree(p));
  size_t n = need + need / 4 + 4096;
  HIP_OR_DIE(hipHostMalloc(&p, n, hipHostMallocDefault));
  cap = n;
}

bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();  // plain host memory: clear the sticky error
    return false;
  }
  return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

// Page-locked host memory HIP knows about (hipHostMalloc, hipHostRegister).
bool is_pinned_host(const void* p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

// Where a reference-named call runs.  Device pointers always go to the kernels
// (in place).  Host pointers go to the kernels too, staged over PCIe, unless the
// call is small -- below "host_cpu_cutoff" values the two PCIe crossings and the
// launch latency (~100 us) cost more than the host codec takes -- or the process
// has no gfx950 device, in which case the host codec is all there is (the
// reference never fails: include/streamvbyte.h:64 there).  The svbgpu_* API
// never takes this path.
// What the probe costs: hipPointerGetAttributes is a lock and a search of the runtime's allocation map,
// 0.1-0.3 us a pointer -- more than the host codec needs for a dozen values.  A process without any HIP
// device skips it (nothing can be a device pointer there; the count is asked once), and a caller that
// only ever passes host memory can say so: svbgpu_set_option("host_pointers_only", 1), or
// SVB_HOST_POINTERS_ONLY=1 in the environment of a program that cannot be changed.
bool process_has_devices() {
  static const bool any = [] {
    int n = 0;
    const bool ok = hipGetDeviceCount(&n) == hipSuccess && n > 0;
    if (!ok) (void)hipGetLastError();
    return ok;
  }();
  return any;
}
// Memory on the calling thread's own stack is host memory without asking anybody (the reference's
// own test programs keep their arrays there: tests/perf.c:48-50, tests/unit.c:93-96).
bool on_own_stack(const void* p) {
  struct Bounds { uintptr_t lo = 0, hi = 0; };
  thread_local const Bounds b = [] {
    Bounds r;
    pthread_attr_t at;
    if (pthread_getattr_np(pthread_self(), &at) == 0) {
      void* addr = nullptr;
      size_t size = 0;
      if (pthread_attr_getstack(&at, &addr, &size) == 0) {
        r.lo = (uintptr_t)addr;
        r.hi = r.lo + size;
      }
      pthread_attr_destroy(&at);
    }
    return r;
  }();
  const uintptr_t a = (uintptr_t)p;
  return a >= b.lo && a < b.hi;
}
bool host_pointers_only() {
  static const bool env = [] { const char* e = std::getenv("SVB_HOST_POINTERS_ONLY"); return e && *e && *e != '0'; }();
  return env || o_host_pointers_only.load(std::memory_order_relaxed) != 0;
}
bool run_on_host(size_t count, std::initializer_list<const void*> ptrs) {
  if (!process_has_devices()) return true;
  if (!host_pointers_only())
    for (const void* p : ptrs)
      if (!on_own_stack(p) && is_device_ptr(p)) return false;
  if (count < (size_t)o_host_cpu_cutoff.load(std::memory_order_relaxed)) return true;
  return !dev_info().ok;
}

// Host <-> device copies of the host-pointer API.  HIP never sees a pageable
// caller pointer: such memory moves through the context's own page-locked
// chunks with plain memcpy on this thread, two chunks in flight (option
// "host_stage_pageable"; an abort once seen with eight threads handing pageable
// pointers to hipMemcpyAsync could not be brought back with staging off, so
// this is a precaution).
// copy_in returns when the source has been read, copy_out when the destination
// holds the data; both keep the order of the context's stream.
bool g_stage_pageable() { return o_host_stage_pageable.load(std::memory_order_relaxed) != 0; }
int copy_in(Ctx* c, void* d_dst, const void* h_src, size_t bytes) {
  if (is_pinned_host(h_src) || !g_stage_pageable()) {
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
    return SVBGPU_OK;
  }
  size_t i = 0;
  for (size_t off = 0; off < bytes; off += kStageChunk, ++i) {
    const size_t len = std::min(kStageChunk, bytes - off);
    const int k = (int)(i & 1);
    if (i >= 2) HIP_TRY(hipEventSynchronize(c->stage_ev[k]));
    std::memcpy(c->stage[k], static_cast<const uint8_t*>(h_src) + off, len);
    HIP_TRY(hipMemcpyAsync(static_cast<uint8_t*>(d_dst) + off, c->stage[k], len, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipEventRecord(c->stage_ev[k], c->stream));
  }
  // the chunks are free again once the stream has passed them
  HIP_TRY(hipStreamSynchronize(c->stream));
  return SVBGPU_OK;
}
int copy_out(Ctx* c, void* h_dst, const void* d_src, size_t bytes) {
  if (is_pinned_host(h_dst) || !g_stage_pageable()) {
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SVBGPU_OK;
  }
  const size_t nchunks = (bytes + kStageChunk - 1) / kStageChunk;
  for (size_t i = 0; i <= nchunks; ++i) {
    if (i < nchunks) {
      const size_t off = i * kStageChunk, len = std::min(kStageChunk, bytes - off);
      HIP_TRY(hipMemcpyAsync(c->stage[i & 1], static_cast<const uint8_t*>(d_src) + off, len, hipMemcpyDeviceToHost,
                             c->stream));
      HIP_TRY(hipEventRecord(c->stage_ev[i & 1], c->stream));
    }
    if (i >= 1) {  // chunk i-1 has landed: hand it to the caller while chunk i is on its way
      const size_t off = (i - 1) * kStageChunk, len = std::min(kStageChunk, bytes - off);
      HIP_TRY(hipEventSynchronize(c->stage_ev[(i - 1) & 1]));
      std::memcpy(static_cast<uint8_t*>(h_dst) + off, c->stage[(i - 1) & 1], len);
    }
  }
  return SVBGPU_OK;
}
#define COPY_OR_DIE(expr)                 \
  do {                                    \
    int rc_ = (expr);                     \
    if (rc_) die("host copy", rc_);       \
  } while (0)

uint64_t fetch_u64(Ctx* c, const uint64_t* d_word) {
  HIP_OR_DIE(hipMemcpyAsync(c->h_res, d_word, 8, hipMemcpyDeviceToHost, c->stream));
  HIP_OR_DIE(hipStreamSynchronize(c->stream));
  return c->h_res[0];
}
void order_after_callers_work(std::initializer_list<const void*> ptrs) {
  for (const void* p : ptrs)
    if (is_device_ptr(p)) {
      HIP_OR_DIE(hipDeviceSynchronize());
      return;
    }
}
// Data bytes of the stream whose directory is in the context's workspace: the sum of its mega tile sums.
uint64_t fetch_total_data(Ctx* c, uint32_t count) {
  const WsLayout L = ws_layout(count);
  uint32_t* h = reinterpret_cast<uint32_t*>(c->h_sums);
  HIP_OR_DIE(hipMemcpyAsync(h, ws_at<u32>(c->ws, L.off_mega_sum), 4 * L.n_mega, hipMemcpyDeviceToHost, c->stream));
  HIP_OR_DIE(hipStreamSynchronize(c->stream));
  uint64_t total = 0;
  for (u32 m = 0; m < L.n_mega; ++m) total += h[m];
  return total;
}
void check_scan(Ctx* c) {
  if (o_debug.load(std::memory_order_relaxed) & 128u) {
    // diagnostic only: the status read as it was before the concurrency fix, an asynchronous copy into
    // a variable on the calling thread's stack (tools/stress_threads.py ... stack)
    u32 err = 0;
    HIP_OR_DIE(hipMemcpyAsync(&err, c->ws, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_OR_DIE(hipStreamSynchronize(c->stream));
    if (err) die("device scan", SVBGPU_ESCAN);
    return;
  }
  HIP_OR_DIE(hipMemcpyAsync(c->h_res + 2, c->ws, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_OR_DIE(hipStreamSynchronize(c->stream));
  if (reinterpret_cast<const uint32_t*>(c->h_res + 2)[0]) die("device scan", SVBGPU_ESCAN);
}

size_t encode_any(int variant, unsigned flags, const uint32_t* in, uint32_t count, uint8_t* out,
                  uint32_t prev) {
  if (count == 0) return 0;
  if (run_on_host(count, {in, out})) return svbcpu::encode_any(variant, flags & SVBGPU_DELTA, in, count, out, prev);
  CtxGuard g;
  Ctx* c = g.c;
  order_after_callers_work({in, out});
  grow(c->ws, c->ws_cap, svbgpu_workspace_bytes(count));
  const uint32_t* d_in = in;
  if (!is_device_ptr(in)) {
    grow(c->a, c->a_cap, (size_t)count * 4);
    COPY_OR_DIE(copy_in(c, c->a, in, (size_t)count * 4));
    d_in = static_cast<const uint32_t*>(c->a);
  }
  const bool out_dev = is_device_ptr(out);
  uint8_t* d_out = out;
  if (!out_dev) {
    grow(c->b, c->b_cap, streamvbyte_max_compressedbytes(count));
    d_out = static_cast<uint8_t*>(c->b);
  }
  int rc = svbgpu_encode(variant, flags, d_in, count, d_out, prev, c->ws, c->ws_cap, c->d_res, c->stream);
  if (rc) die("encode", rc);
  const uint64_t n = fetch_u64(c, c->d_res);
  check_scan(c);
  if (!out_dev) COPY_OR_DIE(copy_out(c, out, d_out, n));
  return (size_t)n;
}

size_t decode_any(int variant, unsigned flags, const uint8_t* in, uint32_t* out, uint32_t count,
                  uint32_t prev) {
  if (count == 0) return 0;
  if (run_on_host(count, {in, out})) return svbcpu::decode_any(variant, flags & SVBGPU_DELTA, in, out, count, prev);
  CtxGuard g;
  Ctx* c = g.c;
  order_after_callers_work({in, out});
  grow(c->ws, c->ws_cap, svbgpu_workspace_bytes(count));
  const size_t keylen = ((size_t)count + 3) / 4;
  const uint8_t* d_in = in;
  int rc;
  if (!is_device_ptr(in)) {
    // The stream's length is not known up front: stage the keys, let the
    // device sum them, then stage exactly the data bytes they promise.
    grow(c->a, c->a_cap, keylen + 16);
    COPY_OR_DIE(copy_in(c, c->a, in, keylen));
    rc = build_directory(variant, static_cast<const uint8_t*>(c->a), count, c->ws, c->stream);
    if (rc) die("decode (key scan)", rc);
    const uint64_t total = fetch_total_data(c, count);  // data bytes of the stream
    if (c->a_cap < keylen + total + 16) {
      grow(c->a, c->a_cap, keylen + total + 16);
      COPY_OR_DIE(copy_in(c, c->a, in, keylen + total));
    } else {
      COPY_OR_DIE(copy_in(c, static_cast<uint8_t*>(c->a) + keylen, in + keylen, total));
    }
    d_in = static_cast<const uint8_t*>(c->a);
  } else {
    rc = build_directory(variant, d_in, count, c->ws, c->stream);
    if (rc) die("decode (key scan)", rc);
  }
  const bool out_dev = is_device_ptr(out);
  uint32_t* d_out = out;
  if (!out_dev) {
    grow(c->b, c->b_cap, (size_t)count * 4);
    d_out = static_cast<uint32_t*>(c->b);
  }
  rc = run_decode(g.o, variant, flags, d_in, d_out, count, prev, c->ws, c->d_res, c->stream);
  if (rc) die("decode", rc);
  if (!out_dev) COPY_OR_DIE(copy_out(c, out, d_out, (size_t)count * 4));
  const uint64_t n = fetch_u64(c, c->d_res);
  if ((flags & SVBGPU_DELTA) && g.o.delta_lookback) check_scan(c);  // only the look-back decoder (which clears it first) uses the error word
  return (size_t)n;
}

size_t size_any(int variant, const uint32_t* in, uint32_t count) {
  if (count == 0) return 0;
  if (run_on_host(count, {in}))
    return variant == SVBGPU_1234 ? svbcpu::compressedbytes<svbcpu::k1234>(in, count) : svbcpu::compressedbytes<svbcpu::k0124>(in, count);
  CtxGuard g;
  Ctx* c = g.c;
  order_after_callers_work({in});
  const uint32_t* d_in = in;
  if (!is_device_ptr(in)) {
    grow(c->a, c->a_cap, (size_t)count * 4);
    COPY_OR_DIE(copy_in(c, c->a, in, (size_t)count * 4));
    d_in = static_cast<const uint32_t*>(c->a);
  }
  int rc = svbgpu_compressedbytes(variant, 0, d_in, count, 0, c->d_res, c->stream);
  if (rc) die("compressedbytes", rc);
  return (size_t)fetch_u64(c, c->d_res);
}

bool validate_any(int variant, const uint8_t* in, size_t in_length, uint32_t count) {
  // empty rules need no device work (reference src/streamvbyte_decode.c:91-92)
  if (in_length == 0 || count == 0) return in_length == (size_t)count;
  const size_t keylen = ((size_t)count + 3) / 4;
  if (keylen > in_length) return false;
  if (run_on_host(count, {in}))
    return variant == SVBGPU_1234 ? svbcpu::validate<svbcpu::k1234>(in, in_length, count) : svbcpu::validate<svbcpu::k0124>(in, in_length, count);
  CtxGuard g;
  Ctx* c = g.c;
  order_after_callers_work({in});
  grow(c->ws, c->ws_cap, sizeof(WsHeader));
  const uint8_t* d_in = in;
  if (!is_device_ptr(in)) {
    grow(c->a, c->a_cap, keylen + 16);
    COPY_OR_DIE(copy_in(c, c->a, in, keylen));
    d_in = static_cast<const uint8_t*>(c->a);
  }
  int rc = svbgpu_validate(variant, d_in, in_length, count, reinterpret_cast<uint32_t*>(c->d_res + 1),
                           c->ws, c->ws_cap, c->stream);
  if (rc) die("validate_stream", rc);
  HIP_OR_DIE(hipMemcpyAsync(c->h_res, c->d_res + 1, 8, hipMemcpyDeviceToHost, c->stream));
  HIP_OR_DIE(hipStreamSynchronize(c->stream));
  return (reinterpret_cast<uint32_t*>(c->h_res)[0]) != 0;
}

void zigzag_any(int mode, const void* in, void* out, size_t n, int32_t prev) {
  if (n == 0) return;
  if (run_on_host(n, {in, out})) return svbcpu::zigzag(mode, in, out, n, prev);
  CtxGuard g;
  Ctx* c = g.c;
  order_after_callers_work({in, out});
  const bool in_dev = is_device_ptr(in), out_dev = is_device_ptr(out);
  const void* d_in = in;
  void* d_out = out;
  if (!in_dev) {
    grow(c->a, c->a_cap, n * 4);
    COPY_OR_DIE(copy_in(c, c->a, in, n * 4));
    d_in = c->a;
  }
  if (!out_dev) {
    grow(c->b, c->b_cap, n * 4);
    d_out = c->b;
  }
  int rc;
  if (mode == 3) {
    // The device scan counts in 32 bits; a longer array (N is a size_t in the
    // reference, include/streamvbyte_zigzag.h:29) goes in pieces, each seeded with
    // the last value of the piece before it.
    const size_t piece = (size_t)g.o.zigzag_piece;
    grow(c->ws, c->ws_cap, svbgpu_workspace_bytes((uint32_t)std::min(n, piece)));
    rc = SVBGPU_OK;
    for (size_t off = 0; off < n && !rc; off += piece) {
      const size_t len = std::min(piece, n - off);
      if (off) {
        HIP_OR_DIE(hipMemcpyAsync(c->h_res + 3, static_cast<int32_t*>(d_out) + off - 1, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_OR_DIE(hipStreamSynchronize(c->stream));
        prev = reinterpret_cast<const int32_t*>(c->h_res + 3)[0];
        check_scan(c);
      }
      rc = svbgpu_zigzag_delta_decode(static_cast<const uint32_t*>(d_in) + off, static_cast<int32_t*>(d_out) + off,
                                      len, prev, c->ws, c->ws_cap, c->stream);
    }
  } else {
    rc = zigzag_launch(mode, static_cast<const uint32_t*>(d_in), static_cast<uint32_t*>(d_out), n,
                       (uint32_t)prev, c->stream);
  }
  if (rc) die("zigzag", rc);
  if (!out_dev) COPY_OR_DIE(copy_out(c, out, d_out, n * 4));
  HIP_OR_DIE(hipStreamSynchronize(c->stream));
  if (mode == 3) check_scan(c);
}

}  // namespace

extern "C" {

size_t streamvbyte_encode(const uint32_t* in, uint32_t length, uint8_t* out) {
  return encode_any(SVBGPU_1234, 0, in, length, out, 0);
}
size_t streamvbyte_encode_0124(const uint32_t* in, uint32_t length, uint8_t* out) {
  return encode_any(SVBGPU_0124, 0, in, length, out, 0);
}
size_t streamvbyte_compressedbytes(const uint32_t* in, uint32_t length) {
  return size_any(SVBGPU_1234, in, length);
}
size_t streamvbyte_compressedbytes_0124(const uint32_t* in, uint32_t length) {
  return size_any(SVBGPU_0124, in, length);
}
size_t streamvbyte_decode(const uint8_t* in, uint32_t* out, uint32_t length) {
  return decode_any(SVBGPU_1234, 0, in, out, length, 0);
}
size_t streamvbyte_decode_0124(const uint8_t* in, uint32_t* out, uint32_t length) {
  return decode_any(SVBGPU_0124, 0, in, out, length, 0);
}
bool streamvbyte_validate_stream(const uint8_t* in, size_t inLength, uint32_t outLength) {
  return validate_any(SVBGPU_1234, in, inLength, outLength);
}
bool streamvbyte_validate_stream_0124(const uint8_t* in, size_t inLength, uint32_t outLength) {
  return validate_any(SVBGPU_0124, in, inLength, outLength);
}
size_t streamvbyte_delta_encode(const uint32_t* in, uint32_t length, uint8_t* out, uint32_t prev) {
  return encode_any(SVBGPU_1234, SVBGPU_DELTA, in, length, out, prev);
}
size_t streamvbyte_delta_decode(const uint8_t* in, uint32_t* out, uint32_t length, uint32_t prev) {
  return decode_any(SVBGPU_1234, SVBGPU_DELTA, in, out, length, prev);
}
void zigzag_encode(const int32_t* in, uint32_t* out, size_t N) { zigzag_any(0, in, out, N, 0); }
void zigzag_decode(const uint32_t* in, int32_t* out, size_t N) { zigzag_any(1, in, out, N, 0); }
void zigzag_delta_encode(const int32_t* in, uint32_t* out, size_t N, int32_t prev) {
  zigzag_any(2, in, out, N, prev);
}
void zigzag_delta_decode(const uint32_t* in, int32_t* out, size_t N, int32_t prev) {
  zigzag_any(3, in, out, N, prev);
}

}  // extern "C"

#include "svb_frame.h"
#include "svb_listframe.h"


// =============================================================================
// Batches over the devices of one node (include/streamvbyte_multi.h)
// =============================================================================
struct svbmulti {
  int ndev = 0;
  int dev[SVBMULTI_MAX_DEVICES];
  hipStream_t stream[SVBMULTI_MAX_DEVICES];
  ncclComm_t comm[SVBMULTI_MAX_DEVICES];
  bool comms = false;
  // per device: this shard's sizes padded to the longest shard, everybody's padded sizes, block sums
  u64* pad[SVBMULTI_MAX_DEVICES];
  u64* gathered[SVBMULTI_MAX_DEVICES];
  u64* blk[SVBMULTI_MAX_DEVICES];
  size_t cap_longest = 0, cap_blocks = 0;
};

namespace {
Rccl g_rccl;
std::mutex g_rccl_mu;
thread_local std::string g_multi_error;

struct DeviceGuard {
  int prev = -1;
  DeviceGuard() { (void)hipGetDevice(&prev); }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int multi_fail_hip(hipError_t e, const char* what) {
  g_last_hip_error = e;
  g_multi_error = std::string(what) + ": " + hipGetErrorString(e);
  return SVBGPU_EHIP;
}
int multi_fail_rccl(ncclResult_t r, const char* what) {
  g_multi_error = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
  return SVBMULTI_ERCCL;
}
#define MULTI_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return multi_fail_hip(e_, #expr); } while (0)
#define MULTI_RCCL(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return multi_fail_rccl(r_, #expr); } while (0)

// Room for shards of up to `longest` streams and jobs of up to `blocks` blocks of 4096 streams.
int multi_reserve(svbmulti* m, size_t longest, size_t blocks) {
  if (longest <= m->cap_longest && blocks <= m->cap_blocks) return SVBGPU_OK;
  const size_t nl = std::max(longest, m->cap_longest), nb = std::max(blocks, m->cap_blocks);
  for (int i = 0; i < m->ndev; ++i) {
    MULTI_HIP(hipSetDevice(m->dev[i]));
    MULTI_HIP(hipStreamSynchronize(m->stream[i]));  // earlier calls may still read the old buffers
    if (m->pad[i]) (void)hipFree(m->pad[i]);
    if (m->gathered[i]) (void)hipFree(m->gathered[i]);
    if (m->blk[i]) (void)hipFree(m->blk[i]);
    m->pad[i] = m->gathered[i] = m->blk[i] = nullptr;
    MULTI_HIP(hipMalloc(&m->pad[i], 8 * std::max<size_t>(nl, 1)));
    MULTI_HIP(hipMalloc(&m->gathered[i], 8 * std::max<size_t>(nl, 1) * m->ndev));
    MULTI_HIP(hipMalloc(&m->blk[i], 8 * std::max<size_t>(nb, 1)));
  }
  m->cap_longest = nl;
  m->cap_blocks = nb;
  return SVBGPU_OK;
}
}  // namespace

extern "C" {

const char* svbmulti_last_error(void) { return g_multi_error.c_str(); }

int svbgpu_partition_streams(const uint32_t* counts, uint64_t nstreams, int world, uint64_t* cut) {
  if (world < 1 || !cut || (nstreams && !counts)) return SVBGPU_EINVAL;
  unsigned long long total = 0;
  for (uint64_t s = 0; s < nstreams; ++s) total += counts[s];
  // cut[r] = the first stream boundary at or past r / world of the values
  cut[0] = 0;
  uint64_t s = 0;
  unsigned long long before = 0;  // values in streams 0 .. s-1
  for (int r = 1; r < world; ++r) {
    const unsigned long long target = (unsigned long long)((unsigned __int128)total * (unsigned)r / (unsigned)world);
    while (s < nstreams && before < target) before += counts[s++];
    cut[r] = s;
  }
  cut[world] = nstreams;
  return SVBGPU_OK;
}

int svbmulti_create(const int* devices, int ndev, svbmulti_t** out) {
  if (!out) return SVBGPU_EINVAL;
  *out = nullptr;
  if (!devices || ndev < 1 || ndev > SVBMULTI_MAX_DEVICES) return SVBGPU_EINVAL;
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have < 1) return SVBGPU_ENODEVICE;
  for (int i = 0; i < ndev; ++i) {
    if (devices[i] < 0 || devices[i] >= have) return SVBGPU_EINVAL;
    for (int j = 0; j < i; ++j)
      if (devices[j] == devices[i]) return SVBGPU_EINVAL;
  }
  DeviceGuard guard;
  for (int i = 0; i < ndev; ++i) {
    MULTI_HIP(hipSetDevice(devices[i]));
    if (!dev_info().ok) return SVBGPU_ENODEVICE;
  }
  {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (!g_rccl.load(g_multi_error)) return SVBMULTI_ERCCL;
  }
  svbmulti* m = new svbmulti();
  m->ndev = ndev;
  for (int i = 0; i < ndev; ++i) {
    m->dev[i] = devices[i];
    m->stream[i] = nullptr;
    m->pad[i] = m->gathered[i] = m->blk[i] = nullptr;
  }
  auto fail = [&](int rc) { svbmulti_destroy(m); return rc; };
  for (int i = 0; i < ndev; ++i) {
    hipError_t e = hipSetDevice(m->dev[i]);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->stream[i], hipStreamNonBlocking);
    if (e != hipSuccess) return fail(multi_fail_hip(e, "hipStreamCreateWithFlags"));
  }
  const ncclResult_t r = g_rccl.CommInitAll(m->comm, ndev, m->dev);
  if (r != ncclSuccess) return fail(multi_fail_rccl(r, "ncclCommInitAll"));
  m->comms = true;
  *out = m;
  return SVBGPU_OK;
}

void svbmulti_destroy(svbmulti_t* m) {
  if (!m) return;
  DeviceGuard guard;
  for (int i = 0; i < m->ndev; ++i) {
    if (hipSetDevice(m->dev[i]) != hipSuccess) continue;
    if (m->stream[i]) (void)hipStreamSynchronize(m->stream[i]);
    if (m->comms) (void)g_rccl.CommDestroy(m->comm[i]);
    if (m->pad[i]) (void)hipFree(m->pad[i]);
    if (m->gathered[i]) (void)hipFree(m->gathered[i]);
    if (m->blk[i]) (void)hipFree(m->blk[i]);
    if (m->stream[i]) (void)hipStreamDestroy(m->stream[i]);
  }
  delete m;
}

int svbmulti_ndev(const svbmulti_t* m) { return m ? m->ndev : 0; }
void* svbmulti_stream(svbmulti_t* m, int i) { return m && i >= 0 && i < m->ndev ? m->stream[i] : nullptr; }

int svbmulti_synchronize(svbmulti_t* m) {
  if (!m) return SVBGPU_EINVAL;
  DeviceGuard guard;
  for (int i = 0; i < m->ndev; ++i) {
    MULTI_HIP(hipSetDevice(m->dev[i]));
    MULTI_HIP(hipStreamSynchronize(m->stream[i]));
  }
  return SVBGPU_OK;
}

int svbmulti_encode_batch(svbmulti_t* m, int variant, unsigned flags, const svbmulti_shard_t* shards) {
  if (!m || !shards || bad_variant(variant) || bad_flags(flags)) return SVBGPU_EINVAL;
  ShardStarts S{};
  S.n = (u32)m->ndev;
  size_t longest = 0;
  for (int i = 0; i < m->ndev; ++i) {
    if (!shards[i].d_sizes_all) return SVBGPU_EINVAL;
    S.start[i + 1] = S.start[i] + shards[i].nstreams;
    longest = std::max<size_t>(longest, shards[i].nstreams);
  }
  const u64 n_all = S.start[m->ndev];
  if (n_all >= (1ull << 32)) return SVBGPU_EINVAL;
  S.longest = (u32)std::max<size_t>(longest, 1);
  const unsigned blocks = (unsigned)((n_all + 4095) / 4096);
  DeviceGuard guard;
  int rc = multi_reserve(m, longest, blocks);
  if (rc) return rc;
  // 1. every shard on its own device and stream; sizes land in the padded send buffer
  for (int i = 0; i < m->ndev; ++i) {
    const svbmulti_shard_t& h = shards[i];
    MULTI_HIP(hipSetDevice(m->dev[i]));
    if (h.nstreams < S.longest) {
      rc = fill_u32(m->pad[i] + h.nstreams, 0u, 8 * (size_t)(S.longest - h.nstreams), m->stream[i]);
      if (rc) return rc;
    }
    rc = svbgpu_encode_batch(variant, flags, h.d_values, h.d_value_off, h.d_bytes, h.d_byte_off, h.d_counts, h.d_prev,
                             h.nstreams, m->pad[i], m->stream[i]);
    if (rc) return rc;
  }
  // 2. the one exchange: 8 bytes per stream, every device receives every shard's sizes
  MULTI_RCCL(g_rccl.GroupStart());
  for (int i = 0; i < m->ndev; ++i) {
    const ncclResult_t r = g_rccl.AllGather(m->pad[i], m->gathered[i], S.longest, ncclUint64, m->comm[i], m->stream[i]);
    if (r != ncclSuccess) {
      (void)g_rccl.GroupEnd();
      return multi_fail_rccl(r, "ncclAllGather");
    }
  }
  MULTI_RCCL(g_rccl.GroupEnd());
  // 3. job order without the padding, and the byte directory, on every device
  for (int i = 0; i < m->ndev && blocks; ++i) {
    MULTI_HIP(hipSetDevice(m->dev[i]));
    hipLaunchKernelGGL(k_multi_compact, dim3(blocks), dim3(1024), 0, m->stream[i], m->gathered[i], S, shards[i].d_sizes_all,
                       m->blk[i]);
    if (shards[i].d_offsets_all)
      hipLaunchKernelGGL(k_multi_offsets, dim3(blocks), dim3(1024), 0, m->stream[i], shards[i].d_sizes_all, n_all, m->blk[i],
                         shards[i].d_offsets_all);
    MULTI_HIP(hipGetLastError());
  }
  for (int i = 0; i < m->ndev && !blocks; ++i)  // an empty job still has a total
    if (shards[i].d_offsets_all) {
      MULTI_HIP(hipSetDevice(m->dev[i]));
      rc = fill_u32(shards[i].d_offsets_all, 0u, 8, m->stream[i]);
      if (rc) return rc;
    }
  return SVBGPU_OK;
}

int svbmulti_decode_batch(svbmulti_t* m, int variant, unsigned flags, const svbmulti_shard_t* shards) {
  if (!m || !shards || bad_variant(variant) || bad_flags(flags)) return SVBGPU_EINVAL;
  DeviceGuard guard;
  for (int i = 0; i < m->ndev; ++i) {
    const svbmulti_shard_t& h = shards[i];
    MULTI_HIP(hipSetDevice(m->dev[i]));
    const int rc = svbgpu_decode_batch(variant, flags, h.d_bytes, h.d_byte_off, const_cast<uint32_t*>(h.d_values), h.d_value_off,
                                       h.d_counts, h.d_prev, h.nstreams, nullptr, m->stream[i]);
    if (rc) return rc;
  }
  return SVBGPU_OK;
}

}  // extern "C"

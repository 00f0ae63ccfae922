// Frames (include/streamvbyte_frame.h): header handling on the host and a
// three-deep pipeline over consecutive frames in which the host-to-device copy,
// the kernels and the device-to-host copy of neighbouring frames overlap.
// Included at the end of svb_api.hip (uses its context pool).
//
// The reference has no container; its callers store count and length next to
// the stream themselves (tests/writeseq.c:19-46 there).
#pragma once

#include "../../include/streamvbyte_frame.h"

namespace {

constexpr uint32_t kFrameMagic = 0x46425653u;  // "SVBF"
constexpr int kFrameSlots = 4;
constexpr size_t kFrameLag = 3;  // < kFrameSlots + 1: a slot's stage 2 is queued before its next stage 1

struct FrameHdr {
  uint32_t magic;
  uint8_t version, variant, flags, zero8;
  uint32_t count, prev;
  uint64_t stream_bytes, zero64;
};
static_assert(sizeof(FrameHdr) == SVBFRAME_HEADER_BYTES, "frame header is 32 bytes");

inline uint64_t round16(uint64_t x) { return (x + 15) & ~(uint64_t)15; }

// kFrameSlots contexts with their buffers sized up front (growing a buffer
// later would free memory that queued work still uses).
struct FrameSlots {
  Ctx* c[kFrameSlots] = {};
  hipEvent_t ev[kFrameSlots] = {};    // result word of the slot's frame is back
  hipEvent_t up[kFrameSlots] = {};    // the slot's host-to-device copy has ended
  hipEvent_t down[kFrameSlots] = {};  // the slot's device-to-host copy has ended
  int n = 0;
  bool pinned_in = false, pinned_out = false;  // the caller's buffers are page-locked
  int init(size_t a_bytes, size_t b_bytes, uint32_t max_count) {
    for (n = 0; n < kFrameSlots; ++n) {
      c[n] = ctx_acquire();
      grow(c[n]->a, c[n]->a_cap, a_bytes);
      grow(c[n]->b, c[n]->b_cap, b_bytes);
      grow(c[n]->ws, c[n]->ws_cap, svbgpu_workspace_bytes(max_count));
      // the slot's sticky error word (d_res[7]) and the workspace header start clear
      if (fill_u32(c[n]->ws, 0u, sizeof(WsHeader), c[n]->stream) != SVBGPU_OK ||
          fill_u32(c[n]->d_res + 7, 0u, 8, c[n]->stream) != SVBGPU_OK) {
        ctx_release(c[n]);
        return SVBGPU_EHIP;
      }
      if (hipEventCreateWithFlags(&ev[n], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&up[n], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&down[n], hipEventDisableTiming) != hipSuccess) {
        ctx_release(c[n]);
        return SVBGPU_EHIP;
      }
    }
    return SVBGPU_OK;
  }
  // Copies of one direction run one after the other in frame order (each waits
  // for the previous frame's), so that a frame's upload ends early and its
  // kernels and download overlap the next frames' uploads; left alone, the
  // slots' copies share the link, end together and the two directions take turns.
  // Pageable caller memory goes through the slot's own page-locked chunks
  // (copy_in / copy_out of svb_api.hip, which block): correct, not overlapped.
  int copy_up(size_t i, void* dst, const void* src, size_t bytes) {
    const int s = (int)(i % kFrameSlots), p = (int)((i + kFrameSlots - 1) % kFrameSlots);
    if (!pinned_in) return copy_in(c[s], dst, src, bytes);
    if (i) HIP_TRY(hipStreamWaitEvent(c[s]->stream, up[p], 0));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c[s]->stream));
    HIP_TRY(hipEventRecord(up[s], c[s]->stream));
    return SVBGPU_OK;
  }
  int copy_down(size_t i, void* dst, const void* src, size_t bytes) {
    const int s = (int)(i % kFrameSlots), p = (int)((i + kFrameSlots - 1) % kFrameSlots);
    if (!pinned_out) return copy_out(c[s], dst, src, bytes);
    if (i) HIP_TRY(hipStreamWaitEvent(c[s]->stream, down[p], 0));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c[s]->stream));
    HIP_TRY(hipEventRecord(down[s], c[s]->stream));
    return SVBGPU_OK;
  }
  // Every look-back call of a frame begins by clearing its workspace header, error
  // word included, so a give-up in frame i would be gone by frame i + kFrameSlots:
  // right after a frame's kernels its error word is OR-ed into the slot's sticky
  // word, which nothing clears until the call ends.
  int keep_error(int s) {
    hipLaunchKernelGGL(k_keep_error, dim3(1), dim3(64), 0, c[s]->stream, reinterpret_cast<const u32*>(c[s]->ws),
                       reinterpret_cast<u32*>(c[s]->d_res + 7));
    HIP_TRY(hipGetLastError());
    return SVBGPU_OK;
  }
  int scan_status(int s) {  // the slot's sticky error word
    HIP_TRY(hipMemcpyAsync(c[s]->h_res + 2, c[s]->d_res + 7, 4, hipMemcpyDeviceToHost, c[s]->stream));
    HIP_TRY(hipStreamSynchronize(c[s]->stream));
    return reinterpret_cast<const uint32_t*>(c[s]->h_res + 2)[0] ? SVBGPU_ESCAN : SVBGPU_OK;
  }
  int drain() {
    int rc = SVBGPU_OK;
    for (int i = 0; i < n; ++i)
      if (hipStreamSynchronize(c[i]->stream) != hipSuccess) rc = SVBGPU_EHIP;
    return rc;
  }
  ~FrameSlots() {
    for (int i = 0; i < n; ++i) {
      (void)hipStreamSynchronize(c[i]->stream);
      (void)hipEventDestroy(ev[i]);
      (void)hipEventDestroy(up[i]);
      (void)hipEventDestroy(down[i]);
      ctx_release(c[i]);
    }
  }
};

}  // namespace

extern "C" {

size_t svbframe_max_bytes(uint32_t count) {
  return SVBFRAME_HEADER_BYTES + (size_t)round16(streamvbyte_max_compressedbytes(count));
}

size_t svbframe_max_bytes_many(size_t n, uint32_t frame_values) {
  if (n == 0 || frame_values == 0) return 0;
  const size_t full = n / frame_values, rest = n % frame_values;
  return full * svbframe_max_bytes(frame_values) + (rest ? svbframe_max_bytes((uint32_t)rest) : 0);
}

int svbframe_peek(const uint8_t* frame, size_t avail, svbframe_info* info) {
  if (!frame || !info) return SVBGPU_EINVAL;
  if (avail < SVBFRAME_HEADER_BYTES) return SVBFRAME_ETRUNCATED;
  FrameHdr h;
  std::memcpy(&h, frame, sizeof(h));
  if (h.magic != kFrameMagic || h.version != 1 || h.variant > 1 ||
      (h.flags & ~(SVBGPU_DELTA | SVBGPU_ZIGZAG)) || h.zero8 || h.zero64)
    return SVBFRAME_EBADFRAME;
  const uint64_t keylen = ((uint64_t)h.count + 3) / 4;
  // a stream of `count` values holds its keys and at most 4 bytes per value
  if (h.stream_bytes < (h.count ? keylen : 0) || h.stream_bytes > keylen + 4ull * h.count) return SVBFRAME_EBADFRAME;
  if (!(h.flags & SVBGPU_DELTA) && h.prev) return SVBFRAME_EBADFRAME;
  info->count = h.count;
  info->prev = h.prev;
  info->variant = h.variant;
  info->flags = h.flags;
  info->stream_bytes = h.stream_bytes;
  info->frame_bytes = SVBFRAME_HEADER_BYTES + round16(h.stream_bytes);
  if (info->frame_bytes > avail) {
    // the last frame of a buffer may come without its padding
    if (SVBFRAME_HEADER_BYTES + h.stream_bytes > avail) return SVBFRAME_ETRUNCATED;
    info->frame_bytes = avail;
  }
  return SVBFRAME_OK;
}

int svbframe_scan(const uint8_t* frames, size_t bytes, size_t* n_frames, size_t* n_values) {
  if (!frames && bytes) return SVBGPU_EINVAL;
  size_t off = 0, nf = 0, nv = 0;
  while (off < bytes) {
    svbframe_info inf;
    const int rc = svbframe_peek(frames + off, bytes - off, &inf);
    if (rc) return rc;
    off += (size_t)inf.frame_bytes;
    nv += inf.count;
    ++nf;
  }
  if (n_frames) *n_frames = nf;
  if (n_values) *n_values = nv;
  return SVBFRAME_OK;
}

void* svbframe_alloc_pinned(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
void svbframe_free_pinned(void* p) {
  if (p) (void)hipHostFree(p);
}

int svbframe_encode_many(const uint32_t* in, size_t n, uint32_t frame_values, int variant,
                         unsigned flags, uint8_t* out, size_t out_capacity, size_t* written) {
  if (written) *written = 0;
  if (bad_variant(variant) || bad_flags(flags) || !written || (n && (!in || !out || !frame_values)))
    return SVBGPU_EINVAL;
  if (n == 0) return SVBFRAME_OK;
  if (out_capacity < svbframe_max_bytes_many(n, frame_values)) return SVBFRAME_ECAPACITY;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  const size_t nframes = (n + frame_values - 1) / frame_values;
  const Options opt = options();
  HostSerial serial(opt);
  FrameSlots S;
  S.pinned_in = is_pinned_host(in);
  S.pinned_out = is_pinned_host(out);
  int rc = S.init((size_t)frame_values * 4, streamvbyte_max_compressedbytes(frame_values), frame_values);
  if (rc) return rc;

  auto count_of = [&](size_t i) -> uint32_t {
    return (uint32_t)std::min<size_t>(frame_values, n - i * frame_values);
  };
  // stage 1: values up, encode, size back
  auto stage1 = [&](size_t i) -> int {
    Ctx* c = S.c[i % kFrameSlots];
    const uint32_t cnt = count_of(i);
    const uint32_t* src = in + i * frame_values;
    const uint32_t prev = (flags & SVBGPU_DELTA) && i ? src[-1] : 0u;
    int r = S.copy_up(i, c->a, src, (size_t)cnt * 4);
    if (r) return r;
    r = svbgpu_encode(variant, flags, static_cast<const uint32_t*>(c->a), cnt, static_cast<uint8_t*>(c->b),
                          prev, c->ws, c->ws_cap, c->d_res, c->stream);
    if (r) return r;
    r = S.keep_error((int)(i % kFrameSlots));
    if (r) return r;
    HIP_TRY(hipMemcpyAsync(c->h_res, c->d_res, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(S.ev[i % kFrameSlots], c->stream));
    return SVBGPU_OK;
  };
  // stage 2: the size is known, so is the frame's place: header by the host, payload down
  size_t off = 0;
  auto stage2 = [&](size_t i) -> int {
    Ctx* c = S.c[i % kFrameSlots];
    HIP_TRY(hipEventSynchronize(S.ev[i % kFrameSlots]));
    const uint64_t sz = c->h_res[0];
    const uint32_t cnt = count_of(i);
    FrameHdr h = {};
    h.magic = kFrameMagic;
    h.version = 1;
    h.variant = (uint8_t)variant;
    h.flags = (uint8_t)flags;
    h.count = cnt;
    h.prev = (flags & SVBGPU_DELTA) && i ? in[i * frame_values - 1] : 0u;
    h.stream_bytes = sz;
    std::memcpy(out + off, &h, sizeof(h));
    const int r = S.copy_down(i, out + off + SVBFRAME_HEADER_BYTES, c->b, sz);
    if (r) return r;
    const uint64_t padded = round16(sz);
    std::memset(out + off + SVBFRAME_HEADER_BYTES + sz, 0, padded - sz);
    off += SVBFRAME_HEADER_BYTES + padded;
    return SVBGPU_OK;
  };
  // the host stays kFrameLag frames ahead of the results it waits for
  for (size_t i = 0; i < nframes + kFrameLag && !rc; ++i) {
    if (i < nframes) rc = stage1(i);
    if (!rc && i >= kFrameLag) rc = stage2(i - kFrameLag);
  }
  const int rd = S.drain();
  if (!rc) rc = rd;
  for (int s = 0; s < S.n && !rc; ++s) rc = S.scan_status(s);
  if (!rc) *written = off;
  return rc;
}

int svbframe_decode_many(const uint8_t* frames, size_t bytes, uint32_t* out, size_t out_capacity,
                         size_t* n_values) {
  if (!n_values || (!frames && bytes)) return SVBGPU_EINVAL;
  *n_values = 0;
  struct Ref { size_t off, voff; svbframe_info inf; };
  std::vector<Ref> fr;
  size_t off = 0, total = 0;
  uint64_t max_stream = 0;
  uint32_t max_count = 0;
  while (off < bytes) {
    Ref r;
    const int rc = svbframe_peek(frames + off, bytes - off, &r.inf);
    if (rc) return rc;
    r.off = off;
    r.voff = total;
    off += (size_t)r.inf.frame_bytes;
    total += r.inf.count;
    max_stream = std::max(max_stream, r.inf.stream_bytes);
    max_count = std::max(max_count, r.inf.count);
    if (r.inf.count) fr.push_back(r);
  }
  *n_values = total;
  if (!out) return SVBFRAME_OK;
  if (total > out_capacity) return SVBFRAME_ECAPACITY;
  if (fr.empty()) return SVBFRAME_OK;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  const Options opt = options();
  HostSerial serial(opt);
  FrameSlots S;
  S.pinned_in = is_pinned_host(frames);
  S.pinned_out = is_pinned_host(out);
  int rc = S.init((size_t)max_stream + 16, (size_t)max_count * 4, max_count);
  if (rc) return rc;

  // stage 1: payload up, validate against the header, verdict back
  auto stage1 = [&](size_t i) -> int {
    Ctx* c = S.c[i % kFrameSlots];
    const Ref& f = fr[i];
    int r = S.copy_up(i, c->a, frames + f.off + SVBFRAME_HEADER_BYTES, f.inf.stream_bytes);
    if (r) return r;
    r = svbgpu_validate(f.inf.variant, static_cast<const uint8_t*>(c->a), f.inf.stream_bytes, f.inf.count,
                            reinterpret_cast<uint32_t*>(c->d_res + 1), c->ws, c->ws_cap, c->stream);
    if (r) return r;
    HIP_TRY(hipMemcpyAsync(c->h_res + 1, c->d_res + 1, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(S.ev[i % kFrameSlots], c->stream));
    return SVBGPU_OK;
  };
  // stage 2: only a payload that keeps its promises is decoded
  auto stage2 = [&](size_t i) -> int {
    Ctx* c = S.c[i % kFrameSlots];
    const Ref& f = fr[i];
    HIP_TRY(hipEventSynchronize(S.ev[i % kFrameSlots]));
    if (reinterpret_cast<const uint32_t*>(c->h_res + 1)[0] != 1u) return SVBFRAME_EINVALID;
    int r = build_directory(f.inf.variant, static_cast<const uint8_t*>(c->a), f.inf.count, c->ws, c->stream);
    if (r) return r;
    r = run_decode(opt, f.inf.variant, f.inf.flags, static_cast<const uint8_t*>(c->a), static_cast<uint32_t*>(c->b),
                   f.inf.count, f.inf.prev, c->ws, nullptr, c->stream);
    if (r) return r;
    if ((f.inf.flags & SVBGPU_DELTA) && opt.delta_lookback) {  // the only decoder that waits on other workgroups
      r = S.keep_error((int)(i % kFrameSlots));
      if (r) return r;
    }
    return S.copy_down(i, out + f.voff, c->b, (size_t)f.inf.count * 4);
  };
  for (size_t i = 0; i < fr.size() + kFrameLag && !rc; ++i) {
    if (i < fr.size()) rc = stage1(i);
    if (!rc && i >= kFrameLag) rc = stage2(i - kFrameLag);
  }
  const int rd = S.drain();
  if (!rc) rc = rd;
  for (int s = 0; s < S.n && !rc; ++s) rc = S.scan_status(s);
  return rc;
}

}  // extern "C"

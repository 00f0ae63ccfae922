// List frames (include/streamvbyte_frame.h): many independent lists per frame
// behind a directory of counts and sizes, coded and decoded by the batched
// kernels, through the same three-deep pipeline as the single-stream frames.
// Included at the end of svb_api.hip, after svb_frame.h.
//
// The reference has neither a container nor a batch call; its callers loop over
// streamvbyte_encode / streamvbyte_decode list by list and keep counts and
// lengths themselves (tests/writeseq.c:19-46, tests/unit.c:99-173 there).
#pragma once

namespace {

constexpr uint32_t kListMagic = 0x4C425653u;  // "SVBL"

struct ListHdr {
  uint32_t magic;
  uint8_t version, variant, flags, zero8;
  uint32_t n_lists, zero32;
  uint64_t n_values, payload_bytes;
};
static_assert(sizeof(ListHdr) == SVBFRAME_HEADER_BYTES, "list frame header is 32 bytes");

inline size_t list_dir_bytes(uint32_t n_lists) { return 2 * (size_t)round16(4ull * n_lists); }

// A frame's directory as the kernels want it, the same layout on the host
// (page-locked image, the first `upload` bytes) and on the device.
struct ListDir {
  uint64_t* value_off;   // first value of list j, in elements
  uint64_t* byte_off;    // first byte of list j
  uint32_t* counts;
  uint32_t* sizes32;     // decode: from the frame; encode: made by k_scan_sizes
  uint64_t* sizes;       // encode only, device only: bytes written per list
  uint64_t* packed_off;  // encode only, device only
  static size_t even(size_t n) { return (n + 1) & ~(size_t)1; }
  static size_t upload_bytes(size_t n, bool with_sizes) { return 16 * n + 4 * even(n) * (with_sizes ? 2 : 1); }
  static size_t device_bytes(size_t n) { return 32 * n + 8 * even(n); }
  ListDir(void* base, size_t n) {
    uint8_t* p = static_cast<uint8_t*>(base);
    value_off = reinterpret_cast<uint64_t*>(p);
    byte_off = value_off + n;
    counts = reinterpret_cast<uint32_t*>(byte_off + n);
    sizes32 = counts + even(n);
    sizes = reinterpret_cast<uint64_t*>(sizes32 + even(n));
    packed_off = sizes + n;
  }
};

// Header of the list frame at `frame`; *frame_bytes = offset of the next frame.
int list_peek(const uint8_t* frame, size_t avail, ListHdr* h, size_t* frame_bytes) {
  if (avail < SVBFRAME_HEADER_BYTES) return SVBFRAME_ETRUNCATED;
  std::memcpy(h, frame, sizeof(*h));
  if (h->magic != kListMagic || h->version != 1 || h->variant > 1 ||
      (h->flags & ~(SVBGPU_DELTA | SVBGPU_ZIGZAG)) || h->zero8 || h->zero32 || h->n_lists > SVBFRAME_MAX_LISTS)
    return SVBFRAME_EBADFRAME;
  if (h->n_values > (uint64_t)h->n_lists * SVBFRAME_LIST_MAX_VALUES) return SVBFRAME_EBADFRAME;
  // at most 4 data bytes per value, a key byte per 4 values and a partly used key byte per list
  if (h->payload_bytes > 4 * h->n_values + h->n_values / 4 + h->n_lists) return SVBFRAME_EBADFRAME;
  const uint64_t head = SVBFRAME_HEADER_BYTES + list_dir_bytes(h->n_lists);
  uint64_t fb = head + round16(h->payload_bytes);
  if (fb > avail) {  // the last frame of a buffer may come without its padding
    if (head + h->payload_bytes > avail) return SVBFRAME_ETRUNCATED;
    fb = avail;
  }
  *frame_bytes = (size_t)fb;
  return SVBFRAME_OK;
}

// The directory must add up to the header and every size must be possible for
// its count; whether a list's keys agree with its size is the device's check.
// Fills the directory image (offsets inside the frame) and, if asked,
// list_start[0 .. n_lists) = first_value + running count.
int list_read_dir(const uint8_t* frame, const ListHdr& h, ListDir* img, uint64_t* list_start, uint64_t first_value) {
  const uint8_t* pc = frame + SVBFRAME_HEADER_BYTES;
  const uint8_t* ps = pc + round16(4ull * h.n_lists);
  uint64_t nv = 0, nb = 0;
  for (uint32_t j = 0; j < h.n_lists; ++j) {
    uint32_t c, s;
    std::memcpy(&c, pc + 4ull * j, 4);
    std::memcpy(&s, ps + 4ull * j, 4);
    const uint64_t keylen = ((uint64_t)c + 3) / 4;
    if (c > SVBFRAME_LIST_MAX_VALUES || s < keylen || s > keylen + 4ull * c) return SVBFRAME_EBADFRAME;
    if (img) {
      img->value_off[j] = nv;
      img->byte_off[j] = nb;
      img->counts[j] = c;
      img->sizes32[j] = s;
    }
    if (list_start) list_start[j] = first_value + nv;
    nv += c;
    nb += s;
  }
  return nv == h.n_values && nb == h.payload_bytes ? SVBFRAME_OK : SVBFRAME_EBADFRAME;
}

unsigned list_grid(uint32_t n_lists) {
  return stream_grid(((unsigned long long)n_lists + kWaves - 1) / kWaves, 8);
}

// The slots of the frame pipeline with the buffers list frames add.
int list_slots_init(FrameSlots& S, size_t a_bytes, size_t b_bytes, size_t p_bytes, size_t max_lists, bool dir_with_sizes) {
  const int rc = S.init(std::max<size_t>(a_bytes, 16), std::max<size_t>(b_bytes, 16), 1);
  if (rc) return rc;
  for (int s = 0; s < S.n; ++s) {
    if (p_bytes) grow(S.c[s]->p, S.c[s]->p_cap, p_bytes);
    grow(S.c[s]->dir, S.c[s]->dir_cap, ListDir::device_bytes(max_lists));
    grow_pinned(S.c[s]->hdir, S.c[s]->hdir_cap, ListDir::upload_bytes(max_lists, dir_with_sizes));
  }
  return SVBGPU_OK;
}

}  // namespace

extern "C" {

size_t svbframe_lists_max_bytes(size_t n_values, size_t n_lists, uint32_t frame_values) {
  if (n_lists == 0 || frame_values == 0) return 0;
  // Two neighbouring frames hold more than frame_values values together unless
  // the first was closed for holding SVBFRAME_MAX_LISTS lists.
  const size_t frames = 2 * (n_values / frame_values) + n_lists / SVBFRAME_MAX_LISTS + 2;
  // header and three paddings per frame; 8 directory bytes, a partly used key byte per list
  return frames * (SVBFRAME_HEADER_BYTES + 48) + 9 * n_lists + 4 * n_values + n_values / 4 + 1;
}

int svbframe_encode_lists(const uint32_t* values, const uint64_t* list_start, size_t n_lists,
                          uint32_t frame_values, int variant, unsigned flags, uint8_t* out,
                          size_t out_capacity, size_t* written) {
  if (written) *written = 0;
  if (bad_variant(variant) || bad_flags(flags) || !written || (n_lists && (!list_start || !out || !frame_values)))
    return SVBGPU_EINVAL;
  if (n_lists == 0) return SVBFRAME_OK;
  // worst: device bytes when every list gets streamvbyte_max_compressedbytes; tight: the most the packed payload can be
  struct Part { size_t first; uint32_t n_lists; uint64_t n_values, worst, tight; };
  std::vector<Part> parts;
  Part cur = {0, 0, 0, 0, 0};
  for (size_t i = 0; i < n_lists; ++i) {
    if (list_start[i + 1] < list_start[i]) return SVBGPU_EINVAL;
    const uint64_t c = list_start[i + 1] - list_start[i];
    if (c > SVBFRAME_LIST_MAX_VALUES) return SVBGPU_EINVAL;
    if (cur.n_lists && (cur.n_lists == SVBFRAME_MAX_LISTS || cur.n_values + c > frame_values)) {
      parts.push_back(cur);
      cur = Part{i, 0, 0, 0, 0};
    }
    ++cur.n_lists;
    cur.n_values += c;
    cur.worst += streamvbyte_max_compressedbytes((uint32_t)c);
    cur.tight += (c + 3) / 4 + 4 * c;
  }
  parts.push_back(cur);
  if (list_start[n_lists] != list_start[0] && !values) return SVBGPU_EINVAL;
  size_t need = 0, max_values = 0, max_worst = 0, max_lists = 0;
  for (const Part& f : parts) {
    need += SVBFRAME_HEADER_BYTES + list_dir_bytes(f.n_lists) + (size_t)round16(f.tight);
    max_values = std::max<size_t>(max_values, f.n_values);
    max_worst = std::max<size_t>(max_worst, f.worst);
    max_lists = std::max<size_t>(max_lists, f.n_lists);
  }
  if (out_capacity < need) return SVBFRAME_ECAPACITY;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  const Options opt = options();
  HostSerial serial(opt);
  FrameSlots S;
  S.pinned_in = values && is_pinned_host(values);
  S.pinned_out = is_pinned_host(out);
  int rc = list_slots_init(S, max_values * 4, max_worst, std::max<size_t>(max_worst, 16), max_lists, false);
  if (rc) return rc;

  // stage 1: values and directory up, every list coded into a place that fits its
  // worst case, sizes scanned, streams moved to their packed places, total back
  auto stage1 = [&](size_t i) -> int {
    Ctx* c = S.c[i % kFrameSlots];
    const Part& f = parts[i];
    ListDir h(c->hdir, f.n_lists), d(c->dir, f.n_lists);
    const uint64_t base = list_start[f.first];
    uint64_t worst = 0;
    for (uint32_t j = 0; j < f.n_lists; ++j) {
      const uint64_t cnt = list_start[f.first + j + 1] - list_start[f.first + j];
      h.value_off[j] = list_start[f.first + j] - base;
      h.byte_off[j] = worst;
      h.counts[j] = (uint32_t)cnt;
      worst += streamvbyte_max_compressedbytes((uint32_t)cnt);
    }
    int r = f.n_values ? S.copy_up(i, c->a, values + base, (size_t)f.n_values * 4) : SVBGPU_OK;
    if (r) return r;
    HIP_TRY(hipMemcpyAsync(c->dir, c->hdir, ListDir::upload_bytes(f.n_lists, false), hipMemcpyHostToDevice, c->stream));
    r = svbgpu_encode_batch(variant, flags, static_cast<const uint32_t*>(c->a), d.value_off, static_cast<uint8_t*>(c->b),
                            d.byte_off, d.counts, nullptr, f.n_lists, d.sizes, c->stream);
    if (r) return r;
    hipLaunchKernelGGL(k_scan_sizes, dim3(1), dim3(kBlock), 0, c->stream, d.sizes, f.n_lists, d.packed_off, d.sizes32,
                       c->d_res);
    hipLaunchKernelGGL(k_pack_lists, dim3(list_grid(f.n_lists)), dim3(kBlock), 0, c->stream,
                       static_cast<const uint8_t*>(c->b), d.byte_off, static_cast<uint8_t*>(c->p), d.packed_off, d.sizes,
                       f.n_lists);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->h_res, c->d_res, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(S.ev[i % kFrameSlots], c->stream));
    return SVBGPU_OK;
  };
  // stage 2: the payload's size is known, so is the frame's place
  size_t off = 0;
  auto stage2 = [&](size_t i) -> int {
    Ctx* c = S.c[i % kFrameSlots];
    const Part& f = parts[i];
    ListDir d(c->dir, f.n_lists);
    HIP_TRY(hipEventSynchronize(S.ev[i % kFrameSlots]));
    const uint64_t payload = c->h_res[0];
    ListHdr h = {};
    h.magic = kListMagic;
    h.version = 1;
    h.variant = (uint8_t)variant;
    h.flags = (uint8_t)flags;
    h.n_lists = f.n_lists;
    h.n_values = f.n_values;
    h.payload_bytes = payload;
    uint8_t* o = out + off;
    std::memcpy(o, &h, sizeof(h));
    const size_t used = 4 * (size_t)f.n_lists, padded = (size_t)round16(used);
    uint8_t* pc = o + SVBFRAME_HEADER_BYTES;
    for (uint32_t j = 0; j < f.n_lists; ++j) {
      const uint32_t cnt = (uint32_t)(list_start[f.first + j + 1] - list_start[f.first + j]);
      std::memcpy(pc + 4 * (size_t)j, &cnt, 4);
    }
    std::memset(pc + used, 0, padded - used);
    uint8_t* ps = pc + padded;
    int r = S.copy_down(i, ps, d.sizes32, used);
    if (r) return r;
    std::memset(ps + used, 0, padded - used);
    uint8_t* pp = ps + padded;
    if (payload) {
      r = S.copy_down(i, pp, c->p, (size_t)payload);
      if (r) return r;
    }
    std::memset(pp + payload, 0, (size_t)(round16(payload) - payload));
    off += SVBFRAME_HEADER_BYTES + 2 * padded + (size_t)round16(payload);
    return SVBGPU_OK;
  };
  for (size_t i = 0; i < parts.size() + kFrameLag && !rc; ++i) {
    if (i < parts.size()) rc = stage1(i);
    if (!rc && i >= kFrameLag) rc = stage2(i - kFrameLag);
  }
  const int rd = S.drain();
  if (!rc) rc = rd;
  if (!rc) *written = off;
  return rc;
}

int svbframe_scan_lists(const uint8_t* frames, size_t bytes, size_t* n_frames, size_t* n_lists,
                        size_t* n_values) {
  if (!frames && bytes) return SVBGPU_EINVAL;
  size_t off = 0, nf = 0, nl = 0, nv = 0;
  while (off < bytes) {
    ListHdr h;
    size_t fb = 0;
    const int rc = list_peek(frames + off, bytes - off, &h, &fb);
    if (rc) return rc;
    off += fb;
    nl += h.n_lists;
    nv += (size_t)h.n_values;
    ++nf;
  }
  if (n_frames) *n_frames = nf;
  if (n_lists) *n_lists = nl;
  if (n_values) *n_values = nv;
  return SVBFRAME_OK;
}

int svbframe_decode_lists(const uint8_t* frames, size_t bytes, uint32_t* values, size_t values_capacity,
                          uint64_t* list_start, size_t lists_capacity, size_t* n_lists, size_t* n_values) {
  if (!n_lists || !n_values || (!frames && bytes)) return SVBGPU_EINVAL;
  *n_lists = *n_values = 0;
  struct Ref { size_t off, voff, loff; ListHdr h; };
  std::vector<Ref> fr;
  size_t off = 0, tl = 0, tv = 0, max_payload = 0, max_values = 0, max_lists = 0;
  while (off < bytes) {
    Ref r;
    size_t fb = 0;
    const int rc = list_peek(frames + off, bytes - off, &r.h, &fb);
    if (rc) return rc;
    r.off = off;
    r.voff = tv;
    r.loff = tl;
    off += fb;
    tl += r.h.n_lists;
    tv += (size_t)r.h.n_values;
    max_payload = std::max<size_t>(max_payload, r.h.payload_bytes);
    max_values = std::max<size_t>(max_values, r.h.n_values);
    max_lists = std::max<size_t>(max_lists, r.h.n_lists);
    fr.push_back(r);
  }
  *n_lists = tl;
  *n_values = tv;
  if (!values) return SVBFRAME_OK;
  if (!list_start) return SVBGPU_EINVAL;
  if (tv > values_capacity || tl > lists_capacity) return SVBFRAME_ECAPACITY;
  // every directory is read before the device sees anything
  for (const Ref& f : fr) {
    const int rc = list_read_dir(frames + f.off, f.h, nullptr, list_start + f.loff, f.voff);
    if (rc) return rc;
  }
  list_start[tl] = tv;
  // frames of empty lists are done
  fr.erase(std::remove_if(fr.begin(), fr.end(), [](const Ref& f) { return f.h.n_values == 0; }), fr.end());
  if (fr.empty()) return SVBFRAME_OK;
  if (!dev_info().ok) return SVBGPU_ENODEVICE;
  const Options opt = options();
  HostSerial serial(opt);
  FrameSlots S;
  S.pinned_in = is_pinned_host(frames);
  S.pinned_out = is_pinned_host(values);
  int rc = list_slots_init(S, max_payload + 16, max_values * 4, 0, max_lists, true);
  if (rc) return rc;

  // stage 1: directory and payload up, every list checked against its entry, verdict back
  auto stage1 = [&](size_t i) -> int {
    Ctx* c = S.c[i % kFrameSlots];
    const Ref& f = fr[i];
    ListDir h(c->hdir, f.h.n_lists), d(c->dir, f.h.n_lists);
    int r = list_read_dir(frames + f.off, f.h, &h, nullptr, 0);
    if (r) return r;
    const uint8_t* payload = frames + f.off + SVBFRAME_HEADER_BYTES + list_dir_bytes(f.h.n_lists);
    r = S.copy_up(i, c->a, payload, (size_t)f.h.payload_bytes);
    if (r) return r;
    HIP_TRY(hipMemcpyAsync(c->dir, c->hdir, ListDir::upload_bytes(f.h.n_lists, true), hipMemcpyHostToDevice, c->stream));
    uint32_t* d_ok = reinterpret_cast<uint32_t*>(c->d_res + 1);
    r = fill_u32(d_ok, 1u, 4, c->stream);
    if (r) return r;
    if (f.h.variant == SVBGPU_1234)
      hipLaunchKernelGGL((k_validate_lists<V1234, u32, false>), dim3(list_grid(f.h.n_lists)), dim3(kBlock), 0, c->stream,
                         static_cast<const uint8_t*>(c->a), d.byte_off, d.counts, d.sizes32, f.h.n_lists, d_ok);
    else
      hipLaunchKernelGGL((k_validate_lists<V0124, u32, false>), dim3(list_grid(f.h.n_lists)), dim3(kBlock), 0, c->stream,
                         static_cast<const uint8_t*>(c->a), d.byte_off, d.counts, d.sizes32, f.h.n_lists, d_ok);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->h_res + 1, c->d_res + 1, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(S.ev[i % kFrameSlots], c->stream));
    return SVBGPU_OK;
  };
  // stage 2: only lists that keep their promises are decoded
  auto stage2 = [&](size_t i) -> int {
    Ctx* c = S.c[i % kFrameSlots];
    const Ref& f = fr[i];
    ListDir d(c->dir, f.h.n_lists);
    HIP_TRY(hipEventSynchronize(S.ev[i % kFrameSlots]));
    if (reinterpret_cast<const uint32_t*>(c->h_res + 1)[0] != 1u) return SVBFRAME_EINVALID;
    const int r = svbgpu_decode_batch(f.h.variant, f.h.flags, static_cast<const uint8_t*>(c->a), d.byte_off,
                                      static_cast<uint32_t*>(c->b), d.value_off, d.counts, nullptr, f.h.n_lists,
                                      nullptr, c->stream);
    if (r) return r;
    return S.copy_down(i, values + f.voff, c->b, (size_t)f.h.n_values * 4);
  };
  for (size_t i = 0; i < fr.size() + kFrameLag && !rc; ++i) {
    if (i < fr.size()) rc = stage1(i);
    if (!rc && i >= kFrameLag) rc = stage2(i - kFrameLag);
  }
  const int rd = S.drain();
  if (!rc) rc = rd;
  return rc;
}

}  // extern "C"

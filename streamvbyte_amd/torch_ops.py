"""torch.ops.svb: the device-pointer API of libstreamvbyte.so.2 as PyTorch custom
operators on device tensors.

    import streamvbyte_amd.torch_ops          # registers the operators
    stream, size = torch.ops.svb.encode(values, variant, flags, prev)
    values       = torch.ops.svb.decode(stream, count, variant, flags, prev)
    ok           = torch.ops.svb.validate(stream, in_length, count, variant)
    size         = torch.ops.svb.compressedbytes(values, variant, flags, prev)

`values` is an int32 tensor holding the uint32 bits (torch has no uint32 arithmetic),
`stream` a uint8 tensor readable 16 bytes past the stream (what encode returns is),
`variant` 0 = 1234 / 1 = 0124, `flags` a bit set of DELTA = 1 and ZIGZAG = 2 as in
include/streamvbyte_gpu.h.  Everything is enqueued on torch's current stream; results
the reference returns by value (include/streamvbyte.h:23,64 there) come back as
one-element int64 tensors so that nothing synchronises.  The operators only move
pointers: all arithmetic is the HIP library's, and without it (or without a gfx950
device) they raise.
"""
import torch

from . import DeviceCodec, streamvbyte_max_compressedbytes


def _codec(device, count):
    """A workspace and result words of this call's own, allocated on torch's current stream.

    The library's workspace (offset directory, look-back words, size words) may be used by one call at a
    time.  A workspace kept per thread or per device would be shared by calls that one thread issues on
    different torch streams without synchronising, and they would overwrite each other's directory.
    torch's caching allocator is stream-aware: a block allocated while stream s is current is handed
    out again on another stream only after the work s had queued when it was freed has finished, so a
    fresh `torch.empty` per call is exactly the ownership the library asks for, at the cost of a cache
    lookup (no hipMalloc in steady state).  Nothing needs zeroing: every call clears what it reads."""
    return DeviceCodec(max(count, 1), torch.device("cuda", device.index if device.index is not None
                                                   else torch.cuda.current_device()), zero=False)


def _check_values(values):
    if not (values.is_cuda and values.dtype == torch.int32 and values.is_contiguous() and values.dim() == 1):
        raise ValueError("values: a contiguous 1-D int32 device tensor (the uint32 bits)")


@torch.library.custom_op("svb::encode", mutates_args=(), device_types="cuda")
def encode(values: torch.Tensor, variant: int, flags: int, prev: int) -> tuple[torch.Tensor, torch.Tensor]:
    _check_values(values)
    c = _codec(values.device, values.numel())
    out = torch.empty(streamvbyte_max_compressedbytes(values.numel()), dtype=torch.uint8, device=values.device)
    _, size = c.encode(values, out=out, variant=variant, flags=flags, prev=prev)
    return out, size


@encode.register_fake
def _(values, variant, flags, prev):
    n = values.numel()
    return (values.new_empty(streamvbyte_max_compressedbytes(n), dtype=torch.uint8),
            values.new_empty(1, dtype=torch.int64))


@torch.library.custom_op("svb::decode", mutates_args=(), device_types="cuda")
def decode(stream: torch.Tensor, count: int, variant: int, flags: int, prev: int) -> torch.Tensor:
    if not (stream.is_cuda and stream.dtype == torch.uint8 and stream.is_contiguous()):
        raise ValueError("stream: a contiguous uint8 device tensor, readable 16 bytes past the stream")
    c = _codec(stream.device, count)
    out, _ = c.decode(stream, count, variant=variant, flags=flags, prev=prev)
    return out


@decode.register_fake
def _(stream, count, variant, flags, prev):
    return stream.new_empty(count, dtype=torch.int32)


@torch.library.custom_op("svb::compressedbytes", mutates_args=(), device_types="cuda")
def compressedbytes(values: torch.Tensor, variant: int, flags: int, prev: int) -> torch.Tensor:
    _check_values(values)
    return _codec(values.device, values.numel()).compressedbytes(values, variant=variant, flags=flags, prev=prev)


@compressedbytes.register_fake
def _(values, variant, flags, prev):
    return values.new_empty(1, dtype=torch.int64)


@torch.library.custom_op("svb::validate", mutates_args=(), device_types="cuda")
def validate(stream: torch.Tensor, in_length: int, count: int, variant: int) -> torch.Tensor:
    c = _codec(stream.device, count)
    return c.validate(stream, in_length, count, variant=variant) & 1


@validate.register_fake
def _(stream, in_length, count, variant):
    return stream.new_empty(1, dtype=torch.int64)

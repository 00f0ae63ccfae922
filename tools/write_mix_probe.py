#!/usr/bin/env python3
"""What the memory system gives for write-heavy streams (torch's own element-wise kernels, 2^28 elements):
a pure fill, a uint8 -> int32 conversion (1 B read per 4 B written, the byte mix of a dense differential
decode), an int32 copy (4 B read per 4 B written) and an int32 -> uint8 narrowing (4 B read per 1 B written)."""
import torch
dev = "cuda:0"
n = 1 << 28
x8 = torch.ones(n, dtype=torch.uint8, device=dev)
x32 = torch.ones(n, dtype=torch.int32, device=dev)
y32 = torch.empty(n, dtype=torch.int32, device=dev)
y8 = torch.empty(n, dtype=torch.uint8, device=dev)
def t(f, nbytes, name, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:34s} {ms:.4f} ms  {nbytes / ms / 1e9:6.2f} TB/s  ({nbytes / ms / 1e9 / 8:.3f} of 8 TB/s)")
t(lambda: y32.fill_(7), 4 * n, "fill int32 (write only)")
t(lambda: y32.copy_(x8), 5 * n, "uint8 -> int32 (1 B in, 4 B out)")
t(lambda: y32.copy_(x32), 8 * n, "int32 copy (4 B in, 4 B out)")
t(lambda: y8.copy_(x32), 5 * n, "int32 -> uint8 (4 B in, 1 B out)")
t(lambda: x32.sum(), 4 * n, "sum int32 (read only)")

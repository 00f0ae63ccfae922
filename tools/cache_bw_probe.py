import torch, time
dev = "cuda:0"
def bw(n_elems, iters=30):
    x = torch.ones(n_elems, dtype=torch.int32, device=dev)
    y = torch.empty_like(x)
    for _ in range(3): s = x.sum()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): s = x.sum()
    e1.record(); torch.cuda.synchronize()
    rd = 4 * n_elems * iters / (e0.elapsed_time(e1) * 1e-3) / 1e12
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    cp = 8 * n_elems * iters / (e0.elapsed_time(e1) * 1e-3) / 1e12
    print(f"{4*n_elems/2**20:8.0f} MiB  read-only {rd:6.2f} TB/s   copy {cp:6.2f} TB/s (read+write)")
for lg in (22, 23, 24, 25, 26, 27, 28, 29):
    bw(1 << lg)

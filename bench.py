#!/usr/bin/env python3
"""StreamVByte on MI355X: device-resident decode / encode throughput.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2n 28] [--quick]

Workload (every rank): one stream of 2^28 uint32 whose byte widths are uniform
over {1,2,3,4} (value uniform inside its width class), seed fixed per rank.
The stream is encoded on the GPU, checked, and then decoded K times in the
timed region: one step = one full streamvbyte_decode of the stream through the
C ABI (svbgpu_decode_plan + svbgpu_decode_run, i.e. every kernel of the
decode path).  `value` = 4*n*K*ranks / time in GB/s on the uint32 output.
With N > 1 every rank decodes its own independent stream (streams are the
sharding unit, nothing is exchanged on the data path); the batched posting-list
workload with its RCCL all-gather of sizes is reported under "batched".

One JSON line is printed by rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=28)
    ap.add_argument("--quick", action="store_true", help="skip the secondary workloads and the CPU baseline")
    ap.add_argument("--event-every", type=int, default=4,
                    help="one timed step in this many carries the HIP events that time k_decode and the directory")
    ap.add_argument("--lists", type=int, default=1 << 20, help="posting lists in the batched workload (whole job)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the ranks (nccl = RCCL; gloo only with --stub)")
    ap.add_argument("--force-dist", action="store_true",
                    help="join a process group even with one rank, so that --gpus 1 runs the RCCL branch of the "
                         "batched workload (all-gather of per-list sizes) like the multi-GPU job does")
    ap.add_argument("--stub", action="store_true",
                    help="rank plumbing only: every rank times a host memcpy instead of the GPU workload "
                         "(no GPU needed; used by the CPU test of the launcher)")
    return ap.parse_args()


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher around it: start N fresh
    rank processes of this same script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    in their environment, exactly what torch.distributed.run sets) and wait for
    them.  This parent never touches the GPU -- it has not even imported torch --
    so no process that initialised HIP is ever replaced or forked.  Rank 0
    prints the one JSON line on the stdout the children inherit."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
        s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # Poll: a rank that dies (a HIP failure on one GPU, say) would otherwise leave the others waiting in
    # a barrier until the collective's own timeout, and its error buried; the first failure ends the job.
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = abs(code)
                for q in live:
                    q.terminate()
        if live:
            time.sleep(0.05)
    return rc


def gen_width_mix(torch, n, seed, device):
    """uint32 with byte width uniform over {1,2,3,4}; returns an int32 tensor holding the bits."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty(n, dtype=torch.int32, device=device)
    chunk = 1 << 26
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        w = torch.randint(1, 5, (m,), generator=g, device=device, dtype=torch.int64)
        lo = torch.where(w == 1, torch.zeros_like(w), torch.ones_like(w) << (8 * (w - 1)))
        hi = torch.ones_like(w) << (8 * w)
        r = torch.rand(m, generator=g, device=device, dtype=torch.float64)
        v = lo + (r * (hi - lo).to(torch.float64)).to(torch.int64)
        v = torch.minimum(v, hi - 1)
        out[s:s + m] = torch.where(v >= (1 << 31), v - (1 << 32), v).to(torch.int32)
        del w, lo, hi, r, v
    return out


def time_steps(torch, fn, steps, warmup, dist=None):
    """barrier + sync, K steps, sync + barrier; returns seconds (max over ranks)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if dist is not None:
        dist.barrier()
        t = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return t1 - t0


def event_ms(torch, fn, iters, warmup=2):
    """mean milliseconds of fn() over `iters` back-to-back calls, one pair of HIP events on the current
    stream around all of them (an event record between the calls is a queue packet of its own and
    costs each call about 3 microseconds); three such runs, the middle one is returned too"""
    for _ in range(warmup):
        fn()
    runs = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        runs.append(a.elapsed_time(b) / iters)
    runs.sort()
    return float(np.mean(runs)), float(runs[1])


def cpu_share():
    """CPUs this process may use for the baseline: its affinity set, at most 16 (a one-GPU box is a
    16-CPU share of a much larger machine: os.cpu_count() there counts CPUs that belong to others)."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    return cpus[:16]


def cpu_baseline(values_host):
    """The reference library (or, where it was not built, the oracle port) on the host cores, on the
    same 2^28-value stream the GPU decodes: one thread, then P equal independent streams on P pinned
    threads.  Median of 10 runs after a warm-up run, time.perf_counter (CLOCK_MONOTONIC)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import Oracle, Reference, max_compressedbytes
    try:
        cpu = Reference()
    except FileNotFoundError:
        cpu = Oracle()
    n = values_host.size

    def enc(v, out):
        return cpu._encode(0, v, v.size, out, False, 0)

    def dec(s, out, m):
        return cpu._decode(0, s, out, m, False, 0)

    comp = np.zeros(max_compressedbytes(n), dtype=np.uint8)
    size = enc(values_host, comp)
    back = np.zeros(n, dtype=np.uint32)
    dec(comp, back, n)
    assert np.array_equal(back, values_host)

    def median_of(f, reps, warm=1):
        for _ in range(warm):
            f()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), float(min(ts))

    reps = 10
    t_dec, t_dec_best = median_of(lambda: dec(comp, back, n), reps)
    t_enc, t_enc_best = median_of(lambda: enc(values_host, comp), reps)

    # all cores: P equal independent streams, one pinned thread each (ctypes drops the GIL)
    cpus = cpu_share()
    P = len(cpus)
    per = n // P
    parts = [np.ascontiguousarray(values_host[i * per:(i + 1) * per]) for i in range(P)]
    comps = [np.zeros(max_compressedbytes(per), dtype=np.uint8) for _ in range(P)]
    outs = [np.zeros(per, dtype=np.uint32) for _ in range(P)]
    for i in range(P):
        enc(parts[i], comps[i])

    def run_all(job):
        def pinned(i):
            try:
                os.sched_setaffinity(0, {cpus[i]})  # the calling thread only
            except (AttributeError, OSError):
                pass
            job(i)
        ths = [threading.Thread(target=pinned, args=(i,)) for i in range(P)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()

    t_dec_all, _ = median_of(lambda: run_all(lambda i: dec(comps[i], outs[i], per)), reps)
    t_enc_all, _ = median_of(lambda: run_all(lambda i: enc(parts[i], comps[i])), reps)
    # the reference's own benchmark shape (tests/perf.c:52-75 there): 500 000 values
    # v = r31 >> (s & 31), cache-resident, 100 trials of encode then 100 of decode, equality check
    rng = np.random.default_rng(1)
    c1 = (rng.integers(0, 1 << 31, size=500000, dtype=np.uint64) >> (rng.integers(0, 1 << 31, size=500000, dtype=np.uint64) & 31)).astype(np.uint32)
    c1_comp = np.zeros(max_compressedbytes(c1.size), dtype=np.uint8)
    c1_size = enc(c1, c1_comp)
    c1_back = np.zeros(c1.size, dtype=np.uint32)
    t_c1_enc, _ = median_of(lambda: enc(c1, c1_comp), 100)
    t_c1_dec, _ = median_of(lambda: dec(c1_comp, c1_back, c1.size), 100)
    assert np.array_equal(c1_back, c1)
    gb = 4.0 * n / 1e9
    dbar = (size - (n + 3) // 4) / n
    return {
        "value": round(gb / t_dec, 3), "unit": "GB/s (uint32 out, decode)", "cores": 1, "kind": cpu.kind,
        "sample": f"the whole 2^{int(np.log2(n))}-value stream of the workload ({size} B, DRAM-resident), "
                  f"median of {reps} runs after a warm-up run",
        "decode_Gints_per_s": round(n / t_dec / 1e9, 3), "decode_best_GBps": round(gb / t_dec_best, 3),
        "decode_algorithmic_GBps": round((4.25 + dbar) * n / t_dec / 1e9, 3),
        "encode_GBps": round(gb / t_enc, 3), "encode_Gints_per_s": round(n / t_enc / 1e9, 3),
        "all_cores": {"cores": P, "cores_visible": os.cpu_count(), "pinned": True,
                      "decode_GBps": round(4.0 * per * P / 1e9 / t_dec_all, 3),
                      "decode_Gints_per_s": round(per * P / 1e9 / t_dec_all, 3),
                      "encode_GBps": round(4.0 * per * P / 1e9 / t_enc_all, 3),
                      "note": f"{P} equal independent streams of 2^{int(np.log2(per))} values, one pinned thread each; "
                              "P = this process's CPU affinity, at most the 16-CPU share of a one-GPU box"},
        "perf_c_500k": {"decode_GBps": round(4.0 * c1.size / 1e9 / t_c1_dec, 3),
                        "encode_GBps": round(4.0 * c1.size / 1e9 / t_c1_enc, 3),
                        "decode_Gints_per_s": round(c1.size / 1e9 / t_c1_dec, 3),
                        "encode_Gints_per_s": round(c1.size / 1e9 / t_c1_enc, 3),
                        "stream_bytes": int(c1_size), "cores": 1,
                        "note": "500 000 values r31 >> (s & 31), cache-resident, median of 100 after a warm-up run"},
    }


def list_lengths(nlists, seed):
    """lengths log-uniform on [1,4096], the same array on every rank"""
    u = np.random.default_rng(seed).uniform(0.0, np.log(4096.0), size=nlists)
    return np.clip(np.floor(np.exp(u)), 1, 4096).astype(np.int64)


def gen_posting_lists(torch, counts_np, seed, device):
    """sorted lists with small gaps (1-2 byte deltas) for the given lengths"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    counts = torch.from_numpy(counts_np).to(device)
    nlists = counts.numel()
    voff = torch.zeros(nlists + 1, dtype=torch.int64, device=device)
    voff[1:] = torch.cumsum(counts, 0)
    total = int(voff[-1].item())
    gaps = torch.randint(0, 1 << 12, (total,), generator=g, device=device, dtype=torch.int64)
    small = torch.rand(total, generator=g, device=device) < 0.7
    gaps = torch.where(small, gaps & 0xFF, gaps)
    cs = torch.cumsum(gaps, 0)
    # restart the running sum at every list start
    starts = voff[:-1]
    first = cs[starts] - gaps[starts]
    vals = cs - torch.repeat_interleave(first, counts)
    del cs, gaps, small
    vals32 = torch.where(vals >= (1 << 31), vals - (1 << 32), vals).to(torch.int32)
    return counts.to(torch.int32), voff, vals32


def stub_rank(a, rank, world):
    """The rank plumbing of the real run (process group, barriers around the timed
    region, max over ranks, one JSON line from rank 0) around a host memcpy.
    Measures nothing about the codec; exists so that the N-rank path can be
    exercised without GPUs."""
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(a.backend, rank=rank, world_size=world)
    n = 1 << min(a.log2n, 20)
    src, dst = np.arange(n, dtype=np.uint32), np.empty(n, dtype=np.uint32)
    for _ in range(a.warmup):
        np.copyto(dst, src)
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        np.copyto(dst, src)
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    ranks_seen = torch.tensor([1], dtype=torch.int64)
    if dist:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(ranks_seen)
    t = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "host memcpy GB/s (launcher stub, not the codec)", "stub": True,
                          "value": round(4.0 * n * a.steps * world / t / 1e9, 3), "unit": "GB/s", "n_gpus": world,
                          "ranks_seen": int(ranks_seen.item()), "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": round(t / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                          "config": {"workload": "launcher stub"}}), flush=True)
    if dist:
        dist.destroy_process_group()
    return 0


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a)  # before anything touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.stub:
        return stub_rank(a, rank, world)
    if a.backend != "nccl":
        sys.exit("--backend gloo is for --stub only: the codec runs on GPUs, ranks talk over RCCL")
    import torch
    import streamvbyte_amd as svb

    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        torch.cuda.set_device(local_rank)
        dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        dist = dist_mod
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    n = 1 << a.log2n
    codec = svb.DeviceCodec(n, device)

    # ---- workload ------------------------------------------------------------
    values = gen_width_mix(torch, n, 0x5356425F43320001 + rank, device)
    stream, size_t = codec.encode(values)
    codec.check()
    size = int(size_t.item())
    keylen = (n + 3) // 4
    dbar = (size - keylen) / n
    out = torch.empty(n, dtype=torch.int32, device=device)

    # parity outside the timed region: round trip, sizes, validate, oracle on a slice
    dec, used = codec.decode(stream, n, out=out)
    parity = bool(torch.equal(dec, values)) and int(used.item()) == size
    parity &= int(codec.compressedbytes(values).item()) == size
    parity &= bool(int(codec.validate(stream, size, n).item()) & 1)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import Oracle
    m = min(n, 1 << 22)
    sl = values[:m].cpu().numpy().view(np.uint32)
    ref_stream = Oracle().encode(0, sl)
    got, got_size = codec.encode(values[:m].contiguous())
    parity &= int(got_size.item()) == ref_stream.size and bool(
        np.array_equal(got[: ref_stream.size].cpu().numpy(), ref_stream))
    codec.check()

    # ---- timed region: K full decodes ---------------------------------------
    ev = {"plan": [], "run": []}

    # Every `event_every`-th step carries HIP events (each record is a packet of its own in the queue and
    # costs the step about a microsecond); the others are the two launches and nothing else.
    step_no = [0]

    def step():
        i = step_no[0]
        step_no[0] += 1
        if i % a.event_every:
            codec.decode_plan(stream, n)
            codec.decode_run(stream, n, out)
            return
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        codec.decode_plan(stream, n)
        e1.record()
        codec.decode_run(stream, n, out)
        e2.record()
        ev["plan"].append((e0, e1))
        ev["run"].append((e1, e2))

    def step_warm():
        codec.decode_plan(stream, n)
        codec.decode_run(stream, n, out)

    for _ in range(a.warmup):
        step_warm()
    t = time_steps(torch, step, a.steps, 0, dist)
    run_ms = float(np.mean([x.elapsed_time(y) for x, y in ev["run"]]))
    plan_ms = float(np.mean([x.elapsed_time(y) for x, y in ev["plan"]]))
    value = 4.0 * n * a.steps * world / t / 1e9
    alg_bytes = (4.25 + dbar) * n  # keys 0.25 + data dbar + output 4 bytes per value
    achieved = alg_bytes / (run_ms * 1e-3) / 1e9

    # HBM bytes per k_decode launch from the committed PMC passes (same kernel,
    # same 2^28 workload; rocprofv3 counters cannot be collected from inside this process)
    traffic, traffic_src = None, None
    if a.log2n == 28:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1:]:
            kernels = json.load(open(path))["kernels"]
            # the plain 1234 decoder the timed region launches (template arguments: variant 0, no zigzag,
            # all-zero-key shortcut on, integer output, non-temporal loads and stores as a 2^28-value
            # stream gets them)
            k = (kernels.get("k_decode<0, false, true, 0, 3>") or kernels.get("k_decode<0, false, true, 0>")
                 or kernels.get("k_decode<0, false>"))
            if k:
                traffic, traffic_src = k["hbm_bytes"], os.path.relpath(path, ROOT)

    res = {
        "metric": "device-resident decode GB/s (uint32 out)",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(t / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"single 2^{a.log2n}-integer stream per GPU, byte widths uniform over 1-4, "
                               "streamvbyte_decode (1234), bit-exact",
                   "values_per_stream": n, "stream_bytes": size, "data_bytes_per_value": round(dbar, 4),
                   "streams": world, "parallelism": f"one independent stream per GPU x{world}"},
        "decode_Gints_per_s": round(n * a.steps * world / t / 1e9, 2),
        "parity": "bit-exact" if parity else "MISMATCH",
        "roofline": {"bound": "hbm", "kernel": "k_decode<1234>", "achieved": round(achieved, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": int(alg_bytes),
                     "kernel_ms": round(run_ms, 4), "directory_ms": round(plan_ms, 4),
                     # the timed step (directory + decoder) against the same algorithmic bytes: the
                     # directory's second read of the keys is traffic, not work
                     "whole_step_frac": round(alg_bytes / (t / a.steps) / 1e9 / HBM_PEAK_GBS, 4)},
    }

    # ---- measured ceiling: a streaming kernel that moves the same number of bytes -------------
    # the library's own element-wise map (16-byte loads and stores, a workgroup per chunk like
    # k_decode, trivial arithmetic) on alg_bytes / 8 values: alg_bytes / 2 read + alg_bytes / 2 written
    m_copy = int(alg_bytes // 8)
    cp_src = torch.empty(m_copy, dtype=torch.int32, device=device)
    cp_dst = torch.empty(m_copy, dtype=torch.int32, device=device)
    cp_ms, _ = event_ms(torch, lambda: codec.zigzag(cp_src, "decode", out=cp_dst), 10)
    copy_gbps = 8.0 * m_copy / (cp_ms * 1e-3) / 1e9
    res["roofline"]["copy_ceiling"] = {"GBps": round(copy_gbps, 1), "of_peak": round(copy_gbps / HBM_PEAK_GBS, 4),
                                       "kernel": "k_zigzag<decode>: element-wise map, 16 B per lane, same bytes moved",
                                       "ms": round(cp_ms, 4)}
    res["roofline"]["frac_of_copy"] = round(achieved / copy_gbps, 4)
    del cp_src, cp_dst

    # ---- encode, same stream --------------------------------------------------
    enc_out = torch.empty(svb.streamvbyte_max_compressedbytes(n), dtype=torch.uint8, device=device)
    # the encoders take some 20 ms of back-to-back calls to settle (tools/warmup_curve.py: 0.357 -> 0.337 ms over
    # the first 60 calls after an idle moment; the decoder is level from its first call), hence the longer warm-up
    ENC_WARMUP = 60
    te = time_steps(torch, lambda: codec.encode(values, out=enc_out), max(3, a.steps // 2), ENC_WARMUP, dist)
    codec.check()
    ke = max(3, a.steps // 2)
    res["encode"] = {"value": round(4.0 * n * ke * world / te / 1e9, 2), "unit": "GB/s (uint32 in)",
                     "Gints_per_s": round(n * ke * world / te / 1e9, 2), "ms": round(te / ke * 1e3, 4),
                     "roofline_frac": round(alg_bytes / (te / ke) / 1e9 / HBM_PEAK_GBS, 4), "warmup_calls": ENC_WARMUP}

    if not a.quick:
        extra = {}
        it = 5
        # C3 dense postings: gaps U[0,31], all 1-byte deltas, prev = 0
        g = torch.Generator(device=device)
        g.manual_seed(7 + rank)
        n3 = n
        post = torch.cumsum(torch.randint(0, 32, (n3,), generator=g, device=device, dtype=torch.int64), 0)
        post = torch.where(post >= (1 << 31), post - (1 << 32), post).to(torch.int32)
        s3, z3 = codec.encode(post, variant=svb.V1234, flags=svb.DELTA, prev=0)
        sz3 = int(z3.item())
        d3 = (sz3 - keylen) / n3
        ms_e, _ = event_ms(torch, lambda: codec.encode(post, out=enc_out, variant=svb.V1234, flags=svb.DELTA), it, warmup=ENC_WARMUP)
        ms_d, _ = event_ms(torch, lambda: codec.decode(s3, n3, out=out, flags=svb.DELTA), it)
        codec.check()
        ok3 = bool(torch.equal(out, post))
        extra["delta_postings"] = {"n": n3, "data_bytes_per_value": round(d3, 4), "parity": ok3,
                                   "decode_GBps": round(4.0 * n3 / ms_d / 1e6, 1), "encode_GBps": round(4.0 * n3 / ms_e / 1e6, 1),
                                   "decode_roofline_frac": round((4.25 + d3) * n3 / ms_d / 1e6 / HBM_PEAK_GBS, 4),
                                   "encode_roofline_frac": round((4.25 + d3) * n3 / ms_e / 1e6 / HBM_PEAK_GBS, 4)}
        # the same postings intersected with a sorted array (every 8th value of theirs) while they are decoded
        pu = post.to(torch.int64) & 0xFFFFFFFF
        sorted_b = torch.unique(pu[::8])
        expect_hits = post[torch.isin(pu, sorted_b)]
        sorted_b = torch.where(sorted_b >= (1 << 31), sorted_b - (1 << 32), sorted_b).to(torch.int32)
        del pu
        ms_i, _ = event_ms(torch, lambda: codec.decode_intersect(s3, n3, sorted_b, out=out, variant=svb.V1234), it)
        _, n_hits, _ = codec.decode_intersect(s3, n3, sorted_b, out=out, variant=svb.V1234)
        codec.check()
        k_hits = int(n_hits.item())
        extra["delta_postings"]["intersect"] = {
            "sorted_entries": int(sorted_b.numel()), "hits": k_hits, "ms": round(ms_i, 4),
            "Gvalues_per_s": round(n3 / ms_i / 1e6, 1),
            "parity": bool(k_hits == expect_hits.numel() and torch.equal(out[:k_hits], expect_hits))}
        # ... and with that array delta-coded too: neither side is ever written out decoded
        sb3, zb3 = codec.encode(sorted_b, variant=svb.V1234, flags=svb.DELTA, prev=0)
        sb3 = sb3[: int(zb3.item()) + 16].clone()
        nb3 = int(sorted_b.numel())
        ms_x, _ = event_ms(torch, lambda: codec.intersect_streams(s3, n3, sb3, nb3, out=out), it)
        _, n_hits = codec.intersect_streams(s3, n3, sb3, nb3, out=out)
        codec.check()
        k_hits = int(n_hits.item())
        extra["delta_postings"]["intersect_streams"] = {
            "values_b": nb3, "hits": k_hits, "ms": round(ms_x, 4), "Gvalues_per_s": round((n3 + nb3) / ms_x / 1e6, 1),
            "parity": bool(k_hits == expect_hits.numel() and torch.equal(out[:k_hits], expect_hits))}
        del post, s3, sorted_b, expect_hits, sb3
        # C3 mixed: 1-3 byte gaps, running sum wraps modulo 2^32, prev != 0; at 2^26 values and at the workload's size
        def mixed_gaps(n3b):
            r31 = torch.randint(0, 1 << 31, (n3b,), generator=g, device=device, dtype=torch.int64)
            sh = torch.randint(0, 32, (n3b,), generator=g, device=device, dtype=torch.int64)
            gaps = (r31 >> sh) >> 8
            del r31, sh
            post = (torch.cumsum(gaps, 0) + 0xDEADBEEF) & 0xFFFFFFFF
            del gaps
            post = torch.where(post >= (1 << 31), post - (1 << 32), post).to(torch.int32)
            s3, z3 = codec.encode(post, variant=svb.V1234, flags=svb.DELTA, prev=0xDEADBEEF)
            s3 = s3[: int(z3.item()) + 16].clone()
            d3b = (int(z3.item()) - (n3b + 3) // 4) / n3b
            o3 = out[:n3b]
            enc = lambda: codec.encode(post, out=enc_out, variant=svb.V1234, flags=svb.DELTA, prev=0xDEADBEEF)
            ms_e, _ = event_ms(torch, enc, it, warmup=ENC_WARMUP)
            codec.lib.svbgpu_set_option(b"encode_pipeline", 2)         # the pipelined encoder for differential input too
            ms_ep, _ = event_ms(torch, enc, it, warmup=ENC_WARMUP)
            same = bool(torch.equal(enc_out[: s3.numel() - 16], s3[:-16]))
            codec.lib.svbgpu_set_option(b"encode_pipeline", 3)
            codec.lib.svbgpu_set_option(b"delta_decode_lookback", 0)   # the two-pass decoder, for comparison
            ms_d2, _ = event_ms(torch, lambda: codec.decode(s3, n3b, out=o3, flags=svb.DELTA, prev=0xDEADBEEF), it)
            codec.lib.svbgpu_set_option(b"delta_decode_lookback", 1)
            ms_d, _ = event_ms(torch, lambda: codec.decode(s3, n3b, out=o3, flags=svb.DELTA, prev=0xDEADBEEF), it)
            codec.lib.svbgpu_set_option(b"scan_waves", 8)   # smaller tiles, every wave tile's bytes requested up front
            ms_d8w, _ = event_ms(torch, lambda: codec.decode(s3, n3b, out=o3, flags=svb.DELTA, prev=0xDEADBEEF), it)
            codec.lib.svbgpu_set_option(b"scan_waves", 16)
            codec.check()
            frac = lambda ms: round((4.25 + d3b) * n3b / ms / 1e6 / HBM_PEAK_GBS, 4)
            return {"n": n3b, "data_bytes_per_value": round(d3b, 4), "prev": "0xDEADBEEF",
                    "parity": bool(torch.equal(o3, post)) and same,
                    "decode_two_pass_GBps": round(4.0 * n3b / ms_d2 / 1e6, 1),
                    "decode_GBps": round(4.0 * n3b / ms_d / 1e6, 1), "encode_GBps": round(4.0 * n3b / ms_e / 1e6, 1),
                    "encode_pipelined_GBps": round(4.0 * n3b / ms_ep / 1e6, 1),
                    "decode_roofline_frac": frac(ms_d), "encode_roofline_frac": frac(ms_e),
                    "decode_scan_waves8_roofline_frac": frac(ms_d8w),
                    "encode_pipelined_roofline_frac": frac(ms_ep)}
        extra["delta_mixed_gaps"] = mixed_gaps(min(n, 1 << 26))
        if n > (1 << 26):
            extra["delta_mixed_gaps_full_size"] = mixed_gaps(n)
        # 64-bit values (the 1248 variant; no reference counterpart): widths 1, 2, 4, 8 bytes equally likely
        n8 = min(n, 1 << 27)
        w8 = torch.randint(0, 4, (n8,), generator=g, device=device, dtype=torch.int64)
        r8 = torch.randint(0, 1 << 62, (n8,), generator=g, device=device, dtype=torch.int64)
        v8 = r8 >> (62 - torch.minimum(8 << w8, torch.full_like(w8, 62)))
        del w8, r8
        s8, z8 = codec.encode64(v8)
        s8 = s8.clone()
        d8 = (int(z8.item()) - (n8 + 3) // 4) / n8
        o8 = torch.empty(n8, dtype=torch.int64, device=device)
        e8 = torch.empty((n8 + 3) // 4 + 8 * n8 + 32, dtype=torch.uint8, device=device)
        ms_e8, _ = event_ms(torch, lambda: codec.encode64(v8, out=e8), it, warmup=ENC_WARMUP)
        ms_d8, _ = event_ms(torch, lambda: codec.decode64(s8, n8, out=o8), it)
        codec.check()
        extra["wide_1248"] = {"n": n8, "data_bytes_per_value": round(d8, 4), "parity": bool(torch.equal(o8, v8)),
                              "encode_GBps": round(8.0 * n8 / ms_e8 / 1e6, 1), "decode_GBps": round(8.0 * n8 / ms_d8 / 1e6, 1),
                              "encode_roofline_frac": round((8.25 + d8) * n8 / ms_e8 / 1e6 / HBM_PEAK_GBS, 4),
                              "decode_roofline_frac": round((8.25 + d8) * n8 / ms_d8 / 1e6 / HBM_PEAK_GBS, 4)}
        del v8, s8, o8, e8
        # C4 gradients: int32 ~ round(N(0,0.8)) with 0.1% outliers, zigzag fused into the 0124 codec
        n4 = min(n, 1 << 26)
        gr = torch.round(torch.randn(n4, generator=g, device=device) * 0.8).to(torch.int32)
        outl = torch.rand(n4, generator=g, device=device) < 0.001
        gr = torch.where(outl, torch.randint(-(1 << 20), 1 << 20, (n4,), generator=g, device=device, dtype=torch.int32), gr)
        s4, z4 = codec.encode(gr, variant=svb.V0124, flags=svb.ZIGZAG)
        sz4 = int(z4.item())
        d4 = (sz4 - (n4 + 3) // 4) / n4
        o4 = out[:n4]
        ms_e, _ = event_ms(torch, lambda: codec.encode(gr, out=enc_out, variant=svb.V0124, flags=svb.ZIGZAG), it, warmup=ENC_WARMUP)
        ms_d, _ = event_ms(torch, lambda: codec.decode(s4, n4, out=o4, variant=svb.V0124, flags=svb.ZIGZAG), it)
        ok4 = bool(torch.equal(o4, gr))
        tmp = torch.empty(n4, dtype=torch.int32, device=device)
        ms_uf, _ = event_ms(torch, lambda: (codec.decode(s4, n4, out=tmp, variant=svb.V0124),
                                            codec.zigzag(tmp, "decode", out=o4)), it)
        extra["zigzag_0124_gradients"] = {"n": n4, "data_bytes_per_value": round(d4, 4), "parity": ok4,
                                          "decode_GBps": round(4.0 * n4 / ms_d / 1e6, 1), "encode_GBps": round(4.0 * n4 / ms_e / 1e6, 1),
                                          "decode_unfused_GBps": round(4.0 * n4 / ms_uf / 1e6, 1),
                                          "decode_roofline_frac": round((4.25 + d4) * n4 / ms_d / 1e6 / HBM_PEAK_GBS, 4),
                                          "encode_roofline_frac": round((4.25 + d4) * n4 / ms_e / 1e6 / HBM_PEAK_GBS, 4)}
        # the same gradients as float32: quantise -> zigzag -> encode_0124 and back in one pass each
        step = 1.0 / 64.0
        xf = gr.to(torch.float32) * step
        sq, zq = codec.quantize_encode(xf, 1.0 / step, variant=svb.V0124)
        okq = bool(int(zq.item()) == sz4 and torch.equal(sq[:sz4], s4[:sz4]))
        of = torch.empty(n4, dtype=torch.float32, device=device)
        ms_qe, _ = event_ms(torch, lambda: codec.quantize_encode(xf, 1.0 / step, out=enc_out, variant=svb.V0124), it, warmup=ENC_WARMUP)
        ms_dd, _ = event_ms(torch, lambda: codec.decode_dequantize(s4, n4, step, out=of, variant=svb.V0124), it)
        okq &= bool(torch.equal(of, xf))
        ms_da, _ = event_ms(torch, lambda: codec.decode_dequantize(s4, n4, step, out=of, accumulate=True, variant=svb.V0124), it)
        extra["zigzag_0124_gradients"]["quantized_float32"] = {
            "parity": okq, "quantize_encode_GBps": round(4.0 * n4 / ms_qe / 1e6, 1),
            "decode_dequantize_GBps": round(4.0 * n4 / ms_dd / 1e6, 1),
            "decode_dequantize_accumulate_GBps": round(4.0 * n4 / ms_da / 1e6, 1),
            "accumulate_roofline_frac": round((8.25 + d4) * n4 / ms_da / 1e6 / HBM_PEAK_GBS, 4)}
        del gr, s4, tmp, xf, of, sq
        # validate (0.25 B/value) and exact size (4 B/value) on the main stream
        ms_v, _ = event_ms(torch, lambda: codec.validate(stream, size, n), it)
        ms_s, _ = event_ms(torch, lambda: codec.compressedbytes(values), it)
        extra["validate_stream"] = {"ms": round(ms_v, 4), "roofline_frac": round(0.25 * n / ms_v / 1e6 / HBM_PEAK_GBS, 4)}
        extra["compressedbytes"] = {"ms": round(ms_s, 4), "roofline_frac": round(4.0 * n / ms_s / 1e6 / HBM_PEAK_GBS, 4)}
        res["extra"] = extra

        # ---- batched posting lists, sharded over ranks ----------------------------
        from streamvbyte_amd.sharding import partition_streams, all_gather_sizes
        lengths = list_lengths(a.lists, 20241218)
        cut = partition_streams(lengths, world)  # whole lists per rank, balanced by value count
        nl = int(cut[rank + 1] - cut[rank])
        counts, voff, vals = gen_posting_lists(torch, lengths[cut[rank]:cut[rank + 1]], 1234 + rank, device)
        tot = int(voff[-1].item())
        cap = (counts.to(torch.int64) + 3) // 4 + 4 * counts.to(torch.int64) + 16
        boff = torch.zeros(nl + 1, dtype=torch.int64, device=device)
        boff[1:] = torch.cumsum(cap, 0)
        bbuf = torch.empty(int(boff[-1].item()) + 16, dtype=torch.uint8, device=device)
        back = torch.empty_like(vals)
        sizes = torch.empty(nl, dtype=torch.int64, device=device)
        directory = {}

        def enc_batch():
            codec.encode_batch(vals, voff[:-1], bbuf, boff[:-1], counts, None, flags=svb.DELTA, sizes=sizes)
            if dist:  # per-stream sizes -> global byte directory (RCCL all-gather, 8 B per list)
                directory["sizes"], directory["offsets"] = all_gather_sizes(dist, sizes, cut)

        def dec_batch():
            codec.decode_batch(bbuf, boff[:-1], back, voff[:-1], counts, None, flags=svb.DELTA)

        # 20 timed steps after 10 untimed ones: the first dozen calls on freshly allocated gigabyte buffers
        # run 5-8 % below the steady state, and the fixed cost of the closing synchronisation is spread thinner
        kb, wb = 20, 10
        tb_e = time_steps(torch, enc_batch, kb, 3 * wb, dist)  # encoders settle after some 20 ms of back-to-back calls
        tb_d = time_steps(torch, dec_batch, kb, wb, dist)
        okb = bool(torch.equal(back, vals))
        tot_t = torch.tensor([tot], dtype=torch.int64, device=device)
        if dist:
            dist.all_reduce(tot_t)
        tot_all = int(tot_t.item())
        dB = float((sizes.sum().item() - ((counts.to(torch.int64) + 3) // 4).sum().item()) / tot)
        res["batched"] = {"workload": f"{a.lists} posting lists (log-uniform 1..4096 values), delta coded, whole lists "
                                      f"sharded over {world} GPU(s) by value count, RCCL all-gather of sizes"
                                      + ("" if dist else " (1 rank: no collective)"),
                          "values": tot_all, "data_bytes_per_value": round(dB, 4), "parity": okb,
                          "decode_GBps": round(4.0 * tot_all * kb / tb_d / 1e9, 1),
                          "encode_GBps": round(4.0 * tot_all * kb / tb_e / 1e9, 1),
                          "decode_roofline_frac": round((4.25 + dB) * tot_all * kb / tb_d / 1e9 / HBM_PEAK_GBS / world, 4),
                          "encode_roofline_frac": round((4.25 + dB) * tot_all * kb / tb_e / 1e9 / HBM_PEAK_GBS / world, 4)}
        # the same lists coded without delta (mixed widths)
        tb_e2 = time_steps(torch, lambda: codec.encode_batch(vals, voff[:-1], bbuf, boff[:-1], counts, None, flags=0, sizes=sizes), kb, 2, dist)
        tb_d2 = time_steps(torch, lambda: codec.decode_batch(bbuf, boff[:-1], back, voff[:-1], counts, None, flags=0), kb, 2, dist)
        dB2 = float((sizes.sum().item() - ((counts.to(torch.int64) + 3) // 4).sum().item()) / tot)
        res["batched"]["plain"] = {"data_bytes_per_value": round(dB2, 4), "parity": bool(torch.equal(back, vals)),
                                   "decode_GBps": round(4.0 * tot_all * kb / tb_d2 / 1e9, 1),
                                   "encode_GBps": round(4.0 * tot_all * kb / tb_e2 / 1e9, 1)}
        if world == 1:
            # the same lists with every list's values starting on a 64-byte boundary (offsets rounded up to 16
            # values): what a caller who lays the value array out gains
            c64 = counts.to(torch.int64)
            voff_al = torch.zeros(nl + 1, dtype=torch.int64, device=device)
            voff_al[1:] = torch.cumsum((c64 + 15) // 16 * 16, 0)
            vals_al = torch.zeros(int(voff_al[-1].item()), dtype=vals.dtype, device=device)
            shift = torch.repeat_interleave(voff_al[:-1] - voff[:-1], c64)
            shift += torch.arange(tot, dtype=torch.int64, device=device)
            vals_al[shift] = vals
            back_al = torch.zeros_like(vals_al)
            tb_e3 = time_steps(torch, lambda: codec.encode_batch(vals_al, voff_al[:-1], bbuf, boff[:-1], counts, None, flags=svb.DELTA, sizes=sizes), kb, wb, None)
            tb_d3 = time_steps(torch, lambda: codec.decode_batch(bbuf, boff[:-1], back_al, voff_al[:-1], counts, None, flags=svb.DELTA), kb, wb, None)
            res["batched"]["aligned_starts"] = {
                "note": "every list's values start on a 64-byte boundary; bytes counted as above (the padding is not touched)",
                "parity": bool(torch.equal(back_al[shift], vals)),
                "decode_roofline_frac": round((4.25 + dB) * tot * kb / tb_d3 / 1e9 / HBM_PEAK_GBS, 4),
                "encode_roofline_frac": round((4.25 + dB) * tot * kb / tb_e3 / 1e9 / HBM_PEAK_GBS, 4)}
            del vals_al, back_al, shift
        del vals, back, bbuf

        if rank == 0 and world == 1:
            host = values.cpu().numpy().view(np.uint32)
            res["cpu_baseline"] = cpu_baseline(host)

    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res), flush=True)
    return 0 if parity else 1


if __name__ == "__main__":
    sys.exit(main())

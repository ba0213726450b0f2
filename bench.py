#!/usr/bin/env python3
"""Headline benchmark: batch gzip decompress of 4096 x 1 MiB level-6 text
buffers per GPU (BASELINE.json configs[1]), reported as GiB/s of bytes in + out,
with the other per-GPU shapes attached to the same JSON line as the "secondary"
list: the compress half of the metric (configs[2]: zlib level 6, 4096 x 1 MiB
mixed text/binary), one GPU's shard of configs[3] (8192 x 64 KiB gzip
decompress) and of configs[4] (128 x 4 MiB JSON, raw DEFLATE level 12).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` from a bare shell (no WORLD_SIZE in the environment) starts the N
ranks itself, before anything in this process has touched the GPU, and relays
rank 0's JSON line.  One process per GPU.

Streams are independent, so ranks shard the work with no data-path collective;
RCCL carries the start/stop barrier, the max-over-ranks time and, in the
sharded modes, the gather of per-item sizes and status.
  --scaling weak   (default) every rank decodes its own 4096 streams;
  --scaling strong ONE logical batch (default: BASELINE.json configs[3], 65536 x
                   64 KiB gzip) is partitioned over the ranks by
                   libdeflate-rsx_amd/sharding.py, every rank decodes its range and
                   the u64 sizes + i32 status are all_gathered on the GPUs.

A step = one rsx_batch_decompress_device call over the rank's whole batch,
inputs and outputs resident in HBM.  The compressed streams are produced by the
CPU restatement of the reference compressor (gzip, level 6, sync-flushed 256 KiB
segments); `--distinct` different buffers are generated and physically
replicated in device memory to the full batch so every stream is read from its
own addresses.  Every output row is compared with its source on the device,
outside the timed region.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak
METRIC = "batch GiB/s in+out (compress@L6, decompress) at 1/2/4/8 GPUs; % HBM roofline"


def kernel_sources_hash():
    """sha256 (16 hex digits) over the engine's kernel sources: the measured HBM traffic in bench_traffic.json
    belongs to the kernels it was collected from."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "libdeflate-rsx_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def measured_traffic(key, which="dominant_kernels"):
    """HBM bytes per batch for the default workloads, taken from separate rocprofv3 --pmc passes
    (counters cannot be collected inside a timed run): of the kernel(s) the roofline entry names
    ("dominant_kernels") or of every kernel of the call, checksum and gather included
    ("all_kernels"); None for any other configuration, and None when the kernel sources have changed
    since the counters were collected (the file records their hash)."""
    try:
        with open(os.path.join(ROOT, "bench_traffic.json")) as f:
            doc = json.load(f)
        if doc.get("csrc_sha16") != kernel_sources_hash():
            return None
        e = doc.get(key)
        return e.get(which) if isinstance(e, dict) else None
    except (OSError, ValueError):
        return None


def host_threads():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 16))  # a 1-GPU box shares 16 host cores


def self_launch(args, argv):
    """--gpus N without a launcher: start N ranks under torch.distributed.run as a child of this
    process (which has not touched the GPU) and pass their output and exit code on."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def make_streams(distinct, buf_bytes, fmt, level, seed0, compressor="reference"):
    import corpus
    import oracle_lib as orc
    texts = [corpus.text(buf_bytes, seed0 + i) for i in range(distinct)]
    if compressor == "zlib":   # one stream without joints, a block every 16K symbols
        import zlib
        wbits = {0: -15, 1: 15, 2: 31}[fmt]
        streams = []
        for t in texts:
            co = zlib.compressobj(level, zlib.DEFLATED, wbits)
            streams.append(np.frombuffer(co.compress(t) + co.flush(), dtype=np.uint8))
        return texts, streams
    flat = np.frombuffer(b"".join(texts), dtype=np.uint8)
    in_off = np.arange(distinct + 1, dtype=np.uint64) * np.uint64(buf_bytes)
    out, out_off, sizes, status = orc.batch_compress(flat, in_off, fmt, level, threads=host_threads())
    assert (status == 0).all(), "oracle compressor failed on synthetic text"
    streams = [out[int(out_off[i]):int(out_off[i]) + int(sizes[i])] for i in range(distinct)]
    return texts, streams


def cpu_decompress_baseline(streams, buf_bytes, fmt, target_s):
    """Times the CPU restatement of the reference decoder on the distinct streams
    (all host threads the box grants), repeated until ~target_s of work."""
    import oracle_lib as orc
    th = host_threads()
    d = len(streams)
    flat = np.concatenate(streams)
    in_off = np.zeros(d + 1, dtype=np.uint64)
    np.cumsum([len(s) for s in streams], out=in_off[1:])
    out_off = np.arange(d + 1, dtype=np.uint64) * np.uint64(buf_bytes)
    orc.batch_decompress(flat, in_off, out_off, fmt, threads=th)  # warm-up
    reps = 0
    t0 = time.perf_counter()
    while True:
        _, sizes, status = orc.batch_decompress(flat, in_off, out_off, fmt, threads=th)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= target_s:
            break
    assert (status == 0).all()
    nbytes = reps * (int(in_off[-1]) + int(sizes.sum()))
    return {"value": nbytes / dt / 2**30, "unit": "GiB/s", "cores": th,
            "kind": "port",
            "note": "C restatement of the reference decoder's careful loop (byte-serial match copy), not its BMI2 "
                    "fast loop: a lower bound of the reference's CPU rate",
            "sample": "%d x %d distinct streams of the same workload, %d passes, %.1f s" % (reps, d, reps, dt)}


class Ctx:
    """torch, torch.distributed, the engine package and this rank's place in the job."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        import rsx_pkg
        self.torch, self.dist, self.rsx = torch, dist, rsx_pkg.load()
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
        # RSX_BENCH_REHEARSE=1: a rehearsal of the multi-rank path on a box with fewer GPUs than ranks --
        # ranks share the GPUs that exist and the collectives run over gloo.  Not a measurement.
        self.rehearse = os.environ.get("RSX_BENCH_REHEARSE") == "1"
        if self.rehearse:
            self.local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        if self.world > 1:
            if self.rehearse:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=self.dev)
        self.reduce_device = torch.device("cpu") if self.rehearse else self.dev
        self.stream = torch.cuda.current_stream(self.dev)
        self.eng = self.rsx.Engine(self.local_rank, self.stream.cuda_stream)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.reduce_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.world == 1:
            return int(value)
        t = self.torch.tensor([int(value)], dtype=self.torch.int64, device=self.reduce_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())


def timed_steps(ctx, step, steps, warmup):
    """W untimed steps, then exactly K steps between two barrier + synchronize pairs; a third
    loop reads the per-launch kernel time so that reading events never serialises the timed one."""
    for _ in range(warmup):
        step()
    ctx.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.barrier()
    dt = time.perf_counter() - t0
    kernel_ms = []
    for _ in range(steps):
        step()
        kernel_ms.append(ctx.eng.last_kernel_ms())
    ctx.torch.cuda.synchronize(ctx.dev)
    return ctx.max_over_ranks(dt), float(np.mean(kernel_ms))


def device_batch(ctx, streams, idx):
    """Device copies of the streams idx[0], idx[1], ... (indices into `streams`), flat + offsets."""
    torch = ctx.torch
    lens = np.array([len(streams[i]) for i in idx], dtype=np.uint64)
    in_off = np.zeros(len(idx) + 1, dtype=np.uint64)
    np.cumsum(lens, out=in_off[1:])
    h_in = np.empty(int(in_off[-1]), dtype=np.uint8)
    for k, i in enumerate(idx):
        h_in[int(in_off[k]):int(in_off[k + 1])] = streams[i]
    return torch.from_numpy(h_in).to(ctx.dev), torch.from_numpy(in_off.view(np.int64)).to(ctx.dev), in_off


def verify_rows(ctx, d_out, n, buf, texts, idx, d_status, d_sizes):
    """Every status 0, every size exact and every output row equal to its source, compared on the device."""
    torch = ctx.torch
    st = d_status.cpu().numpy()
    sz = d_sizes.cpu().numpy()
    ok = bool((st == 0).all() and (sz == buf).all())
    want = torch.from_numpy(np.frombuffer(b"".join(texts), dtype=np.uint8).copy()).to(ctx.dev).view(len(texts), buf)
    rows = d_out.view(n, buf)
    sel = torch.from_numpy(np.asarray(idx, dtype=np.int64)).to(ctx.dev)
    bad_rows = []
    for a in range(0, n, 512):   # (pieces: the comparison needs a gathered copy of the sources)
        z = min(n, a + 512)
        eq = (rows[a:z] == want[sel[a:z]]).all(dim=1)
        if not bool(eq.all()):
            bad_rows += [a + int(k) for k in torch.nonzero(~eq).flatten().cpu().numpy()[:8]]
    if not ok or bad_rows:
        bad = np.nonzero((st != 0) | (sz != buf))[0]
        raise SystemExit("bench.py: GPU output does not match the input texts (status values %s, %d streams with a "
                         "bad status or size, first %s; rows that differ: %s)"
                         % (np.unique(st), len(bad), bad[:8], bad_rows[:8]))
    return True


def host_path_rate(ctx, streams, texts, buf, fmt, n=1024):
    """What a compiled binding gets from rsx_batch_decompress with HOST pointers (the only thing
    src/batch.rs hands over): copy in, kernels and copy out included.  Reported beside the headline,
    never as `value`."""
    eng = ctx.rsx.Engine(ctx.local_rank)
    d = len(streams)
    flat = np.concatenate([streams[i % d] for i in range(n)])
    in_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(streams[i % d]) for i in range(n)], out=in_off[1:])
    out_off = np.arange(n + 1, dtype=np.uint64) * np.uint64(buf)
    out = np.empty(n * buf, dtype=np.uint8)
    sizes = np.zeros(n, dtype=np.uint64)
    status = np.zeros(n, dtype=np.int32)

    def call():
        r = eng._lib.rsx_batch_decompress(eng._h, fmt, flat.ctypes.data, in_off.ctypes.data, n, out.ctypes.data,
                                          out_off.ctypes.data, sizes.ctypes.data, status.ctypes.data)
        assert r == 0 and (status == 0).all() and (sizes == buf).all()
    call()   # (untimed: pages of `out` are touched, the workspace and the pinned buffers are allocated)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        call()
    dt = (time.perf_counter() - t0) / reps
    ok = all(out[i * buf:(i + 1) * buf].tobytes() == texts[i % d] for i in (0, 1, n // 2, n - 1))
    eng.close()
    if not ok:
        raise SystemExit("bench.py: host-pointer decompress returned wrong bytes")
    return {"gibps_in_plus_out": (flat.size + out.size) / dt / 2**30, "ms_per_call": dt * 1e3,
            "what": "rsx_batch_decompress, %d x %d KiB gzip streams in pageable host memory, results back in host "
                    "memory (PCIe both ways, slices overlapped)" % (n, buf // 1024)}


def bench_decompress(args, ctx):
    """BASELINE.json configs[1]: batch gzip decompress, n x 1 MiB level-6 text per GPU (weak scaling)."""
    rsx = ctx.rsx
    fmt, level = rsx.FORMAT_GZIP, 6
    buf = args.buf_kib * 1024
    n = args.streams
    distinct = min(args.distinct, n)
    texts, streams = make_streams(distinct, buf, fmt, level, seed0=1000 * ctx.rank, compressor=args.compressor)
    idx = [i % distinct for i in range(n)]   # stream i is a private copy of distinct stream i % distinct
    d_in, d_in_off, in_off = device_batch(ctx, streams, idx)
    total_in, total_out = int(in_off[-1]), n * buf
    torch = ctx.torch
    d_out_off = torch.from_numpy((np.arange(n + 1, dtype=np.uint64) * np.uint64(buf)).view(np.int64)).to(ctx.dev)
    d_out = torch.empty(total_out, dtype=torch.uint8, device=ctx.dev)
    d_sizes = torch.zeros(n, dtype=torch.int64, device=ctx.dev)
    d_status = torch.full((n,), -7, dtype=torch.int32, device=ctx.dev)

    def step():
        ctx.eng.batch_decompress_device(fmt, d_in.data_ptr(), d_in_off.data_ptr(), n, total_in, total_out,
                                        buf, d_out.data_ptr(), d_out_off.data_ptr(), d_sizes.data_ptr(),
                                        d_status.data_ptr())

    dt, k_ms = timed_steps(ctx, step, args.steps, args.warmup)
    ok = verify_rows(ctx, d_out, n, buf, texts, idx, d_status, d_sizes)
    decoded_twice = ctx.eng.last_fallbacks()   # streams the one-kernel decoder took behind the two-kernel one
    job_bytes = ctx.sum_over_ranks(total_in + total_out)
    alg_bytes = total_in + total_out   # per launch on this rank
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    res = {
        "value": job_bytes * args.steps / dt / 2**30, "unit": "GiB/s", "ms_per_step": dt / args.steps * 1e3,
        "scaling": "weak",
        "config": {
            "workload": "batch gzip decompress, %d x %d KiB level-6 enwik-style text per GPU" % (n, args.buf_kib),
            "op": "decompress", "format": "gzip", "level": level, "streams_per_gpu": n,
            "buf_bytes": buf, "distinct_buffers": distinct, "compressed_ratio": total_out / total_in,
            "compressor": ("CPU restatement of the reference, 256 KiB sync-flushed segments"
                           if args.compressor == "reference" else "zlib, one stream without joints"),
            "parallelism": "batch sharded over %d GPU(s), no data-path collective" % ctx.world,
            "verified": ok, "verified_rows": n, "streams_decoded_twice": decoded_twice,
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            # (kernel_ms spans every kernel of the call, and so does the traffic)
            "traffic": measured_traffic("decompress:%dx%dKiB:gzip:%d" % (n, args.buf_kib, level), "all_kernels"),
            "traffic_decoder_kernels": measured_traffic("decompress:%dx%dKiB:gzip:%d" % (n, args.buf_kib, level)),
            "kernel": "inflate_decode_kernel + inflate_expand_kernel (dependent launches; kernel_ms spans the whole call: the marker search that cuts the streams into parts, both decoder kernels, the CRC-32 pass and the trailer check)",
            "kernel_ms": k_ms, "algorithmic_bytes": alg_bytes,
        },
    }
    if ctx.world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_decompress_baseline(streams, buf, fmt, args.cpu_seconds)
    if ctx.world == 1 and args.host_path:
        res["config"]["host_path"] = host_path_rate(ctx, streams, texts, buf, fmt)
    del d_in, d_out
    torch.cuda.empty_cache()
    return res


def bench_compress(args, ctx):
    """BASELINE.json configs[2]: batch zlib compress level 6, n x 1 MiB mixed text/binary per GPU."""
    import corpus
    import oracle_lib as orc
    rsx, torch = ctx.rsx, ctx.torch
    level = args.level
    fmt = rsx.FORMAT_ZLIB if args.corpus == "mixed" else rsx.FORMAT_DEFLATE
    ofmt = orc.FMT_ZLIB if args.corpus == "mixed" else orc.FMT_DEFLATE
    fmt_name = "zlib" if args.corpus == "mixed" else "raw deflate"
    buf = args.buf_kib * 1024
    n = args.streams
    distinct = min(args.distinct, n)
    gen = {"mixed": corpus.mixed, "json": corpus.json_like, "text": corpus.text}[args.corpus]
    datas = [gen(buf, 1000 * ctx.rank + i) for i in range(distinct)]
    h_in = np.frombuffer(b"".join(datas), dtype=np.uint8)
    d_one = torch.from_numpy(h_in.copy()).to(ctx.dev)
    d_in = d_one.repeat(n // distinct + 1)[: n * buf].contiguous()  # private copy per item
    in_off = np.arange(n + 1, dtype=np.uint64) * np.uint64(buf)
    bound = int(rsx.compress_bound(fmt, buf))
    out_off = np.arange(n + 1, dtype=np.uint64) * np.uint64(bound)
    d_in_off = torch.from_numpy(in_off.view(np.int64)).to(ctx.dev)
    d_out_off = torch.from_numpy(out_off.view(np.int64)).to(ctx.dev)
    d_out = torch.zeros(n * bound, dtype=torch.uint8, device=ctx.dev)
    d_sizes = torch.zeros(n, dtype=torch.int64, device=ctx.dev)
    d_status = torch.full((n,), -7, dtype=torch.int32, device=ctx.dev)
    total_in = n * buf

    def step():
        ctx.eng.batch_compress_device(fmt, level, d_in.data_ptr(), d_in_off.data_ptr(), n, total_in, buf,
                                      d_out.data_ptr(), d_out_off.data_ptr(), d_sizes.data_ptr(),
                                      d_status.data_ptr())

    dt, k_ms = timed_steps(ctx, step, args.steps, args.warmup)
    st = d_status.cpu().numpy()
    sz = d_sizes.cpu().numpy()
    ok = bool((st == 0).all())
    if n > distinct:  # every replica of an input must carry the same bytes as its original
        rows = d_out.view(n, bound)
        ok = ok and bool((rows[distinct:] == rows[:n - distinct]).all())
        ok = ok and bool((sz[distinct:] == sz[:n - distinct]).all())
    for i in range(distinct):  # the oracle must inflate every distinct stream to its input
        c = d_out[i * bound: i * bound + int(sz[i])].cpu().numpy().tobytes()
        r, _, back = orc.decompress(c, buf, ofmt)
        ok = ok and r == 0 and back == datas[i]
    if not ok:
        raise SystemExit("bench.py: compressed output failed verification")
    total_out = int(sz.sum())
    job_bytes = ctx.sum_over_ranks(total_in + total_out)
    achieved = (total_in + total_out) / (k_ms * 1e-3) / 1e9
    res = {
        "value": job_bytes * args.steps / dt / 2**30, "unit": "GiB/s", "ms_per_step": dt / args.steps * 1e3,
        "scaling": "weak",
        "config": {"workload": "batch %s compress level %d, %d x %d KiB %s per GPU"
                               % (fmt_name, level, n, args.buf_kib,
                                  {"mixed": "mixed text/binary", "json": "JSON", "text": "text"}[args.corpus]),
                   "op": "compress", "format": fmt_name, "level": level, "streams_per_gpu": n,
                   "buf_bytes": buf, "distinct_buffers": distinct,
                   "compressed_ratio": total_in / total_out, "verified": ok},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": measured_traffic("compress:%dx%dKiB:%s:%d" % (n, args.buf_kib, args.corpus, level), "all_kernels"),
                     "traffic_match_and_emit_kernels": measured_traffic("compress:%dx%dKiB:%s:%d" % (n, args.buf_kib, args.corpus, level)),
                     "kernel": "deflate_chunks_kernel (dominant; kernel_ms spans the whole call: match, emit, checksum, plan, place)", "kernel_ms": k_ms,
                     "algorithmic_bytes": total_in + total_out},
    }
    if ctx.world == 1 and not args.no_cpu_baseline:
        th = host_threads()
        flat = np.frombuffer(b"".join(datas), dtype=np.uint8)
        off = np.arange(distinct + 1, dtype=np.uint64) * np.uint64(buf)
        t1 = time.perf_counter()
        reps = 0
        while True:
            _, _, csz, cst = orc.batch_compress(flat, off, ofmt, level, threads=th)
            reps += 1
            cdt = time.perf_counter() - t1
            if cdt >= args.cpu_seconds:
                break
        res["cpu_baseline"] = {"value": reps * (distinct * buf + int(csz.sum())) / cdt / 2**30,
                               "unit": "GiB/s", "cores": th, "kind": "port",
                               "note": "C restatement of the reference compressor (scalar match-length loop)",
                               "sample": "%d passes over %d distinct buffers, %.1f s" % (reps, distinct, cdt),
                               "compressed_ratio": distinct * buf / float(csz.sum())}
    del d_in, d_out
    torch.cuda.empty_cache()
    return res


def bench_strong(args, ctx):
    """BASELINE.json configs[3]: ONE logical batch of gzip streams (default 65536 x 64 KiB) spread over the
    ranks: partition by compressed bytes, every rank decodes its range, sizes and status are gathered on
    the GPUs (RCCL), three totals are all-reduced.  The total work is fixed as N grows."""
    from libdeflate_rsx_amd import sharding
    rsx, torch = ctx.rsx, ctx.torch
    fmt, level = rsx.FORMAT_GZIP, 6
    buf = args.buf_kib * 1024
    n_total = args.streams
    distinct = min(args.distinct, n_total)
    texts, streams = make_streams(distinct, buf, fmt, level, seed0=7000)   # the same logical batch on every rank
    lens_all = np.array([len(streams[i % distinct]) for i in range(n_total)], dtype=np.int64)
    ranges = sharding.partition(lens_all, ctx.world)
    s, e = ranges[ctx.rank]
    n = e - s
    idx = [i % distinct for i in range(s, e)]
    d_in, d_in_off, in_off = device_batch(ctx, streams, idx)
    total_in, total_out = int(in_off[-1]), n * buf
    d_out_off = torch.from_numpy((np.arange(n + 1, dtype=np.uint64) * np.uint64(buf)).view(np.int64)).to(ctx.dev)
    d_out = torch.empty(max(total_out, 1), dtype=torch.uint8, device=ctx.dev)
    d_sizes = torch.zeros(max(n, 1), dtype=torch.int64, device=ctx.dev)
    d_status = torch.full((max(n, 1),), -7, dtype=torch.int32, device=ctx.dev)
    gathered = {}

    def step():
        if n:
            ctx.eng.batch_decompress_device(fmt, d_in.data_ptr(), d_in_off.data_ptr(), n, total_in, total_out,
                                            buf, d_out.data_ptr(), d_out_off.data_ptr(), d_sizes.data_ptr(),
                                            d_status.data_ptr())
        # the exchange step of the path: sizes + status of the whole batch on every rank, three totals
        if ctx.rehearse:
            ctx.torch.cuda.synchronize(ctx.dev)
        gathered["r"] = sharding.gather_results(d_sizes[:n], d_status[:n], ranges,
                                                ctx.dist if ctx.world > 1 else None, total_in=total_in)

    dt, k_ms = timed_steps(ctx, step, args.steps, args.warmup)
    if n:
        verify_rows(ctx, d_out[:total_out], n, buf, texts, idx, d_status[:n], d_sizes[:n])
    all_sizes, all_status, totals = gathered["r"]
    ok = bool((all_status == 0).all() and (all_sizes == buf).all() and len(all_sizes) == n_total
              and totals == (int(lens_all.sum()), n_total * buf, 0))
    if not ok:
        raise SystemExit("bench.py: gathered sizes / status / totals of the sharded batch are wrong: %s" % (totals,))
    job_bytes = totals[0] + totals[1]
    alg_bytes = total_in + total_out
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if n else 0.0
    return {
        "value": job_bytes * args.steps / dt / 2**30, "unit": "GiB/s", "ms_per_step": dt / args.steps * 1e3,
        "scaling": "strong",
        "config": {
            "workload": "one batch of %d x %d KiB gzip level-6 text streams sharded over %d GPU(s), sizes/status "
                        "gathered over RCCL" % (n_total, args.buf_kib, ctx.world),
            "op": "decompress", "format": "gzip", "level": level, "streams_total": n_total,
            "streams_this_rank": n, "buf_bytes": buf, "distinct_buffers": distinct,
            "compressed_ratio": n_total * buf / float(lens_all.sum()),
            "parallelism": "contiguous ranges balanced by compressed bytes; all_gather(u64 sizes, i32 status) + "
                           "all_reduce(3 x i64) per step", "verified": ok, "verified_rows": n,
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
            "kernel": "inflate_decode_kernel + inflate_expand_kernel (dependent launches; kernel_ms spans the whole call: the marker search that cuts the streams into parts, both decoder kernels, the CRC-32 pass and the trailer check)",
            "kernel_ms": k_ms, "algorithmic_bytes": alg_bytes,
        },
    }


def bench_large(args, ctx):
    """One large buffer (src/api.rs:22-57: the reference cuts inputs above 256 KiB into 256 KiB chunks, compresses
    them in parallel and joins them with sync-flush markers, src/compress/mod.rs:699-772): a batch of ONE item of
    --large-mib MiB of text, gzip level --level.  The chunks are the unit of parallelism: 64 MiB = 256 chunks = one
    workgroup per CU.  The output is inflated on the GPU (one stream, cut at the chunk joints into parts that are
    decoded side by side: reported, not the metric) and by the oracle, and compared with the input."""
    import corpus
    import oracle_lib as orc
    rsx, torch = ctx.rsx, ctx.torch
    fmt, level = rsx.FORMAT_GZIP, args.level
    size = args.large_mib << 20
    piece = min(size, 8 << 20)
    data = b"".join(corpus.text(piece, 31000 + 1000 * ctx.rank + k) for k in range((size + piece - 1) // piece))[:size]
    d_in = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(ctx.dev)
    bound = int(rsx.compress_bound(fmt, size))
    d_in_off = torch.tensor([0, size], dtype=torch.int64, device=ctx.dev)
    d_out_off = torch.tensor([0, bound], dtype=torch.int64, device=ctx.dev)
    d_out = torch.zeros(bound, dtype=torch.uint8, device=ctx.dev)
    d_sizes = torch.zeros(1, dtype=torch.int64, device=ctx.dev)
    d_status = torch.full((1,), -7, dtype=torch.int32, device=ctx.dev)

    def step():
        ctx.eng.batch_compress_device(fmt, level, d_in.data_ptr(), d_in_off.data_ptr(), 1, size, size,
                                      d_out.data_ptr(), d_out_off.data_ptr(), d_sizes.data_ptr(), d_status.data_ptr())

    dt, k_ms = timed_steps(ctx, step, args.steps, args.warmup)
    csize = int(d_sizes.item())
    if int(d_status.item()) != 0:
        raise SystemExit("bench.py: compressing the large buffer failed with status %d" % int(d_status.item()))
    # inflate the stream on the GPU (a single stream: one workgroup) and with the oracle
    d_back = torch.empty(size, dtype=torch.uint8, device=ctx.dev)
    b_off = torch.tensor([0, size], dtype=torch.int64, device=ctx.dev)
    c_off = torch.tensor([0, csize], dtype=torch.int64, device=ctx.dev)
    b_sizes = torch.zeros(1, dtype=torch.int64, device=ctx.dev)
    b_status = torch.full((1,), -7, dtype=torch.int32, device=ctx.dev)
    def inflate():
        ctx.eng.batch_decompress_device(fmt, d_out.data_ptr(), c_off.data_ptr(), 1, csize, size, size,
                                        d_back.data_ptr(), b_off.data_ptr(), b_sizes.data_ptr(), b_status.data_ptr())
    inflate()   # (untimed: the workspace grows)
    ctx.eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        inflate()
    ctx.eng.synchronize()
    inflate_ms = (time.perf_counter() - t0) / 3 * 1e3   # whole call: marker search, both decoder kernels, CRC-32
    parts, whole = ctx.eng.last_parts(), ctx.eng.last_whole()
    ok = int(b_status.item()) == 0 and int(b_sizes.item()) == size and bool((d_back == d_in).all())
    comp = d_out[:csize].cpu().numpy().tobytes()
    r, used, back = orc.decompress(comp, size, orc.FMT_GZIP)
    ok = ok and r == 0 and used == csize and back == data
    if not ok:
        raise SystemExit("bench.py: the large buffer did not survive the round trip")
    job_bytes = ctx.sum_over_ranks(size + csize)
    achieved = (size + csize) / (k_ms * 1e-3) / 1e9
    return {
        "value": job_bytes * args.steps / dt / 2**30, "unit": "GiB/s", "ms_per_step": dt / args.steps * 1e3,
        "scaling": "weak",
        "config": {"workload": "one %d MiB text buffer per GPU, gzip compress level %d, batch of 1 (chunk-parallel: "
                               "%d x 256 KiB chunks joined by sync-flush markers)" % (args.large_mib, level, size >> 18),
                   "op": "compress", "format": "gzip", "level": level, "buf_bytes": size,
                   "compressed_ratio": size / csize, "verified": ok,
                   "single_stream_inflate_ms": inflate_ms, "single_stream_inflate_parts": parts, "single_stream_inflate_whole": whole,
                   "single_stream_inflate_gibps": (size + csize) / (inflate_ms * 1e-3) / 2**30},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel": "deflate_chunks_kernel",
                     "kernel_ms": k_ms, "algorithmic_bytes": size + csize},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=None,
                    help="streams per GPU (weak scaling, default 4096) or in the whole batch (strong, default 65536)")
    ap.add_argument("--buf-kib", type=int, default=None, help="uncompressed KiB per stream (1024; strong: 64)")
    ap.add_argument("--distinct", type=int, default=64, help="distinct buffers generated on the CPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other legs of the default run")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work per cpu_baseline sample")
    ap.add_argument("--host-path", action="store_true", default=None,
                    help="also time rsx_batch_decompress through host pointers (default: with the secondary legs)")
    ap.add_argument("--level", type=int, default=6, help="compress only")
    ap.add_argument("--corpus", choices=["mixed", "json", "text"], default="mixed",
                    help="compress only: mixed = zlib (headline), json/text = raw deflate")
    ap.add_argument("--op", choices=["decompress", "compress", "large"], default="decompress",
                    help="decompress = gzip level-6 text (headline) with the compress leg attached; "
                         "compress = zlib level-6 mixed alone; large = one --large-mib MiB buffer, batch of 1")
    ap.add_argument("--large-mib", type=int, default=64, help="--op large: size of the single buffer")
    ap.add_argument("--compressor", choices=["reference", "zlib"], default="reference",
                    help="decompress only: who made the streams (reference = the headline's)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    argv = sys.argv[1:]
    args = ap.parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))
    strong = args.scaling == "strong"
    if args.streams is None:
        args.streams = 65536 if strong else 4096
    if args.buf_kib is None:
        args.buf_kib = 64 if strong else 1024

    if args.host_path is None:
        args.host_path = not args.no_secondary and args.op == "decompress" and not strong
    ctx = Ctx(args)
    if strong:
        res = bench_strong(args, ctx)
    elif args.op == "compress":
        res = bench_compress(args, ctx)
    elif args.op == "large":
        res = bench_large(args, ctx)
    else:
        res = bench_decompress(args, ctx)
        if not args.no_secondary:
            # the other per-GPU shapes of BASELINE.json: configs[2] (the compress half of the metric), one GPU's
            # shard of configs[3] (65536 x 64 KiB gzip over 8 GPUs) and of configs[4] (1024 x 4 MiB JSON, raw level 12)
            def leg(**over):
                a = argparse.Namespace(**vars(args))
                a.host_path = False
                a.cpu_seconds = min(args.cpu_seconds, 8.0)
                for k, v in over.items():
                    setattr(a, k, v)
                return a
            legs = [(bench_compress, leg(corpus="mixed", level=6)),
                    (bench_decompress, leg(streams=8192, buf_kib=64, steps=min(args.steps, 5))),
                    (bench_compress, leg(streams=128, buf_kib=4096, corpus="json", level=12, distinct=16,
                                         steps=min(args.steps, 2), warmup=1))]
            res["secondary"] = []
            for fn, a in legs:
                sec = fn(a, ctx)
                entry = {"metric": METRIC, "value": sec["value"], "unit": sec["unit"], "steps": a.steps, "warmup": a.warmup,
                         "ms_per_step": sec["ms_per_step"], "config": sec["config"], "roofline": sec["roofline"]}
                if "cpu_baseline" in sec:
                    entry["cpu_baseline"] = sec["cpu_baseline"]
                res["secondary"].append(entry)
    if ctx.rank == 0:
        line = {"metric": METRIC, "value": res["value"], "unit": res["unit"], "n_gpus": ctx.world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
                "higher_is_better": True, "scaling": res["scaling"], "vs_baseline": None, "dtype": "u8",
                "data": "synthetic"}
        for k in ("config", "roofline", "cpu_baseline", "secondary"):
            if k in res:
                line[k] = res[k]
        print(json.dumps(line), flush=True)
    if ctx.world > 1:
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()

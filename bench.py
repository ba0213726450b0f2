#!/usr/bin/env python3
"""Headline benchmark: batch gzip decompress of 4096 x 1 MiB level-6 text
buffers per GPU (BASELINE.json configs[1]), reported as GiB/s of bytes in + out.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  Streams are independent, so ranks shard the batch with no
data-path collective (weak scaling: every rank decodes its own 4096 streams);
RCCL is used only for the start/stop barrier and the max-over-ranks reduce.

A step = one rsx_batch_decompress_device call over the whole per-rank batch,
inputs and outputs resident in HBM.  The compressed streams are produced by the
CPU restatement of the reference compressor (gzip, level 6, four sync-flushed
256 KiB segments per buffer); `--distinct` different buffers are generated and
physically replicated in device memory to the full batch so every stream is
read from its own addresses.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak


def measured_traffic(key):
    """HBM bytes per launch of the dominant kernel for the default workloads, taken from
    separate rocprofv3 --pmc passes (counters cannot be collected inside a timed run);
    None for any other configuration."""
    try:
        with open(os.path.join(ROOT, "bench_traffic.json")) as f:
            return json.load(f).get(key)
    except OSError:
        return None


def host_threads():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 16))  # a 1-GPU box shares 16 host cores


def make_streams(distinct, buf_bytes, fmt, level, seed0):
    import corpus
    import oracle_lib as orc
    texts = [corpus.text(buf_bytes, seed0 + i) for i in range(distinct)]
    flat = np.frombuffer(b"".join(texts), dtype=np.uint8)
    in_off = np.arange(distinct + 1, dtype=np.uint64) * np.uint64(buf_bytes)
    out, out_off, sizes, status = orc.batch_compress(flat, in_off, fmt, level, threads=host_threads())
    assert (status == 0).all(), "oracle compressor failed on synthetic text"
    streams = [out[int(out_off[i]):int(out_off[i]) + int(sizes[i])] for i in range(distinct)]
    return texts, streams


def cpu_baseline(streams, buf_bytes, fmt, target_s=12.0):
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
    return {"value": nbytes / dt / 2**30, "unit": "GiB/s", "cores": th, "kind": "port",
            "sample": "%d x %d distinct streams of the same workload, %d passes, %.1f s"
                      % (reps, d, reps, dt)}


def bench_compress(args, rsx, torch, dist, rank, local_rank, world, dev):
    """BASELINE.json configs[2]: batch zlib compress level 6, n x 1 MiB mixed text/binary."""
    import corpus
    import oracle_lib as orc
    level = args.level
    fmt = rsx.FORMAT_ZLIB if args.corpus == "mixed" else rsx.FORMAT_DEFLATE
    ofmt = orc.FMT_ZLIB if args.corpus == "mixed" else orc.FMT_DEFLATE
    fmt_name = "zlib" if args.corpus == "mixed" else "raw deflate"
    buf = args.buf_kib * 1024
    n = args.streams
    distinct = min(args.distinct, n)
    gen = {"mixed": corpus.mixed, "json": corpus.json_like, "text": corpus.text}[args.corpus]
    datas = [gen(buf, 1000 * rank + i) for i in range(distinct)]
    h_in = np.frombuffer(b"".join(datas), dtype=np.uint8)
    d_one = torch.from_numpy(h_in.copy()).to(dev)
    d_in = d_one.repeat(n // distinct + 1)[: n * buf].contiguous()  # private copy per item
    in_off = np.arange(n + 1, dtype=np.uint64) * np.uint64(buf)
    bound = int(rsx.compress_bound(fmt, buf))
    out_off = np.arange(n + 1, dtype=np.uint64) * np.uint64(bound)
    d_in_off = torch.from_numpy(in_off.view(np.int64)).to(dev)
    d_out_off = torch.from_numpy(out_off.view(np.int64)).to(dev)
    d_out = torch.zeros(n * bound, dtype=torch.uint8, device=dev)
    d_sizes = torch.zeros(n, dtype=torch.int64, device=dev)
    d_status = torch.full((n,), -7, dtype=torch.int32, device=dev)
    eng = rsx.Engine(local_rank, torch.cuda.current_stream(dev).cuda_stream)
    total_in = n * buf

    def step():
        eng.batch_compress_device(fmt, level, d_in.data_ptr(), d_in_off.data_ptr(), n, total_in, buf,
                                  d_out.data_ptr(), d_out_off.data_ptr(), d_sizes.data_ptr(),
                                  d_status.data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms = []
    for _ in range(args.steps):
        step()
        kernel_ms.append(eng.last_kernel_ms())
    torch.cuda.synchronize(dev)
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
    job_bytes = total_in + total_out
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=args.reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        b = torch.tensor([job_bytes], dtype=torch.int64, device=args.reduce_device)
        dist.all_reduce(b, op=dist.ReduceOp.SUM)
        job_bytes = int(b.item())
    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        achieved = (total_in + total_out) / (k_ms * 1e-3) / 1e9
        res = {
            "metric": "batch GiB/s in+out (compress@L6, decompress) at 1/2/4/8 GPUs; % HBM roofline",
            "value": job_bytes * args.steps / dt / 2**30, "unit": "GiB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "batch %s compress level %d, %d x %d KiB %s per GPU"
                                   % (fmt_name, level, n, args.buf_kib,
                                      {"mixed": "mixed text/binary", "json": "JSON", "text": "text"}[args.corpus]),
                       "op": "compress", "format": fmt_name, "level": level, "streams_per_gpu": n,
                       "buf_bytes": buf, "distinct_buffers": distinct,
                       "compressed_ratio": total_in / total_out, "verified": ok},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": measured_traffic("compress:%dx%dKiB:%s:%d" % (n, args.buf_kib, args.corpus, level)),
                         "kernel": "deflate_chunks_kernel", "kernel_ms": k_ms,
                         "algorithmic_bytes": total_in + total_out},
        }
        if world == 1 and not args.no_cpu_baseline:
            th = host_threads()
            flat = np.frombuffer(b"".join(datas), dtype=np.uint8)
            off = np.arange(distinct + 1, dtype=np.uint64) * np.uint64(buf)
            t1 = time.perf_counter()
            reps = 0
            while True:
                _, _, csz, cst = orc.batch_compress(flat, off, ofmt, level, threads=th)
                reps += 1
                cdt = time.perf_counter() - t1
                if cdt >= 12.0:
                    break
            res["cpu_baseline"] = {"value": reps * (distinct * buf + int(csz.sum())) / cdt / 2**30,
                                   "unit": "GiB/s", "cores": th, "kind": "port",
                                   "sample": "%d passes over %d distinct buffers, %.1f s" % (reps, distinct, cdt),
                                   "compressed_ratio": distinct * buf / float(csz.sum())}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=4096, help="streams per GPU")
    ap.add_argument("--buf-kib", type=int, default=1024, help="uncompressed KiB per stream")
    ap.add_argument("--distinct", type=int, default=64, help="distinct buffers generated on the CPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--level", type=int, default=6, help="compress only")
    ap.add_argument("--corpus", choices=["mixed", "json", "text"], default="mixed",
                    help="compress only: mixed = zlib (headline), json/text = raw deflate")
    ap.add_argument("--op", choices=["decompress", "compress"], default="decompress",
                    help="decompress = gzip level-6 text (headline); compress = zlib level-6 mixed")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import rsx_pkg
    rsx = rsx_pkg.load()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # RSX_BENCH_REHEARSE=1: a rehearsal of the multi-rank path on a box with fewer GPUs than ranks --
    # ranks share the GPUs that exist and the two reductions run over gloo.  Not a measurement.
    rehearse = os.environ.get("RSX_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    args.reduce_device = torch.device("cpu") if rehearse else dev

    if args.op == "compress":
        return bench_compress(args, rsx, torch, dist, rank, local_rank, world, dev)
    fmt, level = rsx.FORMAT_GZIP, 6
    buf = args.buf_kib * 1024
    n = args.streams
    distinct = min(args.distinct, n)
    texts, streams = make_streams(distinct, buf, fmt, level, seed0=1000 * rank)

    # replicate to the full batch: stream i is a private copy of distinct stream i % distinct
    lens = np.array([len(streams[i % distinct]) for i in range(n)], dtype=np.uint64)
    in_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(lens, out=in_off[1:])
    total_in = int(in_off[-1])
    h_in = np.empty(total_in, dtype=np.uint8)
    for i in range(n):
        h_in[int(in_off[i]):int(in_off[i + 1])] = streams[i % distinct]
    out_off = np.arange(n + 1, dtype=np.uint64) * np.uint64(buf)
    total_out = n * buf

    d_in = torch.from_numpy(h_in).to(dev)
    d_in_off = torch.from_numpy(in_off.view(np.int64)).to(dev)
    d_out_off = torch.from_numpy(out_off.view(np.int64)).to(dev)
    d_out = torch.empty(total_out, dtype=torch.uint8, device=dev)
    d_sizes = torch.zeros(n, dtype=torch.int64, device=dev)
    d_status = torch.full((n,), -7, dtype=torch.int32, device=dev)
    del h_in

    stream = torch.cuda.current_stream(dev)
    eng = rsx.Engine(local_rank, stream.cuda_stream)

    def step():
        eng.batch_decompress_device(fmt, d_in.data_ptr(), d_in_off.data_ptr(), n, total_in, total_out,
                                    buf, d_out.data_ptr(), d_out_off.data_ptr(), d_sizes.data_ptr(),
                                    d_status.data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(None)
    barrier()
    dt = time.perf_counter() - t0
    # per-launch kernel time: a second, separately timed loop so reading the
    # events does not serialise the timed region above
    for k in range(args.steps):
        step()
        kernel_ms[k] = eng.last_kernel_ms()
    torch.cuda.synchronize(dev)

    # verification outside the timed region: every status 0, sizes exact, and the
    # first `distinct` outputs byte-identical to the original texts
    st = d_status.cpu().numpy()
    sz = d_sizes.cpu().numpy()
    ok = bool((st == 0).all() and (sz == buf).all())
    host_out = d_out[: distinct * buf].cpu().numpy()
    for i in range(distinct):
        ok = ok and host_out[i * buf:(i + 1) * buf].tobytes() == texts[i]
    if not ok:
        bad = np.nonzero((st != 0) | (sz != buf))[0]
        for i in bad[:6]:   # where the output of a failed stream first departs from its text
            got = d_out[int(i) * buf:(int(i) + 1) * buf].cpu().numpy()
            want = np.frombuffer(texts[int(i) % distinct], dtype=np.uint8)
            diff = np.nonzero(got != want)[0]
            p0 = int(diff[0]) if len(diff) else -1
            shift = [k for k in range(1, 64) if p0 >= 0 and p0 + k + 64 <= buf
                     and (got[p0 + k:p0 + k + 64] == want[p0:p0 + 64]).all()]
            if p0 >= 0:
                print("[bench]   want %r" % want[max(0, p0 - 24):p0 + 72].tobytes(), file=sys.stderr)
                print("[bench]   got  %r" % got[max(0, p0 - 24):p0 + 72].tobytes(), file=sys.stderr)
            print("[bench] stream %d status %d first mismatch at %d (%d differing bytes), "
                  "output continues the text after an insertion of %s bytes"
                  % (i, st[i], p0, len(diff), shift[:1] or "?"), file=sys.stderr)
        raise SystemExit("bench.py: GPU output does not match the input texts (status values %s, %d bad "
                         "streams, first %s, their status %s sizes %s)"
                         % (np.unique(st), len(bad), bad[:8], st[bad[:8]], sz[bad[:8]]))

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=args.reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        b = torch.tensor([total_in + total_out], dtype=torch.int64, device=args.reduce_device)
        dist.all_reduce(b, op=dist.ReduceOp.SUM)
        job_bytes = int(b.item())
    else:
        job_bytes = total_in + total_out

    if rank == 0:
        alg_bytes = total_in + total_out  # per launch on this rank
        k_ms = float(np.mean(kernel_ms))
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        res = {
            "metric": "batch GiB/s in+out (compress@L6, decompress) at 1/2/4/8 GPUs; % HBM roofline",
            "value": job_bytes * args.steps / dt / 2**30,
            "unit": "GiB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "batch gzip decompress, %d x %d KiB level-6 enwik-style text per GPU"
                            % (n, args.buf_kib),
                "op": "decompress", "format": "gzip", "level": level, "streams_per_gpu": n,
                "buf_bytes": buf, "distinct_buffers": distinct,
                "compressed_ratio": total_out / total_in,
                "compressor": "CPU restatement of the reference, 256 KiB sync-flushed segments",
                "parallelism": "batch sharded over %d GPU(s), no data-path collective" % world,
                "verified": ok,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": measured_traffic("decompress:%dx%dKiB:gzip:%d" % (n, args.buf_kib, level)),
                "kernel": "inflate_kernel", "kernel_ms": k_ms, "algorithmic_bytes": alg_bytes,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(streams, buf, fmt)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

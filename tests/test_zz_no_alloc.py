"""rsx_batch_decompress_device_noalloc (static device scratch, no allocation) against
rsx_batch_decompress_device on the same inputs: identical status, sizes and bytes, on two
calls in a row, the second with a smaller batch (the scratch keeps the first call's contents)."""
import numpy as np
import pytest

import corpus
import oracle_lib as o
import rsx_pkg

pytestmark = pytest.mark.gpu


def _device_batch(torch, dev, streams, slot_sizes):
    n = len(streams)
    in_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(s) for s in streams], out=in_off[1:])
    out_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(np.asarray(slot_sizes, dtype=np.uint64), out=out_off[1:])
    h_in = np.frombuffer(b"".join(streams), dtype=np.uint8).copy()
    return dict(
        n=n, total_in=int(in_off[-1]), total_out=int(out_off[-1]), max_out=int(max(slot_sizes)),
        d_in=torch.from_numpy(h_in).to(dev),
        d_in_off=torch.from_numpy(in_off.view(np.int64)).to(dev),
        d_out_off=torch.from_numpy(out_off.view(np.int64)).to(dev),
        out_off=out_off)


def _run(torch, dev, eng, fmt, b, noalloc):
    d_out = torch.full((b["total_out"] + 64,), 0xA5, dtype=torch.uint8, device=dev)
    d_sizes = torch.full((b["n"],), -1, dtype=torch.int64, device=dev)
    d_status = torch.full((b["n"],), -7, dtype=torch.int32, device=dev)
    call = eng.batch_decompress_device_noalloc if noalloc else eng.batch_decompress_device
    call(fmt, b["d_in"].data_ptr(), b["d_in_off"].data_ptr(), b["n"], b["total_in"], b["total_out"], b["max_out"],
         d_out.data_ptr(), b["d_out_off"].data_ptr(), d_sizes.data_ptr(), d_status.data_ptr())
    torch.cuda.synchronize(dev)
    return d_out.cpu().numpy(), d_sizes.cpu().numpy(), d_status.cpu().numpy()


def test_noalloc_matches_device_entry_point():
    import torch
    rsx = rsx_pkg.load()
    dev = torch.device("cuda", 0)
    eng = rsx.Engine(0, torch.cuda.current_stream(dev).cuda_stream)
    texts = [corpus.text(300000 + 1000 * i, 40 + i) for i in range(6)] + [corpus.mixed(70000, 3), b"", b"x" * 5000]
    for fmt, ofmt in ((rsx.FORMAT_GZIP, 2), (rsx.FORMAT_ZLIB, 1), (rsx.FORMAT_DEFLATE, 0)):
        streams = [o.compress(t, 6, ofmt) for t in texts]
        damaged = bytearray(streams[1])
        damaged[len(damaged) // 2] ^= 0x55
        first = _device_batch(torch, dev, streams + [bytes(damaged)], [len(t) for t in texts] + [len(texts[1])])
        # second call: fewer and smaller items, one slot too small
        second = _device_batch(torch, dev, streams[6:] + streams[:2],
                               [len(t) for t in texts[6:]] + [len(texts[0]), len(texts[1]) - 1])
        for b in (first, second):
            want = _run(torch, dev, eng, fmt, b, noalloc=False)
            got = _run(torch, dev, eng, fmt, b, noalloc=True)
            assert (got[2] == want[2]).all(), (got[2], want[2])
            assert (got[1] == want[1]).all()
            for i in range(b["n"]):
                if want[2][i] == 0:
                    a, z = int(b["out_off"][i]), int(b["out_off"][i]) + int(want[1][i])
                    assert (got[0][a:z] == want[0][a:z]).all(), i
        # the plain results are the oracle's
        want = _run(torch, dev, eng, fmt, first, noalloc=True)
        for i, t in enumerate(texts):
            assert want[2][i] == 0 and want[1][i] == len(t)
            a = int(first["out_off"][i])
            assert want[0][a:a + len(t)].tobytes() == t
        if fmt != rsx.FORMAT_DEFLATE:   # (raw DEFLATE has no checksum: a damaged stream may still decode)
            assert want[2][len(texts)] != 0
    # limits are refused with an engine-level error and nothing runs
    p = first["d_in_off"].data_ptr()   # (never dereferenced: the call is refused before anything runs)
    with pytest.raises(rsx.RsxError, match="too large for the static scratch"):
        eng.batch_decompress_device_noalloc(rsx.FORMAT_DEFLATE, p, p, 8193, 0, 0, 0, p, p, p, p)
    with pytest.raises(rsx.RsxError, match="too large for the static scratch"):
        eng.batch_decompress_device_noalloc(rsx.FORMAT_DEFLATE, p, p, 4, 0, (1 << 30) + 1, 1 << 20, p, p, p, p)
    eng.close()

"""Which zlib settings make valid streams that the one-kernel decoder has to take (GPU box, from the repo root)."""
import os
import sys
import zlib

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(_ROOT, "tests")); sys.path.insert(0, _ROOT)
import corpus
import rsx_pkg

rsx = rsx_pkg.load()
eng = rsx.default_engine()
STRAT = {"default": zlib.Z_DEFAULT_STRATEGY, "filtered": zlib.Z_FILTERED, "huffman": zlib.Z_HUFFMAN_ONLY, "rle": zlib.Z_RLE,
         "fixed": zlib.Z_FIXED}
datas = {"text": corpus.text(300000, 31), "mixed": corpus.mixed(300000, 32), "json": corpus.json_like(300000, 33),
         "random": corpus.random_bytes(300000, 34), "text8k": corpus.text(8000, 35)}
for mem in (1, 2, 3, 4, 6, 8, 9):
    for sname, strat in STRAT.items():
        for flush in (None, zlib.Z_SYNC_FLUSH, zlib.Z_BLOCK):
            streams, caps, names = [], [], []
            for name, d in datas.items():
                for lvl in (1, 6, 9):
                    co = zlib.compressobj(lvl, zlib.DEFLATED, 15, mem, strat)
                    parts = []
                    step = len(d) if flush is None else 20000
                    for at in range(0, len(d), step):
                        parts.append(co.compress(d[at:at + step]))
                        if flush is not None:
                            parts.append(co.flush(flush))
                    parts.append(co.flush())
                    streams.append(b"".join(parts)); caps.append(len(d)); names.append("%s/L%d" % (name, lvl))
            got, status = eng.batch_decompress(streams, caps, 1)
            assert all(int(s) == 0 for s in status) and all(g == datas[n.split("/")[0]] for g, n in zip(got, names))
            fb = eng.last_fallbacks()
            if fb:
                who = []
                for s, c, nm in zip(streams, caps, names):
                    eng.batch_decompress([s], [c], 1)
                    if eng.last_fallbacks():
                        who.append(nm)
                print("memLevel %d %-8s flush %-4s: %2d of %d decoded twice: %s" % (mem, sname, flush, fb, len(streams), " ".join(who)), flush=True)
print("done")

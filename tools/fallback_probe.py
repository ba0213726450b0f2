"""Lists which valid streams of a small zoo the one-kernel decoder has to take (GPU box, from the repo root)."""
import os
import sys
import zlib

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(_ROOT, "tests")); sys.path.insert(0, _ROOT)
import corpus
import oracle_lib as o
import rsx_pkg

rsx = rsx_pkg.load()
eng = rsx.default_engine()
WBITS = {0: -15, 1: 15, 2: 31}
datas = {"text300k": corpus.text(300000, 21), "mixed400k": corpus.mixed(400000, 22), "json200k": corpus.json_like(200000, 23),
         "random70k": corpus.random_bytes(70000, 24), "zeros100k": bytes(100000), "a": b"a", "empty": b"", "abab": b"ab" * 5000,
         "text5k": corpus.text(5000, 25)}
for fmt in (0, 1, 2):
    for name, d in datas.items():
        cases = [("oracle L%d" % lvl, o.compress(d, lvl, fmt)) for lvl in (1, 6, 9)]
        for lvl in (0, 1, 6, 9):
            co = zlib.compressobj(lvl, zlib.DEFLATED, WBITS[fmt])
            cases.append(("zlib L%d" % lvl, co.compress(d) + co.flush()))
        for who, z in cases:
            if z is None:
                continue
            for cap in (len(d), len(d) + 7):
                got, status = eng.batch_decompress([z], [cap], fmt)
                fb = eng.last_fallbacks()
                if fb or int(status[0]) != 0 or got[0] != d:
                    print("fmt %d %-10s %-9s cap +%d: status %d fallbacks %d ok %s (stream %d bytes)" %
                          (fmt, name, who, cap - len(d), int(status[0]), fb, got[0] == d, len(z)), flush=True)
print("done")

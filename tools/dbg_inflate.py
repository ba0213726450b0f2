import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, corpus, oracle_lib as o, rsx_pkg
rsx = rsx_pkg.load()
eng = rsx.Engine(0)
import zlib
def check(name, t, level=6):
    c = o.compress(t, level, 0)
    if c is None:
        z = zlib.compressobj(level, zlib.DEFLATED, -15)
        c = z.compress(t) + z.flush()
    out, st = eng.batch_decompress([c], [len(t)+64], 0)
    got = out[0]
    ok = got == t
    print(name, "len", len(t), "comp", len(c), "status", st[0], "ok", ok)
    if got is not None and not ok:
        a = np.frombuffer(got, dtype=np.uint8); b = np.frombuffer(t, dtype=np.uint8)
        n = min(len(a), len(b)); d = np.nonzero(a[:n] != b[:n])[0]
        print("   mismatches", len(d), "first", d[:12])
for size in (20000, 33000, 40000, 50000, 70000, 100000):
    check("text", corpus.text(size, 5))
rs = np.random.RandomState(1)
check("random", rs.bytes(100000))
check("zeros", bytes(200000))
check("period7", (b"abcdefg" * 30000)[:200000])
check("mixed", corpus.mixed(200000, 3))
check("period300", (bytes(range(256)) + b"0123456789" * 5)[:300] * 1000)
check("json", corpus.json_like(300000, 4), 9)
check("text1M", corpus.text(1 << 20, 77))

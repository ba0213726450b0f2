import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import corpus, rsx_pkg, zlib
rsx = rsx_pkg.load(); eng = rsx.default_engine()
sets = {"text": [corpus.text(1 << 20, s) for s in range(4)], "mixed": [corpus.mixed(1 << 20, 7 + s) for s in range(4)],
        "json": [corpus.json_like(1 << 20, 21 + s) for s in range(2)]}
for lvl in [int(x) for x in (sys.argv[1:] or ["1", "6", "9"])]:
    out = []
    for k, v in sets.items():
        comp, st = eng.batch_compress(v, 0, lvl)
        assert all(zlib.decompress(c, -15) == d for c, d in zip(comp, v))
        out.append("%s %.4f" % (k, sum(map(len, v)) / sum(map(len, comp))))
    print("L%d" % lvl, "  ".join(out), flush=True)

import sys, gzip, zlib
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import corpus, rsx_pkg
rsx = rsx_pkg.load(); eng = rsx.default_engine()
a, b = corpus.text(300000, 1), corpus.text(200000, 2)
ga, gb = gzip.compress(a, 6), gzip.compress(b, 6)
for name, data, cap in (("two members", ga + gb, 600000), ("member + zeros", ga + bytes(100), 600000), ("member + junk", ga + b"hello world junk 1234567", 600000)):
    out, st = eng.batch_decompress([data], [cap], rsx.FORMAT_GZIP)
    print(name, list(st), None if out[0] is None else (len(out[0]), out[0] == a))

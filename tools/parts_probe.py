"""Decodes one large gzip stream (compressed by the oracle and by the engine) and prints how many parts it was cut into
and whether it had to be decoded as a whole after all (GPU box, from the repo root)."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import corpus, oracle_lib as o, rsx_pkg
rsx = rsx_pkg.load()
eng = rsx.default_engine()
for size in (1 << 20, 5 << 20, 16 << 20):
    data = corpus.text(size, 11)
    for who in ("oracle", "engine"):
        if who == "oracle":
            stream = o.compress(data, 6, o.FMT_GZIP)
        else:
            comp, st = eng.batch_compress([data], rsx.FORMAT_GZIP, 6)
            stream = comp[0]
        got, status = eng.batch_decompress([stream], [len(data)], rsx.FORMAT_GZIP)
        print(size, who, list(status), got[0] == data, "parts", eng.last_parts(), "whole", eng.last_whole(), flush=True)

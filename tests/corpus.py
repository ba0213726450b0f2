"""Seeded synthetic inputs shared by the tests and bench.py (no files, no network)."""
import random

import numpy as np


def _vocab(rnd, n_words):
    letters = b"etaoinshrdlcumwfgypbvkjxqz"
    weights = [12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4, 2.2, 2.0,
               2.0, 1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07]
    words = []
    for _ in range(n_words):
        k = rnd.randint(2, 12)
        words.append(bytes(rnd.choices(letters, weights, k=k)))
    return words


_VOCAB = {}


def _decorated_vocab(n_words):
    """(plain, link, tag) byte-string variants of one shared vocabulary"""
    if n_words not in _VOCAB:
        words = _vocab(random.Random(12345), n_words)
        _VOCAB[n_words] = (np.array([w + b" " for w in words], dtype=object),
                           np.array([b"[[" + w + b"]] " for w in words], dtype=object),
                           np.array([b"<" + w + b">" for w in words], dtype=object),
                           np.array([w + b". " for w in words], dtype=object))
    return _VOCAB[n_words]


def text(n_bytes, seed=0, n_words=50000, zipf_s=1.12):
    """'enwik-style' text: Zipf-distributed words from a synthetic vocabulary with
    sentences, [[links]], <tags> and numbers."""
    plain, link, tag, stop = _decorated_vocab(n_words)
    rs = np.random.RandomState(seed)
    parts = []
    total = 0
    while total < n_bytes:
        ranks = rs.zipf(zipf_s, size=1 << 15)
        ranks = ranks[ranks <= n_words] - 1
        x = rs.rand(ranks.size)
        tok = plain[ranks]
        m = x < 0.01
        tok[m] = link[ranks[m]]
        m = (x >= 0.01) & (x < 0.015)
        tok[m] = tag[ranks[m]]
        m = (x >= 0.015) & (x < 0.03)
        tok[m] = [b"%d " % v for v in rs.randint(0, 100000, size=int(m.sum()))]
        m = (x >= 0.03) & (x < 0.10)
        tok[m] = stop[ranks[m]]
        m = x >= 0.995
        tok[m] = [t + b"\n" for t in tok[m]]
        chunk = b"".join(tok.tolist())
        parts.append(chunk)
        total += len(chunk)
    return b"".join(parts)[:n_bytes]


def mixed(n_bytes, seed=0):
    """alternating 64 KiB regions: text, little-endian u32 records, 1/8 random bytes"""
    rs = np.random.RandomState(seed)
    out = bytearray()
    k = 0
    while len(out) < n_bytes:
        kind = k % 3
        if kind == 0:
            out += text(65536, seed * 131 + k)
        elif kind == 1:
            base = rs.randint(0, 1 << 30)
            rec = np.arange(base, base + 16384, dtype="<u4")
            out += rec.tobytes()
        else:
            b = np.zeros(65536, dtype=np.uint8)
            idx = rs.rand(65536) < 0.125
            b[idx] = rs.randint(0, 256, size=int(idx.sum()))
            out += b.tobytes()
        k += 1
    return bytes(out[:n_bytes])


def json_like(n_bytes, seed=0):
    rnd = random.Random(seed)
    keys = ["id", "timestamp", "user", "status", "region", "latency_ms", "bytes", "method", "path",
            "code", "agent", "flag"]
    enums = ["ok", "error", "pending", "retry", "us-east-1", "eu-west-2", "GET", "POST", "PUT"]
    out = bytearray(b"[")
    i = 0
    while len(out) < n_bytes:
        obj = ",".join('"%s":%s' % (k, ('"%s"' % rnd.choice(enums)) if j % 3 == 1 else
                                    ('"2026-%02d-%02dT%02d:%02d:%02dZ"' % (rnd.randint(1, 12), rnd.randint(1, 28),
                                                                      rnd.randint(0, 23), rnd.randint(0, 59),
                                                                      rnd.randint(0, 59))) if j % 3 == 2
                                    else str(rnd.randint(0, 5000)))
                       for j, k in enumerate(keys))
        out += b"{" + obj.encode() + b"},"
        i += 1
    return bytes(out[:n_bytes])


def random_bytes(n_bytes, seed=0):
    return np.random.RandomState(seed).randint(0, 256, size=n_bytes, dtype=np.uint8).tobytes()


def periodic(period, n_bytes, seed=0):
    """repeating pattern of the given period (tests/offset_tests.rs shapes)"""
    p = random_bytes(period, seed + period)
    return (p * (n_bytes // period + 1))[:n_bytes]

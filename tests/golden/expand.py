"""Expands the data descriptors used by tests/golden/*.json into bytes."""


def expand(d):
    if "ascii" in d:
        return d["ascii"].encode("latin-1")
    if "hex" in d:
        return bytes.fromhex(d["hex"])
    g = d["gen"]
    if g == "fill":
        return bytes([d["byte"]]) * d["n"]
    if g == "mod":  # byte i = (i * mul) % mod
        return bytes(((i * d["mul"]) % d["mod"]) & 0xFF for i in range(d["n"]))
    if g == "repeat_prefix":  # first n bytes of pattern repeated
        p = d["pattern"].encode("latin-1")
        return (p * (d["n"] // len(p) + 1))[: d["n"]]
    raise ValueError(g)

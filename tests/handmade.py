"""Dynamic DEFLATE block headers assembled bit by bit: cases no compressor writes (shared by the oracle's
CPU test and the GPU parity test)."""

class Bits:
    """LSB-first bit writer (RFC 1951 3.1.1); Huffman codewords go in most significant bit first."""

    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, value, nbits):
        self.acc |= value << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 0xFF)
            self.acc >>= 8
            self.n -= 8

    def code(self, codeword, nbits):
        self.put(int(format(codeword, "0%db" % nbits)[::-1], 2) if nbits else 0, nbits)

    def bytes(self):
        return bytes(self.out) + (bytes([self.acc]) if self.n else b"")


def canonical(lens):
    """codeword of every symbol with a non-zero length (RFC 1951 3.2.2)"""
    codes, code = {}, 0
    for l in range(1, 16):
        for s, sl in enumerate(lens):
            if sl == l:
                codes[s] = code
                code += 1
        code <<= 1
    return codes


def dynamic_block(litlen_lens, offset_lens, precode_lens, tokens, literals, final=True):
    """One dynamic block whose code lengths are sent as the given precode tokens ((symbol, extra value)
    pairs), followed by the literals and the end of block."""
    w = Bits()
    w.put(1 if final else 0, 1)
    w.put(2, 2)
    w.put(len(litlen_lens) - 257, 5)
    w.put(len(offset_lens) - 1, 5)
    perm = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    w.put(19 - 4, 4)
    for sym in perm:
        w.put(precode_lens[sym], 3)
    pc = canonical(precode_lens)
    for sym, extra in tokens:
        w.code(pc[sym], precode_lens[sym])
        if sym >= 16:
            w.put(extra, {16: 2, 17: 3, 18: 7}[sym])
    lc = canonical(litlen_lens)
    for b in literals:
        w.code(lc[b], litlen_lens[b])
    w.code(lc[256], litlen_lens[256])
    return w.bytes()



def header_cases():
    """(streams, caps, text): a 2200-bit header of single code lengths, three headers with runs of every
    kind (the last with a run across the end of the lengths, which is clamped), a repeat as the first token
    whole and cut inside its codeword, and truncations of the long header all along."""
    # precode: symbol 0 one bit, 1..5 two to six bits, 8 and 9 seven bits, the run symbols unused
    pl = [0] * 19
    pl[0], pl[1], pl[2], pl[3], pl[4], pl[5], pl[8], pl[9] = 1, 2, 3, 4, 5, 6, 7, 7
    litlen = [8 if i % 5 != 4 else 9 for i in range(286)]   # 229 eights and 57 nines: one 512th too many
    eights = [i for i in range(286) if litlen[i] == 8]
    for i in eights[:3]:
        litlen[i] = 9                                         # 226 / 256 + 60 / 512 = 1
    offset = [4, 4] + [5] * 28
    text = bytes(range(0, 250, 7)) * 3
    long_header = dynamic_block(litlen, offset, pl, [(l, 0) for l in litlen + offset], text)
    assert len(long_header) > 300
    streams, caps = [long_header], [len(text)]
    # runs of every kind (repeat the previous length, short and long runs of zeros), one of them across the
    # end of the lengths (clamped, src/decompress/mod.rs:462-467): 254 eights, four nines, 28 zeros, then
    # two fours and 28 fives for the offsets
    for zeros, clamp in (([(18, 17)], False), ([(17, 7), (17, 7), (17, 5)], False), ([(18, 17)], True)):
        pl2 = [0] * 19
        pl2[16], pl2[8], pl2[9], pl2[zeros[0][0]], pl2[4], pl2[5] = 2, 2, 2, 3, 4, 4
        toks = [(8, 0)] + [(16, 3)] * 42 + [(8, 0), (9, 0), (16, 0)] + zeros
        toks += [(4, 0), (4, 0), (5, 0)] + [(16, 3)] * 4 + [(16, 3 if clamp else 0)]
        lit2 = [8] * 254 + [9] * 4 + [0] * 28
        streams.append(dynamic_block(lit2, [4, 4] + [5] * 28, pl2, toks, bytes(range(200))))
        caps.append(200)
    # a repeat with nothing in front of it, whole and cut inside its codeword
    bad = dynamic_block(litlen, offset, pl2, [(16, 0)], b"")
    streams += [bad, bad[:9]]
    caps += [10, 10]
    # truncations of the long header and of its data
    for cut in list(range(0, len(long_header), 7)) + [len(long_header) - 1]:
        streams.append(long_header[:cut])
        caps.append(len(text))
    return streams, caps, text

"""A plain DEFLATE block walker for the tests (RFC 1951): per block its type, the code lengths its dynamic header
declares and the symbol frequencies of its body.  Slow and simple on purpose: it is what the compressor's headers
are checked against, and shares no code with the engine or the oracle."""

LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
PERM = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


def _decoder(lens):
    """canonical Huffman: {(length, codeword read MSB first): symbol}"""
    maxl = max(lens) if lens else 0
    count = [0] * (maxl + 2)
    for l in lens:
        if l:
            count[l] += 1
    nxt, code = [0] * (maxl + 2), 0
    for l in range(1, maxl + 1):
        code = (code + count[l - 1]) << 1
        nxt[l] = code
    table = {}
    for sym, l in enumerate(lens):
        if l:
            table[(l, nxt[l])] = sym
            nxt[l] += 1
    return table


def walk(data, max_blocks=None):
    """-> list of dicts: type, bits, litlen_lens, offset_lens (dynamic blocks), litlen_freq[288], offset_freq[32]"""
    pos = 0

    def bits(n):
        nonlocal pos
        v = 0
        for i in range(n):
            v |= ((data[pos >> 3] >> (pos & 7)) & 1) << i
            pos += 1
        return v

    def symbol(table):
        code = l = 0
        while True:
            code = (code << 1) | bits(1)
            l += 1
            if (l, code) in table:
                return table[(l, code)]
            if l > 15:
                raise ValueError("no such codeword")

    blocks = []
    while True:
        start = pos
        final, typ = bits(1), bits(2)
        blk = {"type": typ, "litlen_freq": [0] * 288, "offset_freq": [0] * 32}
        if typ == 0:
            pos = (pos + 7) & ~7
            n = bits(16)
            bits(16)
            pos += 8 * n
        elif typ == 3:
            raise ValueError("reserved block type")
        else:
            if typ == 1:
                ll, dl = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8, [5] * 30
            else:
                hlit, hdist, hclen = bits(5) + 257, bits(5) + 1, bits(4) + 4
                pre = [0] * 19
                for i in range(hclen):
                    pre[PERM[i]] = bits(3)
                pt, lens = _decoder(pre), []
                while len(lens) < hlit + hdist:
                    s = symbol(pt)
                    if s < 16:
                        lens.append(s)
                    elif s == 16:
                        lens += [lens[-1]] * (3 + bits(2))
                    elif s == 17:
                        lens += [0] * (3 + bits(3))
                    else:
                        lens += [0] * (11 + bits(7))
                ll, dl = lens[:hlit], lens[hlit:hlit + hdist]
                blk["litlen_lens"], blk["offset_lens"] = ll, dl
            lt, dt = _decoder(ll), _decoder(dl)
            while True:
                s = symbol(lt)
                blk["litlen_freq"][s] += 1
                if s == 256:
                    break
                if s > 256:
                    bits(LEN_EXTRA[s - 257])
                    d = symbol(dt)
                    blk["offset_freq"][d] += 1
                    bits(DIST_EXTRA[d])
        blk["bits"] = pos - start
        blocks.append(blk)
        if final or (max_blocks and len(blocks) >= max_blocks):
            return blocks

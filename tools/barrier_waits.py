#!/usr/bin/env python3
"""Finds workgroup barriers that can be reached with an LDS access still outstanding.

usage: barrier_waits.py <listing.s> [<kernel-name-substring>]

The listing is what `hipcc -S --offload-arch=gfx950` writes.  For every kernel (or the ones whose
symbol contains the substring) the basic blocks are cut at labels and branches, and one forward
data-flow pass carries a single bit per block: "a ds_* instruction was issued on some path to
here and no s_waitcnt with lgkmcnt(0) has followed it".  Every s_barrier reached with that bit set
is printed with the instructions in front of it.  A wave that enters the barrier in that state has
LDS writes the other waves of the workgroup may not see after the barrier.
"""
import re
import sys


def kernels(lines):
    out, name, start = [], None, 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", l)
        if m and not l.startswith(".L"):
            if name is not None:
                out.append((name, start, i))
            name, start = m.group(1), i
        if l.strip().startswith(".end_amdhsa_kernel") or l.strip() == "s_endpgm" and False:
            pass
    if name is not None:
        out.append((name, start, len(lines)))
    return out


def analyse(lines, start, end):
    # instructions and labels of the function body
    ins = []      # (line_no, text)
    labels = {}   # label -> index into ins of the first instruction after it
    for i in range(start + 1, end):
        t = lines[i].split(";")[0].rstrip()
        s = t.strip()
        if not s:
            continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if s.startswith(".") or s.endswith(":"):
            if s.startswith(".Lfunc_end"):
                break
            continue
        ins.append((i + 1, s))
    if not ins:
        return 0, []
    # block leaders
    leaders = {0} | set(labels.values())
    for k, (_, s) in enumerate(ins):
        op = s.split()[0]
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            if k + 1 < len(ins):
                leaders.add(k + 1)
    order = sorted(x for x in leaders if x < len(ins))
    block_of = {}
    blocks = []
    for bi, a in enumerate(order):
        b = order[bi + 1] if bi + 1 < len(order) else len(ins)
        blocks.append((a, b))
        block_of[a] = bi
    succ = [[] for _ in blocks]
    for bi, (a, b) in enumerate(blocks):
        s = ins[b - 1][1]
        op = s.split()[0]
        tgt = s.split()[-1]
        if op == "s_branch":
            if tgt in labels and labels[tgt] in block_of:
                succ[bi].append(block_of[labels[tgt]])
            continue
        if op in ("s_endpgm", "s_setpc_b64"):
            continue
        if op.startswith("s_cbranch") and tgt in labels and labels[tgt] in block_of:
            succ[bi].append(block_of[labels[tgt]])
        if bi + 1 < len(blocks):
            succ[bi].append(bi + 1)

    def transfer(bi, pending, report=None):
        a, b = blocks[bi]
        for k in range(a, b):
            s = ins[k][1]
            op = s.split()[0]
            if op.startswith("ds_") and not op.startswith(("ds_bpermute", "ds_permute", "ds_swizzle")):
                pending = True      # the cross-lane ds_* forms move registers only, no LDS memory
            elif op == "s_waitcnt" and re.search(r"lgkmcnt\(0\)", s):
                pending = False
            elif op == "s_waitcnt" and not re.search(r"[a-z]+cnt\(", s):
                # numeric form: lgkmcnt is bits 11:8
                try:
                    v = int(s.split()[1], 0)
                    if (v >> 8) & 0xF == 0:
                        pending = False
                except ValueError:
                    pass
            elif op == "s_barrier" and report is not None:
                report.append((k, pending))
        return pending

    entry = [False] * len(blocks)
    changed = True
    while changed:
        changed = False
        for bi in range(len(blocks)):
            out = transfer(bi, entry[bi])
            for s in succ[bi]:
                if out and not entry[s]:
                    entry[s] = True
                    changed = True
    found = []
    for bi in range(len(blocks)):
        transfer(bi, entry[bi], found)
    bad = []
    for k, pending in found:
        if pending:
            bad.append((ins[k][0], [x[1] for x in ins[max(0, k - 6):k + 1]]))
    return len(found), bad


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else None
    lines = open(path).read().split("\n")
    for name, a, b in kernels(lines):
        if want and want not in name:
            continue
        nb, bad = analyse(lines, a, b)
        if nb == 0:
            continue
        print("%s: %d barriers, %d reachable with an LDS access outstanding" % (name, nb, len(bad)))
        for line, ctx in bad:
            print("  line %d:" % line)
            for c in ctx:
                print("      " + c)


if __name__ == "__main__":
    main()

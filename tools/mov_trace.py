#!/usr/bin/env python3
"""Register copies in a kernel of a gfx950 listing, each with the instruction that made the copied value and the first
instruction that reads the copy: python tools/mov_trace.py <listing.s> <kernel name prefix>"""
import re, sys
s = open(sys.argv[1]).read()
f = [x for x in re.split(r"\n\s*\.globl\s+", s)[1:] if x.startswith(sys.argv[2])][0]
body = f.split(".end_amdhsa_kernel")[0]
lines = [l.strip() for l in body.split("\n") if l.startswith("\t") and l.strip() and l.strip()[0] not in ".;"]
pat = re.compile(r"v\[(\d+):(\d+)\]")


def reads(t, num, skip_dst=True):
    ops = t.split(None, 1)[1] if " " in t else ""
    parts = ops.split(",")
    if skip_dst:
        parts = parts[1:]
    ops = ",".join(parts)
    if re.search(r"\bv%d\b" % num, ops):
        return True
    return any(int(a) <= num <= int(b) for a, b in pat.findall(ops))


def writes(t, num):
    ops = t.split(None, 1)[1] if " " in t else ""
    d = ops.split(",")[0].strip()
    return d == "v%d" % num or any(int(a) <= num <= int(b) for a, b in pat.findall(d))


n_copy = 0
for n, l in enumerate(lines):
    if not l.startswith("v_mov_b32"):
        continue
    dst, src = [x.strip() for x in l.split(None, 1)[1].split(",")[:2]]
    if not src.startswith("v"):
        print(n, l)
        continue
    n_copy += 1
    num, sn = int(dst[1:]), int(src[1:])
    prod = next(((m, lines[m]) for m in range(n - 1, -1, -1) if writes(lines[m], sn)), None)
    use = next(((m, lines[m]) for m in range(n + 1, len(lines)) if reads(lines[m], num)), None)
    later = []
    for m in range(n + 1, len(lines)):
        if reads(lines[m], sn):
            later.append((m, lines[m]))
        if writes(lines[m], sn) or len(later) >= 2:
            break
    print(n, l, "\n     made by", prod, "\n     read by", use, "\n     source read later by", later)
print(n_copy, "register copies of", len(lines), "instructions")

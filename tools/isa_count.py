#!/usr/bin/env python3
"""Instruction census of the kernels in a gfx950 assembly listing (hipcc -S)."""
import collections, re, sys
s = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/mppi_Reacher7Dof.s").read()
pat = sys.argv[2] if len(sys.argv) > 2 else "rollout_kernel"
for f in re.split(r"\n\s*\.globl\s+", s)[1:]:
    name = f.split("\n", 1)[0].strip()
    if pat not in name:
        continue
    body = f.split(".end_amdhsa_kernel")[0]
    ins = []
    for l in body.split("\n"):
        t = l.strip()
        if not l.startswith("\t") or not t or t[0] in ".;":
            continue
        ins.append(t.split()[0])
    c = collections.Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print(name[:64], "total", sum(c.values()), "valu", valu, "mov", c["v_mov_b32_e32"], "accvgpr",
          sum(v for k, v in c.items() if "accvgpr" in k), "lane", sum(v for k, v in c.items() if "lane" in k),
          "scratch", sum(v for k, v in c.items() if "scratch" in k))
    print("   ", c.most_common(12))

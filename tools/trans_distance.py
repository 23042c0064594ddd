#!/usr/bin/env python3
"""For every transcendental instruction of a kernel in an assembly listing: how many instructions later its
result is first read (a lone wave stalls for the unit's latency when that distance is short).
    python tools/trans_distance.py /tmp/mppi_Reacher7Dof.s 'rollout_kernelIfN4mppi11Reacher7DofELb1'"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(sys.argv[2]), l)][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
ins = [l.strip() for l in lines[start:end] if l.strip() and not l.strip().startswith((";", "."))]
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_", "v_exp_", "v_log_")
hist = Counter()
for i, s in enumerate(ins):
    if s.startswith(TRANS):
        m = re.match(r"(\S+)\s+(v\d+)", s)
        dst = m.group(2)
        d = None
        for j in range(i + 1, min(i + 60, len(ins))):
            ops = ins[j].split(None, 1)
            if len(ops) > 1 and re.search(r"\b%s\b" % dst, ops[1].split(",", 1)[1] if "," in ops[1] else ""):
                d = j - i
                break
            if ins[j].startswith(("s_cbranch", "s_branch")):
                d = "br+%d" % (j - i)
                break
        hist[(m.group(1), d)] += 1
        print("%5d %-16s first use after %s" % (i, m.group(1), d))

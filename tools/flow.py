#!/usr/bin/env python3
"""Control-flow skeleton of one kernel in a device assembly listing: every label and branch with the number
of instructions between them (listing: make -C trajopt_amd/csrc asm).    python tools/flow.py /tmp/mppi_Reacher7Dof.s 'rollout_kernelIfN4mppi11Reacher7DofELb1'"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(sys.argv[2]), l)][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
n = 0
for l in lines[start:end]:
    s = l.strip()
    if not s or s.startswith(";"):
        continue
    if s.startswith(".LBB"):
        print("%5d  %s" % (n, s[:110]))
        n = 0
    elif s.startswith(("s_cbranch", "s_branch", "s_endpgm")):
        print("%5d      %s" % (n, s))
        n = 0
    elif not s.startswith("."):
        n += 1

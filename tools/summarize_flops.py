#!/usr/bin/env python3
"""Floating-point operations the rollout kernel executes, from hardware counters and the kernel's listing.

    python tools/summarize_flops.py <pmc dir> [<asm listing>] > profiles/<round>_rollout_flops.json

The pmc dir holds a rocprofv3 pass over tools/ksweep.py 4096 with SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32 (one count
per wave instruction, packed or not).  A packed instruction (v_pk_fma_f32 and friends) works on two values per lane;
its share among the instructions of each kind is taken from the listing (make -C trajopt_amd/csrc asm) of the
planning-only kernel variant.  Executed FLOP per lane and substep = 2 FMA (1 + packed share) + MUL (1 + share)
+ ADD (1 + share) + TRANS."""
import collections
import json
import re
import sys

from summarize_pmc import K, H, FRAME_SKIP, launches

ASM = sys.argv[2] if len(sys.argv) > 2 else "/tmp/mppi_Reacher7Dof.s"
KERNEL = "rollout_kernelIfN4mppi11Reacher7DofELb1"


def packed_shares():
    s = open(ASM).read()
    for f in re.split(r"\n\s*\.globl\s+", s)[1:]:
        if KERNEL not in f.split("\n", 1)[0]:
            continue
        c = collections.Counter()
        for line in f.split(".end_amdhsa_kernel")[0].split("\n"):
            t = line.strip()
            if line.startswith("\t") and t and t[0] not in ".;":
                c[t.split()[0]] += 1
        fma = sum(v for k, v in c.items() if k.startswith(("v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32")))
        mul = sum(v for k, v in c.items() if k.startswith("v_mul_f32"))
        add = sum(v for k, v in c.items() if k.startswith(("v_add_f32", "v_sub_f32", "v_subrev_f32")))
        return {"fma": c["v_pk_fma_f32"] / (fma + c["v_pk_fma_f32"]), "mul": c["v_pk_mul_f32"] / (mul + c["v_pk_mul_f32"]),
                "add": c["v_pk_add_f32"] / (add + c["v_pk_add_f32"])}
    raise SystemExit("kernel not found in " + ASM)


vals, n = launches(sys.argv[1])
waves, substeps = (K + 63) // 64, H * FRAME_SKIP
per = {k: v / (waves * substeps) for k, v in vals.items()}
sh = packed_shares()
flop = (2.0 * per["SQ_INSTS_VALU_FMA_F32"] * (1 + sh["fma"]) + per["SQ_INSTS_VALU_MUL_F32"] * (1 + sh["mul"])
        + per["SQ_INSTS_VALU_ADD_F32"] * (1 + sh["add"]) + per["SQ_INSTS_VALU_TRANS_F32"])
print(json.dumps({"per_wave_substep": per, "packed_share_in_listing": sh, "launches_averaged": n,
                  "executed_flop_per_lane_substep": flop, "executed_flop_per_env_step": flop * FRAME_SKIP,
                  "note": "counters: one count per wave instruction; a packed instruction counts once and does two operations per "
                          "lane (share from the static listing of the planning-only kernel)"}, indent=1, sort_keys=True))

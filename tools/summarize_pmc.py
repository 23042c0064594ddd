#!/usr/bin/env python3
"""Turn rocprofv3 --pmc output (counter_collection.csv) into the per-launch summaries kept under profiles/.

    python tools/summarize_pmc.py sq    <dir> [<dir> ...] > profiles/<round>_rollout_pmc_sq.json
    python tools/summarize_pmc.py hbm   <fetch_dir> <write_dir> > profiles/<round>_rollout_traffic.json
    python tools/summarize_pmc.py hbm_of <kernel name part> <algorithmic bytes> <fetch_dir> <write_dir>      (any kernel)

Counters are averaged over the launches of rollout_kernel<float, Reacher7Dof> found in the files.  The HBM
summary applies the gfx950 correction for FETCH_SIZE (it tallies 128-byte read requests as 64 bytes, so
coalesced streaming reads are reported at half their size; WRITE_SIZE is exact).
"""
import csv
import glob
import json
import os
import sys

KERNEL = "rollout_kernel<float, mppi::Reacher7Dof,"      # both variants (general, planning-only)
K, H, NV, FRAME_SKIP = 4096, 64, 7, 2


def launches(directory, kernel=None):
    kernel = kernel or KERNEL
    per = {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if kernel in row["Kernel_Name"]:
                per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
                per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {name: sum(v.values()) / len(v) for name, v in per.items()}, max((len(v) for v in per.values()), default=0)


def main():
    mode, dirs = sys.argv[1], sys.argv[2:]
    if mode == "sq":
        out, n = {}, 0
        for d in dirs:
            vals, n = launches(d)
            out.update(vals)
        waves, substeps = (K + 63) // 64, H * FRAME_SKIP
        res = dict(out)
        res["launches_averaged"] = n
        res["note"] = ("rocprofv3 --pmc, %s, K=%d H=%d (%d waves x %d substeps); SQ_* cycle counters are in quad-cycles"
                       % (KERNEL, K, H, waves, substeps))
        res["per_wave_substep"] = {k: v / (waves * substeps) for k, v in out.items()}
        print(json.dumps(res, indent=1, sort_keys=True))
    elif mode == "hbm_of":
        kernel, alg = dirs[0], float(dirs[1])
        fetch, n = launches(dirs[2], kernel)
        write, _ = launches(dirs[3], kernel)
        f_kb, w_kb = fetch["FETCH_SIZE"], write["WRITE_SIZE"]
        print(json.dumps({
            "kernel": kernel, "launches_averaged": n,
            "FETCH_SIZE_KB_per_launch": f_kb, "WRITE_SIZE_KB_per_launch": w_kb,
            "correction": "gfx950: FETCH_SIZE counts half the bytes of coalesced streaming reads (doubled here); WRITE_SIZE exact",
            "hbm_bytes_per_launch": 2.0 * f_kb * 1024 + w_kb * 1024,
            "algorithmic_bytes_per_launch": alg,
        }, sort_keys=True))
    else:
        fetch, n = launches(dirs[0])
        write, _ = launches(dirs[1])
        f_kb, w_kb = fetch["FETCH_SIZE"], write["WRITE_SIZE"]
        print(json.dumps({
            "kernel": KERNEL, "K": K, "H": H, "launches_averaged": n,
            "FETCH_SIZE_KB_per_launch": f_kb, "WRITE_SIZE_KB_per_launch": w_kb,
            "correction": "gfx950: FETCH_SIZE counts half the bytes of coalesced streaming reads (doubled here); WRITE_SIZE exact",
            "hbm_bytes_per_launch": 2.0 * f_kb * 1024 + w_kb * 1024,
            "algorithmic_bytes_per_launch": K * H * NV * 4 + K * 4 + H * NV * 8,
            "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/ksweep.py 4096",
        }, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Averages of the counters in a rocprofv3 --pmc output directory for the kernels whose name contains argv[2]:
    python tools/pmc_kernel.py <dir> <substring>"""
import csv, glob, json, os, sys
per = {}
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if sys.argv[2] in row["Kernel_Name"]:
            per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
print(json.dumps({k: sum(v.values()) / len(v) for k, v in per.items()}, indent=1, sort_keys=True))

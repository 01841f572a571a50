#!/usr/bin/env python3
"""Mean of every counter per kernel from rocprofv3 --pmc passes (one *counter_collection.csv per pass directory).
usage: pmc_kernel_means.py <out.json> <pass_dir> [<pass_dir> ...]   (kernels of namespace d3p only)"""
import collections
import csv
import glob
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per_dispatch = collections.defaultdict(float)   # a counter is reported per XCD / SE instance: sum them per dispatch
        names = {}
        for r in csv.DictReader(open(f)):
            if "d3p::" not in r["Kernel_Name"]:
                continue
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for (disp, counter), v in per_dispatch.items():
            acc[names[disp]][counter].append(v)
out = {}
for k, counters in acc.items():
    short = k.split("(")[0].replace("void ", "")
    out[short] = {"dispatches": max(len(v) for v in counters.values())}
    for c, v in sorted(counters.items()):
        out[short][c] = sum(v) / len(v)
    o = out[short]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in o and "SQ_BUSY_CYCLES" in o and o["SQ_BUSY_CYCLES"]:
        o["mfma_busy_over_sq_busy"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / o["SQ_BUSY_CYCLES"]
    if "SQ_VALU_MFMA_COEXEC_CYCLES" in o and o.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        o["coexec_over_mfma_busy"] = o["SQ_VALU_MFMA_COEXEC_CYCLES"] / o["SQ_VALU_MFMA_BUSY_CYCLES"]
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))

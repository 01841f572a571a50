#!/usr/bin/env python3
"""usage: fetch_calibration_report.py <fetch_pass_dir> <write_pass_dir> <out.json>  (passes of tools/probes/fetch_calibration.py)"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_hbm_copy" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in acc.items()}


moved = float(1 << 30)
fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"bytes_moved_each_way_per_launch": moved, "by_kernel": {}}
for k in fetch:
    targ = k.split("k_hbm_copy<", 1)[1].split(",", 1)[0]          # the element type: the template's first argument
    width = 16 if "vector" in targ else 8 if "long" in targ else 4
    out["by_kernel"][f"{width}_bytes_per_lane"] = {
        "kernel": k, "FETCH_SIZE_KiB": fetch[k], "WRITE_SIZE_KiB": write.get(k),
        "fetch_reported_over_moved": fetch[k] * 1024.0 / moved,
        "write_reported_over_moved": write[k] * 1024.0 / moved if k in write else None}
out["note"] = ("coalesced streaming copy of 1 GiB (nontemporal loads / stores, 4 loads in flight per thread): the factor by which "
               "FETCH_SIZE x 1024 under- or over-reports the bytes read, per access width")
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))

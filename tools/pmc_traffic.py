#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into HBM bytes per launch of
the dominant kernel, following /opt/skills/guides/MI355X_MICROARCH.md section HBM:
  * counters come from separate passes (TCC slots), values are in KiB;
  * on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of wide (16 B/lane) coalesced streaming
    reads -> doubled; WRITE_SIZE is exact for 16-B stores.
usage: pmc_traffic.py <fetch_pass_dir> <write_pass_dir> <out.json>"""
import csv
import glob
import json
import sys


def per_launch(d, counter, kernel="k_logreg_main"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals) / len(vals), len(vals)


fetch_kib, n1 = per_launch(sys.argv[1], "FETCH_SIZE")
write_kib, n2 = per_launch(sys.argv[2], "WRITE_SIZE")
fetch_bytes = 2.0 * fetch_kib * 1024.0
write_bytes = write_kib * 1024.0
out = {"kernel": "k_logreg_main", "launches": [n1, n2], "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
       "fetch_bytes_corrected_x2": fetch_bytes, "write_bytes": write_bytes,
       "hbm_bytes_per_launch": fetch_bytes + write_bytes,
       "note": "FETCH_SIZE doubled per the gfx950 correction for 16-B/lane coalesced reads; separate --pmc passes"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))

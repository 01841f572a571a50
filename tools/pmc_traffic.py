#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into HBM bytes per DP-VI step / per launch of
the dominant step kernel, following /opt/skills/guides/MI355X_MICROARCH.md section HBM:
  * counters come from separate passes (TCC slots), values are in KiB;
  * on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of wide (16 B/lane) coalesced streaming
    reads -> doubled; WRITE_SIZE is exact for 16-B stores.
The dominant kernel is the k_logreg_* kernel with the largest summed FETCH_SIZE (a chained launch covers up to 128 steps,
so the figures are normalised by the number of steps the profiled command ran: warmup + steps).
The doubling is CALIBRATED for the table rows only (two 16-B loads per lane and row: B x 4 d bytes per step, every row a miss --
the 2 GB table is random-gathered and far beyond the 256 MiB Infinity Cache); everything else the kernel reads -- labels (scattered
4-B loads), indices and sample keys (coalesced 4- / 8-B), the step's normals, tagged-word polls, returning atomics -- is not that
access pattern, so its raw count is reported both ways (x 1 and x 2): `split` holds the two parts, `hbm_bytes_per_step` stays the
UPPER bound (everything doubled), `hbm_bytes_per_step_lower` doubles the rows only.  A calibration of the counter per access width
on a known byte count (tools/probes/fetch_calibration.py under the same --pmc passes) is attached when its file is given.
usage: pmc_traffic.py <fetch_pass_dir> <write_pass_dir> <out.json> <total_steps> [<committed file name> <commit> [<B> <d> [<calibration.json>]]]"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_logreg" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[4])
kernel = max(fetch, key=lambda k: sum(fetch[k]))
fetch_kib, write_kib = sum(fetch[kernel]), sum(write.get(kernel, [0.0]))
n1, n2 = len(fetch[kernel]), len(write.get(kernel, []))
fetch_bytes = 2.0 * fetch_kib * 1024.0
write_bytes = write_kib * 1024.0
out = {"source": sys.argv[5] if len(sys.argv) > 5 else None, "commit": sys.argv[6] if len(sys.argv) > 6 else None,
       "kernel": kernel, "launches": [n1, n2], "steps": steps,
       "FETCH_SIZE_KiB_raw_sum": fetch_kib, "WRITE_SIZE_KiB_raw_sum": write_kib,
       "fetch_bytes_corrected_x2_per_step": fetch_bytes / steps, "write_bytes_per_step": write_bytes / steps,
       "hbm_bytes_per_step": (fetch_bytes + write_bytes) / steps,
       "hbm_bytes_per_launch": (fetch_bytes + write_bytes) / max(n1, 1),
       "note": "FETCH_SIZE doubled per the gfx950 correction for 16-B/lane coalesced reads; separate --pmc passes; "
               "normalised by the DP-VI steps of the profiled command"}
B, d = (int(sys.argv[7]), int(sys.argv[8])) if len(sys.argv) > 8 else (4096, 512)
rows = B * 4 * d                                   # table-row bytes per step (algorithmic: every selected row once)
raw_fetch = fetch_kib * 1024.0 / steps             # the counter as reported, per step
other_raw = raw_fetch - rows / 2.0                 # what is left once the rows (reported at 1/2) are taken out
out["split"] = {
    "table_row_bytes_per_step": rows, "table_rows_reported_as": rows / 2.0,
    "other_fetch_raw_per_step": other_raw, "other_fetch_if_doubled": 2.0 * other_raw,
    "write_bytes_per_step": write_bytes / steps,
    "hbm_bytes_per_step_lower": rows + other_raw + write_bytes / steps,
    "hbm_bytes_per_step_upper": rows + 2.0 * other_raw + write_bytes / steps,
    "note": "rows: B x 4 d bytes, 16 B per lane (the x 2 correction applies); other: labels, indices, keys, normals, parameter / "
            "accumulator words, polls and returning atomics (correction uncalibrated: between x 1 and x 2)"}
out["hbm_bytes_per_step_lower"] = out["split"]["hbm_bytes_per_step_lower"]
if len(sys.argv) > 9:
    try:
        out["counter_calibration"] = json.load(open(sys.argv[9]))
    except OSError:
        pass
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))

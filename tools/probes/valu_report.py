#!/usr/bin/env python3
"""Joins valu_probe's timings with the SQ counters of the same binary: executed VALU instructions per wave and example,
cycles per wave64 VALU instruction per SIMD at 1 / 2 / 4 waves per SIMD.  usage: valu_report.py <out dir>"""
import collections
import csv
import glob
import json
import sys

O = sys.argv[1]
runs = [json.loads(l) for l in open(O + "/valu_probe.jsonl") if l.startswith("{")]
f = glob.glob(O + "/valu_pmc/**/*counter_collection.csv", recursive=True)
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
if f:
    for r in csv.DictReader(open(f[0])):
        pmc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = []
for run in runs:
    key = [k for k in pmc if "<%d>" % run["threads"] in k or "Li%dE" % run["threads"] in k]
    rec = dict(run)
    if key:
        c = {n: sum(v) / len(v) for n, v in pmc[key[0]].items()}          # averages per launch
        waves = c.get("SQ_WAVES", 0.0)
        if waves:
            per_wave_example = c["SQ_INSTS_VALU"] / waves / run["examples_per_wave"]
            rec["pmc"] = {k: round(v, 1) for k, v in c.items()}
            rec["valu_instr_per_example_executed"] = round(per_wave_example, 1)
            simd_instr = run["waves_per_simd"] * run["examples_per_wave"] * per_wave_example
            rec["cycles_per_wave64_valu_instr_executed"] = round(run["us_per_launch"] * run["shader_clock_mhz"] / simd_instr, 3)
            # SQ_ACTIVE_INST_VALU counts quad-cycles (MI355X_MICROARCH.md): x4 = cycles a VALU instruction of some wave was active
            rec["active_valu_cycles_per_instr"] = round(4.0 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"], 3)
    out.append(rec)
print(json.dumps(out, indent=1))

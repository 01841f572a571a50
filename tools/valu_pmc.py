#!/usr/bin/env python3
"""Reduces a rocprofv3 --pmc pass (SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES) over the headline
command of bench.py to the step kernel's wave64 VALU instructions PER DP-VI STEP: the counter per launch / the steps a launch covers
(a chained launch covers up to 128; the profiled command's warmup + steps are whole launches of `steps_per_launch`).
usage: valu_pmc.py <pass_dir> <out.json> <steps_per_launch> [<commit>]"""
import collections
import csv
import glob
import json
import sys

pass_dir, out_path, spl = sys.argv[1], sys.argv[2], int(sys.argv[3])
commit = sys.argv[4] if len(sys.argv) > 4 else None
per_dispatch = collections.defaultdict(float)    # a counter is reported per XCD / SE instance: summed per dispatch
names = {}
for f in glob.glob(pass_dir + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_logreg" not in r["Kernel_Name"]:
            continue
        per_dispatch[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"]
by_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
for (disp, counter), v in per_dispatch.items():
    by_kernel[names[disp]][counter].append(v)
kernel = max(by_kernel, key=lambda k: sum(by_kernel[k].get("SQ_INSTS_VALU", [0.0])))
c = by_kernel[kernel]
# (only launches that cover a whole prepared batch: the largest instruction count is a full launch; shorter ones are dropped)
full = max(c["SQ_INSTS_VALU"])
keep = [i for i, v in enumerate(c["SQ_INSTS_VALU"]) if v > 0.9 * full]


def mean(name):
    vals = c.get(name)
    return sum(vals[i] for i in keep) / len(keep) if vals and len(vals) == len(c["SQ_INSTS_VALU"]) else None


insts, active, wave_cycles, busy, waves = (mean(n) for n in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES"))
out = {"kernel": kernel, "commit": commit, "launches_counted": len(keep), "launches_seen": len(c["SQ_INSTS_VALU"]), "steps_per_launch": spl,
       "SQ_INSTS_VALU_per_launch": insts, "SQ_ACTIVE_INST_VALU_per_launch": active, "SQ_WAVE_CYCLES_per_launch": wave_cycles,
       "SQ_BUSY_CYCLES_per_launch": busy, "SQ_WAVES_per_launch": waves,
       "valu_instructions_per_step": insts / spl,
       "active_valu_cycles_per_step": active / spl if active is not None else None,
       "valu_instructions_per_wave": insts / waves if waves else None,
       "active_valu_cycles_per_instruction": active / insts if active else None,
       "note": "wave64 VALU instructions issued by the step kernel (summed over the XCDs), per DP-VI step; SQ_ACTIVE_INST_VALU = cycles a "
               "SIMD's VALU was executing (bench.py's roofline.valu prices the count against the nominal and the measured issue rates)"}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out))

#!/usr/bin/env python3
"""Per-kernel counter table of the VAE step (tools/vae_pmc.sh: rocprofv3 --pmc passes over tools/time_vae_step.py), with ONE
normalisation for the matrix pipe:

    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDS_PER_COUNTER_UNIT)

both counters taken in the SAME pass and summed over their instances per dispatch, where the unit factor is not assumed but
CALIBRATED: the same pass over tools/probes/mfma_probe (back-to-back v_mfma_f32_32x32x2_f32 on register operands, 155.2 of
157.3 TFLOP/s = 0.987 busy by its own timing, profiles/r02_mfma_probe.jsonl) gives the counter ratio of a kernel whose matrix pipe is
known to be busy; a kernel's fraction = its ratio / the probe's ratio x 0.987.  The clock is the measured one:
GRBM_GUI_ACTIVE / instances / kernel duration.
usage: vae_gemm_pmc.py <out.json> <commit> <probe_pass_dir> <pass_dir> [<pass_dir> ...]"""
import collections
import csv
import glob
import json
import sys

PROBE_BUSY = 155.2 / 157.3


def read(dirs, want):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            per = collections.defaultdict(float)
            names, times = {}, {}
            for r in csv.DictReader(open(f)):
                if not want(r["Kernel_Name"]):
                    continue
                per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = r["Kernel_Name"]
                times[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3   # us
            for (disp, counter), v in per.items():
                acc[names[disp]][counter].append(v)
            for disp, t in times.items():
                dur[names[disp]].append(t)
    out = {}
    for k, counters in acc.items():
        short = k.split("(")[0].replace("void ", "")
        o = {"dispatches": max(len(v) for v in counters.values()), "us_under_pmc": sum(dur[k]) / len(dur[k])}
        for c, v in sorted(counters.items()):
            o[c] = sum(v) / len(v)
        out[short] = o
    return out


out_path, commit, probe_dir, dirs = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
probe = read([probe_dir], lambda n: "k_mfma" in n)
kern = read(dirs, lambda n: "d3p::" in n)
cal = None
for k, o in probe.items():
    if o.get("SQ_VALU_MFMA_BUSY_CYCLES") and o.get("GRBM_GUI_ACTIVE"):
        r = o["SQ_VALU_MFMA_BUSY_CYCLES"] / o["GRBM_GUI_ACTIVE"]
        if cal is None or r > cal[1]:
            cal = (k, r, o)
table = {}
for k, o in kern.items():
    if o.get("GRBM_GUI_ACTIVE"):
        if o.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None and cal:
            o["mfma_busy_frac"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / o["GRBM_GUI_ACTIVE"] / cal[1] * PROBE_BUSY
        if o.get("SQ_BUSY_CYCLES"):
            o["sq_busy_over_gui_active"] = o["SQ_BUSY_CYCLES"] / o["GRBM_GUI_ACTIVE"]
    if o.get("SQ_INSTS_LDS") and o.get("SQ_LDS_BANK_CONFLICT") is not None and o.get("SQ_ACTIVE_INST_LDS"):
        o["lds_bank_conflict_cycles_over_lds_active"] = o["SQ_LDS_BANK_CONFLICT"] / o["SQ_ACTIVE_INST_LDS"]
    table[k] = o
res = {"commit": commit, "command": "rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/time_vae_step.py (tools/vae_pmc.sh)",
       "normalisation": "mfma_busy_frac = (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE of the kernel) / (the same ratio of the calibration "
                        "kernel) x %.3f -- the calibration kernel is tools/probes/mfma_probe's back-to-back fp32 MFMA loop, whose own timing is "
                        "%.3f of the fp32 MFMA peak; counters summed over their instances per dispatch, means over the dispatches" % (PROBE_BUSY, PROBE_BUSY),
       "calibration": None if cal is None else {"kernel": cal[0], "ratio": cal[1], "counters": cal[2]},
       "kernels": table}
json.dump(res, open(out_path, "w"), indent=1)
for k, o in sorted(table.items(), key=lambda kv: -kv[1].get("us_under_pmc", 0) * kv[1]["dispatches"]):
    print(k[:70], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in o.items() if a in ("dispatches", "us_under_pmc", "mfma_busy_frac", "sq_busy_over_gui_active", "lds_bank_conflict_cycles_over_lds_active")})

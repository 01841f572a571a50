"""One step's launches in order from a rocprofv3 --kernel-trace CSV: start, gap to the previous kernel's end, duration, grid, name.
    python tools/kernel_timeline.py <dir or *_kernel_trace.csv> <marker kernel substring> [steps back from the end = 3]
The step printed runs from one launch of the marker kernel to the next."""
import csv, glob, os, sys

path, marker = sys.argv[1], sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-back], idx[-back + 1]
t0 = prev = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    grid = "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    print("%8.1f gap %5.1f dur %6.1f  grid %14s wg %4s  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, grid, r["Workgroup_Size_X"], r["Kernel_Name"][:80]))
    prev = e

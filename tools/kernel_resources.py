#!/usr/bin/env python3
"""Register / scratch / occupancy report of the kernels of one HIP source (developer tool).

    python tools/kernel_resources.py d3p_amd/csrc/d3p_dpvi.hip [name-filter ...]

Compiles the file for gfx950 with -Rpass-analysis=kernel-resource-usage (works without a GPU) and prints one
line per kernel whose demangled name contains any of the filters."""
import os
import re
import subprocess
import sys
import tempfile


def main():
    src = sys.argv[1]
    filters = sys.argv[2:] or [""]
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o",
                            os.path.join(tmp, "o.o"), "-Rpass-analysis=kernel-resource-usage"],
                           capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        raise SystemExit(r.returncode)
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    names = [b.split("\n")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    for b, d in zip(blocks, dem):
        d = d.replace("d3p::", "").split("(")[0]
        if not any(f in d for f in filters):
            continue
        def get(pat):
            m = re.search(pat, b)
            return m.group(1) if m else "?"
        vals = [get(r"VGPRs: (\d+)"), get(r"AGPRs: (\d+)"), get(r"SGPRs: (\d+)"), get(r"ScratchSize \[bytes/lane\]: (\d+)"),
                get(r"Occupancy \[waves/SIMD\]: (\d+)"), get(r"LDS Size \[bytes/block\]: (\d+)")]
        print("%-72s VGPR %4s AGPR %3s SGPR %4s scratch %5s occ %s LDS %s" % ((d,) + tuple(vals)))


if __name__ == "__main__":
    main()

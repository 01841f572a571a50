#!/usr/bin/env python3
"""Known byte counts for calibrating FETCH_SIZE / WRITE_SIZE per access width (MI355X_MICROARCH.md, HBM: "other access widths are
uncalibrated: calibrate on a known byte count in your own access pattern"): d3p_hbm_copy over 1 GiB with 16, 8 and 4 bytes per lane,
three launches each.  Run it under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE`;
tools/probes/fetch_calibration_report.py turns the two passes into bytes-reported / bytes-moved per width."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import d3p_amd._lib as L  # noqa: E402

L.require_device()
lib = L.load()
n = 1 << 30
src = torch.empty(n, dtype=torch.uint8, device="cuda").fill_(3)
dst = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for width in (16, 8, 4):
    for _ in range(3):
        L.check(lib.d3p_hbm_copy(L.stream_ptr(), L.ptr(dst), L.ptr(src), n, width))
    torch.cuda.synchronize()
assert bool((dst[:: 1 << 20] == 3).all())
print("fetch_calibration: 3 copies of 1 GiB at 16, 8 and 4 bytes per lane")

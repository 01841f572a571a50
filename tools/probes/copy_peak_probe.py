#!/usr/bin/env python3
"""What device-to-device copy rate does THIS box reach?  d3p_hbm_copy (16 B per lane) over 2 GiB with several grids, nontemporal and
plain loads / stores, against the runtime's own copy (torch `dst.copy_(src)`); median of 5 per variant, GB/s = 2 x size / time."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import d3p_amd._lib as L  # noqa: E402

L.require_device()
lib = L.load()
n = 2 << 30
src = torch.empty(n, dtype=torch.uint8, device="cuda").fill_(1)
dst = torch.empty(n, dtype=torch.uint8, device="cuda")


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    r = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        r.append(2.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    return round(sorted(r)[2], 1)


out = {"torch_copy_": timed(lambda: dst.copy_(src))}
for temporal in (False, True):
    if temporal:
        os.environ["D3P_COPY_TEMPORAL"] = "1"
    for grid in (1024, 2048, 4096, 8192, 16384, 65536):
        os.environ["D3P_COPY_GRID"] = str(grid)
        out[f"k_hbm_copy grid {grid} {'plain' if temporal else 'nontemporal'}"] = timed(
            lambda: L.check(lib.d3p_hbm_copy(L.stream_ptr(), L.ptr(dst), L.ptr(src), n, 16)))
print(json.dumps(out, indent=1))

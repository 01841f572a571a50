"""Times d3p_poisson_select_batch (the Poisson mask of `steps` consecutive batches over an N-row table: k_poisson_flags,
k_poisson_scan, k_poisson_write) by itself, with HIP events.  Under rocprofv3 (--kernel-trace --stats, or --pmc) the same
command gives the per-kernel split and the VALU counters:  python3 tools/time_poisson_select.py [N] [steps] [reps]"""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
import d3p_amd._lib as L
from d3p_amd._lib import check, ptr, stream_ptr

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
B = 4096
q = B / N
lib = L.load()
L.require_device()
dev = torch.device("cuda:0")
keys = torch.randint(0, 2**31 - 1, (steps, 16), dtype=torch.int32, device=dev)
per = lib.d3p_poisson_select_workspace(N)
ws = torch.empty(per * steps, dtype=torch.uint8, device=dev)
cutoff = int(B * 1.1)
idx = torch.empty((steps, cutoff), dtype=torch.int32, device=dev)
counts = torch.zeros((steps, 2), dtype=torch.int32, device=dev)


def run():
    check(lib.d3p_poisson_select_batch(stream_ptr(), 0, ptr(keys), 16, C.c_float(q), N, cutoff, 0, ptr(idx), cutoff, ptr(counts), 2,
                                       steps, ptr(ws), ws.numel()))


run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (reps * steps)
print(f"poisson select, N = {N}, {steps} steps per call: {us:.2f} us per step ({N / 16 / us * 1e-3:.1f} G ChaCha blocks/s); "
      f"mean selected {counts[:, 0].float().mean().item():.1f}")

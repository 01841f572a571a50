"""Which torch streams run CONCURRENTLY with the current (default) stream on this box?

A HIP process has a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default); streams beyond that share one, and two streams that
share a hardware queue are serialised.  The tests that play a second rank on a side stream need a stream that does NOT share
the default stream's queue.  For each of the next N streams of torch's pool: a 50 ms spin kernel on the stream, then a tiny
kernel on the default stream; concurrent = the tiny kernel finished long before the spin kernel did.
"""
import sys
import time

import torch


def concurrent_with_default(s, spin_cycles=100_000_000):
    torch.cuda.synchronize()
    x = torch.zeros(8, device="cuda")
    e_side, e_main = torch.cuda.Event(enable_timing=False), torch.cuda.Event(enable_timing=False)
    with torch.cuda.stream(s):
        torch.cuda._sleep(spin_cycles)
        e_side.record()
    x.add_(1.0)
    e_main.record()
    t0 = time.perf_counter()
    e_main.synchronize()
    t_main = time.perf_counter() - t0
    e_side.synchronize()
    t_side = time.perf_counter() - t0
    return t_main < 0.5 * t_side, t_main, t_side


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    out = []
    for i in range(n):
        s = torch.cuda.Stream()
        ok, tm, ts = concurrent_with_default(s)
        out.append(ok)
        print(f"stream {i:2d} id {s.stream_id} cuda_stream {s.cuda_stream:#x}: concurrent with the default stream: {ok}  (main {tm * 1e3:.1f} ms, side {ts * 1e3:.1f} ms)")
    print("pattern:", "".join("C" if o else "s" for o in out))

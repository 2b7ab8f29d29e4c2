#!/usr/bin/env python3
"""Is the spread of tools/debug/placement.py a property of the memory?  Times a plain fill (torch, one store stream) of N
separately allocated 7.3 GB buffers, twice each."""
import sys

import torch

n_buf = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = 914358272
bufs = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(n_buf)]
for rep in range(2):
    out = []
    for b in bufs:
        b.fill_(1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            b.fill_(2.0)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out.append("%.3f ms (%.2f TB/s) @ 0x%x" % (ms, 8 * n / ms * 1e-9, b.data_ptr()))
    print(" | ".join(out), flush=True)

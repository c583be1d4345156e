#!/usr/bin/env python3
"""Achievable HBM bandwidth on this GPU for traffic volumes like the fine-level multigrid kernels (torch copy / triad)."""
import torch
dev = 'cuda:0'
for mb in (59, 118, 236, 944):
    n = mb * 1000 * 1000 // 8
    a = torch.empty(n, dtype=torch.float64, device=dev)
    b = torch.ones(n, dtype=torch.float64, device=dev)
    c = torch.ones(n, dtype=torch.float64, device=dev)
    for name, fn, vol in (('copy', lambda: a.copy_(b), 2), ('triad', lambda: torch.add(b, c, alpha=2.0, out=a), 3)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print('%-5s %4d MB per array: %7.1f us  %6.0f GB/s' % (name, mb, us, vol * n * 8 / us / 1e3))

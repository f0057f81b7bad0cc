#!/usr/bin/env python3
"""Probe (GPU box): cost of pinned host allocation and D2H rates, to size the host-result path."""
import time

import torch

n = 8192 * 8192 * 6
dev = torch.empty(n, dtype=torch.float64, device="cuda:0").fill_(1.0)
torch.cuda.synchronize()
for label, mk in (("pageable", lambda: torch.empty(n, dtype=torch.float64)), ("pinned", lambda: torch.empty(n, dtype=torch.float64, pin_memory=True))):
    t0 = time.perf_counter()
    host = mk()
    t_alloc = time.perf_counter() - t0
    for rep in range(2):
        t0 = time.perf_counter()
        host.copy_(dev)
        torch.cuda.synchronize()
        t_copy = time.perf_counter() - t0
        print(f"{label}: alloc {t_alloc * 1e3:7.1f} ms, D2H pass {rep}: {t_copy * 1e3:7.1f} ms = {n * 8 / t_copy / 1e9:5.1f} GB/s", flush=True)
    del host

#!/usr/bin/env python3
"""One fused residual kernel, a few launches - the program the counter passes of tools/exp/pmc_kernel.sh run:
    python3 tools/exp/eval_job.py <eq> <layout ny|nt> <BxTxXxY> [reps]     eq: ns | continuity | momentum | energy | induction | gauss"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cp_pre_amd import residuals as R            # noqa: E402

eq, layout, shp = sys.argv[1], sys.argv[2], sys.argv[3]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
B, T, X, Y = (int(v) for v in shp.split("x"))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
F = 3 if eq == "ns" else 6
w = torch.rand(B, F, X, Y, T, device=dev, generator=g).add_(0.5).permute(0, 1, 4, 2, 3) if layout == "nt" else \
    torch.rand(B, F, T, X, Y, device=dev, generator=g).add_(0.5)
fn = (lambda: R.NavierStokes(1e-2, 1.0 / X, 1.0 / Y).residual_momentum(w, boundary=True)) if eq == "ns" else \
     (lambda: getattr(R.MHD(), "residual_" + eq)(w, boundary=True))
bpc = {"ns": 16, "continuity": 16, "momentum": 28, "energy": 28, "induction": 20, "gauss": 12}[eq]
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    torch.cuda.synchronize()
    del r
print(f"{eq} {layout} [{B},{T},{X},{Y}]: {e0.elapsed_time(e1):.3f} ms  {bpc * B * T * X * Y / e0.elapsed_time(e1) / 1e6:.0f} GB/s of {bpc * B * T * X * Y / 1e9:.2f} GB algorithmic")

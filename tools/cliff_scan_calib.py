#!/usr/bin/env python3
"""Scan of the calibration functions over realistic calibration-set sizes and layouts (MI355X): per-cell
calibrate (10 alpha levels in one call and one at a time, as the scripts do), modulation_func, ncf_metric_joint,
emp_cov, on device tensors in the reference layout and in the surrogate's Nt-fastest memory order.
Prints effective GB/s of ONE read of the scores per function call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cp_pre_amd import inductive_cp as icp
dev = torch.device("cuda:0")


def timeit(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


alphas = [float(a) for a in icp.ALPHA_LEVELS]
for n in (100, 500, 1000):
    for (T, N) in ((20, 64), (30, 128), (10, 256), (60, 256)):
        for layout in ("reference", "surrogate"):
            if layout == "reference":
                r = torch.randn(n, T, N, N, device=dev)
            else:
                r = torch.randn(n, N, N, T, device=dev).permute(0, 3, 1, 2)
            s = r.abs()
            nbytes = 4 * r.numel()
            a = nbytes / timeit(lambda: icp.calibrate_multi(s, n, alphas)) / 1e6
            b = nbytes / timeit(lambda: icp.calibrate(s, n, 0.1)) / 1e6
            mod = icp.modulation_func(r, None)
            c = nbytes / timeit(lambda: icp.modulation_func(r, None)) / 1e6
            d = nbytes / timeit(lambda: icp.ncf_metric_joint(r, None, mod)) / 1e6
            d2 = nbytes / timeit(lambda: icp.ncf_metric_joint(r, None, mod, crop=1)) / 1e6
            q = icp.calibrate(s, n, 0.1)
            e = nbytes / timeit(lambda: icp.emp_cov([-q, q], r)) / 1e6
            print(f"{layout:9s} n={n:4d} [{T},{N},{N}]: calibrate x10 {a:6.0f}  calibrate x1 {b:6.0f}  modulation {c:6.0f}  "
                  f"joint score {d:6.0f} (crop=1: {d2:6.0f})  emp_cov {e:6.0f}   (GB/s of one read of the scores)", flush=True)
            del r, s, mod, q

#!/usr/bin/env python3
"""Scan realistic shapes / layouts for performance cliffs (MI355X): every fused residual and a single operator
on the reference layout [BS,F,Nt,Nx,Ny], the surrogate's native layout [BS,F,Nx,Ny,Nt] (seen through
permute(0,1,4,2,3), Marginal/NS_Residuals_CP.py:282) and their 1-D counterparts, for the T_out / grid sizes the
reference scripts use.  Prints algorithmic GB/s and flags anything under 2 TB/s.
    python tools/cliff_scan.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cp_pre_amd import residuals as R
from cp_pre_amd.convops_2d import ConvOperator
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


low = []
for N in (64, 100, 128, 200, 256):
    for T in (10, 20, 30, 40, 60):
        B = max(4, (1 << 26) // (T * N * N))
        cells = B * T * N * N
        for layout in ("reference", "surrogate"):
            if layout == "reference":
                v = torch.empty(B, 6, T, N, N, device=dev).uniform_(0.5, 1.5)
            else:
                v = torch.empty(B, 6, N, N, T, device=dev).uniform_(0.5, 1.5).permute(0, 1, 4, 2, 3)
            ns, wave, mhd, L = R.NavierStokes(0.01, 1 / N, 1 / N), R.PRE_Wave(0.01, 0.02), R.MHD(), ConvOperator(("x", "y"), 2)
            ops = {"wave": (8, lambda: wave.residual(v[:, 0], True)), "laplacian": (8, lambda: L(v[:, 1])),
                   "ns_mom": (16, lambda: ns.residual_momentum(v[:, :3], True)), "ns_cont": (12, lambda: ns.residual_continuity(v[:, :2], True)),
                   "mhd_ind": (20, lambda: mhd.residual_induction(v, True)), "mhd_energy": (28, lambda: mhd.residual_energy(v, True)),
                   "mhd_mom": (28, lambda: mhd.residual_momentum(v, True)), "mhd_cont": (16, lambda: mhd.residual_continuity(v, True))}
            row = []
            for name, (bpc, fn) in ops.items():
                gbs = bpc * cells / timeit(fn) / 1e6
                row.append(f"{name} {gbs:5.0f}")
                if gbs < 2000:
                    low.append((layout, B, T, N, name, gbs))
            print(f"{layout:9s} [{B},{T},{N},{N}]: " + "  ".join(row), flush=True)
            del v
for X in (100, 200, 512, 1024):
    for T in (20, 30, 100, 200):
        B = max(16, (1 << 25) // (T * X))
        for layout in ("reference", "surrogate"):
            u = (torch.empty(B, T, X, device=dev) if layout == "reference" else torch.empty(B, X, T, device=dev).permute(0, 2, 1)).uniform_(0.5, 1.5)
            bur, adv = R.Burgers(2 / X, 1.25 / T, 0.002), R.Advection(1.0, 0.005, 0.01)
            a = 8 * u.numel() / timeit(lambda: bur.residual(u, True)) / 1e6
            b = 8 * u.numel() / timeit(lambda: adv.residual(u, True)) / 1e6
            print(f"{layout:9s} 1-D [{B},{T},{X}]: burgers {a:5.0f}  advection {b:5.0f}", flush=True)
            for name, gbs in (("burgers", a), ("advection", b)):
                if gbs < 2000:
                    low.append((layout, B, T, X, name, gbs))
print("\nunder 2 TB/s:")
for item in low:
    print("  ", item)

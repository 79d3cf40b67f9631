#!/usr/bin/env python3
"""Interleaved A/B of several builds of libcp_pre_hip.so in ONE process (boxes differ by several per cent):
    python tools/exp/var_ab.py name=path.so [name=path.so ...] [--reps 5] [--eqs momentum,energy]
The first library is the reference: every other one must reproduce its residuals bit for bit."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cp_pre_amd import _lib                      # noqa: E402
from cp_pre_amd import residuals as R            # noqa: E402


def handle(path):
    _lib._lib = None
    _lib.SO_PATH = path
    return _lib.load()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--eqs", default="momentum,energy,continuity,induction")
    ap.add_argument("--shapes", default="1024x64x256x256,512x10x512x512")
    ap.add_argument("--tol", type=float, default=0.0, help="allowed tensor-scale relative difference from the first library "
                    "(0: bit for bit; different code shapes contract their multiply-adds differently)")
    ap.add_argument("--layout", choices=["ny", "nt"], default="ny",
                    help="nt: the fields as [B,6,Nx,Ny,Nt].permute(0,1,4,2,3) - the surrogate's Nt-fastest layout")
    args = ap.parse_args()
    names, libs = [], {}
    for spec in args.libs:
        n, p = spec.split("=", 1)
        names.append(n)
        libs[n] = handle(os.path.abspath(p))
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    bpc = {"momentum": 28, "energy": 28, "continuity": 16, "induction": 20, "ns": 16, "wave": 8}
    for shp in args.shapes.split(","):
        B, T, X, Y = (int(v) for v in shp.split("x"))
        if args.layout == "nt":
            w = torch.rand(B, 6, X, Y, T, device=dev, generator=g).add_(0.5).permute(0, 1, 4, 2, 3)
        else:
            w = torch.rand(B, 6, T, X, Y, device=dev, generator=g).add_(0.5)
        mhd, ns = R.MHD(), R.NavierStokes(1e-2, 1.0 / X, 1.0 / Y)
        cells = B * T * X * Y
        for eq in args.eqs.split(","):
            wave = R.PRE_Wave(dt=0.005, dx=0.01, c=1.0)
            fn = (lambda: ns.residual_momentum(w[:, :3], boundary=True)) if eq == "ns" else \
                 (lambda: wave.residual(w[:, 0], boundary=True)) if eq == "wave" else \
                 (lambda eq=eq: getattr(mhd, "residual_" + eq)(w, boundary=True))
            times = {n: [] for n in names}
            ref = None
            for rep in range(args.reps + 1):
                for n in names:
                    _lib._lib = libs[n]
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    r = fn()
                    e1.record()
                    torch.cuda.synchronize()
                    if rep == 0:
                        if ref is None:
                            ref = r
                        elif not torch.equal(r, ref):
                            d = float((r - ref).abs().max() / ref.abs().max())
                            print(f"  ({n} differs from {names[0]} on {eq}: {d:.2e} tensor-scale)", flush=True)
                            assert d <= args.tol, (n, eq, d)
                    else:
                        times[n].append(e0.elapsed_time(e1))
                    del r
            del ref
            base = None
            line = f"[{B},{T},{X},{Y}]{' Nt-fastest' if args.layout == 'nt' else ''} {eq:11s}"
            for n in names:
                t = sorted(times[n])[len(times[n]) // 2]
                base = base or t
                line += f"  {n} {t:7.3f} ms {bpc[eq] * cells / t / 1e6:5.0f} GB/s ({t / base:.3f})"
            print(line, flush=True)
        del w
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

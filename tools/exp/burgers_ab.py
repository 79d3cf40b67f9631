#!/usr/bin/env python3
"""Interleaved A/B of the 1-D residual kernels (Burgers, advection: [B,T,X] mapped to [1,B,T,X], the batch axis marched):
    python tools/exp/burgers_ab.py name=path.so [...]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cp_pre_amd import _lib                      # noqa: E402
from cp_pre_amd import residuals as R            # noqa: E402


def handle(path):
    _lib._lib = None
    _lib.SO_PATH = path
    return _lib.load()


names, libs = [], {}
for spec in sys.argv[1:]:
    n, p = spec.split("=", 1)
    names.append(n)
    libs[n] = handle(os.path.abspath(p))
dev = torch.device("cuda:0")
for (B, T, X) in ((8192, 200, 512), (65536, 200, 512), (256, 100, 200), (20000, 100, 256)):
    u = torch.rand(B, T, X, device=dev) + 0.5
    for name, op in (("burgers", R.Burgers(2.0 / X, 1.25 / T, 0.002)), ("advection", R.Advection(1.0, 0.005, 0.01, disc=2))):
        times, ref = {k: [] for k in names}, None
        for rep in range(10):
            for k in names:
                _lib._lib = libs[k]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = op.residual(u, boundary=True)
                e1.record()
                torch.cuda.synchronize()
                if rep == 0:
                    ref = r if ref is None else ref
                    assert torch.equal(r, ref), (k, name)
                else:
                    times[k].append(e0.elapsed_time(e1))
                if r is not ref:
                    del r
        line, base = f"[{B},{T},{X}] {name:9s}", None
        for k in names:
            t = sorted(times[k])[len(times[k]) // 2]
            base = base or t
            line += f"  {k} {t:7.3f} ms {8 * B * T * X / t / 1e6:5.0f} GB/s ({t / base:.3f})"
        print(line, flush=True)
        del ref
    del u

#!/usr/bin/env python3
"""Interleaved A/B of the moments + segment-maxima pass (pre_moments_segmax_f64) across builds: python tools/exp/moments_ab.py name=path.so ..."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cp_pre_amd import _lib                      # noqa: E402
from cp_pre_amd import pipeline                  # noqa: E402


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
for (n, T, X, Y) in ((4096, 64, 128, 512), (1024, 64, 256, 256), (512, 32, 256, 256), (4096, 16, 128, 512)):
    res = torch.randn(n, T, X, Y, device=dev)
    crop = (1, 1, 1)
    times, ref = {k: [] for k in names}, None
    for rep in range(8):
        for k in names:
            _lib._lib = libs[k]
            mom = pipeline.HipOps.zeros_moments((T - 2) * X * Y, dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            seg = pipeline.HipOps.add_moments_segmax(res, mom, crop)
            e1.record()
            torch.cuda.synchronize()
            if rep == 0:
                if ref is None:
                    ref = (mom.clone(), seg.clone())
                else:
                    assert torch.allclose(mom, ref[0], rtol=1e-12, atol=0.0) and torch.equal(seg, ref[1]), k
            else:
                times[k].append(e0.elapsed_time(e1))
            del mom, seg
    line, base = f"[{n},{T},{X},{Y}]", None
    for k in names:
        t = sorted(times[k])[len(times[k]) // 2]
        base = base or t
        line += f"  {k} {t:7.3f} ms {4 * n * (T - 2) * X * Y / t / 1e6:5.0f} GB/s ({t / base:.3f})"
    print(line, flush=True)
    del res, ref

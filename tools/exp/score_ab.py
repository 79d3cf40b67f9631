#!/usr/bin/env python3
"""Interleaved A/B of the full joint score pass (pre_joint_score_f32) across builds of libcp_pre_hip.so in ONE process:
    python tools/exp/score_ab.py name=path.so [...] [--shapes 1024x64x256x256,...]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cp_pre_amd import _lib                      # noqa: E402
from cp_pre_amd import pipeline                  # noqa: E402


def handle(path):
    _lib._lib = None
    _lib.SO_PATH = path
    return _lib.load()


ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--reps", type=int, default=9)
ap.add_argument("--shapes", default="1024x64x256x256,1024x64x128x512,8192x1x200x512,512x32x256x256,256x20x512x512")
args = ap.parse_args()
names, libs = [], {}
for spec in args.libs:
    n, p = spec.split("=", 1)
    names.append(n)
    libs[n] = handle(os.path.abspath(p))
dev = torch.device("cuda:0")
for shp in args.shapes.split(","):
    n, T, X, Y = (int(v) for v in shp.split("x"))
    res = torch.randn(n, T, X, Y, device=dev)
    mod = torch.rand(T, X, Y, device=dev) + 0.5
    crop = (1, 1, 1) if T > 2 else (0, 1, 1)
    times, ref = {k: [] for k in names}, None
    for rep in range(args.reps + 1):
        for k in names:
            _lib._lib = libs[k]
            sc = pipeline.HipOps.zeros_scores(n, dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            pipeline.HipOps.max_scores(res, mod, crop, sc)
            e1.record()
            torch.cuda.synchronize()
            if rep == 0:
                ref = sc if ref is None else ref
                assert torch.equal(sc, ref), k
            else:
                times[k].append(e0.elapsed_time(e1))
    line, base = f"[{n},{T},{X},{Y}]", None
    for k in names:
        t = sorted(times[k])[len(times[k]) // 2]
        base = base or t
        line += f"  {k} {t:7.3f} ms {4 * res.numel() / t / 1e6:5.0f} GB/s ({t / base:.3f})"
    print(line, flush=True)
    del res, mod

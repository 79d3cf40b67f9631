#!/usr/bin/env python3
"""Interleaved A/B of the per-cell select on a POWER-OF-TWO row pitch (a caller's dense [n, 2^k] device tensor) and on the
padded pitch, across builds: python tools/exp/select_pitch_ab.py name=path.so [...] [--ns ...]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cp_pre_amd import _lib                      # noqa: E402
from cp_pre_amd import inductive_cp as icp       # noqa: E402


def handle(path):
    _lib._lib = None
    _lib.SO_PATH = path
    return _lib.load()


ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--ns", default="256,512,768,1024,1200,1500,2048,3000,4096,8192")
ap.add_argument("--reps", type=int, default=7)
args = ap.parse_args()
names, libs = [], {}
for spec in args.libs:
    n, p = spec.split("=", 1)
    names.append(n)
    libs[n] = handle(os.path.abspath(p))
dev = torch.device("cuda:0")
alphas = [float(a) for a in icp.ALPHA_LEVELS]
for n in (int(x) for x in args.ns.split(",")):
    M = 1 << max(16, 32 - (4 * n - 1).bit_length())
    for pitch in (M, M + 64):
        torch.manual_seed(n)
        buf = torch.randn(n * pitch, device=dev).abs_()
        s = buf.as_strided((n, M), (pitch, 1))
        s[:, 5] = 1.0
        ks = [icp.kth_index(n, n, a) for a in alphas]
        want = torch.sort(s[:, :4096], dim=0).values[ks]
        want_end = torch.sort(s[:, -4096:], dim=0).values[ks]
        times = {k: [] for k in names}
        for rep in range(args.reps + 1):
            for k in names:
                _lib._lib = libs[k]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                q = icp.kth_axis0(s, ks)
                e1.record()
                torch.cuda.synchronize()
                if rep == 0:
                    assert torch.equal(q[:, :4096], want) and torch.equal(q[:, -4096:], want_end), (k, n, pitch)
                else:
                    times[k].append(e0.elapsed_time(e1))
                del q
        line, base = f"n={n:5d} M={M:8d} pitch M{'+64' if pitch > M else '   '}", None
        for k in names:
            t = sorted(times[k])[len(times[k]) // 2]
            base = base or t
            line += f"  {k} {t:6.3f} ms {4 * n * M / t / 1e6:5.0f} GB/s ({t / base:.3f})"
        print(line, flush=True)
        del buf, s, want, want_end

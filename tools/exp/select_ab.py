#!/usr/bin/env python3
"""Interleaved A/B of the per-cell select across several builds of libcp_pre_hip.so in ONE process:
    python tools/exp/select_ab.py name=path.so [name=path.so ...] [--ns 512,1000,2048] [--reps 7]
|N(0,1)| scores [n, M] with rows M + 64 floats apart (pipeline.row_padded), 10 ranks; every library's result is checked
against torch.sort on the first 4096 cells."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cp_pre_amd import _lib                      # noqa: E402
from cp_pre_amd import inductive_cp as icp       # noqa: E402


def handle(path):
    _lib._lib = None
    _lib.SO_PATH = path
    return _lib.load()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--ns", default="256,384,512,600,768,1000,1024,1500,2048")
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
        pitch = M + 64
        torch.manual_seed(n)
        buf = torch.randn(n * pitch, device=dev).abs_()
        s = buf.as_strided((n, M), (pitch, 1))
        s[:, 5] = 1.0
        ks = [icp.kth_index(n, n, a) for a in alphas]
        want = torch.sort(s[:, :4096], dim=0).values[ks]
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
                    assert torch.equal(q[:, :4096], want), (k, n)
                else:
                    times[k].append(e0.elapsed_time(e1))
                del q
        line, base = f"n={n:5d} M={M:8d}", None
        for k in names:
            t = sorted(times[k])[len(times[k]) // 2]
            base = base or t
            line += f"  {k} {t:6.3f} ms {4 * n * M / t / 1e6:5.0f} GB/s ({t / base:.3f})"
        print(line, flush=True)
        del buf, s, want
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

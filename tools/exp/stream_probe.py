#!/usr/bin/env python3
"""n = 4096 streaming select under two libraries, by number of ranks: python tools/exp/stream_probe.py a=lib.so b=lib.so"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cp_pre_amd import _lib, inductive_cp as icp

def handle(path):
    _lib._lib = None
    _lib.SO_PATH = path
    return _lib.load()

libs = {}
for spec in sys.argv[1:]:
    k, p = spec.split("=", 1)
    libs[k] = handle(os.path.abspath(p))
dev = torch.device("cuda:0")
for n in (3000, 3800, 4096):
    M = 262144 + 64
    s = torch.randn(n, M, device=dev).abs_()
    for nk in (1, 3, 10):
        alphas = [0.5] if nk == 1 else [0.1, 0.5, 0.9] if nk == 3 else [float(a) for a in icp.ALPHA_LEVELS]
        ks = [icp.kth_index(n, n, a) for a in alphas]
        line = f"n={n} nk={nk}"
        for k, lib in libs.items():
            _lib._lib = lib
            ts = []
            for rep in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); q = icp.kth_axis0(s, ks); e1.record(); torch.cuda.synchronize()
                if rep: ts.append(e0.elapsed_time(e1))
            t = sorted(ts)[len(ts) // 2]
            line += f"  {k} {t:.3f} ms {4*n*M/t/1e6:.0f} GB/s"
        print(line, flush=True)
    del s

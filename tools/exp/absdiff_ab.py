#!/usr/bin/env python3
"""Interleaved A/B of pre_absdiff_f32 (|a - b|, three distinct buffers) across builds: python tools/exp/absdiff_ab.py name=path.so ..."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cp_pre_amd import _lib                      # noqa: E402


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
for n in (1 << 31, 838860800, 1 << 24, 1000003):
    a = torch.randn(n, device=dev)
    b = torch.randn(n, device=dev)
    want = (a - b).abs()
    times = {k: [] for k in names}
    for rep in range(10):
        for k in names:
            out = torch.empty_like(a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(libs[k].pre_absdiff_f32(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), n, _lib.stream()), "absdiff")
            e1.record()
            torch.cuda.synchronize()
            if rep == 0:
                assert torch.equal(out, want), k
                o1 = torch.empty_like(a)
                _lib.check(libs[k].pre_absdiff_f32(_lib.ptr(a), None, _lib.ptr(o1), n, _lib.stream()), "abs")
                assert torch.equal(o1, a.abs()), k
                del o1
            else:
                times[k].append(e0.elapsed_time(e1))
            del out
    line, base = f"n={n:11d}", None
    for k in names:
        t = sorted(times[k])[len(times[k]) // 2]
        base = base or t
        line += f"  {k} {t:7.3f} ms {12 * n / t / 1e6:5.0f} GB/s ({t / base:.3f})"
    print(line, flush=True)
    del a, b, want

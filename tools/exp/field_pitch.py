#!/usr/bin/env python3
"""Does a power-of-two distance between the fields of `vars` [BS,F,Nt,Nx,Ny] cost the fused kernels?  (C4: 64 x 256 x 256
cells per field = 16 MiB: the four field streams of the induction residual then share their low address bits.)
Times MHD induction / continuity on a contiguous tensor and on one whose fields are `pad` floats further apart."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cp_pre_amd import residuals as R            # noqa: E402

dev = torch.device("cuda:0")
B, F, T, X, Y = 1024, 6, 64, 256, 256
per = T * X * Y
mhd = R.MHD()
g = torch.Generator(device=dev).manual_seed(0)


def make(pad):
    buf = torch.empty(B * F * (per + pad), device=dev)
    buf.uniform_(0.5, 1.5, generator=g)
    return buf.as_strided((B, F, T, X, Y), (F * (per + pad), per + pad, X * Y, Y, 1))


pads = [int(a) for a in sys.argv[1:]] or [0, 1024]
ws = {pad: make(pad) for pad in pads}                    # all resident at once: the variants alternate launch by launch
for name, fn, bpc in (("induction", mhd.residual_induction, 20), ("continuity", mhd.residual_continuity, 16),
                      ("momentum", mhd.residual_momentum, 28)):
    ts = {pad: [] for pad in pads}
    for rep in range(9):
        for pad in pads:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(ws[pad], boundary=True)
            e1.record()
            torch.cuda.synchronize()
            del r
            if rep:
                ts[pad].append(e0.elapsed_time(e1))
    print(f"{name:12s}", "   ".join(f"pad {pad}: {sorted(v)[len(v) // 2]:.3f} ms ({bpc * B * per / sorted(v)[len(v) // 2] / 1e6:.0f} GB/s)"
                                     for pad, v in ts.items()), flush=True)

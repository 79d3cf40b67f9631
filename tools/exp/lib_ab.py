#!/usr/bin/env python3
"""A/B of two builds of libcp_pre_hip.so inside ONE process (the boxes of the pool differ by several per cent, and the
first process after an idle period runs faster: only an interleaved comparison in one job says anything).

    python tools/exp/lib_ab.py [--old tools/exp/prev/libcp_pre_hip.so] [--reps 6]

Times the fused residual kernels of C3 (an x-slab, quarter batch), C4 (MHD induction) and the JOREK temperature
residual with HIP events, alternating the two libraries launch by launch."""
import argparse
import ctypes
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
    ap.add_argument("--old", default=os.path.join(ROOT, "tools/exp/prev/libcp_pre_hip.so"))
    ap.add_argument("--reps", type=int, default=6)
    args = ap.parse_args()
    new_path = _lib.SO_PATH
    libs = {"old": handle(args.old), "new": handle(new_path)}
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)

    def measure(name, nbytes, fn):
        times = {"old": [], "new": []}
        for rep in range(args.reps + 1):
            for tag in ("old", "new"):
                _lib._lib = libs[tag]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = fn()
                e1.record()
                torch.cuda.synchronize()
                del r
                if rep:                                   # the first round warms both up
                    times[tag].append(e0.elapsed_time(e1))
        o, n = (sorted(times[t])[len(times[t]) // 2] for t in ("old", "new"))
        print(f"{name:48s} old {o:8.3f} ms ({nbytes / o / 1e6:6.0f} GB/s)   new {n:8.3f} ms ({nbytes / n / 1e6:6.0f} GB/s)   "
              f"new/old {n / o:.3f}", flush=True)

    B, T, X, Y = 1024, 64, 128, 512
    v = torch.rand(B, 3, T, X + 2, Y, device=dev, generator=g).add_(0.5)
    ns = R.NavierStokes(1e-2, 1 / 512, 1 / 512, nu=1e-3)
    out = torch.empty(B, T, X, Y, device=dev)
    measure("C3 NS momentum x-slab [1024,64,128(+2),512]", 16 * B * T * X * Y,
            lambda: ns.residual_momentum(v[:, :, :, 1:-1], boundary=True, out=out, halo_x=True))
    del v, out
    torch.cuda.empty_cache()
    B4 = 1024
    w = torch.rand(B4, 6, 64, 256, 256, device=dev, generator=g).add_(0.5)
    mhd = R.MHD()
    cells = B4 * 64 * 256 * 256
    measure("C4 MHD induction [1024,64,256,256]", 20 * cells, lambda: mhd.residual_induction(w, boundary=True))
    measure("MHD momentum [1024,64,256,256]", 28 * cells, lambda: mhd.residual_momentum(w, boundary=True))
    measure("MHD energy [1024,64,256,256]", 28 * cells, lambda: mhd.residual_energy(w, boundary=True))
    measure("MHD continuity [1024,64,256,256]", 16 * cells, lambda: mhd.residual_continuity(w, boundary=True))
    del w
    torch.cuda.empty_cache()
    # the single-field kernels of C2 / C5 / C1
    u2 = torch.randn(512, 32, 256, 256, device=dev, generator=g)
    wave = R.PRE_Wave(dt=0.005, dx=0.01, c=1.0, device=dev)
    measure("C2 wave additive kernel [512,32,256,256]", 8 * u2.numel(), lambda: wave.residual(u2, boundary=True))
    u5 = torch.randn(8192, 200, 512, device=dev, generator=g)
    bur = R.Burgers(2.0 / 512, 1.25 / 200, 0.002)
    measure("C5 Burgers [8192,200,512]", 8 * u5.numel(), lambda: bur.residual(u5, boundary=True))
    adv = R.Advection(1.0, 0.005, 0.01, disc=2)
    measure("advection additive kernel [8192,200,512]", 8 * u5.numel(), lambda: adv.residual(u5, boundary=True))
    del u2, u5
    w = torch.rand(B4, 6, 64, 256, 256, device=dev, generator=g).add_(0.5)
    # JOREK: [BS,F,Nx,Ny,Nt] in the script; here the fields Ny-contiguous ([BS,F,Nt,Nx,Ny] permuted to the script's axes)
    jv = w[:, :3].permute(0, 1, 3, 4, 2)                          # -> unstack_fields gives [BS,Nt,Nx,Ny] views, Ny fastest
    jk = R.JOREK(torch.linspace(1.0, 2.0, 256), device=dev)
    measure("JOREK temperature [1024,64,256,256]", 16 * cells, lambda: jk.residual_temperature(jv, boundary=True))
    measure("JOREK continuity [1024,64,256,256]", 12 * cells, lambda: jk.residual_continuity(jv, boundary=True))
    for tag in ("old", "new"):
        _lib._lib = libs[tag]
        libs[tag + "_res"] = mhd.residual_induction(w[:64], boundary=True)
    print("induction results identical:", torch.equal(libs["old_res"], libs["new_res"]))


if __name__ == "__main__":
    main()

"""A/B of the march_kernel tile shape (PRE_TUNE_TILE, read once per process) on the MHD / NS functors.
    for t in default 4x64 8x32 16x16; do PRE_TUNE_TILE=$t python tools/exp/tile_ab.py; done"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cp_pre_amd import _lib, residuals as R
if os.environ.get("PROBE_SO"):                       # an experimental build of the library (tools/exp/*.so)
    _lib.SO_PATH = os.path.abspath(os.environ["PROBE_SO"])

dev = torch.device("cuda:0")


def timeit(fn, reps=8, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    tag = os.environ.get("PRE_TUNE_TILE", "default")
    for (B, T, X, Y) in [(1024, 64, 256, 256), (512, 10, 512, 512)]:
        cells = B * T * X * Y
        v = torch.empty(B, 6, T, X, Y, device=dev).uniform_(0.5, 1.5)
        mhd, ns = R.MHD(), R.NavierStokes(0.01, 1 / X, 1 / Y)
        out = torch.empty(B, T, X, Y, device=dev)
        for name, fn, bpc in (("mhd_induction", lambda: mhd.residual_induction(v, True), 20),
                              ("mhd_momentum", lambda: mhd.residual_momentum(v, True), 28),
                              ("mhd_energy", lambda: mhd.residual_energy(v, True), 28),
                              ("mhd_continuity", lambda: mhd.residual_continuity(v, True), 16),
                              ("ns_momentum", lambda: ns.residual_momentum(v[:, :3], True, out=out), 16)):
            ms = timeit(fn)
            print(f"{tag:8s} [{B},{T},{X},{Y}] {name:15s} {ms:8.3f} ms  {bpc * cells / ms / 1e6:7.1f} GB/s", flush=True)
        del v, out


if __name__ == "__main__":
    main()

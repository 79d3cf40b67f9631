"""Experiment: reference-layout grids with 64 < Ny <= 128 - regular tiles vs the flat form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cp_pre_amd import _lib
if os.environ.get('PRE_SO'):
    _lib.SO_PATH = os.environ['PRE_SO']
from cp_pre_amd import residuals as R
dev = torch.device("cuda:0")
def timeit(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (X, Y) in [(64, 64), (80, 80), (100, 100), (128, 128), (200, 100)]:
    B, T = max(8, (1 << 27) // (20 * X * Y)), 20
    v = torch.empty(B, 6, T, X, Y, device=dev).uniform_(0.5, 1.5)
    cells = B * T * X * Y
    ns, wave, mhd = R.NavierStokes(0.01, 1 / X, 1 / Y), R.PRE_Wave(0.01, 0.02), R.MHD()
    a = 16 * cells / timeit(lambda: ns.residual_momentum(v[:, :3], True)) / 1e6
    b = 8 * cells / timeit(lambda: wave.residual(v[:, 0], True)) / 1e6
    c = 20 * cells / timeit(lambda: mhd.residual_induction(v, True)) / 1e6
    print(f"[{B},{T},{X},{Y}]: ns {a:6.0f} GB/s  wave {b:6.0f}  mhd_induction {c:6.0f}", flush=True)
    del v

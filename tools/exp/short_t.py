"""Experiment: zero-copy Nt-fastest views with SHORT Nt (what real surrogate outputs look like: T_out = 10..32)."""
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
wave = R.PRE_Wave(0.01, 0.02)
ns = R.NavierStokes(0.01, 1 / 256, 1 / 256)
for T in [int(v) for v in os.environ.get('PRE_TS', '10,12,16,32,64').split(',')]:
    B = 2048 // T * 4
    phys = torch.empty(B, 3, 256, 256, T, device=dev).uniform_(0.5, 1.5)        # [BS,F,Nx,Ny,Nt]
    v = phys.permute(0, 1, 4, 2, 3)                                             # [BS,F,Nt,Nx,Ny], Nt fastest
    cells = B * T * 256 * 256
    ms = timeit(lambda: wave.residual(v[:, 0], True)); a = 8 * cells / ms / 1e6
    ms = timeit(lambda: wave.residual(v[:, 0].contiguous(), True)); b = 8 * cells / ms / 1e6
    ms = timeit(lambda: ns.residual_momentum(v, True)); c = 16 * cells / ms / 1e6
    ms = timeit(lambda: ns.residual_momentum(v.contiguous(), True)); d = 16 * cells / ms / 1e6
    print(f"T={T:3d} B={B:4d}: wave zero-copy {a:6.0f} GB/s | via contiguous() {b:6.0f} || ns zero-copy {c:6.0f} | via contiguous() {d:6.0f}", flush=True)
    del phys, v

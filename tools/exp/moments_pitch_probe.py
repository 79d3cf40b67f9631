"""pre_moments_segmax_f64 / pre_moments_axis0_f64 on C5-shaped data [n, 1, 200, 512] with the samples 102400 floats apart
(dense) and 102400 + 64: does the row pitch cost the moments pass what it costs the select?"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cp_pre_amd import _lib
dev = torch.device("cuda:0")
lib = _lib.load()
for (n, T, X, Y) in ((65536, 1, 200, 512), (8192, 1, 200, 512), (1024, 64, 256, 256)):
    M = T * X * Y
    for pad in (0, 64, 192):
        pitch = M + pad
        buf = torch.randn(n * pitch, device=dev)
        mom = torch.zeros(2, M, dtype=torch.float64, device=dev)
        TC, NS = (T + 15) // 16, (X * Y + 63) // 64
        seg = torch.empty(n * TC * NS, dtype=torch.int32, device=dev)
        def run(which):
            if which == "segmax":
                _lib.check(lib.pre_moments_segmax_f64(_lib.ptr(buf), pitch, n, T, X, Y, 1, 1, _lib.ptr(mom[0]), _lib.ptr(mom[1]), _lib.ptr(seg), _lib.stream()), "ms")
            else:
                _lib.check(lib.pre_moments_axis0_f64(_lib.ptr(buf), None, n, M, pitch, _lib.ptr(mom[0]), _lib.ptr(mom[1]), _lib.stream()), "m")
        out = []
        for which in ("segmax", "plain"):
            for _ in range(2):
                run(which)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run(which)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            out.append(f"{which} {ms:.3f} ms {4*n*M/ms/1e6:.0f} GB/s")
        print(f"[{n},{T},{X},{Y}] pitch M+{pad}: " + "   ".join(out), flush=True)
        del buf, mom, seg

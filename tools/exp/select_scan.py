"""Per-cell select over the batch axis, every regime of n, GB/s of ONE read of the scores, checked against torch.sort on a
slice.  Two row pitches per n: M = 2^k cells (every row of a tile on the same low address bits) and the same rows 64
floats further apart (what pipeline.row_padded / the drivers' residual buffers give)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cp_pre_amd import _lib, inductive_cp as icp

dev = torch.device("cuda:0")
lib = _lib.load()
alphas = [float(a) for a in icp.ALPHA_LEVELS]
ns = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else
      "20,50,100,128,130,160,168,170,200,256,300,400,512,640,800,1000,1024,1500,2048,3000,4096,8192").split(",")]
for n in ns:
    M = 1 << max(16, 32 - (4 * n - 1).bit_length())     # a power of two, 2.1-4.3 GB of scores
    res = []
    for pitch in (M, M + 64):
        torch.manual_seed(n)
        buf = torch.randn(n * pitch, device=dev).abs_()
        s = buf.as_strided((n, M), (pitch, 1))
        s[:, 5] = 1.0
        ks = [icp.kth_index(n, n, a) for a in alphas if icp.quantile_level(n, a) <= 1]
        for _ in range(3):
            q = icp.kth_axis0(s, ks)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            q = icp.kth_axis0(s, ks)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        ok = torch.equal(q[:, :2048], torch.sort(s[:, :2048], dim=0).values[ks])
        res.append(f"pitch M{'+64' if pitch > M else '   '}: {ms:7.3f} ms {4*n*M/ms/1e6:5.0f} GB/s exact={ok}")
        del buf, s
    print(f"n={n:5d} M={M:8d} ({4*n*M/1e9:.2f} GB)  " + "   ".join(res), flush=True)

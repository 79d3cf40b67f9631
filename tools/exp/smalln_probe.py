"""Per-cell select at small n (register sort / 16-row register tiles), GB/s of ONE read, checked against torch.sort."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cp_pre_amd import inductive_cp as icp
dev = torch.device("cuda:0")
alphas = [float(a) for a in icp.ALPHA_LEVELS]
for n in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "20,50,64,100,128,130,144,160,200,256").split(",")]:
    M = (3 << 30) // (4 * n) // 64 * 64 + 64           # ~3 GB, not a power of two
    torch.manual_seed(n)
    s = torch.randn(n, M, device=dev).abs_()
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
    ok = torch.equal(q[:, :4096], torch.sort(s[:, :4096], dim=0).values[ks])
    print(f"n={n} M={M} data={4*n*M/1e9:.2f} GB  {ms:.3f} ms  {4*n*M/ms/1e6:.0f} GB/s of one read  exact={ok}", flush=True)
    del s

"""Per-cell select at mid n (129..2048), GB/s of ONE read of the scores, checked against torch.sort on a slice
(n <= 1024: register tiles, kth_tile_kernel; above: the streaming form)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cp_pre_amd import inductive_cp as icp

dev = torch.device("cuda:0")
alphas = [float(a) for a in icp.ALPHA_LEVELS]
for (n, M) in [(130, 4194304), (200, 4194304), (256, 2621440), (300, 2621440), (400, 2097152), (512, 2097152), (640, 1048576), (800, 1048576), (1000, 1048576), (1024, 1048576), (1500, 524288), (2048, 524288)]:
    torch.manual_seed(n)
    s = torch.randn(n, M, device=dev).abs_()
    s[:, 5] = 1.0                                            # a column of ties
    s[: n // 2, 7] = 0.0
    ks = [icp.kth_index(n, n, a) for a in alphas]
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
    ref = torch.sort(s[:, :4096], dim=0).values[ks]
    ok = torch.equal(q[:, :4096], ref)
    print(f"n={n} M={M} data={4*n*M/1e9:.2f} GB  {ms:.3f} ms  {4*n*M/ms/1e6:.0f} GB/s of one read  exact={ok}", flush=True)
    del s

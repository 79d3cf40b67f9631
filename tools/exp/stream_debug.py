"""Debug helper for kth_stream_kernel: mismatches against torch.sort at a few n."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cp_pre_amd import inductive_cp as icp
dev = torch.device("cuda:0")
alphas = [float(a) for a in icp.ALPHA_LEVELS]
for (n, M, kind) in [(4096, 64 * 40, "abs"), (2049, 200, "abs"), (3000, 1000, "abs"), (6144, 64 * 300 + 3, "abs"), (3000, 65, "ties"), (5000, 200, "signed"), (3000, 63, "const")]:
    torch.manual_seed(n + M)
    s = torch.randn(n, M, device=dev)
    if kind == "abs": s = s.abs()
    if kind == "ties": s = torch.round(s * 2) / 4
    if kind == "const": s = torch.full_like(s, -7.5)
    ks = [icp.kth_index(n, n, a) for a in alphas]
    q = icp.kth_axis0(s, ks)
    torch.cuda.synchronize()
    ref = torch.sort(s, dim=0).values[ks]
    bad = q != ref
    print(f"n={n} M={M} {kind}: mismatches {int(bad.sum())} of {bad.numel()}", flush=True)
    if bad.any():
        idx = torch.nonzero(bad)
        print("  tiles:", sorted(set((idx[:, 1] // 64).tolist()))[:20], "ranks:", sorted(set(idx[:, 0].tolist())))
        for (j, c) in idx[:5].tolist():
            col = torch.sort(s[:, c]).values
            g = q[j, c].item()
            print(f"   j={j} k={ks[j]} cell={c} got={g!r} want={ref[j, c].item()!r} got-rank~{int((col < g).sum())}")

"""Debug helper for kth_tile_kernel: which cells / tiles / ranks differ from torch.sort."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cp_pre_amd import inductive_cp as icp

dev = torch.device("cuda:0")
alphas = [float(a) for a in icp.ALPHA_LEVELS]
for (n, M, special) in [(512, 64 * 6, False), (512, 64 * 300, False), (1000, 64 * 300, False), (512, 64 * 40, True)]:
    torch.manual_seed(n + M)
    s = torch.randn(n, M, device=dev).abs_()
    if special:
        s[:, 5] = 1.0
        s[: n // 2, 70] = 0.0
        s[3, 130] = float("inf")
    ks = [icp.kth_index(n, n, a) for a in alphas]
    q = icp.kth_axis0(s, ks)
    torch.cuda.synchronize()
    ref = torch.sort(s, dim=0).values[ks]
    bad = (q != ref)
    print(f"n={n} M={M} special={special}: mismatches {int(bad.sum())} of {bad.numel()}", flush=True)
    if bad.any():
        idx = torch.nonzero(bad)
        tiles = sorted(set((idx[:, 1] // 64).tolist()))
        print("  tiles with mismatches:", tiles[:40], "count", len(tiles))
        print("  ranks with mismatches:", sorted(set(idx[:, 0].tolist())))
        for (j, c) in idx[:8].tolist():
            col = torch.sort(s[:, c]).values
            g = q[j, c].item()
            pos = int((col < g).sum())
            print(f"   rank j={j} k={ks[j]} cell={c} (tile {c//64} lane {c%64}) got={g!r} want={ref[j, c].item()!r} got-is-rank~{pos}")

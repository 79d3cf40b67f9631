import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cp_pre_amd import inductive_cp as icp
dev = torch.device("cuda:0")
alphas = [float(a) for a in icp.ALPHA_LEVELS]
for n in (130, 256):
    M = (3 << 30) // (4 * n) // 64 * 64 + 64
    torch.manual_seed(n)
    s = torch.randn(n, M, device=dev).abs_()
    s[:, 5] = 1.0
    ks = [icp.kth_index(n, n, a) for a in alphas if icp.quantile_level(n, a) <= 1]
    q = icp.kth_axis0(s, ks)
    torch.cuda.synchronize()
    bad_total = 0
    for c0 in range(0, M, 1 << 20):
        c1 = min(M, c0 + (1 << 20))
        ref = torch.sort(s[:, c0:c1], dim=0).values[ks]
        neq = (q[:, c0:c1] != ref)
        if neq.any():
            idx = neq.nonzero()
            bad_total += len(idx)
            if bad_total <= 40:
                for j, c in idx[:10].tolist():
                    print(n, "rank", j, ks[j], "cell", c0 + c, "tile", (c0 + c) // 64, "lane", (c0 + c) % 64, "got", float(q[j, c0 + c]), "want", float(ref[j, c]))
    print("n", n, "M", M, "bad", bad_total, flush=True)

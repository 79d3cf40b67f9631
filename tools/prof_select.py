import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cp_pre_amd import inductive_cp as icp
dev = torch.device("cuda:0")
alphas = [float(a) for a in icp.ALPHA_LEVELS]
for (n, M) in [(4096, 524288), (256, 2621440)]:
    s = torch.randn(n, M, device=dev).abs_()
    ks = [icp.kth_index(n, n, a) for a in alphas]
    for _ in range(3):
        icp.kth_axis0(s, ks)
    torch.cuda.synchronize()
    del s

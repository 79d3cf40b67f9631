"""Tap-list kernels on the surrogate layout under rocprofv3 (--pmc FETCH_SIZE / SQ counters)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cp_pre_amd.convops_2d import ConvOperator
dev = torch.device("cuda:0")
D4 = ConvOperator(("x", "y"), 2, taylor_order=4)
for nt in (10, 40):
    xs = torch.randn(256 * 40 // nt, 256, 256, nt, device=dev).permute(0, 3, 1, 2)
    for _ in range(3):
        y = D4(xs)
    torch.cuda.synchronize()
    print(nt, xs.shape, 8 * xs.numel() / 1e9, "GB algorithmic")

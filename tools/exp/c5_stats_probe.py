import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cp_pre_amd import _lib, inductive_cp as icp, pipeline
dev = torch.device("cuda:0")
alphas = [float(a) for a in icp.ALPHA_LEVELS]
B = 65536
res = torch.randn(B, 1, 200, 512, device=dev)
orig = pipeline.HipOps.zeros_prune_stats
for rep in range(3):
    for tag in ("stats", "nostats"):
        pipeline.HipOps.zeros_prune_stats = orig if tag == "stats" else None
        ts = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            jc = pipeline.JointCalibration(B, dev)
            jc.add_slab(res, crop=(0, 1, 1))
            q = jc.finish(alphas)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"B={B} {tag}: calibrate {sorted(ts)[len(ts)//2]:.3f} ms", flush=True)

"""Per-cell select at one (n, M) on |N(0,1)| scores, a few launches: for rocprofv3 --pmc / --kernel-trace runs."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cp_pre_amd import _lib, inductive_cp as icp
if os.environ.get("PROBE_SO"):                       # an experimental build of the library (tools/exp/*.so)
    _lib.SO_PATH = os.path.abspath(os.environ["PROBE_SO"])

n, M = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
s = torch.randn(n, M, device=dev).abs_()
ks = [icp.kth_index(n, n, float(a)) for a in icp.ALPHA_LEVELS]
for _ in range(reps):
    q = icp.kth_axis0(s, ks)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    q = icp.kth_axis0(s, ks)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"n={n} M={M} {ms:.3f} ms  {4*n*M/ms/1e6:.0f} GB/s of one read", flush=True)

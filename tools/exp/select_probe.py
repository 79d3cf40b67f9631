"""Per-cell select on |N(0,1)| scores at a few n: run under rocprofv3 (--kernel-trace --stats, or --pmc FETCH_SIZE)
to read sweeps per tile (FETCH_SIZE x2 / data bytes) and time per launch."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cp_pre_amd import _lib, inductive_cp as icp
if os.environ.get("PROBE_SO"):                       # an experimental build of the library (tools/exp/*.so)
    _lib.SO_PATH = os.path.abspath(os.environ["PROBE_SO"])
    print("library:", _lib.SO_PATH, flush=True)

dev = torch.device("cuda:0")
alphas = [float(a) for a in icp.ALPHA_LEVELS]
for (n, M) in [(256, 2621440), (1024, 2621440), (2048, 1048576), (4096, 524288), (8192, 262144)]:
    s = torch.randn(n, M, device=dev).abs_()
    ks = [icp.kth_index(n, n, a) for a in alphas]
    for _ in range(3):
        q = icp.kth_axis0(s, ks)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        q = icp.kth_axis0(s, ks)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"n={n} M={M} data={4*n*M/1e9:.2f} GB  {ms:.3f} ms  {4*n*M/ms/1e6:.0f} GB/s per sweep-equivalent", flush=True)
    del s

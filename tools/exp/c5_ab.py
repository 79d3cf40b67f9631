"""C5 joint calibration (Burgers residual -> moments + segment maxima -> branch-and-bound score -> q-hat), two builds of the
library alternating in one process: python tools/exp/c5_ab.py [--alt tools/exp/prev/libcp_pre_hip_prev.so]"""
import argparse, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cp_pre_amd import _lib, inductive_cp as icp, pipeline
ap = argparse.ArgumentParser()
ap.add_argument("--alt", default=None)
args = ap.parse_args()
dev = torch.device("cuda:0")
main_path = _lib.SO_PATH
libs = {"new": _lib.load()}
if args.alt:
    _lib._lib, _lib.SO_PATH = None, args.alt
    libs["alt"] = _lib.load()
    _lib.SO_PATH = main_path
alphas = [float(a) for a in icp.ALPHA_LEVELS]
for B in (8192, 65536):
    res = torch.randn(B, 1, 200, 512, device=dev)
    for rep in range(3):
        for tag, lib in libs.items():
            _lib._lib = lib
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
            print(f"B={B} {tag}: calibrate {sorted(ts)[len(ts)//2]:.3f} ms  q0={float(q[0]):.6f}", flush=True)
    del res

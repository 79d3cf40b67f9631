import sys, os, torch
sys.path.insert(0, "/root/repo")
os.chdir("/root/repo")
import bench
from cp_pre_amd import residuals as R
dev = torch.device("cuda:0")
B,T,X,Y = 1024,64,256,256
# mimic measure_others: c1, c2 first, then c4
import argparse
for name in ("c1","c2"):
    cfg = bench.CONFIGS[name]; shp = cfg["shape"]
    a = argparse.Namespace(config=name, mode=cfg["mode"], batch=shp[0], nt=shp[1], nx=shp[2], ny=shp[3] if len(shp)==4 else 0, steps=5, warmup=2, no_prune=False, scaling="weak", slab=0, slab_axis="x", no_parity=True)
    bench.run_secondary(a, cfg, dev, None, 0, 1, {})
v = torch.empty(B,6,T,X,Y,device=dev)
for i in range(6): bench.synth_(v[:,i], 10+i, positive=i in (0,3))
fn = R.MHD().residual_induction
ts=[]
for k in range(12):
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = fn(v, boundary=True); e1.record(); torch.cuda.synchronize()
    ts.append(round(e0.elapsed_time(e1),2)); del r
print("induction per-launch ms:", ts)

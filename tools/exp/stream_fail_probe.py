import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cp_pre_amd import _lib, inductive_cp as icp
_lib.SO_PATH = os.path.join(ROOT, "tools/exp/var/libcp_pre_hip.kadbg.so")
lib = _lib.load()
dev = torch.device("cuda:0")
prev = np.zeros(8, dtype=np.uint64)
for n in (3000, 3800, 4096):
    M = 262144 + 64
    s = torch.randn(n, M, device=dev).abs_()
    for nk in (3, 10):
        alphas = [0.1, 0.5, 0.9] if nk == 3 else [float(a) for a in icp.ALPHA_LEVELS]
        ks = [icp.kth_index(n, n, a) for a in alphas]
        icp.kth_axis0(s, ks); torch.cuda.synchronize()
        buf = np.zeros(8, dtype=np.uint64)
        lib.pre_debug_ka_fail(ctypes.c_void_p(buf.ctypes.data))
        d = buf - prev
        print(f"n={n} nk={nk}: tiles {d[0]} many {d[1]} bad {d[2]} nan {d[3]} mean cmax {d[4]/max(d[0],1):.1f} max cmax {buf[5]} max ptr[0] {buf[6]}", flush=True)
        prev = buf.copy(); prev[5] = 0; prev[6] = 0

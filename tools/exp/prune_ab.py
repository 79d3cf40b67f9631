"""A/B of the joint calibration driver with and without the branch-and-bound score over reference-sized sets
(gpurun -- python tools/exp/prune_ab.py).  Prints ms per add_slab + finish."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cp_pre_amd import pipeline


def bench(n, shape, prune, reps=20, wild=False, slabs=1):
    dev = torch.device("cuda:0")
    torch.manual_seed(n + shape[0])
    res = torch.randn(n, *shape, device=dev)
    if wild:            # worst case for the bounds: the scale changes by orders of magnitude from cell to cell
        res *= torch.exp(3.0 * torch.randn(*shape, device=dev))
    al = [0.1 * k + 0.05 for k in range(10)]

    frac = [None]

    def once():
        jc = pipeline.JointCalibration(n, dev, prune=prune)
        for _ in range(slabs):                  # a stream of `slabs` slabs (the same tensor: the timing is what matters)
            jc.add_slab(res, crop=(1, 1, 1))
        q = jc.finish(al)
        frac[0] = jc.score_pass_read_frac()
        return q
    for _ in range(3):
        q = once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        q = once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3 / slabs, q, frac[0]


if __name__ == "__main__":
    pipeline.HipOps.PRUNE_MIN_CELLS = pipeline.HipOps.PRUNE_MIN_SAMPLES = 0
    for n in (100, 500, 1000, 4000):
        for shape in ((20, 64, 64), (30, 128, 128), (10, 256, 256), (60, 256, 256)):
            if n * shape[0] * shape[1] * shape[2] * 4 > 40e9:
                continue
            a, qa, _ = bench(n, shape, False)
            b, qb, fb = bench(n, shape, True)
            if n >= 1000:
                for slabs in (1, 4):            # one slab: only the in-kernel sweep adapts; a stream also drops the bounds after slab 1
                    aw, qaw, _ = bench(n, shape, False, wild=True, slabs=slabs)
                    bw, qbw, fw = bench(n, shape, True, wild=True, slabs=slabs)
                    print(f"n={n:5d} {list(shape)!s:16s} WILD per-cell scale e^(3 N(0,1)), {slabs} slab(s): full {aw:8.3f} ms  adaptive {bw:8.3f} ms per slab  "
                          f"x{aw/bw:5.2f}  read {fw:.2f} of the pruned slabs' segments  same={bool(torch.allclose(qaw, qbw, rtol=1e-5, equal_nan=True))}", flush=True)
            print(f"n={n:5d} {list(shape)!s:16s} cells={n*shape[0]*shape[1]*shape[2]:>12d}  full {a:8.3f} ms  pruned {b:8.3f} ms  x{a/b:5.2f}  "
                  f"read {fb:.3f}  same={bool(torch.allclose(qa, qb, rtol=1e-5, equal_nan=True))}", flush=True)

#!/usr/bin/env python3
"""bench.py - residual-cells/s of the PRE hot path (eval + calibrate) on MI355X.

Workload (BASELINE.json configs[2], the one the north_star target is quoted on): 2-D
Navier-Stokes momentum residual of (u, v, p) on [4096, 64, 512, 512] fp32 per rank, followed
by conformal calibration over the 4096 samples at the reference's 10 alpha levels.

One field of that shape is 275 GB, so the tensor is streamed as 8 t-slabs of
[4096, 10, 512, 512] (8 interior planes + the 2 halo planes their stencils read); the three
input slabs (129 GB) and the residual slab (43 GB) are resident in HBM before the timed
region.  Synthetic data: one resident slab of B + 7 samples stands in for the 8 slab positions, slab
position s reading the batch window [s, s + B) of it (825 GB of distinct input cannot be resident; the
arithmetic and traffic per slab do not depend on the values).  One STEP = the whole [4096,64,512,512] job = 8 slab passes of
    fused NS-momentum residual (one HIP launch)  ->  calibration on the resident residual slab
and the final q-hat selection.  cells/step = 4096*64*512*512 (uncropped grid).

--mode joint (default; Joint/NS_Residuals_CP.py recipe): per-cell moments -> modulation ->
    per-sample max|r|/sigma -> scalar q-hat x 10.  N>1: batch-sharded, weak scaling, one RCCL
    all-reduce of the moments per slab and ONE all-gather of the per-sample scores.
--mode marginal (Marginal/NS_Residuals_CP.py recipe): |r| -> per-cell q-hat x 10 by radix
    select over the batch axis.  N>1: all-to-all (batch-sharded -> cell-sharded) per slab.

--config c1|c2|c4|c5 run the other BASELINE.json configurations (per-rank shard sizes, whole
tensor resident, no slabs); they are secondary measurements quoted in DESIGN.md, not the
contract line.  Default: c3.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (fused
residual kernel, HIP events on its stream) and `cpu_baseline` (the CPU oracle = the
reference's own F.conv3d arithmetic, timed on this box's host cores; N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


# BASELINE.json configs: per-rank shapes (C4/C5 are quoted sharded over 8 GPUs), the fused kernel
# that evaluates them, its algorithmic bytes per cell and the script's CP flavour.
CONFIGS = {
    "c1": dict(kind="advection", shape=(256, 100, 200), bpc=8, mode="marginal", kernel="march_kernel<Linear1,8,64>",
               title="C1 1D advection additive kernel (Marginal/Advection_Residuals_CP.py)"),
    "c2": dict(kind="wave", shape=(512, 32, 256, 256), bpc=8, mode="marginal", kernel="march_kernel<Linear1,8,64>",
               title="C2 2D wave additive kernel (Marginal/Wave_Residuals_CP.py)"),
    "c3": dict(kind="ns", shape=(4096, 64, 512, 512), bpc=16, mode="joint", kernel="march_kernel<NSMomentum<0>,8,64>",
               title="C3 2D Navier-Stokes momentum residual"),
    "c4": dict(kind="mhd", shape=(1024, 64, 256, 256), bpc=20, mode="joint", kernel="march_kernel<MHDInduction<0>,8,64>",
               title="C4 2D MHD induction residual (Marginal/MHD_Residuals_CP.py:323), 8192/8 samples per rank"),
    "c5": dict(kind="burgers", shape=(8192, 200, 512), bpc=8, mode="joint", kernel="march_kernel<Burgers<0>,8,64>",
               title="C5 1D Burgers residual (Joint/Burgers_Residuals_CP.py), 65536/8 samples per rank"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=["joint", "marginal"], default=None, help="default: the config's own flavour")
    ap.add_argument("--batch", type=int, default=None, help="samples per rank (default: the config's)")
    ap.add_argument("--nt", type=int, default=None)
    ap.add_argument("--nx", type=int, default=None)
    ap.add_argument("--ny", type=int, default=None)
    ap.add_argument("--slab", type=int, default=8, help="interior planes per t-slab")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    shp = cfg["shape"]
    args.mode = args.mode or cfg["mode"]
    args.batch = args.batch or shp[0]
    args.nt = args.nt or shp[1]
    args.nx = args.nx or shp[2]
    args.ny = args.ny or (shp[3] if len(shp) == 4 else 0)
    return args


def synth_(field, seed, positive=False):
    """Fill one fp32 field [B, T, X(, Y)] in place with the synthetic data of SURVEY 8(d): a smooth mode
    sin(2 pi kx x) cos(2 pi ky y) cos(w t) plus 0.01 N(0,1) noise per cell (strictly positive U(0.5,1.5) for
    densities / pressures).  Generated on the device, no temporary of the field's size."""
    g = torch.Generator(device=field.device).manual_seed(seed)
    if positive:
        return field.uniform_(0.5, 1.5, generator=g)
    field.normal_(0.0, 0.01, generator=g)
    dims = field.shape[1:]
    kx, ky, w = 1 + seed % 3, 1 + (seed // 3) % 3, 1 + (seed // 9) % 2
    ax = [torch.linspace(0, 1, n, device=field.device) for n in dims]
    smooth = torch.cos(2 * math.pi * w * ax[0]).reshape(-1, *([1] * (len(dims) - 1)))
    smooth = smooth * torch.sin(2 * math.pi * kx * ax[1]).reshape(1, -1, *([1] * (len(dims) - 2)))
    if len(dims) == 3:
        smooth = smooth * torch.cos(2 * math.pi * ky * ax[2]).reshape(1, 1, -1)
    return field.add_(smooth)                               # broadcast over the batch axis


SYNTH = "synthetic smooth-plus-noise fields (sin*cos*cos mode + 0.01 N(0,1); U(0.5,1.5) for rho, p), generated on device"


def run_secondary(args, cfg, dev, group, rank, world):
    """c1/c2/c4/c5: whole per-rank tensor resident; one step = fused residual + calibration."""
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd import pipeline
    from cp_pre_amd import residuals as R
    B, T, X, Y = args.batch, args.nt, args.nx, args.ny
    alphas = [float(a) for a in icp.ALPHA_LEVELS]
    absolute = args.mode == "marginal"
    torch.manual_seed(1234 + rank)
    kind = cfg["kind"]
    if kind in ("advection", "burgers"):
        u = synth_(torch.empty(B, T, X, device=dev), 100 * rank + 1)
        u += 1.0                                            # advected / Burgers quantity of order one
        op = R.Advection(1.0, 0.005, 0.01, disc=2) if kind == "advection" else R.Burgers(2.0 / X, 1.25 / T, 0.002)
        evaluate = lambda: op.residual(u, boundary=True, absolute=absolute).unsqueeze(1)      # [B,1,T,X]
        crop, cells = (0, 1, 1), B * T * X
    elif kind == "wave":
        u = synth_(torch.empty(B, T, X, Y, device=dev), 100 * rank + 2)
        op = R.PRE_Wave(dt=0.005, dx=0.01, c=1.0)
        evaluate = lambda: op.residual(u, boundary=True, absolute=absolute)
        crop, cells = (1, 1, 1), B * T * X * Y
    else:
        v = torch.empty(B, 6, T, X, Y, device=dev)              # rho, u, v, p, Bx, By (Marginal/MHD_Residuals_CP.py:225)
        for i in range(6):
            synth_(v[:, i], 100 * rank + 10 + i, positive=i in (0, 3))
        op = R.MHD()
        evaluate = lambda: op.residual_induction(v, boundary=True, absolute=absolute)
        crop, cells = (1, 1, 1), B * T * X * Y
    ev = []

    def step():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        res = evaluate()
        e1.record()
        ev.append((e0, e1))
        if args.mode == "joint":
            jc = pipeline.JointCalibration(B, dev, group=group)
            jc.add_slab(res, crop=crop)
            return jc.finish(alphas)
        return pipeline.marginal_qhat(res, alphas, group=group)

    def sync():
        torch.cuda.synchronize()
        if group is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    ev.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        q = step()
    sync()
    elapsed = time.perf_counter() - t0
    if group is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        kms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
        launch_bytes = cfg["bpc"] * cells
        achieved = launch_bytes / (kms * 1e-3) / 1e9
        shape = [B, T, X] + ([Y] if Y else [])
        print(json.dumps({
            "metric": "residual-cells/s (PRE eval+calibrate)", "value": cells * world * args.steps / elapsed,
            "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": SYNTH + "; whole per-rank tensor resident",
            "config": {"workload": f"{cfg['title']} {shape} per rank, {args.mode} CP, 10 alpha levels", "mode": args.mode,
                       "parallelism": f"batch-sharded x{world}"},
            "roofline": {"bound": "hbm", "kernel": cfg["kernel"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": kms,
                         "algorithmic_bytes_per_launch": launch_bytes}}), flush=True)


def pmc_traffic(args):
    """HBM bytes per launch of the fused residual kernel from the committed rocprofv3 PMC passes
    (profiles/rNN/pmc_hbm_c3.json, written by tools/distill_profiles.py).  Counters cannot be
    read from inside this process; the file is used only when it was collected on this exact
    workload, otherwise `traffic` stays null."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_hbm_c3.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        w = d.get("workload", {})
        if (w.get("batch"), w.get("slab"), w.get("nx"), w.get("ny")) == (args.batch, args.slab, args.nx, args.ny):
            best = d
    return best


def host_cores():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota
    (a 1-GPU box of the pool shows 256 logical CPUs but is limited to 16)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(args, alphas):
    """The oracle (reference arithmetic: F.conv3d per operator + torch elementwise + numpy
    calibration) on a bounded sample of the same workload, on this box's host cores."""
    import numpy as np
    from oracle import conformal as oc
    from oracle import residuals as orr
    T = args.slab + 2
    dt, dx, dy = 1e-2, 1.0 / args.nx, 1.0 / args.ny
    threads = host_cores()
    torch.set_num_threads(threads)

    def run(nb):
        g = torch.Generator().manual_seed(0)
        v = torch.rand(nb, 3, T, args.nx, args.ny, generator=g) + 0.5
        t0 = time.perf_counter()
        res = orr.ns_momentum(v, dt, dx, dy, boundary=False).contiguous().numpy()
        if args.mode == "joint":
            mod = oc.modulation_func(res, np.zeros_like(res))
            sc = oc.ncf_metric_joint(res, np.zeros_like(res), mod)
            [oc.calibrate(sc, nb, a) for a in alphas if oc.quantile_level(nb, a) <= 1]
        else:
            s = np.abs(res)
            [oc.calibrate(s, nb, a) for a in alphas if oc.quantile_level(nb, a) <= 1]
        return time.perf_counter() - t0

    t_probe = run(4)
    nb = int(max(4, min(256, 4 * args.cpu_seconds / max(t_probe, 1e-3))))
    t = run(nb)
    cells = nb * args.slab * args.nx * args.ny            # useful (interior-plane) cells, as in `value`
    return {"value": cells / t, "unit": "cells/s", "cores": threads, "kind": "port",
            "sample": f"oracle NS-momentum + {args.mode} calibrate on [{nb},{T},{args.nx},{args.ny}] x3 fields "
                      f"(one slab, {nb}/{args.batch} of the batch), {t:.1f} s on {threads} torch threads"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
    # PRE_BENCH_REHEARSE=1: several ranks on ONE GPU over gloo - a plumbing rehearsal of the N>1
    # path on a single-GPU box (RCCL refuses two ranks per device); never a measurement.
    rehearse = os.environ.get("PRE_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    group = None
    if world > 1 or os.environ.get("PRE_BENCH_FORCE_GROUP") == "1":     # the latter: RCCL at world size 1 (plumbing check)
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)     # "nccl" is RCCL on ROCm
        group = dist.group.WORLD
        # create the communicator (and its xGMI rings) now, not inside the first timed collective
        warm = torch.zeros(1, dtype=torch.float64, device=dev)
        dist.all_reduce(warm)
        dist.all_gather([torch.empty_like(warm) for _ in range(dist.get_world_size())], warm)
        torch.cuda.synchronize()
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"

    if args.config != "c3":
        run_secondary(args, CONFIGS[args.config], dev, group, rank, world)
        if group is not None:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return

    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd import pipeline
    from cp_pre_amd.residuals import NavierStokes

    B, T, X, Y = args.batch, args.slab + 2, args.nx, args.ny
    n_slabs = max(1, args.nt // args.slab)
    alphas = [float(a) for a in icp.ALPHA_LEVELS]
    dt, dx, dy = 1e-2, 1.0 / X, 1.0 / Y
    ns = NavierStokes(dt, dx, dy, nu=1e-3)

    # resident synthetic slab: vars[:, i] views of one [B,3,T,X,Y] tensor, like the reference's `vars`
    torch.manual_seed(1234 + rank)
    # n_slabs - 1 extra samples: slab position s reads the batch window [s, s + B), so no two slab passes of a
    # step see the same input (and no layer of the memory system could serve one from another)
    vars_ = torch.empty(B + n_slabs - 1, 3, T, X, Y, dtype=torch.float32, device=dev)
    for i in range(3):
        synth_(vars_[:, i], 100 * rank + 20 + i)
    res = torch.empty(B, T, X, Y, dtype=torch.float32, device=dev)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(n_slabs * (args.steps + args.warmup))]
    ev_used = []

    def step(k):
        jc = pipeline.JointCalibration(B, dev, group=group) if args.mode == "joint" else None
        q = None
        for s in range(n_slabs):
            e0, e1 = ev[k * n_slabs + s]
            e0.record()
            # the slab's first and last plane are halo planes: every consumer crops them
            ns.residual_momentum(vars_[s:s + B], boundary=True, absolute=(args.mode == "marginal"), out=res, skip_t_rim=True)
            e1.record()
            ev_used.append((k, e0, e1))
            if jc is not None:
                jc.add_slab(res, crop=(1, 1, 1))
            else:
                q = pipeline.marginal_qhat(res, alphas, group=group)     # [10, T, X, Y]; caller keeps planes 1..T-2
        return jc.finish(alphas) if jc is not None else q

    def sync():
        torch.cuda.synchronize()
        if group is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    sync()
    t0 = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.steps):
        qhat = step(k)
    sync()
    elapsed = time.perf_counter() - t0
    if group is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())

    cells_per_step = B * n_slabs * args.slab * X * Y * world          # whole job, all ranks
    value = cells_per_step * args.steps / elapsed

    if rank == 0:
        durs = [e0.elapsed_time(e1) for (k, e0, e1) in ev_used if k >= args.warmup]     # ms, this rank
        kms = sum(durs) / len(durs)
        # 3 fields x T planes read, (T-2) interior planes written (the two halo planes of the slab are
        # neither computed nor stored, PRE_FLAG_INTERIOR_T): 12 B/cell in, 4 B/cell out
        launch_bytes = (12 * T + 4 * (T - 2)) * B * X * Y
        achieved = launch_bytes / (kms * 1e-3) / 1e9
        pmc = pmc_traffic(args)
        out = {
            "metric": "residual-cells/s (PRE eval+calibrate)",
            "value": value, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": SYNTH + "; one resident t-slab, slab position s reads the batch window [s, s+B) of it",
            "config": {"workload": f"C3 2D Navier-Stokes momentum residual [{B},{args.nt},{X},{Y}] x3 fields per rank, "
                                   f"{args.mode} CP, 10 alpha levels; streamed as {n_slabs} t-slabs of [{B},{T},{X},{Y}]",
                       "mode": args.mode, "batch_per_rank": B, "parallelism": f"batch-sharded x{world}"},
            "roofline": {"bound": "hbm", "kernel": "march_kernel<NSMomentum<0>,8,64>",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc["traffic_bytes_per_launch"] if pmc else None,
                         "traffic_source": pmc["source"] if pmc else None,
                         "avg_launch_ms": kms, "algorithmic_bytes_per_launch": launch_bytes},
            "qhat_first_last": [float(qhat.reshape(len(alphas), -1)[0, 0]), float(qhat.reshape(len(alphas), -1)[-1, 0])],
        }
        if world == 1 and not args.no_cpu_baseline:
            del vars_, res
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(args, alphas)
        print(json.dumps(out), flush=True)
    if group is not None:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py - residual-cells/s of the PRE hot path (eval + calibrate) on MI355X.

Workload (BASELINE.json configs[2], the one the north_star target is quoted on): 2-D
Navier-Stokes momentum residual of (u, v, p) on [4096, 64, 512, 512] fp32 per rank, followed
by conformal calibration over the 4096 samples at the reference's 10 alpha levels.

One field of that shape is 275 GB, so the tensor is streamed as SLABS, all 4096 samples of a slab at a time (a per-cell
std / quantile needs every sample of its cell).  Default: x-slabs (--slab-axis x) - the T axis whole (the kernel marches
along it with the planes in registers) and the reference's interior rows 1 .. Nx-2 (Marginal/NS_Residuals_CP.py:240 crops
rows 0 and Nx-1) cut into runs of --slab rows (default: the thickest whose resident set fits the free HBM: 128 -> 128, 128,
127, 127), each read with the row before and the row after it (PRE_FLAG_HALO_X: real rows of the grid, 2 rows re-read per
128).  The three input slabs [B,64,S+2,512] (209 GB at S = 128) and the residual buffer [B,64,S,512] (69 GB) are resident
in HBM before the timed region.  (--slab-axis t: t-slabs of --slab planes + 2 halo planes, rounds 1-2.)  Synthetic data:
one resident slab of B + n_slabs - 1 samples stands in for the slab positions, slab position s reading the batch window
[s, s + B) of it (825 GB of distinct input cannot be resident; the arithmetic and traffic per slab do not depend on the
values).  One STEP = the whole [4096,64,512,512] job = n_slabs passes of
    fused NS-momentum residual (one HIP launch)  ->  calibration on the resident residual slab
and the final q-hat selection.  cells/step = the cells the launches COMPUTE: 4096*64*510*512 (x-slabs: the reference's rows
0 and 511 exist only to be cropped and are not computed; the t and y rims are computed and cropped as in the reference).

--mode joint (default; Joint/NS_Residuals_CP.py recipe): per-cell moments -> modulation ->
    per-sample max|r|/sigma -> scalar q-hat x 10.  N>1: batch-sharded, weak scaling, one RCCL
    all-reduce of the moments per slab and ONE all-gather of the per-sample scores.
--mode marginal (Marginal/NS_Residuals_CP.py recipe): |r| -> per-cell q-hat x 10 by radix
    select over the batch axis.  N>1: all-to-all (batch-sharded -> cell-sharded) per slab.

--config c1|c2|c4|c5 run the other BASELINE.json configurations (per-rank shard sizes, whole
tensor resident, no slabs); they are secondary measurements quoted in DESIGN.md, not the
contract line.  Default: c3.  --equation continuity|momentum|energy|induction|gauss: which of C4's five MHD residuals
(Marginal/MHD_Residuals_CP.py:225-278; default induction, the script's own default :323).

--gpus N: one process per GPU.  Launched by the driver under torch.distributed.run (WORLD_SIZE set) the
process is one rank; launched bare with N > 1 it SPAWNS the N ranks itself (a fresh
`python -m torch.distributed.run --standalone` child, before anything touches the GPU) and exits with
the child's code; WORLD_SIZE != N is an error (exit 2).  --scaling weak (default): --batch samples per
rank; strong: the --batch samples are split over the ranks.

The default line (--config c3, joint, one GPU) also carries `secondary`: the other BASELINE configurations measured in
the same process after the C3 joint job (whose number stays `value`) - C3 marginal on the same resident slab, the per-rank
job of the strong-scaled 8-GPU curve ([512,64,512,512] x3 whole grid, joint and marginal, through a one-rank RCCL group) in
the reference layout and in the reference callers' Nt-fastest layout (`c3_rank8_ntfast`), C4 the way its script runs it
(`c4_marginal_rank8[_ntfast]`: |residual| written plane-major + ONE select at n = 8192), C1, C2, the C4 shard for each of its
five equations and with Nt-fastest fields, the C5 shard, C5 at its single-GPU size [65536,200,512], and the launch-bound
steps (C1, C2, C5 shard) as HIP graph replays - 2 warm-up + 5 steps each (the MEDIAN step), so that every config's number is
one the driver's own run produced (--no-secondary skips them).  Every entry carries `parity`: after its timed loop two
samples are re-evaluated by the CPU oracle (the checker, never the thing measured) and compared with what the HIP kernel
wrote (marginal entries: three q-hat cells against torch.sort as well); above 1e-5 the process exits 3.  The LAST key of the
line, `summary`, repeats [ms per step, roofline fraction, parity] of every config in ~0.7 KB.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (fused
residual kernel, HIP events on its stream; `achieved` = SURVEY 8(d) bytes: 16 B x the interior
cells one launch computes, halo planes NOT counted) and `cpu_baseline` (the CPU oracle = the
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
# The driver reads ONE JSON line from stdout.  RCCL prints a five-line version banner to stdout when its communicator is
# created (a run of round 5 came back with the banner in front of the line, and one - without `python -u` - with an empty
# stdout and rc 0), so a process that computes keeps a private duplicate of stdout for the line and points fd 1 at stderr
# for everything else, libraries included (`claim_stdout`, called once the launcher logic has decided that this process
# is a rank and not the spawner of the ranks, whose children must inherit the real stdout).
_LINE_FD = None


def claim_stdout():
    global _LINE_FD
    if _LINE_FD is None:
        sys.stdout.flush()
        _LINE_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    fd = _LINE_FD if _LINE_FD is not None else 1
    try:
        sys.stdout.flush()
    except Exception:
        pass
    while line:
        line = line[os.write(fd, line):]

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
RES_TOL = 1e-5                 # north_star: residuals within 1e-5 rel fp32 of the reference ConvOperator arithmetic


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

# C4's five equations (Marginal/MHD_Residuals_CP.py:225-278): fields read, algorithmic bytes per cell = 4 (F + 1) (SURVEY
# 8d), the kernel that evaluates each.  The script's default is induction (:323); SURVEY Appendix A sizes C4 on the
# six-field ones.
MHD_EQUATIONS = {
    "continuity": dict(bpc=16, kernel="march_kernel<MHDContinuity<0>,8,64>", lines="225-231"),
    "momentum": dict(bpc=28, kernel="march_kernel<MHDMomentum<0>,8,64>", lines="234-243"),
    "energy": dict(bpc=28, kernel="march_kernel<MHDEnergy<0>,8,64>", lines="247-256"),
    "induction": dict(bpc=20, kernel="march_kernel<MHDInduction<0>,8,64>", lines="259-268"),
    "gauss": dict(bpc=12, kernel="march_kernel<Linear2,8,64>", lines="271-278"),
}


def mhd_config(eq):
    e = MHD_EQUATIONS[eq]
    return dict(CONFIGS["c4"], bpc=e["bpc"], kernel=e["kernel"], equation=eq,
                title=f"C4 2D MHD {eq} residual (Marginal/MHD_Residuals_CP.py:{e['lines']}), 8192/8 samples per rank")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--equation", choices=sorted(MHD_EQUATIONS), default="induction",
                    help="c4: which of the five MHD residuals (default: the script's own, induction)")
    ap.add_argument("--layout", choices=["ny", "nt"], default="ny",
                    help="c4: nt = the fields in the surrogate's Nt-fastest layout, [BS,F,Nx,Ny,Nt].permute(0,1,4,2,3), as the "
                         "reference script passes them (Marginal/MHD_Residuals_CP.py:326-346); ny = contiguous [BS,F,Nt,Nx,Ny]")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=["joint", "marginal"], default=None, help="default: the config's own flavour")
    ap.add_argument("--batch", type=int, default=None, help="samples per rank (default: the config's)")
    ap.add_argument("--nt", type=int, default=None)
    ap.add_argument("--nx", type=int, default=None)
    ap.add_argument("--ny", type=int, default=None)
    ap.add_argument("--slab", type=int, default=0,
                    help="c3: rows per x-slab / interior planes per t-slab; 0 = the largest of N, N/2, N/3, N/4, ... whose "
                         "resident set fits the free HBM")
    ap.add_argument("--slab-axis", choices=["x", "t"], default="x",
                    help="c3: cut the grid into x-slabs (T whole, one halo ROW per side: 2/rows re-read; default) or into "
                         "t-slabs (two halo PLANES per slab: 2/planes re-read; rounds 1-3)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch samples per rank; strong: --batch samples in total, split over the ranks")
    ap.add_argument("--plumbing-check", action="store_true",
                    help="initialise the process group, run one all-reduce, print the rank count and exit (no compute)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default c3 joint line only: do not measure the other BASELINE configs after it")
    ap.add_argument("--no-parity", action="store_true",
                    help="c3: skip the post-timing oracle check of two samples of the last slab (profiling passes)")
    ap.add_argument("--no-prune", action="store_true",
                    help="joint mode: read the whole residual in the score pass instead of the branch-and-bound form")
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


def launch_plan(gpus, environ, argv, script=None):
    """What `bench.py --gpus N` has to do given the environment, decided before any GPU call:
    ("run", world)   - this process is one of `world` ranks (or the only one);
    ("spawn", cmd)   - bare `--gpus N` with N > 1: start N ranks under torch.distributed.run and relay the exit code;
    ("error", text)  - WORLD_SIZE contradicts --gpus (exit 2: a wrong-size run must never report a number)."""
    ws = environ.get("WORLD_SIZE")
    if ws is None:
        if gpus <= 1:
            return "run", 1
        return "spawn", [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1",
                         "--nnodes=1", "--nproc-per-node", str(gpus), script or os.path.abspath(__file__), *argv]
    if int(ws) != gpus:
        return "error", f"bench.py: WORLD_SIZE={ws} but --gpus {gpus}: launch with --nproc-per-node {gpus} (or pass --gpus {ws})"
    return "run", int(ws)


def note(msg):
    """Progress on stderr (rank 0's stdout carries the ONE JSON line and nothing else)."""
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def resident_bytes(B, n, slab, other):
    """HBM held by the c3 driver: three input slabs of B + n_slabs - 1 samples with their two halo planes (t-slabs) or
    rows (x-slabs) + the residual buffer of the slab's own planes / rows.  ``n``: extent of the slab axis that is cut
    (x-slabs: the interior rows, Nx - 2), ``other``: cells per unit of it (X*Y for t-slabs, T*Y for x-slabs)."""
    slabs = split_slabs(n, slab)
    S = max(slabs)
    return ((B + len(slabs) - 1) * 3 * (S + 2) + B * S) * other * 4 + B * 256      # (+ the residual rows' pad)


def split_slabs(nt, slab):
    """Interior planes per t-slab: nt planes in ceil(nt/slab) slabs of nearly equal size (64, 13 -> 13,13,13,13,12)."""
    n = max(1, -(-nt // slab))
    return [nt // n + (1 if i < nt % n else 0) for i in range(n)]


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


class C3Stream:
    """The C3 job as bench.py streams it - what `main()` times and what tests/test_gpu_fullsize.py::test_full_size_c3_xslab
    checks against the oracle: the resident synthetic slab, the residual buffer in each of its layouts, one slab pass.

    x-slabs (default): the reference's interior rows 1 .. X-2 (Marginal/NS_Residuals_CP.py:240 crops rows 0 and X-1) in
    runs of `slab` rows; a run of `sl` rows reads rows [0, sl + 2) of the resident slab - its own rows 1 .. sl and the row
    before / after them, real rows of the grid (PRE_FLAG_HALO_X) - with the T axis whole, and writes its own rows only.
    t-slabs: `sl` planes + one halo plane per side (PRE_FLAG_OUT_INTERIOR_T).  Slab position s reads the batch window
    [s, s + B) of the resident slab, so no two passes of a step see the same input."""

    def __init__(self, B, T, X, Y, slab, slab_axis, dev, rank=0, nu=1e-3):
        from cp_pre_amd.residuals import NavierStokes
        self.B, self.T, self.X, self.Y, self.dev = B, T, X, Y, dev
        self.xs = slab_axis == "x"
        self.n_axis, self.other = ((X - 2, T * Y) if self.xs else (T, X * Y))
        self.slabs = split_slabs(self.n_axis, slab)               # rows (x) / interior planes (t) per slab position
        self.S = max(self.slabs)
        self.dt, self.dx, self.dy, self.nu = 1e-2, 1.0 / X, 1.0 / Y, nu
        self.ns = NavierStokes(self.dt, self.dx, self.dy, nu=nu)
        n_slabs, S = len(self.slabs), self.S
        # vars[:, i] views of one [B',3,T,S+2,Y] (x-slabs) or [B',3,S+2,X,Y] (t-slabs) tensor, like the reference's `vars`
        self.vars_ = torch.empty((B + n_slabs - 1, 3, T, S + 2, Y) if self.xs else (B + n_slabs - 1, 3, S + 2, X, Y),
                                 dtype=torch.float32, device=dev)
        for i in range(3):
            synth_(self.vars_[:, i], 100 * rank + 20 + i)
        # residual buffer: the slab's OWN rows / planes only (room for either padded layout)
        self.res_buf = torch.empty(B * (self.other * S + 64 * (T if self.xs else S)), dtype=torch.float32, device=dev)
        # cells within `crop` of the slab's rim are excluded from the scores: the y rim always; x-slabs: the grid's own t
        # rim (every plane is computed, zero padding beyond, as the reference's conv3d has it), no rows (all of a slab's
        # rows are interior rows of the grid); t-slabs: the x rim, no planes (all interior)
        self.crop = (1, 0, 1) if self.xs else (0, 1, 1)

    def rshape(self, sl):
        return (self.B, self.T, sl, self.Y) if self.xs else (self.B, sl, self.X, self.Y)

    def plane(self, sl):
        """Cells of one time plane of the residual."""
        return sl * self.Y if self.xs else self.X * self.Y

    def views(self, mode, sharded=False):
        """The residual buffer as each slab thickness sees it: {sl: tensor [B, ...]}."""
        B, Y, buf, rshape, plane = self.B, self.Y, self.res_buf, self.rshape, self.plane
        if mode == "marginal" and sharded:
            # time-major [planes][B][plane] seen as [B,planes,..]: plane t of all local samples is one contiguous block, the
            # send block of the all-to-all that hands plane t to rank t % world (pipeline.marginal_qhat: no pack copy)
            # (samples of a plane 64 floats further apart than a plane is long: profiles/r03/row_pitch.txt)
            return {sl: buf.as_strided(rshape(sl), (plane(sl) + 64, B * (plane(sl) + 64), Y, 1)) for sl in set(self.slabs)}
        if mode == "marginal":
            # rows 64 floats further apart than they are long: a power-of-two distance between the rows of a cell's column
            # (2^22 cells) costs the per-cell select ~10 % (pipeline.row_padded, profiles/r03/row_pitch.txt)
            return {sl: buf.as_strided(rshape(sl), (rshape(sl)[1] * plane(sl) + 64, plane(sl), Y, 1)) for sl in set(self.slabs)}
        return {sl: buf[:B * rshape(sl)[1] * plane(sl)].view(rshape(sl)) for sl in set(self.slabs)}

    def inputs(self, s, sl):
        """The view of the resident slab that slab position s (sl rows / planes) hands the kernel."""
        if self.xs:
            return self.vars_[s:s + self.B, :, :, 1:sl + 1]             # rows 0 and sl + 1 are read as halo rows
        return self.vars_[s:s + self.B, :, :sl + 2]

    def eval_slab(self, s, sl, res, absolute=False):
        """ONE fused launch: the residual of slab position s into `res` (a view from `views`)."""
        if self.xs:
            return self.ns.residual_momentum(self.inputs(s, sl), boundary=True, absolute=absolute, out=res, halo_x=True)
        return self.ns.residual_momentum(self.inputs(s, sl), boundary=True, absolute=absolute, out=res, skip_t_rim=True)

    def oracle_inputs(self, s, sl, i):
        """Host copy of what sample i of slab position s reads, halo rows / planes included: [1,3,T,sl+2,Y] / [1,3,sl+2,X,Y]."""
        v = self.vars_[s + i, :, :, 0:sl + 2] if self.xs else self.vars_[s + i, :, :sl + 2]
        return v.unsqueeze(0).cpu()

    def oracle_crop(self, ref):
        """The slab's own rows / planes of an oracle residual evaluated on `oracle_inputs` with boundary=True."""
        return ref[:, :, 1:-1] if self.xs else ref[:, 1:-1]

    def free(self):
        self.vars_ = self.res_buf = None
        torch.cuda.empty_cache()


def c3_parity(stream, res, s, sl, samples, absolute=False):
    """CHECKER, outside every timed region: samples `samples` of the residual slab `res` the HIP kernel wrote for slab
    position s, against the CPU oracle (oracle/residuals.py::ns_momentum = the reference's own F.conv3d arithmetic,
    Marginal/NS_Residuals_CP.py:231-240) evaluated on the same rows with their halo rows.  Tensor-scale relative error."""
    from oracle import residuals as orr
    worst = 0.0
    for i in samples:
        v = stream.oracle_inputs(s, sl, i)
        ref = stream.oracle_crop(orr.ns_momentum(v, stream.dt, stream.dx, stream.dy, nu=stream.nu, boundary=True))[0]
        ref = ref.abs() if absolute else ref                          # (the marginal score: |.| epilogue of the kernel)
        got = res[i].cpu()
        worst = max(worst, float((got - ref).abs().max() / ref.abs().max()))
    return worst


def oracle_parity(kind, eq, inputs, res, samples, absolute):
    """CHECKER, outside every timed region: samples `samples` of a residual tensor the HIP kernel wrote, against the CPU
    oracle (oracle/residuals.py = the reference's own F.conv3d / F.conv2d arithmetic per operator) evaluated on the same
    inputs with the coefficients `run_secondary` uses; |.| of it for the marginal scores.  Tensor-scale relative error."""
    from oracle import residuals as orr
    worst = 0.0
    for i in samples:
        x = inputs[i:i + 1].cpu()
        if kind == "advection":
            ref = orr.advection_residual(x, 1.0, 2, 0.005, 0.01, boundary=True)
        elif kind == "burgers":
            ref = orr.burgers_residual(x, 2.0 / x.shape[2], 1.25 / x.shape[1], 0.002, boundary=True)
        elif kind == "wave":
            ref = orr.wave_residual(x, 1.0, 0.005, 0.01, boundary=True)
        elif kind == "ns":
            ref = orr.ns_momentum(x, 1e-2, 1.0 / x.shape[3], 1.0 / x.shape[4], nu=1e-3, boundary=True)
        else:
            ref = getattr(orr, "mhd_" + eq)(x, boundary=True)
        ref = ref[0].abs() if absolute else ref[0]
        got = res[i].cpu().reshape(ref.shape)
        worst = max(worst, float((got - ref).abs().max() / ref.abs().max()))
    return worst


def surrogate_layout(B, F, T, X, Y, dev):
    """Fields as the reference's callers hand them over (Marginal/NS_Residuals_CP.py:282-287, MHD_Residuals_CP.py:326-346):
    the surrogate's output [BS,F,Nx,Ny,Nt] seen through permute(0,1,4,2,3) - a [BS,F,Nt,Nx,Ny] view whose FASTEST axis is
    Nt.  Consumed in place (the library relabels its axes), the residual comes back in the same memory order."""
    return torch.empty(B, F, X, Y, T, dtype=torch.float32, device=dev).permute(0, 1, 4, 2, 3)


def run_secondary(args, cfg, dev, group, rank, world, par):
    """c1/c2/c4/c5: whole per-rank tensor resident; one step = fused residual + calibration.  Returns the JSON line
    (rank 0; None elsewhere).  cfg["layout"] == "nt" (c4): the fields in the surrogate's Nt-fastest layout."""
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
        crop, cells, inputs = (0, 1, 1), B * T * X, u
    elif kind == "wave":
        u = synth_(torch.empty(B, T, X, Y, device=dev), 100 * rank + 2)
        op = R.PRE_Wave(dt=0.005, dx=0.01, c=1.0)
        # marginal: the residual's rows (samples) 64 floats further apart than they are long - a power-of-two row
        # distance costs the per-cell select that follows 4 % (pipeline.row_padded, profiles/r03/row_pitch.txt)
        wout = pipeline.row_padded(B, (T, X, Y), device=dev) if (args.mode == "marginal" and group is None) else None
        evaluate = lambda: op.residual(u, boundary=True, absolute=absolute, out=wout)
        crop, cells, inputs = (1, 1, 1), B * T * X * Y, u
    else:
        # rho, u, v, p, Bx, By (Marginal/MHD_Residuals_CP.py:225)
        v = surrogate_layout(B, 6, T, X, Y, dev) if cfg.get("layout") == "nt" else torch.empty(B, 6, T, X, Y, device=dev)
        for i in range(6):
            synth_(v[:, i], 100 * rank + 10 + i, positive=i in (0, 3))
        op = R.MHD()
        fn = getattr(op, "residual_" + cfg.get("equation", "induction"))
        evaluate = lambda: fn(v, boundary=True, absolute=absolute)
        crop, cells, inputs = (1, 1, 1), B * T * X * Y, v
    ev = []
    pruned = [False]
    last_jc = [None]

    def step():
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        res = evaluate()
        e1.record()
        ev.append((e0, e1, e2))
        if args.mode == "joint":
            jc = pipeline.JointCalibration(B, dev, group=group, prune=not args.no_prune)
            last_jc[0] = jc
            pruned[0] = jc.prune and pipeline.HipOps.can_prune(res, crop)
            jc.add_slab(res, crop=crop)
            q = jc.finish(alphas)
        else:
            q = pipeline.marginal_qhat(res, alphas, group=group)
        e2.record()
        return q

    def sync():
        torch.cuda.synchronize()
        if group is not None:
            torch.distributed.barrier(group=group)
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
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX, group=group)
        elapsed = float(tt.item())
    if rank == 0:
        # (this function also serves `--config c1|c2|c4|c5` as a line of its own: the mean there, the median as a secondary)
        avg = med if getattr(args, "median", False) else (lambda xs: sum(xs) / len(xs))
        kms = avg([a.elapsed_time(b) for a, b, _ in ev])
        cms = avg([b.elapsed_time(c) for _, b, c in ev])                     # calibration: everything after the residual kernel
        if getattr(args, "median", False):                                   # (a secondary: the median step, device time)
            elapsed = 1e-3 * med([a.elapsed_time(c) for a, _, c in ev]) * args.steps
        pmc = pmc_traffic(args, cfg.get("pmc_key", cfg.get("equation")))
        launch_bytes = cfg["bpc"] * cells
        achieved = launch_bytes / (kms * 1e-3) / 1e9
        shape = [B, T, X] + ([Y] if Y else [])
        calib = {"calibrate_ms": cms}
        if not getattr(args, "no_parity", False):
            # CHECKER (outside the timed region): the first and the last sample of one more evaluation against the CPU oracle
            res_chk = evaluate()
            torch.cuda.synchronize()
            err = oracle_parity(kind, cfg.get("equation", "induction"), inputs, res_chk, (0, B - 1), absolute)
            calib["parity"] = {"residual_rel_err": err, "tol": RES_TOL, "samples": 2, "ok": bool(err <= RES_TOL)}
            del res_chk
        if args.mode == "marginal":         # the per-cell select reads the scores once, algorithmically (4 B per cell and pass)
            calib["select_one_read_gbs"] = 4.0 * cells / (cms * 1e-3) / 1e9
        elif last_jc[0] is not None and pruned[0]:
            calib["score_pass_read_frac"] = last_jc[0].score_pass_read_frac()
        return {
            "metric": "residual-cells/s (PRE eval+calibrate)", "value": cells * world * args.steps / elapsed,
            "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": SYNTH + "; whole per-rank tensor resident",
            "config": {"workload": f"{cfg['title']} {shape} per rank, {args.mode} CP, 10 alpha levels", "mode": args.mode, **par,
                       **({"score_pass": "branch-and-bound (same scores as the full pass for the same modulation)" if pruned[0] else "full"}
                          if args.mode == "joint" else {})},
            "roofline": {"bound": "hbm", "kernel": cfg["kernel"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc["traffic_bytes_per_launch"] if pmc else None,
                         "traffic_source": pmc["source"] if pmc else None, "avg_launch_ms": kms,
                         "algorithmic_bytes_per_launch": launch_bytes},
            **calib}
    return None


def med(xs):
    """Median of a secondary's per-step event intervals (5 steps each): one step that had to wait for the host - a
    hipMalloc inside the caching allocator after the previous config's buffers went back to the driver - would otherwise
    decide a secondary's kernel_ms / frac (seen once in round 6: 22.2 ms "mean" over launches of 15.9, 15.9, 47, ...).  The
    headline's roofline stays the MEAN over its launches, as the contract says."""
    xs = sorted(xs)
    return xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])


def secondary_entry(line):
    """The compact form of a config's line inside the default line's `secondary` object."""
    r = line["roofline"]
    e = {"workload": line["config"]["workload"], "ms_per_step": line["ms_per_step"], "cells_per_s": line["value"],
         "steps": line["steps"], "warmup": line["warmup"], "kernel": r["kernel"], "kernel_ms": r["avg_launch_ms"],
         "achieved_gbs": r["achieved"], "frac": r["frac"], "traffic": r["traffic"]}
    for k in ("calibrate_ms", "select_one_read_gbs", "score_pass_read_frac", "score_pass", "parity"):
        if k in line:
            e[k] = line[k]
        elif k in line["config"]:
            e[k] = line["config"][k]
    return e


def summary_of(out):
    """{config: [ms per step, fraction of the 8 TB/s HBM peak its residual kernel reaches (SURVEY 8d bytes), oracle parity =
    worst tensor-scale relative error of the checked samples]} for the headline and every secondary, compact (3 significant
    digits) - the last key of the line.  Short keys: c4_<equation> = the C4 shard's joint-CP job per equation (c4_ind =
    `c4_shard`), *_nt = the same fed the reference callers' Nt-fastest views, c4_marg = `c4_marginal_rank8` (kernel + select
    at n = 8192), r8 = `c3_strong_rank8`, *_g = the launch-bound step as one HIP graph replay ([ms])."""
    def r3(x):
        return None if x is None else float(f"{x:.3g}")

    def par(e):
        p = e.get("parity") if isinstance(e, dict) else None
        return r3(p.get("residual_rel_err")) if p else None
    s = {"c3": [r3(out["ms_per_step"]), r3(out["roofline"]["frac"]), par(out)]}
    short = {"c3_marginal": "c3_marg", "c4_shard": "c4_ind", "c4_continuity": "c4_cont", "c4_momentum": "c4_mom", "c4_energy": "c4_en",
             "c4_gauss": "c4_gauss", "c4_shard_ntfast": "c4_ind_nt", "c4_marginal_rank8": "c4_marg", "c4_marginal_rank8_ntfast": "c4_marg_nt",
             "c5_shard": "c5", "c5_whole": "c5_whole", "c1": "c1", "c2": "c2"}
    for key, e in (out.get("secondary") or {}).items():
        if not isinstance(e, dict) or "error" in e:
            s[short.get(key, key.replace("_graph", "_g"))] = "error"
        elif key in ("c3_strong_rank8", "c3_rank8_ntfast"):
            tag = "c3_r8" if key == "c3_strong_rank8" else "c3_r8_nt"
            for mode in ("joint", "marginal"):
                if mode in e:
                    s[tag + ("_j" if mode == "joint" else "_m")] = [r3(e[mode]["ms_per_step"]), r3(e[mode]["frac"]), par(e)]
        elif key in short:
            s[short[key]] = [r3(e.get("ms_per_step")), r3(e.get("frac")), par(e)]
        elif key.endswith("_graph"):                     # the step as ONE HIP graph replay: [ms], equal to the eager step or "differs"
            s[key.replace("_graph", "_g")] = [r3(e.get("ms_per_step_graph"))] if e.get("replay_equals_eager") else "differs"
    return s


def measure_others(dev):
    """C1 (the reference's own CPU-sized case: launch-bound here), C2, the C4 shard for each of its five equations
    (`c4_shard` = induction, the script's default), the C5 shard and C5 whole, one after the other (each frees its tensors
    on return)."""
    found = {}
    jobs = [("c1", "c1", None, None), ("c2", "c2", None, None)]
    jobs += [("c4_shard" if eq == "induction" else "c4_" + eq, "c4", None, eq)
             for eq in ("induction", "continuity", "momentum", "energy", "gauss")]
    # the C4 shard fed the way Marginal/MHD_Residuals_CP.py:326-346 feeds it: cal_pred.permute(0,1,4,2,3), Nt fastest
    jobs += [("c4_shard_ntfast", "c4", None, "induction:nt")]
    jobs += [("c5_shard", "c5", None, None), ("c5_whole", "c5", 65536, None)]
    for key, name, batch, eq in jobs:
        note(f"secondary {key}")
        layout = None
        if eq and ":" in eq:
            eq, layout = eq.split(":")
        cfg = mhd_config(eq) if eq else CONFIGS[name]
        if layout == "nt":
            cfg = dict(cfg, layout="nt", pmc_key=eq + "_ntfast", kernel=cfg["kernel"].replace("march_kernel<MHDInduction<0>,8,64>", "flat_march_kernel<MHDInduction<3>>"),
                       title=cfg["title"] + ", fields in the surrogate's Nt-fastest layout [BS,F,Nx,Ny,Nt].permute(0,1,4,2,3) (:326-346)")
        shp = cfg["shape"]
        a = argparse.Namespace(config=name, mode=cfg["mode"], batch=batch or shp[0], nt=shp[1], nx=shp[2],
                               ny=shp[3] if len(shp) == 4 else 0, steps=5, warmup=2, no_prune=False, scaling="weak", slab=0,
                               slab_axis="x", no_parity=False, median=True)
        if batch is not None:
            cfg = dict(cfg, title="C5 1D Burgers residual (Joint/Burgers_Residuals_CP.py) at its single-GPU size", pmc_key="whole")
        try:
            line = run_secondary(a, cfg, dev, None, 0, 1, {})
            found[key] = secondary_entry(line)
        except Exception as e:                                    # (a secondary must never cost the contract line)
            found[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.empty_cache()
    return found


def measure_graph(dev, name, reps=None):
    """The launch-bound configs as ONE HIP graph replay per step.  C1 (a 40 us kernel + a 40 us select behind ~10 host
    launches), C2 (1.6 ms + 1.1 ms) and the C5 shard (1.3 ms + six calibration launches of 0.01-0.6 ms): the same step as the
    eager secondary - fused residual -> calibration at the 10 levels - recorded ONCE and replayed; the library never
    allocates, synchronises or reads device data on the host path, and the joint stream takes the fixed branch-and-bound
    route (`prune="always"`: no counter is read back).  A replay is checked against an eager run
    (tests/test_gpu_parity.py::test_marginal_step_is_hip_graph_capturable, ::test_joint_stream_is_hip_graph_capturable_...)."""
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd import pipeline
    from cp_pre_amd import residuals as R
    shp = CONFIGS[name]["shape"]
    alphas = [float(a) for a in icp.ALPHA_LEVELS]
    cells = 1
    for d in shp:
        cells *= d
    if name == "c1":
        u = synth_(torch.empty(*shp, device=dev), 1) + 1.0
        op = R.Advection(1.0, 0.005, 0.01, disc=2)
        step = lambda: pipeline.marginal_qhat(op.residual(u, boundary=True, absolute=True), alphas)
        what = "marginal step (residual + per-cell q-hat x10)"
    elif name == "c2":
        u = synth_(torch.empty(*shp, device=dev), 2)
        op = R.PRE_Wave(dt=0.005, dx=0.01, c=1.0)
        wout = pipeline.row_padded(shp[0], shp[1:], device=dev)
        step = lambda: pipeline.marginal_qhat(op.residual(u, boundary=True, absolute=True, out=wout), alphas)
        what = "marginal step (residual + per-cell q-hat x10)"
    else:
        u = synth_(torch.empty(*shp, device=dev), 1) + 1.0
        op = R.Burgers(2.0 / shp[2], 1.25 / shp[1], 0.002)

        def step():
            jc = pipeline.JointCalibration(shp[0], dev, prune="always")
            jc.add_slab(op.residual(u, boundary=True).unsqueeze(1), crop=(0, 1, 1))
            return jc.finish(alphas)
        what = "joint step (residual + moments + modulation + branch-and-bound scores + q-hat x10)"
    reps = reps or (200 if name == "c1" else 30)

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    eager = timed(step, reps)
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        step()
    cur.wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        q = step()
    graph.replay()
    torch.cuda.synchronize()
    same = bool(torch.equal(q, step()))
    replay = timed(graph.replay, reps)
    return {"workload": f"{name.upper()} {list(shp)} {what} as ONE HIP graph replay", "steps": reps,
            "ms_per_step_eager": eager, "ms_per_step_graph": replay, "cells_per_s_graph": cells / (replay * 1e-3),
            "replay_equals_eager": same}


def one_rank_group(dev):
    """An RCCL ("nccl" on ROCm) process group of ONE rank on this GPU: what every collective of the sharded path costs when
    nothing has to travel.  Returns (group, note); (None, reason) when the communicator cannot be created here."""
    import socket
    import torch.distributed as dist
    try:
        if not dist.is_initialized():
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
        warm = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(warm)
        torch.cuda.synchronize()
        return dist.group.WORLD, "RCCL, world size 1"
    except Exception as e:                                        # never lose the contract line to a communicator problem
        return None, f"no RCCL group on this box: {type(e).__name__}: {e}"[:200]


def measure_strong_rank(dev, alphas, world=8):
    """`c3_strong_rank8`: the per-rank job of the STRONG-scaled 8-GPU curve north_star's ">= 6x" speaks of - C3's 4096
    samples split over 8 ranks = [512,64,512,512] x3 per rank, the whole grid resident (137 GB: one slab of all 510
    interior rows, no halo re-read), joint and marginal CP through the sharded code path with a process group of one rank
    (every collective issued, nothing on the wire).  Numerator of the projected speed-up in DESIGN.md 5; a one-GPU
    measurement, not a multi-GPU one."""
    from cp_pre_amd import pipeline
    shp = CONFIGS["c3"]["shape"]
    B, T, X, Y = shp[0] // world, shp[1], shp[2], shp[3]
    group, how = one_rank_group(dev)
    note(f"secondary c3_strong_rank8 ({how})")
    st = C3Stream(B, T, X, Y, X - 2, "x", dev)
    assert st.slabs == [X - 2]
    out = {"workload": f"C3 strong-scaled over {world} ranks: the per-rank job [{B},{T},{X},{Y}] x3, whole grid resident (one slab of "
                       f"{X - 2} interior rows, no halo re-read), calibration through the sharded path with {how}",
           "group": how, "steps": 5, "warmup": 2}
    cells = B * T * (X - 2) * Y
    for mode in ("joint", "marginal"):
        res = st.views(mode, sharded=(mode == "marginal" and group is not None))[X - 2]
        ev = []

        def step():
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            st.eval_slab(0, X - 2, res, absolute=(mode == "marginal"))
            e1.record()
            if mode == "joint":
                jc = pipeline.JointCalibration(B, dev, group=group)
                jc.add_slab(res, crop=st.crop)
                q = jc.finish(alphas)
            else:
                q = pipeline.marginal_qhat(res, alphas, group=group)
            e2.record()
            ev.append((e0, e1, e2))
            return q
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ev.clear()
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 5
        kms = med([a.elapsed_time(b) for a, b, _ in ev])
        cms = med([b.elapsed_time(c) for _, b, c in ev])
        ach = 16.0 * cells / (kms * 1e-3) / 1e9
        out[mode] = {"ms_per_step": ms, "cells_per_s_per_rank": cells / (ms * 1e-3), "kernel_ms": kms, "calibrate_ms": cms,
                     "achieved_gbs": ach, "frac": ach / HBM_PEAK_GBS}
        if mode == "joint":                                         # CHECKER (outside the timed loop): two samples vs the oracle
            err = c3_parity(st, res, 0, X - 2, (0, B - 1))
            out["parity"] = {"residual_rel_err": err, "tol": RES_TOL, "samples": 2, "ok": bool(err <= RES_TOL)}
        del res
    st.free()
    return out


def plane_major(B, T, X, Y, layout, dev, pad=64):
    """The residual / score buffer of the sharded marginal flow, as the logical [B,T,X,Y] view the residual kernel writes
    through, + its description for the exchange (planes, cells per plane, floats from one sample's plane to the next's).
    layout "ny" (reference layout, Ny fastest): memory [T][B][X*Y + pad] (`pipeline.time_major`) - plane t of all local
    samples is one contiguous block, the send block of the all-to-all that hands plane t to rank t % world.
    layout "nt" (the surrogate's layout, Nt fastest - what the reference script passes): the fastest axis cannot be the
    one the planes are cut along, so the planes are those of Nx: memory [X][B][Y*T + pad]."""
    from cp_pre_amd import pipeline
    if layout == "ny":
        out = pipeline.time_major(B, (T, X, Y), pad=pad, device=dev)
        return out, out, T, X * Y, X * Y + pad
    per, pitch = Y * T, Y * T + pad
    buf = torch.empty(X * B * pitch, dtype=torch.float32, device=dev)
    out = buf.as_strided((B, T, X, Y), (pitch, 1, B * pitch, T))
    return out, out.permute(0, 2, 3, 1), X, per, pitch            # [B,X,Y,T]: "time-major" with Nx in the plane role


def measure_c4_marginal(dev, alphas, layout, world=8, eq="induction", steps=5, warmup=2):
    """`c4_marginal_rank8`: BASELINE config 4 the way Marginal/MHD_Residuals_CP.py runs it - MARGINAL CP (:408-418:
    ncf_scores = |residual|, calibrate per cell over the n_cal = 8192 samples) - as the per-rank job of the 8-way sharded
    flow (pipeline.marginal_qhat):  (1) residual_<eq> of the rank's 1024 samples with the |.| epilogue, written PLANE-MAJOR
    (the send blocks of the exchange);  (2) [the exchange: plane k goes to rank k % 8 - wire time, projected in DESIGN 5,
    nothing to measure on one GPU];  (3) ONE select launch at n = 8192 over the planes the rank owns (1/8 of them), at
    the shape and pitch the receive staging has: [planes/8][8192][cells + 64].  The other seven ranks' rows are stood in
    for by the rank's own scores of seven other planes (the plane-major buffer [P][1024][pitch] IS [P/8][8192][pitch]:
    same bytes, same distribution, no copy).  Also timed: the same scores selected at n = 1024 over all planes (what a
    one-rank run of the flow does).  layout: "ny" (synthetic benchmark layout) / "nt" (the script's own:
    cal_pred.permute(0,1,4,2,3), :326-346)."""
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd import pipeline
    from cp_pre_amd import residuals as R
    B, T, X, Y = CONFIGS["c4"]["shape"]
    v = surrogate_layout(B, 6, T, X, Y, dev) if layout == "nt" else torch.empty(B, 6, T, X, Y, device=dev)
    for i in range(6):
        synth_(v[:, i], 10 + i, positive=i in (0, 3))
    fn = getattr(R.MHD(), "residual_" + eq)
    out, tm, P, per, pitch = plane_major(B, T, X, Y, layout, dev)
    assert pipeline._is_time_major(tm) and P % world == 0
    own, n_all = P // world, B * world
    ks1, ks8 = [icp.kth_index(B, B, a) for a in alphas], [icp.kth_index(n_all, n_all, a) for a in alphas]
    q8 = torch.empty(own, len(alphas), per, dtype=torch.float32, device=dev)
    ev = []

    def step():
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        fn(v, boundary=True, absolute=True, out=out)
        e[1].record()
        q1 = pipeline.marginal_qhat(tm, alphas)                                     # n = 1024, all P planes, one launch
        e[2].record()
        pipeline.HipOps.kth_planes(tm, world * B * pitch, pitch, own, n_all, per, ks8, q8, per, len(alphas) * per)
        e[3].record()
        ev.append(e)
        return q1
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    ev.clear()
    for _ in range(steps):
        q1 = step()
    torch.cuda.synchronize()
    kms, s1, s8 = (med([e[i].elapsed_time(e[i + 1]) for e in ev]) for i in range(3))
    # CHECKERS (outside the timed region): two samples of |res| against the CPU oracle; three cells of both q-hat fields
    # against torch.sort of their columns, bit for bit
    err = oracle_parity("mhd", eq, v, out, (0, B - 1), True)
    flat = tm.as_strided((own, n_all, per), (world * B * pitch, pitch, 1))           # the n = 8192 view of the same bytes
    cells_ok = True
    for (pl, c) in ((0, 0), (own // 2, per // 2 + 1), (own - 1, per - 1)):
        col = torch.sort(flat[pl, :, c].contiguous()).values
        cells_ok = cells_ok and bool(torch.equal(q8[pl, :, c], col[ks8]))
        col1 = torch.sort(tm.as_strided((B, P, per), (pitch, B * pitch, 1))[:, pl, c].contiguous()).values
        cells_ok = cells_ok and bool(torch.equal(q1.reshape(len(alphas), P, per)[:, pl, c], col1[ks1]))
    cells = B * T * X * Y
    bpc = MHD_EQUATIONS[eq]["bpc"]
    ach = bpc * cells / (kms * 1e-3) / 1e9
    kern = MHD_EQUATIONS[eq]["kernel"] if layout == "ny" else MHD_EQUATIONS[eq]["kernel"].replace("march_kernel<", "flat_march_kernel<").replace("<0>,8,64>", "<3>>")
    return {"workload": f"C4 marginal CP as Marginal/MHD_Residuals_CP.py:323-350,408-418 runs it, per-rank job of the 8-way sharded flow: "
                        f"|residual_{eq}| of [{B},6,{T},{X},{Y}] ({'Nt' if layout == 'nt' else 'Ny'} fastest) written plane-major "
                        f"([{P}][{B}][{per}+64]), then ONE select at n = {n_all} over the {own} planes a rank owns",
            "layout": layout, "steps": steps, "warmup": warmup, "kernel": kern, "kernel_ms": kms, "achieved_gbs": ach,
            "frac": ach / HBM_PEAK_GBS, "select_n1024_all_planes_ms": s1, "select_n1024_one_read_gbs": 4.0 * cells / (s1 * 1e-3) / 1e9,
            "select_ms": s8, "select_one_read_gbs": 4.0 * n_all * own * per / (s8 * 1e-3) / 1e9,
            "ms_per_step": kms + s8, "cells_per_s_per_rank": cells / ((kms + s8) * 1e-3),
            "exchange": "not on one GPU: 7/8 of the rank's scores (15.0 GB) leave over xGMI between (1) and (3); projected in DESIGN 5",
            "parity": {"residual_rel_err": err, "tol": RES_TOL, "samples": 2, "ok": bool(err <= RES_TOL),
                       "qhat_cells_equal_sorted_columns": cells_ok}}


def measure_strong_rank_ntfast(dev, alphas, world=8, steps=5, warmup=2):
    """`c3_rank8_ntfast`: the per-rank job of the strong-scaled C3 curve ([512,64,512,512] x3, whole grid) fed the way
    Marginal/NS_Residuals_CP.py:282-287 feeds it - sur.permute(0,1,4,2,3) of a [BS,F,Nx,Ny,Nt = 64] buffer, Nt fastest: the
    library relabels its axes (flat_march_kernel<NSMomentum<3>>: Nt = 64 < 96 merges the Ny and Nt axes into one row) and
    writes the residual in the same memory order; joint CP on it where it lies (the moments / score kernels take any dense
    axis order), marginal CP on a row-padded buffer of that order."""
    from cp_pre_amd import _lib
    from cp_pre_amd import pipeline
    from cp_pre_amd.residuals import NavierStokes
    shp = CONFIGS["c3"]["shape"]
    B, T, X, Y = shp[0] // world, shp[1], shp[2], shp[3]
    group, how = one_rank_group(dev)
    v = surrogate_layout(B, 3, T, X, Y, dev)
    for i in range(3):
        synth_(v[:, i], 20 + i)
    ns = NavierStokes(1e-2, 1.0 / X, 1.0 / Y, nu=1e-3)
    cells = B * T * X * Y                                          # every row is computed (zero padding beyond the grid)
    out = {"workload": f"C3 strong-scaled over {world} ranks, per-rank job [{B},{T},{X},{Y}] x3 in the surrogate's Nt-fastest layout "
                       f"([BS,F,Nx,Ny,Nt].permute(0,1,4,2,3), Marginal/NS_Residuals_CP.py:282-287), whole grid resident, {how}",
           "group": how, "steps": steps, "warmup": warmup, "kernel": "flat_march_kernel<NSMomentum<3>>"}
    for mode in ("joint", "marginal"):
        res = _lib.empty_like_layout(v[:, 0], score_rows=(mode == "marginal"))
        ev = []

        def step():
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            ns.residual_momentum(v, boundary=True, absolute=(mode == "marginal"), out=res)
            e1.record()
            if mode == "joint":
                jc = pipeline.JointCalibration(B, dev, group=group)
                jc.add_slab(res, crop=(1, 1, 1))
                q = jc.finish(alphas)
            else:
                q = pipeline.marginal_qhat(res, alphas, group=group)
            e2.record()
            ev.append((e0, e1, e2))
            return q
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        ev.clear()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        kms = med([a.elapsed_time(b) for a, b, _ in ev])
        cms = med([b.elapsed_time(c) for _, b, c in ev])
        ach = 16.0 * cells / (kms * 1e-3) / 1e9
        out[mode] = {"ms_per_step": ms, "cells_per_s_per_rank": cells / (ms * 1e-3), "kernel_ms": kms, "calibrate_ms": cms,
                     "achieved_gbs": ach, "frac": ach / HBM_PEAK_GBS}
        if mode == "joint":                                         # CHECKER: two samples against the CPU oracle
            err = oracle_parity("ns", None, v, res, (0, B - 1), False)
            out["parity"] = {"residual_rel_err": err, "tol": RES_TOL, "samples": 2, "ok": bool(err <= RES_TOL)}
        del res
        torch.cuda.empty_cache()
    return out


def pmc_traffic(args, equation=None):
    """HBM bytes per launch of the fused residual kernel from the committed rocprofv3 PMC passes
    (profiles/rNN/pmc_hbm_<config>[_<equation>].json, written by tools/distill_profiles.py).  Counters cannot be
    read from inside this process; the file is used only when it was collected on this exact
    workload (latest round wins), otherwise `traffic` stays null."""
    import glob
    best = None
    want = {"batch": args.batch, "nt": args.nt, "nx": args.nx, "ny": args.ny}
    if args.config == "c3":
        want["slab"] = args.slab
        want["slab_axis"] = args.slab_axis
        want["rows"] = "interior"
    names = [f"pmc_hbm_{args.config}_{equation}.json"] if equation else [f"pmc_hbm_{args.config}.json"]
    for f in sorted(sum((glob.glob(os.path.join(ROOT, "profiles", "r*", n)) for n in names), [])):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        w = d.get("workload", {})
        if all(w.get(k, "t" if k == "slab_axis" else None) == v for k, v in want.items()):
            best = d
    # counters of ANOTHER kernel are not this run's traffic: the file records the hash of the kernel's source it was
    # collected on (and the commit), and is ignored when the source has changed since
    if best is not None and best.get("kernel_src_sha16") != kernel_src_sha16():
        return None
    return best


def kernel_src_sha16():
    """sha256 (first 16 hex digits) of the source of the fused residual kernels (every config's dominant kernel is a
    march_kernel of csrc/star_march.hip) - what ties a committed PMC summary to the code it was collected on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("star_march.hip", "common.h"):
        h.update(open(os.path.join(ROOT, "cp_pre_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


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


def cpu_baseline(args, alphas, slab, xs=False):
    """The oracle (reference arithmetic: F.conv3d per operator + torch elementwise + numpy
    calibration) on a bounded sample of the same workload, on this box's host cores."""
    import numpy as np
    from oracle import conformal as oc
    from oracle import residuals as orr
    # one slab with its two halo planes (t-slabs) / rows (x-slabs), as the device job streams it
    T, X = (args.nt, slab + 2) if xs else (slab + 2, args.nx)
    dt, dx, dy = 1e-2, 1.0 / args.nx, 1.0 / args.ny
    threads = host_cores()
    torch.set_num_threads(threads)

    def run(nb):
        g = torch.Generator().manual_seed(0)
        v = torch.rand(nb, 3, T, X, args.ny, generator=g) + 0.5
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
    cells = nb * slab * (args.nt if xs else args.nx) * args.ny       # the slab's own rows / planes, as in `value`
    return {"value": cells / t, "unit": "cells/s", "cores": threads, "kind": "port",
            "sample": f"oracle NS-momentum + {args.mode} calibrate on [{nb},{T},{X},{args.ny}] x3 fields "
                      f"(one slab, {nb}/{args.batch} of the batch), {t:.1f} s on {threads} torch threads"}


def init_ranks(args, world):
    """(rank, device, group, rccl_ranks).  The process group is RCCL ("nccl" on ROCm) with one rank per GPU;
    PRE_BENCH_REHEARSE=1 puts all ranks on GPU 0 over gloo - a plumbing rehearsal of the N>1 path on a
    single-GPU box (RCCL refuses two ranks per device), never a measurement."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearse = os.environ.get("PRE_BENCH_REHEARSE") == "1"
    have_gpu = torch.cuda.is_available()
    if not have_gpu and not args.plumbing_check:
        raise SystemExit("bench.py needs the MI355X (no CPU fallback)")
    dev = torch.device("cpu")
    if have_gpu:
        if rehearse:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    group, ranks = None, 1
    if world > 1 or os.environ.get("PRE_BENCH_FORCE_GROUP") == "1":     # the latter: RCCL at world size 1 (plumbing check)
        import torch.distributed as dist
        if rehearse or not have_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)     # "nccl" is RCCL on ROCm
        group = dist.group.WORLD
        # create the communicator (and its xGMI rings) now, not inside the first timed collective
        warm = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(warm)
        dist.all_gather([torch.empty_like(warm) for _ in range(dist.get_world_size())], warm)
        if have_gpu:
            torch.cuda.synchronize()
        ranks = int(warm.item())                                # every rank contributed a 1
        assert ranks == dist.get_world_size() == world, (ranks, dist.get_world_size(), world)
    return rank, dev, group, ranks


def main():
    args = parse()
    plan, what = launch_plan(args.gpus, os.environ, sys.argv[1:])
    if plan == "error":
        print(what, file=sys.stderr, flush=True)
        return 2
    if plan == "spawn":
        # a FRESH child, started before this process has touched the GPU (never exec after HIP init)
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "4")
        return subprocess.run(what, env=env).returncode
    world = what
    claim_stdout()
    if args.scaling == "strong":
        if args.batch % world:
            print(f"bench.py: --scaling strong needs --batch ({args.batch}) divisible by the rank count ({world})",
                  file=sys.stderr, flush=True)
            return 2
        args.batch //= world
    rank, dev, group, rccl_ranks = init_ranks(args, world)
    par = {"parallelism": f"batch-sharded x{world} ({args.scaling} scaling: {args.batch} samples per rank)",
           "rccl_ranks": rccl_ranks}

    def done(code=0):
        """Every rank leaves with the SAME exit code: the post-timing checks run on rank 0 only, and a launcher that sees
        one rank fail and seven succeed reports whichever it looks at."""
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            if world > 1:
                c = torch.tensor([int(code)], dtype=torch.int64, device=dev)
                torch.distributed.all_reduce(c, op=torch.distributed.ReduceOp.MAX)
                code = int(c.item())
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return code

    if args.plumbing_check:
        if rank == 0:
            emit({"plumbing_check": True, "n_gpus": world, "rccl_ranks": rccl_ranks, "scaling": args.scaling,
                  "batch_per_rank": args.batch, "backend": torch.distributed.get_backend() if group else None})
        return done()

    if args.config != "c3":
        cfg = mhd_config(args.equation) if args.config == "c4" else CONFIGS[args.config]
        if args.config == "c4" and args.layout == "nt":
            cfg = dict(cfg, layout="nt", pmc_key=args.equation + "_ntfast",
                       kernel=cfg["kernel"].replace("march_kernel<", "flat_march_kernel<").replace("<0>,8,64>", "<3>>"),
                       title=cfg["title"] + ", fields in the surrogate's Nt-fastest layout [BS,F,Nx,Ny,Nt].permute(0,1,4,2,3) (:326-346)")
        line = run_secondary(args, cfg, dev, group, rank, world, par)
        if line is not None:
            emit(line)
        return done(3 if (line is not None and not line.get("parity", {}).get("ok", True)) else 0)

    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd import pipeline

    B, T, X, Y = args.batch, args.nt, args.nx, args.ny
    xs = args.slab_axis == "x"
    # The grid does not fit the HBM next to its residual (C3: 3 x 275 GB of fields), so the job streams SLABS of it, all
    # samples of a slab at a time (a per-cell std / quantile needs every sample of its cell): see C3Stream.
    n_axis, other = (X - 2, T * Y) if xs else (T, X * Y)
    free = torch.cuda.mem_get_info(dev)[0]
    if not args.slab:
        # fewer, thicker slabs re-read fewer halo rows / planes; the resident set must fit the free HBM (t-slabs of 16
        # planes at the full batch: 301 GB of the 309 GB the device reports; x-slabs of 128 rows: 279 GB).
        # Smaller per-rank batches (--scaling strong, --batch) afford thicker slabs, up to the whole axis (no halo).
        # sharded marginal CP: the exchange receives ONE plane of all ranks' samples at a time (no send staging: the
        # residual is written time-major, pipeline.time_major)
        per_plane = (X * Y if not xs else (n_axis // 4 + 2) * Y)
        extra = 4 * world * B * per_plane if (args.mode == "marginal" and world > 1) else 0
        extra += (1 << 30) if group is not None else 0                 # headroom for RCCL's own scratch beyond the warm-up collective
        if xs:      # the interior rows in 1, 2, 4, ... runs (510 rows: 510, 255, 128, 64, ...), down to 16 rows
            div = [-(-n_axis // d) for d in (1, 2, 4, 8, 16, 32, 64) if -(-n_axis // d) >= 16] or [n_axis]
        else:
            div = [n_axis, (n_axis + 1) // 2, (n_axis + 2) // 3, (n_axis + 3) // 4, 16, 13, 8]
        cands = sorted(set(c for c in div if 0 < c <= n_axis), reverse=True)
        idx = next((i for i, c in enumerate(cands) if resident_bytes(B, n_axis, c, other) + extra <= free - (4 << 30)),
                   len(cands) - 1)
        if group is not None:
            # the ranks must stream the SAME slabs (the per-slab collectives carry one slab's cells): free memory differs
            # a little from GPU to GPU, so take the most conservative choice of any rank
            t = torch.tensor([idx], dtype=torch.int64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=group)
            idx = int(t.item())
        args.slab = cands[idx]
    alphas = [float(a) for a in icp.ALPHA_LEVELS]

    torch.manual_seed(1234 + rank)
    need = resident_bytes(B, n_axis, args.slab, other)
    short = need > free - (2 << 30)
    if short:
        print(f"bench.py: --slab {args.slab} needs {need / 1e9:.0f} GB resident, {free / 1e9:.0f} GB free", file=sys.stderr, flush=True)
    if group is not None:                                    # every rank leaves, or none (a lone exit would hang the others)
        t = torch.tensor([int(short)], dtype=torch.int64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=group)
        short = bool(t.item())
    if short:
        return 2
    st = C3Stream(B, T, X, Y, args.slab, args.slab_axis, dev, rank=rank)
    slabs, n_slabs, S, crop = st.slabs, len(st.slabs), st.S, st.crop

    ev_used = []
    last_jc = [None]

    def step(k, res_of, prune=not args.no_prune, mode=args.mode):
        """One whole job: every slab position evaluated into its view of the residual buffer (`res_of`: passed by every
        caller, never captured - the buffer must be freeable) and calibrated."""
        jc = pipeline.JointCalibration(B, dev, group=group, prune=prune) if mode == "joint" else None
        last_jc[0] = jc
        q = None
        for s, sl in enumerate(slabs):
            res = res_of[sl]
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            st.eval_slab(s, sl, res, absolute=(mode == "marginal"))
            e1.record()
            if jc is not None:
                jc.add_slab(res, crop=crop)
            else:
                q = pipeline.marginal_qhat(res, alphas, group=group)     # [10, *slab cells]
            e2.record()
            ev_used.append((k, sl, e0, e1, e2))
        return jc.finish(alphas) if jc is not None else q

    def sync():
        torch.cuda.synchronize()
        if group is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    res_main = st.views(args.mode, sharded=group is not None)
    note(f"c3 {args.mode}: {len(slabs)} slabs of {slabs}, {args.warmup} + {args.steps} steps")
    for k in range(args.warmup):
        step(k, res_main)
    sync()
    t0 = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.steps):
        qhat = step(k, res_main)
    sync()
    elapsed = time.perf_counter() - t0
    if group is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())

    cells_per_step = B * sum(slabs) * other * world                   # the cells the launches compute, all ranks
    value = cells_per_step * args.steps / elapsed

    # The branch-and-bound score pass makes `value` depend on the data (how much of the residual the bounds let it
    # skip).  Say by how much: the share of the pass that was read, and the same step with the full pass - one more
    # step, timed apart, not part of `value`.
    read_frac = ms_full = qhat_full = None
    if args.mode == "joint" and not args.no_prune:
        read_frac = last_jc[0].score_pass_read_frac()
        step(args.warmup + args.steps, res_main, prune=False)         # (untimed: first use of the full-pass kernels)
        sync()
        t1 = time.perf_counter()
        qhat_full = step(args.warmup + args.steps + 1, res_main, prune=False)
        sync()
        ms_full = 1e3 * (time.perf_counter() - t1)
        if group is not None:
            tt = torch.tensor([ms_full], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            ms_full = float(tt.item())

    code = 0
    if rank == 0:
        timed = [(sl, e0.elapsed_time(e1)) for (k, sl, e0, e1, _) in ev_used if args.warmup <= k < args.warmup + args.steps]     # ms, this rank
        kms = sum(d for _, d in timed) / len(timed)
        # 3 fields read + 1 residual written = 16 B per cell the launch COMPUTES (its own rows / interior
        # planes); the two halo rows / planes each slab re-reads are overhead, reported apart
        launch_bytes = sum(16 * B * sl * other for sl, _ in timed) / len(timed)
        halo_bytes = sum((12 * (sl + 2) + 4 * sl) * B * other for sl, _ in timed) / len(timed)
        achieved = launch_bytes / (kms * 1e-3) / 1e9
        pmc = pmc_traffic(args)
        out = {
            "metric": "residual-cells/s (PRE eval+calibrate)",
            "value": value, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32",
            "data": SYNTH + f"; one resident {'x' if xs else 't'}-slab, slab position s reads the batch window [s, s+B) of it",
            "config": {"workload": f"C3 2D Navier-Stokes momentum residual [{B},{args.nt},{X},{Y}] x3 fields per rank, "
                                   f"{args.mode} CP, 10 alpha levels; streamed as " +
                                   (f"{n_slabs} x-slabs of {slabs} rows = the reference's interior rows 1..{X - 2} "
                                    f"(Marginal/NS_Residuals_CP.py:240 crops rows 0 and {X - 1}: not computed), +2 halo rows each, T whole"
                                    if xs else f"{n_slabs} t-slabs of {slabs} interior planes (+2 halo planes each)"),
                       "mode": args.mode, "batch_per_rank": B, "slab_axis": args.slab_axis, "slab": args.slab, **par,
                       "cells_per_step": cells_per_step, "uncropped_grid_cells": B * T * X * Y * world,
                       # the per-rank slab plan: resident bytes, and what the slab halo re-reads cost (strong scaling
                       # shrinks the per-rank batch, which affords thicker slabs - up to the whole axis, no halo)
                       "slab_plan": {"axis": args.slab_axis, "slabs": slabs,
                                     "resident_gb": round(resident_bytes(B, n_axis, args.slab, other) / 1e9, 1),
                                     "input_read_per_cell_computed": round(sum(sl + 2 for sl in slabs) / sum(slabs), 4)},
                       "scaling_note": (f"weak: every rank streams its own {B}-sample batch" if args.scaling == "weak" else
                                        "strong: one calibration set split over the ranks (north_star's '>= 6x 1->8 GPUs' "
                                        "speaks of this curve)"),
                       # `value` is measured with the fields resident in HBM (contract); if the 3 fields came from host
                       # memory instead, PCIe Gen5 x16 (63 GB/s spec) would bound the job at 63e9 / 12 B per cell
                       "inputs": "resident in HBM", "host_fed_bound_cells_per_s": 63e9 / 12.0,
                       **({"score_pass": "branch-and-bound (same scores as the full pass for the same modulation; adaptive: "
                                         "flagged samples and wasteful streams take the full pass)"
                           if not args.no_prune and pipeline.HipOps.can_prune(res_main[slabs[0]], crop) else "full"}
                          if args.mode == "joint" else {})},
            # data dependence of `value`: share of the score pass's segments that were read (the synthetic residuals are
            # noise-like: tight bounds), and the same step with the full score pass (--no-prune)
            "score_pass_read_frac": read_frac, "ms_per_step_full_score_pass": ms_full,
            "roofline": {"bound": "hbm", "kernel": "march_kernel<NSMomentum<0>,8,64>",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc["traffic_bytes_per_launch"] if pmc else None,
                         "traffic_source": pmc["source"] if pmc else None,
                         "avg_launch_ms": kms, "algorithmic_bytes_per_launch": launch_bytes,
                         "bytes_incl_slab_halo": halo_bytes,
                         "launches_timed": len(timed)},
            "qhat_first_last": [float(qhat.reshape(len(alphas), -1)[0, 0]), float(qhat.reshape(len(alphas), -1)[-1, 0])],
        }
        if args.mode == "marginal":
            sel = [e1.elapsed_time(e2) for (k, sl, _, e1, e2) in ev_used if args.warmup <= k < args.warmup + args.steps]
            out["select_ms_per_slab"] = sum(sel) / len(sel)
            out["select_one_read_gbs"] = 4.0 * B * sum(slabs) / n_slabs * other / (out["select_ms_per_slab"] * 1e-3) / 1e9
        if not args.no_parity:
            # CHECKER (outside every timed region): what the last slab pass left in the residual buffer against the CPU
            # oracle on two samples (first and last of the batch window), and the pruned q-hats against the full pass's
            note("parity check against the CPU oracle (2 samples of the last slab)")
            s_last, sl_last = n_slabs - 1, slabs[-1]
            res_last = res_main[sl_last]
            if args.mode == "marginal":                                  # |res| is in the buffer: re-evaluate signed
                st.eval_slab(s_last, sl_last, res_last, absolute=False)
                torch.cuda.synchronize()
            err = c3_parity(st, res_last, s_last, sl_last, (0, B - 1))
            out["parity"] = {"residual_rel_err": err, "tol": RES_TOL, "samples": 2, "slab": [s_last, sl_last],
                             "oracle": "oracle/residuals.py::ns_momentum (F.conv3d per operator, Marginal/NS_Residuals_CP.py:231-240) "
                                       "on the same rows with their halo rows, host"}
            if qhat_full is not None:
                rel = float(((qhat - qhat_full).abs() / qhat_full.abs()).max())
                out["parity"]["qhat_pruned_vs_full_pass_rel"] = rel
                out["parity"]["qhat_tol"] = 1e-6
                if not rel <= 1e-6:
                    code = 3
            if not err <= RES_TOL:
                code = 3
            del res_last                                                 # (a view: it would keep the 69 GB buffer alive)
        if world == 1 and args.mode == "joint" and group is None and not args.no_secondary and not args.no_prune:
            # every other BASELINE config under the same clock: C3 marginal on the slab that is already resident, then
            # (C3's buffers freed) the strong-scaled per-rank job, C1, C2, C4 x5, the C5 shard and C5 whole.  Not part
            # of `value`.
            sec = {}
            note("secondary c3_marginal")
            mres, k0, n2 = st.views("marginal"), args.warmup + args.steps + 2, 2 + 5
            for k in range(k0, k0 + 2):
                step(k, mres, mode="marginal")
            sync()
            t2 = time.perf_counter()
            for k in range(k0 + 2, k0 + n2):
                qm = step(k, mres, mode="marginal")
            sync()
            ms = 1e3 * (time.perf_counter() - t2) / 5
            # CHECKER (outside the timed region): the last slab's |res| is still in the buffer - three cells of its q-hat
            # field against torch.sort of their 4096-sample columns, bit for bit
            a_last, ks_m = mres[slabs[-1]], [icp.kth_index(B, B, a) for a in alphas]
            cells_ok = True
            for (t, x, y) in ((0, 0, 0), (a_last.shape[1] // 2, a_last.shape[2] // 2, Y // 2 + 1), (a_last.shape[1] - 1, a_last.shape[2] - 1, Y - 1)):
                col = torch.sort(a_last[:, t, x, y].contiguous()).values
                cells_ok = cells_ok and bool(torch.equal(qm[:, t, x, y], col[ks_m]))
            err_m = c3_parity(st, a_last, n_slabs - 1, slabs[-1], (0, B - 1), absolute=True)      # |res| of the last slab vs |oracle|
            if not (cells_ok and err_m <= RES_TOL):
                code = 3
            del a_last, qm
            mt = [(sl, e0.elapsed_time(e1), e1.elapsed_time(e2)) for (k, sl, e0, e1, e2) in ev_used if k >= k0 + 2]
            kms_m = med([d for _, d, _ in mt])
            sel_m = med([d for _, _, d in mt])
            ach = sum(16 * B * sl * other for sl, _, _ in mt) / len(mt) / (kms_m * 1e-3) / 1e9
            sec["c3_marginal"] = {
                "workload": out["config"]["workload"].replace("joint CP", "marginal CP (per-cell q-hat over the 4096 samples)"),
                "ms_per_step": ms, "cells_per_s": cells_per_step / (ms * 1e-3), "steps": 5, "warmup": 2,
                "kernel": "march_kernel<NSMomentum<0>,8,64>", "kernel_ms": kms_m, "achieved_gbs": ach, "frac": ach / HBM_PEAK_GBS,
                "traffic": (lambda pm: pm["traffic_bytes_per_launch"] if pm else None)(pmc_traffic(args, "marginal")),
                "select_ms_per_slab": sel_m,
                "select_one_read_gbs": 4.0 * B * sum(slabs) / n_slabs * other / (sel_m * 1e-3) / 1e9,
                "parity": {"residual_rel_err": err_m, "tol": RES_TOL, "samples": 2, "ok": bool(err_m <= RES_TOL),
                           "qhat_cells_equal_sorted_columns": cells_ok}}
            del mres
        del res_main
        ev_used.clear()
        last_jc[0] = None
        st.free()
        held = torch.cuda.memory_allocated(dev)                   # C3's 278 GB must be gone before anything else is allocated
        if world == 1 and args.mode == "joint" and group is None and not args.no_secondary and not args.no_prune:
            if held >= (8 << 30):
                sec["error"] = f"C3's buffers were not released ({held / 1e9:.1f} GB still allocated): the other configs were not measured"
                out["secondary"] = sec
        if world == 1 and args.mode == "joint" and group is None and not args.no_secondary and not args.no_prune and held < (8 << 30):
            try:
                sec["c3_strong_rank8"] = measure_strong_rank(dev, alphas)
            except Exception as e:                                # (a secondary must never cost the contract line)
                sec["c3_strong_rank8"] = {"error": f"{type(e).__name__}: {e}"[:300]}
                torch.cuda.empty_cache()
            for key, fn in (("c3_rank8_ntfast", lambda: measure_strong_rank_ntfast(dev, alphas)),
                            ("c4_marginal_rank8", lambda: measure_c4_marginal(dev, alphas, "ny")),
                            ("c4_marginal_rank8_ntfast", lambda: measure_c4_marginal(dev, alphas, "nt"))):
                note(f"secondary {key}")
                try:
                    sec[key] = fn()
                except Exception as e:
                    sec[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
                torch.cuda.empty_cache()
            sec.update(measure_others(dev))
            for gname in ("c1", "c2", "c5"):                      # (last: a capture that fails must not cost the others)
                try:
                    note(f"secondary {gname}_graph")
                    sec[gname + "_graph"] = measure_graph(dev, gname)
                    if not sec[gname + "_graph"]["replay_equals_eager"]:
                        note(f"GRAPH REPLAY DIFFERS from the eager step in {gname}_graph")
                        code = 3
                except Exception as e:
                    sec[gname + "_graph"] = {"error": f"{type(e).__name__}: {e}"[:300]}
                torch.cuda.empty_cache()
            out["secondary"] = sec
            # a secondary whose residual missed the oracle (or whose q-hat cells missed their sorted columns) fails the run
            for key, e in sec.items():
                par_ = e.get("parity") if isinstance(e, dict) else None
                if par_ and (not par_.get("ok", True) or par_.get("qhat_cells_equal_sorted_columns") is False):
                    note(f"PARITY FAILED in secondary {key}: {par_}")
                    code = 3
        if world == 1 and not args.no_cpu_baseline:
            note("cpu baseline (the oracle on the host cores)")
            out["cpu_baseline"] = cpu_baseline(args, alphas, S, xs)
        # LAST key: every config's [ms per step, roofline fraction of its residual kernel, oracle parity] in under 1 KB, so
        # that a record which keeps only the tail of the line still shows all of them
        out["summary"] = summary_of(out)
        note("done")
        emit(out)
    return done(code)


if __name__ == "__main__":
    sys.exit(main())

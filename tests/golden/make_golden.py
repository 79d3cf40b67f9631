#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

Run once, here, with /root/reference present:  python tests/golden/make_golden.py
The GPU box never sees /root/reference; only the .npz files below travel.

What is executed from the reference (nothing of it is copied into this repo):
  * ``Utils/ConvOps_2d.py`` / ``Utils/ConvOps_1d.py``  - imported; every constructor
    combination -> kernels.npz; ``D(x)`` on seeded inputs -> apply.npz
  * ``Other_UQ/Evaluation/PRE_estimations.py``          - imported; PRE_Wave / PRE_NS /
    PRE_MHD ``.residual`` -> residuals.npz
  * the residual *function definitions* of ``Marginal/NS_Residuals_CP.py``,
    ``Marginal/MHD_Residuals_CP.py``, ``Joint/Burgers_Residuals_CP.py`` and the additive
    kernel assignment of ``Marginal/Advection_Residuals_CP.py``: those scripts cannot be
    imported (top-level ``from Neural_PDE...``), so the defs are located with ``ast`` at
    generation time, compiled from the reference file itself and executed against the
    reference's own ConvOperator instances -> residuals.npz
  * the reduced-MHD residuals of ``Marginal/JOREK_residuals_CP.py`` (:188-247): the operator constructions, the
    coefficient tensors and the ``residual_continuity`` / ``residual_temperature`` defs, compiled from the file and
    executed on seeded fields and a seeded R grid -> jorek.npz
  * ``filter_sims_joint`` (Joint/Burgers_Residuals_CP.py:298-300) and ``filter_sims_within_bounds``
    (Active_Learning/Advection_AL_Marginal.py:169-198) - defined inside the reference tree -
    compiled from it the same way -> filters.npz
  * ``Utils/ConvOps_Spatial.py``, ``Utils/boundary_conditions.py``, ``Utils/VectorConvOps_Spatial.py``
    - imported on CPU -> spatial.npz
  * the inline joint-CP recipe of ``Tests/test_advection_inv_sampling_marginal.py`` - the only place the
    reference tree itself states what modulation / joint score / joint coverage compute: the
    ``modulation = np.std(...)`` assignment (:428), the ``conf_metric_joint`` def (:430-431) and the
    prediction-set + coverage statements of the alpha loop (:464-465), compiled from that file with
    ``ast`` and executed on seeded arrays -> conformal_ref.npz (pins a12, a13 and joint coverage)
  * conformal.npz is BUILD-DEFINED (numpy, oracle/conformal.py) and now matters for ``calibrate``
    only: its source, ``Neural_PDE.UQ.inductive_cp``, is absent and nothing in the reference tree
    restates it, so q-hat's rank convention stays "parity unpinned".
"""
import ast
import importlib.util
import os
import sys
import warnings

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("MPLBACKEND", "Agg")
warnings.filterwarnings("ignore")
sys.path[:0] = [REF, os.path.join(REF, "Utils")]

from Utils.ConvOps_2d import ConvOperator as Ref2D   # noqa: E402
from Utils.ConvOps_1d import ConvOperator as Ref1D   # noqa: E402

_spec = importlib.util.spec_from_file_location(
    "PRE_estimations", os.path.join(REF, "Other_UQ/Evaluation/PRE_estimations.py"))
pre = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(pre)

DOMAINS_2D = ["t", "x", "y", ("x", "y"), ("x", "y", "t"), "z", None]
DOMAINS_1D = ["t", "x", ("x", "t"), "y", None]
ORDERS = [0, 1, 2, 3]
TAYLORS = [2, 4, 6]
SCALES = [1.0, 0.37]


def dom_tag(d):
    return "none" if d is None else ("".join(d) if isinstance(d, tuple) else d)


def gen_kernels():
    out = {}
    for tag, cls, doms in (("2d", Ref2D, DOMAINS_2D), ("1d", Ref1D, DOMAINS_1D)):
        for d in doms:
            for o in ORDERS:
                for ty in TAYLORS:
                    for s in SCALES:
                        key = f"{tag}|{dom_tag(d)}|{o}|{ty}|{s}"
                        op = cls(d, o, scale=s, taylor_order=ty)
                        if hasattr(op, "kernel"):
                            out[key] = op.kernel.numpy()
                        else:
                            out[key] = np.zeros((0,), np.float32)     # "no kernel"
    np.savez_compressed(os.path.join(HERE, "kernels.npz"), **out)
    return out


def gen_apply(kernels):
    out = {}
    torch.manual_seed(0)
    x4 = {"a": torch.randn(3, 7, 9, 11), "b": torch.randn(1, 1, 16, 16), "c": torch.randn(2, 5, 4, 8)}
    x3 = {"a": torch.randn(2, 5, 8), "b": torch.randn(3, 1, 12), "c": torch.randn(1, 9, 7)}
    for k, v in x4.items():
        out[f"in4|{k}"] = v.numpy()
    for k, v in x3.items():
        out[f"in3|{k}"] = v.numpy()
    for key, kern in kernels.items():
        if kern.size == 0 or not key.endswith("|1.0"):
            continue
        tag = key.split("|")[0]
        cls, xs = (Ref2D, x4) if tag == "2d" else (Ref1D, x3)
        op = cls()
        op.kernel = torch.from_numpy(kern)
        for name, x in xs.items():
            out[f"out|{key}|{name}"] = op(x).numpy()
    # additive composites (README.md:47-54)
    w = pre.PRE_Wave(dt=0.01, dx=0.02, c=1.0)
    out["kern|wave"] = w.D.kernel.numpy()
    for name, x in x4.items():
        out[f"out|wave|{name}"] = w.D(x).numpy()
    np.savez_compressed(os.path.join(HERE, "apply.npz"), **out)


def ref_defs(relpath, names):
    """Compile the named top-level function defs straight from a reference script."""
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path).read(), path)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in body} == set(names), (relpath, [n.name for n in body])
    return compile(ast.Module(body=body, type_ignores=[]), path, "exec")


def ref_assign_value(relpath, target_src):
    """Compile the RHS of the first top-level assignment to ``target_src``."""
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path).read(), path)
    for n in tree.body:
        if isinstance(n, ast.Assign) and ast.unparse(n.targets[0]) == target_src:
            return compile(ast.Expression(n.value), path, "eval")
    raise KeyError(target_src)


def ref_loop_body(relpath, must_contain, skip_targets=()):
    """Compile the body of the first top-level ``for`` loop whose source contains ``must_contain``,
    leaving out assignments to ``skip_targets`` (names the caller supplies instead)."""
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path).read(), path)
    for n in tree.body:
        if isinstance(n, ast.For) and must_contain in ast.unparse(n):
            body = [b for b in n.body
                    if not (isinstance(b, ast.Assign) and ast.unparse(b.targets[0]) in skip_targets)]
            return compile(ast.Module(body=body, type_ignores=[]), path, "exec"), [ast.unparse(b) for b in body]
    raise KeyError(must_contain)


def ops2d():
    return dict(D_t=Ref2D(domain="t", order=1), D_x=Ref2D(domain="x", order=1),
                D_y=Ref2D(domain="y", order=1), D_xx_yy=Ref2D(domain=("x", "y"), order=2))


def gen_residuals():
    out = {}
    g = torch.Generator().manual_seed(1)
    v6 = torch.rand(4, 6, 6, 10, 12, generator=g) + 0.5          # U(0.5,1.5): rho, p > 0
    u1 = torch.rand(5, 9, 14, generator=g) + 0.5
    out["vars6"], out["u1d"] = v6.numpy(), u1.numpy()
    dt, dx, dy = 0.01, 1.0 / 64, 1.0 / 48
    out["coef"] = np.array([dt, dx, dy], np.float64)

    # packaged classes (imported)
    for b in (False, True):
        out[f"PRE_Wave|{int(b)}"] = pre.PRE_Wave(dt=0.01, dx=0.02, c=1.0).residual(v6[:, :1], boundary=b).contiguous().numpy()
        out[f"PRE_NS|{int(b)}"] = pre.PRE_NS(dt, dx, dy).residual(v6[:, :3], boundary=b).contiguous().numpy()
        out[f"PRE_MHD|{int(b)}"] = pre.PRE_MHD(dt, dx, dy).residual(v6, boundary=b).contiguous().numpy()

    # script-level defs, compiled from the reference files
    ns = dict(ops2d(), dx=dx, dy=dy, dt=dt, nu=0.001)
    exec(ref_defs("Marginal/NS_Residuals_CP.py",
                  ["residual_continuity", "residual_momentum", "periodic_bc_residual"]), ns)
    for b in (False, True):
        out[f"ns_continuity|{int(b)}"] = ns["residual_continuity"](v6[:, :2], boundary=b).contiguous().numpy()
        out[f"ns_momentum|{int(b)}"] = ns["residual_momentum"](v6[:, :3], boundary=b).contiguous().numpy()
    for wall in ("top", "bottom", "left", "right"):
        out[f"ns_periodic_bc|{wall}"] = ns["periodic_bc_residual"](v6[:, 0], wall=wall).contiguous().numpy()

    mhd = dict(ops2d(), gamma=5 / 3)
    names = ["residual_continuity", "residual_momentum", "residual_energy", "residual_induction", "residual_gauss"]
    exec(ref_defs("Marginal/MHD_Residuals_CP.py", names), mhd)
    for n in names:
        for b in (False, True):
            out[f"mhd_{n.split('_')[1]}|{int(b)}"] = mhd[n](v6, boundary=b).contiguous().numpy()

    bdx, bdt, bnu = 2.0 / 14, 1.25 / 9, 0.002
    out["burgers_coef"] = np.array([bdx, bdt, bnu], np.float64)
    bur = dict(D_t=Ref1D(domain="t", order=1), D_x=Ref1D(domain="x", order=1), D_xx=Ref1D(domain="x", order=2),
               dx=torch.tensor(bdx, dtype=torch.float32), dt=torch.tensor(bdt, dtype=torch.float32),
               nu=torch.tensor(bnu, dtype=torch.float32))
    exec(ref_defs("Joint/Burgers_Residuals_CP.py", ["residual"]), bur)
    for b in (False, True):
        out[f"burgers|{int(b)}"] = bur["residual"](u1, boundary=b).contiguous().numpy()

    adv = dict(D_t=Ref1D(domain="t", order=1), D_x=Ref1D(domain="x", order=1), v=1.0, disc=2, dt=0.005, dx=0.01)
    kadv = eval(ref_assign_value("Marginal/Advection_Residuals_CP.py", "D.kernel"), adv)
    D = Ref1D()
    D.kernel = kadv
    out["advection_kernel"] = kadv.numpy()
    out["advection|1"] = D(u1).numpy()
    out["advection|0"] = D(u1)[..., 1:-1, 1:-1].contiguous().numpy()
    np.savez_compressed(os.path.join(HERE, "residuals.npz"), **out)


def gen_jorek():
    """Reduced-MHD (JOREK) residuals, ``Marginal/JOREK_residuals_CP.py:188-247`` (twin: ``Joint/JOREK_residuals_CP.py``).
    Executed from the reference file: the operator constructions ``D_t .. D_ZZ = ConvOperator(...)`` (:201-205, with
    the globals they see at that point of the script: alpha = beta = 1 (:192) and gamma ALREADY re-bound to the
    adiabatic index tensor 5/3 (:199), which is therefore the ``scale`` of D_RR / D_ZZ), the coefficient tensors D, K,
    gamma (:196-199) and the defs ``unstack_fields``, ``residual_continuity``, ``residual_temperature`` - on seeded
    fields and a seeded R grid.  R is a 1-D tensor (``x_grid``) that the expressions broadcast against [BS,Nt,Nx,Ny]
    fields, i.e. along the LAST axis: the fixture grid is square like the reference's data."""
    rel = "Marginal/JOREK_residuals_CP.py"
    out = {}
    g = torch.Generator().manual_seed(11)
    # vars as the script holds them: [BS, F, Nx, Ny, Nt] (the surrogate's layout); unstack_fields permutes every field
    # to [BS, Nt, Nx, Ny] (:84-95), so the operators see Nt-fastest views
    v3 = torch.rand(4, 3, 12, 12, 6, generator=g) + 0.5                     # rho, phi, T in U(0.5, 1.5)
    R = torch.linspace(1.2, 2.3, 12) + 0.01 * torch.rand(12, generator=g)   # major radius grid, > 0
    dx = torch.tensor(0.1, dtype=torch.float32)
    dt = torch.tensor(0.02, dtype=torch.float32)
    env = dict(ConvOperator=Ref2D, torch=torch, np=np, alpha=1, beta=1)
    for name in ("D", "mu", "K", "gamma"):                                  # :196-199 (gamma: the LAST top-level binding)
        path = os.path.join(REF, rel)
        tree = ast.parse(open(path).read(), path)
        val = [n for n in tree.body if isinstance(n, ast.Assign) and ast.unparse(n.targets[0]) == name][-1].value
        env[name] = eval(compile(ast.Expression(val), path, "eval"), env)
    for name in ("D_t", "D_R", "D_Z", "D_RR", "D_ZZ"):                      # :201-205
        env[name] = eval(ref_assign_value(rel, name), env)
        out[f"kernel|{name}"] = env[name].kernel.numpy()
    env.update(R=R, dx=dx, dy=dx, dt=dt, field=["rho", "phi", "T"])
    exec(ref_defs(rel, ["unstack_fields", "residual_continuity", "residual_temperature"]), env)
    out["vars3"], out["R"] = v3.numpy(), R.numpy()
    out["coef"] = np.array([float(dx), float(dx), float(dt), float(env["D"]), float(env["K"]), float(env["gamma"])], np.float64)
    for b in (False, True):
        out[f"continuity|{int(b)}"] = env["residual_continuity"](v3, boundary=b).contiguous().numpy()
        out[f"continuity_norms|{int(b)}"] = env["residual_continuity"](v3, boundary=b, norms=True).contiguous().numpy()
        out[f"temperature|{int(b)}"] = env["residual_temperature"](v3, boundary=b).contiguous().numpy()
    np.savez_compressed(os.path.join(HERE, "jorek.npz"), **out)


def gen_conformal():
    """BUILD-DEFINED vectors (numpy): not reference-derived."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import conformal as oc
    out = {"_note": np.array("build-defined, not reference-derived (Neural_PDE.UQ.inductive_cp absent); authoritative for "
                             "calibrate only - modulation / joint score / joint coverage are pinned by conformal_ref.npz")}
    rng = np.random.default_rng(7)
    for n in (7, 100, 256):
        s = np.abs(rng.standard_normal((n, 5, 6))).astype(np.float32)
        s[: n // 3, 0, 0] = s[0, 0, 0]                       # ties
        r = rng.standard_normal((n, 5, 6)).astype(np.float32)
        out[f"scores|{n}"], out[f"res|{n}"] = s, r
        mod = oc.modulation_func(r, np.zeros_like(r))
        out[f"mod|{n}"] = mod
        js = oc.ncf_metric_joint(r, np.zeros_like(r), mod)
        out[f"jscore|{n}"] = js
        for i, a in enumerate(oc.ALPHA_LEVELS):
            try:
                out[f"k|{n}|{i}"] = np.array(oc.kth_index(n, a))
                out[f"qhat|{n}|{i}"] = oc.calibrate(s, n, a)
                qj = oc.calibrate(js, n, a)
                out[f"qhat_joint|{n}|{i}"] = qj
                out[f"cov|{n}|{i}"] = np.array(oc.emp_cov([-out[f"qhat|{n}|{i}"], out[f"qhat|{n}|{i}"]], r))
                out[f"cov_joint|{n}|{i}"] = np.array(oc.emp_cov_joint([-qj * mod, qj * mod], r))
            except ValueError:
                out[f"k|{n}|{i}"] = np.array(-1)             # level > 1: numpy raises
    np.savez_compressed(os.path.join(HERE, "conformal.npz"), **out)


def gen_conformal_ref():
    """REFERENCE-EXECUTED vectors for modulation / joint score / joint prediction sets / joint coverage.

    ``Tests/test_advection_inv_sampling_marginal.py`` cannot be imported (top-level ``from Neural_PDE...``,
    data and weights absent), so its statements are compiled one by one from the file and run on seeded
    [n, Nt, Nx] residual arrays.  ``calibrate`` is absent from the reference, so q-hat is SUPPLIED (fixed
    values and, per alpha, the build's numpy q-hat): what is pinned is everything around it."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import conformal as oc
    rel = "Tests/test_advection_inv_sampling_marginal.py"
    mod_rhs = ref_assign_value(rel, "modulation")                        # :428 (first top-level assignment)
    score_def = ref_defs(rel, ["conf_metric_joint"])                     # :430-431
    loop, stmts = ref_loop_body(rel, ".all(axis", skip_targets=("qhat",))   # :464-465
    assert len(stmts) == 2 and stmts[0].startswith("prediction_sets") and "emp_cov_res.append" in stmts[1], stmts
    out = {"_note": np.array("executed from the reference: Tests/test_advection_inv_sampling_marginal.py:428,430-431,464-465; "
                             "q-hat supplied (calibrate is absent from the reference)")}
    rng = np.random.default_rng(23)
    for n in (7, 100, 256):
        shape = (n, 9, 14)
        cal = rng.standard_normal(shape).astype(np.float32) * np.linspace(0.5, 2.0, 14, dtype=np.float32)
        pred = rng.standard_normal((31,) + shape[1:]).astype(np.float32) * 0.05
        val = (pred + 0.8 * rng.standard_normal(pred.shape) * np.linspace(0.5, 2.0, 14)).astype(np.float32)
        ns = {"np": np, "cal_residual": torch.from_numpy(cal), "pred_residual": torch.from_numpy(pred),
              "val_residual": torch.from_numpy(val)}
        ns["modulation"] = eval(mod_rhs, ns)
        exec(score_def, ns)
        scores = ns["conf_metric_joint"](ns["cal_residual"][:, 1:-1, 1:-1].numpy())
        out[f"cal|{n}"], out[f"pred|{n}"], out[f"val|{n}"] = cal, pred, val
        out[f"mod|{n}"], out[f"jscore|{n}"] = ns["modulation"], scores
        qs = [1.0, 2.5, 4.0]
        for a in oc.ALPHA_LEVELS:
            try:
                qs.append(float(oc.calibrate(scores, n, a)))
            except ValueError:
                pass
        out[f"qhats|{n}"] = np.array(qs, np.float64)
        covs = []
        for i, q in enumerate(qs):
            ns["qhat"], ns["emp_cov_res"] = np.float32(q), []
            exec(loop, ns)
            covs.append(ns["emp_cov_res"][0])
            if i < 3:
                out[f"lo|{n}|{i}"], out[f"hi|{n}|{i}"] = ns["prediction_sets"]
        out[f"cov_joint|{n}"] = np.array(covs, np.float64)
    np.savez_compressed(os.path.join(HERE, "conformal_ref.npz"), **out)


def gen_filters():
    """The two coverage filters that ARE defined inside the reference tree, compiled from it:
    ``filter_sims_joint`` (Joint/Burgers_Residuals_CP.py:298-300) and
    ``filter_sims_within_bounds`` (Active_Learning/Advection_AL_Marginal.py:169-198)."""
    ns = {"np": np}
    exec(ref_defs("Joint/Burgers_Residuals_CP.py", ["filter_sims_joint"]), ns)
    exec(ref_defs("Active_Learning/Advection_AL_Marginal.py", ["filter_sims_within_bounds"]), ns)
    rng = np.random.default_rng(11)
    y = rng.standard_normal((40, 9, 13)).astype(np.float32)
    q = np.abs(rng.standard_normal((9, 13))).astype(np.float32) + 0.5
    y[3] = np.clip(y[3], -q, q)                       # one sample fully inside, edges exactly on the bound
    out = {"y": y, "q": q}
    for scale in (1.0, 2.0, 3.5):
        out[f"joint|{scale}"] = ns["filter_sims_joint"]([-scale * q, scale * q], y)
        for thr in (0.25, 0.5, 0.9):
            for within in (False, True):
                out[f"within|{scale}|{thr}|{int(within)}"] = ns["filter_sims_within_bounds"](-scale * q, scale * q, y, thr, within=within)
    np.savez_compressed(os.path.join(HERE, "filters.npz"), **out)


def gen_spatial():
    """Utils/ConvOps_Spatial.py, Utils/boundary_conditions.py, Utils/VectorConvOps_Spatial.py imported on
    CPU.  Gradient / Vector_Gradient hard-code device='cuda' for their sub-operators and cannot be
    built here; their pieces (padding + the two first-derivative operators) are all dumped."""
    import importlib
    sp = importlib.import_module("ConvOps_Spatial")
    vs = importlib.import_module("VectorConvOps_Spatial")
    bcm = importlib.import_module("boundary_conditions")
    out = {}
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 1, 12, 16, generator=g)
    y = torch.randn(3, 1, 12, 16, generator=g)
    out["x"], out["y"] = x.numpy(), y.numpy()
    for d in ["x", "y", ("x", "y"), "t", None]:
        for o in (0, 1, 2):
            for ty in (2, 4, 6):
                for sc in (1.0, 0.37):
                    op = sp.ConvOperator(d, o, scale=sc, taylor_order=ty, device="cpu")
                    key = f"k|{dom_tag(d)}|{o}|{ty}|{sc}"
                    if hasattr(op, "kernel"):
                        out[key] = op.kernel.detach().numpy()
                        if sc == 1.0:
                            out["conv|" + key[2:]] = op(x).detach().numpy() if hasattr(op, "__call__") and callable(getattr(op, "forward", None)) else op.convolution(x).detach().numpy()
                    else:
                        out[key] = np.zeros((0,), np.float32)
    for bc in ("periodic", "dirichlet", "neumann", "outflow", "symmetric", "free_slip"):
        for ks in (3, 5):
            m = bcm.BoundaryManager(kernel_size=(ks, ks))
            m.set_all_boundaries(bc_type=bc, value=0.75)
            out[f"pad|{bc}|{ks}"] = m.pad_signal(x).numpy()
    m = bcm.BoundaryManager(kernel_size=3)
    m.set_boundary_type("left", "dirichlet", 1.5); m.set_boundary_type("right", "neumann")
    m.set_boundary_type("top", "symmetric"); m.set_boundary_type("bottom", "periodic")
    out["pad|mixed|3"] = m.pad_signal(x).numpy()
    for bc in ("periodic", "dirichlet", "neumann", "symmetric"):
        for ty in (2, 4):
            L = vs.Laplace(scale=1.7, taylor_order=ty, boundary_cond=bc, device="cpu")
            out[f"laplace|{bc}|{ty}"] = L(x).detach().numpy()
        Lv = vs.Laplace(scale=0.5, boundary_cond=bc, scalar=False, device="cpu")
        out[f"laplace_vec|{bc}"] = Lv(x, y).detach().numpy()
        Dv = vs.Divergence(scale=2.0, boundary_cond=bc, device="cpu")
        out[f"divergence|{bc}"] = Dv(x, y).detach().numpy()
        Cv = vs.Curl(scale=2.0, boundary_cond=bc, device="cpu")
        out[f"curl|{bc}"] = Cv(x, y).detach().numpy()
    v2 = torch.cat((x, y), dim=1)
    out["dot"] = vs.dot(v2, v2.flip(1)).numpy()
    out["cross"] = vs.cross(v2, v2 * 2).numpy()
    np.savez_compressed(os.path.join(HERE, "spatial.npz"), **out)


def gen_spectral():
    """conv='spectral', differentiate, integrate of the three ConvOperator files (torch.fft on CPU)."""
    import importlib
    sp = importlib.import_module("ConvOps_Spatial")
    out = {}
    g = torch.Generator().manual_seed(9)
    x4, x3, xs = torch.randn(2, 6, 8, 10, generator=g), torch.randn(3, 9, 12, generator=g), torch.randn(2, 1, 9, 12, generator=g)
    out["x4"], out["x3"], out["xs"] = x4.numpy(), x3.numpy(), xs.numpy()
    cases = [("2d_lap", Ref2D(("x", "y"), 2), x4), ("2d_t1", Ref2D("t", 1), x4), ("1d_x2", Ref1D("x", 2), x3),
             ("1d_xt", Ref1D(("x", "t"), 2), x3), ("sp_lap", sp.ConvOperator(("x", "y"), 2, device="cpu"), xs),
             ("sp_x1", sp.ConvOperator("x", 1, scale=0.5, device="cpu"), xs)]
    for name, op, x in cases:
        out[f"{name}|kernel"] = op.kernel.detach().numpy()
        out[f"{name}|spectral"] = op.spectral_convolution(x).detach().numpy()
        if name.startswith(("2d", "sp")):
            out[f"{name}|spectral_inv"] = op.spectral_convolution(x, inverse=True).detach().numpy()
        for corr in (False, True):
            for sl in (False, True):
                out[f"{name}|diff|{int(corr)}|{int(sl)}"] = op.differentiate(x, correlation=corr, slice_pad=sl).detach().numpy()
                out[f"{name}|int|{int(corr)}|{int(sl)}"] = op.integrate(x, correlation=corr, slice_pad=sl).detach().numpy()
    np.savez_compressed(os.path.join(HERE, "spectral.npz"), **out)


if __name__ == "__main__":
    if "--jorek" in sys.argv:
        gen_jorek()
        sys.exit(0)
    if "spectral" in sys.argv[1:]:
        gen_spectral()
        sys.exit(0)
    if "conformal" in sys.argv[1:]:
        gen_conformal()
        sys.exit(0)
    if "conformal_ref" in sys.argv[1:]:
        gen_conformal_ref()
        sys.exit(0)
    if "filters" in sys.argv[1:]:
        gen_filters()
        sys.exit(0)
    if "spatial" in sys.argv[1:]:
        gen_spatial()
        sys.exit(0)
    ks = gen_kernels()
    gen_apply(ks)
    gen_residuals()
    gen_jorek()
    gen_conformal()
    gen_conformal_ref()
    gen_filters()
    gen_spatial()
    gen_spectral()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))

"""Pin the CPU oracle against golden vectors produced by running the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import conformal as oc
from oracle import convops as ocv
from oracle import residuals as orr
from oracle.cstencil import xcorr_c

TOL = 1e-6   # oracle vs reference on the SAME CPU arithmetic: only summation order may differ


def _parse(key):
    tag, dom, order, taylor, scale = key.split("|")
    dom = {"none": None, "xy": ("x", "y"), "xyt": ("x", "y", "t"), "xt": ("x", "t")}.get(dom, dom)
    return tag, dom, int(order), int(taylor), float(scale)


def test_kernel_construction_matches_reference(golden):
    ks = golden["kernels"]
    assert len(ks.files) == (7 + 5) * 4 * 3 * 2
    n_with = 0
    for key in ks.files:
        tag, dom, order, taylor, scale = _parse(key)
        build = ocv.build_kernel_2d if tag == "2d" else ocv.build_kernel_1d
        mine = build(dom, order, scale, taylor)
        ref = ks[key]
        if ref.size == 0:
            assert mine is None, key
        else:
            n_with += 1
            assert mine is not None, key
            assert tuple(mine.shape) == ref.shape, key
            assert np.array_equal(mine.numpy(), ref), key        # bit-for-bit
    assert n_with > 40


def test_reference_quirks_are_reproduced(golden):
    ks = golden["kernels"]
    # domain 'y' == domain 't' 
    assert np.array_equal(ks["2d|y|1|2|1.0"], ks["2d|t|1|2|1.0"])
    assert np.array_equal(ocv.build_kernel_2d("y", 1).numpy(), ocv.build_kernel_2d("t", 1).numpy())
    # taylor 4: 5x5 stencil sits at time index 1, not the centre
    k5 = ocv.build_kernel_2d(("x", "y"), 2, taylor_order=4).numpy()
    assert k5.shape == (5, 5, 5) and np.count_nonzero(k5[1]) == 9 and np.count_nonzero(k5[2]) == 0
    # no kernel cases
    assert ocv.build_kernel_2d(("x", "y"), 1) is None
    assert ocv.build_kernel_1d("x", 3) is None
    assert ocv.build_kernel_2d(None, None) is None
    with pytest.raises(ValueError, match="Unknown Convolution Method"):
        ocv.ConvOperator2D("x", 1, conv=False)


@pytest.mark.parametrize("impl", ["torch", "numpy", "c"])
def test_apply_matches_reference(golden, impl):
    ap = golden["apply"]
    fn = {"torch": lambda x, k: ocv.xcorr_torch(torch.from_numpy(x), torch.from_numpy(k)).numpy(),
          "numpy": ocv.xcorr_numpy, "c": xcorr_c}[impl]
    n = 0
    for key in ap.files:
        if not key.startswith("out|"):
            continue
        parts = key.split("|")
        name = parts[-1]
        if parts[1] == "wave":
            kern, x = ap["kern|wave"], ap[f"in4|{name}"]
        else:
            kkey = "|".join(parts[1:-1])
            kern = golden["kernels"][kkey]
            x = ap[("in4|" if parts[1] == "2d" else "in3|") + name]
        got = fn(x, kern)
        assert got.shape == ap[key].shape, key
        assert rel_err(got, ap[key]) <= TOL, (key, rel_err(got, ap[key]))
        n += 1
    assert n > 60


def test_residuals_match_reference(golden):
    g = golden["residuals"]
    v6 = torch.from_numpy(g["vars6"])
    u1 = torch.from_numpy(g["u1d"])
    dt, dx, dy = g["coef"].tolist()
    bdx, bdt, bnu = g["burgers_coef"].tolist()
    cases = {
        "PRE_Wave": lambda b: orr.wave_residual(v6[:, 0], 1.0, 0.01, 0.02, boundary=b),
        "PRE_NS": lambda b: orr.ns_momentum(v6[:, :3], dt, dx, dy, boundary=b),
        "PRE_MHD": lambda b: orr.mhd_energy(v6, boundary=b),
        "ns_continuity": lambda b: orr.ns_continuity(v6[:, :2], dx, dy, boundary=b),
        "ns_momentum": lambda b: orr.ns_momentum(v6[:, :3], dt, dx, dy, boundary=b),
        "mhd_continuity": lambda b: orr.mhd_continuity(v6, boundary=b),
        "mhd_momentum": lambda b: orr.mhd_momentum(v6, boundary=b),
        "mhd_energy": lambda b: orr.mhd_energy(v6, boundary=b),
        "mhd_induction": lambda b: orr.mhd_induction(v6, boundary=b),
        "mhd_gauss": lambda b: orr.mhd_gauss(v6, boundary=b),
        "burgers": lambda b: orr.burgers_residual(u1, bdx, bdt, bnu, boundary=b),
        "advection": lambda b: orr.advection_residual(u1, 1.0, 2, 0.005, 0.01, boundary=b),
    }
    for name, fn in cases.items():
        for b in (0, 1):
            ref = g[f"{name}|{b}"]
            got = fn(bool(b)).numpy()
            assert got.shape == ref.shape, name
            assert rel_err(got, ref) <= TOL, (name, b, rel_err(got, ref))
    assert np.array_equal(orr.advection_kernel(1.0, 2, 0.005, 0.01).numpy(), g["advection_kernel"])
    for wall in ("top", "bottom", "left", "right"):
        got = orr.periodic_bc_residual(v6[:, 0], dx, wall=wall).numpy()
        assert rel_err(got, g[f"ns_periodic_bc|{wall}"]) <= TOL


def test_jorek_residuals_match_reference(golden):
    """Reduced-MHD residuals (Marginal/JOREK_residuals_CP.py:207-243) against the vectors executed from the script's own
    operator constructions and defs (tests/golden/make_golden.py::gen_jorek); the operators' kernels bit for bit."""
    g = golden["jorek"]
    v3, R = torch.from_numpy(g["vars3"]), torch.from_numpy(g["R"])
    dx, dy, dt, D, K, gamma = g["coef"].tolist()
    t = lambda x: torch.tensor(x, dtype=torch.float32)
    o = orr.OpsJorek()
    for name in ("D_t", "D_R", "D_Z", "D_RR", "D_ZZ"):
        assert np.array_equal(getattr(o, name).kernel.numpy(), g[f"kernel|{name}"]), name
    for b in (0, 1):
        cases = {"continuity": orr.jorek_continuity(v3, R, D, boundary=bool(b)),
                 "continuity_norms": orr.jorek_continuity(v3, R, D, boundary=bool(b), norms=True, dx=t(dx), dy=t(dy), dt=t(dt)),
                 "temperature": orr.jorek_temperature(v3, R, K, gamma, boundary=bool(b))}
        for name, got in cases.items():
            ref = g[f"{name}|{b}"]
            assert tuple(got.shape) == ref.shape, (name, b)
            assert rel_err(got.numpy(), ref) <= TOL, (name, b, rel_err(got.numpy(), ref))


def test_conformal_build_defined_vectors(golden):
    """Oracle vs committed numpy vectors (BUILD-DEFINED: parity with the reference is unpinned)."""
    g = golden["conformal"]
    for n in (7, 100, 256):
        s, r = g[f"scores|{n}"], g[f"res|{n}"]
        mod = oc.modulation_func(r, np.zeros_like(r))
        assert np.array_equal(mod, g[f"mod|{n}"])
        js = oc.ncf_metric_joint(r, np.zeros_like(r), mod)
        assert np.array_equal(js, g[f"jscore|{n}"])
        srt = np.sort(s, axis=0)
        for i, a in enumerate(oc.ALPHA_LEVELS):
            k = int(g[f"k|{n}|{i}"])
            if k < 0:
                with pytest.raises(ValueError):
                    oc.calibrate(s, n, a)
                continue
            assert oc.kth_index(n, a) == k
            q = oc.calibrate(s, n, a)
            assert np.array_equal(q, g[f"qhat|{n}|{i}"])
            assert np.array_equal(q, srt[k])            # q-hat is exactly an input value
            assert float(oc.emp_cov([-q, q], r)) == float(g[f"cov|{n}|{i}"])
            qj = oc.calibrate(js, n, a)
            assert float(oc.emp_cov_joint([-qj * mod, qj * mod], r)) == float(g[f"cov_joint|{n}|{i}"])


def test_conformal_reference_executed_vectors():
    """a12 / a13 / joint coverage PINNED: the oracle against what the reference's own statements
    (Tests/test_advection_inv_sampling_marginal.py:428 modulation, :430-431 conf_metric_joint, :464-465
    prediction sets + coverage), compiled from that file by tests/golden/make_golden.py, produced."""
    from conftest import load_golden
    g = load_golden("conformal_ref.npz")
    assert "executed from the reference" in str(g["_note"])
    for n in (7, 100, 256):
        cal, pred, val = g[f"cal|{n}"], g[f"pred|{n}"], g[f"val|{n}"]
        cal_c, pred_c, val_c = cal[:, 1:-1, 1:-1], pred[:, 1:-1, 1:-1], val[:, 1:-1, 1:-1]
        mod = oc.modulation_func(cal_c, np.zeros_like(cal_c))
        assert mod.dtype == np.float32 and np.array_equal(mod, g[f"mod|{n}"])
        js = oc.ncf_metric_joint(cal_c, np.zeros_like(cal_c), mod)
        assert js.dtype == np.float32 and np.array_equal(js, g[f"jscore|{n}"])
        qs, covs = g[f"qhats|{n}"], g[f"cov_joint|{n}"]
        assert len(qs) == len(covs) >= 3 + 9
        for i, q in enumerate(qs):
            sets = [pred_c - np.float32(q) * mod, pred_c + np.float32(q) * mod]
            if i < 3:
                assert np.array_equal(sets[0], g[f"lo|{n}|{i}"]) and np.array_equal(sets[1], g[f"hi|{n}|{i}"])
            assert float(oc.emp_cov_joint(sets, val_c)) == float(covs[i]), (n, i)
        assert covs.min() < covs.max() <= 1.0 and len(np.unique(covs)) >= 3      # the vectors discriminate


def test_inline_joint_recipe_of_reference_tests():
    """Tests/test_advection_inv_sampling_marginal.py:428-431,465 restated with numpy only."""
    rng = np.random.default_rng(0)
    cal = rng.standard_normal((50, 8, 9)).astype(np.float32)
    modulation = np.std(cal, axis=0)
    score = np.max(np.abs(cal) / modulation, axis=(1, 2))
    assert np.array_equal(oc.modulation_func(cal, np.zeros_like(cal)), modulation)
    assert np.array_equal(oc.ncf_metric_joint(cal, np.zeros_like(cal), modulation), score)
    q = oc.calibrate(score, len(score), 0.1)
    sets = [-q * modulation, q * modulation]
    cov = ((cal >= sets[0]).all(axis=(1, 2)) & (cal <= sets[1]).all(axis=(1, 2))).mean()
    assert oc.emp_cov_joint(sets, cal) == cov


def test_reference_defined_filters(golden_filters=None):
    """filter_sims_joint / filter_sims_within_bounds are defined in the reference tree itself:
    the oracle is pinned to outputs of those very functions (tests/golden/filters.npz)."""
    from conftest import load_golden
    g = load_golden("filters.npz")
    y, q = g["y"], g["q"]
    n = 0
    for key in g.files:
        parts = key.split("|")
        if parts[0] == "joint":
            sc = float(parts[1])
            assert np.array_equal(oc.filter_sims_joint([-sc * q, sc * q], y), g[key]), key
            n += 1
        elif parts[0] == "within":
            sc, thr, within = float(parts[1]), float(parts[2]), bool(int(parts[3]))
            assert np.array_equal(oc.filter_sims_within_bounds(-sc * q, sc * q, y, thr, within=within), g[key]), key
            n += 1
    assert n == 3 + 18

"""Seeded randomised differential tests (MI355X): the HIP path against the oracle on shapes, memory
layouts, kernels and calibration parameters drawn at random - tile edges, ragged widths, offset /
permuted / sliced views, tap sets on and off the star, ties and rank extremes.  The seed is fixed:
the cases are the same on every run (raise PRE_FUZZ_CASES to widen the sweep by hand)."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import conformal as oc
from oracle import residuals as orr
from oracle.cstencil import xcorr_c

pytestmark = pytest.mark.gpu
CASES = int(os.environ.get("PRE_FUZZ_CASES", "60"))
RES_TOL = 1e-5


@pytest.fixture(scope="module")
def gpu():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def _random_view(rng, shape, gen):
    """A CPU tensor of logical ``shape`` with a random memory layout: contiguous, offset slice of a
    larger buffer, a permuted buffer (some other axis fastest), or a strided (::2) slice."""
    kind = rng.choice(["contig", "slice", "perm", "step"], p=[0.3, 0.3, 0.3, 0.1])
    if kind == "contig":
        return torch.randn(*shape, generator=gen)
    if kind == "slice":
        pad = [int(rng.integers(0, 4)) for _ in shape]
        big = torch.randn(*[s + 2 * p for s, p in zip(shape, pad)], generator=gen)
        return big[tuple(slice(p, p + s) for s, p in zip(shape, pad))]
    if kind == "perm":
        perm = list(rng.permutation(len(shape) - 1) + 1)              # batch stays first
        phys = [shape[0]] + [shape[p] for p in perm]
        inv = [0] + [perm.index(a) + 1 for a in range(1, len(shape))]
        return torch.randn(*phys, generator=gen).permute(*inv)
    big = torch.randn(*(list(shape[:-1]) + [2 * shape[-1]]), generator=gen)
    return big[..., ::2]


def _random_kernel(rng, gen, nd):
    kind = rng.choice(["star", "planar", "dense", "sparse"])
    k = int(rng.choice([3, 3, 5, 7]))
    if kind == "star":
        ker = torch.zeros(*([3] * nd))
        for ax in range(nd):
            for off in (0, 2):
                idx = [1] * nd
                idx[ax] = off
                ker[tuple(idx)] = float(rng.standard_normal())
        ker[tuple([1] * nd)] = float(rng.standard_normal())
        return ker
    if kind == "planar":                                               # Taylor-4/6 like: one slab of a k^nd kernel
        ker = torch.zeros(*([k] * nd))
        slab = int(rng.integers(0, k))
        sub = torch.randn(*([k] * (nd - 1)), generator=gen) * (torch.rand(*([k] * (nd - 1)), generator=gen) < 0.5)
        ker[slab] = sub
        return ker
    ker = torch.randn(*([k] * nd), generator=gen)
    if kind == "sparse":
        ker = ker * (torch.rand(*([k] * nd), generator=gen) < 0.15)
    return ker


def test_fuzz_convolution_against_c_oracle(gpu):
    from cp_pre_amd.convops_1d import ConvOperator as Conv1D
    from cp_pre_amd.convops_2d import ConvOperator as Conv2D
    rng = np.random.default_rng(2024)
    gen = torch.Generator().manual_seed(2024)
    widths = [1, 2, 3, 4, 5, 7, 8, 17, 31, 64, 65, 100, 130, 255, 256, 257, 260, 515]
    for case in range(CASES):
        nd = int(rng.choice([2, 3]))
        if nd == 3:
            shape = (int(rng.integers(1, 4)), int(rng.integers(1, 12)), int(rng.integers(1, 40)), int(rng.choice(widths)))
        else:
            shape = (int(rng.integers(1, 6)), int(rng.integers(1, 40)), int(rng.choice(widths)))
        x = _random_view(rng, shape, gen)
        ker = _random_kernel(rng, gen, nd)
        D = (Conv2D if nd == 3 else Conv1D)()
        D.kernel = ker
        got = D(x.to(gpu)) if rng.random() < 0.7 else D(x)              # device tensor, or a CPU tensor staged through
        want = xcorr_c(x.contiguous().numpy(), ker.numpy())
        assert tuple(got.shape) == shape
        assert rel_err(got.cpu().numpy(), want) <= RES_TOL, (case, shape, tuple(x.stride()), tuple(ker.shape))


def test_fuzz_fused_residuals_against_oracle(gpu):
    from cp_pre_amd import residuals as R
    rng = np.random.default_rng(7)
    gen = torch.Generator().manual_seed(7)
    for case in range(max(8, CASES // 4)):
        B, T, X = int(rng.integers(1, 4)), int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 10, 15, 21, 30, 47, 64, 95])), int(rng.integers(1, 30))
        Y = int(rng.choice([4, 8, 12, 60, 64, 68, 128, 200, 256, 260, 510, 512]))
        if T > 8 and Y > 128:
            Y = int(rng.choice([4, 8, 12, 60, 64, 68, 128]))            # keep the CPU oracle quick
        boundary = bool(rng.random() < 0.5)
        if not boundary and min(T, X, Y) < 3:
            boundary = True
        dt, dx, dy = (float(v) for v in rng.uniform(0.005, 0.1, 3))
        layout = rng.choice(["vars", "nt_fastest", "offset"])
        if layout == "vars":
            v = torch.rand(B, 6, T, X, Y, generator=gen) + 0.5
        elif layout == "nt_fastest":
            v = (torch.rand(B, 6, X, Y, T, generator=gen) + 0.5).permute(0, 1, 4, 2, 3)     # Marginal/NS_Residuals_CP.py:282
        else:
            v = (torch.rand(B, 6, T, X + 2, Y + 5, generator=gen) + 0.5)[..., 1:X + 1, 3:Y + 3]
        vd = v.to(gpu)
        checks = [
            (R.NavierStokes(dt, dx, dy).residual_momentum(vd[:, :3], boundary), orr.ns_momentum(v[:, :3], dt, dx, dy, boundary=boundary)),
            (R.NavierStokes(dt, dx, dy).residual_continuity(vd[:, :2], boundary), orr.ns_continuity(v[:, :2], dx, dy, boundary=boundary)),
            (R.MHD().residual_induction(vd, boundary), orr.mhd_induction(v, boundary=boundary)),
            (R.MHD().residual_energy(vd, boundary), orr.mhd_energy(v, boundary=boundary)),
            (R.MHD().residual_momentum(vd, boundary), orr.mhd_momentum(v, boundary=boundary)),
            (R.MHD().residual_continuity(vd, boundary), orr.mhd_continuity(v, boundary=boundary)),
        ]
        for i, (got, want) in enumerate(checks):
            assert tuple(got.shape) == tuple(want.shape), (case, i)
            if want.numel():
                assert rel_err(got.cpu().numpy(), want.numpy()) <= RES_TOL, (case, i, layout, (B, T, X, Y), boundary)


def test_fuzz_1d_and_wave_residuals_against_oracle(gpu):
    from cp_pre_amd import residuals as R
    rng = np.random.default_rng(11)
    gen = torch.Generator().manual_seed(11)
    for case in range(max(10, CASES // 3)):
        B, T = int(rng.integers(1, 7)), int(rng.integers(1, 40))
        X = int(rng.choice([3, 4, 5, 8, 63, 64, 66, 100, 128, 200, 256, 258, 512, 515]))
        boundary = bool(rng.random() < 0.5) or min(T, X) < 3
        layout = rng.choice(["contig", "nt_fastest", "offset"])
        if layout == "contig":
            u = torch.rand(B, T, X, generator=gen) + 0.5
        elif layout == "nt_fastest":
            u = (torch.rand(B, X, T, generator=gen) + 0.5).permute(0, 2, 1)          # Marginal/Advection_Residuals_CP.py:231
        else:
            u = (torch.rand(B, T + 3, X + 6, generator=gen) + 0.5)[:, 2:T + 2, 5:X + 5]
        dx, dt, nu = (float(v) for v in rng.uniform(0.002, 0.05, 3))
        ud = u.to(gpu)
        got = R.Burgers(dx, dt, nu).residual(ud, boundary)
        want = orr.burgers_residual(u, dx, dt, nu, boundary=boundary)
        assert tuple(got.shape) == tuple(want.shape)
        if want.numel():
            assert rel_err(got.cpu().numpy(), want.numpy()) <= RES_TOL, (case, "burgers", layout, (B, T, X), boundary)
        got = R.Advection(1.3, dt, dx, disc=2).residual(ud, boundary, absolute=True)
        want = orr.advection_residual(u, 1.3, 2, dt, dx, boundary=boundary).abs()
        if want.numel():
            assert rel_err(got.cpu().numpy(), want.numpy()) <= RES_TOL, (case, "advection", layout, (B, T, X), boundary)
        Y = int(rng.choice([3, 4, 16, 33, 64]))
        w = torch.randn(B, max(T, 1), Y, X, generator=gen)
        b3 = boundary or min(T, Y, X) < 3
        got = R.PRE_Wave(dt=dt, dx=dx, c=1.1).residual(w.to(gpu), boundary=b3)
        want = orr.wave_residual(w, 1.1, dt, dx, boundary=b3)
        if want.numel():
            assert rel_err(got.cpu().numpy(), want.numpy()) <= RES_TOL, (case, "wave", (B, T, Y, X), b3)


def test_fuzz_calibration_against_numpy(gpu):
    from cp_pre_amd import inductive_cp as icp
    rng = np.random.default_rng(99)
    for case in range(max(10, CASES // 3)):
        n = int(rng.choice([2, 3, 17, 64, 100, 255, 256, 257, 1000, 1025, 2048, 5000]))
        cells = tuple(int(v) for v in rng.choice([1, 2, 3, 5, 16, 33, 63, 64, 65, 130], size=int(rng.integers(1, 4))))
        if n * int(np.prod(cells)) > 4_000_000:
            cells = cells[:1]
        kind = rng.choice(["normal", "ties", "const", "signed"])
        s = rng.standard_normal((n,) + cells).astype(np.float32)
        if kind == "ties":
            s = np.round(s * 2) / 2                                      # heavy duplicates
        elif kind == "const":
            s[:] = 0.25
        if kind != "signed":
            s = np.abs(s)
        sd = torch.from_numpy(s).to(gpu)
        for alpha in rng.choice(oc.ALPHA_LEVELS, size=3, replace=False):
            if oc.quantile_level(n, alpha) > 1:
                with pytest.raises(ValueError):
                    icp.calibrate(sd, n, alpha)
                continue
            got = icp.calibrate(sd, n, alpha)
            assert np.array_equal(np.asarray(got.cpu()), oc.calibrate(s, n, alpha)), (case, n, cells, kind, alpha)
        # joint recipe pieces on the same data
        if n >= 3 and kind in ("normal", "signed"):
            b = rng.standard_normal(s.shape).astype(np.float32)
            mod = icp.modulation_func(sd, torch.from_numpy(b).to(gpu))
            mref = oc.modulation_func(s, b)
            sc = icp.ncf_metric_joint(sd, torch.from_numpy(b).to(gpu), torch.from_numpy(mref).to(gpu))
            if int(np.prod(cells)) > 1:       # numpy reduces axis 0 of [n, M>1] row by row: the kernel's order, bit for bit
                assert np.array_equal(np.asarray(mod.cpu()), mref), (case, "modulation")
            else:                             # a single cell is a contiguous 1-D reduction: numpy sums pairwise (<= 1 ulp apart)
                assert rel_err(np.asarray(mod.cpu()), mref) <= 1e-6, (case, "modulation, M=1")
            assert np.array_equal(np.asarray(sc.cpu()), oc.ncf_metric_joint(s, b, mref)), (case, "score")
            q = float(oc.calibrate(np.asarray(sc.cpu()), n, 0.25)) if oc.quantile_level(n, 0.25) <= 1 else 1.0
            sets = [b - q * mref, b + q * mref]
            dsets = [torch.from_numpy(v).to(gpu) for v in sets]
            assert float(icp.emp_cov(dsets, sd)) == pytest.approx(float(oc.emp_cov(sets, s)), abs=1e-12)
            assert float(icp.emp_cov_joint(dsets, sd)) == pytest.approx(float(oc.emp_cov_joint(sets, s)), abs=1e-12)


def test_fuzz_spatial_family_mixed_boundaries(gpu):
    """Utils/VectorConvOps_Spatial.py classes with a different boundary type on every side (set through the
    BoundaryManager, boundary_conditions.py:29-40) against the oracle's pad-then-valid-conv recipe."""
    from cp_pre_amd import vector_convops_spatial as VS
    from oracle import spatial as osp
    rng = np.random.default_rng(314)
    gen = torch.Generator().manual_seed(314)
    types = ["dirichlet", "neumann", "outflow", "periodic", "symmetric"]
    kinds = {"gradient": VS.Gradient, "laplace": VS.Laplace, "divergence": VS.Divergence, "curl": VS.Curl,
             "vector_gradient": VS.Vector_Gradient}
    for case in range(max(12, CASES // 2)):
        B, X = int(rng.integers(1, 5)), int(rng.integers(2, 40))
        Y = int(rng.choice([2, 3, 4, 5, 8, 31, 64, 65, 100, 128, 256, 260]))
        a, b = torch.randn(B, 1, X, Y, generator=gen), torch.randn(B, 1, X, Y, generator=gen)
        kind = str(rng.choice(list(kinds)))
        sides = {s: str(rng.choice(types)) for s in ("left", "right", "top", "bottom")}
        vals = {s: float(rng.standard_normal()) for s in sides}
        scale = float(rng.uniform(0.2, 3.0))
        ref = osp.VectorOp(kind, scale=scale, boundary_cond="periodic")
        ref.types, ref.values = dict(sides), dict(vals)
        op = kinds[kind](scale=scale, boundary_cond="periodic", device=gpu)
        for s in sides:
            op.bc.set_boundary_type(s, sides[s], vals[s])
        with torch.no_grad():
            got = op(a.to(gpu), b.to(gpu)) if kind != "laplace" else op(a.to(gpu))
        want = ref(a, b) if kind != "laplace" else ref(a)
        assert tuple(got.shape) == tuple(want.shape), (case, kind)
        assert rel_err(got.cpu().numpy(), want.numpy()) <= RES_TOL, (case, kind, sides, (B, X, Y))


def test_fuzz_round2_select_paths(gpu):
    """The per-cell select of round 2 (register sort for n <= 128, sampled value-linear first digit, general
    radix form, 32-bit counters): random n, cell counts, distributions and ranks against torch.sort."""
    from cp_pre_amd import inductive_cp as icp
    rng = np.random.default_rng(2024)
    g = torch.Generator(device=gpu).manual_seed(2024)
    for case in range(max(16, CASES // 2)):
        n = int(rng.choice([1, 2, 5, 63, 64, 65, 100, 127, 128, 129, 300, 777, 2047, 2048, 2049, 3000, 6144, 6145, 9001]))
        M = int(rng.choice([1, 3, 63, 64, 65, 200, 1000]))
        kind = str(rng.choice(["abs", "signed", "log", "ties", "sorted", "spike", "tiny", "const"]))
        s = torch.randn(n, M, device=gpu, generator=g)
        if kind == "abs":
            s = s.abs() * (0.1 + 10 * torch.rand(M, device=gpu, generator=g))
        elif kind == "log":
            s = torch.exp(s * float(rng.choice([1.0, 8.0, 30.0])))
        elif kind == "ties":
            s = torch.round(s * float(rng.choice([1.0, 4.0]))) / 4
        elif kind == "sorted":
            s = torch.sort(s, dim=0, descending=bool(rng.integers(0, 2))).values
        elif kind == "spike":
            s = s.abs()
            s[torch.rand(n, M, device=gpu, generator=g) < 0.02] *= 1e8
        elif kind == "tiny":
            s = 1.0 + 1e-6 * s                                            # spread far below the magnitude
        elif kind == "const":
            s = torch.full_like(s, -7.5)
        nk = int(rng.integers(1, 11))
        ks = sorted(int(k) for k in rng.integers(0, n, size=nk))
        if rng.random() < 0.3:
            ks = sorted(set(ks) | {0, n - 1})
        got = icp.kth_axis0(s, ks)
        ref = torch.sort(s, dim=0).values[ks]
        assert torch.equal(got, ref), (case, n, M, kind, ks)


def test_fuzz_round5_select_tags_and_pools(gpu):
    """Round 5 moved the "which list wants this row" bookkeeping of the register tiles (241 <= n <= 2048) and of the
    streaming form between 2049 and 4096 rows into the histogram words (a tag above the count) and made the candidate
    lists exact-size segments of a pool: random n across both ranges and their edges, partial last tiles, distributions
    that fill lists of every length (quantised, bimodal, heavy-tailed, nearly constant), ties that exhaust pools, NaN /
    +-inf sprinkled in a few cells, random and neighbouring ranks - against torch.sort (NaN cells: every rank NaN)."""
    from cp_pre_amd import inductive_cp as icp
    rng = np.random.default_rng(5005)
    g = torch.Generator(device=gpu).manual_seed(5005)
    ns = [241, 255, 256, 257, 300, 383, 384, 385, 500, 512, 513, 600, 767, 768, 769, 1000, 1024, 1025, 1200, 1536, 1537, 2000, 2047,
          2048, 2049, 2100, 2500, 3000, 3333, 3840, 4000, 4095, 4096, 4097, 5000, 6144, 8192, 9216, 9217, 11000, 12288, 12289]
    for case in range(max(24, CASES)):
        n = int(rng.choice(ns))
        M = int(rng.choice([33, 64, 65, 130, 200, 641]))
        kind = str(rng.choice(["abs", "quant", "bimodal", "log", "narrow", "ties", "sorted"]))
        s = torch.randn(n, M, device=gpu, generator=g)
        if kind == "abs":
            s = s.abs() * (0.1 + 10 * torch.rand(M, device=gpu, generator=g))
        elif kind == "quant":                                   # levels with 8 .. 40 elements each: long lists, full pools
            lv = max(2, n // int(rng.choice([8, 14, 22, 30, 40])))
            s = torch.floor(torch.rand(n, M, device=gpu, generator=g) * lv) / lv - 0.3
        elif kind == "bimodal":
            s = s * 0.05 + (torch.rand(n, M, device=gpu, generator=g) < 0.5).float() * 7.0
        elif kind == "log":
            s = torch.exp(s * float(rng.choice([1.0, 6.0, 20.0])))
        elif kind == "narrow":
            s = 3.0 + 1e-5 * s
        elif kind == "ties":
            s = torch.round(s * 2.0) / 2.0
        elif kind == "sorted":
            s = torch.sort(s.abs(), dim=0, descending=bool(rng.integers(0, 2))).values
        if rng.random() < 0.4:                                  # a few special cells
            s[int(rng.integers(0, n)), 1] = float("nan")
            s[int(rng.integers(0, n)), min(7, M - 1)] = float("inf")
            s[int(rng.integers(0, n)), min(9, M - 1)] = float("-inf")
        nk = int(rng.integers(1, 11))
        ks = sorted(int(k) for k in rng.integers(0, n, size=nk))
        if rng.random() < 0.4:
            k0 = int(rng.integers(0, n - 3))
            ks = sorted(set(ks[:7]) | {k0, k0 + 1, k0 + 2})    # neighbouring ranks: shared rows, shared lists
        got = icp.kth_axis0(s, ks)
        ref = torch.sort(s, dim=0).values[ks]
        nanmask = torch.isnan(s).any(dim=0)
        assert torch.isnan(got[:, nanmask]).all(), (case, n, M, kind)
        ok = ~nanmask
        assert torch.equal(got[:, ok], ref[:, ok]), (case, n, M, kind, ks, (got[:, ok] != ref[:, ok]).nonzero()[:4].tolist())


@pytest.mark.parametrize("n", [1024, 2048, 4095, 4096, 4097, 8192])
@pytest.mark.parametrize("M", [64, 130, 200])
def test_select_count_boundary_under_the_list_tag(gpu, n, M):
    """The list tag sits above a 12-bit count in the histogram words of the register tiles (n <= 1024 rows) and of the
    streaming tag form (n <= 4096): a column of n - 2 identical values and two outliers puts n - 2 elements in ONE bucket -
    the largest count the window can produce (the column's extremes occupy other rows) - at the sizes where it meets the
    mask (round-5 advice).  Mixed with ordinary cells in the same tiles; M = 130, 200: partial last tiles (whose idle
    lanes must not send the tile to the general form).  Bit-exact against torch.sort, every rank."""
    from cp_pre_amd import inductive_cp as icp
    g = torch.Generator(device=gpu).manual_seed(n + M)
    s = torch.randn(n, M, device=gpu, generator=g).abs_()
    for c, (lo, hi) in ((3, (-5.0, 9.0)), (M - 1, (0.25, 0.75)), (M // 2, (-1e30, 1e30))):
        s[:, c] = 0.5
        s[int(torch.randint(0, n, (1,), generator=torch.Generator().manual_seed(c)).item()), c] = lo
        s[(int(torch.randint(0, n - 1, (1,), generator=torch.Generator().manual_seed(c + 1)).item()) + 1) % n, c] = hi
    s[:, 5] = 0.5                                                            # a flat column next to them
    ks = sorted({0, 1, 2, n // 2, n - 3, n - 2, n - 1} | {icp.kth_index(n, n, 0.1), icp.kth_index(n, n, 0.9)})
    got = icp.kth_axis0(s, ks)
    ref = torch.sort(s, dim=0).values[ks]
    assert torch.equal(got, ref), (n, M, (got != ref).nonzero()[:4].tolist())


def test_fuzz_round2_flat_tap_list_and_joint_score(gpu):
    """Random tap sets on random SHORT-Nt surrogate layouts (flat tap-list kernel; 3-D and 1-D operators), and the
    cropped joint score on the same memory order (flat quad walk), against the C / numpy oracles."""
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd.convops_1d import ConvOperator as C1
    from cp_pre_amd.convops_2d import ConvOperator as C2
    rng = np.random.default_rng(77)
    gen = torch.Generator().manual_seed(77)
    for case in range(max(12, CASES // 3)):
        nt = int(rng.choice([2, 3, 5, 10, 12, 20, 31, 40, 63]))
        B, X = int(rng.integers(1, 4)), int(rng.integers(1, 20))
        Y = int(rng.choice([1, 2, 4, 7, 16, 30, 64]))
        sur = torch.randn(B, X, Y, nt, generator=gen)                     # [BS,Nx,Ny,Nt]
        v = sur.permute(0, 3, 1, 2)
        k = _random_kernel(rng, gen, 3)
        if all(s % 2 == 1 for s in k.shape):
            D = C2()
            D.kernel = k
            got = D(v.to(gpu))
            assert rel_err(got.cpu().numpy(), xcorr_c(v.contiguous().numpy(), k.numpy())) <= RES_TOL, (case, "3d", v.shape, k.shape)
        s1 = torch.randn(B + 2, X + 3, nt, generator=gen)                 # [BS,Nx,Nt] -> [BS,Nt,Nx]
        u1 = s1.permute(0, 2, 1)
        k2 = _random_kernel(rng, gen, 2)
        D1 = C1()
        D1.kernel = k2
        got = D1(u1.to(gpu))
        assert rel_err(got.cpu().numpy(), xcorr_c(u1.contiguous().numpy(), k2.numpy())) <= RES_TOL, (case, "1d", u1.shape, k2.shape)
        # joint score with a crop on the surrogate memory order
        n = int(rng.integers(2, 30))
        res = torch.randn(n, X + 2, Y + 2, nt + 2, generator=gen).permute(0, 3, 1, 2)          # logical [n,T,X,Y], T fastest
        rn = res.contiguous().numpy()
        inner = rn[:, 1:-1, 1:-1, 1:-1]
        mod_full = np.abs(np.random.default_rng(case).standard_normal(rn.shape[1:])).astype(np.float32) + 0.1
        want = oc.ncf_metric_joint(inner, np.zeros_like(inner), mod_full[1:-1, 1:-1, 1:-1])
        got = icp.ncf_metric_joint(res.to(gpu), None, torch.from_numpy(mod_full).to(gpu), crop=1)
        assert np.array_equal(got.cpu().numpy(), want), (case, "joint score", rn.shape)


def test_fuzz_round2_pruned_joint_score(gpu):
    """Branch-and-bound joint score against the full pass on random slabs: plane counts around the 16-plane chunks, row and
    segment counts, crops, per-cell scales from smooth to wild (loose bounds: everything is evaluated), spikes, zero and
    NaN cells, scores carried over from an earlier slab - bit-identical scores for the same modulation, and the fused
    moments against the plain moments pass."""
    from cp_pre_amd import pipeline
    ops = pipeline.HipOps
    rng = np.random.default_rng(99)
    gen = torch.Generator().manual_seed(99)
    for case in range(max(16, CASES // 2)):
        n = int(rng.integers(1, 40))
        T = int(rng.choice([1, 2, 3, 4, 5, 15, 16, 17, 31, 33]))
        Y = int(rng.choice([1, 7, 63, 64, 65, 100, 200, 201, 256, 320]))
        X = int(rng.integers(1, 30))
        crop = (int(rng.integers(0, 3)), int(rng.integers(0, 3)), int(rng.integers(0, 3)))
        if T - 2 * crop[0] < 1:
            crop = (0, crop[1], crop[2])
        scale = torch.exp(float(rng.choice([0.0, 0.3, 3.0])) * torch.randn(T, X, Y, generator=gen))
        res = (torch.randn(n, T, X, Y, generator=gen) * scale).to(gpu)
        kind = rng.choice(["plain", "spike", "zero", "nan", "const"])
        t_in, x_in, y_in = T // 2, X // 2, Y // 2
        if T - 2 * crop[0] < 1 or X - 2 * crop[1] < 1 or Y - 2 * crop[2] < 1:
            crop = (0, 0, 0)
        if kind == "spike":
            res[int(rng.integers(0, n)), t_in, x_in, y_in] = 1e6
        elif kind == "zero":
            res[int(rng.integers(0, n))] = 0.0
        elif kind == "nan":
            res[int(rng.integers(0, n)), t_in, x_in, y_in] = float("nan")
        elif kind == "const":
            res[:, t_in, x_in, y_in] = 0.0                                # modulation 0 and residual 0: 0/0
        ct = crop[0]
        planes = T - 2 * ct
        old = (ops.PRUNE_MIN_CELLS, ops.PRUNE_MIN_SAMPLES)
        ops.PRUNE_MIN_CELLS = ops.PRUNE_MIN_SAMPLES = 0
        try:
            assert ops.can_prune(res, crop), (case, res.shape, crop)
            m_ref, m_new = ops.zeros_moments(planes * X * Y, gpu), ops.zeros_moments(planes * X * Y, gpu)
            ops.add_moments(res, m_ref, skip_t=ct)
            segmax = ops.add_moments_segmax(res, m_new, crop)
            assert torch.allclose(m_ref, m_new, rtol=1e-13, atol=0.0, equal_nan=True), (case, "moments")
            mod = ops.std_from_moments(m_ref, n, (T, X, Y), 0.0, like=res, skip_t=ct)
            carried = torch.rand(n, generator=gen).to(gpu) * float(rng.choice([0.0, 3.0, 6.0]))     # an earlier slab's scores
            s_full, s_pr = carried.clone(), carried.clone()
            ops.max_scores(res, mod, crop, s_full)
            ops.max_scores_pruned(res, mod, segmax, crop, s_pr)
            assert torch.equal(torch.nan_to_num(s_full, nan=-1.0), torch.nan_to_num(s_pr, nan=-1.0)), \
                (case, kind, tuple(res.shape), crop, s_full, s_pr)
        finally:
            ops.PRUNE_MIN_CELLS, ops.PRUNE_MIN_SAMPLES = old


@pytest.mark.parametrize("shape", [(2, 7, 19, 64), (1, 40, 9, 320), (3, 5, 8, 256), (2, 1, 12, 128), (5, 2, 30, 516)])
def test_three_plane_tap_sets_accumulator_march_vs_oracle(gpu, shape):
    """csrc/acc_march.hip: tap sets on three adjacent time planes within one or two cells of the centre (a dense 3^3
    kernel, the additive wave kernel with a Taylor-4 Laplacian, sparse radius-2 sets with centre-only rows) on the
    reference layout - partial row tiles, a 64-column last tile, T = 1 and 2, a T axis long enough to be cut into
    segments, a batch-strided view, |.| - against the C oracle."""
    from cp_pre_amd import _dispatch, _lib
    from cp_pre_amd.convops_2d import ConvOperator
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape[0], 2, *shape[1:], generator=g)[:, 1]                     # batch stride 2 * T * X * Y
    kernels = {"dense3": torch.randn(3, 3, 3, generator=g)}
    k5 = torch.zeros(5, 5, 5)
    k5[1:4, 1:4, 1:4] = ConvOperator("t", 2).kernel
    kernels["wave_taylor4"] = k5 - 0.25 * ConvOperator(("x", "y"), 2, taylor_order=4).kernel
    sp = torch.zeros(5, 5, 5)
    sp[1:4] = torch.randn(3, 5, 5, generator=g) * (torch.rand(3, 5, 5, generator=g) < 0.3)
    sp[2, 2, 2] = 1.0
    kernels["sparse_r2"] = sp
    one = torch.zeros(3, 3, 3)
    one[0, 2, 0], one[2, 0, 2] = 2.0, -3.0                                            # two corner taps on the outer planes
    kernels["corners"] = one
    for name, ker in kernels.items():
        D = ConvOperator()
        D.kernel = ker
        want = xcorr_c(x.contiguous().numpy(), ker.numpy())
        got = D(x.to(gpu))
        assert rel_err(got.cpu().numpy(), want) <= RES_TOL, (name, shape)
        got_abs = _dispatch._xcorr_impl(x.to(gpu), ker, 3, flags=_lib.PRE_FLAG_ABS)
        assert rel_err(got_abs.cpu().numpy(), np.abs(want)) <= RES_TOL, (name, shape, "abs")

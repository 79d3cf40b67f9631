"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI via the drop-in
Python surface, against (1) the committed golden vectors produced by the reference and
(2) the CPU oracle on seeded inputs.

Tolerances (north_star): residual tensors within 1e-5 tensor-scale relative error
(max|a-b|/max|b|, SURVEY.md 7 'hard parts'); conformal q-hat within 1e-6 relative - and in
fact bit-exact wherever the scores are bit-exact, because a radix select returns an input.
"""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

RES_TOL = 1e-5
QHAT_TOL = 1e-6


@pytest.fixture(scope="module")
def gpu():
    assert torch.cuda.is_available(), "pytest -m gpu needs the MI355X"
    from cp_pre_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _parse(key):
    tag, dom, order, taylor, scale = key.split("|")
    dom = {"none": None, "xy": ("x", "y"), "xyt": ("x", "y", "t"), "xt": ("x", "t")}.get(dom, dom)
    return tag, dom, int(order), int(taylor), float(scale)


# ---------------------------------------------------------------- a4/a5: ConvOperator
def test_convoperator_matches_reference_golden(gpu, golden):
    """Every constructible operator of both files on the reference's own D(x) dumps (odd sizes,
    Nt=1, Y not a multiple of 4: generic kernel; 5^3 / 7^3 Taylor kernels included)."""
    from cp_pre_amd.convops_1d import ConvOperator as C1
    from cp_pre_amd.convops_2d import ConvOperator as C2
    ap = golden["apply"]
    n = 0
    for key in ap.files:
        if not key.startswith("out|") or key.startswith("out|wave"):
            continue
        parts = key.split("|")
        name = parts[-1]
        tag, dom, order, taylor, scale = _parse("|".join(parts[1:-1]))
        op = (C2 if tag == "2d" else C1)(dom, order, scale=scale, taylor_order=taylor, device=gpu)
        x = torch.from_numpy(ap[("in4|" if tag == "2d" else "in3|") + name]).to(gpu)
        got = op(x)
        assert got.is_cuda and got.shape == x.shape
        assert rel_err(got.cpu().numpy(), ap[key]) <= RES_TOL, key
        n += 1
    assert n > 60


def test_additive_kernel_wave_golden_cpu_and_gpu_tensors(gpu, golden):
    from cp_pre_amd.residuals import PRE_Wave
    ap = golden["apply"]
    w = PRE_Wave(dt=0.01, dx=0.02, c=1.0)
    assert np.array_equal(w.D.kernel.numpy(), ap["kern|wave"])
    for name in "abc":
        x = torch.from_numpy(ap[f"in4|{name}"])
        got_cpu = w.D(x)                 # CPU tensor in -> staged through the GPU -> CPU tensor out
        assert not got_cpu.is_cuda
        assert rel_err(got_cpu.numpy(), ap[f"out|wave|{name}"]) <= RES_TOL
        assert torch.equal(w.D(x.to(gpu)).cpu(), got_cpu)


@pytest.mark.parametrize("shape", [(2, 5, 16, 64), (1, 1, 8, 256), (3, 9, 11, 260), (2, 20, 40, 512), (1, 3, 7, 1028)])
def test_streaming_star_kernel_vs_oracle(gpu, shape):
    """Aligned, y-contiguous views take the LDS/sliding-window kernel: every tile shape,
    partial tiles in x and y, t-segments, against the oracle."""
    from cp_pre_amd.convops_2d import ConvOperator
    from oracle.cstencil import xcorr_c
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    D = ConvOperator()
    k = torch.zeros(3, 3, 3)
    for i, idx in enumerate([(1, 1, 1), (0, 1, 1), (2, 1, 1), (1, 0, 1), (1, 2, 1), (1, 1, 0), (1, 1, 2)]):
        k[idx] = 0.3 * (i + 1) * (-1) ** i
    D.kernel = k
    got = D(x.to(gpu)).cpu().numpy()
    assert rel_err(got, xcorr_c(x.numpy(), k.numpy())) <= RES_TOL


def test_strided_views_and_dense_kernels_vs_oracle(gpu):
    """What real callers pass (SURVEY 8b layout pitfalls): vars[:,i] of [BS,F,Nt,Nx,Ny] (streaming
    kernel, batch stride F*vol) and a permuted surrogate output with Nt fastest (generic kernel);
    plus a dense 27-tap kernel (generic kernel)."""
    from cp_pre_amd.convops_2d import ConvOperator
    from oracle.cstencil import xcorr_c
    g = torch.Generator().manual_seed(3)
    vars_ = torch.randn(3, 4, 6, 12, 32, generator=g)
    D = ConvOperator(("x", "y"), 2)
    got = D(vars_.to(gpu)[:, 2])
    assert rel_err(got.cpu().numpy(), xcorr_c(vars_[:, 2].contiguous().numpy(), D.kernel.numpy())) <= RES_TOL
    surrogate = torch.randn(3, 2, 12, 32, 6, generator=g)           # [BS,F,Nx,Ny,Nt]
    view = surrogate.permute(0, 1, 4, 2, 3)[:, 0]                   # Marginal/Wave_Residuals_CP.py:216
    got = ConvOperator("t", 2)(view.to(gpu))
    assert rel_err(got.cpu().numpy(), xcorr_c(view.contiguous().numpy(), ConvOperator("t", 2).kernel.numpy())) <= RES_TOL
    dense = torch.randn(3, 3, 3, generator=g)
    Dd = ConvOperator()
    Dd.kernel = dense
    x = torch.randn(2, 5, 9, 16, generator=g)
    assert rel_err(Dd(x.to(gpu)).cpu().numpy(), xcorr_c(x.numpy(), dense.numpy())) <= RES_TOL
    # fused residual on views whose base is 4 bytes off a 16-byte boundary (a sliced vars tensor)
    from cp_pre_amd.residuals import NavierStokes
    from oracle import residuals as orr
    wide = torch.rand(3, 3, 6, 10, 70, generator=g) + 0.5
    sl = wide[..., 1:65]
    got = NavierStokes(0.01, 0.1, 0.1).residual_momentum(wide.to(gpu)[..., 1:65], boundary=True)
    assert rel_err(got.cpu().numpy(), orr.ns_momentum(sl, 0.01, 0.1, 0.1, boundary=True).numpy()) <= RES_TOL
    # convolution(field, kernel) replaces the operator's kernel (Utils/ConvOps_2d.py:146-147)
    Dd.convolution(x.to(gpu), D.kernel)
    assert Dd.kernel is D.kernel


def test_tiled_tap_list_kernel_vs_oracle(gpu):
    """Tap sets off the 7-point star on views with a long unit-stride axis (LDS-tiled tap-list kernel):
    Taylor-4/6 Laplacians (taps on kernel slab 1, Utils/ConvOps_2d.py:70-71), dense 3^3/5^3/7^3 kernels,
    widths that are not multiples of 4 or of the tile, more rows than one tile, offset views,
    the Nt-fastest surrogate layout, |.| epilogue, and the adjoint (flipped taps) used by autograd."""
    from cp_pre_amd.convops_2d import ConvOperator
    from oracle.cstencil import xcorr_c
    g = torch.Generator().manual_seed(11)
    cases = [("t4", ConvOperator(("x", "y"), 2, taylor_order=4).kernel), ("t6", ConvOperator(("x", "y"), 2, taylor_order=6).kernel),
             ("d3", torch.randn(3, 3, 3, generator=g)), ("d5", torch.randn(5, 5, 5, generator=g)),
             ("d7", torch.randn(7, 7, 7, generator=g) * (torch.rand(7, 7, 7, generator=g) < 0.2))]
    shapes = [(2, 3, 20, 64), (1, 5, 37, 130), (2, 2, 16, 259), (1, 9, 5, 515)]
    for name, k in cases:
        D = ConvOperator()
        D.kernel = k
        for shape in shapes:
            x = torch.randn(*shape, generator=g)
            want = xcorr_c(x.numpy(), k.numpy())
            assert rel_err(D(x.to(gpu)).cpu().numpy(), want) <= RES_TOL, (name, shape)
        wide = torch.randn(2, 4, 19, 140, generator=g)
        view = wide[:, 1:, 2:, 3:133]                                    # offset base, row stride != width
        got = D(wide.to(gpu)[:, 1:, 2:, 3:133])
        assert rel_err(got.cpu().numpy(), xcorr_c(view.contiguous().numpy(), k.numpy())) <= RES_TOL, name
        surrogate = torch.randn(2, 9, 12, 70, generator=g)               # [BS,Nx,Ny,Nt]: Nt is the unit-stride axis
        v = surrogate.permute(0, 3, 1, 2)
        got = D(v.to(gpu))
        assert got.stride() == v.stride()
        assert rel_err(got.cpu().numpy(), xcorr_c(v.contiguous().numpy(), k.numpy())) <= RES_TOL, name
    # |.| epilogue through the C ABI, and the adjoint pass (autograd)
    from cp_pre_amd import _dispatch
    k = cases[3][1]
    x = torch.randn(2, 4, 18, 100, generator=g)
    got = _dispatch.xcorr(x.to(gpu), k, 3, flags=1)                      # PRE_FLAG_ABS
    assert rel_err(got.cpu().numpy(), np.abs(xcorr_c(x.numpy(), k.numpy()))) <= RES_TOL
    xg = x.to(gpu).requires_grad_(True)
    D = ConvOperator()
    D.kernel = k
    D(xg).square().sum().backward()
    xc = x.clone().requires_grad_(True)
    torch.nn.functional.conv3d(xc[:, None], k[None, None], padding=2).square().sum().backward()
    assert rel_err(xg.grad.cpu().numpy(), xc.grad.numpy()) <= 1e-4


def test_single_plane_tap_sets_row_march_vs_oracle(gpu):
    """Tap sets that read one input plane per output plane (register-window row march, plane_taps_kernel): Taylor-4 / 6
    Laplacians, in-plane crosses and dense in-plane 3x3 / 5x5 / 7x7 kernels on every kernel slab (plane offsets -3..3),
    widths from one partial strip to several strips, fewer rows than the window, many rows (several segments), batch
    / plane strides of views, the Nt-fastest layout with a long Nt (the plane axis is then Nx), |.|; and the fall-back
    to the tiled kernel when the width is not a multiple of 4 or the view is not 16-byte aligned."""
    from cp_pre_amd import _dispatch
    from cp_pre_amd.convops_2d import ConvOperator
    from oracle.cstencil import xcorr_c
    g = torch.Generator().manual_seed(23)

    def in_plane(k, slab, dense, r):
        out = torch.zeros(k, k, k)
        c = k // 2
        if dense:
            blk = torch.randn(2 * r + 1, 2 * r + 1, generator=g)
            out[slab, c - r:c + r + 1, c - r:c + r + 1] = blk
        else:
            out[slab, c - r:c + r + 1, c] = torch.randn(2 * r + 1, generator=g)
            out[slab, c, c - r:c + r + 1] = torch.randn(2 * r + 1, generator=g)
        return out

    kernels = [("t4", ConvOperator(("x", "y"), 2, taylor_order=4).kernel), ("t6", ConvOperator(("x", "y"), 2, taylor_order=6).kernel)]
    for k, r in ((3, 1), (5, 2), (7, 3), (7, 1), (5, 1)):
        for slab in range(k):
            kernels.append((f"cross k{k} r{r} slab{slab}", in_plane(k, slab, False, r)))
        kernels.append((f"dense k{k} r{r}", in_plane(k, int(torch.randint(0, k, (1,), generator=g)), True, r)))
    rows_only = torch.zeros(5, 5, 5)
    rows_only[2, :, 2] = torch.randn(5, generator=g)
    rows_only[2, 2, 2] = 0.0
    rows_only[2, 1, 0] = 0.7                                              # a row with a single off-centre tap
    kernels.append(("rows", rows_only))
    shapes = [(2, 3, 20, 64), (1, 4, 3, 128), (2, 2, 70, 260), (1, 2, 9, 516), (1, 3, 150, 72)]
    for name, k in kernels:
        D = ConvOperator()
        D.kernel = k
        for shape in shapes:
            x = torch.randn(*shape, generator=g)
            want = xcorr_c(x.numpy(), k.numpy())
            assert rel_err(D(x.to(gpu)).cpu().numpy(), want) <= RES_TOL, (name, shape)
    for name, k in kernels[:2] + kernels[-3:]:
        D = ConvOperator()
        D.kernel = k
        wide = torch.randn(3, 6, 19, 144, generator=g)
        view = wide[1:, 1:5, 2:, 4:132]                                  # offset base (16-byte aligned), strides != extents
        got = D(wide.to(gpu)[1:, 1:5, 2:, 4:132])
        assert rel_err(got.cpu().numpy(), xcorr_c(view.contiguous().numpy(), k.numpy())) <= RES_TOL, name
        off = wide[:, :, :, 3:131]                                       # misaligned base: tiled kernel
        got = D(wide.to(gpu)[:, :, :, 3:131])
        assert rel_err(got.cpu().numpy(), xcorr_c(off.contiguous().numpy(), k.numpy())) <= RES_TOL, name
        odd = torch.randn(2, 3, 21, 130, generator=g)                    # width % 4 != 0: tiled kernel
        assert rel_err(D(odd.to(gpu)).cpu().numpy(), xcorr_c(odd.numpy(), k.numpy())) <= RES_TOL, name
        surrogate = torch.randn(2, 11, 14, 72, generator=g)              # [BS,Nx,Ny,Nt]: Nt is the unit-stride axis
        v = surrogate.permute(0, 3, 1, 2)
        got = D(v.to(gpu))
        assert got.stride() == v.stride()
        assert rel_err(got.cpu().numpy(), xcorr_c(v.contiguous().numpy(), k.numpy())) <= RES_TOL, name
        x = torch.randn(2, 4, 18, 100, generator=g)
        got = _dispatch.xcorr(x.to(gpu), k, 3, flags=1)                  # PRE_FLAG_ABS
        assert rel_err(got.cpu().numpy(), np.abs(xcorr_c(x.numpy(), k.numpy()))) <= RES_TOL, name
    # many segments and strips at once, against the tiled kernel's result on a misaligned copy of the same data
    big = torch.randn(3, 5, 300, 1028, generator=g).to(gpu)
    D = ConvOperator()
    D.kernel = kernels[1][1]
    pad = torch.zeros(3, 5, 300, 1029, device=gpu)
    pad[..., 1:] = big
    assert rel_err(D(big).cpu().numpy(), D(pad[..., 1:]).cpu().numpy()) <= 1e-6


def test_two_and_three_plane_tap_sets_vs_oracle(gpu):
    """Tap sets on two or three input planes (the LDS-tiled tap-list kernel, with aligned quads and without): the additive
    wave kernel with a Taylor-4 / Taylor-6 Laplacian (README.md:47-54 with taylor_order 4 / 6: three planes with different
    row reach), dense 3^3, dense in-plane blocks on 2 or 3 arbitrary slabs of 5^3 / 7^3 kernels, planes holding a single
    off-centre tap; few and many rows, several strips, strided views, the Nt-fastest layout, |.|."""
    from cp_pre_amd import _dispatch
    from cp_pre_amd.convops_2d import ConvOperator
    from oracle.cstencil import xcorr_c
    g = torch.Generator().manual_seed(29)
    kernels = []
    for to in (4, 6):
        lap = ConvOperator(("x", "y"), 2, taylor_order=to).kernel
        dtt = torch.zeros_like(lap)                                       # the 3^3 time stencil centred in the 5^3 / 7^3 kernel
        o = lap.shape[0] // 2 - 1
        dtt[o:o + 3, o:o + 3, o:o + 3] = ConvOperator("t", 2).kernel
        kernels.append((f"wave taylor-{to}", dtt - 0.25 * lap))
    kernels.append(("dense 3^3", torch.randn(3, 3, 3, generator=g)))
    for k, r, slabs in ((5, 2, (0, 3)), (5, 1, (1, 2, 4)), (7, 3, (0, 6)), (7, 2, (2, 3, 5)), (7, 3, (1, 3, 4)), (3, 1, (0, 2))):
        kk = torch.zeros(k, k, k)
        c = k // 2
        for sl in slabs:
            kk[sl, c - r:c + r + 1, c - r:c + r + 1] = torch.randn(2 * r + 1, 2 * r + 1, generator=g) * \
                (torch.rand(2 * r + 1, 2 * r + 1, generator=g) < 0.6)
        kk[slabs[0], c, c] = 1.5                                          # (never an empty plane)
        kernels.append((f"blocks k{k} r{r} {slabs}", kk))
    lone = torch.zeros(5, 5, 5)
    lone[1, 0, 4] = 0.3                                                   # one off-centre tap on a plane of its own
    lone[3, 2, 1:4] = torch.tensor([1.0, -2.0, 1.0])
    kernels.append(("lone tap", lone))
    shapes = [(2, 3, 20, 64), (1, 4, 7, 128), (2, 2, 70, 260), (1, 3, 9, 516), (1, 2, 150, 72), (1, 1, 8, 64), (1, 2, 1, 68)]
    for name, k in kernels:
        D = ConvOperator()
        D.kernel = k
        for shape in shapes:
            x = torch.randn(*shape, generator=g)
            want = xcorr_c(x.numpy(), k.numpy())
            assert rel_err(D(x.to(gpu)).cpu().numpy(), want) <= RES_TOL, (name, shape)
        wide = torch.randn(3, 6, 19, 144, generator=g)
        view = wide[1:, 1:5, 2:, 4:132]
        got = D(wide.to(gpu)[1:, 1:5, 2:, 4:132])
        assert rel_err(got.cpu().numpy(), xcorr_c(view.contiguous().numpy(), k.numpy())) <= RES_TOL, name
        surrogate = torch.randn(2, 11, 14, 72, generator=g)              # [BS,Nx,Ny,Nt]: Nt is the unit-stride axis
        v = surrogate.permute(0, 3, 1, 2)
        got = D(v.to(gpu))
        assert got.stride() == v.stride()
        assert rel_err(got.cpu().numpy(), xcorr_c(v.contiguous().numpy(), k.numpy())) <= RES_TOL, name
        x = torch.randn(2, 4, 18, 100, generator=g)
        got = _dispatch.xcorr(x.to(gpu), k, 3, flags=1)                  # PRE_FLAG_ABS
        assert rel_err(got.cpu().numpy(), np.abs(xcorr_c(x.numpy(), k.numpy()))) <= RES_TOL, name
    # many tiles at once: aligned quads against a misaligned copy of the same data
    big = torch.randn(2, 5, 300, 1028, generator=g).to(gpu)
    pad = torch.zeros(2, 5, 300, 1029, device=gpu)
    pad[..., 1:] = big
    for name, k in kernels[:3]:
        D = ConvOperator()
        D.kernel = k
        assert rel_err(D(big).cpu().numpy(), D(pad[..., 1:]).cpu().numpy()) <= 1e-6, name


@pytest.mark.parametrize("nt", [10, 12, 20, 30, 40, 63])
def test_flat_tap_list_kernel_short_nt_vs_oracle(gpu, nt):
    """Tap sets off the 7-point star on the surrogate's native layout [BS,Nx,Ny,Nt] with a SHORT Nt (the reference's
    T_out = 10..40) - the flat tap-list kernel: Taylor-4/6 Laplacians (Utils/ConvOps_2d.py:36-62), dense 3^3 / 5^3
    kernels, the additive wave kernel with a Taylor-4 Laplacian (pad_kernel idiom), |.| epilogue, adjoint, and the
    1-D operator on [BS,Nx,Nt] data; every boundary cell included (zero padding), against the C oracle."""
    from cp_pre_amd import _dispatch
    from cp_pre_amd.convops_1d import ConvOperator as C1
    from cp_pre_amd.convops_2d import ConvOperator
    from oracle.cstencil import xcorr_c
    g = torch.Generator().manual_seed(100 + nt)
    lap4 = ConvOperator(("x", "y"), 2, taylor_order=4).kernel
    k5 = torch.zeros(5, 5, 5)
    k5[1:4, 1:4, 1:4] = ConvOperator("t", 2).kernel
    cases = [("t4", lap4), ("t6", ConvOperator(("x", "y"), 2, taylor_order=6).kernel), ("wave4", k5 - 0.25 * lap4),
             ("d3", torch.randn(3, 3, 3, generator=g)), ("d5", torch.randn(5, 5, 5, generator=g))]
    for (B, X, Y) in [(2, 9, 14), (1, 33, 6), (3, 4, 70)]:
        if (Y * nt) % 4:
            Y += 1 if ((Y + 1) * nt) % 4 == 0 else 2 if ((Y + 2) * nt) % 4 == 0 else 3
        sur = torch.randn(B, X, Y, nt, generator=g)                       # [BS,Nx,Ny,Nt]
        v = sur.permute(0, 3, 1, 2)                                       # the reference's [BS,Nt,Nx,Ny] view
        for name, k in cases:
            D = ConvOperator()
            D.kernel = k
            got = D(v.to(gpu))
            assert got.stride() == v.stride(), name
            want = xcorr_c(v.contiguous().numpy(), k.numpy())
            assert rel_err(got.cpu().numpy(), want) <= RES_TOL, (name, nt, (B, X, Y))
        got = _dispatch.xcorr(v.to(gpu), lap4, 3, flags=1)
        assert rel_err(got.cpu().numpy(), np.abs(xcorr_c(v.contiguous().numpy(), lap4.numpy()))) <= RES_TOL
    # adjoint (autograd through the flat kernel: flipped taps)
    sur = torch.randn(2, 8, 10, nt, generator=g)
    xg = sur.to(gpu).permute(0, 3, 1, 2).requires_grad_(True)
    D = ConvOperator()
    D.kernel = cases[4][1]
    D(xg).square().sum().backward()
    xc = sur.permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    torch.nn.functional.conv3d(xc[:, None], cases[4][1][None, None], padding=2).square().sum().backward()
    assert rel_err(xg.grad.cpu().numpy(), xc.grad.numpy()) <= 1e-4
    # 1-D: surrogate [BS,Nx,Nt] -> permute(0,2,1) (Joint/Burgers_Residuals_CP.py:217), dense and 5x5 kernels
    s1 = torch.randn(5, 36, nt, generator=g)
    u1 = s1.permute(0, 2, 1)
    for k in (torch.randn(3, 3, generator=g), torch.randn(5, 5, generator=g)):
        D1 = C1()
        D1.kernel = k
        got = D1(u1.to(gpu))
        assert rel_err(got.cpu().numpy(), xcorr_c(u1.contiguous().numpy(), k.numpy())) <= RES_TOL


def test_vector_ops_vs_oracle(gpu):
    from cp_pre_amd import vector_convops as V
    from oracle import convops as ocv
    g = torch.Generator().manual_seed(5)
    a, b = torch.randn(2, 6, 10, 16, generator=g), torch.randn(2, 6, 10, 16, generator=g)
    ad, bd = a.to(gpu), b.to(gpu)
    for name in ("Divergence", "Curl"):
        got = getattr(V, name)()(ad, bd).cpu().numpy()
        assert rel_err(got, getattr(ocv, name)()(a, b).numpy()) <= RES_TOL, name
    for name in ("Gradient", "Laplace"):
        got = getattr(V, name)()(ad, bd).cpu().numpy()
        assert got.shape == (2,) + tuple(a.shape)
        assert rel_err(got, getattr(ocv, name)()(a, b).numpy()) <= RES_TOL, name
    assert torch.equal(V.cross((ad, bd), (bd, ad)).cpu(), ocv.cross((a, b), (b, a)))
    assert torch.equal(V.dot((ad, bd), (bd, ad)).cpu(), ocv.dot((a, b), (b, a)))


# ---------------------------------------------------------------- a9: fused residuals
def _residual_cases(v6, u1, g):
    from cp_pre_amd import residuals as R
    dt, dx, dy = g["coef"].tolist()
    bdx, bdt, bnu = g["burgers_coef"].tolist()
    ns = lambda f: R.NavierStokes(dt, dx, dy, fused=f)            # noqa: E731
    mhd = lambda f: R.MHD(fused=f)                                # noqa: E731
    return {
        "PRE_Wave": lambda b, f: R.PRE_Wave(dt=0.01, dx=0.02, c=1.0).residual(v6[:, :1], boundary=b),
        "PRE_NS": lambda b, f: R.PRE_NS(dt, dx, dy, fused=f).residual(v6[:, :3], boundary=b),
        "PRE_MHD": lambda b, f: R.PRE_MHD(dt, dx, dy, fused=f).residual(v6, boundary=b),
        "ns_continuity": lambda b, f: ns(f).residual_continuity(v6[:, :2], boundary=b),
        "ns_momentum": lambda b, f: ns(f).residual_momentum(v6[:, :3], boundary=b),
        "mhd_continuity": lambda b, f: mhd(f).residual_continuity(v6, boundary=b),
        "mhd_momentum": lambda b, f: mhd(f).residual_momentum(v6, boundary=b),
        "mhd_energy": lambda b, f: mhd(f).residual_energy(v6, boundary=b),
        "mhd_induction": lambda b, f: mhd(f).residual_induction(v6, boundary=b),
        "mhd_gauss": lambda b, f: mhd(f).residual_gauss(v6, boundary=b),
        "burgers": lambda b, f: R.Burgers(bdx, bdt, bnu, fused=f).residual(u1, boundary=b),
        "advection": lambda b, f: R.Advection(1.0, 0.005, 0.01, disc=2).residual(u1, boundary=b),
    }


@pytest.mark.parametrize("fused", [True, False])
def test_residuals_match_reference_golden(gpu, golden, fused):
    """All residual equations, fused single pass and operator-by-operator composition, against
    outputs of the reference's own residual functions (Y=12 / X=14: generic + streaming mix)."""
    g = golden["residuals"]
    v6 = torch.from_numpy(g["vars6"]).to(gpu)
    u1 = torch.from_numpy(g["u1d"]).to(gpu)
    for name, fn in _residual_cases(v6, u1, g).items():
        for b in (0, 1):
            got = fn(bool(b), fused)
            assert got.is_cuda
            ref = g[f"{name}|{b}"]
            assert tuple(got.shape) == ref.shape, name
            assert rel_err(got.cpu().numpy(), ref) <= RES_TOL, (name, b, fused)


def test_periodic_bc_residual_matches_reference_golden(gpu, golden):
    """``NavierStokes.periodic_bc_residual`` (Marginal/NS_Residuals_CP.py:468-478) - one ``pre_edge_residual_f32`` launch
    - against the outputs of the reference's own def on the same field (``residuals.npz`` ``ns_periodic_bc|*``, made by
    executing it): all four walls, a device tensor and a CPU tensor (which comes home to the CPU), a strided
    ``vars[:, i]`` view, the surrogate's Nt-fastest layout and a bare [Nx,Ny] plane.  (a - b) * dx in fp32 both sides:
    bit for bit."""
    from cp_pre_amd.residuals import NavierStokes
    g = golden["residuals"]
    dt, dx, dy = g["coef"].tolist()
    v6 = torch.from_numpy(g["vars6"])
    u_cpu = v6[:, 0]
    walls = ("top", "bottom", "left", "right")
    ns = NavierStokes(dt, dx, dy)
    for wall in walls:
        ref = g[f"ns_periodic_bc|{wall}"]
        dev = ns.periodic_bc_residual(v6.to(gpu)[:, 0], wall=wall)          # strided vars[:, 0] view of [BS,6,Nt,Nx,Ny]
        assert dev.is_cuda and tuple(dev.shape) == ref.shape
        assert np.array_equal(dev.cpu().numpy(), ref), wall
        home = ns.periodic_bc_residual(u_cpu, wall=wall)                    # CPU tensor in -> CPU tensor out
        assert not home.is_cuda and np.array_equal(home.numpy(), ref), wall
        nt_fast = u_cpu.permute(0, 2, 3, 1).contiguous().to(gpu).permute(0, 3, 1, 2)      # memory [BS,Nx,Ny,Nt]
        assert not nt_fast.is_contiguous()
        assert np.array_equal(ns.periodic_bc_residual(nt_fast, wall=wall).cpu().numpy(), ref), wall
        plane = ns.periodic_bc_residual(u_cpu[1, 2].to(gpu), wall=wall)     # [Nx,Ny] -> [Ny] / [Nx]
        assert np.array_equal(plane.cpu().numpy(), ref[1, 2]), wall
        six = ns.periodic_bc_residual(v6.to(gpu), wall=wall)                # [BS,6,Nt,Nx,Ny]: every field at once
        assert np.array_equal(six[:, 0].cpu().numpy(), ref), wall
    assert ns.periodic_bc_residual(torch.empty(0, 4, 6, 8, device=gpu), wall="left").shape == (0, 4, 6)
    with pytest.raises(KeyError):
        ns.periodic_bc_residual(u_cpu.to(gpu), wall="front")


@pytest.mark.parametrize("fused", [True, False])
def test_jorek_residuals_match_reference_golden(gpu, golden, fused, monkeypatch):
    """Reduced-MHD residuals (Marginal/JOREK_residuals_CP.py:207-243): fused single pass and operator-by-operator
    composition against vectors executed from the script's own operator constructions and defs
    (tests/golden/jorek.npz) - continuity, continuity with norms=True, temperature, cropped and uncropped; the five
    operator kernels bit for bit.  The fused route must really be the fused kernel (its entry point is counted)."""
    from cp_pre_amd import _lib
    from cp_pre_amd import residuals as R
    g = golden["jorek"]
    v3 = torch.from_numpy(g["vars3"]).to(gpu)                              # [BS, F, Nx, Ny, Nt]
    dx, dy, dt, D, K, gamma = g["coef"].tolist()
    jo = R.JOREK(torch.from_numpy(g["R"]), D=D, K=K, gamma=gamma, dx=dx, dy=dy, dt=dt, fused=fused)
    for name in ("D_t", "D_R", "D_Z", "D_RR", "D_ZZ"):
        assert np.array_equal(getattr(jo, name).kernel.cpu().numpy(), g[f"kernel|{name}"]), name
    calls = []
    real = _lib.load().pre_residual_jorek_f32
    monkeypatch.setattr(_lib.load(), "pre_residual_jorek_f32", lambda *a: (calls.append(1), real(*a))[1])
    for b in (0, 1):
        cases = {"continuity": jo.residual_continuity(v3, boundary=bool(b)),
                 "continuity_norms": jo.residual_continuity(v3, boundary=bool(b), norms=True),
                 "temperature": jo.residual_temperature(v3, boundary=bool(b))}
        for name, got in cases.items():
            ref = g[f"{name}|{b}"]
            assert got.is_cuda and tuple(got.shape) == ref.shape, (name, b)
            assert rel_err(got.cpu().numpy(), ref) <= RES_TOL, (name, b, fused, rel_err(got.cpu().numpy(), ref))
    assert len(calls) == (6 if fused else 0)
    a = jo.residual_temperature(v3, boundary=True, absolute=True)
    assert rel_err(a.cpu().numpy(), np.abs(g["temperature|1"])) <= RES_TOL and bool((a >= 0).all())


def test_jorek_streaming_sizes_and_layouts_vs_oracle(gpu):
    """The fused reduced-MHD kernels at sizes with several tiles: the surrogate's Nt-fastest layout (the script's,
    through unstack_fields) with a short and a long Nt, a copy in the reference layout (Ny fastest), user-modified
    operator kernels (general-star instantiation), y_axis_fix against its own composition; a wrong-length R raises."""
    from cp_pre_amd import residuals as R
    from oracle import residuals as orr
    g = torch.Generator().manual_seed(23)
    for (B, N, Nt) in ((3, 64, 10), (2, 128, 24), (2, 40, 100)):
        v3 = torch.rand(B, 3, N, N, Nt, generator=g) + 0.5
        Rg = torch.linspace(1.0, 2.0, N) + 0.01 * torch.rand(N, generator=g)
        jo = R.JOREK(Rg, dx=0.1, dy=0.1, dt=0.02)
        d3 = v3.to(gpu)
        t = lambda x: torch.tensor(x, dtype=torch.float32)
        pairs = {
            "continuity": (jo.residual_continuity(d3, True), orr.jorek_continuity(v3, Rg, 3.4, boundary=True)),
            "continuity_norms": (jo.residual_continuity(d3, True, norms=True),
                                 orr.jorek_continuity(v3, Rg, 3.4, boundary=True, norms=True, dx=t(0.1), dy=t(0.1), dt=t(0.02))),
            "temperature": (jo.residual_temperature(d3, True), orr.jorek_temperature(v3, Rg, boundary=True)),
        }
        for name, (got, ref) in pairs.items():
            assert tuple(got.shape) == tuple(ref.shape) == (B, Nt, N, N)
            assert rel_err(got.cpu().numpy(), ref.numpy()) <= RES_TOL, (name, B, N, Nt)
        # the same fields laid out Ny-fastest (a [BS,F,Nt,Nx,Ny] tensor permuted into the script's axis order)
        ref_layout = d3.permute(0, 1, 4, 2, 3).contiguous().permute(0, 1, 3, 4, 2)
        assert ref_layout.stride(3) == 1 and torch.equal(ref_layout, d3)
        got = jo.residual_temperature(ref_layout, True)
        assert rel_err(got.cpu().numpy(), pairs["temperature"][1].numpy()) <= RES_TOL, ("ny-fastest", N, Nt)
    # user-modified operators (taps on all three axes: the general-star instantiation) == their composition
    jo2, jo2c = R.JOREK(Rg), R.JOREK(Rg, fused=False)
    for o, oc_ in ((jo2.D_Z, jo2c.D_Z), (jo2.D_RR, jo2c.D_RR)):
        k = o.kernel.clone()
        k[1, 1, 0], k[1, 1, 2], k[0, 1, 1] = 0.3, -0.7, 0.2
        o.kernel = k
        oc_.kernel = k.clone()
    assert rel_err(jo2.residual_temperature(d3, True).cpu().numpy(), jo2c.residual_temperature(d3, True).cpu().numpy()) <= RES_TOL
    fix, fixc = R.JOREK(Rg, y_axis_fix=True), R.JOREK(Rg, y_axis_fix=True, fused=False)
    a, b = fix.residual_continuity(d3, True), fixc.residual_continuity(d3, True)
    assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= RES_TOL
    assert rel_err(a.cpu().numpy(), jo.residual_continuity(d3, True).cpu().numpy()) > 1e-3      # not the reference's D_Z
    with pytest.raises(RuntimeError):
        R.JOREK(Rg[:-1]).residual_continuity(d3, True)


def test_fused_residuals_streaming_sizes_vs_oracle(gpu):
    """Same equations at sizes that take the fused streaming kernels (Y % 4 == 0, several tiles)."""
    from cp_pre_amd import residuals as R
    from oracle import residuals as orr
    g = torch.Generator().manual_seed(11)
    v6 = torch.rand(3, 6, 9, 20, 264, generator=g) + 0.5
    d6 = v6.to(gpu)
    dt, dx, dy = 0.01, 1 / 64, 1 / 48
    ns, mhd = R.NavierStokes(dt, dx, dy), R.MHD()
    pairs = {
        "ns_momentum": (ns.residual_momentum(d6[:, :3], True), orr.ns_momentum(v6[:, :3], dt, dx, dy, boundary=True)),
        "ns_continuity": (ns.residual_continuity(d6[:, :2], True), orr.ns_continuity(v6[:, :2], dx, dy, boundary=True)),
        "mhd_continuity": (mhd.residual_continuity(d6, True), orr.mhd_continuity(v6, boundary=True)),
        "mhd_momentum": (mhd.residual_momentum(d6, True), orr.mhd_momentum(v6, boundary=True)),
        "mhd_energy": (mhd.residual_energy(d6, True), orr.mhd_energy(v6, boundary=True)),
        "mhd_induction": (mhd.residual_induction(d6, True), orr.mhd_induction(v6, boundary=True)),
        "mhd_gauss": (mhd.residual_gauss(d6, True), orr.mhd_gauss(v6, boundary=True)),
    }
    for name, (got, ref) in pairs.items():
        assert rel_err(got.cpu().numpy(), ref.numpy()) <= RES_TOL, name
    u1 = torch.rand(70, 33, 128, generator=g) + 0.5
    got = R.Burgers(2 / 128, 1.25 / 33, 0.002).residual(u1.to(gpu), boundary=True)
    assert rel_err(got.cpu().numpy(), orr.burgers_residual(u1, 2 / 128, 1.25 / 33, 0.002, boundary=True).numpy()) <= RES_TOL
    # absolute=True is |residual| (marginal score), identical bits to abs() of the residual
    a = ns.residual_momentum(d6[:, :3], True, absolute=True)
    assert torch.equal(a, ns.residual_momentum(d6[:, :3], True).abs())
    # y_axis_fix (physically intended D_y; NOT reference parity) differs from strict mode and
    # equals the composition with an Ny-axis operator
    fix = R.NavierStokes(dt, dx, dy, y_axis_fix=True)
    rf = fix.residual_momentum(d6[:, :3], True)
    assert rel_err(rf.cpu().numpy(), R.NavierStokes(dt, dx, dy, y_axis_fix=True, fused=False)
                   .residual_momentum(d6[:, :3], True).cpu().numpy()) <= RES_TOL
    assert rel_err(rf.cpu().numpy(), pairs["ns_momentum"][1].numpy()) > 1e-3


def test_linearity_and_shift_properties_large(gpu):
    """Size-independent properties at a BASELINE-scale plane (512x512): D(a*x+b*y) = a*D(x)+b*D(y);
    a constant field has zero Laplacian away from the zero-padded rim; D_t of a t-ramp is 2."""
    from cp_pre_amd.convops_2d import ConvOperator
    g = torch.Generator(device="cpu").manual_seed(2)
    x = torch.randn(4, 8, 512, 512, generator=g).to(gpu)
    y = torch.randn(4, 8, 512, 512, generator=g).to(gpu)
    L = ConvOperator(("x", "y"), 2, device=gpu)
    lhs = L(2.0 * x - 0.5 * y)
    rhs = 2.0 * L(x) - 0.5 * L(y)
    assert (lhs - rhs).abs().max().item() <= 1e-5 * rhs.abs().max().item()
    ones = torch.ones(2, 4, 512, 512, device=gpu)
    assert L(ones)[..., 1:-1, 1:-1].abs().max().item() == 0.0
    assert (L(ones)[..., 0, 1:-1] == -1.0).all()                      # zero padding, not periodic
    ramp = torch.arange(8, dtype=torch.float32, device=gpu).view(1, 8, 1, 1).expand(2, 8, 512, 512).contiguous()
    assert (ConvOperator("t", 1, device=gpu)(ramp)[:, 1:-1] == 2.0).all()


# ---------------------------------------------------------------- a10-a14: conformal
def test_conformal_golden_vectors(gpu, golden):
    """Build-defined numpy vectors (parity with the reference's absent inductive_cp is unpinned)."""
    from cp_pre_amd import inductive_cp as icp
    from oracle import conformal as oc
    g = golden["conformal"]
    for n in (7, 100, 256):
        s, r = g[f"scores|{n}"], g[f"res|{n}"]
        mod = icp.modulation_func(r, np.zeros_like(r))
        assert mod.dtype == np.float32 and np.array_equal(mod, g[f"mod|{n}"])       # numpy-order std: bit exact
        js = icp.ncf_metric_joint(r, np.zeros_like(r), mod)
        assert np.array_equal(js, g[f"jscore|{n}"])
        alphas = [a for i, a in enumerate(oc.ALPHA_LEVELS) if int(g[f"k|{n}|{i}"]) >= 0]
        multi = icp.calibrate_multi(s, n, alphas)
        j = 0
        for i, a in enumerate(oc.ALPHA_LEVELS):
            if int(g[f"k|{n}|{i}"]) < 0:
                with pytest.raises(ValueError):
                    icp.calibrate(s, n, a)
                continue
            q = icp.calibrate(s, n, a)
            assert np.array_equal(q, g[f"qhat|{n}|{i}"]) and np.array_equal(multi[j], q)
            qj = icp.calibrate(js, n, a)
            assert qj == g[f"qhat_joint|{n}|{i}"]
            assert icp.emp_cov([-q, q], r) == pytest.approx(float(g[f"cov|{n}|{i}"]), abs=1e-12)
            assert icp.emp_cov_joint([-qj * mod, qj * mod], r) == pytest.approx(float(g[f"cov_joint|{n}|{i}"]), abs=1e-12)
            j += 1


def test_conformal_reference_executed_vectors(gpu):
    """a12 modulation_func, a13 ncf_metric_joint and joint coverage against vectors produced by EXECUTING the
    reference's own statements (Tests/test_advection_inv_sampling_marginal.py:428,430-431,464-465; see
    tests/golden/make_golden.py::gen_conformal_ref).  Bit-exact: fp32 numpy-order std, IEEE division, max."""
    from conftest import load_golden
    from cp_pre_amd import inductive_cp as icp
    g = load_golden("conformal_ref.npz")
    for n in (7, 100, 256):
        cal, pred, val = g[f"cal|{n}"], g[f"pred|{n}"], g[f"val|{n}"]
        cal_c, pred_c, val_c = (np.ascontiguousarray(a[:, 1:-1, 1:-1]) for a in (cal, pred, val))
        for b in (None, np.zeros_like(cal_c)):                       # "no second argument", both spellings
            mod = icp.modulation_func(cal_c, b)
            assert mod.dtype == np.float32 and np.array_equal(mod, g[f"mod|{n}"]), n
            js = icp.ncf_metric_joint(cal_c, b, mod)
            assert js.dtype == np.float32 and np.array_equal(js, g[f"jscore|{n}"]), n
        # device tensors, uncropped input + fused interior crop: same numbers
        cal_d = torch.from_numpy(cal).to(gpu)
        mod_d = icp.modulation_func(cal_d[:, 1:-1, 1:-1], None)
        assert np.array_equal(mod_d.cpu().numpy(), g[f"mod|{n}"])
        js_d = icp.ncf_metric_joint(cal_d[:, 1:-1, 1:-1], None, mod_d)
        assert np.array_equal(js_d.cpu().numpy(), g[f"jscore|{n}"])
        qs, covs = g[f"qhats|{n}"], g[f"cov_joint|{n}"]
        for i, q in enumerate(qs):
            sets = [pred_c - np.float32(q) * mod, pred_c + np.float32(q) * mod]
            assert icp.emp_cov_joint(sets, val_c) == float(covs[i]), (n, i)
            assert np.array_equal(icp.filter_sims_joint(sets, val_c),
                                  (val_c >= sets[0]).all(axis=(1, 2)) & (val_c <= sets[1]).all(axis=(1, 2)))
        # the scripts' modulation_func(res, np.zeros(res.shape)) promotes to float64: <= 1e-6 of the pinned fp32 value
        mod64 = icp.modulation_func(cal_c, np.zeros(cal_c.shape))
        assert mod64.dtype == np.float64 and np.max(np.abs(mod64 - g[f"mod|{n}"]) / g[f"mod|{n}"]) <= 2e-6


@pytest.mark.parametrize("n,cells", [(33, (5, 7)), (129, (70,)), (256, (9, 33)), (512, (3, 40, 50)), (1000, (4099,)), (4096, (2, 16, 48))])
def test_marginal_qhat_bit_exact_vs_numpy(gpu, n, cells):
    """Per-cell radix select over the batch axis: ragged cell counts, ties, negatives, +-0, inf."""
    from cp_pre_amd import inductive_cp as icp
    from oracle import conformal as oc
    rng = np.random.default_rng(n)
    s = (rng.standard_normal((n,) + cells) * np.exp(rng.standard_normal(cells) * 3)).astype(np.float32)
    s[: n // 4, ..., 0] = 1.5
    s[n // 2, ..., -1] = np.inf
    s[0].flat[:3] = [0.0, -0.0, -np.inf]
    alphas = [a for a in oc.ALPHA_LEVELS if oc.quantile_level(n, a) <= 1]
    got = icp.calibrate_multi(torch.from_numpy(s).to(gpu), n, alphas).cpu().numpy()
    for j, a in enumerate(alphas):
        assert np.array_equal(got[j], oc.calibrate(s, n, a)), (n, a)
    # |res| scores as the pipeline produces them
    a_ = np.abs(s[:, ..., 1:-1]) if s.ndim > 2 else np.abs(s)
    a_[~np.isfinite(a_)] = 0
    q = icp.calibrate(a_, n, 0.1)
    assert np.array_equal(q, oc.calibrate(a_, n, 0.1))


@pytest.mark.parametrize("n,M", [(65536, 4096), (70001, 130), (100000, 64)])
def test_marginal_qhat_large_n_wide_counters(gpu, n, M):
    """n >= 65536 (BASELINE C5 has 65536 samples; Marginal/Burgers_Residuals_CP.py:272 would select per cell over
    all of them): the 32-bit-counter instantiation, against torch.sort on the device, the 10 reference ranks
    plus the extremes."""
    from cp_pre_amd import inductive_cp as icp
    g = torch.Generator(device=gpu).manual_seed(n)
    s = torch.randn(n, M, device=gpu, generator=g).abs_() * (0.5 + torch.rand(M, device=gpu, generator=g) * 8)
    s[: n // 3, 0] = 2.0                                                # a third of one column tied
    s[:, 1] = 7.25                                                      # a constant column
    s[17, 2] = float("inf")
    ks = sorted({icp.kth_index(n, n, float(a)) for a in icp.ALPHA_LEVELS} | {0, 1, n - 2, n - 1})
    got = icp.kth_axis0(s, ks)
    ref = torch.sort(s, dim=0).values[ks]
    assert torch.equal(got, ref)


@pytest.mark.parametrize("n", [40, 130, 200, 255, 256, 1500, 2049, 5000, 8192, 9000, 12000])
def test_marginal_qhat_window_and_fallback_paths(gpu, n):
    """The sample-guided first digit must never decide the RESULT: columns whose sampled rows are unrepresentative
    (sorted data, outliers only between the sampled rows, mixed signs, huge dynamic range, heavy ties, denormals)
    and ranks at both extremes, against torch.sort.  Bit-exact."""
    from cp_pre_amd import inductive_cp as icp
    g = torch.Generator(device=gpu).manual_seed(1000 + n)
    M = 200
    s = torch.randn(n, M, device=gpu, generator=g)
    s[:, :20] = s[:, :20].abs()
    s[:, 20:40] = torch.sort(s[:, 20:40], dim=0).values                 # ordered calibration sets
    s[:, 40:50] = torch.sort(s[:, 40:50], dim=0, descending=True).values
    stride = max(1, n // 256)
    off = torch.arange(n, device=gpu) % stride != 0                     # rows the sampler never reads (if any)
    if off.any():
        s[:, 50:60] = torch.where(off[:, None], s[:, 50:60] * 1e6, s[:, 50:60] * 1e-6)
    s[:, 60:70] = torch.exp(s[:, 60:70] * 20)                           # ~170 binades
    s[:, 70:80] = torch.round(s[:, 70:80] * 2) / 2                      # heavy ties (a handful of distinct values)
    s[:, 80:90] = s[:, 80:90] * 1e-42                                   # denormals
    s[: n // 2, 90:95] = 3.0
    s[n // 2:, 90:95] = -3.0                                            # two values only
    s[::3, 95:100] = 0.0
    s[1::3, 95:100] = -0.0
    # a constant sample (every sampled row equal) hiding infinities / a tiny spread between the sampled rows
    s[:, 100:104] = 2.0
    s[:, 104:108] = 2.0
    if off.any():
        idx = torch.nonzero(off)[:, 0]
        s[idx[: max(1, len(idx) // 3)], 100:102] = float("inf")
        s[idx[-max(1, len(idx) // 3):], 102:104] = float("-inf")
        s[idx, 104:108] = 2.0 + 1e-6 * torch.randn(len(idx), 4, device=gpu, generator=g)
    # np.quantile: a NaN anywhere in a cell's column makes every quantile of that cell NaN (the other cells of the
    # tile are unaffected; the tile takes the general form)
    s[3, 108:112] = float("nan")
    s[n // 2, 110:112] = float("nan")
    s[n - 1, 111] = float("inf")
    has_nan = torch.isnan(s).any(dim=0)
    ks = sorted({icp.kth_index(n, n, float(a)) for a in icp.ALPHA_LEVELS if icp.quantile_level(n, float(a)) <= 1}
                | {0, 1, n // 2, n - 2, n - 1})
    for group in (ks[:10], ks[-10:]):
        got = icp.kth_axis0(s, group)
        ref = torch.sort(s, dim=0).values[group]
        same = torch.where(has_nan[None, :], torch.isnan(got), got == ref)
        assert bool(same.all()), (n, group)
    with np.errstate(all="ignore"):
        col = s[:, 109].cpu().numpy()
        assert np.isnan(np.quantile(col, 0.5, method="higher")) and np.isnan(icp.kth_axis0(s, [n // 2])[0, 109].item())


@pytest.mark.parametrize("n", [129, 200, 256, 257, 300, 511, 512, 513, 640, 777, 1000, 1023, 1024])
def test_marginal_qhat_register_resident_tiles(gpu, n):
    """128 < n <= 1024 (kth_tile_kernel: a persistent workgroup holds a 64-cell tile in registers - 16 / 32 rows per
    thread with two workgroups per CU, 64 with one -, exact [min, max] window, next tile prefetched into the same
    registers; tiles its fast form cannot finish are marked and redone by the streaming form once its loop is done): ragged cell counts (fewer tiles than CUs, several tiles per workgroup, a partial
    last tile), one rank / ten / more than ten, both extremes, and every column kind that decides a path - constant,
    two-valued, heavy ties, one huge outlier (stretched window -> marked tile), infinities of either sign, NaNs,
    denormals, sorted columns, mixed signs - against torch.sort, bit for bit."""
    from cp_pre_amd import inductive_cp as icp
    g = torch.Generator(device=gpu).manual_seed(7000 + n)
    for M in (1, 63, 64, 65, 200, 4099, 40000):
        s = torch.randn(n, M, device=gpu, generator=g) * torch.exp(2 * torch.randn(M, device=gpu, generator=g))
        if M >= 200:
            s[:, :20] = s[:, :20].abs()
            s[:, 20] = -4.0                                              # constant column
            s[: n // 2, 21] = 3.0
            s[n // 2:, 21] = -3.0                                        # two values
            s[:, 22:26] = torch.round(s[:, 22:26]) / 2                   # heavy ties: buckets above the list capacity
            s[n // 3, 26] = 1e30                                         # one outlier stretches the window
            s[5, 27] = float("inf")
            s[n - 1, 28] = float("-inf")
            s[:, 29] = float("inf")                                      # a column of infinities only
            s[3, 30] = float("nan")
            s[:, 31] = float("nan")                                      # a column of NaNs only
            s[n - 1, 32] = float("nan")
            s[7, 32] = float("inf")
            s[:, 33:36] = s[:, 33:36] * 1e-42                            # denormals
            s[:, 36] = torch.sort(s[:, 36]).values
            s[:, 37] = torch.sort(s[:, 37], descending=True).values
            s[::3, 38] = 0.0
            s[1::3, 38] = -0.0
            s[:, 39] = 1.0 + 1e-7 * torch.randn(n, device=gpu, generator=g)   # spread of a few ulps
            s[:, 130:140] = s[:, 130:140].abs()                          # (a clean tile next to the special one)
        if M == 40000:
            s[:, 20000:20064] = torch.round(s[:, 20000:20064] * 4)       # a whole tile of ties deep in a workgroup's run
        has_nan = torch.isnan(s).any(dim=0)
        ks_all = sorted({0, 1, n // 2, n - 2, n - 1} | {icp.kth_index(n, n, float(a)) for a in icp.ALPHA_LEVELS
                                                      if icp.quantile_level(n, float(a)) <= 1})
        ref_sorted = torch.sort(s, dim=0).values
        for group in (ks_all[:1], ks_all[:10], ks_all[-10:], ks_all):
            got = icp.kth_axis0(s, group)
            ref = ref_sorted[group]
            same = torch.where(has_nan[None, :], torch.isnan(got), got == ref)
            assert bool(same.all()), (n, M, group, torch.nonzero(~same)[:5].tolist())


@pytest.mark.parametrize("n", [129, 131, 192, 250, 256])
def test_marginal_qhat_register_tiles_16_rows(gpu, n):
    """128 < n <= 256 (kth_tile_kernel with 16 rows per thread, two workgroups per CU; round 2's two-lanes-per-cell
    register sort, which this replaced, drew these cases): cell counts around the tile sizes, one rank and more than
    ten, both extremes, ranks 127 / 128, ties, a NaN / an infinity in either half - against torch.sort, bit for bit."""
    from cp_pre_amd import inductive_cp as icp
    g = torch.Generator(device=gpu).manual_seed(n)
    for M in (1, 31, 32, 33, 127, 128, 129, 1000, 100003):
        s = torch.randn(n, M, device=gpu, generator=g) * torch.exp(2 * torch.randn(M, device=gpu, generator=g))
        s[: n // 2, 0] = 1.25                                             # ties across the two halves
        if M > 3:
            s[:, 1] = -4.0                                                # a constant column
            s[5, 2] = float("nan")                                        # NaN in the first half
            s[n - 1, 3] = float("nan")                                    # ... in the second
        if M > 6:
            s[130 % n, 4] = float("inf")
            s[7, 5] = float("-inf")
            s[:, 6] = torch.sort(s[:, 6]).values                          # sorted: every row of the 2nd half above the 1st
        has_nan = torch.isnan(s).any(dim=0)
        ks_all = sorted({0, 1, 63, 64, 126, 127, 128, 129 % n, n // 2, n - 2, n - 1}
                        | {icp.kth_index(n, n, float(a)) for a in icp.ALPHA_LEVELS})
        for group in (ks_all[:1], ks_all[:10], ks_all[-10:], ks_all):
            got = icp.kth_axis0(s, group)
            ref = torch.sort(s, dim=0).values[group]
            same = torch.where(has_nan[None, :], torch.isnan(got), got == ref)
            assert bool(same.all()), (n, M, group)


def test_joint_recipe_vs_numpy(gpu):
    """modulation -> per-sample score -> scalar q-hat -> bounds -> joint coverage; q-hat within 1e-6."""
    from cp_pre_amd import inductive_cp as icp
    from oracle import conformal as oc
    rng = np.random.default_rng(1)
    n = 2000
    res = (rng.standard_normal((n, 6, 10, 12)) * (1 + rng.random((6, 10, 12)))).astype(np.float32)
    pred = (res[:500] * 0.9).astype(np.float32)
    # (1) fp32 second argument: numpy computes in fp32 -> bit-exact
    b32 = (0.1 * rng.standard_normal(res.shape)).astype(np.float32)
    mod = icp.modulation_func(res, b32)
    assert np.array_equal(mod, oc.modulation_func(res, b32))
    sc = icp.ncf_metric_joint(res, b32, mod)
    assert np.array_equal(sc, oc.ncf_metric_joint(res, b32, mod))
    # (2) the scripts' np.zeros(res.shape) (float64): numpy promotes to float64
    z = np.zeros(res.shape)
    mod64 = icp.modulation_func(res, z)
    ref64 = oc.modulation_func(res, z)
    assert mod64.dtype == ref64.dtype == np.float64
    assert np.max(np.abs(mod64 - ref64) / ref64) <= QHAT_TOL
    sc64 = icp.ncf_metric_joint(res, z, mod64)
    ref_sc = oc.ncf_metric_joint(res, z, ref64)
    assert np.max(np.abs(sc64 - ref_sc) / ref_sc) <= QHAT_TOL
    for a in oc.ALPHA_LEVELS:
        q, qr = icp.calibrate(sc64, n, a), oc.calibrate(ref_sc, n, a)
        assert abs(q - qr) <= QHAT_TOL * abs(qr)
        sets = [-qr * ref64, qr * ref64]
        assert icp.emp_cov_joint(sets, pred) == pytest.approx(oc.emp_cov_joint(sets, pred), abs=1e-12)
        assert np.array_equal(icp.filter_sims_joint(sets, pred), oc.filter_sims_joint(sets, pred))
    # eps of Joint/MHD_Residuals_CP.py:350 and the fused interior crop
    assert np.array_equal(icp.modulation_func(res, b32, eps=1e-6), oc.modulation_func(res, b32) + np.float32(1e-6))
    full = torch.from_numpy(res).to(gpu)
    m_full = icp.modulation_func(full, None)
    s_crop = icp.ncf_metric_joint(full, None, m_full, crop=1).cpu().numpy()
    inner = res[:, 1:-1, 1:-1, 1:-1]
    assert np.array_equal(s_crop, oc.ncf_metric_joint(inner, np.zeros_like(inner), oc.modulation_func(inner, np.zeros_like(inner))))


def test_joint_score_propagates_nan_like_numpy(gpu):
    """np.max(np.abs(a-b)/mod) propagates NaN: a NaN residual cell, a NaN modulation, and 0/0 (a constant cell:
    modulation 0 with residual 0) make that sample's score NaN - also through the streaming driver - and a NaN
    score makes the scalar q-hat NaN (np.quantile); inf/0 stays inf.  Samples without such cells are unaffected."""
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd import pipeline
    from oracle import conformal as oc
    rng = np.random.default_rng(3)
    n = 40
    res = rng.standard_normal((n, 6, 10, 16)).astype(np.float32)
    res[:, 2, 3, 4] = 0.0                                   # constant cell -> modulation 0 -> 0/0 in every sample
    mod = oc.modulation_func(res, np.zeros_like(res))
    assert mod[2, 3, 4] == 0.0
    with np.errstate(all="ignore"):
        ref = oc.ncf_metric_joint(res, np.zeros_like(res), mod)
    assert np.isnan(ref).all()
    got = icp.ncf_metric_joint(res, None, icp.modulation_func(res, None))
    assert np.isnan(got).all()
    # NaN in ONE sample only; zero modulation under a non-zero residual gives inf, not NaN
    res2 = rng.standard_normal((n, 6, 10, 16)).astype(np.float32)
    res2[7, 1, 5, 9] = np.nan
    mod2 = np.abs(rng.standard_normal((6, 10, 16))).astype(np.float32) + 0.1
    mod2[4, 4, 4] = 0.0
    with np.errstate(all="ignore"):
        ref2 = oc.ncf_metric_joint(res2, np.zeros_like(res2), mod2)
    got2 = icp.ncf_metric_joint(res2, None, mod2)
    assert np.isnan(ref2[7]) and np.isinf(np.delete(ref2, 7)).all()
    assert np.array_equal(got2, ref2, equal_nan=True)
    mod2[4, 4, 4] = 0.5
    mod2[0, 0, 0] = np.nan                                  # NaN modulation at a cell the crop removes / keeps
    with np.errstate(all="ignore"):
        ref3 = oc.ncf_metric_joint(res2[:, 1:-1, 1:-1, 1:-1], np.zeros_like(res2[:, 1:-1, 1:-1, 1:-1]), mod2[1:-1, 1:-1, 1:-1])
        ref4 = oc.ncf_metric_joint(res2, np.zeros_like(res2), mod2)
    got3 = icp.ncf_metric_joint(torch.from_numpy(res2).to(gpu), None, torch.from_numpy(mod2).to(gpu), crop=1).cpu().numpy()
    assert np.array_equal(got3, ref3, equal_nan=True) and np.isnan(ref3).sum() == 1
    assert np.array_equal(icp.ncf_metric_joint(res2, None, mod2), ref4, equal_nan=True) and np.isnan(ref4).all()
    # the streaming driver: a NaN residual makes that cell's modulation NaN (np.std), hence EVERY sample's score,
    # and the NaN survives the later slabs' max-accumulation and calibrate
    jc = pipeline.JointCalibration(n, gpu)
    r = torch.from_numpy(res2).to(gpu)
    for t0 in (0, 2, 4):
        jc.add_slab(r[:, t0:t0 + 2].contiguous(), crop=(0, 1, 1))
    q = jc.finish([0.1, 0.5])
    with np.errstate(all="ignore"):
        inner = res2[:, :, 1:-1, 1:-1]
        ref5 = oc.ncf_metric_joint(inner, np.zeros_like(inner), oc.modulation_func(inner, np.zeros_like(inner)))
    assert np.isnan(ref5).all() and torch.isnan(jc.all_scores).all() and torch.isnan(q).all()
    with np.errstate(all="ignore"):
        assert np.isnan(icp.calibrate(ref2, n, 0.1)) and np.isnan(np.quantile(ref2, 0.5, method="higher"))


@pytest.mark.parametrize("shape,crop", [((37, 5, 24, 256), (0, 1, 1)), ((16, 3, 8, 192), (0, 1, 1)), ((64, 8, 40, 512), (0, 1, 1)),
                                        ((9, 16, 12, 320), (0, 1, 1)), ((21, 40, 8, 128), (1, 1, 1)), ((70, 1, 36, 256), (0, 1, 1)),
                                        ((12, 19, 4, 64), (2, 0, 3)), ((33, 4, 16, 64), (1, 1, 1)), ((25, 6, 21, 201), (1, 1, 1)),
                                        ((40, 1, 100, 200), (0, 1, 1)), ((11, 17, 3, 7), (0, 0, 1)), ((12, 2, 1, 63), (0, 0, 0))])
def test_pruned_joint_score_equals_full_pass(gpu, shape, crop, monkeypatch):
    """Branch-and-bound joint score (segment maxima from the fused moments pass + pre_joint_score_pruned_f32) against the
    full pass: identical moments, identical scores for the same modulation, identical q-hat, slab after slab; the
    segment maxima themselves against torch; a NaN residual, a zero-modulation cell, an outlier sample and a sample
    that is zero everywhere.  Shapes: the C3 slab form (<= 16 planes, no t crop), several 16-plane chunks with a t crop
    (C4), one plane (C5 arrives as [n,1,Nt,Nx]) and the 4-plane form."""
    from cp_pre_amd import pipeline
    B, T, X, Y = shape
    ct, cx, cy = crop
    g = torch.Generator().manual_seed(sum(shape))
    alphas = [0.1, 0.5, 0.9]
    ops = pipeline.HipOps
    monkeypatch.setattr(ops, "PRUNE_MIN_CELLS", 0)
    monkeypatch.setattr(ops, "PRUNE_MIN_SAMPLES", 0)
    full_jc, pruned_jc = pipeline.JointCalibration(B, gpu, prune=False), pipeline.JointCalibration(B, gpu)
    planes = T - 2 * ct
    TC = (planes + 15) // 16
    for slab in range(3):
        res = (torch.randn(B, T, X, Y, generator=g) * (0.5 + torch.rand(T, X, Y, generator=g))).to(gpu)
        if slab == 1:
            res[3, :, X // 2:X // 2 + 2, 70 % Y:] *= 40.0                  # an outlier sample
            res[B - 1] = 0.0                                              # residual exactly zero everywhere
        if slab == 2 and B > 10:
            res[7, T // 2, X // 2, 40 % Y] = float("nan")
        if slab == 0:
            res[:, :, min(3, X - 1), 13 % Y] = 0.25                       # a constant cell: modulation 0 there
        assert ops.can_prune(res, crop)
        # fused moments + segment maxima == plain moments (over the planes inside the t crop), maxima against torch
        cells = planes * X * Y
        m_ref, m_new = ops.zeros_moments(cells, gpu), ops.zeros_moments(cells, gpu)
        ops.add_moments(res, m_ref, skip_t=ct)
        segmax = ops.add_moments_segmax(res, m_new, crop)
        NS = (X * Y + 63) // 64
        assert tuple(segmax.shape) == (B, TC, NS)
        # (fp64 sums of fp32 values: equal up to the order of the additions)
        assert torch.allclose(m_ref, m_new, rtol=1e-13, atol=0.0, equal_nan=True)
        a = res[:, ct:T - ct].abs()
        if cy:
            a[..., :cy] = 0.0
            a[..., Y - cy:] = 0.0
        if cx:
            a[:, :, :cx] = 0.0
            a[:, :, X - cx:] = 0.0
        a = torch.nn.functional.pad(a.reshape(B, planes, X * Y), (0, NS * 64 - X * Y))     # flattened plane, whole segments
        a = torch.nn.functional.pad(a.permute(0, 2, 1), (0, TC * 16 - planes)).reshape(B, NS, 64, TC, 16)
        nanseg = torch.isnan(a).any(-1).any(2)                                             # [B, NS, TC]
        ref = torch.where(nanseg, torch.full((), float("nan"), device=gpu), torch.nan_to_num(a, nan=0.0).amax(-1).amax(2))
        ref = ref.permute(0, 2, 1)
        assert torch.equal(torch.nan_to_num(segmax.view(torch.float32), nan=-5.0), torch.nan_to_num(ref, nan=-5.0)), (slab, shape)
        # same modulation -> same scores, accumulated over the slabs
        mod = ops.std_from_moments(m_ref, B, (T, X, Y), 0.0, like=res, skip_t=ct)
        s_full, s_pr = full_jc.scores.clone(), full_jc.scores.clone()
        ops.max_scores(res, mod, crop, s_full)
        ops.max_scores_pruned(res, mod, segmax, crop, s_pr)
        assert torch.equal(torch.nan_to_num(s_full, nan=-1.0), torch.nan_to_num(s_pr, nan=-1.0)), (slab, shape, s_full, s_pr)
        # and through the drivers
        m1 = full_jc.add_slab(res, crop=crop)
        m2 = pruned_jc.add_slab(res, crop=crop)
        assert torch.allclose(m1, m2, rtol=1e-6, atol=0.0, equal_nan=True)
        assert torch.allclose(full_jc.scores, pruned_jc.scores, rtol=1e-5, atol=0.0, equal_nan=True)
    q1, q2 = full_jc.finish(alphas), pruned_jc.finish(alphas)
    assert torch.allclose(q1, q2, rtol=1e-5, atol=0.0, equal_nan=True)
    # tensors the pruned form does not take fall back silently: no plane inside the crop, too many segments, a view
    # that is not dense; and small tensors by default
    assert not ops.can_prune(res, (T, 1, 1))
    assert not ops.can_prune(torch.empty(1, 16 * 170, 256, 64, device=gpu), (0, 0, 0))       # 43520 segments > the LDS list
    assert Y < 3 or not ops.can_prune(res[..., ::2], crop)                 # not dense
    monkeypatch.undo()
    assert not ops.can_prune(res, crop)


def test_pruned_joint_score_whole_t_shard(gpu, monkeypatch):
    """A strong-scaling C3 shard keeps the whole T axis resident: 64 planes x 512 x 512 = 16384 segments per sample, a work
    list of 64 KiB + the kernel's static LDS - beyond the 64 KiB a dynamic allocation gets by default, inside the 160 KiB of
    a gfx950 workgroup (round 3).  Pruned == full pass; smooth data, so almost nothing is read."""
    from cp_pre_amd import pipeline
    ops = pipeline.HipOps
    monkeypatch.setattr(ops, "PRUNE_MIN_CELLS", 0)
    monkeypatch.setattr(ops, "PRUNE_MIN_SAMPLES", 0)
    B, T, X, Y = 96, 64, 512, 512
    g = torch.Generator(device=gpu).manual_seed(5)
    res = torch.randn(B, T, X, Y, device=gpu, generator=g) * (0.5 + torch.rand(T, X, Y, device=gpu, generator=g))
    res[7, 40, 100:102, 300:310] *= 30.0
    crop = (0, 1, 1)
    assert ops.can_prune(res, crop) and ((T + 15) // 16) * ((X * Y + 63) // 64) == 16384
    full, pr = pipeline.JointCalibration(B, gpu, prune=False), pipeline.JointCalibration(B, gpu)
    full.add_slab(res, crop=crop)
    pr.add_slab(res, crop=crop)
    assert pr.prune_stats is not None and int(pr.prune_stats[1]) == B * 16384
    assert torch.allclose(full.scores, pr.scores, rtol=1e-5, atol=0.0)
    assert torch.equal(torch.argmax(full.scores), torch.argmax(pr.scores)) and int(torch.argmax(pr.scores)) == 7


@pytest.mark.parametrize("shape", [(300, 16, 64, 256), (64, 5, 24, 128)])
def test_pruned_joint_score_adapts_to_wild_modulation(gpu, shape, monkeypatch):
    """The adaptive route of the branch-and-bound score (round 3).  A per-cell scale that jumps by orders of magnitude
    between neighbouring cells makes max |r| / min mod a useless bound: the kernel then flags such a sample for the full
    pass over the flagged samples (stats[2] counts them), the driver reads the counters once after its first pruned slab and takes the plain passes
    from then on - and the scores, the modulation and q-hat stay those of the full pass.  Also: a negative and a
    subnormal modulation (eps < 0 / scores of ~1e-40 scale) go through both passes alike."""
    from cp_pre_amd import pipeline
    B, T, X, Y = shape
    ops = pipeline.HipOps
    monkeypatch.setattr(ops, "PRUNE_MIN_CELLS", 0)
    monkeypatch.setattr(ops, "PRUNE_MIN_SAMPLES", 0)
    g = torch.Generator().manual_seed(B + T)
    crop = (0, 1, 1)
    alphas = [0.1, 0.5, 0.9]
    scale = torch.exp(3.0 * torch.randn(T, X, Y, generator=g)).to(gpu)
    full_jc, ad_jc = pipeline.JointCalibration(B, gpu, prune=False), pipeline.JointCalibration(B, gpu)
    for slab in range(3):
        res = torch.randn(B, T, X, Y, generator=g).to(gpu) * scale
        assert ops.can_prune(res, crop)
        m1, m2 = full_jc.add_slab(res, crop=crop), ad_jc.add_slab(res, crop=crop)
        assert torch.allclose(m1, m2, rtol=1e-6, atol=0.0)
        assert torch.allclose(full_jc.scores, ad_jc.scores, rtol=1e-5, atol=0.0)
        if slab == 0:
            read, total, swept = (int(v) for v in ad_jc.prune_stats.tolist())
            assert total == B * ((T + 15) // 16) * ((X * Y + 63) // 64) and swept > B // 2 and read > 0.25 * total
        if slab == 1:
            assert ad_jc.prune is False                                    # the stream gave the bounds up at its second slab
    assert ad_jc.score_pass_read_frac() > 0.25
    assert torch.allclose(full_jc.finish(alphas), ad_jc.finish(alphas), rtol=1e-5, atol=0.0)
    # smooth data keeps them (at a few hundred samples sigma-hat is noisy enough for the odd sample to be swept whole)
    sm_jc = pipeline.JointCalibration(B, gpu)
    res = torch.randn(B, T, X, Y, generator=g).to(gpu)
    sm_jc.add_slab(res, crop=crop)
    if B >= 256:                         # (with a few dozen samples the bounds are loose on any data)
        assert sm_jc.prune is True and int(sm_jc.prune_stats[2]) < B // 4 and sm_jc.score_pass_read_frac() < 0.25
    # same modulation -> same scores, with a negative modulation cell (full pass: a quotient <= 0 never raises the
    # maximum) and subnormal modulations (the skip-the-divide test of js_update is not valid there: always divided)
    for kind in ("negative", "subnormal"):
        res = torch.randn(B, T, X, Y, generator=g).to(gpu)
        mom = ops.zeros_moments(T * X * Y, gpu)
        segmax = ops.add_moments_segmax(res, mom, crop)
        mod = ops.std_from_moments(mom, B, (T, X, Y), 0.0, like=res)
        if kind == "negative":
            mod[T // 2, X // 2, 5::7] = -0.5
        else:
            res = res * 1e-40
            mom = ops.zeros_moments(T * X * Y, gpu)
            segmax = ops.add_moments_segmax(res, mom, crop)
            mod = (torch.rand(T, X, Y, generator=g) * 3e-40 + 1e-41).to(gpu)
            assert float(mod.max()) < 1.2e-38
        s_full, s_pr = ops.zeros_scores(B, gpu), ops.zeros_scores(B, gpu)
        ops.max_scores(res, mod, crop, s_full)
        ops.max_scores_pruned(res, mod, segmax, crop, s_pr)
        assert torch.equal(s_full, s_pr), kind
        if kind == "subnormal":          # ... and both are the exactly rounded quotients (numpy divides the same way)
            want = (res.abs() / mod)[:, :, 1:-1, 1:-1].amax(dim=(1, 2, 3))
            assert torch.equal(s_full, want)


@pytest.mark.parametrize("order", [(2, 3, 1), (3, 1, 2), (1, 3, 2)])
def test_pruned_joint_score_on_permuted_layouts(gpu, order, monkeypatch):
    """The branch-and-bound score pass on residuals whose cell axes are permuted in memory (``order`` = the logical axes
    1..3 = t, x, y from slowest to fastest; (2,3,1) is the surrogate's Nt-fastest layout [n,Nx,Ny,Nt]): segments are laid
    over the MEMORY order, the crop follows the axes - same modulation, scores and q-hat as the full pass on the same view
    and as the pruned pass on a contiguous copy, over two slabs."""
    from cp_pre_amd import pipeline
    ops = pipeline.HipOps
    monkeypatch.setattr(ops, "PRUNE_MIN_CELLS", 0)
    monkeypatch.setattr(ops, "PRUNE_MIN_SAMPLES", 0)
    g = torch.Generator().manual_seed(sum(10 ** i * o for i, o in enumerate(order)))
    n, crop, alphas = 40, (1, 2, 1), [0.1, 0.5, 0.9]
    logical = (12, 18, 37)                                                # (T, X, Y)
    phys = [logical[a - 1] for a in order]
    back = [0] + [order.index(k) + 1 for k in (1, 2, 3)]                  # memory -> logical
    jcs = {k: pipeline.JointCalibration(n, gpu, prune=p) for k, p in (("view", True), ("view_full", False), ("copy", True))}
    inner = tuple(slice(c, e - c) for c, e in zip(crop, logical))
    for slab in range(2):
        base = (torch.randn(n, *phys, generator=g) * (0.5 + torch.rand(*phys, generator=g))).to(gpu)
        res = base.permute(*back)                                         # logical [n,T,X,Y], memory order `phys`
        assert tuple(res.shape[1:]) == logical and not res.is_contiguous() and ops.can_prune(res, crop)
        mods = {"view": jcs["view"].add_slab(res, crop=crop), "view_full": jcs["view_full"].add_slab(res, crop=crop),
                "copy": jcs["copy"].add_slab(res.contiguous(), crop=crop)}
        for k in ("view_full", "copy"):
            assert torch.allclose(mods["view"][inner], mods[k][inner], rtol=1e-6, atol=0.0), (order, slab, k)
            assert torch.allclose(jcs["view"].scores, jcs[k].scores, rtol=1e-5, atol=0.0), (order, slab, k)
    q = {k: jc.finish(alphas) for k, jc in jcs.items()}
    assert torch.allclose(q["view"], q["view_full"], rtol=1e-5) and torch.allclose(q["view"], q["copy"], rtol=1e-5)


def test_scalar_kth_large_and_absdiff(gpu):
    from cp_pre_amd import _lib, inductive_cp as icp
    rng = np.random.default_rng(9)
    s = rng.standard_normal(65536).astype(np.float32)
    srt = np.sort(s)
    ks = [0, 1, 3277, 32768, 65535]
    got = icp.kth_axis0(torch.from_numpy(s).to(gpu), ks).cpu().numpy()
    assert np.array_equal(got, srt[ks])
    a = torch.randn(1000003, device=gpu)
    b = torch.randn(1000003, device=gpu)
    out = torch.empty_like(a)
    _lib.check(_lib.load().pre_absdiff_f32(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), a.numel(), _lib.stream()), "absdiff")
    assert torch.equal(out, (a - b).abs())


# ---------------------------------------------------------------- streaming pipeline (HipOps)
def test_pipeline_slabs_equal_whole_tensor(gpu):
    """t-slab streaming (what bench.py runs) == whole-tensor recipe: NS residual -> joint CP and
    marginal CP, every stage on the HIP path, against the oracle on the full tensor."""
    from cp_pre_amd import pipeline
    from cp_pre_amd.residuals import NavierStokes
    from oracle import conformal as oc
    from oracle import residuals as orr
    g = torch.Generator().manual_seed(21)
    n, T, X, Y = 40, 14, 12, 64
    v = torch.rand(n, 3, T, X, Y, generator=g) + 0.5
    dt, dx, dy = 0.01, 1 / 64, 1 / 64
    ref = orr.ns_momentum(v, dt, dx, dy, boundary=False).contiguous().numpy()         # [n,T-2,X-2,Y-2]
    ns = NavierStokes(dt, dx, dy)
    vd = v.to(gpu)
    alphas = [0.1, 0.5, 0.9]
    jc = pipeline.JointCalibration(n, gpu)
    qm_parts = []
    for t0 in range(0, T - 2, 4):                         # slabs of 4 interior planes + 2 halo planes
        slab = vd[:, :, t0:t0 + 6]
        res = ns.residual_momentum(slab, boundary=True)
        jc.add_slab(res, crop=(1, 1, 1))
        qm_parts.append(pipeline.marginal_qhat(res.abs(), alphas)[:, 1:-1, 1:-1, 1:-1])
    q = jc.finish(alphas).cpu().numpy()
    mod_ref = oc.modulation_func(ref.astype(np.float64), np.zeros(ref.shape))
    sc_ref = oc.ncf_metric_joint(ref, np.zeros(ref.shape), mod_ref)
    for j, a in enumerate(alphas):
        qr = oc.calibrate(sc_ref, n, a)
        assert abs(q[j] - qr) <= 1e-5 * abs(qr)          # residual tolerance (1e-5) propagates into the score
    qm = torch.cat(qm_parts, dim=1).cpu().numpy()
    qm_ref = np.stack([oc.calibrate(np.abs(ref), n, a) for a in alphas])
    assert qm.shape == qm_ref.shape
    assert rel_err(qm, qm_ref) <= RES_TOL


# ---------------------------------------------------------------- edge cases
def test_edge_shapes_and_layouts(gpu):
    """Empty batch, single-plane / single-row / single-column grids, Y not a multiple of 4,
    misaligned (offset-by-one) views, batch-1 huge plane: every route must agree with the oracle."""
    from cp_pre_amd.convops_1d import ConvOperator as C1
    from cp_pre_amd.convops_2d import ConvOperator as C2
    from oracle.cstencil import xcorr_c
    g = torch.Generator().manual_seed(77)
    L = C2(("x", "y"), 2)
    Dt = C2("t", 2)
    assert L(torch.empty(0, 3, 4, 8, device=gpu)).shape == (0, 3, 4, 8)
    for shape in [(1, 1, 1, 4), (2, 1, 1, 1), (1, 3, 1, 8), (1, 2, 5, 1), (2, 3, 4, 7), (1, 1, 1, 1024), (1, 2, 1030, 4),
                  (2, 4, 9, 101), (1, 3, 20, 1030), (2, 2, 33, 510), (3, 5, 6, 5)]:       # odd widths: streaming + <=3 tail columns
        x = torch.randn(*shape, generator=g)
        for D in (L, Dt):
            got = D(x.to(gpu)).cpu().numpy()
            assert rel_err(got, xcorr_c(x.numpy(), D.kernel.numpy())) <= RES_TOL, shape
    # a view whose base pointer is 4 bytes off a 16-byte boundary, and one with a padded row pitch
    big = torch.randn(2, 4, 6, 40, generator=g).to(gpu)
    off = big[..., 1:33]
    assert rel_err(L(off).cpu().numpy(), xcorr_c(off.cpu().contiguous().numpy(), L.kernel.numpy())) <= RES_TOL
    pitched = big[..., :32]
    assert rel_err(L(pitched).cpu().numpy(), xcorr_c(pitched.cpu().contiguous().numpy(), L.kernel.numpy())) <= RES_TOL
    # 1-D: [BS,1,Nt,Nx] channel form and tiny extents
    D1 = C1("x", 2)
    x3 = torch.randn(3, 1, 9, 16, generator=g)
    assert rel_err(D1(x3.to(gpu)).cpu().numpy(), xcorr_c(x3[:, 0].numpy(), D1.kernel.numpy())) <= RES_TOL
    for shape in [(1, 1, 1), (4, 1, 5), (2, 7, 1), (0, 4, 4)]:
        x = torch.randn(*shape, generator=g)
        got = D1(x.to(gpu)).cpu().numpy()
        if x.numel():
            assert rel_err(got, xcorr_c(x.numpy(), D1.kernel.numpy())) <= RES_TOL, shape
        assert got.shape == shape
    # zero kernel / operator with no taps
    Z = C2()
    Z.kernel = torch.zeros(3, 3, 3)
    assert Z(torch.randn(1, 2, 3, 4, generator=g).to(gpu)).abs().max().item() == 0.0


def test_empty_batches_through_the_residual_classes(gpu):
    """A zero-sample batch returns empty tensors of the right shape (as the reference's torch expressions do)."""
    from cp_pre_amd import residuals as R
    v = torch.zeros(0, 6, 5, 8, 16, device=gpu)
    assert R.NavierStokes(0.01, 0.1, 0.1).residual_momentum(v[:, :3], True).shape == (0, 5, 8, 16)
    assert R.NavierStokes(0.01, 0.1, 0.1).residual_continuity(v[:, :2]).shape == (0, 3, 6, 14)
    assert R.MHD().residual_induction(v, True).shape == (0, 5, 8, 16)
    assert R.Burgers(0.1, 0.1, 0.01).residual(torch.zeros(0, 6, 16, device=gpu), True).shape == (0, 6, 16)
    assert R.PRE_Wave(0.01, 0.02).residual(v[:, :1], boundary=True).shape == (0, 5, 8, 16)


def test_calibration_edge_cases(gpu):
    """n=1, all-equal scores (ties everywhere), M not a multiple of the 64-cell tile, ranks 0 and n-1,
    more ranks than one launch group (12 > 10), negative and mixed-sign scores."""
    from cp_pre_amd import inductive_cp as icp
    rng = np.random.default_rng(5)
    one = torch.from_numpy(rng.standard_normal((1, 7)).astype(np.float32)).to(gpu)
    assert torch.equal(icp.kth_axis0(one, [0]).cpu()[0], one.cpu()[0])
    ties = torch.full((300, 130), 2.5, device=gpu)
    assert (icp.kth_axis0(ties, [0, 150, 299]) == 2.5).all()
    s = rng.standard_normal((777, 193)).astype(np.float32)
    ks = [0, 1, 5, 77, 100, 200, 388, 500, 600, 700, 775, 776]
    got = icp.kth_axis0(torch.from_numpy(s).to(gpu), ks).cpu().numpy()
    assert np.array_equal(got, np.sort(s, axis=0)[ks])
    unordered = [500, 0, 776, 77]
    got = icp.kth_axis0(torch.from_numpy(s).to(gpu), unordered).cpu().numpy()
    assert np.array_equal(got, np.sort(s, axis=0)[unordered])
    with pytest.raises(RuntimeError):
        icp.kth_axis0(torch.from_numpy(s).to(gpu), [777])
    with pytest.raises(ValueError):
        icp.calibrate(s, 777, 0.0001)                      # level above 1, like numpy
    # coverage with scalar, per-cell and per-sample bounds
    from oracle import conformal as oc
    y = rng.standard_normal((50, 6, 7)).astype(np.float32)
    q = np.abs(rng.standard_normal((6, 7))).astype(np.float32)
    assert icp.emp_cov([-q, q], y) == pytest.approx(oc.emp_cov([-q, q], y), abs=1e-12)
    assert icp.emp_cov([y - 0.5, y + q], y * 1.2) == pytest.approx(oc.emp_cov([y - 0.5, y + q], y * 1.2), abs=1e-12)
    assert icp.emp_cov_joint([-3 * q, 3 * q], y) == pytest.approx(oc.emp_cov_joint([-3 * q, 3 * q], y), abs=1e-12)
    assert icp.emp_cov([np.float32(-1.0), np.float32(1.0)], y) == pytest.approx(oc.emp_cov([-1.0, 1.0], y), abs=1e-12)
    # cell counts that are multiples of 4 take the float4 kernel: ragged sample counts, both kinds of bounds, a big case
    for n_, cells in ((1, (4,)), (53, (8, 12)), (200, (3, 20, 52)), (1000, (30, 64, 64))):
        yv = rng.standard_normal((n_,) + cells).astype(np.float32)
        qv = np.abs(rng.standard_normal(cells)).astype(np.float32)
        yv[0].flat[0] = qv.flat[0]                                    # exactly on the bound: inside
        assert icp.emp_cov([-qv, qv], yv) == pytest.approx(oc.emp_cov([-qv, qv], yv), abs=1e-12)
        if n_ <= 200:
            assert icp.emp_cov([yv - 0.5, yv + qv], yv * 1.2) == pytest.approx(oc.emp_cov([yv - 0.5, yv + qv], yv * 1.2), abs=1e-12)


# ---------------------------------------------------------------- autograd
def test_autograd_matches_torch_conv(gpu):
    """D(u) inside a loss on device (Physics_Informed/Wave_FNO_PI.py:202-228): gradients w.r.t. the
    field (adjoint stencil = flipped taps, same HIP kernel) and w.r.t. D.kernel, against
    torch autograd through the oracle's F.conv3d / F.conv2d on CPU."""
    from cp_pre_amd.convops_1d import ConvOperator as C1
    from cp_pre_amd.convops_2d import ConvOperator as C2
    from cp_pre_amd.residuals import NavierStokes
    from oracle import residuals as orr
    from oracle.convops import xcorr_torch
    g = torch.Generator().manual_seed(4)
    for C, shape, dom, kw in [(C2, (2, 5, 6, 8), ("x", "y"), {}), (C2, (2, 6, 7, 8), ("x", "y"), {"taylor_order": 4}),
                              (C2, (1, 4, 4, 16), "t", {}), (C1, (3, 6, 8), "x", {})]:
        x = torch.randn(*shape, generator=g)
        w = torch.randn(*shape, generator=g)
        D = C(dom, 2, **kw)
        k_ref = D.kernel.clone().requires_grad_(True)
        x_ref = x.clone().requires_grad_(True)
        (xcorr_torch(x_ref, k_ref) * w).sum().backward()
        D.kernel = D.kernel.to(gpu).requires_grad_(True)
        xd = x.to(gpu).requires_grad_(True)
        out = D(xd)
        assert out.requires_grad
        (out * w.to(gpu)).sum().backward()
        assert rel_err(xd.grad.cpu().numpy(), x_ref.grad.numpy()) <= RES_TOL
        assert rel_err(D.kernel.grad.cpu().numpy(), k_ref.grad.numpy()) <= 1e-4      # k^nd long fp32 sums
    # 1-D operator on a [BS,1,Nt,Nx] field (Utils/ConvOps_1d.py:130-150 accepts both): the gradient keeps the channel
    import torch.nn.functional as F
    x4 = torch.randn(3, 1, 6, 8, generator=g)
    w3 = torch.randn(3, 6, 8, generator=g)
    D1 = C1("x", 2)
    k_ref, x_ref = D1.kernel.clone().requires_grad_(True), x4.clone().requires_grad_(True)
    (F.conv2d(x_ref, k_ref[None, None], padding=1).squeeze(1) * w3).sum().backward()
    D1.kernel = D1.kernel.to(gpu).requires_grad_(True)
    x4d = x4.to(gpu).requires_grad_(True)
    (D1(x4d) * w3.to(gpu)).sum().backward()
    assert x4d.grad.shape == x4.shape and rel_err(x4d.grad.cpu().numpy(), x_ref.grad.numpy()) <= RES_TOL
    assert rel_err(D1.kernel.grad.cpu().numpy(), k_ref.grad.numpy()) <= 1e-4
    # a residual used as a physics loss: fused route steps aside, gradients flow through the composition
    v = (torch.rand(2, 3, 5, 6, 16, generator=g) + 0.5)
    v_ref = v.clone().requires_grad_(True)
    orr.ns_momentum(v_ref, 0.01, 1 / 16, 1 / 16, boundary=False).pow(2).mean().backward()
    vd = v.to(gpu).requires_grad_(True)
    NavierStokes(0.01, 1 / 16, 1 / 16).residual_momentum(vd).pow(2).mean().backward()
    assert rel_err(vd.grad.cpu().numpy(), v_ref.grad.numpy()) <= 1e-4
    with torch.no_grad():                                   # and no graph is built when not asked for
        assert not C2("x", 1)(vd[:, 0]).requires_grad


def test_fused_residuals_stay_differentiable(gpu):
    """Fields that require grad (a surrogate's output outside torch.no_grad(), a physics-informed loss) still take
    the fused HIP forward; a backward recomputes the composed route.  Forward == the fused result, gradients ==
    those of the operator-by-operator expression (fused=False)."""
    from cp_pre_amd import residuals as R
    g = torch.Generator().manual_seed(77)
    v = (torch.rand(2, 6, 6, 12, 64, generator=g) + 0.5).to(gpu)
    u1 = (torch.rand(3, 10, 64, generator=g) + 0.5).to(gpu)
    cases = [("ns_momentum", lambda f, x: R.NavierStokes(0.01, 0.05, 0.04, fused=f).residual_momentum(x[:, :3], True), v),
             ("ns_continuity", lambda f, x: R.NavierStokes(0.01, 0.05, 0.04, fused=f).residual_continuity(x[:, :2], True), v),
             ("mhd_induction", lambda f, x: R.MHD(fused=f).residual_induction(x, True, absolute=True), v),
             ("mhd_energy", lambda f, x: R.MHD(fused=f).residual_energy(x, False), v),
             ("burgers", lambda f, x: R.Burgers(0.03, 0.06, 0.002, fused=f).residual(x, True), u1)]
    for name, fn, x in cases:
        with torch.no_grad():
            plain = fn(True, x)
        xa = x.clone().requires_grad_(True)
        ya = fn(True, xa)
        assert ya.requires_grad and torch.equal(ya.detach(), plain), name          # the fused forward, bit for bit
        w = torch.randn(ya.shape, generator=g).to(gpu)
        (ya * w).sum().backward()
        xb = x.clone().requires_grad_(True)
        (fn(False, xb) * w).sum().backward()
        assert rel_err(xa.grad.cpu().numpy(), xb.grad.cpu().numpy()) <= 1e-5, name


def test_kernel_gradient_single_pass(gpu):
    """pre_stencil3d_wgrad_f32 (d loss / d kernel in one pass, used by autograd when the kernel requires grad -
    Physics_Informed/Wave_FNO_PI.py:206 sets it) against torch's conv backward, 3-D and 2-D, odd and ragged sizes,
    a strided input view, and a 5^3 kernel that takes the composed route."""
    import torch.nn.functional as F
    from cp_pre_amd.convops_1d import ConvOperator as Conv1D
    from cp_pre_amd.convops_2d import ConvOperator
    g = torch.Generator().manual_seed(41)
    for shape, kshape in (((3, 7, 19, 70), (3, 3, 3)), ((2, 5, 33, 300), (3, 3, 3)), ((1, 2, 17, 515), (3, 3, 3)),
                          ((4, 20, 130), (3, 3)), ((2, 4, 9, 20), (5, 5, 5))):
        nd = len(kshape)
        x, k = torch.randn(*shape, generator=g), torch.randn(*kshape, generator=g)
        wide = torch.zeros(shape[:-1] + (shape[-1] + 3,))
        wide[..., 2:-1] = x
        xd = wide.to(gpu)[..., 2:-1].requires_grad_(True)                 # offset, non-dense rows
        kd = k.to(gpu).requires_grad_(True)
        D = (ConvOperator if nd == 3 else Conv1D)()
        D.kernel = kd
        (D(xd) ** 2).sum().backward()
        xc, kc = x.clone().requires_grad_(True), k.clone().requires_grad_(True)
        conv = F.conv3d if nd == 3 else F.conv2d
        (conv(xc[:, None], kc[None, None], padding=kshape[0] // 2) ** 2).sum().backward()
        assert rel_err(kd.grad.cpu().numpy(), kc.grad.numpy()) <= 1e-5, (shape, kshape)
        assert rel_err(xd.grad.cpu().numpy(), xc.grad.numpy()) <= 1e-5, (shape, kshape)


def test_permuted_surrogate_layout_large(gpu):
    """[BS,F,Nx,Ny,Nt] surrogate output seen through permute(0,1,4,2,3): large views are re-laid
    out on the device and take the streaming kernels; results equal the oracle on the same view."""
    from cp_pre_amd.residuals import NavierStokes, PRE_Wave
    from oracle import residuals as orr
    g = torch.Generator().manual_seed(8)
    sur = torch.rand(5, 3, 24, 64, 10, generator=g) + 0.5               # [BS,F,Nx,Ny,Nt]
    view = sur.permute(0, 1, 4, 2, 3)
    assert view.stride(-1) != 1 and view[:, 0].numel() >= 1 << 16
    got = NavierStokes(0.01, 1 / 24, 1 / 64).residual_momentum(sur.to(gpu).permute(0, 1, 4, 2, 3))
    ref = orr.ns_momentum(view, 0.01, 1 / 24, 1 / 64, boundary=False)
    assert rel_err(got.cpu().numpy(), ref.numpy()) <= RES_TOL
    w = PRE_Wave(0.01, 0.02).residual(sur.to(gpu).permute(0, 1, 4, 2, 3)[:, :1])
    assert rel_err(w.cpu().numpy(), orr.wave_residual(view[:, 0], 1.0, 0.01, 0.02).numpy()) <= RES_TOL


def test_reference_defined_filters_on_gpu(gpu):
    """filter_sims_joint (Joint/Burgers_Residuals_CP.py:298-300) and filter_sims_within_bounds
    (Active_Learning/Advection_AL_Marginal.py:169-198) against outputs of the reference's own functions."""
    from conftest import load_golden
    from cp_pre_amd import inductive_cp as icp
    g = load_golden("filters.npz")
    y, q = g["y"], g["q"]
    for key in g.files:
        parts = key.split("|")
        if parts[0] == "joint":
            sc = float(parts[1])
            assert np.array_equal(icp.filter_sims_joint([-sc * q, sc * q], y), g[key]), key
        elif parts[0] == "within":
            sc, thr, within = float(parts[1]), float(parts[2]), bool(int(parts[3]))
            got = icp.filter_sims_within_bounds(-sc * q, sc * q, y, thr, within=within)
            assert got.dtype == bool and np.array_equal(got, g[key]), key
    yt = torch.from_numpy(y).to(gpu)
    assert icp.filter_sims_within_bounds(-q, q, yt, 0.5, within=True).is_cuda


def test_reference_script_shaped_usage_through_compat_shims(gpu):
    """The lines a Marginal/ or Joint/ script executes, with the reference's own import paths
    (cp_pre_amd/compat on sys.path) and CPU tensors / numpy arrays exactly as the scripts pass them:
    Marginal/Wave_Residuals_CP.py:168-184,280-290 and Joint/Burgers_Residuals_CP.py:171-187,272-285."""
    import importlib
    import os
    import sys
    from conftest import ROOT
    from oracle import conformal as oc
    from oracle import residuals as orr
    sys.path.insert(0, os.path.join(ROOT, "cp_pre_amd", "compat"))
    try:
        for m in [k for k in sys.modules if k == "Utils" or k.startswith("Utils.") or k.startswith("Neural_PDE")]:
            del sys.modules[m]
        ConvOperator = importlib.import_module("Utils.ConvOps_2d").ConvOperator
        cp = importlib.import_module("Neural_PDE.UQ.inductive_cp")
        g = torch.Generator().manual_seed(13)
        c, dt, dx = 1.0, 0.01, 0.02
        D_tt = ConvOperator('t', 2)
        D_xx_yy = ConvOperator(('x', 'y'), 2)
        D = ConvOperator()
        cc = torch.tensor(c, dtype=torch.float32)
        D.kernel = D_tt.kernel - (cc * dt / dx) ** 2 * D_xx_yy.kernel
        uu = torch.randn(30, 8, 12, 16, generator=g)                    # CPU tensor, like cal_pred.permute(...)[:,0]
        res = D(uu)[..., 1:-1, 1:-1, 1:-1]
        assert not res.is_cuda
        ref = orr.wave_residual(uu, c, dt, dx)
        assert rel_err(res.numpy(), ref.numpy()) <= RES_TOL
        ncf_scores = np.abs(res.numpy())
        for alpha in np.arange(0.05, 0.95 + 0.1, 0.1):
            qhat = cp.calibrate(scores=ncf_scores, n=len(ncf_scores), alpha=alpha)
            assert np.array_equal(qhat, oc.calibrate(np.abs(res.numpy()), len(ncf_scores), alpha))
            cov = cp.emp_cov([-qhat, +qhat], res.numpy())
            assert cov == pytest.approx(oc.emp_cov([-qhat, qhat], res.numpy()), abs=1e-12)
        # joint flavour on a 1-D problem
        ConvOperator1 = importlib.import_module("Utils.ConvOps_1d").ConvOperator
        D_t, D_x, D_xx = ConvOperator1(domain='t', order=1), ConvOperator1(domain='x', order=1), ConvOperator1(domain='x', order=2)
        bdx, bdt, nu = (torch.tensor(v, dtype=torch.float32) for v in (2 / 64, 1.25 / 20, 0.002))
        u1 = torch.rand(40, 20, 64, generator=g) + 0.5
        r1 = (bdx * D_t(u1) + bdt * u1 * D_x(u1) - nu * D_xx(u1) * (2 * bdt / bdx))[..., 1:-1, 1:-1]
        assert rel_err(r1.numpy(), orr.burgers_residual(u1, 2 / 64, 1.25 / 20, 0.002).numpy()) <= RES_TOL
        rn = r1.numpy()
        modulation = cp.modulation_func(rn, np.zeros(rn.shape))
        scores = cp.ncf_metric_joint(rn, np.zeros(rn.shape), modulation)
        mref = oc.modulation_func(rn, np.zeros(rn.shape))
        sref = oc.ncf_metric_joint(rn, np.zeros(rn.shape), mref)
        assert np.max(np.abs(scores - sref) / sref) <= QHAT_TOL
        q = cp.calibrate(scores=scores, n=len(scores), alpha=0.5)
        assert abs(q - oc.calibrate(sref, len(sref), 0.5)) <= QHAT_TOL * abs(q)
        sets = [-q * modulation, +q * modulation]
        assert cp.emp_cov_joint(sets, rn) == pytest.approx(oc.emp_cov_joint(sets, rn), abs=1e-12)
        assert np.array_equal(cp.filter_sims_joint(sets, rn), oc.filter_sims_joint(sets, rn))
    finally:
        sys.path.remove(os.path.join(ROOT, "cp_pre_amd", "compat"))
        for m in [k for k in sys.modules if k == "Utils" or k.startswith("Utils.") or k.startswith("Neural_PDE")]:
            del sys.modules[m]


def test_intents_of_the_reference_test_scripts(gpu):
    """The reference's Tests/*.py are eyeball scripts; what each means to show, as assertions:
    Tests/test_wave.py:146-169 and Tests/test_advection.py:271-282 (individual kernels == additive kernel ==
    spectral convolution), Tests/test_convops.py:31-79 (Laplace / Divergence / Gradient classes == the scalar
    operators, 128^2 Gaussian), Tests/findiff_test.py:34-39 (scaled first derivative vs the analytic one),
    Tests/NS_vector_convops.py:131-176 (vector classes reproduce the scalar NS continuity residual),
    Tests/MM_FinDiff.py (the stencil as a matrix: W @ u == conv)."""
    from cp_pre_amd import vector_convops as V
    from cp_pre_amd.convops_1d import ConvOperator as Conv1D
    from cp_pre_amd.convops_2d import ConvOperator, get_stencil
    g = torch.Generator().manual_seed(17)
    # test_wave
    c, dt, dx = 1.0, 0.01, 0.02
    u = torch.randn(2, 12, 33, 33, generator=g).to(gpu)
    D_tt, D_xx_yy = ConvOperator('t', 2), ConvOperator(('x', 'y'), 2)
    individual = D_tt(u) - (c * dt / dx) ** 2 * D_xx_yy(u)
    D = ConvOperator()
    D.kernel = D_tt.kernel - (c * dt / dx) ** 2 * D_xx_yy.kernel
    additive, spectral = D(u), D.spectral_convolution(u)
    assert rel_err(additive.cpu().numpy(), individual.cpu().numpy()) <= RES_TOL
    assert rel_err(spectral.cpu().numpy(), individual.cpu().numpy()) <= 1e-4
    # test_advection
    v = 1.0
    uu = torch.randn(3, 30, 200, generator=g).to(gpu)
    D_t, D_x = Conv1D(domain='t', order=1), Conv1D(domain='x', order=1)
    D1 = Conv1D()
    D1.kernel = D_t.kernel + (v * dt / dx) * D_x.kernel
    individual = D_t(uu) + (v * dt / dx) * D_x(uu)
    assert rel_err(D1(uu).cpu().numpy(), individual.cpu().numpy()) <= RES_TOL
    assert rel_err(D1.spectral_convolution(uu).cpu().numpy(), individual.cpu().numpy()) <= 1e-4
    # test_convops: 2D Gaussian, field [1,1,128,128]
    x = np.linspace(-1, 1, 128)
    xx, yy = np.meshgrid(x, x)
    field = torch.tensor(np.exp(-50 * (xx ** 2 + yy ** 2)), dtype=torch.float32).view(1, 1, 128, 128).to(gpu)
    lap = V.Laplace()(field, field)
    assert torch.equal(lap[0], ConvOperator(('x', 'y'), 2)(field)) and torch.equal(lap[0], lap[1])
    Dx, Dy = ConvOperator('x', 1), ConvOperator('y', 1)
    assert rel_err(V.Divergence()(field, field).cpu().numpy(), (Dx(field) + Dy(field)).cpu().numpy()) <= RES_TOL
    gx, gy = V.Gradient()(field, field)
    assert torch.equal(gx, Dx(field)) and torch.equal(gy, Dy(field))
    # findiff_test: f = sin(x) cos(y) on [0, 2pi]^2, D_x scaled by 1/(2 dx) against the analytic d/dx
    n = 256
    xs = np.linspace(0, 2 * np.pi, n)
    h = xs[1] - xs[0]
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    f = torch.tensor(np.sin(X) * np.cos(Y), dtype=torch.float32)
    fx = ConvOperator(('x'), 1, scale=1 / (2 * h))(f[None, None].to(gpu))[0, 0].cpu().numpy()      # [1,1,Nx,Ny]: d/dx on axis 2
    assert np.max(np.abs(fx[1:-1, 1:-1] - (np.cos(X) * np.cos(Y))[1:-1, 1:-1])) < 2e-3              # O(h^2) + fp32
    # NS_vector_convops: continuity residual through Divergence == scalar operators
    uvel, vvel = torch.randn(2, 6, 20, 24, generator=g).to(gpu), torch.randn(2, 6, 20, 24, generator=g).to(gpu)
    assert rel_err(V.Divergence()(uvel, vvel).cpu().numpy(), (Dx(uvel) + Dy(vvel)).cpu().numpy()) <= RES_TOL
    # MM_FinDiff: the 5-point Laplacian as a matrix acting on the flattened plane
    m = 12
    st = get_stencil(2, 2).numpy()
    W = np.zeros((m * m, m * m), np.float32)
    for i in range(m):
        for j in range(m):
            for di in (-1, 0, 1):
                for dj in (-1, 0, 1):
                    if 0 <= i + di < m and 0 <= j + dj < m:
                        W[i * m + j, (i + di) * m + (j + dj)] = st[1 + di, 1 + dj]
    plane = torch.randn(m, m, generator=g)
    got = ConvOperator(('x', 'y'), 2)(plane[None, None].to(gpu))[0, 0].cpu().numpy()
    assert rel_err(got, (W @ plane.numpy().reshape(-1)).reshape(m, m)) <= RES_TOL


def test_marginal_step_is_hip_graph_capturable(gpu):
    """The library never allocates, synchronises or reads device data back on the host path (operator kernels
    on the CPU, ranks computed on the host), so a whole eval + |.| + per-cell q-hat step records into a HIP
    graph on torch's capture stream and replays on new data (C1 shape: Marginal/Advection_Residuals_CP.py)."""
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd import pipeline
    from cp_pre_amd import residuals as R
    alphas = [float(a) for a in icp.ALPHA_LEVELS]
    g = torch.Generator().manual_seed(29)
    u = (torch.rand(256, 100, 200, generator=g) + 0.5).to(gpu)
    op = R.Advection(1.0, 0.005, 0.01, disc=2)

    def step():
        return pipeline.marginal_qhat(op.residual(u, boundary=True, absolute=True), alphas)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        q = step()
    u.copy_((torch.rand(256, 100, 200, generator=g) + 0.5).to(gpu))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(q, step())


def test_joint_stream_is_hip_graph_capturable_with_a_fixed_route(gpu, monkeypatch):
    """``JointCalibration(prune="always")`` (and ``"never"``) never reads a device counter on the host: a two-slab
    stream - Burgers residual, moments + segment maxima, branch-and-bound score with its flagged pass, scalar q-hats -
    records into a HIP graph and replays on new data with the result of an eager run.  (The default, adaptive policy
    synchronises once, when the second slab arrives: it is refused by the capture, which is what the option is for.)"""
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd import pipeline
    from cp_pre_amd import residuals as R
    monkeypatch.setattr(pipeline.HipOps, "PRUNE_MIN_CELLS", 0)
    monkeypatch.setattr(pipeline.HipOps, "PRUNE_MIN_SAMPLES", 0)
    alphas = [0.1, 0.5, 0.9]
    g = torch.Generator().manual_seed(31)
    u = (torch.rand(300, 2, 40, 128, generator=g) + 0.5).to(gpu)            # two "slabs" of [300, 40, 128]
    op = R.Burgers(2.0 / 128, 1.25 / 40, 0.002)

    def step(policy):
        jc = pipeline.JointCalibration(300, gpu, prune=policy)
        for sl in range(2):
            jc.add_slab(op.residual(u[:, sl], boundary=True).unsqueeze(1), crop=(0, 1, 1))
        return jc.finish(alphas)

    for policy in ("always", "never"):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step(policy)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            q = step(policy)
        u.copy_((torch.rand(300, 2, 40, 128, generator=g) + 0.5).to(gpu))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(q, step(policy)), policy
    assert torch.allclose(step("always"), step("never"), rtol=1e-6, atol=0.0)


def test_c_abi_client(gpu, tmp_path):
    """tests/c_abi/abi_check.c: a C99 program (gcc, HIP runtime only) drives the library and checks stencil,
    fused NS residual, |a-b| and per-cell order statistics against its own C loops, plus the error codes."""
    import subprocess
    from test_host_cpu import _build_c_client
    exe = _build_c_client(tmp_path / "abi_check")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "all checks passed" in run.stdout, run.stdout + run.stderr


def test_graft_smoke(gpu):
    import __graft_entry__ as ge
    ge.smoke()


def test_zero_copy_surrogate_layout_end_to_end(gpu):
    """[BS,F,Nx,Ny,Nt] surrogate output through permute(0,1,4,2,3) (Marginal/NS_Residuals_CP.py:282):
    the library relabels its axes, the residual keeps the Nt-fastest memory layout (no copy), and
    scores / modulation / q-hat computed on that layout equal the oracle on the logical tensors.
    Also the X-fastest layout and the 1-D [BS,F,Nx,Nt] case."""
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd import pipeline
    from cp_pre_amd.residuals import MHD, Burgers, NavierStokes, PRE_Wave
    from cp_pre_amd.convops_2d import ConvOperator
    from oracle import conformal as oc
    from oracle import residuals as orr
    g = torch.Generator().manual_seed(31)
    sur = torch.rand(48, 6, 20, 24, 16, generator=g) + 0.5                 # [BS,F,Nx,Ny,Nt]
    view = sur.permute(0, 1, 4, 2, 3)                                      # [BS,F,Nt,Nx,Ny], Nt fastest
    dview = sur.to(gpu).permute(0, 1, 4, 2, 3)
    dt, dx, dy = 0.01, 1 / 20, 1 / 24
    res = NavierStokes(dt, dx, dy).residual_momentum(dview[:, :3], boundary=True)
    assert res.stride(1) == 1 and not res.is_contiguous()                 # stayed in the surrogate's layout
    ref = orr.ns_momentum(view[:, :3], dt, dx, dy, boundary=True)
    assert rel_err(res.cpu().numpy(), ref.numpy()) <= RES_TOL
    for name, got, want in [
        ("mhd_induction", MHD().residual_induction(dview, True), orr.mhd_induction(view, boundary=True)),
        ("mhd_energy", MHD().residual_energy(dview, True), orr.mhd_energy(view, boundary=True)),
        ("wave", PRE_Wave(0.01, 0.02).residual(dview[:, 0], True), orr.wave_residual(view[:, 0], 1.0, 0.01, 0.02, boundary=True)),
        ("laplace", ConvOperator(("x", "y"), 2)(dview[:, 1]), __import__("oracle.convops", fromlist=["x"]).ConvOperator2D(("x", "y"), 2)(view[:, 1])),
    ]:
        assert got.stride(1) == 1, name
        assert rel_err(got.cpu().numpy(), want.numpy()) <= RES_TOL, name
    # calibration directly on the permuted residual
    rn = ref.numpy()
    alphas = [0.1, 0.5, 0.9]
    q = icp.calibrate_multi(res.abs(), res.shape[0], alphas)
    assert q.shape == (3,) + tuple(res.shape[1:])
    assert rel_err(q.cpu().numpy(), np.stack([oc.calibrate(np.abs(rn), len(rn), a) for a in alphas])) <= RES_TOL
    exact = torch.from_numpy(rn).to(gpu).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)     # same values, permuted layout
    assert not exact.is_contiguous()
    assert np.array_equal(icp.calibrate(exact.abs(), len(rn), 0.1).cpu().numpy(), oc.calibrate(np.abs(rn), len(rn), 0.1))
    assert np.array_equal(icp.modulation_func(exact, None).cpu().numpy(), oc.modulation_func(rn, np.zeros_like(rn)))
    mod = icp.modulation_func(exact, None)
    inner = rn[:, 1:-1, 1:-1, 1:-1]
    want = oc.ncf_metric_joint(inner, np.zeros_like(inner), oc.modulation_func(inner, np.zeros_like(inner)))
    assert np.array_equal(icp.ncf_metric_joint(exact, None, mod, crop=1).cpu().numpy(), want)
    jc = pipeline.JointCalibration(len(rn), gpu)
    jc.add_slab(exact)
    qj = jc.finish(alphas).cpu().numpy()
    for j, a in enumerate(alphas):
        assert abs(qj[j] - oc.calibrate(want, len(want), a)) <= 1e-6 * abs(qj[j])
    # X-fastest layout [BS,Nt,Ny,Nx] viewed as [BS,Nt,Nx,Ny]
    xf = torch.rand(6, 8, 16, 32, generator=g)
    v2 = xf.permute(0, 1, 3, 2)
    got = ConvOperator("x", 2)(xf.to(gpu).permute(0, 1, 3, 2))
    assert got.stride(2) == 1
    assert rel_err(got.cpu().numpy(), __import__("oracle.convops", fromlist=["x"]).ConvOperator2D("x", 2)(v2).numpy()) <= RES_TOL
    # 1-D surrogate [BS,F,Nx,Nt] -> permute(0,1,3,2)[:,0] (Joint/Burgers_Residuals_CP.py:217)
    s1 = torch.rand(64, 1, 128, 40, generator=g) + 0.5
    u1 = s1.permute(0, 1, 3, 2)[:, 0]
    got = Burgers(2 / 128, 1.25 / 40, 0.002).residual(s1.to(gpu).permute(0, 1, 3, 2)[:, 0], boundary=True)
    assert got.stride(1) == 1
    assert rel_err(got.cpu().numpy(), orr.burgers_residual(u1, 2 / 128, 1.25 / 40, 0.002, boundary=True).numpy()) <= RES_TOL


def test_interior_t_fast_path_equals_full_path(gpu):
    """PRE_FLAG_INTERIOR_T (rim planes neither computed nor stored) + moments over the interior planes
    only: same q-hat, same interior residual, same modulation as the full-slab route."""
    from cp_pre_amd import pipeline
    from cp_pre_amd.residuals import NavierStokes
    g = torch.Generator().manual_seed(17)
    v = (torch.rand(30, 3, 10, 16, 64, generator=g) + 0.5).to(gpu)
    ns = NavierStokes(0.01, 1 / 16, 1 / 64)
    full = ns.residual_momentum(v, boundary=True)
    out = torch.full_like(full, float("nan"))
    fast = ns.residual_momentum(v, boundary=True, out=out, skip_t_rim=True)
    assert torch.equal(fast[:, 1:-1], full[:, 1:-1])
    assert torch.isnan(fast[:, 0]).all() and torch.isnan(fast[:, -1]).all()          # rim untouched
    alphas = [0.1, 0.5, 0.9]
    a = pipeline.JointCalibration(30, gpu)
    a.add_slab(fast, crop=(1, 1, 1))                                                # interior-only moments
    qa = a.finish(alphas)

    class FullOps(pipeline.HipOps):        # the full-slab route: every plane reduced (all 10 are valid in `full`), crop in the score
        interior_planes = staticmethod(lambda res, crop: 0)
    b = pipeline.JointCalibration(30, gpu, ops=FullOps)
    b.add_slab(full, crop=(1, 1, 1))
    qb = b.finish(alphas)
    assert not torch.isnan(b.modulation[0][0]).any()                                # (the rim planes WERE reduced here)
    assert torch.equal(qa, qb)
    assert torch.equal(a.modulation[0][1:-1], b.modulation[0][1:-1]) and torch.isnan(a.modulation[0][0]).all()

    # PRE_FLAG_OUT_INTERIOR_T: the output buffer holds the interior planes only (bench.py's t-slab driver)
    guard = torch.full((30 * 8 * 16 * 64 + 2 * 16 * 64,), float("nan"), device=gpu)
    inner = guard[16 * 64:-16 * 64].view(30, 8, 16, 64)
    got = ns.residual_momentum(v, boundary=True, out=inner, skip_t_rim=True)
    assert got.data_ptr() == inner.data_ptr() and torch.equal(got, full[:, 1:-1])
    assert torch.isnan(guard[:16 * 64]).all() and torch.isnan(guard[-16 * 64:]).all()      # nothing written around it
    assert torch.equal(ns.residual_momentum(v, out=inner, skip_t_rim=True), full[:, 1:-1, 1:-1, 1:-1])
    absd = ns.residual_momentum(v, boundary=True, absolute=True, out=inner, skip_t_rim=True)
    assert torch.equal(absd, full[:, 1:-1].abs())
    c = pipeline.JointCalibration(30, gpu)
    c.add_slab(ns.residual_momentum(v, boundary=True, out=inner, skip_t_rim=True), crop=(0, 1, 1))
    assert torch.equal(c.finish(alphas), qb) and torch.equal(c.modulation[0], b.modulation[0][1:-1])
    # a strided sub-slab of a bigger resident slab (ragged last slab position) and a batch window
    big = (torch.rand(33, 3, 12, 16, 64, generator=g) + 0.5).to(gpu)
    sub = big[2:32, :, :7]
    ref = ns.residual_momentum(sub.contiguous(), boundary=True)
    out5 = torch.empty(30, 5, 16, 64, device=gpu)
    assert torch.equal(ns.residual_momentum(sub, boundary=True, out=out5, skip_t_rim=True), ref[:, 1:-1])
    # not applicable: Nt-fastest views (the skipped rim is on the logical t axis), wrong shapes
    perm = v.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)
    with pytest.raises((RuntimeError, ValueError)):
        ns.residual_momentum(perm, boundary=True, out=torch.empty(30, 8, 16, 64, device=gpu), skip_t_rim=True)
    with pytest.raises(ValueError):
        ns.residual_momentum(v, boundary=True, out=torch.empty(30, 7, 16, 64, device=gpu), skip_t_rim=True)


def test_index_arithmetic_beyond_2_31_elements(gpu):
    """Tensors with more than 2^31 elements (BASELINE C3 slabs have 1e10): the last samples of a big
    batch must equal the same samples evaluated on their own, for the fused residual, the moments /
    joint score and the per-cell select (64-bit offsets everywhere)."""
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd import pipeline
    from cp_pre_amd.residuals import NavierStokes
    B, T, X, Y = 300, 10, 512, 512                       # 300*3*10*512*512 = 2.36e9 elements (9.4 GB)
    assert B * 3 * T * X * Y > 2 ** 31
    v = torch.empty(B, 3, T, X, Y, device=gpu).uniform_(0.5, 1.5)
    ns = NavierStokes(0.01, 1 / X, 1 / Y)
    res = ns.residual_momentum(v, boundary=True)
    tail = ns.residual_momentum(v[-2:].clone(), boundary=True)
    assert torch.equal(res[-2:], tail)
    head = ns.residual_momentum(v[:2].clone(), boundary=True)
    assert torch.equal(res[:2], head)
    # calibration over > 2^31 residual elements: [2200, 1M cells]
    del v, res
    n, M = 2200, 1 << 20
    assert n * M > 2 ** 31
    s = torch.empty(n, M, device=gpu).normal_()
    ks = [0, n // 2, n - 1]
    q = icp.kth_axis0(s, ks)
    for cols in (slice(0, 256), slice(M - 256, M)):
        ref = torch.sort(s[:, cols], dim=0).values[ks]
        assert torch.equal(q[:, cols], ref)
    mom = pipeline.HipOps.zeros_moments(M, gpu)
    pipeline.HipOps.add_moments(s, mom)
    ref_sum = s[:, -128:].double().sum(0)
    assert torch.allclose(mom[0, -128:], ref_sum, rtol=1e-12, atol=1e-9)
    mod = torch.ones(1, 1, M, device=gpu)
    sc = pipeline.HipOps.zeros_scores(n, gpu)
    pipeline.HipOps.max_scores(s.view(n, 1, 1, M), mod, (0, 0, 0), sc)
    assert torch.equal(sc[-3:], s[-3:].abs().amax(1))


# ---------------------------------------------------------------- 2-D spatial family
def test_spatial_family_matches_reference_golden(gpu):
    """Utils/ConvOps_Spatial.py valid conv and the boundary-conditioned vector operators of
    Utils/VectorConvOps_Spatial.py against outputs of the reference itself; the 3x3 cases run the
    fused boundary-mapping kernel, the Taylor-4 ones the pad + stencil fallback."""
    from conftest import load_golden
    from cp_pre_amd import vector_convops_spatial as V
    from cp_pre_amd.convops_spatial import ConvOperator
    from oracle import spatial as osp
    g = load_golden("spatial.npz")
    x, y = torch.from_numpy(g["x"]).to(gpu), torch.from_numpy(g["y"]).to(gpu)
    doms = {"none": None, "xy": ("x", "y")}
    n = 0
    for key in g.files:
        if not key.startswith("conv|"):
            continue
        _, dom, order, taylor, scale = key.split("|")
        op = ConvOperator(doms.get(dom, dom), int(order), scale=float(scale), taylor_order=int(taylor), device=gpu)
        got = op(x)
        assert got.is_cuda and tuple(got.shape) == g[key].shape, key
        assert rel_err(got.detach().cpu().numpy(), g[key]) <= RES_TOL, key
        n += 1
    assert n >= 9
    # the reference's spatial kernels always require grad (scale is a grad-requiring leaf): with grad
    # mode on every call takes the composed, differentiable route; under no_grad the fused kernels run
    for grad in (True, False):
      with torch.set_grad_enabled(grad):
        for bc in ("periodic", "dirichlet", "neumann", "symmetric"):
            for ty in (2, 4):
                L = V.Laplace(scale=1.7, taylor_order=ty, boundary_cond=bc, device=gpu)
                assert rel_err(L(x).detach().cpu().numpy(), g[f"laplace|{bc}|{ty}"]) <= RES_TOL, (bc, ty)
            Lv = V.Laplace(scale=0.5, boundary_cond=bc, scalar=False, device=gpu)
            assert rel_err(Lv(x, y).detach().cpu().numpy(), g[f"laplace_vec|{bc}"]) <= RES_TOL
            D = V.Divergence(scale=2.0, boundary_cond=bc, device=gpu)
            assert rel_err(D(x, y).detach().cpu().numpy(), g[f"divergence|{bc}"]) <= RES_TOL, bc
            C = V.Curl(scale=2.0, boundary_cond=bc, device=gpu)
            assert rel_err(C(x, y).detach().cpu().numpy(), g[f"curl|{bc}"]) <= RES_TOL, bc
            # Gradient / Vector_Gradient (sub-operators live on cuda in the reference): vs the oracle restatement
            G = V.Gradient(scale=1.3, boundary_cond=bc)
            assert rel_err(G(x, y).detach().cpu().numpy(), osp.VectorOp("gradient", 1.3, 2, bc)(x.cpu(), y.cpu()).numpy()) <= RES_TOL
            VG = V.Vector_Gradient(scale=1.3, boundary_cond=bc)
            assert rel_err(VG(x, y).detach().cpu().numpy(), osp.VectorOp("vector_gradient", 1.3, 2, bc)(x.cpu(), y.cpu()).numpy()) <= RES_TOL
    assert torch.equal(V.dot(torch.cat((x, y), 1), torch.cat((y, x), 1)).cpu(), torch.from_numpy(g["dot"]))
    # mixed per-side boundaries and a dirichlet value, larger streaming-sized planes, partial tiles
    gen = torch.Generator().manual_seed(2)
    big = torch.randn(5, 1, 37, 264, generator=gen)
    L = V.Laplace(scale=1.0, device=gpu)
    for spec in [dict(left=("dirichlet", 1.5), right=("neumann", 0.0), top=("symmetric", 0.0), bottom=("periodic", 0.0)),
                 dict(left=("periodic", 0.0), right=("periodic", 0.0), top=("dirichlet", -2.0), bottom=("outflow", 0.0))]:
        types = {k: v[0] for k, v in spec.items()}
        values = {k: v[1] for k, v in spec.items()}
        for side, (t, v) in spec.items():
            L.bc.set_boundary_type(side, t, v)
        ref = osp.conv_valid(osp.pad_signal(big, 3, types, values), L.laplace.kernel.detach().cpu())
        assert rel_err(L(big.to(gpu)).detach().cpu().numpy(), ref.numpy()) <= RES_TOL, spec
        with torch.no_grad():
            assert rel_err(L(big.to(gpu)).cpu().numpy(), ref.numpy()) <= RES_TOL, spec
    with torch.no_grad():
        assert V._fused1(big.to(gpu), L.laplace, L.bc) is not None        # 3x3 cross + mapped boundaries: one fused pass
        L4 = V.Laplace(taylor_order=4, device=gpu)
        assert V._fused1(big.to(gpu), L4.laplace, L4.bc) is None           # 5x5 Taylor stencil: pad + stencil fallback
        Ls = V.Laplace(device=gpu)
        Ls.bc.set_boundary_type("left", "symmetric")                       # left symmetric + right periodic: no fused mapping
        assert V._fused1(big.to(gpu), Ls.laplace, Ls.bc) is None
    # grad mode: the forward is still the fused pass; a backward recomputes through the differentiable pad + stencil
    # recipe (CNS.py trains through these operators), so gradients equal those of the composed route
    Dv = V.Divergence(scale=2.0, device=gpu)
    xg, yg = big.to(gpu).requires_grad_(True), (0.5 * big).to(gpu).requires_grad_(True)
    out = Dv(xg, yg)
    assert out.requires_grad
    out.pow(2).sum().backward()
    xc, yc = big.to(gpu).requires_grad_(True), (0.5 * big).to(gpu).requires_grad_(True)
    Dc = V.Divergence(scale=2.0, device=gpu)        # a second instance: the kernels are non-leaf (scale * stencil) tensors
    (Dc.grad_x(Dc.bc.pad_signal(xc)) + Dc.grad_y(Dc.bc.pad_signal(yc))).pow(2).sum().backward()
    assert rel_err(xg.grad.cpu().numpy(), xc.grad.cpu().numpy()) <= 1e-5
    assert rel_err(yg.grad.cpu().numpy(), yc.grad.cpu().numpy()) <= 1e-5


def test_spectral_family_matches_reference_golden(gpu):
    """conv='spectral' / differentiate / integrate (libcp_pre_fft.so - hipFFT plus fused
    embed / spectrum-multiply / crop kernels) of all three ConvOperator files against outputs of the
    reference's own methods."""
    from conftest import load_golden
    from cp_pre_amd.convops_1d import ConvOperator as Conv1D
    from cp_pre_amd.convops_2d import ConvOperator as Conv2D
    from cp_pre_amd.convops_spatial import ConvOperator as ConvS
    g = load_golden("spectral.npz")
    x4, x3, xs = (torch.from_numpy(g[k]) for k in ("x4", "x3", "xs"))
    cases = {"2d_lap": (Conv2D(("x", "y"), 2), x4), "2d_t1": (Conv2D("t", 1), x4), "1d_x2": (Conv1D("x", 2), x3),
             "1d_xt": (Conv1D(("x", "t"), 2), x3), "sp_lap": (ConvS(("x", "y"), 2, device="cpu"), xs),
             "sp_x1": (ConvS("x", 1, scale=0.5, device="cpu"), xs)}
    n = 0
    for name, (op, x) in cases.items():
        assert np.array_equal(op.kernel.detach().numpy(), g[f"{name}|kernel"])
        with torch.no_grad():
            for key in [k for k in g.files if k.startswith(name + "|") and not k.endswith("kernel")]:
                parts = key.split("|")
                if parts[1] == "spectral":
                    got = op.spectral_convolution(x)
                elif parts[1] == "spectral_inv":
                    got = op.spectral_convolution(x, inverse=True)
                elif parts[1] == "diff":
                    got = op.differentiate(x, correlation=bool(int(parts[2])), slice_pad=bool(int(parts[3])))
                else:
                    got = op.integrate(x, correlation=bool(int(parts[2])), slice_pad=bool(int(parts[3])))
                assert not got.is_cuda and tuple(got.shape) == g[key].shape, key      # CPU in -> hipFFT -> CPU out
                assert rel_err(got.numpy(), g[key]) <= 1e-4, (key, rel_err(got.numpy(), g[key]))
                n += 1
    assert n >= 50
    from oracle.convops import xcorr_torch
    D = Conv2D(("x", "y"), 2)
    got = D.spectral_convolution(x4.to(gpu))
    assert got.is_cuda and torch.allclose(got.cpu(), xcorr_torch(x4, D.kernel), atol=1e-4)


def test_spectral_native_route_equals_torch_fft_composition(gpu):
    """libcp_pre_fft.so against the same recipe composed from torch.fft ops (the differentiable route), on
    odd and even sizes, strided views, 5^3 kernels, and a batch that is staged in several hipFFT chunks."""
    from cp_pre_amd import _spectral as S
    from cp_pre_amd.convops_2d import ConvOperator
    g = torch.Generator().manual_seed(23)
    kernels = [ConvOperator(("x", "y"), 2).kernel, ConvOperator("t", 1).kernel, torch.randn(3, 3, 3, generator=g),
               ConvOperator(("x", "y"), 2, taylor_order=4).kernel]
    fields = [torch.randn(3, 6, 9, 12, generator=g), torch.randn(2, 7, 8, 11, generator=g),
              torch.randn(2, 12, 9, 5, generator=g).permute(0, 3, 2, 1)]
    tol = 2e-5
    for k in kernels:
        for x in fields:
            xd = x.to(gpu)
            a, b = S.fft_xcorr(xd, k), S._torch_fft_xcorr(xd, k)
            assert a.shape == b.shape and rel_err(a.cpu().numpy(), b.cpu().numpy()) <= tol, "xcorr"
            # PRE_FFT_INVERT on a well-conditioned eps (the singular eps=1e-6 case amplifies round-off 1e6 times)
            a = S._native_fft_xcorr_inverse(xd, k, eps=0.3)
            x_pad = torch.nn.functional.pad(xd[:, None], [p for d in reversed(range(3)) for p in (k.shape[d] // 2,) * 2])
            if x_pad.size(-1) % 2:
                x_pad = torch.nn.functional.pad(x_pad, [0, 1])
            kf = torch.fft.rfftn(torch.nn.functional.pad(k.to(gpu)[None, None], [v for i in reversed(range(2, 5))
                                                                                 for v in (0, x_pad.size(i) - k.size(i - 2))]),
                                 dim=(2, 3, 4))
            b = torch.fft.irfftn(torch.fft.rfftn(x_pad, dim=(2, 3, 4)) / (torch.conj(kf) + 0.3), dim=(2, 3, 4))
            b = b[:, 0, :xd.shape[1], :xd.shape[2], :xd.shape[3]]
            assert a.shape == b.shape and rel_err(a.cpu().numpy(), b.cpu().numpy()) <= 1e-4, "xcorr inverse"
            for corr in (False, True):
                for sp in (False, True):
                    a, b = S.differentiate(xd, k, corr, sp), S._torch_differentiate(xd, k, corr, sp)
                    assert a.shape == b.shape and rel_err(a.cpu().numpy(), b.cpu().numpy()) <= tol, ("diff", corr, sp)
                    a, b = S._native_integrate(xd, k, corr, sp, 0.3), S._torch_integrate(xd, k, corr, sp, 0.3)
                    assert a.shape == b.shape and rel_err(a.cpu().numpy(), b.cpu().numpy()) <= 1e-4, ("integ", corr, sp)
    # 1-D file (nd=2) and the spatial family ([B,C,X,Y], channels kept)
    k2 = torch.randn(3, 3, generator=g)
    x3 = torch.randn(4, 9, 16, generator=g).to(gpu)
    assert rel_err(S.fft_xcorr(x3, k2).cpu().numpy(), S._torch_fft_xcorr(x3, k2).cpu().numpy()) <= tol
    x4 = torch.randn(2, 3, 10, 13, generator=g).to(gpu)
    a, b = S.differentiate(x4, k2, True, True, keep_channel=True), S._torch_differentiate(x4, k2, True, True, keep_channel=True)
    assert a.shape == b.shape == (2, 3, 10, 13) and rel_err(a.cpu().numpy(), b.cpu().numpy()) <= tol
    # several hipFFT chunks (ragged last one)
    old = S._STAGE_BYTES
    try:
        S._STAGE_BYTES = 3 * (4 * 8 * 11 * 14 + 8 * 8 * 11 * 8 + 256) + 100          # 3 samples per chunk, batch 7
        x = torch.randn(7, 6, 9, 12, generator=g).to(gpu)
        a = S.fft_xcorr(x, kernels[0])
    finally:
        S._STAGE_BYTES = old
    assert rel_err(a.cpu().numpy(), S._torch_fft_xcorr(x, kernels[0]).cpu().numpy()) <= tol
    # a gradient request takes the torch route and still works
    xg = torch.randn(2, 5, 8, 8, generator=g).to(gpu).requires_grad_(True)
    S.fft_xcorr(xg, kernels[0]).square().sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all()


def test_full_size_properties_c2(gpu):
    """BASELINE config 2 at its full size [512,32,256,256]: two samples against the CPU oracle
    (Marginal/Wave_Residuals_CP.py:170-184; the whole tensor is beyond it in a test) and
    size-independent properties - linearity of the additive wave kernel, |.| idempotence, per-cell
    q-hat non-increasing in alpha and an input value, conformal guarantee on the calibration set
    (at least ceil((n+1)(1-alpha)) of n calibration scores lie within q-hat) for marginal and joint CP."""
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd.residuals import PRE_Wave
    from oracle import residuals as orr
    B, T, X, Y = 512, 32, 256, 256
    g = torch.Generator(device=gpu).manual_seed(3)
    u1 = torch.randn(B, T, X, Y, device=gpu, generator=g)
    u2 = torch.randn(B, T, X, Y, device=gpu, generator=g)
    w = PRE_Wave(dt=0.005, dx=0.01, c=1.0)
    r1, r2 = w.residual(u1, boundary=True), w.residual(u2, boundary=True)
    lin = w.residual(0.75 * u1 - 1.25 * u2, boundary=True)
    assert (lin - (0.75 * r1 - 1.25 * r2)).abs().max().item() <= 1e-5 * lin.abs().max().item()
    a1 = w.residual(u1, boundary=True, absolute=True)
    assert torch.equal(a1, r1.abs()) and torch.equal(a1.abs(), a1)
    for b0 in (0, B - 1):                                                   # two samples against the CPU oracle
        ref = orr.wave_residual(u1[b0:b0 + 1].cpu(), 1.0, 0.005, 0.01, boundary=True)[0]
        assert rel_err(r1[b0].cpu().numpy(), ref.numpy()) <= RES_TOL
    del u2, r2, lin
    n = B
    alphas = [float(a) for a in icp.ALPHA_LEVELS]
    q = icp.calibrate_multi(a1, n, alphas)                                  # [10, T, X, Y]
    assert (q[:-1] >= q[1:]).all()                                          # alpha ascending -> q-hat non-increasing
    for j, a in enumerate(alphas):
        k = icp.kth_index(n, n, a)
        inside = (a1 <= q[j]).sum(0)
        assert int(inside.min()) >= k + 1                                   # q-hat is the (k+1)-th smallest score
    cell = a1[:, 7, 100, 33].contiguous()
    assert torch.equal(q[:, 7, 100, 33], torch.sort(cell).values[[icp.kth_index(n, n, a) for a in alphas]])
    # joint recipe on the same residuals
    mod = icp.modulation_func(r1, None)
    sc = icp.ncf_metric_joint(r1, None, mod, crop=1)
    qj = icp.calibrate_multi(sc, n, alphas)
    assert (qj[:-1] >= qj[1:]).all()
    inner = r1[:, 1:-1, 1:-1, 1:-1]
    for j in (0, 4, 9):
        cov = icp.emp_cov_joint([-(qj[j] * mod)[1:-1, 1:-1, 1:-1], (qj[j] * mod)[1:-1, 1:-1, 1:-1]], inner)
        assert cov >= (icp.kth_index(n, n, alphas[j]) + 1) / n - 2.0 / n    # knife-edge samples sit exactly on the bound


def _sharded_worker(rank, world, port, n_local, shape, out_dir):
    """One of two ranks sharing cuda:0 (gloo rendezvous; RCCL refuses two ranks per device): the product back end
    (pipeline.HipOps) under batch sharding."""
    import torch.distributed as dist
    from cp_pre_amd import pipeline
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        pipeline.HipOps.PRUNE_MIN_CELLS = pipeline.HipOps.PRUNE_MIN_SAMPLES = 0      # the shards take the branch-and-bound score
        full = torch.from_numpy(np.load(os.path.join(out_dir, "res.npy")))
        mine = full[rank * n_local:(rank + 1) * n_local].to(dev)
        alphas = [0.1, 0.3, 0.5, 0.7, 0.9]
        jc = pipeline.JointCalibration(n_local, dev, group=dist.group.WORLD)
        T = shape[0]
        for s in range(2):                                              # two t-slabs with halo planes
            jc.add_slab(mine[:, s * (T - 2) // 2:s * (T - 2) // 2 + (T - 2) // 2 + 2].contiguous(), crop=(1, 1, 1))
        q = jc.finish(alphas)
        qm = pipeline.marginal_qhat(mine.abs().contiguous(), alphas, group=dist.group.WORLD, stage_bytes=4 * n_local * world * 1000)
        # the zero-copy exchange of a time-major score buffer (round 3): plane t to rank t % world, no pack
        tm = pipeline.time_major(n_local, shape, device=dev)
        tm.copy_(mine.abs())
        for ov in (False, True):
            assert torch.equal(pipeline.marginal_qhat(tm, alphas, group=dist.group.WORLD, overlap=ov), qm), ov
        tmp = pipeline.time_major(n_local, shape, pad=64, device=dev)    # padded planes: the pad travels with the plane
        tmp.copy_(mine.abs())
        for ov in (False, True):
            assert torch.equal(pipeline.marginal_qhat(tmp, alphas, group=dist.group.WORLD, overlap=ov), qm), ov
        # bounded receive staging: runs of 2 planes per rank, one select launch per run (pre_kth_axis0_planes_f32)
        two = 2 * 4 * n_local * world * (tmp.stride(0))
        for ov in (False, True):
            assert torch.equal(pipeline.marginal_qhat(tmp, alphas, group=dist.group.WORLD, overlap=ov, stage_bytes=two), qm), ov
        # the surrogate's Nt-fastest layout (memory [n,X,Y,T]) over three slabs, rank 1's samples "wild" (a modulation
        # nothing like the residual's scale: every sample flagged, read whole), rank 0's prunable: the ranks' own read
        # fractions straddle PRUNE_GIVE_UP, the GROUP's decides for both, and the moments all-reduce has the same length
        # on both routes (round-3 advice: rank-local decisions made it M on one rank, M - 2 planes on the other)
        pw = torch.from_numpy(np.load(os.path.join(out_dir, "res_perm.npy")))[rank * n_local:(rank + 1) * n_local].to(dev)
        jp = pipeline.JointCalibration(n_local, dev, group=dist.group.WORLD)
        routes = []
        for s in range(3):
            slab = pw[:, 4 * s:4 * s + 6].permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
            assert not slab.is_contiguous() and pipeline.HipOps.interior_planes(slab, (1, 1, 1)) == 1
            jp.add_slab(slab, crop=(1, 1, 1))
            routes.append(bool(jp.prune))
        qp = jp.finish(alphas)
        mine_frac = jp.score_pass_read_frac()
        np.save(os.path.join(out_dir, f"qp_{rank}.npy"), qp.cpu().numpy())
        np.save(os.path.join(out_dir, f"sp_{rank}.npy"), jp.all_scores.cpu().numpy())
        np.save(os.path.join(out_dir, f"routes_{rank}.npy"), np.array(routes + [mine_frac], dtype=np.float64))
        np.save(os.path.join(out_dir, f"q_{rank}.npy"), q.cpu().numpy())
        np.save(os.path.join(out_dir, f"qm_{rank}.npy"), qm.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [33, 100, 200, 300, 500, 700, 1000, 2049, 5000, 9000, 66000])
def test_marginal_qhat_rows_with_a_pitch(gpu, n):
    """pre_kth_axis0_strided_f32 / pipeline.row_padded: the per-cell select over rows that are further apart than they are
    long (every regime of n: register sort, register tiles, streaming fast and general forms, 32-bit counters) equals the
    select over a contiguous copy, bit for bit; ragged cell counts, a NaN and a tie column included; the fused residual
    kernel writes through the padded view."""
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd import pipeline
    g = torch.Generator(device=gpu).manual_seed(n)
    for cells in ((7, 19), (1000,), (3, 8, 64)) if n < 60000 else ((130,),):
        s = pipeline.row_padded(n, cells, pad=64 if n % 2 else 36, device=gpu)
        assert not s.is_contiguous() and s[0].is_contiguous()
        s.copy_(torch.randn((n,) + cells, device=gpu, generator=g).abs_())
        s.view(n, -1)[: n // 2, 3] = 1.25
        if s[0].numel() > 5:
            s.view(n, -1)[n // 3, 5] = float("nan")
        ks = sorted({0, n // 2, n - 1} | {icp.kth_index(n, n, float(a)) for a in icp.ALPHA_LEVELS if icp.quantile_level(n, float(a)) <= 1})[:10]
        got, want = icp.kth_axis0(s, ks), icp.kth_axis0(s.contiguous(), ks)
        assert got.shape == want.shape and torch.equal(torch.nan_to_num(got, nan=-1.0), torch.nan_to_num(want, nan=-1.0)), (n, cells)
    if n == 300:
        from cp_pre_amd.residuals import NavierStokes
        B, T, X, Y = 21, 6, 16, 64
        v = (torch.rand(B, 3, T, X, Y, device=gpu, generator=g) + 0.5)
        ns = NavierStokes(1e-2, 1.0 / X, 1.0 / Y, nu=1e-3)
        ref = torch.empty(B, T - 2, X, Y, device=gpu)
        ns.residual_momentum(v, boundary=True, absolute=True, out=ref, skip_t_rim=True)
        pad = pipeline.row_padded(B, (T - 2, X, Y), device=gpu)
        ns.residual_momentum(v, boundary=True, absolute=True, out=pad, skip_t_rim=True)
        assert torch.equal(pad, ref)
        assert torch.equal(pipeline.marginal_qhat(pad, [0.1, 0.5]), pipeline.marginal_qhat(ref, [0.1, 0.5]))


@pytest.mark.parametrize("n", [40, 100, 200, 300, 500, 700, 1000, 1500, 2049, 5000, 66000])
def test_multi_plane_select_one_launch(gpu, n):
    """pre_kth_axis0_planes_f32: the per-cell select of several [n, M] score matrices in ONE launch (every regime of n)
    equals one pre_kth_axis0_strided_f32 launch per plane, bit for bit - planes with a pitch between their rows and a
    gap between the planes, a ragged last tile per plane (M % 64 != 0: a plane's last tile must not read its
    neighbour's cells), NaN / tie / constant columns, results written with arbitrary rank and plane strides."""
    from cp_pre_amd import _lib
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd import pipeline
    lib = _lib.load()
    g = torch.Generator(device=gpu).manual_seed(1000 + n)
    for planes, M, pad, gap in ((5, 200, 8, 24), (3, 64, 0, 0), (7, 1, 3, 1)) if n < 60000 else ((3, 130, 6, 10),):
        pitch = M + pad
        ps = n * pitch + gap
        buf = torch.randn(planes * ps, device=gpu, generator=g).abs_()
        view = buf.as_strided((planes, n, M), (ps, pitch, 1))
        view[:, : n // 2, 0] = 1.25                                          # ties
        if M > 5:
            view[1, n // 3, 5] = float("nan")
            view[:, :, 3] = 0.5                                              # a constant column
        if M > 70:
            view[planes - 1, :, 65] *= 1e-30                                 # tiny values in the last plane only
        ks = sorted({0, n // 2, n - 1} | {icp.kth_index(n, n, float(a)) for a in icp.ALPHA_LEVELS
                                          if icp.quantile_level(n, float(a)) <= 1})[:10]
        nk = len(ks)
        want = torch.stack([icp.kth_axis0(view[p], ks) for p in range(planes)], dim=1)       # [nk, planes, M]
        # (a) results as [nk, planes, M]; (b) as [planes, nk, M + 7] with a gap: q_own's layout in the sharded exchange
        out_a = torch.full((nk, planes, M), -7.0, device=gpu)
        out_b = torch.full((planes, nk, M + 7), -7.0, device=gpu)
        kk = _lib.iarr32(ks)
        with torch.cuda.device(gpu):
            _lib.check(lib.pre_kth_axis0_planes_f32(_lib.ptr(buf), ps, pitch, planes, n, M, kk, nk, _lib.ptr(out_a), planes * M, M,
                                                    _lib.stream()), "planes a")
            _lib.check(lib.pre_kth_axis0_planes_f32(_lib.ptr(buf), ps, pitch, planes, n, M, kk, nk, _lib.ptr(out_b), M + 7,
                                                    nk * (M + 7), _lib.stream()), "planes b")
        f = lambda t: torch.nan_to_num(t, nan=-1.0)
        assert torch.equal(f(out_a), f(want)), (n, planes, M)
        assert torch.equal(f(out_b[:, :, :M]), f(want.transpose(0, 1))), (n, planes, M)
        assert (out_b[:, :, M:] == -7.0).all()                               # nothing written beyond a plane's cells
    # argument checks: planes that overlap their neighbours, rows shorter than M
    one = torch.zeros(4 * 8, device=gpu)
    out = torch.zeros(2 * 8, device=gpu)
    k0 = _lib.iarr32([0])
    assert lib.pre_kth_axis0_planes_f32(_lib.ptr(one), 4, 8, 2, 2, 8, k0, 1, _lib.ptr(out), 8, 8, None) == _lib.PRE_E_RANGE
    assert lib.pre_kth_axis0_planes_f32(_lib.ptr(one), 16, 4, 2, 2, 8, k0, 1, _lib.ptr(out), 8, 8, None) == _lib.PRE_E_RANGE
    assert lib.pre_kth_axis0_planes_f32(_lib.ptr(one), 16, 8, 0, 2, 8, k0, 1, _lib.ptr(out), 8, 8, None) == _lib.PRE_E_NULL
    if n == 300:       # the time-major tensor of a slab driver: all planes in one launch == the contiguous copy's q-hat
        tm = pipeline.time_major(n, (6, 10, 33), pad=64, device=gpu)
        tm.copy_(torch.randn(n, 6, 10, 33, device=gpu, generator=g).abs_())
        alphas = [0.1, 0.5, 0.9]
        assert torch.equal(pipeline.marginal_qhat(tm, alphas), pipeline.marginal_qhat(tm.contiguous(), alphas))


@pytest.mark.parametrize("n", [20, 100, 130, 168, 169, 170, 183, 200, 255, 256, 257, 300, 512, 700, 1000, 1500, 2048, 3000, 4096, 4097,
                               7000, 8192, 9216, 9217, 12000, 12288, 12289])
def test_constant_and_nan_columns_alone_in_their_tile(gpu, n):
    """A constant column (settled by its window: an empty candidate list) and a column with one NaN, each in a tile whose
    other cells are ordinary - so that nothing sends the tile to the streaming form - with SMALL ranks among the requested
    ones: a rank below the pick's network size used to select the padding of the empty list (NaN instead of the constant;
    latent in round 3, whose tests always had a tie column in the same tile).  Also +-inf columns, an all-NaN column, a
    column of -0.0 / +0.0, every register-tile regime and both neighbours of it."""
    from cp_pre_amd import inductive_cp as icp
    g = torch.Generator(device=gpu).manual_seed(7000 + n)
    M = 64 * 9 + 17
    s = torch.randn(n, M, device=gpu, generator=g).abs_()
    s[:, 5] = 1.0                                  # tile 0: constant
    s[n // 3, 70] = float("nan")                   # tile 1: one NaN
    s[:, 130] = float("inf")                       # tile 2: a column of +inf
    s[:, 200] = float("nan")                       # tile 3: a column of NaN
    s[:, 260] = -2.5                               # tile 4: constant, negative
    s[:, 330] = 0.0
    s[::2, 330] = -0.0                             # tile 5: zeros of both signs (equal keys apart, equal values)
    s[n // 2, 390] = float("-inf")                 # tile 6: one -inf
    s[:, 450] = 3e38
    s[0, 450] = -3e38                              # tile 7: a window whose width overflows fp32
    ks = sorted({0, 1, min(7, n - 1), min(13, n - 1), n // 2, n - 2, n - 1})
    got = icp.kth_axis0(s, ks)
    want = torch.sort(s, dim=0).values[ks]
    nanmask = torch.isnan(s).any(dim=0)            # np.quantile: a NaN anywhere -> every quantile of the cell is NaN
    assert torch.isnan(got[:, nanmask]).all(), n
    ok = ~nanmask
    assert torch.equal(got[:, ok], want[:, ok]), (n, (got[:, ok] != want[:, ok]).nonzero()[:5].tolist())


@pytest.mark.parametrize("n", [250, 384, 512, 700, 1000, 1500, 2048, 3000, 4096, 6000, 9000, 12000])
def test_register_tile_lists_share_a_pool(gpu, n):
    """Round 5: the candidate lists of a cell are segments of ONE pool (exactly `count` entries each, taken with an atomic
    add on the cell's pool pointer; 144 / 288 entries per cell), tagged in the histogram word of their row.  Quantised
    scores put 12-30 equal-bucket elements behind every rank: lists of every length up to the pick's 31, cells whose ten
    lists together exhaust the pool (the tile then goes to the streaming form), ranks of a cell that share a row (one list,
    two owners), next to ordinary cells - bit-exact against torch.sort in all of them."""
    from cp_pre_amd import inductive_cp as icp
    g = torch.Generator(device=gpu).manual_seed(9100 + n)
    M = 64 * 12 + 5
    s = torch.randn(n, M, device=gpu, generator=g).abs_()
    for j, levels in enumerate((n // 12, n // 20, n // 28, n // 34, 3, 9)):          # elements per level: 12, 20, 28, 34, n/3, n/9
        cols = slice(64 * (2 * j) + 3, 64 * (2 * j) + 40)                               # a run of cells in every other tile
        q = torch.rand(n, cols.stop - cols.start, device=gpu, generator=g)
        s[:, cols] = torch.floor(q * max(levels, 1)) / max(levels, 1) + 0.25
    alphas = [float(a) for a in icp.ALPHA_LEVELS]
    ks = [icp.kth_index(n, n, a) for a in alphas]
    got = icp.kth_axis0(s, ks)
    want = torch.sort(s, dim=0).values[ks]
    assert torch.equal(got, want), (n, (got != want).nonzero()[:5].tolist())
    ks2 = sorted({0, 1, 2, n // 2, n // 2 + 1, n - 3, n - 2, n - 1})                     # neighbouring ranks: shared rows
    assert torch.equal(icp.kth_axis0(s, ks2), torch.sort(s, dim=0).values[ks2])


@pytest.mark.parametrize("n,S", [(300, 20_000_000), (300, 23_000_000), (1100, 10_000_000), (1100, 11_500_000),
                                 (200, 5_000_000), (200, 5_600_000), (512, 21_000_000)])
def test_select_rows_many_megabytes_apart(gpu, n, S):
    """Rows 20-92 MB apart (a [n, M] view of a buffer whose rows are S floats long): the register forms address a group
    of rows by 32-bit scalar byte offsets from one descriptor base (up to 3 x 16 rows x S x 4 B, unsigned: beyond 2^31
    here) and the two-lane / 32-cell forms the second half of a load by a 32-bit lane offset; past their limits (the
    second S of each pair) the dispatch must fall back to a form without them.  Bit-exact against the select of a
    contiguous copy either way."""
    from cp_pre_amd import inductive_cp as icp
    M = 200
    g = torch.Generator(device=gpu).manual_seed(n + S % 1000)
    buf = torch.empty(n * S, device=gpu)
    view = buf.as_strided((n, M), (S, 1))
    view.copy_(torch.randn(n, M, device=gpu, generator=g).abs_())
    view[:, 7] = 2.0
    ks = sorted({0, n // 2, n - 1} | {icp.kth_index(n, n, float(a)) for a in icp.ALPHA_LEVELS if icp.quantile_level(n, float(a)) <= 1})[:10]
    got = icp.kth_axis0(view, ks)
    want = torch.sort(view.contiguous(), dim=0).values[ks]
    assert torch.equal(got, want), (n, S)


def test_time_major_residual_buffer_and_planewise_qhat(gpu):
    """The t-slab driver's residual buffer for sharded marginal CP: memory [T][B][X][Y] handed to the fused kernel as
    an interior-plane ``out`` view [B,T-2,X,Y] (any batch / time strides over dense planes).  Same numbers as the
    contiguous buffer, bit for bit; q-hat of the time-major view (plane by plane, no copy) == q-hat of a copy."""
    from cp_pre_amd import pipeline
    from cp_pre_amd.residuals import NavierStokes
    B, T, X, Y = 37, 9, 24, 256
    g = torch.Generator().manual_seed(77)
    v = (torch.rand(B, 3, T, X, Y, generator=g) + 0.5).to(gpu)
    ns = NavierStokes(1e-2, 1.0 / X, 1.0 / Y, nu=1e-3)
    ref = torch.empty(B, T - 2, X, Y, device=gpu)
    ns.residual_momentum(v, boundary=True, absolute=True, out=ref, skip_t_rim=True)
    tm = pipeline.time_major(B, (T - 2, X, Y), device=gpu)
    tm.fill_(float("nan"))
    got = ns.residual_momentum(v, boundary=True, absolute=True, out=tm, skip_t_rim=True)
    assert got.data_ptr() == tm.data_ptr() and not tm.is_contiguous() and tm.transpose(0, 1).is_contiguous()
    assert torch.equal(tm, ref)
    alphas = [0.1, 0.5, 0.9]
    assert torch.equal(pipeline.marginal_qhat(tm, alphas), pipeline.marginal_qhat(ref, alphas))
    tmp = pipeline.time_major(B, (T - 2, X, Y), pad=64, device=gpu)   # samples of a plane 64 floats further apart
    tmp.fill_(float("nan"))
    ns.residual_momentum(v, boundary=True, absolute=True, out=tmp, skip_t_rim=True)
    assert pipeline._is_time_major(tmp) and tmp.stride(0) == X * Y + 64 and torch.equal(tmp, ref)
    assert torch.equal(pipeline.marginal_qhat(tmp, alphas), pipeline.marginal_qhat(ref, alphas))


@pytest.mark.timeout(300)
def test_sharded_calibration_two_ranks_on_one_gpu(gpu, tmp_path):
    """Joint (all-reduce of moments + all-gather of scores) and marginal (all-to-all in bounded runs) calibration
    with the HIP back end on two batch shards == the single-process result on the whole batch.  The shards score by
    branch and bound (bounds from the all-reduced modulation), the single process with the full pass."""
    import socket
    import torch.multiprocessing as mp
    from cp_pre_amd import pipeline
    world, n_local, shape = 2, 48, (10, 12, 64)
    rng = np.random.default_rng(5)
    res = (rng.standard_normal((world * n_local,) + shape) * (1 + rng.random(shape))).astype(np.float32)
    np.save(tmp_path / "res.npy", res)
    # permuted-layout stream [n, T=14, X=12, Y=64]: rank 0's half smooth in scale, rank 1's half with a per-cell scale
    # that jumps by 1e4 between neighbouring cells (the group's modulation is then far above most of rank 0's cells and
    # the bounds of every sample are useless on rank 1's: flagged, read whole)
    shp = (14, 12, 64)
    rp = rng.standard_normal((world * n_local,) + shp).astype(np.float32)
    rp[n_local:] *= np.where(rng.random(shp) < 0.5, 1.0, 1e4).astype(np.float32)
    np.save(tmp_path / "res_perm.npy", rp)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_sharded_worker, args=(world, port, n_local, shape, str(tmp_path)), nprocs=world, join=True)
    # the permuted stream: same route on both ranks at every slab, same q-hat as one process on the whole batch
    wp = torch.from_numpy(rp).to(gpu)
    jw = pipeline.JointCalibration(world * n_local, gpu, prune=False)
    for s in range(3):
        jw.add_slab(wp[:, 4 * s:4 * s + 6].contiguous(), crop=(1, 1, 1))
    qp_ref = jw.finish([0.1, 0.3, 0.5, 0.7, 0.9]).cpu().numpy()
    r0, r1 = np.load(tmp_path / "routes_0.npy"), np.load(tmp_path / "routes_1.npy")
    assert np.array_equal(r0[:3], r1[:3]), (r0, r1)                            # the group's decision, not each rank's
    for r in range(world):
        qp = np.load(tmp_path / f"qp_{r}.npy")
        assert np.max(np.abs(qp - qp_ref) / np.abs(qp_ref)) <= 1e-6, (qp, qp_ref)
        assert np.allclose(np.load(tmp_path / f"sp_{r}.npy"), jw.all_scores.cpu().numpy(), rtol=1e-5, atol=0.0)
    alphas = [0.1, 0.3, 0.5, 0.7, 0.9]
    whole = torch.from_numpy(res).to(gpu)
    jc = pipeline.JointCalibration(world * n_local, gpu, prune=False)
    T = shape[0]
    for s in range(2):
        jc.add_slab(whole[:, s * (T - 2) // 2:s * (T - 2) // 2 + (T - 2) // 2 + 2].contiguous(), crop=(1, 1, 1))
    q_ref = jc.finish(alphas).cpu().numpy()
    qm_ref = pipeline.marginal_qhat(whole.abs().contiguous(), alphas).cpu().numpy()
    for r in range(world):
        q = np.load(tmp_path / f"q_{r}.npy")
        assert np.max(np.abs(q - q_ref) / np.abs(q_ref)) <= 1e-6, (q, q_ref)       # fp64 moments, different summation split
        assert np.array_equal(np.load(tmp_path / f"qm_{r}.npy"), qm_ref)           # order statistics: exact


def test_marginal_exchange_overlap_on_rccl_at_world_size_one(gpu):
    """The double-buffered, asynchronous form of the sharded marginal exchange (`overlap=True`) on REAL RCCL - at world size
    one, the only size a one-GPU box offers: `all_to_all_single(async_op=True)` then runs on RCCL's own stream, the select
    of run k - 1 on the compute stream, and what orders them is what orders them at any size - the collective waits for the
    work enqueued before it (so a staging buffer is not overwritten under the select that still reads it), `Work.wait()`
    makes the compute stream wait for the collective.  Several runs per tensor (small staging), both layouts; against the
    group-less select, bit for bit.  (`marginal_qhat` itself skips the exchange for a group of one: `_marginal_planes` and
    `_marginal_cells` are called directly.)"""
    import socket
    import torch.distributed as dist
    from cp_pre_amd import pipeline
    created = not dist.is_initialized()
    if created:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=gpu)
    elif not (dist.get_backend() == "nccl" and dist.get_world_size() == 1):
        pytest.skip("a process group of another kind is already up in this process (earlier in-process code left it)")
    try:
        group = dist.group.WORLD
        g = torch.Generator(device=gpu).manual_seed(77)
        alphas = [0.1, 0.5, 0.9]
        n, T, X, Y = 300, 12, 20, 64
        tm = pipeline.time_major(n, (T, X, Y), pad=64, device=gpu)
        tm.copy_(torch.randn(n, T, X, Y, device=gpu, generator=g).abs_())
        want = pipeline.marginal_qhat(tm.contiguous(), alphas)
        one_plane = 4 * n * tm.stride(0)                                   # staging for ONE plane per run: T runs
        for ov in (False, True):
            for stage in (one_plane, 3 * one_plane, 1 << 30):
                got = pipeline._marginal_planes(tm, alphas, group, pipeline.HipOps, ov, stage)
                assert torch.equal(got, want), (ov, stage)
        # the cell-run form (any other layout): pack -> all-to-all -> select, runs of 1000 cells
        dense = tm.contiguous()
        flat_want = want.reshape(len(alphas), -1)
        for ov in (False, True):
            q = pipeline._marginal_cells(dense, alphas, group, pipeline.HipOps, ov, 4 * n * 1000)
            assert torch.equal(q.reshape(len(alphas), -1), flat_want), ov
        # joint CP through the same one-rank RCCL group: the default moments exchange (two reduce-scatters + one all-gather
        # of sigma-hat, issued on RCCL with nothing on the wire) against the all-reduce form and against no group at all
        res = torch.randn(n, T, X, Y, device=gpu, generator=g)
        qs = []
        for kw in (dict(group=group), dict(group=group, moments="all_reduce"), dict()):
            jc = pipeline.JointCalibration(n, gpu, prune=False, **kw)
            mod = jc.add_slab(res, crop=(1, 1, 1))
            qs.append((jc.finish(alphas), mod, jc.all_scores))
        for q, mod, sc in qs[1:]:
            assert torch.equal(q, qs[0][0]) and torch.equal(sc, qs[0][2])
            assert torch.equal(mod[1:-1], qs[0][1][1:-1]) and mod.shape == qs[0][1].shape
        torch.cuda.synchronize()
    finally:
        if created:                                                        # (only what this test created)
            dist.destroy_process_group()


def test_host_scores_reach_the_select_row_padded(gpu, monkeypatch):
    """The reference scripts hand `calibrate` HOST arrays (`ncf_scores = np.abs(res.numpy())`,
    Marginal/Wave_Residuals_CP.py:280-290): the device copy is this package's, so it is uploaded into rows 64 floats further
    apart than they are long whenever the row length is a multiple of a large power of two (the per-cell select runs 2.5
    instead of 3.4 TB/s on such a pitch at n = 2048: profiles/r05/select_scan.txt) - through the compat import path, a numpy
    [2048, 2^19] array: bit-exact against numpy's own quantile, and the select is entered on the strided view (pitch M + 64).
    Caller-owned DEVICE tensors are selected where they lie (dense pitch), row-padded ones with their pad."""
    import sys
    from conftest import ROOT
    from cp_pre_amd import inductive_cp as icp
    from oracle import conformal as oc
    sys.path.insert(0, os.path.join(ROOT, "cp_pre_amd", "compat"))
    try:
        from Neural_PDE.UQ.inductive_cp import calibrate
    finally:
        sys.path.remove(os.path.join(ROOT, "cp_pre_amd", "compat"))
    seen = []
    real = icp.rows_where_they_lie

    def spy(scores):
        r = real(scores)
        seen.append((tuple(scores.shape), scores.stride(0), None if r is None else r[1]))
        return r
    monkeypatch.setattr(icp, "rows_where_they_lie", spy)
    n, M = 2048, 1 << 19
    rng = np.random.default_rng(19)
    host = np.abs(rng.standard_normal((n, M), dtype=np.float32))
    q = calibrate(scores=host, n=len(host), alpha=0.1)
    assert isinstance(q, np.ndarray) and q.shape == (M,) and q.dtype == np.float32
    assert seen[-1] == ((n, M), M + 64, M + 64), seen[-1]                   # entered on the padded rows
    k = icp.kth_index(n, n, 0.1)
    cols = rng.integers(0, M, size=64)
    assert np.array_equal(q[cols], np.sort(host[:, cols], axis=0)[k])       # (numpy's quantile on all 2^19 columns takes minutes)
    assert np.array_equal(q[:4096], oc.calibrate(host[:, :4096], n, 0.1))
    # a CPU torch tensor [n, T, X, Y] takes the same road and comes back on the CPU
    t4 = torch.from_numpy(host[:512]).reshape(512, 8, 256, 256)
    q4 = icp.calibrate_multi(t4, 512, [0.1, 0.5])
    assert seen[-1][1:] == (M + 64, M + 64) and q4.device.type == "cpu" and q4.shape == (2, 8, 256, 256)
    assert torch.equal(q4.reshape(2, -1)[:, cols], torch.sort(t4.reshape(512, -1)[:, cols], dim=0).values[[icp.kth_index(512, 512, 0.1), icp.kth_index(512, 512, 0.5)]])
    # short rows, few rows, odd row lengths: dense upload (nothing to gain)
    for shape in ((2048, 1000), (200, 1 << 19), (300, 3 * (1 << 14) + 4)):
        h = np.abs(rng.standard_normal(shape, dtype=np.float32))
        qh = calibrate(h, shape[0], 0.5)
        assert seen[-1][2] is None and np.array_equal(qh[:50], oc.calibrate(h[:, :50], shape[0], 0.5)), shape
    # the caller's dense device tensor: selected where it lies
    dev = torch.from_numpy(host[:600]).to(gpu)
    qd = calibrate(dev, 600, 0.1)
    assert seen[-1] == ((600, M), M, None) and qd.is_cuda
    assert torch.equal(qd.cpu()[cols], torch.sort(torch.from_numpy(host[:600][:, cols]), dim=0).values[icp.kth_index(600, 600, 0.1)])


def test_score_outputs_of_the_residuals_are_row_padded_and_selected_in_place(gpu):
    """`residual(..., absolute=True)` is this package's own extension (the |.| epilogue of the marginal score): its output is
    a score matrix about to be selected along the batch axis, so it is allocated row-padded when the row length asks for
    it - in the fields' own memory order (Ny fastest, or the surrogate's Nt fastest) - and `kth_axis0` / `marginal_qhat`
    select it where it lies (no copy: `rows_where_they_lie`).  Values identical to |signed residual|."""
    from cp_pre_amd import _lib
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd import pipeline
    from cp_pre_amd import residuals as R
    B, T, X, Y = 256, 8, 64, 64                                               # M = 2^15 cells per sample
    g = torch.Generator(device=gpu).manual_seed(3)
    M = T * X * Y
    assert _lib.wants_row_pad(B, M) and not _lib.wants_row_pad(240, M) and not _lib.wants_row_pad(B, M + 4)
    alphas = [0.1, 0.5, 0.9]
    for layout in ("ny", "nt"):
        sur = torch.rand(B, 6, X, Y, T, device=gpu, generator=g) + 0.5
        v = sur.permute(0, 1, 4, 2, 3) if layout == "nt" else sur.permute(0, 1, 4, 2, 3).contiguous()
        for name, fn in (("ns", lambda a: R.NavierStokes(0.01, 1 / X, 1 / Y).residual_momentum(v[:, :3], True, absolute=a)),
                         ("induction", lambda a: R.MHD().residual_induction(v, True, absolute=a)),
                         ("gauss", lambda a: R.MHD().residual_gauss(v, True, absolute=a)),
                         ("wave", lambda a: R.PRE_Wave(0.01, 0.02).residual(v[:, 0], True, absolute=a))):
            signed, a = fn(False), fn(True)
            assert signed.stride(0) == M and a.stride(0) == M + 64, (layout, name)
            assert a.stride()[1:] == signed.stride()[1:] and torch.equal(a, signed.abs()), (layout, name)
            lie = icp.rows_where_they_lie(a)
            assert lie is not None and lie[1] == M + 64 and lie[0].data_ptr() == a.data_ptr()
            q = pipeline.marginal_qhat(a, alphas)
            want = torch.sort(signed.abs().contiguous(), dim=0).values[[icp.kth_index(B, B, al) for al in alphas]]
            assert q.shape == want.shape and torch.equal(q, want), (layout, name)
    # a batch below the register-tile sizes stays dense; so does a CPU input's result (it goes home as a dense tensor)
    small = R.MHD().residual_induction(v[:200], True, absolute=True)
    assert small.stride(0) == M
    home = R.MHD().residual_induction(v.cpu(), True, absolute=True)
    assert home.device.type == "cpu" and torch.equal(home, R.MHD().residual_induction(v, True).abs().cpu())
    # out= on the MHD entries: a plane-major (time-major) buffer, the send blocks of the sharded marginal exchange
    vc = v.contiguous()
    tm = pipeline.time_major(B, (T, X, Y), pad=64, device=gpu)
    got = R.MHD().residual_induction(vc, True, absolute=True, out=tm)
    assert got.data_ptr() == tm.data_ptr() and torch.equal(tm, R.MHD().residual_induction(vc, True).abs())
    with pytest.raises(ValueError):
        R.MHD().residual_induction(vc, True, out=torch.empty(B, T, X, Y + 4, device=gpu))


@pytest.mark.parametrize("T", [10, 20, 30, 64])
def test_short_nt_surrogate_layout_flat_form(gpu, T):
    """The surrogate's native [BS,F,Nx,Ny,Nt] layout with the reference's T_out values (20, 30, 40; 10 = a C3 slab):
    Nt is the contiguous axis and is short, so the flat (merged-axis) form of the marching kernel runs - fused
    residuals and single operators, zero-copy, output in the input's memory order."""
    from cp_pre_amd import residuals as R
    from cp_pre_amd.convops_2d import ConvOperator
    from oracle import residuals as orr
    from oracle.cstencil import xcorr_c
    g = torch.Generator().manual_seed(100 + T)
    phys = torch.rand(3, 6, 18, 26, T, generator=g) + 0.5                 # [BS,F,Nx,Ny,Nt]
    v = phys.permute(0, 1, 4, 2, 3)                                       # Marginal/NS_Residuals_CP.py:282
    vd = phys.to(gpu).permute(0, 1, 4, 2, 3)
    dt, dx, dy = 0.01, 0.05, 0.04
    cases = [(R.NavierStokes(dt, dx, dy).residual_momentum(vd[:, :3], True), orr.ns_momentum(v[:, :3], dt, dx, dy, boundary=True)),
             (R.NavierStokes(dt, dx, dy).residual_continuity(vd[:, :2], True), orr.ns_continuity(v[:, :2], dx, dy, boundary=True)),
             (R.MHD().residual_induction(vd, True), orr.mhd_induction(v, boundary=True)),
             (R.MHD().residual_energy(vd, True), orr.mhd_energy(v, boundary=True)),
             (R.PRE_Wave(dt=dt, dx=dx, c=1.0).residual(vd[:, :1], boundary=True), orr.wave_residual(v[:, 0], 1.0, dt, dx, boundary=True))]
    for i, (got, want) in enumerate(cases):
        assert got.stride()[1:] == vd[:, 0].stride()[1:], i               # same memory order as the input view (dense batch)
        assert rel_err(got.cpu().numpy(), want.numpy()) <= RES_TOL, (T, i)
    for dom, order in (("t", 1), ("x", 2), (("x", "y"), 2)):
        D = ConvOperator(dom, order)
        got = D(vd[:, 0])
        assert rel_err(got.cpu().numpy(), xcorr_c(v[:, 0].contiguous().numpy(), D.kernel.numpy())) <= RES_TOL, (T, dom)
    # the reference layout with a narrow grid takes the same form
    small = torch.rand(2, 3, 7, 22, max(T // 2, 6), generator=g) + 0.5
    got = R.NavierStokes(dt, dx, dy).residual_momentum(small.to(gpu), True)
    assert rel_err(got.cpu().numpy(), orr.ns_momentum(small, dt, dx, dy, boundary=True).numpy()) <= RES_TOL


# ---------------------------------------------------------------- x-slabs (round 3): PRE_FLAG_HALO_X
@pytest.mark.parametrize("X,Y,x0,x1", [(40, 256, 1, 17), (40, 256, 8, 39), (21, 256, 5, 12), (30, 128, 3, 14), (40, 64, 5, 38)])
def test_x_slab_with_halo_rows_equals_the_whole_grid_rows(gpu, X, Y, x0, x1):
    """An x-slab driver hands the fused kernel the rows [x0, x1) of a larger grid and says that the rows x0-1 and x1
    exist (``halo_x`` -> PRE_FLAG_HALO_X): the slab's residual is then the whole grid's residual on those rows, bit for
    bit (whole and partial 8-row tiles, the 16x32 and 32x16 tile shapes, |.|, an out buffer, the skipped t rim)."""
    from cp_pre_amd import _dispatch, _lib
    from cp_pre_amd.convops_2d import ConvOperator
    from cp_pre_amd.residuals import NavierStokes
    B, T = 5, 7
    g = torch.Generator().manual_seed(X * 1000 + Y + x0)
    v = (torch.rand(B + 1, 3, T, X, Y, generator=g) + 0.5).to(gpu)[:B]      # (memory beyond the last sample's last row)
    ns = NavierStokes(1e-2, 1.0 / X, 1.0 / Y, nu=1e-3, device=gpu)
    full = ns.residual_momentum(v, boundary=True)
    slab = ns.residual_momentum(v[:, :, :, x0:x1], boundary=True, halo_x=True)
    assert slab.shape == (B, T, x1 - x0, Y) and torch.equal(slab, full[:, :, x0:x1])
    zero = ns.residual_momentum(v[:, :, :, x0:x1], boundary=True)          # without the flag: zero padding at the cut
    assert not torch.equal(zero[:, :, 0], full[:, :, x0]) and torch.equal(zero[:, :, 1:-1], full[:, :, x0 + 1:x1 - 1])
    out = torch.full((B, T, x1 - x0, Y), float("nan"), device=gpu)
    got = ns.residual_momentum(v[:, :, :, x0:x1], boundary=True, absolute=True, out=out, skip_t_rim=True, halo_x=True)
    assert got.data_ptr() == out.data_ptr() and torch.equal(out[:, 1:-1], full[:, 1:-1, x0:x1].abs())
    # a single star-shaped operator through pre_stencil3d_f32 (the wave's additive kernel)
    D = ConvOperator(("x", "y"), 2, device=gpu)
    D.kernel = D.kernel + 0.25 * ConvOperator("t", 2, device=gpu).kernel
    u = v[:, 0]
    assert torch.equal(_dispatch._xcorr_impl(u[:, :, x0:x1], D.kernel, 3, flags=_lib.PRE_FLAG_HALO_X), D(u)[:, :, x0:x1])


@pytest.mark.parametrize("eq", ["continuity", "momentum", "energy", "induction"])
def test_mhd_x_slab_with_halo_rows_equals_the_whole_grid_rows(gpu, eq):
    """The MHD residuals on an x-slab with its halo rows (float-per-thread halo for induction / continuity, float4 halo
    rows for the six-field functors; a partial last tile) == the whole grid's rows, bit for bit."""
    from cp_pre_amd.residuals import MHD
    g = torch.Generator().manual_seed(11)
    v = (torch.rand(4, 6, 6, 37, 256, generator=g) + 0.5).to(gpu)[:3]
    mhd = MHD(device=gpu)
    fn = getattr(mhd, "residual_" + eq)
    full = fn(v, boundary=True)
    for x0, x1 in ((1, 17), (9, 36), (20, 27)):
        slab = fn(v[:, :, :, x0:x1], boundary=True, halo_x=True)
        assert torch.equal(slab, full[:, :, x0:x1]), (x0, x1)
        assert torch.equal(fn(v[:, :, :, x0:x1], boundary=True, absolute=True, halo_x=True), full[:, :, x0:x1].abs())


def test_wave_x_slab_with_halo_rows_equals_the_whole_grid_rows(gpu):
    from cp_pre_amd.residuals import PRE_Wave
    g = torch.Generator().manual_seed(12)
    u = torch.randn(4, 9, 45, 256, generator=g).to(gpu)[:3]
    w = PRE_Wave(dt=0.005, dx=0.01, c=1.0, device=gpu)
    full = w.residual(u, boundary=True)
    for x0, x1 in ((1, 9), (7, 44)):
        assert torch.equal(w.residual(u[:, :, x0:x1], boundary=True, halo_x=True), full[:, :, x0:x1])
        assert torch.equal(w.residual(u[:, :, x0:x1], boundary=True, absolute=True, halo_x=True), full[:, :, x0:x1].abs())


def test_x_slab_flag_is_refused_where_no_kernel_reads_the_halo(gpu):
    """PRE_E_UNSUPPORTED (never a silently zero-padded result): off-star taps, a width that leaves tail columns, the
    1-D entry, a T-contiguous (relabelled) view."""
    from cp_pre_amd import _dispatch, _lib
    from cp_pre_amd.convops_1d import ConvOperator as C1
    from cp_pre_amd.convops_2d import ConvOperator
    from cp_pre_amd.residuals import NavierStokes
    g = torch.Generator().manual_seed(3)
    u = torch.rand(3, 6, 20, 66, generator=g).to(gpu)
    dense = torch.rand(3, 3, 3, generator=g)
    for field, k, nd in ((u[:, :, 2:9, :64], dense, 3), (u[:, :, 2:9], ConvOperator("x", 2, device=gpu).kernel, 3),
                         (u[0][:, 2:9, :64], C1("x", 2, device=gpu).kernel, 2)):
        with pytest.raises(RuntimeError, match="-3"):
            _dispatch._xcorr_impl(field, k, nd, flags=_lib.PRE_FLAG_HALO_X)
    v = (torch.rand(2, 3, 16, 16, 8, generator=g) + 0.5).to(gpu).permute(0, 1, 4, 2, 3)        # Nt fastest
    with pytest.raises((RuntimeError, ValueError)):
        NavierStokes(1e-2, 1 / 16, 1 / 16, device=gpu).residual_momentum(v[:, :, :, 2:9], boundary=True, halo_x=True)


@pytest.mark.parametrize("B,T,X,Y,rows", [(40, 9, 34, 64, 16), (300, 18, 130, 512, 64)])
def test_x_slab_stream_equals_whole_grid_calibration(gpu, B, T, X, Y, rows):
    """The x-slab driver of bench.py (rows [x0, x1) + halo rows, T whole, crop (1, 0, 1)) against the whole grid in one
    piece (crop (1, 1, 1)): the same joint scores / q-hat (moment sums in another order: 1e-6) and, slab by slab, the
    whole grid's per-cell q-hat bit for bit.  The second shape is large enough for the branch-and-bound score pass."""
    from cp_pre_amd import pipeline
    from cp_pre_amd.residuals import NavierStokes
    g = torch.Generator().manual_seed(B + X)
    v = (torch.rand(B, 3, T, X, Y, generator=g) * 0.2 + 0.9).to(gpu)
    ns = NavierStokes(1e-2, 1.0 / X, 1.0 / Y, nu=1e-3, device=gpu)
    alphas = [0.1, 0.5, 0.9]
    whole = ns.residual_momentum(v, boundary=True)
    jc0 = pipeline.JointCalibration(B, gpu, prune=False)
    jc0.add_slab(whole, crop=(1, 1, 1))
    q0 = jc0.finish(alphas)
    qm0 = pipeline.marginal_qhat(whole.abs(), alphas)
    jc = pipeline.JointCalibration(B, gpu)
    assert (X - 2) % rows == 0
    buf = torch.empty(B, T, rows, Y, device=gpu)
    for x0 in range(1, X - 1, rows):
        res = ns.residual_momentum(v[:, :, :, x0:x0 + rows], boundary=True, out=buf, halo_x=True)
        assert torch.equal(res, whole[:, :, x0:x0 + rows])
        if B >= 256:
            assert pipeline.HipOps.can_prune(res, (1, 0, 1))
        jc.add_slab(res, crop=(1, 0, 1))
        assert torch.equal(pipeline.marginal_qhat(res.abs(), alphas), qm0[:, :, x0:x0 + rows])
    q = jc.finish(alphas)
    assert torch.allclose(jc.scores, jc0.scores, rtol=1e-6, atol=0) and torch.allclose(q, q0, rtol=QHAT_TOL, atol=0)


def test_row_stride_beyond_32_bit_offsets_takes_the_generic_route(gpu):
    """The streaming kernels address a plane by 32-bit byte offsets; a view whose rows lie further apart than that (here
    64 M floats: 256 MB per row) is declined by them (PRE_E_UNSUPPORTED) and must still come out right - single operators
    through the strided generic kernel, fused residuals through the composed route."""
    from cp_pre_amd.convops_2d import ConvOperator
    from cp_pre_amd.residuals import NavierStokes
    pitch = 1 << 26
    big = torch.zeros(6 * pitch + 256, device=gpu)
    g = torch.Generator().manual_seed(5)
    vals = (torch.rand(1, 3, 2, 2, 256, generator=g) + 0.5).to(gpu)              # [B, F, T, X, Y] with X = 2 rows per plane
    view = big.as_strided((1, 3, 2, 2, 256), (0, 2 * pitch, 256, pitch, 1))       # rows 2^26 floats apart
    view.copy_(vals)
    D = ConvOperator(("x", "y"), 2, device=gpu)
    assert torch.equal(D(view[:, 0]), D(vals[:, 0].contiguous()))
    ns = NavierStokes(1e-2, 0.1, 0.1, nu=1e-3, device=gpu)
    a, b = ns.residual_momentum(view, boundary=True), ns.residual_momentum(vals, boundary=True)
    assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= RES_TOL


def test_plane_offsets_between_2_and_4_gigabytes_stream(gpu):
    """The streaming kernels' 32-bit plane offsets are UNSIGNED: a view whose last rows lie more than 2^31 bytes into
    the plane (16 rows 128 MB apart) still takes the streaming kernel and comes out bit-identical to the dense copy."""
    from cp_pre_amd.convops_2d import ConvOperator
    pitch = 1 << 25
    X, Y, T = 16, 256, 2
    big = torch.zeros(T * X * pitch + 256, device=gpu)
    g = torch.Generator().manual_seed(6)
    vals = torch.randn(1, T, X, Y, generator=g).to(gpu)
    view = big.as_strided((1, T, X, Y), (0, X * pitch, pitch, 1))
    view.copy_(vals)
    D = ConvOperator(("x", "y"), 2, device=gpu)
    D.kernel = D.kernel + 0.5 * ConvOperator("t", 2, device=gpu).kernel
    assert torch.equal(D(view), D(vals.contiguous()))

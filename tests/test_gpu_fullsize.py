"""BASELINE.json configurations at their FULL per-GPU sizes (far beyond what the CPU oracle can check in a test):
size-independent properties of the hot path, through the same classes bench.py drives.

  C3 slab  [4096,10,512,512] x3 fields   NS momentum residual (Marginal/NS_Residuals_CP.py:231-240), joint + marginal CP
  C4 shard [1024,64,256,256] x6 fields   MHD induction residual (Marginal/MHD_Residuals_CP.py:259-268), joint CP
  C5 shard [8192,200,512]                Burgers residual (Joint/Burgers_Residuals_CP.py:182-187), joint + marginal CP
  C5 whole [65536,200,512]               the same at its N = 1 size (27 GB in, 27 GB residual): the anchor of the 1 -> 8 curve

Properties: the last samples of the big batch equal the same samples evaluated on their own (64-bit indexing, grid
decomposition); |.| epilogue == abs of the signed result; the interior-plane fast paths equal the full path; q-hat is
non-increasing in alpha and an input value; the conformal guarantee holds on the calibration set (at least k+1 of the
n calibration scores lie within q-hat, k = the rank calibrate selects)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    free = torch.cuda.mem_get_info()[0]
    # three BASELINE configurations ride on these tests: a box that cannot hold them is a FAILED run, not a skipped one
    # (under `-m gpu` a skip would silently un-exercise C3, C4 and C5)
    assert free >= 250e9, f"the full-size tests need ~250 GB of free HBM, {free / 1e9:.0f} GB available"
    return torch.device("cuda:0")


def _alphas():
    from cp_pre_amd import inductive_cp as icp
    return [float(a) for a in icp.ALPHA_LEVELS]


def _check_joint(icp, pipeline, res, crop, n, gpu):
    """Streaming joint calibration on `res` + the guarantees; returns (q, modulation, scores)."""
    alphas = _alphas()
    jc = pipeline.JointCalibration(n, gpu)
    assert pipeline.HipOps.can_prune(res, crop)                         # these sizes take the branch-and-bound score pass
    mod = jc.add_slab(res, crop=crop)
    q = jc.finish(alphas)
    sc = jc.all_scores
    full = pipeline.JointCalibration(n, gpu, prune=False)               # ... which must agree with the full pass
    mod_full = full.add_slab(res, crop=crop)
    q_full = full.finish(alphas)
    assert torch.allclose(mod, mod_full, rtol=1e-6, atol=0.0, equal_nan=True)    # (fp64 sums in a different order)
    assert torch.allclose(sc, full.all_scores, rtol=1e-6, atol=0.0) and torch.allclose(q, q_full, rtol=1e-6, atol=0.0)
    assert q.shape == (len(alphas),) and torch.isfinite(q).all() and (q[:-1] >= q[1:]).all()
    srt = torch.sort(sc).values
    for j, a in enumerate(alphas):
        k = icp.kth_index(n, n, a)
        assert q[j] == srt[k]                                           # an input value, the (k+1)-th smallest
        assert int((sc <= q[j]).sum()) >= k + 1                         # conformal guarantee on the calibration set
    return q, mod, sc


@pytest.mark.timeout(300)
def test_full_size_c3_slab(gpu):
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd import pipeline
    from cp_pre_amd.residuals import NavierStokes
    B, T, X, Y = 4096, 10, 512, 512
    alphas = _alphas()
    v = torch.empty(B, 3, T, X, Y, device=gpu)
    g = torch.Generator(device=gpu).manual_seed(11)
    v.uniform_(0.5, 1.5, generator=g)
    ns = NavierStokes(1e-2, 1.0 / X, 1.0 / Y, nu=1e-3)
    full = ns.residual_momentum(v, boundary=True)
    assert full.shape == (B, T, X, Y) and torch.isfinite(full[::512]).all()
    # the tail of the batch alone (offsets beyond 2^31 elements: B*3*T*X*Y = 3.2e10)
    tail = ns.residual_momentum(v[-3:].clone(), boundary=True)
    assert torch.equal(full[-3:], tail)
    mid = ns.residual_momentum(v[2047:2049].clone(), boundary=True)
    assert torch.equal(full[2047:2049], mid)
    # interior-plane fast paths == full path
    inner = torch.empty(B, T - 2, X, Y, device=gpu)
    got = ns.residual_momentum(v, boundary=True, out=inner, skip_t_rim=True)
    assert got.data_ptr() == inner.data_ptr()
    for b0 in range(0, B, 1024):
        assert torch.equal(inner[b0:b0 + 1024], full[b0:b0 + 1024, 1:-1])
    # joint CP on the interior planes == joint CP on the uncropped slab with a t crop
    q1, mod1, sc1 = _check_joint(icp, pipeline, inner, (0, 1, 1), B, gpu)
    q2, mod2, sc2 = _check_joint(icp, pipeline, full, (1, 1, 1), B, gpu)
    assert torch.equal(q1, q2) and torch.equal(sc1, sc2) and torch.equal(mod1, mod2[1:-1])
    del full, v
    # marginal CP: |res| epilogue, per-cell q-hat over the 4096 samples
    v = torch.empty(B, 3, T, X, Y, device=gpu)
    v.uniform_(0.5, 1.5, generator=torch.Generator(device=gpu).manual_seed(11))
    absd = torch.empty_like(inner)
    ns.residual_momentum(v, boundary=True, absolute=True, out=absd, skip_t_rim=True)
    del v
    for b0 in range(0, B, 1024):
        assert torch.equal(absd[b0:b0 + 1024], inner[b0:b0 + 1024].abs())
    del inner
    q = pipeline.marginal_qhat(absd, alphas)                                # [10, T-2, X, Y]
    assert q.shape == (len(alphas), T - 2, X, Y) and (q[:-1] >= q[1:]).all()
    for j in (0, 5, 9):
        k = icp.kth_index(B, B, alphas[j])
        inside = torch.zeros(T - 2, X, Y, dtype=torch.int32, device=gpu)
        for b0 in range(0, B, 512):
            inside += (absd[b0:b0 + 512] <= q[j]).sum(0, dtype=torch.int32)
        assert int(inside.min()) >= k + 1
    for (t, x, y) in ((0, 0, 0), (3, 255, 77), (7, 511, 511)):
        col = torch.sort(absd[:, t, x, y].contiguous()).values
        assert torch.equal(q[:, t, x, y], col[[icp.kth_index(B, B, a) for a in alphas]])


@pytest.mark.timeout(300)
def test_full_size_c4_shard(gpu):
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd import pipeline
    from cp_pre_amd.residuals import MHD
    B, T, X, Y = 1024, 64, 256, 256
    v = torch.empty(B, 6, T, X, Y, device=gpu)
    v.uniform_(0.5, 1.5, generator=torch.Generator(device=gpu).manual_seed(12))
    mhd = MHD()
    res = mhd.residual_induction(v, boundary=True)
    assert res.shape == (B, T, X, Y) and torch.isfinite(res[::128]).all()
    assert torch.equal(res[-2:], mhd.residual_induction(v[-2:].clone(), boundary=True))
    assert torch.equal(res[511:513], mhd.residual_induction(v[511:513].clone(), boundary=True))
    a = mhd.residual_induction(v, boundary=True, absolute=True)
    assert torch.equal(a, res.abs())
    crop_view = mhd.residual_induction(v[:4])                               # boundary=False: the reference's crop
    assert torch.equal(crop_view, res[:4, 1:-1, 1:-1, 1:-1])
    del a, v
    q, mod, sc = _check_joint(icp, pipeline, res, (1, 1, 1), B, gpu)
    # bounds +-q*mod cover at least the guaranteed share of the calibration samples jointly
    alphas = _alphas()
    inner = res[:, 1:-1, 1:-1, 1:-1]
    m = mod[1:-1, 1:-1, 1:-1]
    for j in (0, 9):
        cov = icp.emp_cov_joint([-(q[j] * m), q[j] * m], inner)
        assert cov >= (icp.kth_index(B, B, alphas[j]) + 1) / B - 2.0 / B    # knife-edge samples sit exactly on the bound


@pytest.mark.timeout(300)
def test_full_size_c5_shard(gpu):
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd import pipeline
    from cp_pre_amd.residuals import Burgers
    B, T, X = 8192, 200, 512
    alphas = _alphas()
    u = torch.empty(B, T, X, device=gpu)
    u.uniform_(0.5, 1.5, generator=torch.Generator(device=gpu).manual_seed(13))
    bur = Burgers(2.0 / X, 1.25 / T, 0.002)
    res = bur.residual(u, boundary=True)
    assert res.shape == (B, T, X) and torch.isfinite(res).all()
    assert torch.equal(res[-3:], bur.residual(u[-3:].clone(), boundary=True))
    assert torch.equal(bur.residual(u[:5]), res[:5, 1:-1, 1:-1])
    a = bur.residual(u, boundary=True, absolute=True)
    assert torch.equal(a, res.abs())
    q, mod, sc = _check_joint(icp, pipeline, res.unsqueeze(1), (0, 1, 1), B, gpu)
    m = mod[0, 1:-1, 1:-1]
    for j in (0, 9):
        cov = icp.emp_cov_joint([-(q[j] * m), q[j] * m], res[:, 1:-1, 1:-1])
        assert cov >= (icp.kth_index(B, B, alphas[j]) + 1) / B - 2.0 / B
    # marginal: per-cell q-hat over the 8192 samples of the shard
    qm = pipeline.marginal_qhat(a, alphas)
    assert qm.shape == (len(alphas), T, X) and (qm[:-1] >= qm[1:]).all()
    for j in (0, 9):
        assert int((a <= qm[j]).sum(0).min()) >= icp.kth_index(B, B, alphas[j]) + 1
    col = torch.sort(a[:, 100, 257].contiguous()).values
    assert torch.equal(qm[:, 100, 257], col[[icp.kth_index(B, B, al) for al in alphas]])


@pytest.mark.timeout(600)
def test_full_size_c5_whole(gpu):
    """BASELINE config 5 at its single-GPU size [65536,200,512] (Joint/Burgers_Residuals_CP.py:182-187,272-285): 65 536
    calibration samples resident at once - the select's 32-bit-counter instantiation (n >= 65536), sample offsets
    beyond 2^32 bytes, the branch-and-bound score pass over one 200 x 512 plane per sample."""
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd import pipeline
    from cp_pre_amd.residuals import Burgers
    B, T, X = 65536, 200, 512
    alphas = _alphas()
    u = torch.empty(B, T, X, device=gpu)
    u.uniform_(0.5, 1.5, generator=torch.Generator(device=gpu).manual_seed(14))
    bur = Burgers(2.0 / X, 1.25 / T, 0.002)
    res = bur.residual(u, boundary=True)
    assert res.shape == (B, T, X) and torch.isfinite(res[::4096]).all()
    for b0 in (0, 32767, B - 3):                                            # (byte offsets 0, 13.4e9, 26.8e9)
        assert torch.equal(res[b0:b0 + 3], bur.residual(u[b0:b0 + 3].clone(), boundary=True))
    assert torch.equal(bur.residual(u[-5:]), res[-5:, 1:-1, 1:-1])
    a = bur.residual(u, boundary=True, absolute=True)
    for b0 in range(0, B, 8192):
        assert torch.equal(a[b0:b0 + 8192], res[b0:b0 + 8192].abs())
    del u
    q, mod, sc = _check_joint(icp, pipeline, res.unsqueeze(1), (0, 1, 1), B, gpu)
    m = mod[0, 1:-1, 1:-1]
    for j in (0, 9):
        cov = icp.emp_cov_joint([-(q[j] * m), q[j] * m], res[:, 1:-1, 1:-1])
        assert cov >= (icp.kth_index(B, B, alphas[j]) + 1) / B - 2.0 / B
    del res
    qm = pipeline.marginal_qhat(a, alphas)                                  # per-cell q-hat over all 65 536 samples
    assert qm.shape == (len(alphas), T, X) and (qm[:-1] >= qm[1:]).all()
    for j in (0, 9):
        inside = torch.zeros(T, X, dtype=torch.int32, device=gpu)
        for b0 in range(0, B, 8192):
            inside += (a[b0:b0 + 8192] <= qm[j]).sum(0, dtype=torch.int32)
        assert int(inside.min()) >= icp.kth_index(B, B, alphas[j]) + 1
    for (t, x) in ((0, 0), (100, 257), (199, 511)):
        col = torch.sort(a[:, t, x].contiguous()).values
        assert torch.equal(qm[:, t, x], col[[icp.kth_index(B, B, al) for al in alphas]])

"""BASELINE.json configurations at their FULL per-GPU sizes, through the same classes bench.py drives: a few samples of
every residual against the CPU oracle (oracle/residuals.py = the reference's F.conv3d / F.conv2d arithmetic; whole tensors
are beyond what it can check in a test, two or three samples take seconds) + size-independent properties of the rest.

  C1 whole [256,100,200]                 advection additive kernel (Marginal/Advection_Residuals_CP.py:156-164,234-237): ALL of it vs the oracle
  C3 x-slab [4096,64,128+2,512] x3       THE BENCHMARKED JOB, call for call (bench.C3Stream): NS momentum, joint + marginal CP
  C3 slab  [4096,10,512,512] x3 fields   NS momentum residual (Marginal/NS_Residuals_CP.py:231-240), joint + marginal CP
  C4 shard [1024,64,256,256] x6 fields   each of the five MHD residuals (Marginal/MHD_Residuals_CP.py:225-278), joint CP;
                                         the reference script's own recipe - Nt-fastest views + marginal CP at n = 8192 -
                                         in test_full_size_c4_marginal_ntfast
  C5 shard [8192,200,512]                Burgers residual (Joint/Burgers_Residuals_CP.py:182-187), joint + marginal CP
  C5 whole [65536,200,512]               the same at its N = 1 size (27 GB in, 27 GB residual): the anchor of the 1 -> 8 curve

Properties: the last samples of the big batch equal the same samples evaluated on their own (64-bit indexing, grid
decomposition); |.| epilogue == abs of the signed result; the interior-plane fast paths equal the full path; q-hat is
non-increasing in alpha and an input value; the conformal guarantee holds on the calibration set (at least k+1 of the
n calibration scores lie within q-hat, k = the rank calibrate selects)."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

RES_TOL = 1e-5          # north_star: residuals within 1e-5 rel fp32 (tensor-scale) of the reference arithmetic


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    torch.cuda.empty_cache()                 # (what earlier tests of the session left in the caching allocator is not "in use")
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



def _oracle_err(got, ref):
    return rel_err(got.detach().cpu().numpy(), ref.detach().cpu().numpy())


def test_full_size_c1_vs_oracle(gpu):
    """BASELINE config 1 WHOLE ([256,100,200], the reference's own CPU-sized case) against the oracle end to end:
    Marginal/Advection_Residuals_CP.py:156-164 (additive kernel D_t + (v disc dt/dx) D_x), :234-235 (residual, cropped),
    :237 ff. (|res| -> per-cell q-hat at the 10 alpha levels over the 256 samples)."""
    import bench
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd.residuals import Advection
    from oracle import conformal as oc
    from oracle import residuals as orr
    B, T, X = 256, 100, 200
    u = bench.synth_(torch.empty(B, T, X, device=gpu), 1) + 1.0                 # bench.py's C1 input
    adv = Advection(1.0, 0.005, 0.01, disc=2)
    res = adv.residual(u)                                                       # the reference's crop: [256,98,198]
    ref = orr.advection_residual(u.cpu(), 1.0, 2, 0.005, 0.01)
    assert res.shape == ref.shape == (B, T - 2, X - 2)
    assert _oracle_err(res, ref) <= RES_TOL
    full = adv.residual(u, boundary=True, absolute=True)
    assert _oracle_err(full, orr.advection_residual(u.cpu(), 1.0, 2, 0.005, 0.01, boundary=True).abs()) <= RES_TOL
    scores = res.abs().contiguous()
    sc_host = scores.cpu().numpy()
    alphas = _alphas()
    q = icp.calibrate_multi(scores, B, alphas).cpu().numpy()                    # ONE select for the ten levels
    for j, a in enumerate(alphas):
        # the device select against the oracle's calibrate on the SAME scores: bit for bit
        assert np.array_equal(q[j], oc.calibrate(sc_host, B, a)), a
        assert np.array_equal(icp.calibrate(scores, B, a).cpu().numpy(), q[j])  # the reference's one-level call
    # and end to end (oracle residual -> oracle scores -> oracle q-hat): the residuals differ in their last bits, so do
    # the order statistics picked from them
    q_ref = np.stack([oc.calibrate(np.abs(ref.numpy()), B, a) for a in alphas])
    assert rel_err(q, q_ref) <= RES_TOL


@pytest.mark.timeout(900)
def test_full_size_c3_xslab(gpu):
    """THE JOB THE HEADLINE IS MEASURED ON, call for call: bench.C3Stream builds what `python bench.py` times - the
    resident synthetic x-slab [4096+3,3,64,128+2,512], the residual buffer in its joint and its row-padded marginal
    layout - and `eval_slab` is the timed launch: `residual_momentum(vars_[s:s+4096,:,:,1:sl+1], boundary=True, out=res,
    halo_x=True)`.  Reference job: Marginal/NS_Residuals_CP.py:231-240 (residual), :282-289 (|res| -> calibrate).
      (i)   samples {0, 2047, 4095} of the first and of the last slab position against the CPU oracle on the same rows
            with their halo rows (<= 1e-5);
      (ii)  the streaming joint calibration of the slab, branch-and-bound == full pass, and the whole four-slab stream;
      (iii) the marginal q-hat through the row-padded view on three cells against torch.sort, bit for bit, and the
            |.| epilogue against the oracle."""
    import bench
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd import pipeline
    from oracle import residuals as orr
    torch.cuda.empty_cache()
    B, T, X, Y = bench.CONFIGS["c3"]["shape"]
    alphas = _alphas()
    st = bench.C3Stream(B, T, X, Y, 128, "x", gpu)
    assert st.slabs == [128, 128, 127, 127] and sum(st.slabs) == X - 2 and st.crop == (1, 0, 1)
    assert tuple(st.vars_.shape) == (B + 3, 3, T, 130, Y)
    res_of = st.views("joint")
    for s in (0, 3):
        sl = st.slabs[s]
        res = st.eval_slab(s, sl, res_of[sl])
        assert res.data_ptr() == st.res_buf.data_ptr() and tuple(res.shape) == (B, T, sl, Y) and res.is_contiguous()
        assert bench.c3_parity(st, res, s, sl, (0, 2047, B - 1)) <= RES_TOL                     # (i)
    # (ii) the last slab's calibration: pruned == full pass, q-hat an input value, the conformal guarantee
    _check_joint(icp, pipeline, res, st.crop, B, gpu)
    # ... and the stream as bench.py's step runs it (adaptive route over the four slab positions) against the full passes
    qs = []
    for prune in (True, False):
        jc = pipeline.JointCalibration(B, gpu, prune=prune)
        for s, sl in enumerate(st.slabs):
            jc.add_slab(st.eval_slab(s, sl, res_of[sl]), crop=st.crop)
        qs.append((jc.finish(alphas), jc.all_scores))
    assert torch.allclose(qs[0][0], qs[1][0], rtol=1e-6, atol=0.0) and torch.allclose(qs[0][1], qs[1][1], rtol=1e-6, atol=0.0)
    assert torch.isfinite(qs[0][0]).all() and (qs[0][0][:-1] >= qs[0][0][1:]).all()
    # (iii) marginal CP on the last slab position: |.| epilogue into the row-padded view, per-cell q-hat
    del res, res_of
    s, sl = 3, st.slabs[3]
    a = st.eval_slab(s, sl, st.views("marginal")[sl], absolute=True)
    assert a.stride(0) == T * sl * Y + 64 and not a.is_contiguous()
    v = st.oracle_inputs(s, sl, B - 1)
    ref = st.oracle_crop(orr.ns_momentum(v, st.dt, st.dx, st.dy, nu=st.nu, boundary=True))[0].abs()
    assert _oracle_err(a[B - 1], ref) <= RES_TOL
    q = pipeline.marginal_qhat(a, alphas)
    assert q.shape == (len(alphas), T, sl, Y) and (q[:-1] >= q[1:]).all()
    ks = [icp.kth_index(B, B, al) for al in alphas]
    for (t, x, y) in ((0, 0, 0), (31, 64, 257), (T - 1, sl - 1, Y - 1)):
        col = torch.sort(a[:, t, x, y].contiguous()).values
        assert torch.equal(q[:, t, x, y], col[ks])
    st.free()


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
    # the middle and the tail of the batch (offsets beyond 2^31 elements: B*3*T*X*Y = 3.2e10) against the CPU oracle
    from oracle import residuals as orr
    for b0 in (2047, B - 1):
        assert _oracle_err(full[b0], orr.ns_momentum(v[b0:b0 + 1].cpu(), 1e-2, 1.0 / X, 1.0 / Y, nu=1e-3, boundary=True)[0]) <= RES_TOL
    tail = ns.residual_momentum(v[-3:].clone(), boundary=True)          # (and alone: grid decomposition, 64-bit indexing)
    assert torch.equal(full[-3:], tail)
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


@pytest.fixture(scope="module")
def c4_fields(gpu):
    """The C4 shard's six fields [1024,6,64,256,256] (103 GB), shared by the five equations' tests."""
    v = torch.empty(1024, 6, 64, 256, 256, device=gpu)
    v.uniform_(0.5, 1.5, generator=torch.Generator(device=gpu).manual_seed(12))
    yield v
    del v
    torch.cuda.empty_cache()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("eq", ["induction", "continuity", "momentum", "energy", "gauss"])
def test_full_size_c4_shard(gpu, c4_fields, eq):
    """Each of C4's five equations (Marginal/MHD_Residuals_CP.py:225-231 continuity, :234-243 momentum, :247-256 energy,
    :259-268 induction, :271-278 gauss) on the per-rank shard [1024,6,64,256,256]: samples 511 and 1023 against the CPU
    oracle, the |.| epilogue, the reference's crop, joint CP with its guarantees."""
    from cp_pre_amd import inductive_cp as icp
    from cp_pre_amd import pipeline
    from cp_pre_amd.residuals import MHD
    from oracle import residuals as orr
    B, T, X, Y = 1024, 64, 256, 256
    v = c4_fields
    mhd = MHD()
    fn, ofn = getattr(mhd, "residual_" + eq), getattr(orr, "mhd_" + eq)
    res = fn(v, boundary=True)
    assert res.shape == (B, T, X, Y) and torch.isfinite(res[::128]).all()
    for b0 in (511, B - 1):                                                 # two samples against the CPU oracle
        assert _oracle_err(res[b0], ofn(v[b0:b0 + 1].cpu(), boundary=True)[0]) <= RES_TOL
    assert torch.equal(res[-2:], fn(v[-2:].clone(), boundary=True))
    a = fn(v, boundary=True, absolute=True)
    assert torch.equal(a, res.abs())
    crop_view = fn(v[:4])                                                   # boundary=False: the reference's crop
    assert torch.equal(crop_view, res[:4, 1:-1, 1:-1, 1:-1])
    del a
    q, mod, sc = _check_joint(icp, pipeline, res, (1, 1, 1), B, gpu)
    # bounds +-q*mod cover at least the guaranteed share of the calibration samples jointly
    alphas = _alphas()
    inner = res[:, 1:-1, 1:-1, 1:-1]
    m = mod[1:-1, 1:-1, 1:-1]
    for j in (0, 9):
        cov = icp.emp_cov_joint([-(q[j] * m), q[j] * m], inner)
        assert cov >= (icp.kth_index(B, B, alphas[j]) + 1) / B - 2.0 / B    # knife-edge samples sit exactly on the bound


@pytest.mark.timeout(600)
@pytest.mark.parametrize("layout", ["nt", "ny"])
def test_full_size_c4_marginal_as_the_script_runs_it(gpu, layout):
    """BASELINE config 4 the way Marginal/MHD_Residuals_CP.py runs it: residual_induction(cal_pred.permute(0,1,4,2,3))
    (:323-350: Nt-fastest views; "ny": the synthetic benchmark's contiguous fields) -> ncf_scores = |res| -> per-cell
    calibrate over the n_cal = 8192 samples (:408-418), as the per-rank job of the 8-way sharded flow that bench.py times
    (`secondary.c4_marginal_rank8[_ntfast]`, same function): |residual| of the rank's [1024,6,64,256,256] written
    plane-major, two samples against the CPU oracle; the select at n = 1024 over all planes and at n = 8192 over the
    planes a rank owns (at the receive staging's shape and pitch), three cells each against torch.sort, bit for bit."""
    import bench
    out = bench.measure_c4_marginal(gpu, _alphas(), layout, steps=1, warmup=0)
    par = out["parity"]
    assert par["residual_rel_err"] <= RES_TOL and par["ok"], par
    assert par["qhat_cells_equal_sorted_columns"], par
    assert out["layout"] == layout and out["select_ms"] > 0 and out["kernel_ms"] > 0


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
    from oracle import residuals as orr
    for b0 in (0, B - 1):                                                   # two samples against the CPU oracle
        assert _oracle_err(res[b0], orr.burgers_residual(u[b0:b0 + 1].cpu(), 2.0 / X, 1.25 / T, 0.002, boundary=True)[0]) <= RES_TOL
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
    from oracle import residuals as orr
    for b0 in (32767, B - 1):                                               # (byte offsets 13.4e9, 26.8e9) vs the CPU oracle
        assert _oracle_err(res[b0], orr.burgers_residual(u[b0:b0 + 1].cpu(), 2.0 / X, 1.25 / T, 0.002, boundary=True)[0]) <= RES_TOL
    for b0 in (0, B - 3):
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

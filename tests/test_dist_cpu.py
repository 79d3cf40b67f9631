"""World-size-2 `gloo` tests (CPU) of the sharded calibration logic in cp_pre_amd.pipeline:
the all-reduce of per-cell moments, the all-gather of per-sample scores (joint CP) and the
all-to-all batch->cell re-sharding (marginal CP).  The compute back end injected here is a
numpy/torch-CPU one built on the oracle's definitions - it exists only in this test; the
product back end (pipeline.HipOps) is covered by the -m gpu tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cp_pre_amd import pipeline
from oracle import conformal as oc

ALPHAS = [0.1, 0.25, 0.5, 0.75, 0.9]     # alpha < 1/(n+1) makes the level exceed 1 (numpy raises)


class CpuOps:
    """Same interface as pipeline.HipOps, torch-CPU arithmetic (test double)."""

    @staticmethod
    def zeros_moments(M, device):
        return torch.zeros(2, M, dtype=torch.float64)

    @staticmethod
    def add_moments(res, mom):
        r = res.reshape(res.shape[0], -1).double()
        mom[0] += r.sum(0)
        mom[1] += (r * r).sum(0)

    @staticmethod
    def std_from_moments(mom, n_total, shape, eps, like=None):
        mean = mom[0] / n_total
        var = (mom[1] / n_total - mean * mean).clamp_min(0)
        return (var.sqrt().float() + eps).reshape(shape)

    @staticmethod
    def zeros_scores(n, device):
        return torch.zeros(n)

    @staticmethod
    def max_scores(res, mod, crop, scores):
        ct, cx, cy = crop
        sl = (slice(None), slice(ct, res.shape[1] - ct), slice(cx, res.shape[2] - cx), slice(cy, res.shape[3] - cy))
        e = (res.abs() / mod)[sl].reshape(res.shape[0], -1).amax(1)
        torch.maximum(scores, e, out=scores)

    @staticmethod
    def kth(scores, ks):
        return torch.sort(scores, dim=0).values[list(ks)]


class PlaneOps(CpuOps):
    """... with the one-launch multi-plane select of the device back end (``HipOps.kth_planes``): same addressing
    contract, so the strides ``pipeline`` hands to ``pre_kth_axis0_planes_f32`` are exercised on CPU."""
    calls = []

    @staticmethod
    def kth_planes(base, plane_stride, row_stride, planes, n, per, ks, out, out_rank_stride, out_plane_stride):
        PlaneOps.calls.append(planes)
        src = torch.as_strided(base, (planes, n, per), (plane_stride, row_stride, 1), base.storage_offset())
        dst = torch.as_strided(out, (planes, len(ks), per), (out_plane_stride, out_rank_stride, 1), out.storage_offset())
        dst.copy_(torch.sort(src, dim=1).values[:, list(ks)])


class PruneOps(CpuOps):
    """Test double of the device back end's ADAPTIVE joint route (``HipOps.prune_view`` / ``add_moments_segmax`` /
    ``max_scores_pruned`` / ``interior_planes``): the same layout rules (taken from ``HipOps`` itself - they are pure
    functions of shape, strides and crop) and the same interior-plane moments, with torch-CPU arithmetic.  "How much the
    bounds saved" is scripted: a sample with an entry above 1e3 counts as read whole, the others as 1 % read."""
    dense_order = staticmethod(pipeline.HipOps.dense_order)
    interior_planes = staticmethod(pipeline.HipOps.interior_planes)
    routes = []

    @staticmethod
    def prune_view(res, crop):
        order, dense = pipeline.HipOps.dense_order(res)
        if res.dim() != 4 or not dense:
            return None
        resc = res if order is None else res.permute(0, *order)
        cropc = tuple(crop) if order is None else tuple(crop[o - 1] for o in order)
        return (resc, cropc) if resc.shape[1] > 2 * cropc[0] else None

    @staticmethod
    def _canon(res):
        order, dense = pipeline.HipOps.dense_order(res)
        assert dense
        return (res if order is None else res.permute(0, *order)), order

    @staticmethod
    def add_moments(res, mom, skip_t=0):
        PruneOps.routes.append("plain")
        resc, _ = PruneOps._canon(res)
        a = resc[:, skip_t:resc.shape[1] - skip_t].reshape(resc.shape[0], -1).double()
        assert a.shape[1] == mom.shape[1], (a.shape, mom.shape)
        mom[0] += a.sum(0)
        mom[1] += (a * a).sum(0)

    @staticmethod
    def add_moments_segmax(resc, mom, cropc):
        PruneOps.routes.append("pruned")
        assert resc.is_contiguous()
        a = resc[:, cropc[0]:resc.shape[1] - cropc[0]].reshape(resc.shape[0], -1).double()
        assert a.shape[1] == mom.shape[1], (a.shape, mom.shape)
        mom[0] += a.sum(0)
        mom[1] += (a * a).sum(0)
        return resc.abs().reshape(resc.shape[0], -1).amax(1)

    @staticmethod
    def std_from_moments(mom, n_total, shape, eps, like=None, skip_t=0):
        order = PruneOps._canon(like)[1] if like is not None else None      # (like=None: a flat run of cells)
        mshape = tuple(shape) if order is None else tuple(shape[o - 1] for o in order)
        mean = mom[0] / n_total
        std = (mom[1] / n_total - mean * mean).clamp_min(0).sqrt().float() + eps
        mod = torch.full(mshape, float("nan"))
        mod[skip_t:mshape[0] - skip_t] = std.reshape((mshape[0] - 2 * skip_t,) + mshape[1:])
        return pipeline.icp.uncanon(mod, order, 0)

    @staticmethod
    def zeros_prune_stats(device):
        return torch.zeros(3, dtype=torch.int64)

    @staticmethod
    def max_scores_pruned(resc, modc, segmax, cropc, scores, stats=None):
        sl = (slice(None),) + tuple(slice(c, e - c) for c, e in zip(cropc, resc.shape[1:]))
        torch.maximum(scores, (resc.abs() / modc)[sl].reshape(resc.shape[0], -1).amax(1), out=scores)
        if stats is not None:
            wild = int((segmax > 1e3).sum())
            stats += torch.tensor([wild * 100 + (len(segmax) - wild), len(segmax) * 100, wild])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_local, shape, slabs, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.from_numpy(np.load(os.path.join(out_dir, "res.npy")))
        mine = full[rank * n_local:(rank + 1) * n_local]                  # batch sharding
        jc = pipeline.JointCalibration(n_local, "cpu", eps=0.0, group=dist.group.WORLD, ops=CpuOps)
        T = shape[0]
        step = (T - 2) // slabs
        for s in range(slabs):                                            # t-slabs with one halo plane each side
            jc.add_slab(mine[:, s * step:s * step + step + 2].contiguous(), crop=(1, 1, 1))
        q = jc.finish(ALPHAS)
        # the default moments exchange is reduce-scatter + all-gather of sigma-hat; the all-reduce form (rounds 1-5) must
        # give the same modulation, scores and q-hat bit for bit (each cell's sums are added in rank order either way
        # under gloo; on RCCL the ring order may differ in the last bit of an fp64 sum)
        assert jc.moments == "reduce_scatter"
        ar = pipeline.JointCalibration(n_local, "cpu", eps=0.0, group=dist.group.WORLD, ops=CpuOps, moments="all_reduce")
        for s in range(slabs):
            ar.add_slab(mine[:, s * step:s * step + step + 2].contiguous(), crop=(1, 1, 1))
        q_ar = ar.finish(ALPHAS)
        assert torch.equal(q, q_ar) and torch.equal(jc.all_scores, ar.all_scores)
        for m_rs, m_ar in zip(jc.modulation, ar.modulation):
            assert m_rs.shape == m_ar.shape and torch.equal(m_rs, m_ar)
        qm = pipeline.marginal_qhat(mine.abs().contiguous(), ALPHAS, group=dist.group.WORLD, ops=CpuOps)
        # bounded staging: 100 cells per rank per exchange -> 3 full runs and a ragged one (170 of 200 cells)
        qm_runs = pipeline.marginal_qhat(mine.abs().contiguous(), ALPHAS, group=dist.group.WORLD, ops=CpuOps,
                                         stage_bytes=4 * n_local * world * 100)
        assert torch.equal(qm_runs, qm)
        # the double-buffered asynchronous exchange against the blocking one (the default), 1 / 4 / 8 runs
        for cells_per_rank in (400, 100, 50):
            kw = dict(group=dist.group.WORLD, ops=CpuOps, stage_bytes=4 * n_local * world * cells_per_rank)
            a = pipeline.marginal_qhat(mine.abs().contiguous(), ALPHAS, overlap=True, **kw)
            b = pipeline.marginal_qhat(mine.abs().contiguous(), ALPHAS, overlap=False, **kw)
            assert torch.equal(a, b) and torch.equal(a, qm)
        # a TIME-MAJOR score buffer (what the t-slab driver lets the residual kernel write) is exchanged where it lies,
        # plane t to rank t % world, no pack: 10 planes over 2 / 3 ranks = 5 runs / 3 runs and a ragged one
        tm = pipeline.time_major(n_local, mine.shape[1:])
        tm.copy_(mine.abs())
        assert not tm.is_contiguous() and pipeline._is_time_major(tm)
        for ov in (False, True):
            q_tm = pipeline.marginal_qhat(tm, ALPHAS, group=dist.group.WORLD, ops=CpuOps, overlap=ov)
            assert q_tm.shape == qm.shape and torch.equal(q_tm, qm), ov
        tmp = pipeline.time_major(n_local, mine.shape[1:], pad=5)          # samples of a plane 5 floats further apart than
        tmp.copy_(mine.abs())                                              # a plane is long: the pad travels with the plane
        assert pipeline._is_time_major(tmp) and tmp.stride(0) == mine[0, 0].numel() + 5
        for ov in (False, True):
            assert torch.equal(pipeline.marginal_qhat(tmp, ALPHAS, group=dist.group.WORLD, ops=CpuOps, overlap=ov), qm), ov
        # the one-launch multi-plane select (HipOps.kth_planes' contract): runs of p planes per rank, p = 1, 2 and all
        plane_bytes = 4 * n_local * world * (mine[0, 0].numel() + 5)
        for planes_per_run, ov in ((1, False), (2, False), (2, True), (99, False)):
            PlaneOps.calls.clear()
            got = pipeline.marginal_qhat(tmp, ALPHAS, group=dist.group.WORLD, ops=PlaneOps, overlap=ov,
                                         stage_bytes=planes_per_run * plane_bytes)
            assert torch.equal(got, qm), (planes_per_run, ov)
            own = -(-mine.shape[1] // world)
            assert len(PlaneOps.calls) <= -(-own // min(planes_per_run, own)) and max(PlaneOps.calls) <= planes_per_run
        # ONE local sample per rank, padded planes: the sample stride of such a view says nothing, the planes do (round-3
        # advice: the pitch was taken as the unpadded plane size and the planes were read at the wrong offsets)
        m1 = mine[:1].abs().contiguous()
        q1 = pipeline.marginal_qhat(m1, [0.5], group=dist.group.WORLD, ops=CpuOps)
        for pad in (0, 5):
            t1 = pipeline.time_major(1, m1.shape[1:], pad=pad)
            t1.copy_(m1)
            assert pipeline._is_time_major(t1)
            for ops in (CpuOps, PlaneOps):
                assert torch.equal(pipeline.marginal_qhat(t1, [0.5], group=dist.group.WORLD, ops=ops), q1), (pad, ops)
        one = pipeline.time_major(n_local, (1,) + tuple(mine.shape[2:]))       # a single plane: rank 0 owns it, the others idle
        one.copy_(mine.abs()[:, 3:4])
        assert torch.equal(pipeline.marginal_qhat(one, ALPHAS, group=dist.group.WORLD, ops=CpuOps), qm[:, 3:4])
        with pytest.raises(ValueError):                       # a level above 1 is refused before any collective
            pipeline.marginal_qhat(mine.abs().contiguous(), [1e-6], group=dist.group.WORLD, ops=CpuOps)
        np.save(os.path.join(out_dir, f"q_{rank}.npy"), q.numpy())
        np.save(os.path.join(out_dir, f"scores_{rank}.npy"), jc.all_scores.numpy())
        np.save(os.path.join(out_dir, f"mod_{rank}.npy"), torch.cat([m[1:-1] for m in jc.modulation]).numpy())
        np.save(os.path.join(out_dir, f"qm_{rank}.npy"), qm.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_joint_and_marginal_world2(tmp_path, world):
    n_local, shape, slabs = 9, (10, 7, 11), 2
    rng = np.random.default_rng(0)
    res = (rng.standard_normal((world * n_local,) + shape) * (1 + rng.random(shape))).astype(np.float32)
    np.save(tmp_path / "res.npy", res)
    mp.spawn(_worker, args=(world, _free_port(), n_local, shape, slabs, str(tmp_path)), nprocs=world, join=True)

    inner = res[:, 1:-1, 1:-1, 1:-1]
    mod_ref = oc.modulation_func(inner.astype(np.float64), np.zeros_like(inner, dtype=np.float64))
    sc_ref = oc.ncf_metric_joint(inner, np.zeros_like(inner), mod_ref.astype(np.float32))
    n = world * n_local
    for r in range(world):
        mod = np.load(tmp_path / f"mod_{r}.npy")[:, 1:-1, 1:-1]
        assert np.max(np.abs(mod - mod_ref) / mod_ref) <= 1e-6
        sc = np.load(tmp_path / f"scores_{r}.npy")
        assert sc.shape == (n,) and np.max(np.abs(sc - sc_ref) / sc_ref) <= 1e-6        # rank order == batch order
        q = np.load(tmp_path / f"q_{r}.npy")
        for j, a in enumerate(ALPHAS):
            qr = oc.calibrate(sc_ref, n, a)
            assert abs(q[j] - qr) <= 1e-6 * abs(qr)
        qm = np.load(tmp_path / f"qm_{r}.npy")
        for j, a in enumerate(ALPHAS):
            assert np.array_equal(qm[j], oc.calibrate(np.abs(res), n, a))               # order statistic: exact
    for r in range(1, world):
        assert np.array_equal(np.load(tmp_path / "q_0.npy"), np.load(tmp_path / f"q_{r}.npy"))


def test_single_rank_pipeline_equals_whole_tensor_oracle():
    """group=None: the slab decomposition alone (no collectives) reproduces the whole-tensor recipe."""
    rng = np.random.default_rng(3)
    res = rng.standard_normal((12, 8, 6, 9)).astype(np.float32)
    t = torch.from_numpy(res)
    jc = pipeline.JointCalibration(12, "cpu", ops=CpuOps)
    for s in range(3):
        jc.add_slab(t[:, 2 * s:2 * s + 4].contiguous())
    q = jc.finish(ALPHAS).numpy()
    inner = res[:, 1:-1, 1:-1, 1:-1]
    mod = oc.modulation_func(inner.astype(np.float64), np.zeros_like(inner, dtype=np.float64)).astype(np.float32)
    sc = oc.ncf_metric_joint(inner, np.zeros_like(inner), mod)
    for j, a in enumerate(ALPHAS):
        assert abs(q[j] - oc.calibrate(sc, 12, a)) <= 1e-6 * abs(q[j])
    qm = pipeline.marginal_qhat(t.abs(), ALPHAS, ops=CpuOps).numpy()
    assert np.array_equal(qm[2], oc.calibrate(np.abs(res), 12, 0.5))
    tm = pipeline.time_major(12, t.shape[1:])                 # time-major buffer, no group: plane by plane, same result
    tm.copy_(t.abs())
    assert np.array_equal(pipeline.marginal_qhat(tm, ALPHAS, ops=CpuOps).numpy(), qm)
    tmp = pipeline.time_major(12, t.shape[1:], pad=7)
    tmp.copy_(t.abs())
    assert np.array_equal(pipeline.marginal_qhat(tmp, ALPHAS, ops=CpuOps).numpy(), qm)
    PlaneOps.calls.clear()                                    # ... and with the one-launch form of the device back end
    assert np.array_equal(pipeline.marginal_qhat(tmp, ALPHAS, ops=PlaneOps).numpy(), qm) and PlaneOps.calls == [8]
    assert np.array_equal(pipeline.marginal_qhat(tm, ALPHAS, ops=PlaneOps).numpy(), qm)


def _prune_worker(rank, world, port, n_local, shape, case, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.from_numpy(np.load(os.path.join(out_dir, "res.npy")))
        mine = full[rank * n_local:(rank + 1) * n_local]
        T = shape[0]
        step = (T - 2) // 3

        def stream(moments):
            jc = pipeline.JointCalibration(n_local, "cpu", group=dist.group.WORLD, ops=PruneOps, prune=case["prune"], moments=moments)
            for s in range(3):
                slab = mine[:, s * step:s * step + step + 2]
                if case["layout"] == "nt_fastest":            # the surrogate's [n, Nx, Ny, Nt] memory seen as [n, Nt, Nx, Ny]
                    slab = slab.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
                    assert not slab.is_contiguous() and PruneOps.interior_planes(slab, (1, 1, 1)) == 1
                else:
                    slab = slab.contiguous()
                jc.add_slab(slab, crop=(1, 1, 1))
            return jc, jc.finish(ALPHAS)
        # the all-reduce form of the moments exchange first (its routes are not the ones recorded), then the default
        # reduce-scatter + all-gather form: same routes, bit-identical modulation (NaN rim planes included), scores, q-hat
        ar, q_ar = stream("all_reduce")
        PruneOps.routes.clear()
        jc, q = stream("reduce_scatter")
        assert torch.equal(q, q_ar) and torch.equal(jc.all_scores, ar.all_scores)
        for m_rs, m_ar in zip(jc.modulation, ar.modulation):
            assert m_rs.shape == m_ar.shape and m_rs.stride() == m_ar.stride() and torch.equal(m_rs.nan_to_num(-1.0), m_ar.nan_to_num(-1.0))
        np.save(os.path.join(out_dir, f"q_{rank}.npy"), q.numpy())
        np.save(os.path.join(out_dir, f"scores_{rank}.npy"), jc.all_scores.numpy())
        with open(os.path.join(out_dir, f"routes_{rank}.txt"), "w") as f:
            f.write(",".join(PruneOps.routes))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("case", [
    dict(layout="nt_fastest", prune=True, wild=(1,), routes="pruned,plain,plain"),      # ranks' read fractions straddle 0.25:
    dict(layout="nt_fastest", prune=True, wild=(), routes="pruned,pruned,pruned"),      #   the GROUP decides, both leave
    dict(layout="contiguous", prune=True, wild=(0,), routes="pruned,plain,plain"),
    dict(layout="nt_fastest", prune="always", wild=(1,), routes="pruned,pruned,pruned"),
    dict(layout="nt_fastest", prune="never", wild=(1,), routes="plain,plain,plain"),
], ids=lambda c: f"{c['layout']}-{c['prune']}-wild{len(c['wild'])}")
def test_adaptive_prune_decision_is_collective(tmp_path, case):
    """JointCalibration's give-up decision (drop the branch-and-bound route after the first slab) is taken by the whole
    group from the summed counters, and the all-reduced moment vector has the same length on either route and any dense
    layout: rank 0 streams prunable data, rank 1 'wild' data (every sample read whole), in the surrogate's Nt-fastest
    layout - with rank-local decisions the second slab's all-reduce had mismatched sizes (round-3 advice).  Every rank
    must take the same route on every slab, finish, and agree with the whole-tensor oracle."""
    world, n_local, shape = 2, 8, (11, 6, 7)
    rng = np.random.default_rng(11)
    res = (rng.standard_normal((world * n_local,) + shape) * (1 + rng.random(shape))).astype(np.float32)
    for r in case["wild"]:
        res[r * n_local:(r + 1) * n_local, 4, 2, 3] = 5e3          # every sample of that rank: "read whole"
    np.save(tmp_path / "res.npy", res)
    mp.spawn(_prune_worker, args=(world, _free_port(), n_local, shape, case, str(tmp_path)), nprocs=world, join=True)
    inner = res[:, 1:-1, 1:-1, 1:-1]
    mod_ref = oc.modulation_func(inner.astype(np.float64), np.zeros_like(inner, dtype=np.float64))
    sc_ref = oc.ncf_metric_joint(inner, np.zeros_like(inner), mod_ref.astype(np.float32))
    n = world * n_local
    for r in range(world):
        assert open(tmp_path / f"routes_{r}.txt").read() == case["routes"], r
        sc = np.load(tmp_path / f"scores_{r}.npy")
        assert np.max(np.abs(sc - sc_ref) / sc_ref) <= 1e-6
        q = np.load(tmp_path / f"q_{r}.npy")
        for j, a in enumerate(ALPHAS):
            qr = oc.calibrate(sc_ref, n, a)
            assert abs(q[j] - qr) <= 1e-6 * abs(qr)


def test_prune_policy_argument():
    with pytest.raises(ValueError):
        pipeline.JointCalibration(4, "cpu", ops=CpuOps, prune="sometimes")
    with pytest.raises(ValueError):
        pipeline.JointCalibration(4, "cpu", ops=CpuOps, moments="gossip")
    assert pipeline.JointCalibration(4, "cpu", ops=CpuOps, prune="always").prune_checked
    assert not pipeline.JointCalibration(4, "cpu", ops=CpuOps, prune="never").prune


def test_moment_vector_length_depends_on_layout_alone():
    """``HipOps.interior_planes`` (pure layout arithmetic, no device): the planes the moments leave out are those of the
    slowest MEMORY axis inside its crop, whatever route scores the slab."""
    ip = pipeline.HipOps.interior_planes
    a = torch.empty(5, 6, 7, 8)
    assert ip(a, (1, 1, 1)) == 1 and ip(a, (0, 1, 1)) == 0 and ip(a, (3, 1, 1)) == 0 and ip(a, (2, 0, 0)) == 2
    nt = torch.empty(5, 7, 8, 6).permute(0, 3, 1, 2)               # logical [n,T=6,X=7,Y=8], memory [n,X,Y,T]
    assert ip(nt, (1, 2, 3)) == 2 and ip(nt, (1, 0, 3)) == 0       # the X crop: X is slowest in memory
    assert ip(a[:, :, :, ::2], (1, 1, 1)) == 0                     # not dense: whole planes
    assert pipeline.JointCalibration._plane_cells(nt) == 8 * 6 and pipeline.JointCalibration._plane_cells(a) == 56

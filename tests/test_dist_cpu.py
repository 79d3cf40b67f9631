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

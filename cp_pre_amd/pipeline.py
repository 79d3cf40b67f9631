"""Streaming calibration drivers: residual tensors larger than HBM are processed slab by
slab (or batch chunk by batch chunk) while only small calibration state stays resident.

Hot-path position: the step right after ``residual(...)`` in the reference scripts -
``Joint/Burgers_Residuals_CP.py:272-285`` (modulation -> per-sample score -> scalar q-hat)
and ``Marginal/Wave_Residuals_CP.py:280-290`` (per-cell q-hat) - restated for tensors that
the reference holds whole in host RAM (n_cal <= 1000 there) but BASELINE configs cannot
(C3: 275 GB per field).

``ops`` is the compute back end.  The product back end is :class:`HipOps` (C ABI calls on
device tensors).  Tests inject a numpy back end to exercise the sharding / collective
logic under ``gloo`` on CPU; nothing in this package constructs one.

Multi-GPU (one process per GPU, batch axis sharded, ``group`` = a ``torch.distributed``
process group over RCCL/xGMI):
  * joint: the per-cell moments are summed with ONE all-reduce per slab (fp64, 16 B per cell),
    the per-sample scores travel in ONE all-gather at the end (4 B per sample);
  * marginal: per-cell order statistics need every sample of a cell in one place: an
    all-to-all turns the batch sharding into a cell sharding, each rank selects its cells,
    and an all-gather returns the q-hat field.
"""
from __future__ import annotations

import torch

from . import _lib
from . import inductive_cp as icp


class HipOps:
    """Device back end: every method is one or two ``libcp_pre_hip.so`` launches."""

    interior_t = True      # can reduce only the interior t planes of an uncropped [n,T,X,Y] slab in place

    @staticmethod
    def zeros_moments(M, device):
        return torch.zeros(2, M, dtype=torch.float64, device=device)

    @staticmethod
    def add_moments(res, mom, skip_t=0):
        """``skip_t`` > 0 (contiguous [n,T,X,Y] only): reduce planes skip_t .. T-skip_t-1, which are one
        contiguous run of cells per sample (row stride T*X*Y); ``mom`` then has (T-2*skip_t)*X*Y cells."""
        res, _ = icp.canon(res)                 # cells in memory order; mom is indexed the same way
        n, M = res.shape[0], res.numel() // res.shape[0]
        a, cells = res, M
        if skip_t:
            plane = M // res.shape[1]
            a, cells = res[:, skip_t:], M - 2 * skip_t * plane
        with torch.cuda.device(res.device):
            _lib.check(_lib.load().pre_moments_axis0_f64(_lib.ptr(a), None, n, cells, M, _lib.ptr(mom[0]), _lib.ptr(mom[1]),
                                                         _lib.stream()), "pre_moments_axis0_f64")

    @staticmethod
    def std_from_moments(mom, n_total, shape, eps, like=None, skip_t=0):
        """``like``: the residual slab the moments came from; the modulation is returned as a
        logical [T,X,Y] view with that slab's memory order.  With ``skip_t`` the moments cover the
        interior planes only; the rim planes of the result are NaN (they are cropped anyway)."""
        M = mom.shape[1]
        order = icp.canon(like)[1] if like is not None else None
        mshape = tuple(shape) if order is None else tuple(shape[o - 1] for o in order)
        mod = torch.empty(mshape, dtype=torch.float32, device=mom.device)
        target = mod
        if skip_t:
            mod[:skip_t] = float("nan")
            mod[mshape[0] - skip_t:] = float("nan")
            target = mod[skip_t:mshape[0] - skip_t]
        with torch.cuda.device(mom.device):
            _lib.check(_lib.load().pre_std_from_moments_f32(_lib.ptr(mom[0]), _lib.ptr(mom[1]), n_total, M, float(eps),
                                                            _lib.ptr(target), _lib.stream()), "pre_std_from_moments_f32")
        return icp.uncanon(mod, order, 0)

    @staticmethod
    def zeros_scores(n, device):
        return torch.zeros(n, dtype=torch.float32, device=device)

    @staticmethod
    def max_scores(res, mod, crop, scores):
        """scores[i] = max(scores[i], max_cells |res[i]|/mod) over the cropped interior."""
        res, order = icp.canon(res)
        mod = icp.canon_cells(mod, order)
        n, (T, X, Y) = res.shape[0], res.shape[1:]
        ct, cx, cy = crop if order is None else tuple(crop[o - 1] for o in order)
        with torch.cuda.device(res.device):
            _lib.check(_lib.load().pre_joint_score_f32(_lib.ptr(res), None, _lib.ptr(mod), n, T, X, Y, ct, cx, cy,
                                                       _lib.ptr(scores), _lib.stream()), "pre_joint_score_f32")

    # measured (profiles/r02/prune_ab.txt): the pruned form pays from a few hundred samples and ~1e8 cells on; below,
    # the joint calibration is a few launch latencies either way and sigma-hat is too noisy for tight bounds
    PRUNE_MIN_CELLS = 1 << 27
    PRUNE_MIN_SAMPLES = 256
    PRUNE_MAX_SEGMENTS = (160 * 1024 - 256) // 4      # pre_joint_score_pruned_f32's work list: the LDS of a gfx950 workgroup

    @staticmethod
    def prune_view(res, crop):
        """(contiguous view of ``res`` with its cell axes in memory order, the crop in that order) when the
        branch-and-bound score applies, else None.  It takes a dense [n,T,X,Y] tensor in any axis order (the surrogate's
        Nt-fastest layout included: a segment is defined on the memory order) with at least one plane of the slowest
        cell axis inside its crop and at most ``PRUNE_MAX_SEGMENTS`` segments (64 consecutive cells of the two faster
        axes' flattened plane x 16 planes) per sample - the work list lives in LDS; it pays from a few hundred samples
        and ~1e8 cells on."""
        if res.dim() != 4 or res.numel() < HipOps.PRUNE_MIN_CELLS or res.shape[0] < HipOps.PRUNE_MIN_SAMPLES:
            return None
        resc, order = icp.canon(res)
        if order is None and not res.is_contiguous():
            return None                                            # not dense: canon copied it
        cropc = tuple(crop) if order is None else tuple(crop[o - 1] for o in order)
        A0, A1, A2 = resc.shape[1:]
        planes = A0 - 2 * cropc[0]
        if planes < 1 or ((planes + 15) // 16) * ((A1 * A2 + 63) // 64) > HipOps.PRUNE_MAX_SEGMENTS:
            return None
        return resc, cropc

    @staticmethod
    def can_prune(res, crop):
        return HipOps.prune_view(res, crop) is not None

    @staticmethod
    def add_moments_segmax(res, mom, crop):
        """``add_moments(res, mom, skip_t=crop[0])`` + from the same read the maxima of |res| per sample and segment
        (64 consecutive cells of the flattened (x, y) plane x 16 planes of those inside the t crop; cells within ``crop``
        of the x / y rim excluded): int32 bit patterns [n, ceil(planes/16), ceil(X*Y/64)] for ``max_scores_pruned``."""
        n, (T, X, Y) = res.shape[0], res.shape[1:]
        planes = T - 2 * crop[0]
        segmax = torch.empty(n, (planes + 15) // 16, (X * Y + 63) // 64, dtype=torch.int32, device=res.device)
        with torch.cuda.device(res.device):
            _lib.check(_lib.load().pre_moments_segmax_f64(_lib.ptr(res[:, crop[0]:]), res.stride(0), n, planes, X, Y, crop[1],
                                                          crop[2], _lib.ptr(mom[0]), _lib.ptr(mom[1]), _lib.ptr(segmax),
                                                          _lib.stream()), "pre_moments_segmax_f64")
        return segmax

    @staticmethod
    def zeros_prune_stats(device):
        """[segments read, segments, samples swept whole], accumulated on the device by ``max_scores_pruned``."""
        return torch.zeros(3, dtype=torch.int64, device=device)

    @staticmethod
    def max_scores_pruned(res, mod, segmax, crop, scores, stats=None):
        """Branch-and-bound form of ``max_scores`` for the tensor ``add_moments_segmax`` just read: the same scores for
        the same modulation, bit for bit, but only the segments whose bound (max |res| / min mod) exceeds a sample's
        best score so far are read; a sample whose bounds leave more than a quarter of its segments is flagged instead
        and takes the full pass (``pre_joint_score_flagged_f32``: the other samples' workgroups leave at once)."""
        n, (T, X, Y) = res.shape[0], res.shape[1:]
        planes, mod_in = T - 2 * crop[0], mod[crop[0]:]
        segmin = torch.empty(segmax.shape[1:], dtype=torch.float32, device=res.device)
        flags = torch.empty(n, dtype=torch.int32, device=res.device)
        lib = _lib.load()
        with torch.cuda.device(res.device):
            _lib.check(lib.pre_segmin_mod_f32(_lib.ptr(mod_in), planes, X, Y, crop[1], crop[2], _lib.ptr(segmin), _lib.stream()),
                       "pre_segmin_mod_f32")
            _lib.check(lib.pre_joint_score_pruned_f32(_lib.ptr(res[:, crop[0]:]), res.stride(0), _lib.ptr(mod_in), _lib.ptr(segmax),
                                                      _lib.ptr(segmin), n, planes, X, Y, crop[1], crop[2], _lib.ptr(scores),
                                                      _lib.ptr(flags), _lib.ptr(stats) if stats is not None else None,
                                                      _lib.stream()), "pre_joint_score_pruned_f32")
            _lib.check(lib.pre_joint_score_flagged_f32(_lib.ptr(res), None, _lib.ptr(mod), n, T, X, Y, crop[0], crop[1], crop[2],
                                                       _lib.ptr(flags), _lib.ptr(scores), _lib.stream()),
                       "pre_joint_score_flagged_f32")

    @staticmethod
    def kth(scores, ks):
        return icp.kth_axis0(scores, ks)


def _ranks(n_total, alphas):
    return [icp.kth_index(n_total, n_total, a) for a in alphas]


class JointCalibration:
    """Joint CP over a calibration set streamed as T-slabs of [n_local, T_slab, X, Y] residuals.

    Per slab (all local samples, a few time planes): ``add_slab(res, crop)`` accumulates the
    per-cell moments, all-reduces them across ranks, finishes the modulation for those planes
    and max-accumulates every local sample's score.  ``finish(alphas)`` all-gathers the scores
    and selects the q-hats.  A per-cell std needs all samples but only its own cell, and a
    per-sample max composes across slabs, so one sweep over the data suffices.
    """

    # the bounds are dropped for the rest of the stream when the first pruned slab had to read more than this share
    # of its segments (measured, profiles/r03/prune_ab.txt: beyond it the segment maxima cost more than they save)
    PRUNE_GIVE_UP = 0.25

    def __init__(self, n_local, device, eps=0.0, group=None, ops=None, prune=True):
        self.ops = ops or HipOps
        self.prune = prune
        self.prune_stats = None        # device [segments read, segments, samples swept whole] over the pruned slabs
        self.prune_checked = False
        self.group, self.eps, self.n_local, self.device = group, eps, n_local, device
        self.world = torch.distributed.get_world_size(group) if group is not None else 1
        self.n_total = n_local * self.world
        self.scores = self.ops.zeros_scores(n_local, device)
        self.modulation = []           # one [T_slab, X, Y] array per slab, in call order

    def add_slab(self, res, crop=(1, 1, 1)):
        """``res``: UNCROPPED residual slab [n_local, T_slab, X, Y]; ``crop`` cells per side are excluded
        from the score (the reference's ``[...,1:-1,1:-1,1:-1]``).  The t-rim planes may hold garbage
        (``PRE_FLAG_INTERIOR_T``): they are neither reduced nor scored.
        When the slab allows it (``HipOps.prune_view``: dense, large enough, ...) the moments pass also delivers
        per-segment maxima of |res| and the score pass reads only the segments that can still raise a sample's
        score - the same scores for the same modulation, bit for bit (``prune=False`` forces the full pass).  The
        route adapts to the data: a sample whose bounds prune little is flagged and takes the full pass, and a stream
        whose first pruned slab read more than ``PRUNE_GIVE_UP`` of its segments takes the plain passes from its
        second slab on."""
        ops = self.ops
        M = res[0].numel() if hasattr(res[0], "numel") else res[0].size
        if self.prune and self.prune_stats is not None and not self.prune_checked:
            # ONE host read per stream, when its SECOND slab arrives (a single-slab stream never synchronises for it):
            # were the bounds worth their segment maxima on the first?  (Every rank reads its own counters; the
            # decision only changes which LOCAL kernels run.)
            self.prune_checked = True
            read, total, _ = (int(v) for v in self.prune_stats.tolist())
            if total and read > self.PRUNE_GIVE_UP * total:
                self.prune = False
        view = ops.prune_view(res, crop) if self.prune and getattr(ops, "prune_view", None) else None
        if view is not None:
            resc, cropc = view                                           # cell axes in memory order
            skip = cropc[0]                                              # planes of the slowest axis inside its crop only
        else:
            skip = crop[0] if (getattr(ops, "interior_t", False) and crop[0] > 0 and res.is_contiguous()
                               and res.shape[1] > 2 * crop[0]) else 0
        kw = {"skip_t": skip} if skip else {}
        mom = ops.zeros_moments(M - 2 * skip * (M // (resc if view is not None else res).shape[1]), self.device)
        if view is not None:
            segmax = ops.add_moments_segmax(resc, mom, cropc)
        else:
            ops.add_moments(res, mom, **kw)
        if self.group is not None:
            torch.distributed.all_reduce(mom, group=self.group)          # RCCL: sum of (sum, sumsq) per cell
        mod = ops.std_from_moments(mom, self.n_total, tuple(res.shape[1:]), self.eps, like=res, **kw)
        if view is not None:
            if self.prune_stats is None and getattr(ops, "zeros_prune_stats", None):
                self.prune_stats = ops.zeros_prune_stats(self.device)
            kw2 = {"stats": self.prune_stats} if self.prune_stats is not None else {}
            ops.max_scores_pruned(resc, icp.canon_cells(mod, icp.canon(res)[1]), segmax, cropc, self.scores, **kw2)
        else:
            ops.max_scores(res, mod, crop, self.scores)
        self.modulation.append(mod)
        return mod

    def score_pass_read_frac(self):
        """Share of the residual's segments the pruned score passes of this stream read (None: no pruned slab)."""
        if self.prune_stats is None:
            return None
        read, total, _ = (int(v) for v in self.prune_stats.tolist())
        return read / total if total else None

    def finish(self, alphas):
        scores = self.scores
        if self.group is not None:
            gathered = [torch.empty_like(scores) for _ in range(self.world)]
            torch.distributed.all_gather(gathered, scores, group=self.group)   # RCCL: n_local floats per rank
            scores = torch.cat(gathered)
        self.all_scores = scores
        return self.ops.kth(scores, _ranks(self.n_total, alphas))


def time_major(n_local, cells, pad=0, dtype=torch.float32, device=None):
    """A score / residual buffer for ``marginal_qhat``'s zero-copy exchange: memory [T][n_local][*rest (+ pad)]
    (time-major), returned as its logical [n_local, T, *rest] view.  Plane t of every local sample is then one contiguous
    block - the send block of the all-to-all that gives plane t to rank t % world - so a residual kernel that writes
    through this view (``NavierStokes.residual_momentum(out=...)``: any batch / time strides over dense planes) has
    already packed the exchange.  ``pad``: floats between a sample's plane and the next sample's (64 keeps the rows of a
    power-of-two plane from sharing their low address bits, see :func:`row_padded`; the pad travels with the plane)."""
    T, rest = cells[0], tuple(cells[1:])
    per = 1
    for d in rest:
        per *= d
    pitch = per + pad
    buf = torch.empty(T * n_local * pitch, dtype=dtype, device=device)
    strides, acc = [], 1
    for d in reversed(rest):
        strides.append(acc)
        acc *= d
    return buf.as_strided((n_local, T) + rest, (pitch, n_local * pitch) + tuple(reversed(strides)))


def row_padded(n_local, cells, pad=64, dtype=torch.float32, device=None):
    """A score / residual buffer [n_local, *cells] whose rows are ``pad`` elements further apart than they are long.
    With M = 2^k cells per row every row of a cell's column shares its low address bits, which costs the per-cell select's
    column sweeps 4-11 % on the MI355X (``profiles/r03/row_pitch.txt``); ``marginal_qhat`` / ``kth_axis0`` select
    where the rows lie (``pre_kth_axis0_strided_f32``), the residual kernels write through any batch stride."""
    M = 1
    for d in cells:
        M *= d
    buf = torch.empty(n_local * (M + pad), dtype=dtype, device=device)
    strides, acc = [], 1
    for d in reversed(cells):
        strides.append(acc)
        acc *= d
    return buf.as_strided((n_local,) + tuple(cells), (M + pad,) + tuple(reversed(strides)))


def _is_time_major(scores):
    """[n, T, ...] whose memory is [T][n][... (+ pad)]: dense planes, the samples of a plane ``pitch`` >= plane size
    apart, the planes n * pitch apart (see :func:`time_major`)."""
    if scores.dim() < 3 or scores.shape[1] < 1:
        return False
    n, T = scores.shape[0], scores.shape[1]
    if not scores[0, 0].is_contiguous():
        return False
    per, pitch = scores[0, 0].numel(), scores.stride(0)
    return pitch >= per and (T == 1 or scores.stride(1) == n * pitch) and (n == 1 or T == 1 or pitch < scores.stride(1))


def _marginal_planes(scores, alphas, group, ops, overlap):
    """Sharded per-cell q-hat of a TIME-MAJOR score tensor [n_local, T, *rest]: no pack copy, one all-to-all per run of
    ``world`` planes (plane t goes to rank t % world, which selects over all n_local * world samples of it), one
    all-gather of the q-hat planes at the end.  Bytes on the wire per rank: (world-1)/world of its scores once
    (4 B x n_local x cells), + nk/n_local of that for the q-hats.  Staging: ONE received plane of all samples
    (4 B x n_local x world x cells per plane), twice with ``overlap``."""
    dist = torch.distributed
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_local, T = scores.shape[0], scores.shape[1]
    rest = tuple(scores.shape[2:])
    per = 1
    for d in rest:
        per *= d
    ks = _ranks(n_local * world, alphas)          # raises before any collective if a level exceeds 1
    nk = len(alphas)
    pitch = scores.stride(0) if n_local > 1 else per            # floats from one sample's plane to the next one's
    blk = n_local * pitch                                       # one plane of all local samples, as it lies
    tm = scores.as_strided((T, blk), (blk, 1))                  # the memory as it lies: [T][n_local * pitch], no copy
    runs = (T + world - 1) // world
    nbuf = 2 if (overlap and runs > 1) else 1
    recv = [tm.new_empty(world * blk) for _ in range(nbuf)]
    q_own = tm.new_zeros(runs, nk, per)           # the q-hat planes this rank owns (plane k * world + rank), run by run
    work = [None] * nbuf

    def exchange(k, b):
        t0 = k * world
        pr = min(world, T - t0)                   # planes in this run: ranks >= pr receive nothing
        send = tm[t0:t0 + pr].reshape(-1)         # contiguous: [pr][n_local * pitch]
        in_split = [blk if r < pr else 0 for r in range(world)]
        out_split = [blk if rank < pr else 0] * world
        out = recv[b] if rank < pr else recv[b][:0]
        return dist.all_to_all_single(out, send, out_split, in_split, group=group, async_op=nbuf > 1)

    def select(k):
        b = k % nbuf
        if work[b] is not None:
            work[b].wait()                        # the compute stream waits for run k's exchange
            work[b] = None
        if k * world + rank < T:
            rows = recv[b].as_strided((world * n_local, per), (pitch, 1))        # (rows with the senders' pitch)
            q_own[k] = ops.kth(rows, ks).reshape(nk, per)

    for k in range(runs):
        b = k % nbuf
        work[b] = exchange(k, b)
        if nbuf > 1 and k > 0:
            select(k - 1)
        elif nbuf == 1:
            select(k)
    if nbuf > 1:
        select(runs - 1)
    parts = [torch.empty_like(q_own) for _ in range(world)]
    dist.all_gather(parts, q_own, group=group)                   # nk * (planes owned) * per floats per rank
    # parts[r][k, j] is q-hat j of plane k * world + r
    q = torch.stack(parts, dim=1).reshape(runs * world, nk, per)[:T]          # [T, nk, per]
    return q.transpose(0, 1).reshape((nk, T) + rest)


def marginal_qhat(scores, alphas, group=None, ops=None, stage_bytes=4 << 30, overlap=False):
    """Per-cell q-hat [len(alphas), *cells] of |residual| scores [n_local, *cells].

    Single rank (no group, or a group of one: the exchange is the identity): one multi-rank select per tensor, or per
    plane of a time-major one.  Sharded: all-to-all (batch-sharded -> cell-sharded), local select over all
    ``n_local * world`` samples, ONE all-gather of the result at the end.

    A TIME-MAJOR tensor (:func:`time_major`: what a t-slab driver lets its residual kernel write) is exchanged where it
    lies, plane by plane (``_marginal_planes``): no pack copy, no send staging.  Any other layout takes the cell-run
    form below, which packs each run into a [world, n_local, cells-per-rank] send buffer first.

    The exchange runs over runs of cells sized so that one send and one receive staging buffer hold at most
    ``stage_bytes`` each (a C3 slab of scores is 56 GB; staging it whole next to the fields would not fit in HBM).
    ``overlap``: the staging buffers are double-buffered and the all-to-all of run k is issued asynchronously
    (RCCL runs it on its own stream) before the select of run k-1 is enqueued, so the xGMI transfer of one run hides
    behind the select of the previous one; nothing else is ordered differently, the result is identical.  Off by
    default: the stream ordering it relies on (RCCL's stream against the compute stream, reuse of the staging
    buffers) has only ever run under gloo, whose collectives block the host."""
    ops = ops or HipOps
    n_local, cells = scores.shape[0], tuple(scores.shape[1:])
    world = torch.distributed.get_world_size(group) if group is not None else 1
    tmajor = _is_time_major(scores) and not scores.is_contiguous() and cells[0] > 1
    if world == 1:
        ks = _ranks(n_local, alphas)
        if tmajor:                                # plane by plane: each [n_local, *rest] is contiguous
            return torch.stack([ops.kth(scores[:, t], ks) for t in range(cells[0])], dim=1)
        return ops.kth(scores, ks)
    if tmajor:
        return _marginal_planes(scores, alphas, group, ops, overlap)
    if n_local * world > 0x7fffffff:
        raise ValueError(f"{n_local} x {world} calibration samples exceed the select's 32-bit sample count")
    flat = scores.reshape(n_local, -1)
    M = flat.shape[1]
    ks = _ranks(n_local * world, alphas)          # raises before any collective if a level exceeds 1
    nk = len(alphas)
    per = max(1, min((M + world - 1) // world, int(stage_bytes) // (4 * n_local * world)))   # cells per rank per run
    run = per * world
    runs = (M + run - 1) // run
    nbuf = 2 if (overlap and runs > 1) else 1
    send = [flat.new_empty(world, n_local, per) for _ in range(nbuf)]
    recv = [torch.empty_like(send[0]) for _ in range(nbuf)]
    q_own = flat.new_empty(runs, nk, per)         # the q-hats of the cells this rank owns, run by run
    work = [None] * nbuf

    def pack(k, buf):
        c0 = k * run
        w = min(run, M - c0)
        if w == run:                                           # rank r will own cells [c0 + r*per, c0 + (r+1)*per)
            buf.copy_(flat[:, c0:c0 + run].reshape(n_local, world, per).permute(1, 0, 2))
        else:                                                  # ragged last run: pad with zeros, dropped at the end
            buf.zero_()
            for r in range(world):
                wr = max(0, min(per, w - r * per))
                buf[r, :, :wr] = flat[:, c0 + r * per:c0 + r * per + wr]

    def select(k):
        b = k % nbuf
        if work[b] is not None:
            work[b].wait()                                     # the compute stream waits for run k's exchange
            work[b] = None
        q_own[k] = ops.kth(recv[b].reshape(world * n_local, per), ks)        # every sample of my cells: [nk, per]

    for k in range(runs):
        b = k % nbuf
        pack(k, send[b])
        # RCCL: (world-1)/world of the run leaves over xGMI; issued after the pack (stream order), completes on
        # RCCL's stream while the select below runs
        work[b] = torch.distributed.all_to_all_single(recv[b], send[b], group=group, async_op=nbuf > 1)
        if nbuf > 1 and k > 0:
            select(k - 1)
        elif nbuf == 1:
            select(k)
    if nbuf > 1:
        select(runs - 1)
    parts = [torch.empty_like(q_own) for _ in range(world)]
    torch.distributed.all_gather(parts, q_own, group=group)                  # RCCL: nk * M / world floats per rank
    # parts[r][k, j, i] is the q-hat j of cell k*run + r*per + i
    q = torch.stack(parts, dim=2).permute(1, 0, 2, 3).reshape(nk, runs * run)[:, :M]
    return q.reshape((nk,) + cells)

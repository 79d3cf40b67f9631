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
    _max_segments = None

    @staticmethod
    def prune_max_segments():
        """Longest work list ``pre_joint_score_pruned_f32`` accepts on the current device: it lives in the LDS of one
        workgroup, and the library reports what this device gives one (gfx950: 160 KiB -> 40 896 segments)."""
        if HipOps._max_segments is None:
            HipOps._max_segments = int(_lib.load().pre_joint_score_pruned_max_segments())
        return HipOps._max_segments

    @staticmethod
    def dense_order(res):
        """(cell-axis order in memory, dense?) of [n, *cells] WITHOUT touching the data: order None = as written."""
        if res.is_contiguous():
            return None, True
        if res.dim() >= 3:
            order = sorted(range(1, res.dim()), key=lambda k: (-res.stride(k), k))
            if res.permute(0, *order).is_contiguous():
                return order, True
        return None, False

    @staticmethod
    def interior_planes(res, crop):
        """Planes per side of the slowest cell axis IN MEMORY that lie inside ``crop`` and are therefore left out of the
        moments (they may hold garbage, ``PRE_FLAG_INTERIOR_T``): a function of shape, layout and crop alone, the same
        on the pruned and on the plain route - it fixes the length of the moment vector the ranks all-reduce."""
        if res.dim() != 4:
            return 0
        order, dense = HipOps.dense_order(res)
        if not dense:
            return 0
        slow = 1 if order is None else order[0]
        c0 = int(crop[slow - 1])
        return c0 if (c0 > 0 and res.shape[slow] > 2 * c0) else 0

    @staticmethod
    def prune_view(res, crop):
        """(contiguous view of ``res`` with its cell axes in memory order, the crop in that order) when the
        branch-and-bound score applies, else None.  It takes a dense [n,T,X,Y] tensor in any axis order (the surrogate's
        Nt-fastest layout included: a segment is defined on the memory order) with at least one plane of the slowest
        cell axis inside its crop and at most ``prune_max_segments()`` segments (64 consecutive cells of the two faster
        axes' flattened plane x 16 planes) per sample - the work list lives in LDS; it pays from a few hundred samples
        and ~1e8 cells on."""
        if res.dim() != 4 or res.numel() < HipOps.PRUNE_MIN_CELLS or res.shape[0] < HipOps.PRUNE_MIN_SAMPLES:
            return None
        order, dense = HipOps.dense_order(res)
        if not dense:
            return None
        resc = res if order is None else res.permute(0, *order)
        cropc = tuple(crop) if order is None else tuple(crop[o - 1] for o in order)
        A0, A1, A2 = resc.shape[1:]
        planes = A0 - 2 * cropc[0]
        if planes < 1 or ((planes + 15) // 16) * ((A1 * A2 + 63) // 64) > HipOps.prune_max_segments():
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

    @staticmethod
    def kth_planes(base, plane_stride, row_stride, planes, n, per, ks, out, out_rank_stride, out_plane_stride):
        """Order statistics ``ks`` of ``planes`` score matrices [n, per] in ONE launch (``pre_kth_axis0_planes_f32``):
        plane p starts ``p * plane_stride`` floats behind ``base``'s first element, its rows ``row_stride`` apart; rank j
        of plane p goes to ``out`` at ``j * out_rank_stride + p * out_plane_stride`` floats."""
        with torch.cuda.device(base.device):
            _lib.check(_lib.load().pre_kth_axis0_planes_f32(_lib.ptr(base), plane_stride, row_stride, planes, n, per,
                                                            _lib.iarr32([int(k) for k in ks]), len(ks), _lib.ptr(out),
                                                            out_rank_stride, out_plane_stride, _lib.stream()),
                       "pre_kth_axis0_planes_f32")


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

    def __init__(self, n_local, device, eps=0.0, group=None, ops=None, prune=True, moments="reduce_scatter"):
        """``moments`` (only with a group): how the per-cell moments of a slab become the modulation every rank needs.
        "reduce_scatter" (default): every rank needs sigma-hat (fp32), not the sums (2 x fp64) - the two moment vectors
        are REDUCE-SCATTERED (rank r receives the group's sums for its 1/W of the cells), each rank finishes sigma-hat for
        its own cells, and ONE all-gather of the fp32 sigma-hat hands every rank the whole field: (W-1)/W x (16 + 4) B
        per cell on the wire per rank instead of the all-reduce's 2 x (W-1)/W x 16 B (C3 strong-scaled, W = 8: 284 MB
        instead of 453 MB per rank and step), and sigma-hat is bit-identical on every rank by construction (each cell's
        is computed once).  "all_reduce": the sums are all-reduced and every rank finishes every cell (rounds 1-5).
        Either way the collectives' sizes follow from the slab's shape, layout and crop alone (see ``add_slab``).
        ``prune``: True / "adaptive" (bounds on the first slab, kept or dropped for the rest of the stream by what they
        saved there: ONE host read of two device counters when the second slab arrives - and, in a group, one 16-byte
        all-reduce of them, so that every rank takes the same route), "always" (no host read, no extra collective:
        ``add_slab`` never synchronises and can be captured in a HIP graph), False / "never" (the full score pass)."""
        self.ops = ops or HipOps
        if prune not in (True, False, "adaptive", "always", "never"):
            raise ValueError(f"prune={prune!r}: True / 'adaptive', 'always', False / 'never'")
        if moments not in ("reduce_scatter", "all_reduce"):
            raise ValueError(f"moments={moments!r}: 'reduce_scatter' or 'all_reduce'")
        self.moments = moments
        self.prune = prune not in (False, "never")
        self.prune_checked = prune == "always"     # nothing left to decide
        self.prune_stats = None        # device [segments read, segments, samples swept whole] over the pruned slabs
        self.group, self.eps, self.n_local, self.device = group, eps, n_local, device
        self.world = torch.distributed.get_world_size(group) if group is not None else 1
        self.n_total = n_local * self.world
        self.scores = self.ops.zeros_scores(n_local, device)
        self.modulation = []           # one [T_slab, X, Y] array per slab, in call order

    def _keep_pruning(self):
        """Were the bounds worth their segment maxima on the first pruned slab?  Decided ONCE per stream, when its second
        slab arrives (a single-slab stream never synchronises for it), and by the WHOLE GROUP: the counters of all ranks
        are summed (RCCL all-reduce of two int64) before the threshold is applied, so no rank can leave the pruned route
        alone.  (The collectives that carry data - the moments all-reduce, the score all-gather - have sizes fixed by the
        slab's shape and layout either way, see ``add_slab``; this keeps the ranks' kernels, and so their pace, alike.)"""
        st = self.prune_stats[:2]
        if self.group is not None:
            st = st.clone()
            torch.distributed.all_reduce(st, group=self.group)
        read, total = (int(v) for v in st.tolist())                 # the stream's one host read
        return not (total and read > self.PRUNE_GIVE_UP * total)

    def add_slab(self, res, crop=(1, 1, 1)):
        """``res``: UNCROPPED residual slab [n_local, T_slab, X, Y]; ``crop`` cells per side are excluded
        from the score (the reference's ``[...,1:-1,1:-1,1:-1]``).  The t-rim planes may hold garbage
        (``PRE_FLAG_INTERIOR_T``): they are neither reduced nor scored.
        When the slab allows it (``HipOps.prune_view``: dense, large enough, ...) the moments pass also delivers
        per-segment maxima of |res| and the score pass reads only the segments that can still raise a sample's
        score - the same scores for the same modulation, bit for bit (``prune=False`` forces the full pass).  The
        route adapts to the data: a sample whose bounds prune little is flagged and takes the full pass, and a stream
        whose first pruned slab read more than ``PRUNE_GIVE_UP`` of its segments (summed over the group) takes the plain
        passes from its second slab on.

        Collectives issued per slab, in this order on every rank: [second slab of an adaptive stream only: all-reduce of
        2 int64], then the moments: two reduce-scatters of the fp64 sums / sums of squares [cells reduced, padded to a
        multiple of the group size] + one all-gather of the fp32 sigma-hat (``moments="all_reduce"``: one all-reduce of
        [2, cells reduced]).  ``cells reduced`` follows from the slab's shape,
        its memory layout and ``crop`` ALONE (the planes of the slowest memory axis inside its crop), never from the
        route or from anything a rank measured: ranks may only be grouped if they stream slabs of the same shape and
        layout, and then their collectives match whatever their data."""
        ops = self.ops
        M = res[0].numel() if hasattr(res[0], "numel") else res[0].size
        if self.prune and self.prune_stats is not None and not self.prune_checked:
            self.prune_checked = True
            self.prune = self._keep_pruning()
        view = ops.prune_view(res, crop) if self.prune and getattr(ops, "prune_view", None) else None
        # the planes the moments cover: layout alone decides (the same on the pruned and on the plain route)
        layout = getattr(ops, "interior_planes", None)
        skip = layout(res, crop) if layout else 0
        if view is not None:
            resc, cropc = view                                           # cell axes in memory order
            assert skip == cropc[0], (skip, cropc)                       # (prune_view only takes dense slabs)
        kw = {"skip_t": skip} if skip else {}
        Mm = M - (2 * skip * self._plane_cells(res) if skip else 0)     # cells reduced
        scatter = self.group is not None and self.moments == "reduce_scatter"
        chunk = -(-Mm // self.world)                                    # cells per rank of the scattered form
        momp = ops.zeros_moments(chunk * self.world if scatter else Mm, self.device)
        mom = momp[:, :Mm]                                              # (the pad stays zero: nobody's cells)
        if view is not None:
            segmax = ops.add_moments_segmax(resc, mom, cropc)
        else:
            ops.add_moments(res, mom, **kw)
        if scatter:
            mod = self._modulation_scattered(momp, Mm, chunk, res, skip)
        else:
            if self.group is not None:
                torch.distributed.all_reduce(mom, group=self.group)      # RCCL: sum of (sum, sumsq) per cell
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

    def _modulation_scattered(self, momp, Mm, chunk, res, skip):
        """sigma-hat of a slab from the ranks' local moments ``momp`` [2, chunk * world] (fp64, cells in memory order, zero
        beyond ``Mm``): reduce-scatter of the sums and of the sums of squares (rank r: cells [r * chunk, (r+1) * chunk)),
        the std of the rank's own cells, ONE all-gather of the fp32 result - straight into the modulation's own memory,
        ``skip`` planes of the slowest memory axis in (those planes, inside the crop, are NaN as on the all-reduce route)."""
        dist, g, ops = torch.distributed, self.group, self.ops
        own = momp.new_empty(2, chunk)
        dist.reduce_scatter_tensor(own[0], momp[0], group=g)           # RCCL: (W-1)/W x 8 B per cell out per rank, twice
        dist.reduce_scatter_tensor(own[1], momp[1], group=g)
        sig_own = ops.std_from_moments(own, self.n_total, (chunk,), self.eps)      # fp32 [chunk]: this rank's cells
        shape = tuple(res.shape[1:])
        order = HipOps.dense_order(res)[0]                              # (pure layout arithmetic, any back end)
        mshape = shape if order is None else tuple(shape[o - 1] for o in order)
        plane = 1
        for d in mshape[1:]:
            plane *= d
        head = skip * plane
        buf = sig_own.new_empty(2 * head + chunk * self.world)
        dist.all_gather_into_tensor(buf[head:head + chunk * self.world], sig_own, group=g)   # (W-1)/W x 4 B per cell
        mod = buf[:mshape[0] * plane].view(mshape)
        if skip:
            mod[:skip] = float("nan")
            mod[mshape[0] - skip:] = float("nan")
        return icp.uncanon(mod, order, 0)

    @staticmethod
    def _plane_cells(res):
        """Cells of one plane of the slowest cell axis IN MEMORY of a dense [n, T, X, Y] slab."""
        if res.dim() < 3:
            return 0
        slow = max(range(1, res.dim()), key=lambda k: (res.stride(k), -k))
        return res[0].numel() // res.shape[slow]

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


def _select_planes(ops, base, plane_stride, row_stride, planes, n, per, ks, out, out_rank_stride, out_plane_stride):
    """All ``planes`` score matrices in one launch when the back end can (``kth_planes``), else plane by plane."""
    if getattr(ops, "kth_planes", None):
        ops.kth_planes(base, plane_stride, row_stride, planes, n, per, ks, out, out_rank_stride, out_plane_stride)
        return
    flat, oflat = base.reshape(-1) if base.is_contiguous() else base, out.reshape(-1)
    for pl in range(planes):
        rows = flat.as_strided((n, per), (row_stride, 1), flat.storage_offset() + pl * plane_stride)
        q = ops.kth(rows, ks)                                                         # [nk, per]
        oflat.as_strided((len(ks), per), (out_rank_stride, 1), out.storage_offset() + pl * out_plane_stride).copy_(q)


def _marginal_planes(scores, alphas, group, ops, overlap, stage_bytes):
    """Sharded per-cell q-hat of a TIME-MAJOR score tensor [n_local, T, *rest]: no pack copy.  Plane t goes to rank
    t % world, which selects over all n_local * world samples of it; one all-gather of the q-hat planes at the end.
    A RUN is ``p * world`` planes: p all-to-alls (one per plane a rank receives; each moves one plane of every sender,
    4 B x n_local x cells, to every rank) and then ONE select launch over the p received planes - the tiles of all of
    them in one grid, so the ragged last round of workgroups is paid once per run, not once per plane.  p = as many
    planes as ``stage_bytes`` of receive staging hold (4 B x n_local x world x cells each; at least one; twice the
    staging with ``overlap``).  Bytes on the wire per rank: (world-1)/world of its scores once, + nk/n_local of that
    for the q-hats."""
    dist = torch.distributed
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_local, T = scores.shape[0], scores.shape[1]
    rest = tuple(scores.shape[2:])
    per = 1
    for d in rest:
        per *= d
    ks = _ranks(n_local * world, alphas)          # raises before any collective if a level exceeds 1
    nk = len(alphas)
    # floats from one sample's plane to the next one's.  (A single local sample has no "next": its stride is whatever
    # the view says; the planes then tell - they lie n_local * pitch apart.)
    pitch = scores.stride(0) if n_local > 1 else (scores.stride(1) if T > 1 else per)
    blk = n_local * pitch                                       # one plane of all local samples, as it lies
    tm = scores.as_strided((T, blk), (blk, 1))                  # the memory as it lies: [T][n_local * pitch], no copy
    own = (T + world - 1) // world                              # planes a rank owns at most (plane kk * world + rank)
    p = max(1, min(own, int(stage_bytes) // (4 * world * blk)))
    runs = (own + p - 1) // p
    nbuf = 2 if (overlap and runs > 1) else 1
    recv = [tm.new_empty(p * world * blk) for _ in range(nbuf)]   # [p][world][n_local * pitch]
    q_own = tm.new_zeros(own, nk, per)            # the q-hat planes this rank owns, in order
    work = [[] for _ in range(nbuf)]

    def exchange(k, b):
        for j in range(p):
            t0 = (k * p + j) * world
            pr = min(world, T - t0)               # planes in this exchange: ranks >= pr receive nothing
            if pr <= 0:
                break
            send = tm[t0:t0 + pr].reshape(-1)     # contiguous: [pr][n_local * pitch]
            in_split = [blk if r < pr else 0 for r in range(world)]
            out_split = [blk if rank < pr else 0] * world
            dst = recv[b][j * world * blk:(j + 1) * world * blk]
            out = dst if rank < pr else dst[:0]
            w = dist.all_to_all_single(out, send, out_split, in_split, group=group, async_op=nbuf > 1)
            if nbuf > 1:
                work[b].append(w)

    def select(k):
        b = k % nbuf
        for w in work[b]:
            w.wait()                              # the compute stream waits for run k's exchanges
        work[b] = []
        kk0 = k * p
        mine = sum(1 for j in range(p) if (kk0 + j) * world + rank < T)      # (the first `mine` of the run's p)
        if mine:
            _select_planes(ops, recv[b], world * blk, pitch, mine, world * n_local, per, ks,
                           q_own[kk0:kk0 + mine], per, nk * per)

    for k in range(runs):
        b = k % nbuf
        exchange(k, b)
        if nbuf > 1 and k > 0:
            select(k - 1)
        elif nbuf == 1:
            select(k)
    if nbuf > 1:
        select(runs - 1)
    parts = [torch.empty_like(q_own) for _ in range(world)]
    dist.all_gather(parts, q_own, group=group)                   # nk * (planes owned) * per floats per rank
    # parts[r][kk, j] is q-hat j of plane kk * world + r
    q = torch.stack(parts, dim=1).reshape(own * world, nk, per)[:T]           # [T, nk, per]
    return q.transpose(0, 1).reshape((nk, T) + rest)


def marginal_qhat(scores, alphas, group=None, ops=None, stage_bytes=4 << 30, overlap=False):
    """Per-cell q-hat [len(alphas), *cells] of |residual| scores [n_local, *cells].

    Single rank (no group, or a group of one: the exchange is the identity): one multi-rank select per tensor, or per
    plane of a time-major one.  Sharded: all-to-all (batch-sharded -> cell-sharded), local select over all
    ``n_local * world`` samples, ONE all-gather of the result at the end.

    A TIME-MAJOR tensor (:func:`time_major`: what a slab driver lets its residual kernel write) is exchanged where it
    lies, whole planes at a time (``_marginal_planes``): no pack copy, no send staging, and ONE select launch over all the
    planes a run delivers (single rank: over all planes of the tensor).  Any other layout takes the cell-run form
    below, which packs each run into a [world, n_local, cells-per-rank] send buffer first.

    The exchange runs over runs of cells sized so that one send and one receive staging buffer hold at most
    ``stage_bytes`` each (a C3 slab of scores is 56 GB; staging it whole next to the fields would not fit in HBM).
    ``overlap``: the staging buffers are double-buffered and the all-to-all of run k is issued asynchronously
    (RCCL runs it on its own stream) before the select of run k-1 is enqueued, so the xGMI transfer of one run hides
    behind the select of the previous one; nothing else is ordered differently, the result is identical.  Off by
    default: the stream ordering it relies on (RCCL's stream against the compute stream, reuse of the staging
    buffers) has run under gloo at 2-3 ranks (whose collectives block the host) and on real RCCL at world size ONE only
    (tests/test_gpu_parity.py::test_marginal_exchange_overlap_on_rccl_at_world_size_one) - never across GPUs."""
    ops = ops or HipOps
    n_local, cells = scores.shape[0], tuple(scores.shape[1:])
    world = torch.distributed.get_world_size(group) if group is not None else 1
    tmajor = _is_time_major(scores) and not scores.is_contiguous() and cells[0] > 1
    if world == 1:
        ks = _ranks(n_local, alphas)
        if tmajor:                                # every plane [n_local, *rest (+ pad)] where it lies, ONE launch
            T, per = cells[0], scores[0, 0].numel()
            out = scores.new_empty((len(ks),) + cells)
            pitch = scores.stride(0) if n_local > 1 else per
            _select_planes(ops, scores, scores.stride(1), pitch, T, n_local, per, ks, out, T * per, per)
            return out
        return ops.kth(scores, ks)
    if tmajor:
        return _marginal_planes(scores, alphas, group, ops, overlap, stage_bytes)
    return _marginal_cells(scores, alphas, group, ops, overlap, stage_bytes)


def _marginal_cells(scores, alphas, group, ops, overlap, stage_bytes):
    """The cell-run form of the sharded per-cell q-hat (any layout): runs of cells are packed into a [world, n_local,
    cells-per-rank] send buffer, exchanged with one all-to-all per run (batch-sharded -> cell-sharded) and selected over all
    ``n_local * world`` samples; ONE all-gather of the result at the end.  (A group of one rank exchanges with itself:
    ``marginal_qhat`` skips this function then; the tests call it directly to run the exchange on RCCL at world size 1.)"""
    n_local, cells = scores.shape[0], tuple(scores.shape[1:])
    world = torch.distributed.get_world_size(group)
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

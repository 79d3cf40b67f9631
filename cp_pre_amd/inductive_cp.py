"""Split-conformal calibration on the MI355X: the functions the reference imports from
``Neural_PDE.UQ.inductive_cp`` (``Marginal/NS_Residuals_CP.py:58``), same names and call
signatures: ``calibrate``, ``modulation_func``, ``ncf_metric_joint``, ``emp_cov``,
``emp_cov_joint`` (+ ``filter_sims_joint``, ``Joint/Burgers_Residuals_CP.py:298-300``).

That module is an un-vendored, un-pinned submodule absent from the reference snapshot.
``modulation_func``, ``ncf_metric_joint`` and joint coverage are pinned to vectors produced by
executing the reference's in-tree restatement (``Tests/test_advection_inv_sampling_marginal.py:428,
430-431,464-465`` -> ``tests/golden/conformal_ref.npz``); ``calibrate`` has no source or restatement
anywhere in the reference (PARITY UNPINNED, permanently): it is the standard split-CP quantile
inferred from the call sites, tested against numpy.

    calibrate(scores, n, alpha)        = np.quantile(scores, ceil((n+1)(1-alpha))/n, axis=0, method='higher')
    modulation_func(a, b)              = np.std(a - b, axis=0)
    ncf_metric_joint(a, b, modulation) = np.max(np.abs(a - b)/modulation, axis=(1..))
    emp_cov(sets, y)                   = ((y >= lo) & (y <= hi)).mean()
    emp_cov_joint(sets, y)             = ((y >= lo).all(1..) & (y <= hi).all(1..)).mean()

numpy arrays in -> numpy out (what the reference scripts pass); torch tensors in -> torch
out on the same device.  The arithmetic runs in ``libcp_pre_hip.so`` either way (radix
select, sequential-order std, max-reduce); q-hat is bit-for-bit one of the input scores.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _dispatch, _lib


# ------------------------------------------------------------------ marshalling
def canon(d):
    """(contiguous view [n, *cells in MEMORY order], order) of a dense tensor whose cell axes are
    permuted in memory (e.g. a residual computed in the surrogate's [BS,Nx,Ny,Nt] layout): the
    calibration kernels only need "n rows of M cells", so they run on the data where it lies.
    ``order`` is None for an already-contiguous tensor; non-dense views are copied."""
    if d.is_contiguous():
        return d, None
    if d.dim() >= 3:
        order = sorted(range(1, d.dim()), key=lambda k: (-d.stride(k), k))
        v = d.permute(0, *order)
        if v.is_contiguous():
            return v, order
    return d.contiguous(), None


def canon_cells(m, order):
    """A per-cell array (no batch axis) in the memory order ``order`` chosen by :func:`canon`."""
    return m.contiguous() if order is None else m.permute(*[o - 1 for o in order]).contiguous()


def uncanon(r, order, lead):
    """Inverse of the cell permutation on a result with ``lead`` leading non-cell axes."""
    if order is None:
        return r
    inv = [0] * len(order)
    for pos, o in enumerate(order):
        inv[o - 1] = pos
    return r.permute(*range(lead), *[lead + i for i in inv])


# (why and when score rows are padded: _lib.wants_row_pad)
wants_row_pad, PAD = _lib.wants_row_pad, _lib.PAD


def upload_rows(host, device=None):
    """Device copy [n, *cells] of a host fp32 tensor for a per-cell select: row-padded (``pipeline.row_padded``'s layout)
    when the row length asks for it - the H2D transfer happens anyway, into the strided view instead of a dense one."""
    n = host.shape[0] if host.dim() >= 2 else 0
    M = host[0].numel() if n else 0
    if not (host.dim() >= 2 and wants_row_pad(n, M)):
        return host.to(device or "cuda")
    buf = torch.empty(n * (M + PAD), dtype=torch.float32, device=device or "cuda")
    view = buf.as_strided((n, M), (M + PAD, 1))
    flat = host.reshape(n, M)
    # (torch stages a host -> strided-device copy through a dense device temporary of the copy's size: row blocks of at
    # most 1 GiB keep that temporary small whatever the matrix - a 100 GB score matrix must not need 200 GB for a moment)
    rows = max(1, (1 << 28) // M)
    for r0 in range(0, n, rows):
        view[r0:r0 + rows].copy_(flat[r0:r0 + rows])
    strides, acc = [], 1
    for d in reversed(host.shape[1:]):
        strides.append(acc)
        acc *= d
    return buf.as_strided(tuple(host.shape), (M + PAD,) + tuple(reversed(strides)))


def _dev(x, like=None, rows=False):
    """fp32 device tensor (layout untouched) + a function that converts results back.  ``rows``: the tensor is a score
    matrix about to be selected along axis 0 - a HOST input is uploaded row-padded when that pays (``upload_rows``); a
    device tensor is the caller's and is selected where it lies."""
    if isinstance(x, torch.Tensor):
        if x.dtype != torch.float32:
            raise RuntimeError("cp_pre_amd.inductive_cp works on float32 tensors")
        if rows and not x.is_cuda and x.dim() >= 2:
            _lib.require_gpu()
            origin = x.device
            return upload_rows(x.contiguous()), (lambda r: _dispatch.from_device(r, origin))
        d, origin = _dispatch.to_device(x)
        return d, (lambda r: _dispatch.from_device(r, origin))
    arr = np.asarray(x)
    if arr.dtype != np.float32:
        arr = arr.astype(np.float32)
    _lib.require_gpu()
    host = torch.from_numpy(np.ascontiguousarray(arr))
    return (upload_rows(host) if rows else host.cuda()), (lambda r: r.cpu().numpy())


def _is_zero_like(b):
    """``np.zeros(res.shape)`` is how the scripts say 'no second argument' (Joint/Burgers_..:274)."""
    if b is None:
        return True
    if isinstance(b, np.ndarray):
        return not b.any()
    return False


# ------------------------------------------------------------------ ranks
def quantile_level(n, alpha):
    return math.ceil((n + 1) * (1 - alpha)) / n


def kth_index(n_rows, n, alpha):
    """0-based sorted index numpy's ``method='higher'`` selects: ceil(q*(n_rows-1)), float64."""
    q = np.ceil((n + 1) * (1 - alpha)) / n
    if not (0.0 <= q <= 1.0):
        raise ValueError("Quantiles must be in the range [0, 1]")
    return int(np.ceil((n_rows - 1) * q))


def rows_where_they_lie(scores):
    """(view [n, *cells in MEMORY order], row pitch or None, cell order or None) of a score tensor whose rows can be
    selected in place: dense, or dense rows a fixed pitch >= their length apart (``pipeline.row_padded``; a driver pads
    its residual buffer so that the rows of a cell's column are not a power of two apart), with the cell axes in any
    order in memory (a residual computed in the surrogate's [BS,Nx,Ny,Nt] layout).  Anything else: None (copied)."""
    if scores.is_contiguous():
        return scores, None, None
    if scores.dim() < 2:
        return None
    order = sorted(range(1, scores.dim()), key=lambda k: (-scores.stride(k), k))
    v = scores.permute(0, *order)
    M = v[0].numel()
    if not v[0].is_contiguous() or not (v.shape[0] == 1 or v.stride(0) >= M):
        return None
    pitch = v.stride(0) if (v.shape[0] > 1 and v.stride(0) != M) else None
    return v, pitch, (None if order == list(range(1, scores.dim())) else order)


def kth_axis0(scores, ks):
    """Order statistics ``ks`` (0-based ranks, any order) along axis 0 of a device tensor
    [n, ...] -> [len(ks), ...].  1-D scores use the scalar radix select."""
    lib = _lib.load()
    lie = rows_where_they_lie(scores)
    if lie is not None:
        scores, pitch, cell_order = lie
    else:
        pitch = None
        scores, cell_order = canon(scores)
    n = scores.shape[0]
    with torch.cuda.device(scores.device):
        if scores.dim() == 1:
            order = sorted(range(len(ks)), key=lambda i: ks[i])     # the scalar select wants ascending ranks
            sk = [int(ks[i]) for i in order]
            out = torch.empty(len(sk), dtype=torch.float32, device=scores.device)
            _lib.check(lib.pre_kth_f32(_lib.ptr(scores), n, _lib.iarr64(sk), len(sk), _lib.ptr(out), _lib.stream()),
                       "pre_kth_f32")
            inv = [0] * len(ks)
            for pos, i in enumerate(order):
                inv[i] = pos
            if inv != list(range(len(ks))):      # device-only reorder (no index upload: stays HIP-graph capturable)
                out = out.flip(0) if inv == list(range(len(ks) - 1, -1, -1)) else torch.stack([out[i] for i in inv])
        else:
            M = scores[0].numel()
            out = torch.empty((len(ks),) + tuple(scores.shape[1:]), dtype=torch.float32, device=scores.device)
            _lib.check(lib.pre_kth_axis0_strided_f32(_lib.ptr(scores), pitch or M, n, M, _lib.iarr32([int(k) for k in ks]), len(ks),
                                                     _lib.ptr(out), _lib.stream()), "pre_kth_axis0_strided_f32")   # (out[j] <-> ks[j])
    return uncanon(out, cell_order, 1)


# ------------------------------------------------------------------ the five functions
def calibrate(scores, n, alpha):
    """q-hat over axis 0; array of shape ``scores.shape[1:]`` (scalar for 1-D scores)."""
    d, back = _dev(scores, rows=True)
    k = kth_index(d.shape[0], n, alpha)
    res = back(kth_axis0(d, [k])[0])
    if d.dim() == 1 and not isinstance(scores, torch.Tensor):
        return res[()]                                       # numpy scalar, like np.quantile
    return res


def calibrate_multi(scores, n, alphas):
    """All levels in ONE radix sweep (the reference loops ``calibrate`` over 10 alphas,
    Marginal/Wave_Residuals_CP.py:284-288).  Returns [len(alphas), ...]."""
    d, back = _dev(scores, rows=True)
    ks = [kth_index(d.shape[0], n, a) for a in alphas]
    return back(kth_axis0(d, ks))


def _same_layout(t, order):
    """``t`` ([n, *cells]) as a contiguous tensor in the memory order ``order`` of its partner."""
    return t.contiguous() if order is None else t.permute(0, *order).contiguous()


def _wide(*xs):
    """numpy promotes float32 - float64 to float64 (``modulation_func(res, np.zeros(res.shape))``,
    Joint/Burgers_Residuals_CP.py:274): such calls get the fp64-accumulated route and a float64 result."""
    return any(isinstance(x, np.ndarray) and x.dtype == np.float64 for x in xs)


def modulation_func(a, b=None, eps=0.0):
    """Per-cell population std of (a-b) over axis 0.

    float32 inputs: float32 sequential accumulation in numpy's own operation order
    (bit-identical to ``np.std(a-b, axis=0)`` on float32).  If an argument is a float64 numpy
    array (numpy would then compute in float64) the sums are accumulated in fp64 on the device."""
    da, back = _dev(a)
    da, order = canon(da)
    db = None if _is_zero_like(b) else _same_layout(_dev(b)[0], order)
    n, M = da.shape[0], da.numel() // da.shape[0]
    mod = torch.empty(da.shape[1:], dtype=torch.float32, device=da.device)
    lib = _lib.load()
    with torch.cuda.device(da.device):
        if _wide(a, b):
            mom = torch.zeros(2, M, dtype=torch.float64, device=da.device)
            _lib.check(lib.pre_moments_axis0_f64(_lib.ptr(da), _lib.ptr(db), n, M, M, _lib.ptr(mom[0]), _lib.ptr(mom[1]),
                                                 _lib.stream()), "pre_moments_axis0_f64")
            _lib.check(lib.pre_std_from_moments_f32(_lib.ptr(mom[0]), _lib.ptr(mom[1]), n, M, float(eps), _lib.ptr(mod),
                                                    _lib.stream()), "pre_std_from_moments_f32")
            return back(uncanon(mod, order, 0)).astype(np.float64)
        _lib.check(lib.pre_std_axis0_f32(_lib.ptr(da), _lib.ptr(db), n, M, float(eps), _lib.ptr(mod), _lib.stream()),
                   "pre_std_axis0_f32")
    return back(uncanon(mod, order, 0))


def ncf_metric_joint(a, b, modulation, crop=0):
    """Per-sample max over cells of |a-b|/modulation -> [n].  ``crop`` > 0 restricts the max to
    the interior of the last three axes of UNCROPPED [n,T,X,Y] inputs (fusing the reference's
    ``[...,1:-1,1:-1,1:-1]``)."""
    da, back = _dev(a)
    da, order = canon(da)
    db = None if _is_zero_like(b) else _same_layout(_dev(b)[0], order)
    dm = canon_cells(_dev(modulation)[0], order)
    n = da.shape[0]
    if crop:
        if da.dim() != 4:
            raise ValueError("crop needs [n,T,X,Y] inputs")
        T, X, Y = da.shape[1:]                      # extents in memory order; the crop is the same on each axis
        ct = cx = cy = int(crop)
    else:
        # no crop: any rows x columns factorisation of the cells will do; one row of M cells would leave a single
        # workgroup per sample (measured 0.16 TB/s at n = 100) and a short innermost axis (the surrogate's Nt) would
        # defeat the float4 path: rows of the largest power of two that divides M
        M = da.numel() // n
        # (not longer than 512 either: a block takes 32 rows, and n x M / (32 x 4096) blocks left a calibration set
        # of n = 100 with two workgroups per CU - 2.4 TB/s against 5 with 512-cell rows)
        Y = next((c for c in (512, 256, 128, 64, 32, 16, 8, 4) if M % c == 0), da.shape[-1])
        T, X, ct, cx, cy = 1, M // Y, 0, 0, 0
    scores = torch.zeros(n, dtype=torch.float32, device=da.device)
    with torch.cuda.device(da.device):
        _lib.check(_lib.load().pre_joint_score_f32(_lib.ptr(da), _lib.ptr(db), _lib.ptr(dm), n, T, X, Y, ct, cx, cy,
                                                   _lib.ptr(scores), _lib.stream()), "pre_joint_score_f32")
    res = back(scores)
    return res.astype(np.float64) if _wide(a, b, modulation) else res


def _directed_f32(a, up):
    """float64 bound -> float32 rounded towards +inf (``up``) or -inf, so that comparing a float32
    sample with the float32 bound decides exactly like numpy's float64 comparison
    (``y >= lo64``  <=>  ``y >= ceil32(lo64)``;  ``y <= hi64``  <=>  ``y <= floor32(hi64)``)."""
    if not (isinstance(a, np.ndarray) and a.dtype == np.float64) and not isinstance(a, float):
        return a
    a64 = np.asarray(a, np.float64)
    a32 = a64.astype(np.float32)
    wrong = (a32 < a64) if up else (a32 > a64)
    fixed = np.nextafter(a32, np.float32(np.inf if up else -np.inf), dtype=np.float32)
    return np.where(wrong, fixed, a32).astype(np.float32)


def _bounds(pred_sets, y, outside=False):
    """y as [n, M] rows in MEMORY order (no copy for dense permuted layouts, e.g. the surrogate's Nt-fastest
    one) and the two bounds laid out the same way, per cell ([M]) or per sample and cell ([n, M])."""
    ydev = _dev(y)[0]
    logical = tuple(ydev.shape)                             # [n, *cells] as the caller sees it
    dy, order = canon(ydev)
    n, M = dy.shape[0], dy.numel() // dy.shape[0]

    def lay(t):
        # inside test: y >= lo & y <= hi; outside test: y <= lo | y >= hi  (opposite rounding directions)
        if t.numel() == M and t.dim() >= 1:
            return canon_cells(t.reshape(logical[1:]), order), 0
        if t.numel() == n * M:
            return _same_layout(t.reshape(logical), order), 1
        return None, -1                                      # scalar / other broadcast: materialise below
    lo_t = _dev(_directed_f32(pred_sets[0], up=not outside))[0]
    hi_t = _dev(_directed_f32(pred_sets[1], up=outside))[0]
    (lo, pl), (hi, ph) = lay(lo_t), lay(hi_t)
    if pl != ph or pl < 0:
        ps = 1 if 1 in (pl, ph) else 0
        shape = logical if ps else logical[1:]
        lo = lay(lo_t.expand(shape).contiguous())[0]
        hi = lay(hi_t.expand(shape).contiguous())[0]
    else:
        ps = pl
    return dy, lo, hi, n, M, ps


def emp_cov(pred_sets, y):
    """Fraction of all cells of all samples inside [lo, hi]."""
    dy, lo, hi, n, M, ps = _bounds(pred_sets, y)
    count = torch.zeros(1, dtype=torch.int64, device=dy.device)
    with torch.cuda.device(dy.device):
        _lib.check(_lib.load().pre_cov_count_f32(_lib.ptr(dy), _lib.ptr(lo), _lib.ptr(hi), n, M, ps, _lib.ptr(count),
                                                 _lib.stream()), "pre_cov_count_f32")
    return float(count.item()) / float(n * M)


def filter_sims_joint(pred_sets, y):
    """Per-sample 'every cell inside the band' flags (bool [n])."""
    dy, lo, hi, n, M, ps = _bounds(pred_sets, y)
    inside = torch.ones(n, dtype=torch.uint8, device=dy.device)
    with torch.cuda.device(dy.device):
        _lib.check(_lib.load().pre_cov_joint_f32(_lib.ptr(dy), _lib.ptr(lo), _lib.ptr(hi), n, M, ps, _lib.ptr(inside),
                                                 _lib.stream()), "pre_cov_joint_f32")
    flags = inside.bool()
    return flags if isinstance(y, torch.Tensor) else flags.cpu().numpy()


def filter_sims_within_bounds(lower_bound, upper_bound, samples, threshold, within=False):
    """``Active_Learning/Advection_AL_Marginal.py:169-198``: bool [n], True where at least
    ``threshold`` of a sample's cells lie inside [lo,hi] (``within``) or on/outside the bounds."""
    dy, lo, hi, n, M, ps = _bounds([lower_bound, upper_bound], samples, outside=not within)
    counts = torch.zeros(n, dtype=torch.int32, device=dy.device)
    with torch.cuda.device(dy.device):
        _lib.check(_lib.load().pre_cov_rowcount_f32(_lib.ptr(dy), _lib.ptr(lo), _lib.ptr(hi), n, M, ps, 0 if within else 1,
                                                    _lib.ptr(counts), _lib.stream()), "pre_cov_rowcount_f32")
    flags = (counts.double() / float(M)) >= threshold          # numpy: mean of bools in float64
    return flags if isinstance(samples, torch.Tensor) else flags.cpu().numpy()


def emp_cov_joint(pred_sets, y):
    f = filter_sims_joint(pred_sets, y)
    return float(f.sum().item()) / f.numel() if isinstance(f, torch.Tensor) else float(f.mean())     # count / n in float64, as numpy


ALPHA_LEVELS = np.arange(0.05, 0.95 + 0.1, 0.1)     # Marginal/Wave_Residuals_CP.py:284

"""Spectral (FFT) family of ``ConvOperator`` - SURVEY.md 8(f) rank 4; not on the measured path.

``conv='spectral'``, ``differentiate`` and ``integrate`` of ``Utils/ConvOps_2d.py:153-284`` /
``Utils/ConvOps_1d.py:153-283`` / ``Utils/ConvOps_Spatial.py:139-260`` (and the ``fft_conv`` they call,
``Utils/fft_conv_pytorch/fft_conv.py:35-131``) are all "zero-pad, rfftn, multiply by a function of the
kernel spectrum, irfftn, crop".  They run in ``libcp_pre_fft.so`` (``include/cp_pre_fft.h``,
``csrc/spectral.hip``): one fused embed pass, hipFFT R2C, one multiply pass that evaluates the kernel
spectrum from the taps, hipFFT C2R, one crop/normalise pass - for the multiplicative modes
(``spectral_convolution``, ``differentiate``); the singular 1/(K^+eps) modes keep the reference's
transform structure through ``torch.fft`` (see ``integrate``).  Like every other compute entry of the
package they run on the MI355X only - CPU tensors are staged through the GPU and come back on the CPU,
and without a HIP device the call raises (no CPU fallback).

When a gradient is required (field or kernel ``requires_grad`` with grad mode on) the same recipe is
composed from ``torch.fft`` device ops instead (hipFFT underneath), which autograd can differentiate.
"""
from __future__ import annotations

import ctypes
import functools
import weakref

import numpy as np
import torch
import torch.nn.functional as F
from torch.fft import irfftn, rfftn

from . import _dispatch, _lib

_STAGE_BYTES = 4 << 30          # padded array + half-spectrum staged per hipFFT batch


def _on_gpu(fn):
    @functools.wraps(fn)
    def wrapped(field, kernel, *args, **kwargs):
        _dispatch._check_field(field)
        dev, origin = _dispatch.to_device(field)
        return _dispatch.from_device(fn(dev, kernel.detach().to(dev.device) if not kernel.requires_grad else kernel.to(dev.device),
                                        *args, **kwargs), origin)
    return wrapped


def _with_channel(field, nd):
    return field.unsqueeze(1) if field.dim() == nd + 1 else field


def _kernel_spectrum(kernel, like, dims):
    k = kernel.to(like.device)[None, None]
    grow = [v for i in reversed(range(2, like.ndim)) for v in (0, like.size(i) - k.size(i))]
    return k, rfftn(F.pad(k, grow).float(), dim=dims)


@_on_gpu
def _torch_fft_xcorr(field, kernel, inverse=False, keep_channel=False):
    """``fft_conv(field, K, padding=k//2[, inverse])``: zero-pad by k//2, make the last axis
    even, multiply by the CONJUGATE kernel spectrum (cross-correlation) - or by
    1/(conj(K)+1e-6) when ``inverse`` - and keep the leading ``n - k + 1`` samples."""
    nd = kernel.dim()
    x = _with_channel(field, nd)
    pads = [kernel.shape[d] // 2 for d in range(nd)]
    x = F.pad(x, [p for d in reversed(range(nd)) for p in (pads[d], pads[d])])
    size = x.size()
    if x.size(-1) % 2:
        x = F.pad(x, [0, 1])
    dims = tuple(range(2, x.ndim))
    k, kf = _kernel_spectrum(kernel, x, dims)
    kf = torch.conj(kf)
    if inverse:
        kf = 1 / (kf + 1e-6)
    out = irfftn(rfftn(x.float(), dim=dims) * kf, dim=dims)
    keep = (slice(None), slice(None)) + tuple(slice(0, size[i] - k.size(i) + 1) for i in range(2, x.ndim))
    out = out[keep].contiguous()
    return out if keep_channel else out.squeeze(1)


def _pad_all(field, kernel):
    nd = kernel.dim()
    x = _with_channel(field, nd)
    p = kernel.size(-1) // 2
    return F.pad(x, (p, p) * nd, mode='constant'), tuple(range(2, x.ndim))


def _crop(out, xp, k, slice_pad, keep_channel=False):
    if slice_pad:
        keep = (slice(None), slice(None)) + tuple(slice(0, xp.size(i) - k.size(i) + 1) for i in range(2, xp.ndim))
        out = out[keep].contiguous()
    return out if keep_channel else out.squeeze(1)


@_on_gpu
def _torch_differentiate(field, kernel, correlation=False, slice_pad=True, keep_channel=False):
    xp, dims = _pad_all(field, kernel)
    k, kf = _kernel_spectrum(kernel, xp, dims)
    if correlation:
        kf = torch.conj(kf)
    out = irfftn(rfftn(xp.float(), dim=dims) * kf, dim=dims)
    return _crop(out, xp, k, slice_pad, keep_channel)


@_on_gpu
def _torch_integrate(field, kernel, correlation=False, slice_pad=False, eps=1e-6, keep_channel=False):
    xp, dims = _pad_all(field, kernel)
    k, kf = _kernel_spectrum(kernel, xp, dims)
    inv = 1 / (kf + eps)
    if correlation:
        inv = torch.conj(inv)
    out = irfftn(rfftn(xp, dim=dims) * inv, dim=dims)
    return _crop(out, xp, k, slice_pad, keep_channel)


# ---------------------------------------------------------------- native route (libcp_pre_fft.so)
class _Plan:
    """Owner of one ``pre_fft_t`` (a hipFFT R2C/C2R plan pair for one padded size and batch)."""

    def __init__(self, nd, n, inv_last, batch):
        self.handle = ctypes.c_void_p()
        _lib.check(_lib.load_fft().pre_fft_create(ctypes.byref(self.handle), nd, (ctypes.c_int64 * 3)(*n), inv_last, batch),
                   "pre_fft_create")
        nbytes = ctypes.c_size_t()
        _lib.check(_lib.load_fft().pre_fft_work_bytes(self.handle, ctypes.byref(nbytes)), "pre_fft_work_bytes")
        self.work_bytes = nbytes.value
        weakref.finalize(self, _lib.load_fft().pre_fft_destroy, self.handle)


_plans = {}


def _plan(dev, nd, n, inv_last, batch):
    key = (dev.index, nd, tuple(n), inv_last, batch)
    p = _plans.get(key)
    if p is None:
        if len(_plans) >= 16:
            _plans.pop(next(iter(_plans)))
        with torch.cuda.device(dev):
            p = _plans[key] = _Plan(nd, n, inv_last, batch)
    return p


def _native(field, kernel, pads, even_last, mode, eps, crop, keep_channel):
    """``pads``: zeros added on each side of every transformed axis; ``even_last``: one more zero at the end of the
    last axis when its padded length is odd (fft_conv.py:98-101) - otherwise the inverse transform drops to the
    even length torch.fft.irfftn assumes; ``crop``: keep ``padded - k + 1`` leading samples per axis, or all."""
    _dispatch._check_field(field)
    nd = kernel.dim()
    karr = _dispatch.host_kernel(kernel)
    dev, origin = _dispatch.to_device(field)
    x = _with_channel(dev, nd)                                  # [B, C, *spatial]
    lead, spatial = tuple(x.shape[:2]), tuple(x.shape[2:])
    xb = x.reshape((lead[0] * lead[1],) + spatial)              # a view whenever (B, C) collapse
    batch = xb.shape[0]
    three = lambda v, fill: (fill,) * (3 - nd) + tuple(v)       # noqa: E731
    dims, pad_lo, kd = three(spatial, 1), three(pads, 0), three(karr.shape, 1)
    padded = [d + 2 * p for d, p in zip(dims, pad_lo)]
    n = list(padded)
    if even_last and n[2] % 2:
        n[2] += 1
    inv_last = n[2] - (n[2] % 2)
    avail = (n[0], n[1], inv_last)
    od = tuple(min(a, s - k + 1) for a, s, k in zip(avail, padded, kd)) if crop else avail
    out = torch.empty((batch,) + od, dtype=torch.float32, device=xb.device)
    per_sample = 4 * n[0] * n[1] * n[2] + 8 * n[0] * n[1] * (n[2] // 2 + 1) + 256
    chunk = max(1, min(batch, _STAGE_BYTES // per_sample))
    strides = lambda t: (ctypes.c_int64 * 4)(t.stride(0), *three(t.stride()[1:], 0))     # noqa: E731
    lib, kptr = _lib.load_fft(), karr.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    i64x3 = lambda v: (ctypes.c_int64 * 3)(*v)                                               # noqa: E731
    with torch.cuda.device(xb.device):
        work = None
        for b0 in range(0, batch, chunk):
            nb = min(chunk, batch - b0)
            plan = _plan(xb.device, nd, n, inv_last, nb)
            if work is None or work.numel() < plan.work_bytes:
                work = torch.empty(plan.work_bytes, dtype=torch.uint8, device=xb.device)
            xi, oi = xb[b0:b0 + nb], out[b0:b0 + nb]
            oi4 = oi.reshape((nb,) + od)
            _lib.check(lib.pre_spectral_apply_f32(plan.handle, _lib.ptr(xi), strides(xi), i64x3(dims), i64x3(pad_lo), kptr,
                                                  i64x3(kd), mode, float(eps), _lib.ptr(oi4),
                                                  (ctypes.c_int64 * 4)(*oi4.stride()), i64x3(od), _lib.ptr(work), _lib.stream()),
                       "pre_spectral_apply_f32")
    out = out.reshape(lead + od[3 - nd:])
    if not keep_channel:
        out = out.squeeze(1)
    return _dispatch.from_device(out, origin)


def _wants_grad(field, kernel):
    return torch.is_grad_enabled() and (field.requires_grad or kernel.requires_grad)


def fft_xcorr(field, kernel, inverse=False, keep_channel=False):
    """``fft_conv(field, K, padding=k//2[, inverse])`` (fft_conv.py:35-131): zero-pad by k//2, make the last
    axis even, multiply by the CONJUGATE kernel spectrum (cross-correlation) - or by 1/(conj(K)+1e-6) when
    ``inverse`` - and keep the leading ``n - k + 1`` samples."""
    if inverse or _wants_grad(field, kernel):
        return _torch_fft_xcorr(field, kernel, inverse, keep_channel)          # inverse: see integrate()
    return _native(field, kernel, [s // 2 for s in kernel.shape], True, _lib.PRE_FFT_CONJ, 0.0, True, keep_channel)


def differentiate(field, kernel, correlation=False, slice_pad=True, keep_channel=False):
    """Utils/ConvOps_2d.py:179-228: pad every axis by k_last//2, multiply by K^ (conj for correlation)."""
    if _wants_grad(field, kernel):
        return _torch_differentiate(field, kernel, correlation, slice_pad, keep_channel)
    p = kernel.size(-1) // 2
    return _native(field, kernel, [p] * kernel.dim(), False, _lib.PRE_FFT_CONJ if correlation else 0, 0.0, slice_pad, keep_channel)


def integrate(field, kernel, correlation=False, slice_pad=False, eps=1e-6, keep_channel=False):
    """Utils/ConvOps_2d.py:231-284: as differentiate with 1/(K^ + eps).

    Deconvolution by a difference stencil is singular (K^ = 0 on whole frequency planes), so with the
    reference's eps = 1e-6 the result is round-off amplified 10^6 times, and WHICH round-off depends on
    the structure of the inverse transform: torch.fft.irfftn runs complex transforms over the leading
    axes and a 1-D C2R over the last one, reading every stored bin, whereas a multi-dimensional hipFFT
    C2R plan uses the Hermitian redundancy of the half-spectrum.  On exactly Hermitian input the two
    agree; on the amplified, slightly asymmetric one they differ at the percent level.  To stay within
    1e-4 of the reference the 1/(K^+eps) modes are therefore composed from torch.fft ops (same hipFFT
    kernels, the reference's transform structure); ``libcp_pre_fft.so`` implements PRE_FFT_INVERT too
    (tests compare it on a well-conditioned eps) and ``_native_integrate`` exposes it."""
    return _torch_integrate(field, kernel, correlation, slice_pad, eps, keep_channel)


def _native_integrate(field, kernel, correlation=False, slice_pad=False, eps=1e-6, keep_channel=False):
    p = kernel.size(-1) // 2
    mode = _lib.PRE_FFT_INVERT | (_lib.PRE_FFT_CONJ if correlation else 0)      # conj(1/(K^+eps)) == 1/(conj(K^)+eps)
    return _native(field, kernel, [p] * kernel.dim(), False, mode, eps, slice_pad, keep_channel)


def _native_fft_xcorr_inverse(field, kernel, eps=1e-6, keep_channel=False):
    return _native(field, kernel, [s // 2 for s in kernel.shape], True, _lib.PRE_FFT_CONJ | _lib.PRE_FFT_INVERT, eps, True,
                   keep_channel)

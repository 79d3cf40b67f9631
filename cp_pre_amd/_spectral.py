"""Spectral (FFT) family of ``ConvOperator`` - SURVEY.md 8(f) rank 4, NOT the HIP hot path.

``conv='spectral'``, ``differentiate`` and ``integrate`` of ``Utils/ConvOps_2d.py:153-284`` /
``Utils/ConvOps_1d.py:153-283`` (and the ``fft_conv`` they call,
``Utils/fft_conv_pytorch/fft_conv.py:35-131``) are kept on the class surface as thin
``torch.fft`` compositions executed by hipFFT: like every other compute entry of the package they
run on the MI355X only - CPU tensors are staged through the GPU and come back on the CPU, and without
a HIP device the call raises (no CPU fallback).  No hand-written kernel here; out of the measured path.
"""
from __future__ import annotations

import functools

import torch
import torch.nn.functional as F
from torch.fft import irfftn, rfftn

from . import _dispatch


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
def fft_xcorr(field, kernel, inverse=False, keep_channel=False):
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
def differentiate(field, kernel, correlation=False, slice_pad=True, keep_channel=False):
    xp, dims = _pad_all(field, kernel)
    k, kf = _kernel_spectrum(kernel, xp, dims)
    if correlation:
        kf = torch.conj(kf)
    out = irfftn(rfftn(xp.float(), dim=dims) * kf, dim=dims)
    return _crop(out, xp, k, slice_pad, keep_channel)


@_on_gpu
def integrate(field, kernel, correlation=False, slice_pad=False, eps=1e-6, keep_channel=False):
    xp, dims = _pad_all(field, kernel)
    k, kf = _kernel_spectrum(kernel, xp, dims)
    inv = 1 / (kf + eps)
    if correlation:
        inv = torch.conj(inv)
    out = irfftn(rfftn(xp, dim=dims) * inv, dim=dims)
    return _crop(out, xp, k, slice_pad, keep_channel)

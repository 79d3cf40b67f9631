"""Vector calculus on 2D+time fields: drop-in for ``Utils/VectorConvOps.py:17-81``.

The reference classes cannot be constructed as shipped (their ``requires_grad`` argument
lands in ``ConvOperator``'s ``conv=`` slot and raises ``ValueError``); this
module implements the evident intent: two sub-operators
``ConvOperator(domain[i], order, scale, taylor_order)`` combined exactly as the reference
``__call__`` bodies do.  ``Divergence`` and ``Curl`` are evaluated in ONE fused HIP pass
(``pre_residual_linear2_f32``) when both kernels are 3x3x3 stars, otherwise as two
``pre_stencil3d_f32`` passes plus a device add.
"""
from __future__ import annotations

import ctypes

import torch

from . import _dispatch, _lib
from .convops_2d import ConvOperator


def dot(a, b):
    return a[0] * b[0] + a[1] * b[1]


def cross(a, b):
    return a[0] * b[1] + a[1] * b[0]        # '+', as in Utils/VectorConvOps.py:21-22


def vectorize(a, b):
    return torch.stack((a, b))


def _linear2_fused(f0, k0, f1, k1, ratio, flags):
    d0, d1 = _dispatch.dense27(k0), _dispatch.dense27(k1)
    if d0 is None or d1 is None or f0.shape != f1.shape or f0.dim() != 4 or f0.numel() == 0:
        return None
    _dispatch._check_field(f0)
    _dispatch._check_field(f1)
    a, origin = _dispatch.to_device(f0)
    b, _ = _dispatch.to_device(f1)
    if not _lib.streamable(a, b):
        a, b = a.contiguous(), b.contiguous()
    out = _lib.empty_like_layout(a, score_rows=bool(flags & _lib.PRE_FLAG_ABS) and origin is None)
    fa, fb, fo = _lib.field(a), _lib.field(b), _lib.field(out)
    with torch.cuda.device(a.device):
        rc = _lib.load().pre_residual_linear2_f32(ctypes.byref(fa), ctypes.byref(fb), ctypes.byref(fo), d0, d1,
                                                  float(ratio), *a.shape, flags, _lib.stream())
    if rc == _lib.PRE_E_UNSUPPORTED:
        return None
    _lib.check(rc, "pre_residual_linear2_f32")
    return _dispatch.from_device(out, origin)


def linear2(f0, k0, f1, k1, ratio=1.0, flags=0):
    """``K0(f0) + ratio*K1(f1)`` in one streaming pass; None if the fused kernel declines.  Differentiable: a
    backward recomputes the two single-operator passes (``_dispatch.fused_or_composed``)."""
    if not isinstance(k0, torch.Tensor) or not isinstance(k1, torch.Tensor):
        return None
    with torch.no_grad():
        out = _linear2_fused(f0, k0, f1, k1, ratio, flags)
    if out is None or not _dispatch.needs_grad(f0, f1, k0, k1):
        return out

    def composed(a, ka, b, kb):
        r = _dispatch.xcorr(a, ka, 3) + ratio * _dispatch.xcorr(b, kb, 3)
        return r.abs() if flags & _lib.PRE_FLAG_ABS else r
    return _dispatch._Recompute.apply(out, composed, f0, k0, f1, k1)


class _Pair(ConvOperator):
    def __init__(self, domain=('x', 'y'), order=1, scale=1.0, taylor_order=2, requires_grad=False):
        super().__init__()
        self.grad_x = ConvOperator(domain[0], order, scale, taylor_order, requires_grad=requires_grad)
        self.grad_y = ConvOperator(domain[1], order, scale, taylor_order, requires_grad=requires_grad)


class Divergence(_Pair):
    def __call__(self, input_x, input_y):
        fused = linear2(input_x, self.grad_x.kernel, input_y, self.grad_y.kernel, 1.0)
        return fused if fused is not None else self.grad_x(input_x) + self.grad_y(input_y)


def _stack2(x0, op0, x1, op1):
    """torch.stack((op0(x0), op1(x1))): when nothing needs a gradient and the fields live on the device, the two
    stencil passes write straight into the two slots of the result."""
    ok = all(hasattr(o, "kernel") and o.conv == o.convolution for o in (op0, op1)) and x0.shape == x1.shape and x0.dim() == 4 \
        and x0.is_cuda and x1.is_cuda and not _dispatch.needs_grad(x0, x1, op0.kernel, op1.kernel)
    if not ok:
        return torch.stack((op0(x0), op1(x1)))
    out = torch.empty((2,) + tuple(x0.shape), dtype=torch.float32, device=x0.device)
    _dispatch._xcorr_impl(x0, op0.kernel, 3, 0, out=out[0])
    _dispatch._xcorr_impl(x1, op1.kernel, 3, 0, out=out[1])
    return out


class Gradient(_Pair):
    def __call__(self, input_x, input_y=None):
        if input_y is None:
            input_y = input_x
        return _stack2(input_x, self.grad_x, input_y, self.grad_y)


class Curl(_Pair):
    def __call__(self, input_x, input_y):
        # grad_x(input_y) - grad_y(input_x)
        fused = linear2(input_y, self.grad_x.kernel, input_x, self.grad_y.kernel, -1.0)
        return fused if fused is not None else self.grad_x(input_y) - self.grad_y(input_x)


class Laplace(ConvOperator):
    def __init__(self, domain=('x', 'y'), order=2, scale=1.0, taylor_order=2, requires_grad=False):
        super().__init__()
        self.laplace = ConvOperator(domain, order, scale, taylor_order, requires_grad=requires_grad)

    def __call__(self, input_x, input_y=None):
        if input_y is None:
            input_y = input_x
        return _stack2(input_x, self.laplace, input_y, self.laplace)

"""Host-side marshalling between the Python operator surface and the C ABI.

Nothing here computes field values: kernels (<= 343 floats) are inspected on the host to
build tap lists, tensors are handed to ``libcp_pre_hip.so`` by device pointer + strides on
the current HIP stream.  CPU tensors (what every ``Marginal/`` / ``Joint/`` script passes)
are staged through the GPU and the result is returned on the caller's device - the
arithmetic always runs in the HIP kernels.
"""
from __future__ import annotations

import ctypes
import weakref

import numpy as np
import torch

from . import _lib

_DTYPE_NAMES = {torch.float64: "Double", torch.float16: "Half", torch.bfloat16: "BFloat16", torch.int64: "Long",
                torch.int32: "Int"}


def _dtype_error(dtype):
    # wording of the RuntimeError F.conv3d raises for a dtype mismatch with the fp32 kernel
    return RuntimeError(f"expected scalar type Float but found {_DTYPE_NAMES.get(dtype, str(dtype))}")


_kernel_cache = {}   # id(tensor) -> (weakref, version, ndarray); device kernels only


def host_kernel(kernel):
    """fp32 numpy copy of an operator kernel.  Device kernels are cached per tensor object
    and ``_version`` so the D2H copy (a sync) happens once per distinct kernel value."""
    if not isinstance(kernel, torch.Tensor):
        raise TypeError("operator kernel must be a torch.Tensor")
    if kernel.dtype != torch.float32:
        raise _dtype_error(kernel.dtype)
    if not kernel.is_cuda:
        return np.ascontiguousarray(kernel.detach().numpy())
    key = id(kernel)
    hit = _kernel_cache.get(key)
    if hit is not None and hit[0]() is kernel and hit[1] == kernel._version:
        return hit[2]
    arr = kernel.detach().cpu().numpy().copy()
    if len(_kernel_cache) > 256:
        for k in [k for k, v in _kernel_cache.items() if v[0]() is None]:
            del _kernel_cache[k]
    _kernel_cache[key] = (weakref.ref(kernel), kernel._version, arr)
    return arr


def taps_of(karr):
    """(weights, offsets) of the non-zero taps; offset = index - extent//2 (zero padding
    ``k//2`` of ``Utils/ConvOps_2d.py:149`` / ``Utils/ConvOps_1d.py:150``)."""
    if any(s % 2 == 0 for s in karr.shape):
        raise NotImplementedError("even kernel extents change the output shape in the reference; not supported")
    if any(s > 7 for s in karr.shape):
        raise NotImplementedError("kernel extents above 7 are not supported")
    idx = np.argwhere(karr != 0)
    w = karr[tuple(idx.T)].astype(np.float32) if len(idx) else np.zeros((0,), np.float32)
    off = idx - np.array([s // 2 for s in karr.shape])
    return w, off.astype(np.int32)


def to_device(t):
    """(device tensor, original device).  Raises without a GPU: no CPU fallback.

    Views with a unit-stride axis - the reference layout, and the surrogate's native
    [BS,F,Nx,Ny,Nt] layout seen through ``permute(0,1,4,2,3)`` (Marginal/Wave_Residuals_CP.py:216),
    whose fastest axis is Nt - are consumed in place: the library relabels its axes and the result
    gets the same memory layout.  Only large views with NO unit-stride axis (e.g. ``x[..., ::2]``)
    are re-laid out by a device copy first; small ones go to the generic strided kernel."""
    if t.is_cuda:
        if t.dim() >= 3 and t.numel() >= (1 << 16) and not _lib.streamable(t):
            t = t.contiguous()
        return t, None
    _lib.require_gpu()
    return t.cuda(), t.device


def from_device(out, origin):
    return out if origin is None else out.to(origin)


def _check_field(field):
    if not isinstance(field, torch.Tensor):
        raise TypeError("field must be a torch.Tensor")
    if field.dtype != torch.float32:
        raise _dtype_error(field.dtype)


class _XCorrFn(torch.autograd.Function):
    """Autograd wrapper (``Physics_Informed/Wave_FNO_PI.py:202-228,257-269`` puts
    ``D(y_out)`` in a loss on device).  The adjoint of a zero-padded cross-correlation is the
    cross-correlation with the kernel flipped along every axis - the same HIP kernel with
    mirrored taps.  The kernel gradient (only if ``kernel.requires_grad``) is k^nd shifted
    inner products, composed from torch device ops."""

    @staticmethod
    def forward(ctx, field, kernel, nd):
        ctx.save_for_backward(field, kernel)
        ctx.nd = nd
        return _xcorr_impl(field, kernel, nd, 0)

    @staticmethod
    def backward(ctx, gout):
        field, kernel = ctx.saved_tensors
        nd = ctx.nd
        gf = gk = None
        if ctx.needs_input_grad[0]:
            gf = _xcorr_impl(gout.contiguous(), torch.flip(kernel.detach(), dims=tuple(range(nd))), nd, 0)
            gf = gf.reshape(field.shape)                    # a [BS,1,Nt,Nx] field has its channel squeezed in the output
        if ctx.needs_input_grad[1]:
            g = gout.to(field.device) if gout.device != field.device else gout
            x = field[:, 0] if field.dim() == nd + 2 else field
            g = g[:, 0] if g.dim() == nd + 2 else g
            gk = _kernel_grad_hip(x, g, kernel, nd)
            if gk is None:                                  # extents other than 1/3, or no unit stride: shifted products
                gk = torch.zeros_like(kernel, dtype=torch.float32, device=g.device)
                ext = x.shape[1:]
                for idx in torch.cartesian_prod(*[torch.arange(s) for s in kernel.shape]).tolist():
                    off = [i - s // 2 for i, s in zip(idx, kernel.shape)]
                    src = tuple(slice(max(0, o), e + min(0, o)) for o, e in zip(off, ext))
                    dst = tuple(slice(max(0, -o), e - max(0, o)) for o, e in zip(off, ext))
                    if all(sl.stop > sl.start for sl in src):
                        gk[tuple(idx)] = (g[(slice(None),) + dst] * x[(slice(None),) + src]).sum()
            gk = gk.to(kernel.device)
        return gf, gk, None


def _kernel_grad_hip(x, g, kernel, nd):
    """d loss / d kernel in ONE pass (``pre_stencil3d_wgrad_f32``) for device tensors and kernel extents in {1, 3};
    None if the library declines (the caller then composes k^nd shifted products from torch ops)."""
    if not (x.is_cuda and g.is_cuda and x.dtype == torch.float32 and g.dtype == torch.float32):
        return None
    if any(s not in (1, 3) for s in kernel.shape):
        return None
    if nd == 2:                                             # [B,T,X] with a (kt,kx) kernel == [1,B,T,X] with (1,kt,kx)
        x4, g4, ext = x.unsqueeze(0), g.unsqueeze(0), (1,) + tuple(kernel.shape)
    else:
        x4, g4, ext = x, g, tuple(kernel.shape)
    if x4.stride(-1) != 1:
        x4 = x4.contiguous()
    if g4.stride(-1) != 1:
        g4 = g4.contiguous()
    gk = torch.zeros(ext, dtype=torch.float64, device=x.device)
    fx, fg = _lib.field(x4), _lib.field(g4)
    with torch.cuda.device(x.device):
        rc = _lib.load().pre_stencil3d_wgrad_f32(ctypes.byref(fx), ctypes.byref(fg), *ext, *x4.shape, _lib.ptr(gk), _lib.stream())
    if rc == _lib.PRE_E_UNSUPPORTED:
        return None
    _lib.check(rc, "pre_stencil3d_wgrad_f32")
    return gk.to(torch.float32).reshape(tuple(kernel.shape))


def xcorr(field, kernel, nd, flags=0):
    """Zero-padded single-channel cross-correlation of ``field`` with ``kernel``.

    nd=3: field [BS,Nt,Nx,Ny], kernel k*k*k.   nd=2: field [BS,Nt,Nx] (or [BS,1,Nt,Nx]), kernel k*k.
    Differentiable w.r.t. both arguments when either requires grad.
    """
    if flags == 0 and torch.is_grad_enabled() and isinstance(field, torch.Tensor) and isinstance(kernel, torch.Tensor) \
            and (field.requires_grad or kernel.requires_grad):
        return _XCorrFn.apply(field, kernel, nd)
    return _xcorr_impl(field, kernel, nd, flags)


class _Recompute(torch.autograd.Function):
    """Attach a result computed by a fused (non-differentiable) HIP pass to the autograd graph.  If backward is
    ever called, the same expression is recomputed through the composed, differentiable route (single-operator
    HIP passes + torch ops) and differentiated; a forward that is never differentiated - the common case: model
    outputs that merely carry ``requires_grad``, the reference's always-grad-requiring spatial kernels - costs
    nothing extra."""

    @staticmethod
    def forward(ctx, out, composed, *tensors):
        ctx.composed = composed
        ctx.save_for_backward(*tensors)
        return out.view_as(out)

    @staticmethod
    def backward(ctx, gout):
        needs = ctx.needs_input_grad[2:]
        with torch.enable_grad():
            ins = [t.detach().requires_grad_(n) for t, n in zip(ctx.saved_tensors, needs)]
            y = ctx.composed(*ins)
            wanted = [i for i, n in zip(ins, needs) if n]
            grads = iter(torch.autograd.grad(y, wanted, gout.to(y.device), allow_unused=True)) if wanted else iter(())
        return (None, None) + tuple(next(grads) if n else None for n in needs)


def fused_or_composed(fused, composed, *tensors):
    """``fused()``: the one-pass HIP route, or None if it declines.  ``composed(*tensors)``: the same expression
    from differentiable pieces.  The fused result is used whenever it exists; gradients stay available."""
    with torch.no_grad():
        out = fused()
    if out is None:
        return composed(*tensors)
    return _Recompute.apply(out, composed, *tensors) if needs_grad(*tensors) else out


def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


def _xcorr_impl(field, kernel, nd, flags=0, out=None):
    """``out``: optional fp32 device tensor of the field's shape to write into (e.g. one slot of a stacked result)."""
    _check_field(field)
    karr = host_kernel(kernel)
    if karr.ndim != nd:
        raise RuntimeError(f"expected a {nd}-D kernel, got shape {tuple(karr.shape)}")
    w, off = taps_of(karr)
    if nd == 2 and field.dim() == 4:            # [BS,1,Nt,Nx]: the reference squeezes the channel after conv2d
        if field.shape[1] != 1:
            raise RuntimeError("expected a single-channel [BS,1,Nt,Nx] field")
        field = field[:, 0]
    if field.dim() != nd + 1:
        raise RuntimeError(f"expected a {nd + 1}-D field [BS,Nt,Nx{',Ny' if nd == 3 else ''}], got {tuple(field.shape)}")
    lib = _lib.load()
    dev, origin = to_device(field)
    if out is None:
        out = _lib.empty_like_layout(dev, score_rows=bool(flags & _lib.PRE_FLAG_ABS) and origin is None)
    elif not (out.is_cuda and out.shape == dev.shape and out.dtype == torch.float32):
        raise ValueError("out must be an fp32 device tensor of the field's shape")
    if out.numel() == 0:
        return from_device(out, origin)
    wv = _lib.farr(w) if len(w) else (ctypes.c_float * 1)()
    ov = _lib.iarr32(off.reshape(-1)) if len(w) else (ctypes.c_int32 * 1)()
    with torch.cuda.device(dev.device):
        if nd == 3:
            f, o = _lib.field(dev), _lib.field(out)
            rc = lib.pre_stencil3d_f32(ctypes.byref(f), ctypes.byref(o), wv, ov, len(w), *dev.shape, flags, _lib.stream())
            _lib.check(rc, "pre_stencil3d_f32")
        else:
            rc = lib.pre_stencil2d_f32(_lib.ptr(dev), _lib.iarr64(dev.stride()), _lib.ptr(out), _lib.iarr64(out.stride()),
                                       wv, ov, len(w), *dev.shape, flags, _lib.stream())
            _lib.check(rc, "pre_stencil2d_f32")
    return from_device(out, origin)


def dense27(kernel):
    """27 host floats of a 3x3x3 operator kernel, or None if it has another shape."""
    k = host_kernel(kernel)
    if k.shape != (3, 3, 3):
        return None
    return _lib.farr(k.reshape(-1))


def dense9(kernel):
    k = host_kernel(kernel)
    if k.shape != (3, 3):
        return None
    return _lib.farr(k.reshape(-1))

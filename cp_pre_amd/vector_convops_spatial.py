"""Vector calculus on 2-D spatial fields [BS,Nvar,Nx,Ny] with boundary conditions: drop-in for
``Utils/VectorConvOps_Spatial.py:17-165`` (``Gradient, Laplace, Divergence, Curl, Vector_Gradient``).

Reference recipe per operator: ``bc.pad_signal(input)`` then a VALID conv per sub-operator.  Here,
when the sub-operator kernels are 3x3 crosses and every side's boundary type has a mapping
(periodic / dirichlet / neumann / outflow / symmetric), padding and stencil are ONE HIP pass
(``pre_spatial2d_bc_f32``, ``pre_spatial2d_linear2_bc_f32``): the kernel reads the mapped cell in
place of the out-of-domain neighbour, nothing is padded or copied.  Anything else (5x5 / 7x7 Taylor
stencils, 'free_slip', non-streamable views, autograd) takes the reference recipe literally with
device ops + the HIP stencil pass.

Kept from the reference: ``Gradient`` and ``Vector_Gradient`` build their sub-operators on
``torch.device("cuda")`` with ``requires_grad=True`` regardless of the ``device`` argument
(``:37-38,150-151``); ``cross`` uses ``+``; ``D_y`` differences along Nx (see convops_spatial).
"""
from __future__ import annotations

import ctypes

import torch

from . import _dispatch, _lib
from .boundary_conditions import BoundaryManager
from .convops_spatial import ConvOperator, valid_conv

_BC_MODE = {'dirichlet': 0, 'neumann': 1, 'outflow': 1, 'periodic': 2, 'symmetric': 3}


def dot(a, b):
    return a[:, 0:1] * b[:, 0:1] + a[:, 1:2] * b[:, 1:2]


def cross(a, b):
    return a[:, 0:1] * b[:, 1:2] + a[:, 1:2] * b[:, 0:1]


def vectorize(a, b):
    return torch.cat((a, b), dim=1)


def _bc_struct(bc):
    """``pre_bc_t`` for a BoundaryManager, or None if a side has no fused mapping.

    ``pad_signal`` pads left, right, top, bottom IN SEQUENCE (``boundary_conditions.py:97-179``), and
    its 'periodic' right / bottom padding copies the first column / row of the ALREADY left / top
    padded array - i.e. the left / top padding cell, not the field's first cell.  So the effective
    neighbour of the last column under right='periodic' depends on the left side's type: left periodic
    -> the last column itself, left dirichlet -> that constant, left neumann/outflow -> column 0 (a true
    wrap), left symmetric -> column 1 (no fused mapping).  Reproduced here, not corrected."""
    if bc.pad_left != 1 or bc.pad_top != 1:
        return None
    modes, values = [0] * 4, [0.0] * 4
    for lo, hi, i in (('left', 'right', 0), ('top', 'bottom', 2)):
        tl, th = bc.boundary_types[lo], bc.boundary_types[hi]
        if tl not in _BC_MODE or th not in _BC_MODE:
            return None
        modes[i], values[i] = _BC_MODE[tl], float(bc.boundary_values[lo])
        modes[i + 1], values[i + 1] = _BC_MODE[th], float(bc.boundary_values[hi])
        if th == 'periodic':
            if tl == 'periodic':
                modes[i + 1] = 1                                   # replicate: the last cell itself
            elif tl == 'dirichlet':
                modes[i + 1], values[i + 1] = 0, float(bc.boundary_values[lo])
            elif tl in ('neumann', 'outflow'):
                modes[i + 1] = 2                                   # first cell: a true wrap
            else:
                return None
    return _lib.PreBC((ctypes.c_int * 4)(*modes), (ctypes.c_float * 4)(*values))


def _plane_view(x):
    """[BS,1,Nx,Ny] -> [BS,Nx,Ny] device view for the fused entries, or None."""
    if x.dim() != 4 or x.shape[1] != 1 or x.dtype != torch.float32:
        return None
    return x[:, 0]


def _fused1(x, op, bc, dst=None):
    """op(bc.pad_signal(x)) in one pass, or None.  ``dst``: optional [BS,1,Nx,Ny] device view to write into (a
    channel of a stacked result)."""
    st = _bc_struct(bc)
    k = _dispatch.dense9(op.kernel) if hasattr(op, "kernel") else None
    if st is None or k is None or _plane_view(x) is None:
        return None
    dev, origin = _dispatch.to_device(x)
    v = _plane_view(dev)
    out = torch.empty(dev.shape, dtype=torch.float32, device=dev.device) if dst is None else dst
    with torch.cuda.device(dev.device):
        rc = _lib.load().pre_spatial2d_bc_f32(_lib.ptr(v), _lib.iarr64(v.stride()), _lib.ptr(out),
                                              _lib.iarr64(out[:, 0].stride()), k, ctypes.byref(st), *v.shape, 0, _lib.stream())
    if rc == _lib.PRE_E_UNSUPPORTED:
        return None
    _lib.check(rc, "pre_spatial2d_bc_f32")
    return _dispatch.from_device(out, origin)


def _fused2(x0, op0, x1, op1, ratio, bc):
    """op0(pad(x0)) + ratio*op1(pad(x1)) in one pass, or None."""
    st = _bc_struct(bc)
    if st is None or not (hasattr(op0, "kernel") and hasattr(op1, "kernel")):
        return None
    k0, k1 = _dispatch.dense9(op0.kernel), _dispatch.dense9(op1.kernel)
    if k0 is None or k1 is None:
        return None
    if _plane_view(x0) is None or _plane_view(x1) is None or x0.shape != x1.shape:
        return None
    d0, origin = _dispatch.to_device(x0)
    d1, _ = _dispatch.to_device(x1)
    v0, v1 = _plane_view(d0), _plane_view(d1)
    out = torch.empty(d0.shape, dtype=torch.float32, device=d0.device)
    with torch.cuda.device(d0.device):
        rc = _lib.load().pre_spatial2d_linear2_bc_f32(_lib.ptr(v0), _lib.iarr64(v0.stride()), _lib.ptr(v1), _lib.iarr64(v1.stride()),
                                                      _lib.ptr(out), _lib.iarr64(out[:, 0].stride()), k0, k1, float(ratio),
                                                      ctypes.byref(st), *v0.shape, 0, _lib.stream())
    if rc == _lib.PRE_E_UNSUPPORTED:
        return None
    _lib.check(rc, "pre_spatial2d_linear2_bc_f32")
    return _dispatch.from_device(out, origin)


def _apply(x, op, bc):
    """op(bc.pad_signal(x)): fused pad + stencil pass; differentiable through the pad-then-conv recipe."""
    if not hasattr(op, "kernel") or op.conv != op.convolution:          # kernel-less operator, or conv='spectral'
        return op(bc.pad_signal(x))
    return _dispatch.fused_or_composed(lambda: _fused1(x, op, bc), lambda xx, kk: valid_conv(bc.pad_signal(xx), kk), x, op.kernel)


def _stack2(x0, op0, x1, op1, bc):
    """cat(op0(pad(x0)), op1(pad(x1)), dim=1): both fused passes write straight into the two channels."""
    def composed(a, ka, b, kb):
        return torch.cat((valid_conv(bc.pad_signal(a), ka), valid_conv(bc.pad_signal(b), kb)), dim=1)
    ok = all(hasattr(o, "kernel") and o.conv == o.convolution for o in (op0, op1)) and x0.shape == x1.shape \
        and x0.dim() == 4 and x0.shape[1] == 1 and x0.is_cuda and x1.is_cuda and x0.device == x1.device
    if not ok:
        return torch.cat((_apply(x0, op0, bc), _apply(x1, op1, bc)), dim=1)

    def fused():
        out = torch.empty((x0.shape[0], 2) + tuple(x0.shape[2:]), dtype=torch.float32, device=x0.device)
        if _fused1(x0, op0, bc, out[:, 0:1]) is None or _fused1(x1, op1, bc, out[:, 1:2]) is None:
            return None
        return out
    return _dispatch.fused_or_composed(fused, composed, x0, op0.kernel, x1, op1.kernel)


def _apply2(x0, op0, x1, op1, ratio, bc):
    """op0(pad(x0)) + ratio * op1(pad(x1)), fused when possible, differentiable either way."""
    if not (hasattr(op0, "kernel") and hasattr(op1, "kernel")) or op0.conv != op0.convolution or op1.conv != op1.convolution:
        return op0(bc.pad_signal(x0)) + ratio * op1(bc.pad_signal(x1))
    return _dispatch.fused_or_composed(
        lambda: _fused2(x0, op0, x1, op1, ratio, bc),
        lambda a, ka, b, kb: valid_conv(bc.pad_signal(a), ka) + ratio * valid_conv(bc.pad_signal(b), kb),
        x0, op0.kernel, x1, op1.kernel)


class _WithBC(ConvOperator):
    def _set_bc(self, taylor_order, boundary_cond):
        self.bc = BoundaryManager(kernel_size=(taylor_order + 1, taylor_order + 1))
        self.bc.set_all_boundaries(bc_type=boundary_cond)


class Gradient(_WithBC):          # 1 -> 2
    def __init__(self, domain=('x', 'y'), order=1, scale=1.0, taylor_order=2, boundary_cond='periodic', conv='direct',
                 device=torch.device("cpu"), requires_grad=False):
        super().__init__()
        self.grad_x = ConvOperator(domain[0], order, scale, taylor_order, conv, device=torch.device("cuda"), requires_grad=True)
        self.grad_y = ConvOperator(domain[1], order, scale, taylor_order, conv, device=torch.device("cuda"), requires_grad=True)
        self._set_bc(taylor_order, boundary_cond)

    def __call__(self, input_x, input_y=None):
        if input_y is None:
            input_y = input_x
        return _stack2(input_x, self.grad_x, input_y, self.grad_y, self.bc)


class Laplace(_WithBC):           # 1 -> 1 (scalar) or 2 -> 2 (vector)
    def __init__(self, domain=('x', 'y'), order=2, scale=1.0, taylor_order=2, boundary_cond='periodic', scalar=True,
                 conv='direct', device='cpu', requires_grad=False):
        super().__init__()
        self.laplace = ConvOperator(domain, order, scale, taylor_order, conv, device, requires_grad)
        self.scalar = scalar
        self._set_bc(taylor_order, boundary_cond)

    def __call__(self, input_x, input_y=None):
        if self.scalar == True:                    # noqa: E712
            return _apply(input_x, self.laplace, self.bc)
        if input_y is None:
            input_y = input_x
        return _stack2(input_x, self.laplace, input_y, self.laplace, self.bc)


class Divergence(_WithBC):        # 2 -> 1
    def __init__(self, domain=('x', 'y'), order=1, scale=1.0, taylor_order=2, boundary_cond='periodic', conv='direct',
                 device='cpu', requires_grad=False):
        super().__init__()
        self.grad_x = ConvOperator(domain[0], order, scale, taylor_order, conv, device, requires_grad)
        self.grad_y = ConvOperator(domain[1], order, scale, taylor_order, conv, device, requires_grad)
        self._set_bc(taylor_order, boundary_cond)

    def __call__(self, input_x, input_y):
        return _apply2(input_x, self.grad_x, input_y, self.grad_y, 1.0, self.bc)


class Curl(_WithBC):              # 2 -> 1
    def __init__(self, domain=('x', 'y'), order=1, scale=1.0, taylor_order=2, boundary_cond='periodic', conv='direct',
                 device='cpu', requires_grad=False):
        super().__init__()
        self.grad_x = ConvOperator(domain[0], order, scale, taylor_order, conv, device, requires_grad)
        self.grad_y = ConvOperator(domain[1], order, scale, taylor_order, conv, device, requires_grad)
        self._set_bc(taylor_order, boundary_cond)

    def __call__(self, input_x, input_y):
        return _apply2(input_y, self.grad_x, input_x, self.grad_y, -1.0, self.bc)      # grad_x(y) - grad_y(x)


class Vector_Gradient(_WithBC):   # 2 -> 1
    def __init__(self, domain=('x', 'y'), order=1, scale=1.0, taylor_order=2, boundary_cond='periodic', conv='direct',
                 device=torch.device("cpu"), requires_grad=False):
        self.grad_x = ConvOperator(domain[0], order, scale, taylor_order, conv, device=torch.device("cuda"), requires_grad=True)
        self.grad_y = ConvOperator(domain[1], order, scale, taylor_order, conv, device=torch.device("cuda"), requires_grad=True)
        self._set_bc(taylor_order, boundary_cond)

    def __call__(self, input_x, input_y):
        gxx, gyy = _apply(input_x, self.grad_x, self.bc), _apply(input_y, self.grad_y, self.bc)
        gyx, gxy = _apply(input_x, self.grad_y, self.bc), _apply(input_y, self.grad_x, self.bc)
        return gxx ** 2 + gyy ** 2 + 2 * gyx * gxy

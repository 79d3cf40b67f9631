"""Oracle (TEST INFRASTRUCTURE): CPU restatement of the reference ConvOperator.

Follows ``Utils/ConvOps_2d.py:17-150,288-313`` ([BS,Nt,Nx,Ny] fields, dense
k*k*k kernel, ``F.conv3d``), ``Utils/ConvOps_1d.py:17-150,287-309`` ([BS,Nt,Nx]
fields, k*k kernel, ``F.conv2d``) and ``Utils/VectorConvOps.py:17-81``.
Pinned against ``tests/golden/kernels.npz`` / ``apply.npz`` (generated from the
reference itself by ``tests/golden/make_golden.py``).

Two independent evaluations of the same arithmetic are provided:
``xcorr_torch`` (the reference's own third-party call, ``F.conv3d/conv2d`` with
zero padding ``k//2``) and ``xcorr_numpy`` (explicit shifted sums in numpy,
float32 accumulate, taps in row-major order).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

# 1-D central-difference coefficient rows, centre in the middle
# (ConvOps_2d.py:19-36 / ConvOps_1d.py:20-37).  No 1/2 on the first derivative.
_AXIS_COEFFS = {0: (0.0, 1.0, 0.0), 1: (-1.0, 0.0, 1.0), 2: (1.0, -2.0, 1.0)}

# One arm (centre outwards) of the 2-D Laplacian crosses
# (ConvOps_2d.py:38-61 / ConvOps_1d.py:55-78).
_LAPLACE_ARMS = {
    2: (-4.0, (1.0,)),
    4: (-5 / 2, (4 / 3, -1 / 12)),
    6: (-49 / 18, (3 / 2, -3 / 20, 1 / 90)),
}


def get_stencil(dims, deriv_order, taylor_order=2, one_d_file=False):
    """Square float32 stencil matrix; raises like the reference for anything else.

    ``dims==1``: 3x3 with the 1-D coefficients in COLUMN 1.  In the 2-D file the
    order-0 stencil ignores ``taylor_order`` and orders 1/2 need ``taylor_order==2``
    (ConvOps_2d.py:19-36).  The 1-D file's order-3 literals are missing commas
    (ConvOps_1d.py:39-53): evaluating them raises TypeError, restated here.
    """
    if dims == 1:
        if deriv_order == 0 or (deriv_order in (1, 2) and taylor_order == 2):
            m = np.zeros((3, 3), np.float32)
            m[:, 1] = _AXIS_COEFFS[deriv_order]
            return torch.from_numpy(m)
        if one_d_file and deriv_order == 3 and taylor_order in (2, 4):
            raise TypeError("list indices must be integers or slices, not tuple")
    elif dims == 2:
        if deriv_order == 2 and taylor_order in _LAPLACE_ARMS:
            centre, arm = _LAPLACE_ARMS[taylor_order]
            h = len(arm)
            m = np.zeros((2 * h + 1, 2 * h + 1), np.float64)
            m[h, h] = centre
            for j, w in enumerate(arm, start=1):
                m[h - j, h] = m[h + j, h] = m[h, h - j] = m[h, h + j] = w
            return torch.tensor(m.tolist(), dtype=torch.float32)
    raise ValueError("Invalid stencil parameters")


def kernel_3d(stencil, axis):
    """ConvOps_2d.py:67-79 - the slab index is the literal 1, not k//2."""
    k = stencil.shape[0]
    out = torch.zeros(k, k, k)
    if axis == 0:
        out[1, :, :] = stencil
    elif axis == 1:
        out[:, 1, :] = stencil
    elif axis == 2:
        out[:, :, 1] = stencil
    else:
        raise ValueError("Invalid axis. Must be either 0, 1 or 2")
    return out


_AXIS_2D = {"t": 2, "x": 0, "y": 1, ("x", "y"): 0, ("x", "y", "t"): 0}


def build_kernel_2d(domain, order, scale=1.0, taylor_order=2):
    """Dense 3-D kernel of ``ConvOperator.__init__`` (ConvOps_2d.py:98-125) or None
    when the reference's bare ``except`` leaves the operator without ``.kernel``."""
    try:
        dims = len(domain)
        stencil = get_stencil(dims, order, taylor_order)
        if isinstance(domain, list) or domain not in _AXIS_2D:
            raise ValueError("Invalid Domain. Must be either x,y or t")
        return scale * kernel_3d(stencil, _AXIS_2D[domain])
    except Exception:
        return None


def build_kernel_1d(domain, order, scale=1.0, taylor_order=2):
    """Dense 2-D kernel of the 1-D file's ctor (ConvOps_1d.py:101-120) or None."""
    try:
        dims = len(domain)
        stencil = get_stencil(dims, order, taylor_order, one_d_file=True)
        if domain == "t" or domain == ("x", "t"):
            pass
        elif domain == "x":
            stencil = stencil.T
        else:
            raise ValueError("Invalid Domain. Must be either x or t")
        return scale * stencil
    except Exception:
        return None


def xcorr_torch(field, kernel):
    """The reference's arithmetic call (ConvOps_2d.py:149-150, ConvOps_1d.py:145-150)."""
    if kernel.dim() == 3:
        pad = tuple(s // 2 for s in kernel.shape)
        return F.conv3d(field.unsqueeze(1), kernel[None, None], padding=pad).squeeze(1)
    if field.dim() == 3:
        field = field.unsqueeze(1)
    pad = tuple(s // 2 for s in kernel.shape)
    return F.conv2d(field, kernel[None, None], padding=pad).squeeze(1)


def xcorr_numpy(field, kernel):
    """Same cross-correlation as explicit shifted sums (float32 accumulate).

    out[b, i0, i1(, i2)] = sum_a K[a] * in[b, i + a - K.shape//2], zero outside.
    Output extent follows conv semantics: n + 2*(k//2) - k + 1 per axis (equal to
    n for odd k).
    """
    x = np.asarray(field, np.float32)
    k = np.asarray(kernel, np.float32)
    nd = k.ndim
    pads = [s // 2 for s in k.shape]
    out_shape = [x.shape[0]] + [x.shape[1 + d] + 2 * pads[d] - k.shape[d] + 1 for d in range(nd)]
    xp = np.pad(x, [(0, 0)] + [(p, p) for p in pads])
    out = np.zeros(out_shape, np.float32)
    for idx in np.ndindex(*k.shape):
        w = k[idx]
        if w == 0.0:
            continue
        sl = (slice(None),) + tuple(slice(idx[d], idx[d] + out_shape[1 + d]) for d in range(nd))
        out += w * xp[sl]
    return out


class ConvOperator2D:
    """Restatement of ``Utils/ConvOps_2d.py:86-150,288-313`` (direct path only)."""

    def __init__(self, domain=None, order=None, scale=1.0, taylor_order=2, conv="direct"):
        k = build_kernel_2d(domain, order, scale, taylor_order)
        if k is not None:
            self.kernel = k
        if conv not in ("direct", "spectral"):
            raise ValueError("Unknown Convolution Method")

    def convolution(self, field, kernel=None):
        if kernel is not None:
            self.kernel = kernel
        return xcorr_torch(field, self.kernel)

    def __call__(self, field):
        return self.convolution(field, self.kernel)


class ConvOperator1D(ConvOperator2D):
    """Restatement of ``Utils/ConvOps_1d.py:89-150,287-309`` (direct path only)."""

    def __init__(self, domain=None, order=None, scale=1.0, taylor_order=2, conv="direct"):
        k = build_kernel_1d(domain, order, scale, taylor_order)
        if k is not None:
            self.kernel = k
        if conv not in ("direct", "spectral"):
            raise ValueError("Unknown Convolution Method")


# ---- Utils/VectorConvOps.py:17-81 (intended semantics; the shipped classes cannot be
# constructed because ``requires_grad`` lands in the ``conv=`` slot) ----

def dot(a, b):
    return a[0] * b[0] + a[1] * b[1]


def cross(a, b):
    return a[0] * b[1] + a[1] * b[0]          # '+' as in VectorConvOps.py:21-22


def vectorize(a, b):
    return torch.stack((a, b))


class _Pair:
    def __init__(self, domain=("x", "y"), order=1, scale=1.0, taylor_order=2):
        self.grad_x = ConvOperator2D(domain[0], order, scale, taylor_order)
        self.grad_y = ConvOperator2D(domain[1], order, scale, taylor_order)


class Divergence(_Pair):
    def __call__(self, ix, iy):
        return self.grad_x(ix) + self.grad_y(iy)


class Gradient(_Pair):
    def __call__(self, ix, iy=None):
        iy = ix if iy is None else iy
        return torch.stack((self.grad_x(ix), self.grad_y(iy)))


class Curl(_Pair):
    def __call__(self, ix, iy):
        return self.grad_x(iy) - self.grad_y(ix)


class Laplace:
    def __init__(self, domain=("x", "y"), order=2, scale=1.0, taylor_order=2):
        self.laplace = ConvOperator2D(domain, order, scale, taylor_order)

    def __call__(self, ix, iy=None):
        iy = ix if iy is None else iy
        return torch.stack((self.laplace(ix), self.laplace(iy)))

"""``ConvOperator`` for [BS,Nt,Nx,Ny] fields: drop-in for ``Utils/ConvOps_2d.py:86-313``.

Same constructor signature, attributes (``.kernel`` is a plain, caller-owned, mutable
``torch.Tensor``), additive-kernel idiom (README.md:47-54), quirks and error behaviour as
the reference; ``convolution`` is evaluated by the HIP library (``pre_stencil3d_f32``)
instead of ``F.conv3d`` (``Utils/ConvOps_2d.py:149``).

Reference behaviours kept on purpose (golden-tested):
  * ``domain='y'`` builds the same kernel as ``domain='t'`` (``kernel[:,1,:] = stencil``,
    ``Utils/ConvOps_2d.py:72-73``); pass ``y_axis_fix=True`` to ``kernel_3d`` /
    ``ConvOperator`` for the physically intended Ny stencil (NOT reference parity);
  * Taylor-4/6 stencils land on kernel slab 1, not the centre (``:70-71``);
  * a constructor that cannot build a kernel swallows the error and leaves the operator
    without ``.kernel`` (``:98-125``); only a bad ``conv`` raises ``ValueError``;
  * ``convolution(field, kernel)`` with a kernel REPLACES ``self.kernel`` (``:146-147``).
"""
from __future__ import annotations

import torch

from . import _dispatch

# 1-D central-difference rows (no 1/2 on the first derivative, ConvOps_2d.py:31-36) and the
# arms of the Laplacian crosses (ConvOps_2d.py:38-61), centre value first.
_ROWS = {0: (0.0, 1.0, 0.0), 1: (-1.0, 0.0, 1.0), 2: (1.0, -2.0, 1.0)}
_CROSS = {2: (-4.0, 1.0), 4: (-5 / 2, 4 / 3, -1 / 12), 6: (-49 / 18, 3 / 2, -3 / 20, 1 / 90)}


def get_stencil(dims, deriv_order, taylor_order=2):
    """Square fp32 stencil matrix (``Utils/ConvOps_2d.py:17-63``)."""
    if dims == 1 and (deriv_order == 0 or (deriv_order in (1, 2) and taylor_order == 2)):
        m = torch.zeros(3, 3, dtype=torch.float32)
        m[:, 1] = torch.tensor(_ROWS[deriv_order], dtype=torch.float32)
        return m
    if dims == 2 and deriv_order == 2 and taylor_order in _CROSS:
        centre, *arm = _CROSS[taylor_order]
        h = len(arm)
        m = torch.zeros(2 * h + 1, 2 * h + 1, dtype=torch.float64)
        m[h, h] = centre
        for j, w in enumerate(arm, start=1):
            for idx in ((h - j, h), (h + j, h), (h, h - j), (h, h + j)):
                m[idx] = w
        return m.to(torch.float32)
    raise ValueError("Invalid stencil parameters")


def kernel_3d(stencil, axis, y_axis_fix=False):
    """Embed the stencil in a k*k*k kernel (axes Nt,Nx,Ny) exactly like
    ``Utils/ConvOps_2d.py:67-79``: the slab index is the literal 1."""
    k = stencil.shape[0]
    kern = torch.zeros(k, k, k)
    if axis == 0:
        kern[1, :, :] = stencil
    elif axis == 1:
        if y_axis_fix:                      # intended: taps along Ny at the centre slab
            kern[k // 2, k // 2, :] = stencil[:, k // 2]
        else:
            kern[:, 1, :] = stencil
    elif axis == 2:
        kern[:, :, 1] = stencil
    else:
        raise ValueError("Invalid axis. Must be either 0, 1 or 2")
    return kern


def pad_kernel(grid, kernel):
    """``Utils/ConvOps_2d.py:81-84``."""
    k = kernel.shape[0]
    nt, nx, ny = grid.shape[1], grid.shape[2], grid.shape[3]
    return torch.nn.functional.pad(kernel, (0, nx - k, 0, ny - k, 0, nt - k), "constant", 0)


_AXIS = {"t": 2, "x": 0, "y": 1, ("x", "y"): 0, ("x", "y", "t"): 0}


class ConvOperator:
    """Finite-difference operator applied as a zero-padded cross-correlation on the GPU."""

    def __init__(self, domain=None, order=None, scale=1.0, taylor_order=2, conv='direct', device='cpu',
                 requires_grad=False, y_axis_fix=False):
        try:
            self.domain = domain
            self.dims = len(self.domain)
            self.order = order
            self.stencil = get_stencil(self.dims, self.order, taylor_order)
            if isinstance(domain, list) or domain not in _AXIS:
                raise ValueError("Invalid Domain. Must be either x,y or t")
            self.axis = _AXIS[domain]
            self.kernel = (scale * kernel_3d(self.stencil, self.axis, y_axis_fix)).to(device)
            if requires_grad == True:                      # noqa: E712 - sets an attribute, like the reference (:121-122)
                self.kernel.requires_grad_ = True
        except Exception:                                  # the reference's bare except (:124-125)
            pass

        if conv == 'direct':
            self.conv = self.convolution
        elif conv == 'spectral':
            self.conv = self.spectral_convolution
        else:
            raise ValueError("Unknown Convolution Method")

    # ---- hot path -------------------------------------------------------------------
    def convolution(self, field, kernel=None):
        """``F.conv3d(field[:,None], K[None,None], padding=k//2).squeeze(1)`` on the HIP path."""
        if kernel is not None:
            self.kernel = kernel
        return _dispatch.xcorr(field, self.kernel, nd=3)

    # ---- spectral family (torch.fft / hipFFT pass-through) --------------
    def spectral_convolution(self, field, kernel=None, inverse=False):
        from . import _spectral
        if kernel is not None:
            self.kernel = kernel
        return _spectral.fft_xcorr(field, self.kernel, inverse=inverse)

    def differentiate(self, field, kernel=None, correlation=False, slice_pad=True):
        from . import _spectral
        if kernel is not None:
            self.kernel = kernel
        return _spectral.differentiate(field, self.kernel, correlation, slice_pad)

    def integrate(self, field, kernel=None, correlation=False, slice_pad=False, eps=1e-6):
        from . import _spectral
        if kernel is not None:
            self.kernel = kernel
        return _spectral.integrate(field, self.kernel, correlation, slice_pad, eps)

    def forward(self, field):
        return self.conv(field, self.kernel)

    def __call__(self, inputs):
        return self.forward(inputs)

"""``ConvOperator`` for 2-D spatial fields [BS,1,Nx,Ny]: drop-in for ``Utils/ConvOps_Spatial.py:20-136``
(used by ``Active_Learning/CNS.py`` through ``Utils/VectorConvOps_Spatial.py``).

Differences from the [BS,Nt,Nx,Ny] operator that are reproduced on purpose:
  * the first-derivative stencil is 1/2-scaled (``[-1/2, 0, 1/2]``, ``:34-39``);
  * ``convolution`` is a VALID conv (``F.conv2d`` without padding, ``:135``): the output loses
    ``k//2`` cells per side; boundary conditions are the caller's business (``BoundaryManager``);
  * ``kernel = scale * stencil`` with ``scale`` a float32 0-d tensor that requires grad (``:103-105``);
  * the stencil is NOT transposed for ``domain='y'`` (``:101``): the reference's spatial ``D_y``
    differences along Nx, exactly like its ``D_x``;
  * default ``device=torch.device("cuda")``.
The valid conv is the zero-padded HIP stencil pass with its rim cropped (interior cells never see
the padding).
"""
from __future__ import annotations

import torch

from . import _dispatch
from .convops_2d import _CROSS, _ROWS


def get_stencil(dims, deriv_order, taylor_order=2):
    if dims == 1 and (deriv_order == 0 or (deriv_order in (1, 2) and taylor_order == 2)):
        m = torch.zeros(3, 3, dtype=torch.float32)
        col = torch.tensor(_ROWS[deriv_order], dtype=torch.float32)
        m[:, 1] = col / 2 if deriv_order == 1 else col
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


def pad_kernel(grid, kernel):
    k = kernel.shape[0]
    nt, nx, ny = grid.shape[1], grid.shape[2], grid.shape[3]
    return torch.nn.functional.pad(kernel, (0, nx - k, 0, ny - k, 0, nt - k), "constant", 0)


def valid_conv(field, kernel):
    """``F.conv2d(field, K[None,None])`` for a single-channel [BS,1,Nx,Ny] field on the HIP path."""
    if field.dim() != 4 or field.shape[1] != 1:
        raise RuntimeError(f"expected input [BS, 1, Nx, Ny] (single channel), got {tuple(field.shape)}")
    if kernel.dim() != 2:
        raise RuntimeError(f"expected a 2-D kernel, got shape {tuple(kernel.shape)}")
    p0, p1 = kernel.shape[0] // 2, kernel.shape[1] // 2
    same = _dispatch.xcorr(field, kernel.unsqueeze(0), nd=3)          # [BS,1,Nx,Ny] read as [B,T=1,X,Y]
    return same[..., p0:field.shape[2] - p0, p1:field.shape[3] - p1]


class ConvOperator:
    def __init__(self, domain=None, order=None, scale=1.0, taylor_order=2, conv='direct', device=torch.device("cuda"),
                 requires_grad=False):
        try:
            self.domain = domain
            self.dims = len(self.domain)
            self.order = order
            self.stencil = get_stencil(self.dims, self.order, taylor_order)
            if isinstance(domain, list) or domain not in ('x', 'y', ('x', 'y')):
                raise ValueError("Invalid Domain. Must be either x,y or their combination")
            self.axis = 1 if domain == 'y' else 0
            self.kernel = self.stencil.to(device)
            self.scale = torch.tensor(scale, dtype=torch.float32, device=device, requires_grad=True)
            self.kernel = self.scale * self.kernel
            if requires_grad == True:                     # noqa: E712  (attribute, not a call: reference :107-108)
                self.kernel.requires_grad_ = True
        except Exception:                                 # the reference's bare except (:110-111)
            pass

        if conv == 'direct':
            self.conv = self.convolution
        elif conv == 'spectral':
            self.conv = self.spectral_convolution
        else:
            raise ValueError("Unknown Convolution Method")

    def convolution(self, field, kernel=None):
        if kernel is not None:
            self.kernel = kernel
        return valid_conv(field, self.kernel)

    def spectral_convolution(self, field, kernel=None, inverse=False):
        from . import _spectral
        if kernel is not None:
            self.kernel = kernel
        return _spectral.fft_xcorr(field, self.kernel, inverse=inverse, keep_channel=True)

    def differentiate(self, field, kernel=None, correlation=False, slice_pad=True):
        from . import _spectral
        if kernel is not None:
            self.kernel = kernel
        return _spectral.differentiate(field, self.kernel, correlation, slice_pad, keep_channel=True)

    def integrate(self, field, kernel=None, correlation=False, slice_pad=False, eps=1e-6):
        from . import _spectral
        if kernel is not None:
            self.kernel = kernel
        return _spectral.integrate(field, self.kernel, correlation, slice_pad, eps, keep_channel=True)

    def forward(self, field):
        return self.conv(field, self.kernel)

    def __call__(self, inputs):
        return self.forward(inputs)

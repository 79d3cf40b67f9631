"""``BoundaryManager``: drop-in for ``Utils/boundary_conditions.py:7-211``.

``pad_signal`` is pure data movement (pad / cat on whatever device the signal lives on) and is kept
as device ops; the vector operators of ``vector_convops_spatial`` do not call it on the fast path -
they hand the boundary types to the HIP kernel, which maps the out-of-domain neighbours itself
(``pre_spatial2d_bc_f32``) - but it remains the reference semantics and the fallback.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

_SIDES = ('left', 'right', 'top', 'bottom')


class BoundaryManager:
    SUPPORTED_TYPES = ['dirichlet', 'neumann', 'periodic', 'symmetric', 'free_slip', 'outflow']

    def __init__(self, kernel_size):
        if isinstance(kernel_size, int):
            self.kernel_height = self.kernel_width = kernel_size
        else:
            self.kernel_height, self.kernel_width = kernel_size
        self.pad_left = self.pad_right = self.kernel_width // 2
        self.pad_top = self.pad_bottom = self.kernel_height // 2
        self.boundary_types = {s: 'periodic' for s in _SIDES}
        self.boundary_values = {s: 0.0 for s in _SIDES}

    def set_boundary_type(self, side, bc_type, value=0.0):
        if side not in _SIDES:
            raise ValueError(f"Unknown side: {side}. Use 'left', 'right', 'top', or 'bottom'")
        if bc_type.lower() not in self.SUPPORTED_TYPES:
            raise ValueError(f"Unsupported boundary type: {bc_type}")
        self.boundary_types[side] = bc_type.lower()
        self.boundary_values[side] = value

    def set_all_boundaries(self, bc_type, value=0.0):
        for side in _SIDES:
            self.set_boundary_type(side, bc_type, value)

    @staticmethod
    def _grow(r, bc, value, pad, piece, dim, front):
        if bc == 'dirichlet':
            return F.pad(r, pad, mode='constant', value=value)
        if bc in ('neumann', 'outflow'):
            return F.pad(r, pad, mode='replicate')
        if bc == 'periodic':
            return torch.cat([piece, r] if front else [r, piece], dim=dim)
        if bc == 'symmetric':
            return F.pad(r, pad, mode='reflect')
        return r                      # 'free_slip' is accepted but has no padding rule in the reference

    def pad_signal(self, signal):
        """Left, right, top, bottom - in that order, each on the result of the previous one (``:97-179``)."""
        two_d = signal.dim() == 2
        r = signal[None, None] if two_d else signal
        t, v = self.boundary_types, self.boundary_values
        if self.pad_left > 0:
            r = self._grow(r, t['left'], v['left'], (self.pad_left, 0, 0, 0), r[:, :, :, -self.pad_left:], 3, True)
        if self.pad_right > 0:
            r = self._grow(r, t['right'], v['right'], (0, self.pad_right, 0, 0), r[:, :, :, :self.pad_right], 3, False)
        if self.pad_top > 0:
            r = self._grow(r, t['top'], v['top'], (0, 0, self.pad_top, 0), r[:, :, -self.pad_top:, :], 2, True)
        if self.pad_bottom > 0:
            r = self._grow(r, t['bottom'], v['bottom'], (0, 0, 0, self.pad_bottom), r[:, :, :self.pad_bottom, :], 2, False)
        return r[0, 0] if two_d else r

    def apply_convolution(self, signal, kernel):
        from .convops_spatial import valid_conv
        padded = self.pad_signal(signal)
        if padded.dim() == 2:
            padded = padded[None, None]
        return valid_conv(padded, kernel).squeeze(0).squeeze(0)

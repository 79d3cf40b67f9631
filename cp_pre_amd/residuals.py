"""PDE residual operators (PRE) over surrogate outputs, evaluated on the MI355X.

Mirrors the residual definitions the reference writes inline per experiment script and
packages in ``Other_UQ/Evaluation/PRE_estimations.py:5-80`` (``PRE_Wave``, ``PRE_NS``,
``PRE_MHD`` keep that file's class names, constructor arguments and ``.residual(vars,
boundary)`` signature).  Every class owns reference-style ``ConvOperator`` objects
(``D_t, D_x, D_y, D_xx_yy`` ...), whose ``.kernel`` tensors may be replaced by the caller.

Two evaluation routes, same numbers:
  * fused (default): one streaming HIP pass reads each field once and writes the residual
    once (``pre_residual_*_f32``); the operators' CURRENT dense kernels are handed to the
    library, so the reference's ``D_y == D_t`` construction quirk is inherited, not re-coded;
  * composed (``fused=False``, or automatically when a kernel is not a 3x3x3 star or a view
    is not streamable): the reference expression, operator by operator, each
    ``ConvOperator`` call being a HIP stencil pass and the products/sums torch device ops.

``vars`` is [BS,F,Nt,Nx,Ny]; ``boundary=False`` crops one cell per side like the reference
(a view of the full residual).  ``absolute=True`` returns |residual| (the marginal score).
"""
from __future__ import annotations

import ctypes

import torch

from . import _dispatch, _lib
from .convops_1d import ConvOperator as ConvOperator1D
from .convops_2d import ConvOperator as ConvOperator2D

_CROP3 = (Ellipsis, slice(1, -1), slice(1, -1), slice(1, -1))
_CROP2 = (Ellipsis, slice(1, -1), slice(1, -1))


def _stage(fields):
    """Device views of the fields (+ where the result has to go back to).  Views that do not
    share a unit-stride axis are made contiguous so the fused kernels can stream them."""
    origin = None
    out = []
    for f in fields:
        _dispatch._check_field(f)
        d, o = _dispatch.to_device(f)
        origin = origin or o
        out.append(d)
    if len(out) > 1 and not _lib.streamable(*out):
        out = [d.contiguous() for d in out]
    return out, origin


def _on_device(fields, fn):
    """Composed route: stage the fields once, evaluate ``fn`` on device tensors, return home."""
    devs, origin = _stage(fields)
    return _dispatch.from_device(fn(*devs), origin)


def _fused_call(name, call):
    """Run a fused entry point; None means 'compose instead'."""
    rc = call()
    if rc == _lib.PRE_E_UNSUPPORTED:
        return False
    _lib.check(rc, name)
    return True


class _Residual2D:
    """Shared plumbing: the operator set of ``Marginal/NS_Residuals_CP.py:213-219``."""

    def __init__(self, device='cpu', fused=True, y_axis_fix=False):
        self.fused = fused
        self.D_t = ConvOperator2D(domain='t', order=1, device=device)
        self.D_x = ConvOperator2D(domain='x', order=1, device=device)
        self.D_y = ConvOperator2D(domain='y', order=1, device=device, y_axis_fix=y_axis_fix)
        self.D_x_y = ConvOperator2D(domain=('x', 'y'), order=1, device=device)     # kernel-less, as in the reference
        self.D_xx_yy = ConvOperator2D(domain=('x', 'y'), order=2, device=device)

    def _k27(self, *ops):
        ks = [_dispatch.dense27(o.kernel) for o in ops]
        return None if any(k is None for k in ks) else ks

    def _want_fused(self, *tensors):
        """The fused kernels run whatever the fields' ``requires_grad`` (``_attach`` keeps the result
        differentiable by recomputation); only operator KERNELS that require grad force the composed route, so
        that their gradients flow."""
        if any(isinstance(t, torch.Tensor) and t.numel() == 0 for t in tensors):
            return False                                    # empty batch: the composed route returns empty tensors
        return self.fused and not _dispatch.needs_grad(*[getattr(o, "kernel", None) for o in
                                                         (self.D_t, self.D_x, self.D_y, self.D_xx_yy)])


def _attach(res, fields, composed, absolute):
    """``res``: a fused result (with |.| already applied if ``absolute``), or None.  If a field requires grad the
    result is hooked into autograd: a backward recomputes ``composed`` (operator by operator) and differentiates
    that; a forward that is never differentiated pays nothing."""
    if res is None or not _dispatch.needs_grad(*fields):
        return res
    fn = (lambda *f: _on_device(f, composed).abs()) if absolute else (lambda *f: _on_device(f, composed))
    return _dispatch._Recompute.apply(res, fn, *fields)


def _finish(res, boundary, crop, absolute, already_abs):
    if absolute and not already_abs:
        res = res.abs()
    return res if boundary else res[crop]


# ======================================================================= Navier-Stokes
class NavierStokes(_Residual2D):
    """``Marginal/NS_Residuals_CP.py:203-240``: continuity and momentum residuals of (u, v, p)."""

    def __init__(self, dt, dx, dy, nu=0.001, **kw):
        super().__init__(**kw)
        self.dt, self.dx, self.dy, self.nu = dt, dx, dy, nu

    def residual_continuity(self, vars, boundary=False, absolute=False):
        u, v = vars[:, 0], vars[:, 1]
        ratio = self.dx / self.dy
        res = None
        if self._want_fused(vars):
            from .vector_convops import linear2
            res = linear2(u, self.D_x.kernel, v, self.D_y.kernel, ratio, _lib.PRE_FLAG_ABS if absolute else 0)
        done_abs = res is not None and absolute
        if res is None:
            res = _on_device((u, v), lambda u, v: self.D_x(u) + ratio * self.D_y(v))
        return _finish(res, boundary, _CROP3, absolute, done_abs)

    def residual_momentum(self, vars, boundary=False, absolute=False, out=None, skip_t_rim=False, halo_x=False):
        """``out``: optional preallocated device tensor [BS,Nt,Nx,Ny] for the uncropped residual
        (fused route only; lets a streaming driver reuse one buffer).  ``skip_t_rim``: the caller
        crops the first and last time plane anyway, so they need not be computed or stored
        (``PRE_FLAG_INTERIOR_T``; their content is then unspecified).  With ``skip_t_rim`` ``out`` may
        also be a [BS,Nt-2,Nx,Ny] tensor (contiguous, or any batch / time strides over dense planes): it then receives
        the interior planes only
        (``PRE_FLAG_OUT_INTERIOR_T``; what a t-slab driver that feeds slabs with their two halo planes
        wants) and the result is that tensor, cropped in x and y unless ``boundary``.
        ``halo_x``: ``vars`` is an x-slab ``full[:, :, :, x0:x1]`` (1 <= x0, x1 <= Nx - 1) of a larger grid whose rows
        x0 - 1 and x1 lie in the same memory: they are read as the x-neighbours of the slab's first and last row instead
        of the zero padding (``PRE_FLAG_HALO_X``), so the slab's residual rows are those of the whole grid.  What an
        x-slab driver wants: the T axis stays whole and a slab re-reads 2 rows of Nx_slab instead of 2 planes of
        Nt_slab.  Fused route only (raises otherwise)."""
        u, v, p = vars[:, 0], vars[:, 1], vars[:, 2]
        dt, dx, dy, nu = self.dt, self.dx, self.dy, self.nu
        D_t, D_x, D_y, D_xx_yy = self.D_t, self.D_x, self.D_y, self.D_xx_yy

        def composed(u, v, p):
            res_x = D_t(u)*dx*dy + u*D_x(u)*dt*dy + v*D_y(u)*dt*dx - nu*D_xx_yy(u)*dt + D_x(p)*dt*dy
            res_y = D_t(v)*dx*dy + u*D_x(v)*dt*dx + v*D_y(v)*dt*dy - nu*D_xx_yy(v)*dt + D_y(p)*dt*dx
            return res_x + res_y
        ks = self._k27(self.D_t, self.D_x, self.D_y, self.D_xx_yy) if self._want_fused(vars) else None
        if ks is not None:
            with torch.no_grad():
                (du, dv, dp), origin = _stage((u, v, p))
                if out is None:
                    out = _lib.empty_like_layout(du, score_rows=absolute and origin is None)
                interior = (skip_t_rim and out.dim() == 4 and du.shape[1] >= 3 and
                            tuple(out.shape) == (du.shape[0], du.shape[1] - 2, du.shape[2], du.shape[3]))
                if not (out.is_cuda and out.dtype == torch.float32 and (out.shape == du.shape or interior)):
                    raise ValueError("out must be an fp32 device tensor of the field shape "
                                     "(or, with skip_t_rim, of its interior planes [BS,Nt-2,Nx,Ny])")
                # (the planes of an interior out must be dense, its batch and time strides are free: a t-slab driver that
                # shards the marginal calibration hands in a TIME-MAJOR buffer [Nt-2][BS][Nx][Ny] seen as [BS,Nt-2,Nx,Ny],
                # whose planes then are the contiguous send blocks of its all-to-all - pipeline.marginal_qhat)
                dense_planes = out.dim() == 4 and out.stride(3) == 1 and out.stride(2) == out.shape[3]
                if interior and not (dense_planes and origin is None and not _dispatch.needs_grad(u, v, p)):
                    raise ValueError("an interior-plane out needs device-resident fields, dense [Nx,Ny] planes and no autograd")
                flags = (_lib.PRE_FLAG_ABS if absolute else 0) | (_lib.PRE_FLAG_INTERIOR_T if skip_t_rim else 0) | \
                        (_lib.PRE_FLAG_OUT_INTERIOR_T if interior else 0) | (_lib.PRE_FLAG_HALO_X if halo_x else 0)
                if halo_x and (origin is not None or _dispatch.needs_grad(u, v, p) or
                               any(d.data_ptr() != f.data_ptr() or d.stride(3) != 1 for d, f in zip((du, dv, dp), (u, v, p)))):
                    raise ValueError("halo_x needs device-resident, Ny-contiguous views of a larger grid and no autograd")
                fu, fv, fp, fo = _lib.field(du), _lib.field(dv), _lib.field(dp), _lib.field(out)
                with torch.cuda.device(du.device):
                    ok = _fused_call("pre_residual_ns_momentum_f32", lambda: _lib.load().pre_residual_ns_momentum_f32(
                        ctypes.byref(fu), ctypes.byref(fv), ctypes.byref(fp), ctypes.byref(fo), *ks,
                        float(dt), float(dx), float(dy), float(nu), *du.shape, flags, _lib.stream()))
                if (interior or halo_x) and not ok:
                    raise RuntimeError("pre_residual_ns_momentum_f32: an interior-plane out / halo_x needs Ny-contiguous "
                                       "views and star-shaped operator kernels")
                if interior:
                    return out if boundary else out[..., 1:-1, 1:-1]
            if ok:
                res = _attach(_dispatch.from_device(out, origin), (u, v, p), composed, absolute)
                return _finish(res, boundary, _CROP3, absolute, True)
        if halo_x:
            raise RuntimeError("halo_x: only the fused route reads the halo rows")
        return _finish(_on_device((u, v, p), composed), boundary, _CROP3, absolute, False)

    _WALLS = {'top': 0, 'bottom': 1, 'left': 2, 'right': 3}

    def periodic_bc_residual(self, u, wall='right'):
        """``Marginal/NS_Residuals_CP.py:468-478``: (one edge of every [Nx,Ny] plane minus the opposite one) * dx ->
        ``u.shape[:-2] + (Ny,)`` for 'top' / 'bottom', ``+ (Nx,)`` for 'left' / 'right'.  One ``pre_edge_residual_f32``
        launch on the view where it lies (any strides); an unknown wall raises KeyError as the reference's dict lookup
        would."""
        code = self._WALLS[wall]
        if u.dim() < 2 or u.dtype != torch.float32:
            raise RuntimeError("periodic_bc_residual: float32 field with at least the two plane axes")
        d, origin = _dispatch.to_device(u)
        lead = tuple(d.shape[:-2])
        v = d.reshape((1,) * (4 - d.dim()) + tuple(d.shape)) if d.dim() < 4 else (d if d.dim() == 4 else d.reshape(-1, *d.shape[-3:]))
        B, T, X, Y = v.shape
        out = torch.empty(lead + ((Y,) if code < 2 else (X,)), dtype=torch.float32, device=d.device)
        if out.numel() == 0:                                     # (an empty batch has no device memory to point at)
            return _dispatch.from_device(out, origin)
        with torch.cuda.device(d.device):
            f = _lib.field(v)
            _lib.check(_lib.load().pre_edge_residual_f32(ctypes.byref(f), code, float(self.dx), B, T, X, Y, _lib.ptr(out),
                                                         _lib.stream()), "pre_edge_residual_f32")
        return _dispatch.from_device(out, origin)


class PRE_NS(NavierStokes):
    """``Other_UQ/Evaluation/PRE_estimations.py:24-50``: ``PRE_NS(dt, dx, dy).residual(vars)``."""

    def __init__(self, dt, dx, dy, **kw):
        super().__init__(dt, dx, dy, nu=0.001, **kw)

    def residual(self, vars, boundary=False):
        return self.residual_momentum(vars, boundary)


# ======================================================================= MHD
_MHD_EQ = {'continuity': 0, 'momentum': 1, 'energy': 2, 'induction': 3}


class MHD(_Residual2D):
    """``Marginal/MHD_Residuals_CP.py:204-278``: ideal-MHD residuals of (rho,u,v,p,Bx,By)."""

    def __init__(self, gamma=5 / 3, **kw):
        super().__init__(**kw)
        self.gamma = gamma

    def _fused(self, eq, vars, absolute, halo_x=False, out=None):
        """``halo_x`` (every ``residual_*`` below takes it): ``vars`` is an x-slab ``full[:, :, :, x0:x1]`` whose rows
        x0 - 1 and x1 lie in the same memory and are read as x-neighbours instead of the zero padding
        (``PRE_FLAG_HALO_X``, see ``NavierStokes.residual_momentum``); fused route only.
        ``out`` (likewise): a preallocated fp32 device tensor [BS,Nt,Nx,Ny] for the uncropped residual - any batch / plane
        strides over rows that share the fields' contiguous axis (``pipeline.row_padded`` / ``time_major`` buffers: the
        sharded marginal calibration's send blocks); fused route only (raises otherwise)."""
        ks = self._k27(self.D_t, self.D_x, self.D_y) if self._want_fused(vars) else None
        if ks is None or vars.shape[1] < 6:
            if halo_x or out is not None:
                raise RuntimeError("halo_x / out: only the fused route reads the halo rows / writes a caller's buffer")
            return None
        with torch.no_grad():
            fields, origin = _stage([vars[:, i] for i in range(6)])
        if halo_x and (origin is not None or _dispatch.needs_grad(vars) or
                       any(d.data_ptr() != vars[:, i].data_ptr() or d.stride(3) != 1 for i, d in enumerate(fields))):
            raise ValueError("halo_x needs device-resident, Ny-contiguous views of a larger grid and no autograd")
        given = out is not None
        if given and not (out.is_cuda and out.dtype == torch.float32 and out.shape == fields[0].shape and origin is None
                          and not _dispatch.needs_grad(vars)):
            raise ValueError("out must be an fp32 device tensor of the field shape, the fields device-resident, no autograd")
        if not given:
            out = _lib.empty_like_layout(fields[0], score_rows=absolute and origin is None)
        arr = (_lib.PreField * 6)(*[_lib.field(f) for f in fields])
        fo = _lib.field(out)
        flags = (_lib.PRE_FLAG_ABS if absolute else 0) | (_lib.PRE_FLAG_HALO_X if halo_x else 0)
        with torch.cuda.device(out.device):
            ok = _fused_call("pre_residual_mhd_f32", lambda: _lib.load().pre_residual_mhd_f32(
                _MHD_EQ[eq], arr, ctypes.byref(fo), *ks, float(self.gamma), *out.shape, flags, _lib.stream()))
        if (halo_x or given) and not ok:
            raise RuntimeError("pre_residual_mhd_f32: halo_x / out need views that share a contiguous axis with the fields "
                               "and star-shaped operator kernels")
        return _dispatch.from_device(out, origin) if ok else None

    def residual_continuity(self, vars, boundary=False, absolute=False, halo_x=False, out=None):
        D_t, D_x, D_y = self.D_t, self.D_x, self.D_y
        fields = (vars[:, 0], vars[:, 1], vars[:, 2])

        def composed(rho, u, v):
            return D_t(rho) + u*D_x(rho) + rho*D_x(u) + v*D_y(rho) + rho*D_y(v)
        res = _attach(self._fused('continuity', vars, absolute, halo_x, out), fields, composed, absolute)
        if res is None:
            return _finish(_on_device(fields, composed), boundary, _CROP3, absolute, False)
        return _finish(res, boundary, _CROP3, absolute, True)

    def residual_momentum(self, vars, boundary=False, absolute=False, halo_x=False, out=None):
        D_t, D_x, D_y = self.D_t, self.D_x, self.D_y
        fields = tuple(vars[:, i] for i in range(6))

        def composed(rho, u, v, p, Bx, By):
            res_x = D_t(u) + u*D_x(u) + (1/rho)*D_x(p) - 2*(Bx/rho)*D_x(Bx) + v*D_y(u) - (By/rho)*D_y(Bx) - (Bx/rho)*D_y(By)
            res_y = D_t(v) + u*D_x(v) + (1/rho)*D_y(p) - 2*(By/rho)*D_y(By) + v*D_y(v) - (By/rho)*D_x(Bx) - (Bx/rho)*D_x(By)
            return res_x + res_y
        res = _attach(self._fused('momentum', vars, absolute, halo_x, out), fields, composed, absolute)
        if res is None:
            return _finish(_on_device(fields, composed), boundary, _CROP3, absolute, False)
        return _finish(res, boundary, _CROP3, absolute, True)

    def residual_energy(self, vars, boundary=False, absolute=False, halo_x=False, out=None):
        D_t, D_x, D_y, gamma = self.D_t, self.D_x, self.D_y, self.gamma
        fields = tuple(vars[:, i] for i in range(6))

        def composed(rho, u, v, p, Bx, By):
            p_gas = p - 0.5*(Bx**2 + By**2)
            return (D_t(rho) + u*D_x(p) + v*D_y(p) + (gamma-2)*(u*Bx+v*By)*(D_x(Bx) + D_y(By))
                    + (gamma*p_gas+By**2)*D_x(u) + (gamma*p_gas+Bx**2)*D_y(v) - Bx*By*(D_y(u) + D_x(v)))
        res = _attach(self._fused('energy', vars, absolute, halo_x, out), fields, composed, absolute)
        if res is None:
            return _finish(_on_device(fields, composed), boundary, _CROP3, absolute, False)
        return _finish(res, boundary, _CROP3, absolute, True)

    def residual_induction(self, vars, boundary=False, absolute=False, halo_x=False, out=None):
        D_t, D_x, D_y = self.D_t, self.D_x, self.D_y
        fields = (vars[:, 1], vars[:, 2], vars[:, 4], vars[:, 5])

        def composed(u, v, Bx, By):
            res_x = D_t(Bx) - By*D_y(u) + Bx*D_y(v) - v*D_y(Bx) + u*D_y(By)
            res_y = D_t(By) + By*D_x(u) - Bx*D_x(v) - v*D_x(Bx) + u*D_x(By)
            return res_x + res_y
        res = _attach(self._fused('induction', vars, absolute, halo_x, out), fields, composed, absolute)
        if res is None:
            return _finish(_on_device(fields, composed), boundary, _CROP3, absolute, False)
        return _finish(res, boundary, _CROP3, absolute, True)

    def residual_gauss(self, vars, boundary=False, absolute=False):
        Bx, By = vars[:, 4], vars[:, 5]
        res = None
        if self._want_fused(vars):
            from .vector_convops import linear2
            res = linear2(Bx, self.D_x.kernel, By, self.D_y.kernel, 1.0, _lib.PRE_FLAG_ABS if absolute else 0)
        done_abs = res is not None and absolute
        if res is None:
            res = _on_device((Bx, By), lambda Bx, By: self.D_x(Bx) + self.D_y(By))
        return _finish(res, boundary, _CROP3, absolute, done_abs)


# ======================================================================= reduced MHD (JOREK)
class JOREK:
    """``Marginal/JOREK_residuals_CP.py:188-243`` (twin ``Joint/JOREK_residuals_CP.py``): continuity and temperature
    residuals of the reduced-MHD fields (rho, phi, T) on an (R, Z) grid.

    ``vars`` is what the script holds, [BS, F, Nx, Ny, Nt] (the surrogate's layout); like its ``unstack_fields``
    (:84-95) every field is taken as the [BS, Nt, Nx, Ny] permuted view, zero-copy.  ``R`` is the 1-D radius grid
    tensor; the script's expressions broadcast it along the LAST axis of those views (its grid is square), and so
    does this class - ``len(R)`` must equal Ny.  The five operators are built as the script builds them (:201-205):
    ``D_t`` with ``scale=alpha``, ``D_R`` / ``D_Z`` with ``scale=beta`` and ``D_RR`` / ``D_ZZ`` with the value the
    name ``gamma`` holds at that point of the script - already the adiabatic index (:199), hence the default
    ``lap_scale=None`` -> ``gamma``.  ``domain='y'`` carries the reference's kernel (taps along Nt).  All five
    ``.kernel`` tensors may be replaced by the caller; the fused route hands the current ones to the library."""

    def __init__(self, R, D=3.4, K=2.25 * 1e-7, gamma=5 / 3, alpha=1, beta=1, lap_scale=None, dx=None, dy=None, dt=None,
                 device='cpu', fused=True, y_axis_fix=False):
        self.fused = fused
        self.R = torch.as_tensor(R, dtype=torch.float32)
        self.D, self.K, self.gamma = D, K, gamma
        self.dx, self.dy, self.dt = dx, dy, dt
        lap = torch.tensor(gamma, dtype=torch.float32) if lap_scale is None else lap_scale
        self.D_t = ConvOperator2D(domain='t', order=1, scale=alpha, device=device)
        self.D_R = ConvOperator2D(domain='x', order=1, scale=beta, device=device)
        self.D_Z = ConvOperator2D(domain='y', order=1, scale=beta, device=device, y_axis_fix=y_axis_fix)
        self.D_RR = ConvOperator2D(domain='x', order=2, scale=lap, device=device)
        self.D_ZZ = ConvOperator2D(domain='y', order=2, scale=lap, device=device, y_axis_fix=y_axis_fix)

    @staticmethod
    def unstack_fields(vars):
        """The script's ``unstack_fields(vars, axis=1, ...)``: [BS,F,Nx,Ny,Nt] -> F views [BS,Nt,Nx,Ny]."""
        return [vars[:, i].permute(0, 3, 1, 2) for i in range(vars.shape[1])]

    def _ops(self):
        return (self.D_t, self.D_R, self.D_Z, self.D_RR, self.D_ZZ)

    def _t32(self, x, like):
        return torch.as_tensor(x, dtype=torch.float32).to(like.device)

    def _fused(self, eq, fields, coef, absolute):
        """One streaming pass (``pre_residual_jorek_f32``); None -> compose."""
        if not self.fused or any(f.numel() == 0 for f in fields) or \
                _dispatch.needs_grad(*[getattr(o, "kernel", None) for o in self._ops()]):
            return None
        ks = [_dispatch.dense27(o.kernel) for o in self._ops()]
        if any(k is None for k in ks):
            return None
        with torch.no_grad():
            devs, origin = _stage(fields)
        f0 = devs[0]
        if self.R.numel() != f0.shape[3]:
            raise RuntimeError(f"R has {self.R.numel()} points, the fields' last axis {f0.shape[3]} (the reference "
                               "broadcasts its 1-D R along the last axis)")
        # R as one more field view that repeats the row: unit stride on the axis the fields are contiguous on
        Rd = self.R.to(f0.device)
        if f0.stride(3) == 1:
            Rb = Rd.view(1, 1, 1, -1).expand(f0.shape)
        elif f0.stride(1) == 1:                                           # Nt fastest (the surrogate's memory order)
            Rb = Rd.view(-1, 1).expand(-1, f0.shape[1]).contiguous().as_strided(tuple(f0.shape), (0, 1, 0, f0.shape[1]))
        else:
            return None
        out = _lib.empty_like_layout(f0, score_rows=absolute and origin is None)
        arr = (_lib.PreField * 3)(*[_lib.field(devs[i] if i < len(devs) else devs[0]) for i in range(3)])
        fo, fr = _lib.field(out), _lib.field(Rb)
        with torch.cuda.device(out.device):
            ok = _fused_call("pre_residual_jorek_f32", lambda: _lib.load().pre_residual_jorek_f32(
                eq, arr, ctypes.byref(fr), ctypes.byref(fo), *ks, _lib.farr(coef), *out.shape,
                _lib.PRE_FLAG_ABS if absolute else 0, _lib.stream()))
        return _dispatch.from_device(out, origin) if ok else None

    def residual_continuity(self, vars, boundary=False, norms=False, absolute=False):
        """:207-221."""
        D_t, D_R, D_Z, D_RR, D_ZZ = self._ops()
        rho, phi, _ = self.unstack_fields(vars)
        if norms and (self.dx is None or self.dy is None or self.dt is None):
            raise ValueError("norms=True needs dx, dy, dt")

        def composed(rho, phi):
            R, D = self._t32(self.R, rho), self._t32(self.D, rho)
            if norms:
                dx, dy, dt = (self._t32(v, rho) for v in (self.dx, self.dy, self.dt))
                return 2*dx*dy*D_t(rho) - (dt)*R*(D_R(rho)*D_Z(phi) - D_R(phi)*D_Z(rho)) - (2*dt*dy)*2*rho*D_Z(phi) \
                    - (4*dt)*D*(D_RR(rho) + (1/R)*D_R(rho) + D_ZZ(rho))
            return D_t(rho) - R*(D_R(rho)*D_Z(phi) - D_R(phi)*D_Z(rho)) - 2*rho*D_Z(phi) \
                - D*(D_RR(rho) + (1/R)*D_R(rho) + D_ZZ(rho))
        if norms:      # the script's scalars, folded in fp32 in its own order of operations
            t = lambda v: torch.tensor(v, dtype=torch.float32)
            dx, dy, dt, D = t(self.dx), t(self.dy), t(self.dt), t(self.D)
            coef = [float(2*dx*dy), float(dt), float((2*dt*dy)*2), float((4*dt)*D)]
        else:
            coef = [1.0, 1.0, 2.0, float(torch.tensor(self.D, dtype=torch.float32))]
        res = _attach(self._fused(0, (rho, phi), coef, absolute), (rho, phi), composed, absolute)
        if res is None:
            return _finish(_on_device((rho, phi), composed), boundary, _CROP3, absolute, False)
        return _finish(res, boundary, _CROP3, absolute, True)

    def residual_temperature(self, vars, boundary=False, norms=False, absolute=False):
        """:224-243."""
        if norms:
            raise Exception("Norm not implemented yet")              # (as the reference, :230)
        D_t, D_R, D_Z, D_RR, D_ZZ = self._ops()
        rho, phi, T = self.unstack_fields(vars)

        def composed(rho, phi, T):
            R, K, gamma = self._t32(self.R, rho), self._t32(self.K, rho), self._t32(self.gamma, rho)
            return T*D_t(rho) + rho*D_t(T) - rho*R*(D_R(T)*D_Z(phi) - D_R(phi)*D_Z(T)) + \
                T*R*(D_R(rho)*D_Z(phi) - D_R(phi)*D_Z(rho)) + \
                2*gamma*rho*T*D_Z(phi) + \
                K * (D_RR(T) + (1/R)*D_R(T) + D_ZZ(T))
        t = lambda v: torch.tensor(v, dtype=torch.float32)
        coef = [float(2 * t(self.gamma)), 0.0, 0.0, float(t(self.K))]
        res = _attach(self._fused(1, (rho, phi, T), coef, absolute), (rho, phi, T), composed, absolute)
        if res is None:
            return _finish(_on_device((rho, phi, T), composed), boundary, _CROP3, absolute, False)
        return _finish(res, boundary, _CROP3, absolute, True)


class PRE_MHD(MHD):
    """``Other_UQ/Evaluation/PRE_estimations.py:54-80``: the energy equation."""

    def __init__(self, dt, dx, dy, **kw):
        super().__init__(gamma=5 / 3, **kw)
        self.dt, self.dx, self.dy = dt, dx, dy

    def residual(self, vars, boundary=False):
        return self.residual_energy(vars, boundary)


# ======================================================================= linear: wave / advection
class PRE_Wave:
    """``Other_UQ/Evaluation/PRE_estimations.py:5-21`` / ``Marginal/Wave_Residuals_CP.py:170-184``:
    ONE additive kernel ``D_tt - (c dt/dx)^2 D_xx_yy`` applied in one pass."""

    def __init__(self, dt, dx, c=1.0, device='cpu'):
        D_tt = ConvOperator2D('t', 2, device=device)
        D_xx_yy = ConvOperator2D(('x', 'y'), 2, device=device)
        self.D = ConvOperator2D()
        c = torch.tensor(c, dtype=torch.float32)
        self.D.kernel = D_tt.kernel - ((c * dt / dx) ** 2).to(device) * D_xx_yy.kernel

    def residual(self, uu, boundary=False, absolute=False, halo_x=False, out=None):
        """``out``: optional fp32 device tensor of the field's shape for the uncropped residual (any batch stride over dense
        [Nt,Nx,Ny] blocks: ``pipeline.row_padded`` - the per-cell select that follows is 4 % faster on rows that are not
        a power of two apart).  ``halo_x``: ``uu`` is an x-slab ``full[..., x0:x1, :]`` whose rows x0 - 1 and x1 lie in the same device memory and
        are read instead of the zero padding (``PRE_FLAG_HALO_X``, see ``NavierStokes.residual_momentum``)."""
        uu = uu[:, 0] if uu.dim() == 5 else uu
        flags = (_lib.PRE_FLAG_ABS if absolute else 0) | (_lib.PRE_FLAG_HALO_X if halo_x else 0)
        if halo_x and not (uu.is_cuda and uu.stride(-1) == 1 and not _dispatch.needs_grad(uu, self.D.kernel)):
            raise ValueError("halo_x needs a device-resident, Ny-contiguous view of a larger grid and no autograd")
        if out is not None:
            if _dispatch.needs_grad(uu, self.D.kernel) or not uu.is_cuda:
                raise ValueError("out needs a device-resident field and no autograd")
            res = _dispatch._xcorr_impl(uu, self.D.kernel, 3, flags, out=out)
        else:
            res = _dispatch.xcorr(uu, self.D.kernel, nd=3, flags=flags)  # (raises if no kernel reads the halo rows)
        return res if boundary else res[_CROP3]


class Advection:
    """``Marginal/Advection_Residuals_CP.py:156-164,234-235``: ``D_t + (v disc dt/dx) D_x`` on [BS,Nt,Nx]."""

    def __init__(self, v, dt, dx, disc=2, device='cpu'):
        D_t = ConvOperator1D(domain='t', order=1, device=device)
        D_x = ConvOperator1D(domain='x', order=1, device=device)
        self.D = ConvOperator1D()
        self.D.kernel = D_t.kernel + (v * disc * dt / dx) * D_x.kernel

    def residual(self, uu, boundary=False, absolute=False):
        res = _dispatch.xcorr(uu, self.D.kernel, nd=2, flags=_lib.PRE_FLAG_ABS if absolute else 0)
        return res if boundary else res[_CROP2]


# ======================================================================= Burgers (1-D)
class Burgers:
    """``Joint/Burgers_Residuals_CP.py:171-187``:
    ``dx*D_t(u) + dt*u*D_x(u) - nu*D_xx(u)*(2*dt/dx)`` on [BS,Nt,Nx]."""

    def __init__(self, dx, dt, nu, device='cpu', fused=True):
        self.fused = fused
        self.D_t = ConvOperator1D(domain='t', order=1, device=device)
        self.D_x = ConvOperator1D(domain='x', order=1, device=device)
        self.D_xx = ConvOperator1D(domain='x', order=2, device=device)
        # the script turns the three coefficients into fp32 0-d tensors first
        self.dx, self.dt, self.nu = (torch.tensor(v, dtype=torch.float32) for v in (dx, dt, nu))

    def residual(self, uu, boundary=False, absolute=False):
        dx, dt, nu = self.dx, self.dt, self.nu
        def composed(uu):
            dxd, dtd, nud = (c.to(uu.device) for c in (dx, dt, nu))
            return dxd * self.D_t(uu) + dtd * uu * self.D_x(uu) - nud * self.D_xx(uu) * (2 * dtd / dxd)
        fused = self.fused and uu.numel() > 0 and not _dispatch.needs_grad(self.D_t.kernel, self.D_x.kernel, self.D_xx.kernel)
        ks = [_dispatch.dense9(o.kernel) for o in (self.D_t, self.D_x, self.D_xx)] if fused else [None]
        if all(k is not None for k in ks) and uu.dim() == 3:
            with torch.no_grad():
                (du,), origin = _stage((uu,))
            out = _lib.empty_like_layout(du, score_rows=absolute and origin is None)
            c3 = float(2 * dt / dx)                       # evaluated in fp32 like the reference
            with torch.cuda.device(du.device):
                ok = _fused_call("pre_residual_burgers_f32", lambda: _lib.load().pre_residual_burgers_f32(
                    _lib.ptr(du), _lib.iarr64(du.stride()), _lib.ptr(out), _lib.iarr64(out.stride()), *ks,
                    float(dx), float(dt), float(nu), c3,
                    *du.shape, _lib.PRE_FLAG_ABS if absolute else 0, _lib.stream()))
            if ok:
                res = _attach(_dispatch.from_device(out, origin), (uu,), composed, absolute)
                return res if boundary else res[_CROP2]
        return _finish(_on_device((uu,), composed), boundary, _CROP2, absolute, False)

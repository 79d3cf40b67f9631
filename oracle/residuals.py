"""Oracle (TEST INFRASTRUCTURE): PDE residual algebra of the reference scripts.

Every function evaluates the cited reference expression operator by operator with
``oracle.convops`` (i.e. the reference's ``ConvOperator`` arithmetic) and torch-CPU
elementwise ops in the reference's own evaluation order.  Operators are built
exactly as the scripts build them - in particular ``D_y = ConvOperator('y', 1)``
carries the reference's kernel, whose taps lie along Nt.

``vars`` is ``[BS, F, Nt, Nx, Ny]`` (``[BS, Nt, Nx]`` field for the 1-D cases);
``boundary=False`` crops one cell on every side like the reference.
Pinned by ``tests/golden/residuals.npz`` (PRE_Wave / PRE_NS / PRE_MHD imported
from ``Other_UQ/Evaluation/PRE_estimations.py``; the rest are restated here from
the cited lines because the scripts cannot be imported).
"""
from __future__ import annotations

import torch

from .convops import ConvOperator1D, ConvOperator2D


def _crop(res, boundary, nd=3):
    if boundary:
        return res
    return res[(Ellipsis,) + (slice(1, -1),) * nd]


class Ops2D:
    """The five operators every 2-D script builds (Marginal/NS_Residuals_CP.py:213-219)."""

    def __init__(self):
        self.D_t = ConvOperator2D("t", 1)
        self.D_x = ConvOperator2D("x", 1)
        self.D_y = ConvOperator2D("y", 1)
        self.D_xx_yy = ConvOperator2D(("x", "y"), 2)


# ---------------------------------------------------------------- linear, one kernel
def wave_kernel(c, dt, dx):
    """Marginal/Wave_Residuals_CP.py:170-175; Other_UQ/Evaluation/PRE_estimations.py:8-12."""
    D_tt = ConvOperator2D("t", 2)
    D_xx_yy = ConvOperator2D(("x", "y"), 2)
    c = torch.tensor(c, dtype=torch.float32)
    return D_tt.kernel - (c * dt / dx) ** 2 * D_xx_yy.kernel


def wave_residual(uu, c, dt, dx, boundary=False):
    """Marginal/Wave_Residuals_CP.py:179-184 - uu is [BS,Nt,Nx,Ny]."""
    D = ConvOperator2D()
    D.kernel = wave_kernel(c, dt, dx)
    return _crop(D(uu), boundary)


def advection_kernel(v, disc, dt, dx):
    """Marginal/Advection_Residuals_CP.py:156-164."""
    D_t = ConvOperator1D("t", 1)
    D_x = ConvOperator1D("x", 1)
    return D_t.kernel + (v * disc * dt / dx) * D_x.kernel


def advection_residual(uu, v, disc, dt, dx, boundary=False):
    """Marginal/Advection_Residuals_CP.py:234-235 - uu is [BS,Nt,Nx]."""
    D = ConvOperator1D()
    D.kernel = advection_kernel(v, disc, dt, dx)
    return _crop(D(uu), boundary, nd=2)


# ---------------------------------------------------------------- Burgers (1-D)
def burgers_residual(uu, dx, dt, nu, boundary=False):
    """Joint/Burgers_Residuals_CP.py:171-187 (dx, dt, nu are float32 0-d tensors there)."""
    D_t = ConvOperator1D("t", 1)
    D_x = ConvOperator1D("x", 1)
    D_xx = ConvOperator1D("x", 2)
    dx = torch.tensor(dx, dtype=torch.float32)
    dt = torch.tensor(dt, dtype=torch.float32)
    nu = torch.tensor(nu, dtype=torch.float32)
    res = dx * D_t(uu) + dt * uu * D_x(uu) - nu * D_xx(uu) * (2 * dt / dx)
    return _crop(res, boundary, nd=2)


# ---------------------------------------------------------------- Navier-Stokes
def ns_continuity(vars, dx, dy, boundary=False):
    """Marginal/NS_Residuals_CP.py:222-228."""
    o = Ops2D()
    u, v = vars[:, 0], vars[:, 1]
    res = o.D_x(u) + (dx / dy) * o.D_y(v)
    return _crop(res, boundary)


def ns_momentum(vars, dt, dx, dy, nu=0.001, boundary=False):
    """Marginal/NS_Residuals_CP.py:231-240; Other_UQ/Evaluation/PRE_estimations.py:40-50."""
    o = Ops2D()
    D_t, D_x, D_y, D_xx_yy = o.D_t, o.D_x, o.D_y, o.D_xx_yy
    u, v, p = vars[:, 0], vars[:, 1], vars[:, 2]
    res_x = D_t(u)*dx*dy + u*D_x(u)*dt*dy + v*D_y(u)*dt*dx - nu*D_xx_yy(u)*dt + D_x(p)*dt*dy
    res_y = D_t(v)*dx*dy + u*D_x(v)*dt*dx + v*D_y(v)*dt*dy - nu*D_xx_yy(v)*dt + D_y(p)*dt*dx
    if boundary:
        return res_x + res_y
    return _crop(res_x, False) + _crop(res_y, False)


# ---------------------------------------------------------------- MHD
def mhd_continuity(vars, boundary=False):
    """Marginal/MHD_Residuals_CP.py:225-231."""
    o = Ops2D()
    D_t, D_x, D_y = o.D_t, o.D_x, o.D_y
    rho, u, v = vars[:, 0], vars[:, 1], vars[:, 2]
    res = D_t(rho) + u*D_x(rho) + rho*D_x(u) + v*D_y(rho) + rho*D_y(v)
    return _crop(res, boundary)


def mhd_momentum(vars, boundary=False):
    """Marginal/MHD_Residuals_CP.py:234-243."""
    o = Ops2D()
    D_t, D_x, D_y = o.D_t, o.D_x, o.D_y
    rho, u, v, p, Bx, By = (vars[:, i] for i in range(6))
    res_x = D_t(u) + u*D_x(u) + (1/rho)*D_x(p) - 2*(Bx/rho)*D_x(Bx) + v*D_y(u) - (By/rho)*D_y(Bx) - (Bx/rho)*D_y(By)
    res_y = D_t(v) + u*D_x(v) + (1/rho)*D_y(p) - 2*(By/rho)*D_y(By) + v*D_y(v) - (By/rho)*D_x(Bx) - (Bx/rho)*D_x(By)
    if boundary:
        return res_x + res_y
    return _crop(res_x, False) + _crop(res_y, False)


def mhd_energy(vars, gamma=5 / 3, boundary=False):
    """Marginal/MHD_Residuals_CP.py:247-256; Other_UQ/Evaluation/PRE_estimations.py:70-80."""
    o = Ops2D()
    D_t, D_x, D_y = o.D_t, o.D_x, o.D_y
    rho, u, v, p, Bx, By = (vars[:, i] for i in range(6))
    p_gas = p - 0.5*(Bx**2 + By**2)
    res = D_t(rho) + u*D_x(p) + v*D_y(p) + (gamma-2)*(u*Bx+v*By)*(D_x(Bx) + D_y(By)) + (gamma*p_gas+By**2)*D_x(u) + (gamma*p_gas+Bx**2)*D_y(v) - Bx*By*(D_y(u) + D_x(v))
    return _crop(res, boundary)


def mhd_induction(vars, boundary=False):
    """Marginal/MHD_Residuals_CP.py:259-268."""
    o = Ops2D()
    D_t, D_x, D_y = o.D_t, o.D_x, o.D_y
    u, v, Bx, By = vars[:, 1], vars[:, 2], vars[:, 4], vars[:, 5]
    res_x = D_t(Bx) - By*D_y(u) + Bx*D_y(v) - v*D_y(Bx) + u*D_y(By)
    res_y = D_t(By) + By*D_x(u) - Bx*D_x(v) - v*D_x(Bx) + u*D_x(By)
    if boundary:
        return res_x + res_y
    return _crop(res_x, False) + _crop(res_y, False)


def mhd_gauss(vars, boundary=False):
    """Marginal/MHD_Residuals_CP.py:272-278."""
    o = Ops2D()
    Bx, By = vars[:, 4], vars[:, 5]
    res = o.D_x(Bx) + o.D_y(By)
    return _crop(res, boundary)


class OpsJorek:
    """The five operators of ``Marginal/JOREK_residuals_CP.py:201-205`` with the scales the script's globals hold at
    that point: alpha = beta = 1 (:192), and ``gamma`` already re-bound to the adiabatic index 5/3 (:199) - the
    ``scale`` of D_RR and D_ZZ.  ``domain='y'`` carries the reference's kernel (taps along Nt)."""

    def __init__(self, alpha=1, beta=1, lap_scale=None, gamma=5 / 3):
        lap = torch.tensor(gamma, dtype=torch.float32) if lap_scale is None else lap_scale
        self.D_t = ConvOperator2D("t", 1, scale=alpha)
        self.D_R = ConvOperator2D("x", 1, scale=beta)
        self.D_Z = ConvOperator2D("y", 1, scale=beta)
        self.D_RR = ConvOperator2D("x", 2, scale=lap)
        self.D_ZZ = ConvOperator2D("y", 2, scale=lap)


def jorek_unstack(vars):
    """``unstack_fields`` of the script (:84-95): vars [BS,F,Nx,Ny,Nt] -> F views [BS,Nt,Nx,Ny]."""
    return [vars[:, i].permute(0, 3, 1, 2) for i in range(vars.shape[1])]


def jorek_continuity(vars, R, D=3.4, boundary=False, norms=False, dx=None, dy=None, dt=None, ops=None):
    """Marginal/JOREK_residuals_CP.py:207-221 (twin: Joint/JOREK_residuals_CP.py:210-221).  ``R`` is the 1-D radius
    grid; the expression broadcasts it along the LAST axis of the [BS,Nt,Nx,Ny] fields."""
    o = ops or OpsJorek()
    D_t, D_R, D_Z, D_RR, D_ZZ = o.D_t, o.D_R, o.D_Z, o.D_RR, o.D_ZZ
    rho, phi, T = jorek_unstack(vars)
    D = torch.tensor(D, dtype=torch.float32)
    if norms:
        res = 2*dx*dy*D_t(rho) - (dt)*R*(D_R(rho)*D_Z(phi) - D_R(phi)*D_Z(rho)) - (2*dt*dy)*2*rho*D_Z(phi) - (4*dt)*D*(D_RR(rho) + (1/R)*D_R(rho) + D_ZZ(rho))
    else:
        res = D_t(rho) - R*(D_R(rho)*D_Z(phi) - D_R(phi)*D_Z(rho)) - 2*rho*D_Z(phi) - D*(D_RR(rho) + (1/R)*D_R(rho) + D_ZZ(rho))
    return _crop(res, boundary)


def jorek_temperature(vars, R, K=2.25 * 1e-7, gamma=5 / 3, boundary=False, ops=None):
    """Marginal/JOREK_residuals_CP.py:224-243."""
    o = ops or OpsJorek()
    D_t, D_R, D_Z, D_RR, D_ZZ = o.D_t, o.D_R, o.D_Z, o.D_RR, o.D_ZZ
    rho, phi, T = jorek_unstack(vars)
    K = torch.tensor(K, dtype=torch.float32)
    gamma = torch.tensor(gamma, dtype=torch.float32)
    res = T*D_t(rho) + rho*D_t(T) - rho*R*(D_R(T)*D_Z(phi) - D_R(phi)*D_Z(T)) + \
        T*R*(D_R(rho)*D_Z(phi) - D_R(phi)*D_Z(rho)) + \
        2*gamma*rho*T*D_Z(phi) + \
        K * (D_RR(T) + (1/R)*D_R(T) + D_ZZ(T))
    return _crop(res, boundary)


def periodic_bc_residual(u, dx, wall="right"):
    """Marginal/NS_Residuals_CP.py:468-478."""
    if wall == "top":
        res = u[..., 0, :] - u[..., -1, :]
    if wall == "bottom":
        res = u[..., -1, :] - u[..., 0, :]
    if wall == "left":
        res = u[..., :, 0] - u[..., :, -1]
    if wall == "right":
        res = u[..., :, -1] - u[..., :, 0]
    return res * dx

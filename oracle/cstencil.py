"""Oracle (TEST INFRASTRUCTURE): ctypes binding of ``oracle/liboracle.so``
(``oracle/stencil_ref.c``, built by ``oracle/Makefile``)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "stencil_ref.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _SO


def _lib():
    lib = ctypes.CDLL(build())
    i64, fp = ctypes.c_int64, ctypes.POINTER(ctypes.c_float)
    lib.oracle_xcorr3d_f32.argtypes = [fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, i64, i64, i64, i64]
    lib.oracle_xcorr3d_f32.restype = ctypes.c_int
    return lib


def xcorr_c(field, kernel):
    """Zero-padded cross-correlation of [B,n0,n1,n2] (3-D kernel) or [B,n0,n1] (2-D kernel)."""
    x = np.ascontiguousarray(field, np.float32)
    k = np.ascontiguousarray(kernel, np.float32)
    if k.ndim == 2:
        x4, k3 = x[:, None], k[None]
    else:
        x4, k3 = x, k
    out = np.empty_like(x4)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = _lib().oracle_xcorr3d_f32(x4.ctypes.data_as(fp), out.ctypes.data_as(fp), k3.ctypes.data_as(fp),
                                   *k3.shape, *x4.shape)
    if rc != 0:
        raise ValueError(f"oracle_xcorr3d_f32 rc={rc}")
    return out.reshape(x.shape)

"""ctypes binding of ``libcp_pre_hip.so`` (C ABI: ``include/cp_pre_hip.h``).

The library is the product; this module only marshals device pointers, sizes and the
current HIP stream.  There is no CPU fallback: if the shared object is missing, or no
MI355X is visible when a compute entry point is called, the call raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_double, c_float, c_int, c_int32, c_int64, c_void_p

import torch  # imported first on purpose: the .so must bind to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libcp_pre_hip.so")

PRE_ABI_VERSION = 8            # include/cp_pre_hip.h: the version SIGNATURES below was written for
PRE_OK, PRE_E_NULL, PRE_E_SHAPE, PRE_E_UNSUPPORTED, PRE_E_RANGE = 0, -1, -2, -3, -4
PRE_FLAG_ABS = 1
PRE_FLAG_INTERIOR_T = 2
PRE_FLAG_OUT_INTERIOR_T = 4
PRE_FLAG_HALO_X = 8
_ERR = {PRE_E_NULL: "null pointer / bad size", PRE_E_SHAPE: "unsupported shape",
        PRE_E_UNSUPPORTED: "operator kernels not star-shaped or layout not streamable",
        PRE_E_RANGE: "rank / crop out of range"}


class PreField(ctypes.Structure):
    """``pre_field_t`` / ``pre_out_t`` (same layout): a strided [B,T,X,Y] view (element strides)."""
    _fields_ = [("ptr", c_void_p), ("sB", c_int64), ("sT", c_int64), ("sX", c_int64), ("sY", c_int64)]


class PreBC(ctypes.Structure):
    """``pre_bc_t``: boundary mode / value for left, right (columns) and top, bottom (rows)."""
    _fields_ = [("mode", c_int * 4), ("value", c_float * 4)]


_fp, _fld = c_void_p, POINTER(PreField)
# name -> argtypes; every symbol include/cp_pre_hip.h declares
SIGNATURES = {
    "pre_abi_version": [],
    "pre_stencil3d_f32": [_fld, _fld, POINTER(c_float), POINTER(c_int32), c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p],
    "pre_stencil2d_f32": [_fp, POINTER(c_int64), _fp, POINTER(c_int64), POINTER(c_float), POINTER(c_int32), c_int, c_int64, c_int64, c_int64, c_int, c_void_p],
    "pre_stencil3d_wgrad_f32": [_fld, _fld, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64, _fp, c_void_p],
    "pre_residual_ns_momentum_f32": [_fld, _fld, _fld, _fld] + [POINTER(c_float)] * 4 + [c_float] * 4 + [c_int64] * 4 + [c_int, c_void_p],
    "pre_moments_segmax_f64": [_fp, c_int64, c_int64, c_int64, c_int64, c_int64, c_int, c_int, _fp, _fp, _fp, c_void_p],
    "pre_segmin_mod_f32": [_fp, c_int64, c_int64, c_int64, c_int, c_int, _fp, c_void_p],
    "pre_joint_score_pruned_f32": [_fp, c_int64, _fp, _fp, _fp, c_int64, c_int64, c_int64, c_int64, c_int, c_int, _fp, _fp, _fp, c_void_p],
    "pre_joint_score_flagged_f32": [_fp, _fp, _fp, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, _fp, _fp, c_void_p],
    "pre_residual_linear2_f32": [_fld, _fld, _fld, POINTER(c_float), POINTER(c_float), c_float] + [c_int64] * 4 + [c_int, c_void_p],
    "pre_residual_burgers_f32": [_fp, POINTER(c_int64), _fp, POINTER(c_int64)] + [POINTER(c_float)] * 3 + [c_float] * 4 + [c_int64] * 3 + [c_int, c_void_p],
    "pre_residual_mhd_f32": [c_int, POINTER(PreField), _fld] + [POINTER(c_float)] * 3 + [c_double] + [c_int64] * 4 + [c_int, c_void_p],
    "pre_residual_jorek_f32": [c_int, POINTER(PreField), _fld, _fld] + [POINTER(c_float)] * 6 + [c_int64] * 4 + [c_int, c_void_p],
    "pre_spatial2d_bc_f32": [_fp, POINTER(c_int64), _fp, POINTER(c_int64), POINTER(c_float), POINTER(PreBC), c_int64, c_int64, c_int64, c_int, c_void_p],
    "pre_spatial2d_linear2_bc_f32": [_fp, POINTER(c_int64), _fp, POINTER(c_int64), _fp, POINTER(c_int64), POINTER(c_float), POINTER(c_float),
                                     c_float, POINTER(PreBC), c_int64, c_int64, c_int64, c_int, c_void_p],
    "pre_edge_residual_f32": [_fld, c_int, c_float, c_int64, c_int64, c_int64, c_int64, _fp, c_void_p],
    "pre_absdiff_f32": [_fp, _fp, _fp, c_int64, c_void_p],
    "pre_std_axis0_f32": [_fp, _fp, c_int64, c_int64, c_float, _fp, c_void_p],
    "pre_moments_axis0_f64": [_fp, _fp, c_int64, c_int64, c_int64, _fp, _fp, c_void_p],
    "pre_std_from_moments_f32": [_fp, _fp, c_int64, c_int64, c_float, _fp, c_void_p],
    "pre_joint_score_f32": [_fp, _fp, _fp, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, _fp, c_void_p],
    "pre_kth_f32": [_fp, c_int64, POINTER(c_int64), c_int, _fp, c_void_p],
    "pre_kth_axis0_f32": [_fp, c_int64, c_int64, POINTER(c_int32), c_int, _fp, c_void_p],
    "pre_kth_axis0_strided_f32": [_fp, c_int64, c_int64, c_int64, POINTER(c_int32), c_int, _fp, c_void_p],
    "pre_kth_axis0_planes_f32": [_fp, c_int64, c_int64, c_int64, c_int64, c_int64, POINTER(c_int32), c_int, _fp, c_int64, c_int64, c_void_p],
    "pre_joint_score_pruned_max_segments": [],
    "pre_cov_count_f32": [_fp, _fp, _fp, c_int64, c_int64, c_int, _fp, c_void_p],
    "pre_cov_rowcount_f32": [_fp, _fp, _fp, c_int64, c_int64, c_int, c_int, _fp, c_void_p],
    "pre_cov_joint_f32": [_fp, _fp, _fp, c_int64, c_int64, c_int, _fp, c_void_p],
}

# libcp_pre_fft.so (include/cp_pre_fft.h): the spectral family; separate because it links hipFFT
FFT_SO_PATH = os.path.join(_HERE, "libcp_pre_fft.so")
PRE_FFT_CONJ, PRE_FFT_INVERT = 1, 2
_i64p = POINTER(c_int64)
FFT_SIGNATURES = {
    "pre_fft_abi_version": [],
    "pre_fft_create": [POINTER(c_void_p), c_int, _i64p, c_int64, c_int64],
    "pre_fft_destroy": [c_void_p],
    "pre_fft_work_bytes": [c_void_p, POINTER(ctypes.c_size_t)],
    "pre_spectral_apply_f32": [c_void_p, _fp, _i64p, _i64p, _i64p, POINTER(c_float), _i64p, c_int, c_float, _fp, _i64p, _i64p,
                               c_void_p, c_void_p],
}

_lib = None
_fft = None


def load_fft():
    """ctypes handle of libcp_pre_fft.so (loaded once, after torch so that it binds to the hipFFT / HIP runtime
    torch already loaded); raises loudly if absent."""
    global _fft
    if _fft is None:
        if not os.path.exists(FFT_SO_PATH):
            raise ImportError(f"{FFT_SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = ctypes.CDLL(FFT_SO_PATH)
        for name, argtypes in FFT_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = c_int
        _fft = lib
    return _fft


def load():
    """Load (once) and return the ctypes handle; raise loudly if the extension is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  cp_pre_amd has no CPU fallback.")
        lib = ctypes.CDLL(SO_PATH)
        # first: a stale .so with the same symbol names but other signatures would pass misaligned arguments into a
        # kernel launch
        lib.pre_abi_version.argtypes, lib.pre_abi_version.restype = [], c_int
        have = lib.pre_abi_version()
        if have != PRE_ABI_VERSION:
            raise ImportError(f"{SO_PATH} has ABI version {have}, this binding was written for {PRE_ABI_VERSION}: rebuild it "
                              "(`python -c 'import __graft_entry__ as g; g.build()'`)")
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)        # AttributeError here = header / library out of sync
            fn.argtypes = argtypes
            fn.restype = c_int64 if name.endswith(("_bytes", "_max_segments")) else c_int
        _lib = lib
    return _lib


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("cp_pre_amd: no HIP device visible; the residual / calibration path runs only on "
                           "an MI355X (there is deliberately no CPU fallback)")


def check(rc, what):
    if rc == PRE_OK:
        return
    if rc < 0:
        raise RuntimeError(f"{what}: {_ERR.get(rc, 'error')} (rc={rc})")
    if rc >= 1000:
        raise RuntimeError(f"{what}: hipfftResult {rc - 1000}")
    raise RuntimeError(f"{what}: hipError_t {rc}")


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def field(t):
    """PreField for a 4-D fp32 device view (any strides)."""
    assert t.dim() == 4
    s = t.stride()
    return PreField(t.data_ptr(), s[0], s[1], s[2], s[3])


# A row pitch that is a multiple of a large power of two puts every row of a per-cell select's column tile on the same HBM
# channels: the same kernel on the same bytes runs 2.5 instead of 3.4 TB/s at n = 1200-2048 rows, 2.8 / 2.5 instead of
# 3.3 / 3.0 at n = 4096 / 8192 (profiles/r05/select_scan.txt, column "pitch M"; the register-sort forms below 241 rows do
# not care, and rows up to 32 KiB long lose nothing: profiles/r04/rowpage.txt).  Score matrices THIS package allocates
# (|residual| outputs, device copies of host score arrays) therefore get rows PAD floats further apart than they are
# long; tensors the caller owns are selected where they lie.
PAD_MIN_ROWS, PAD_ROW_MULTIPLE, PAD = 241, 1 << 14, 64


def wants_row_pad(n, M):
    return n >= PAD_MIN_ROWS and M >= PAD_ROW_MULTIPLE and M % PAD_ROW_MULTIPLE == 0


def empty_like_layout(t, score_rows=False):
    """Uninitialised fp32 tensor with ``t``'s shape and ``t``'s axis order in memory (dense): a
    [BS,Nt,Nx,Ny] view of a [BS,Nx,Ny,Nt] buffer gets an output laid out the same way, so the
    streaming kernels read and write along the same contiguous axis.
    ``score_rows``: the result is a score matrix [n, *cells] (an |residual| output, about to be selected along axis 0):
    its rows are PAD floats further apart than they are long when ``wants_row_pad`` says that pays."""
    order = sorted(range(t.dim()), key=lambda d: (-t.stride(d), d))         # slowest axis first
    if score_rows and t.dim() >= 2 and order[0] == 0 and wants_row_pad(t.shape[0], t[0].numel()):
        M = t[0].numel()
        strides, acc = [0] * t.dim(), 1
        for d in reversed(order[1:]):
            strides[d] = acc
            acc *= t.shape[d]
        strides[0] = M + PAD
        buf = torch.empty(t.shape[0] * (M + PAD), dtype=torch.float32, device=t.device)
        return buf.as_strided(tuple(t.shape), tuple(strides))
    if t.is_contiguous():
        return torch.empty(t.shape, dtype=torch.float32, device=t.device)
    strides, acc = [0] * t.dim(), 1
    for d in reversed(order):
        strides[d] = acc
        acc *= t.shape[d]
    return torch.empty_strided(tuple(t.shape), tuple(strides), dtype=torch.float32, device=t.device)


def streamable(*views):
    """True if one of the last three axes has unit stride in every view (library relabels axes)."""
    return any(all(v.stride(ax) == 1 for v in views) for ax in (-1, -2, -3) if views[0].dim() >= -ax)


def farr(values):
    return (c_float * len(values))(*[float(v) for v in values])


def iarr32(values):
    return (c_int32 * len(values))(*[int(v) for v in values])


def iarr64(values):
    return (c_int64 * len(values))(*[int(v) for v in values])

"""Host-side layout logic of round 6, on CPU tensors (no device call): which score matrices get padded rows, which can be
selected where they lie, and the plane-major buffers of the sharded marginal flow in both field layouts."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cp_pre_amd import _lib, pipeline            # noqa: E402
from cp_pre_amd import inductive_cp as icp       # noqa: E402


def test_wants_row_pad_thresholds():
    assert _lib.wants_row_pad(241, 1 << 14) and _lib.wants_row_pad(8192, 1 << 22) and _lib.wants_row_pad(512, 3 << 14)
    assert not _lib.wants_row_pad(240, 1 << 20)               # the register-sort forms do not care about the pitch
    assert not _lib.wants_row_pad(1000, (1 << 14) - 64)       # short rows lose nothing
    assert not _lib.wants_row_pad(1000, (1 << 20) + 64)       # not a multiple of 64 KiB: no aliasing to avoid
    assert not _lib.wants_row_pad(1000, 1 << 13)


def test_empty_like_layout_score_rows():
    t = torch.empty(256, 8, 64, 64)                            # M = 2^15
    dense = _lib.empty_like_layout(t)
    assert dense.is_contiguous() and dense.shape == t.shape
    pad = _lib.empty_like_layout(t, score_rows=True)
    M = 8 * 64 * 64
    assert pad.shape == t.shape and pad.stride() == (M + _lib.PAD, 64 * 64, 64, 1)
    assert pad.untyped_storage().nbytes() == 256 * (M + _lib.PAD) * 4
    # the surrogate's Nt-fastest layout: [B,T,X,Y] view of a [B,X,Y,T] buffer - the result keeps that order, rows padded
    nt = torch.empty(256, 64, 64, 8).permute(0, 3, 1, 2)
    out = _lib.empty_like_layout(nt)
    assert out.stride() == nt.stride() and out.shape == nt.shape
    padn = _lib.empty_like_layout(nt, score_rows=True)
    assert padn.shape == nt.shape and padn.stride() == (M + _lib.PAD, 1, 64 * 8, 8)
    # few rows / odd row lengths / a batch axis that is not the slowest: dense
    assert _lib.empty_like_layout(torch.empty(200, 8, 64, 64), score_rows=True).is_contiguous()
    assert _lib.empty_like_layout(torch.empty(256, 8, 64, 65), score_rows=True).is_contiguous()
    tb = torch.empty(8, 256, 64, 64).permute(1, 0, 2, 3)       # time-major: the batch axis is not the slowest in memory
    assert _lib.empty_like_layout(tb, score_rows=True).stride() == tb.stride()


def test_rows_where_they_lie():
    n, T, X, Y = 300, 4, 6, 8
    M = T * X * Y
    dense = torch.arange(n * M, dtype=torch.float32).reshape(n, T, X, Y)
    v, pitch, order = icp.rows_where_they_lie(dense)
    assert v is dense and pitch is None and order is None
    padded = pipeline.row_padded(n, (T, X, Y), pad=64)
    padded.copy_(dense)
    v, pitch, order = icp.rows_where_they_lie(padded)
    assert pitch == M + 64 and order is None and v.data_ptr() == padded.data_ptr() and torch.equal(v, dense)
    # rows with a pitch AND the cell axes permuted in memory (an |residual| output in the surrogate's layout)
    buf = torch.zeros(n * (M + 64))
    nt = buf.as_strided((n, T, X, Y), (M + 64, 1, Y * T, T))
    nt.copy_(dense)
    v, pitch, order = icp.rows_where_they_lie(nt)
    assert pitch == M + 64 and order == [2, 3, 1] and v.shape == (n, X, Y, T) and v[0].is_contiguous()
    assert torch.equal(icp.uncanon(v, order, 1), dense)        # the cell permutation is undone on results the same way
    # dense permuted (no pitch): a contiguous view in memory order
    ntd = torch.empty(n, X, Y, T).permute(0, 3, 1, 2)
    v, pitch, order = icp.rows_where_they_lie(ntd)
    assert pitch is None and order == [2, 3, 1] and v.is_contiguous()
    # not selectable in place: strided cells, overlapping rows, a non-contiguous 1-D tensor
    assert icp.rows_where_they_lie(dense[..., ::2]) is None
    assert icp.rows_where_they_lie(torch.zeros(M + 3).as_strided((4, M), (1, 1))) is None
    assert icp.rows_where_they_lie(torch.zeros(10)[::2]) is None
    one = padded[:1]                                            # a single row: its stride says nothing
    v, pitch, order = icp.rows_where_they_lie(one.permute(0, 1, 3, 2))
    assert v is not None and pitch is None


def test_plane_major_buffers_of_the_sharded_marginal_flow():
    import bench
    B, T, X, Y = 6, 8, 16, 16
    for layout, P, per in (("ny", T, X * Y), ("nt", X, Y * T)):
        out, tm, planes, cells, pitch = bench.plane_major(B, T, X, Y, layout, torch.device("cpu"), pad=64)
        assert out.shape == (B, T, X, Y) and planes == P and cells == per and pitch == per + 64
        assert pipeline._is_time_major(tm) and tm.shape[0] == B and tm.shape[1] == P
        assert out.stride(3 if layout == "ny" else 1) == 1      # the kernel's contiguous axis: Ny / Nt
        # plane p of every local sample is one contiguous block of B * pitch floats: the send block of the exchange
        src = torch.arange(B * T * X * Y, dtype=torch.float32).reshape(B, T, X, Y)
        out.copy_(src)
        flat = tm.as_strided((P, B, per), (B * pitch, pitch, 1))
        want = src.permute(1, 0, 2, 3).reshape(P, B, per) if layout == "ny" else src.permute(2, 0, 3, 1).reshape(P, B, per)
        assert torch.equal(flat, want), layout
        # and the same bytes ARE the receive staging of 2 ranks' rows for half the planes: [P/2][2B][pitch]
        recv = tm.as_strided((P // 2, 2 * B, per), (2 * B * pitch, pitch, 1))
        assert torch.equal(recv[1, B + 2], flat[3, 2])
    sur = bench.surrogate_layout(3, 6, 5, 7, 8, torch.device("cpu"))
    assert sur.shape == (3, 6, 5, 7, 8) and sur.stride(2) == 1 and sur[0, 0].permute(1, 2, 0).is_contiguous()      # memory [.., Nx, Ny, Nt]

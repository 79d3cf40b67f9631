"""CPU: the 2-D spatial family - oracle and host-side mirrors against golden
vectors produced by importing Utils/ConvOps_Spatial.py, Utils/boundary_conditions.py and
Utils/VectorConvOps_Spatial.py (tests/golden/make_golden.py::gen_spatial)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from cp_pre_amd.boundary_conditions import BoundaryManager
from cp_pre_amd.convops_spatial import ConvOperator, get_stencil
from oracle import spatial as osp

DOMS = {"none": None, "xy": ("x", "y")}


@pytest.fixture(scope="module")
def g():
    return load_golden("spatial.npz")


def test_kernels_bit_for_bit_oracle_and_product(g):
    n = 0
    for key in g.files:
        if not key.startswith("k|"):
            continue
        _, dom, order, taylor, scale = key.split("|")
        dom = DOMS.get(dom, dom)
        ref = g[key]
        mine = osp.build_kernel(dom, int(order), float(scale), int(taylor))
        op = ConvOperator(dom, int(order), scale=float(scale), taylor_order=int(taylor), device="cpu")
        if ref.size == 0:
            assert mine is None and not hasattr(op, "kernel"), key
        else:
            assert np.array_equal(mine.numpy(), ref), key
            assert np.array_equal(op.kernel.detach().numpy(), ref), key
            assert op.kernel.requires_grad and op.scale.requires_grad          # scale is a grad-requiring leaf (:103)
            n += 1
    assert n >= 20
    # the spatial quirks: half-scaled first derivative, and 'y' == 'x'
    assert get_stencil(1, 1)[:, 1].tolist() == [-0.5, 0.0, 0.5]
    assert np.array_equal(g["k|y|1|2|1.0"], g["k|x|1|2|1.0"])
    assert not hasattr(ConvOperator("t", 1, device="cpu"), "kernel")
    with pytest.raises(ValueError, match="Unknown Convolution Method"):
        ConvOperator("x", 1, conv="nope", device="cpu")


def test_oracle_valid_conv_and_padding(g):
    x = torch.from_numpy(g["x"])
    for key in g.files:
        if key.startswith("conv|"):
            k = torch.from_numpy(g["k|" + key[5:]])
            assert rel_err(osp.conv_valid(x, k).numpy(), g[key]) <= 1e-6, key
        if key.startswith("pad|") and "mixed" not in key:
            _, bc, ks = key.split("|")
            types, values = osp._all(bc, 0.75)
            assert np.array_equal(osp.pad_signal(x, (int(ks), int(ks)), types, values).numpy(), g[key]), key
            m = BoundaryManager(kernel_size=(int(ks), int(ks)))
            m.set_all_boundaries(bc, value=0.75)
            assert np.array_equal(m.pad_signal(x).numpy(), g[key]), key
    m = BoundaryManager(kernel_size=3)
    m.set_boundary_type("left", "dirichlet", 1.5); m.set_boundary_type("right", "neumann")
    m.set_boundary_type("top", "symmetric"); m.set_boundary_type("bottom", "periodic")
    assert np.array_equal(m.pad_signal(x).numpy(), g["pad|mixed|3"])
    assert m.pad_signal(x[0, 0]).shape == (14, 18)                    # 2-D signals keep their rank
    with pytest.raises(ValueError):
        m.set_boundary_type("front", "periodic")
    with pytest.raises(ValueError):
        m.set_boundary_type("left", "absorbing")


def test_oracle_vector_ops(g):
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    for bc in ("periodic", "dirichlet", "neumann", "symmetric"):
        for ty in (2, 4):
            got = osp.VectorOp("laplace", 1.7, ty, bc)(x)
            assert rel_err(got.numpy(), g[f"laplace|{bc}|{ty}"]) <= 1e-6, (bc, ty)
        assert rel_err(osp.VectorOp("laplace", 0.5, 2, bc, scalar=False)(x, y).numpy(), g[f"laplace_vec|{bc}"]) <= 1e-6
        assert rel_err(osp.VectorOp("divergence", 2.0, 2, bc)(x, y).numpy(), g[f"divergence|{bc}"]) <= 1e-6
        assert rel_err(osp.VectorOp("curl", 2.0, 2, bc)(x, y).numpy(), g[f"curl|{bc}"]) <= 1e-6

"""cp_pre_amd - MI355X-native physics-residual evaluation and conformal calibration.

A from-scratch gfx950 implementation of ONE hot path of gitvicky/CP-PRE: the
``ConvOperator`` finite-difference stencil evaluation of PDE residuals over batched
surrogate outputs, the nonconformity-score reductions and the conformal-quantile
calibration, behind the reference's own Python surface.  The compute lives in
``libcp_pre_hip.so`` (hand-written HIP, C ABI in ``include/cp_pre_hip.h``); this package
is the thin host side.  See DESIGN.md.

    from cp_pre_amd.convops_2d import ConvOperator          # Utils/ConvOps_2d.py
    from cp_pre_amd.convops_1d import ConvOperator          # Utils/ConvOps_1d.py
    from cp_pre_amd.vector_convops import *                 # Utils/VectorConvOps.py
    from cp_pre_amd.residuals import PRE_Wave, PRE_NS, PRE_MHD   # Other_UQ/Evaluation/PRE_estimations.py
    from cp_pre_amd.inductive_cp import *                   # Neural_PDE.UQ.inductive_cp

or, for unmodified reference scripts, put ``cp_pre_amd/compat`` on ``sys.path`` and keep
``from Utils.ConvOps_2d import ConvOperator``.
"""
from . import _lib  # noqa: F401  (imports torch first, see _lib)

__all__ = ["convops_1d", "convops_2d", "vector_convops", "residuals", "inductive_cp", "pipeline", "dist"]
__version__ = "0.1.0"

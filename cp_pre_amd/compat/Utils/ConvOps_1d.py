"""Import-path shim: ``from Utils.ConvOps_1d import ConvOperator`` (Marginal/Advection_Residuals_CP.py:156)."""
from cp_pre_amd.convops_1d import ConvOperator, get_stencil, pad_kernel  # noqa: F401

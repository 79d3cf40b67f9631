"""Import-path shim: ``from Utils.ConvOps_2d import ConvOperator`` (Marginal/NS_Residuals_CP.py:213)."""
from cp_pre_amd.convops_2d import *  # noqa: F401,F403
from cp_pre_amd.convops_2d import ConvOperator, get_stencil, kernel_3d, pad_kernel  # noqa: F401

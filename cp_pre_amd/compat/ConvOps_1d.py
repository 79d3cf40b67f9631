"""Import-path shim: ``from ConvOps_1d import ConvOperator`` (Tests/test_advection.py:268)."""
from cp_pre_amd.convops_1d import *  # noqa: F401,F403
from cp_pre_amd.convops_1d import ConvOperator, get_stencil, pad_kernel  # noqa: F401

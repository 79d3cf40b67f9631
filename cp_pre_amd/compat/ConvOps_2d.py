"""Import-path shim: ``from ConvOps_2d import ConvOperator`` (README.md:37, scripts run from inside Utils/)."""
from cp_pre_amd.convops_2d import *  # noqa: F401,F403
from cp_pre_amd.convops_2d import ConvOperator, get_stencil, kernel_3d, pad_kernel  # noqa: F401

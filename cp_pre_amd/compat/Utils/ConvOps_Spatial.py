"""Import-path shim: ``from ConvOps_Spatial import *`` (Utils/VectorConvOps_Spatial.py:13)."""
from cp_pre_amd.convops_spatial import ConvOperator, get_stencil, pad_kernel  # noqa: F401

"""Import-path shim: ``from Utils.VectorConvOps_Spatial import *`` (Active_Learning/CNS.py:4)."""
from cp_pre_amd.convops_spatial import ConvOperator, get_stencil, pad_kernel  # noqa: F401
from cp_pre_amd.boundary_conditions import BoundaryManager  # noqa: F401
from cp_pre_amd.vector_convops_spatial import dot, cross, vectorize, Gradient, Laplace, Divergence, Curl, Vector_Gradient  # noqa: F401

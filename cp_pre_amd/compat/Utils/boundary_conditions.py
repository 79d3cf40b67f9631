"""Import-path shim: ``from boundary_conditions import *`` (Utils/VectorConvOps_Spatial.py:14)."""
from cp_pre_amd.boundary_conditions import BoundaryManager  # noqa: F401

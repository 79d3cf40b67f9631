"""Import-path shim: ``from Utils.VectorConvOps import *`` (Tests/NS_vector_convops.py:154)."""
from cp_pre_amd.convops_2d import *  # noqa: F401,F403
from cp_pre_amd.vector_convops import dot, cross, vectorize, Divergence, Gradient, Curl, Laplace  # noqa: F401

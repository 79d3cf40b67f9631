"""Import-path shim: ``from PRE_estimations import *`` (Other_UQ/Evaluation/Eval.py:44) - the packaged
residual classes ``PRE_Wave``, ``PRE_NS``, ``PRE_MHD`` (Other_UQ/Evaluation/PRE_estimations.py:5-80)."""
from cp_pre_amd.residuals import PRE_MHD, PRE_NS, PRE_Wave  # noqa: F401

__all__ = ["PRE_Wave", "PRE_NS", "PRE_MHD"]

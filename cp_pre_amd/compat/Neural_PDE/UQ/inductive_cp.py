"""Import-path shim: ``from Neural_PDE.UQ.inductive_cp import *`` (Marginal/NS_Residuals_CP.py:58)."""
from cp_pre_amd.inductive_cp import calibrate, modulation_func, ncf_metric_joint, emp_cov, emp_cov_joint, filter_sims_joint, filter_sims_within_bounds  # noqa: F401

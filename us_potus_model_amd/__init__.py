"""MI355X-native HMC/NUTS sampler for The Economist's 2020 poll model (hot path only).

Layout: csrc/ (HIP kernels + the C ABI of include/potus_hmc.h), sampler.py (host mirror of the
reference's `$sample()` / `rstan::extract()` surface), dataprep.py / synthetic.py (Stan data
lists), diagnostics.py (R-hat / ESS), _abi.py (ctypes structs).
"""
from . import _abi  # noqa: F401
from .sampler import (Handle, PotusError, PotusModel, StanFit, backtest_scores, check_convergence, device_diagnostics, device_diagnostics_of_block,  # noqa: F401
                      load_library, posterior_summary, run_many, sampling)

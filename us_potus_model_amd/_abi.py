"""ctypes mirror of include/potus_hmc.h (the C ABI of libpotus_hmc.so).

The structs follow the Stan data block (scripts/model/poll_model_2020.stan:1-41) and the
cmdstanr `$sample()` argument surface (scripts/model/final_2016.R:533-541).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

VARIANT_FULL = 0     # scripts/model/poll_model_2020.stan
VARIANT_NO_MODE = 1  # scripts/model/poll_model_2020_no_mode_adjustment.stan
VARIANTS = {"full": VARIANT_FULL, "no_mode_adjustment": VARIANT_NO_MODE}
METRIC_DIAG, METRIC_DENSE = 0, 1
METRICS = {"diag_e": METRIC_DIAG, "dense_e": METRIC_DENSE}
STORAGE_F64, STORAGE_F32 = 0, 1
N_SAMPLER_COLS = 7
SAMPLER_COLS = ("lp__", "accept_stat__", "stepsize__", "treedepth__", "n_leapfrog__",
                "divergent__", "energy__")

_I32P = C.POINTER(C.c_int32)
_F64P = C.POINTER(C.c_double)

_INT_VECS = ("state", "day_state", "day_national", "poll_state", "poll_national",
             "poll_mode_state", "poll_mode_national", "poll_pop_state", "poll_pop_national",
             "n_democrat_national", "n_two_share_national", "n_democrat_state",
             "n_two_share_state")
_DBL_VECS = ("unadjusted_national", "unadjusted_state", "mu_b_prior", "state_weights")


class PotusData(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in ("N_national_polls", "N_state_polls", "T", "S", "P", "M", "Pop")]
        + [(n, _I32P) for n in _INT_VECS]
        + [(n, _F64P) for n in _DBL_VECS]
        + [(n, C.c_double) for n in ("sigma_c", "sigma_m", "sigma_pop",
                                     "sigma_measure_noise_national", "sigma_measure_noise_state",
                                     "sigma_e_bias")]
        + [("state_covariance_0", _F64P)]
        + [(n, C.c_double) for n in ("random_walk_scale", "mu_b_T_scale", "polling_bias_scale")]
        + [("variant", C.c_int32)]
    )


class PotusOpts(C.Structure):
    _fields_ = [
        ("chains", C.c_int32), ("chain_id_offset", C.c_int32),
        ("num_warmup", C.c_int32), ("num_samples", C.c_int32), ("max_depth", C.c_int32),
        ("init_buffer", C.c_int32), ("term_buffer", C.c_int32), ("window", C.c_int32),
        ("delta", C.c_double), ("gamma", C.c_double), ("kappa", C.c_double), ("t0", C.c_double),
        ("stepsize", C.c_double), ("init_radius", C.c_double),
        ("seed", C.c_uint64), ("device", C.c_int32), ("save_warmup", C.c_int32),
        ("cus_per_chain", C.c_int32), ("metric", C.c_int32), ("twin", C.c_int32), ("metric_storage", C.c_int32),
        ("pooled_metric", C.c_int32), ("reserved_", C.c_int32),
    ]


def make_data(data: dict, variant: str | int = "full"):
    """Build a PotusData from the R-style named list. Returns (struct, keepalive).

    Extra entries of `data` are ignored, as Stan ignores them (the 2008/2012 lists carry
    an unused `sigma_a`, final_2012.R:500-542); entries the no-mode variant does not
    declare may be absent.
    """
    v = VARIANTS[variant] if isinstance(variant, str) else int(variant)
    d = PotusData()
    keep = []
    for n in ("N_national_polls", "N_state_polls", "T", "S", "P"):
        setattr(d, n, int(data[n]))
    d.M = int(data.get("M", 0) or 0)
    d.Pop = int(data.get("Pop", 0) or 0)
    for n in _INT_VECS:
        if n in data and data[n] is not None:
            a = np.ascontiguousarray(np.asarray(data[n]).reshape(-1), dtype=np.int32)
            keep.append(a)
            setattr(d, n, a.ctypes.data_as(_I32P))
    for n in _DBL_VECS:
        if n in data and data[n] is not None:
            a = np.ascontiguousarray(np.asarray(data[n]).reshape(-1), dtype=np.float64)
            keep.append(a)
            setattr(d, n, a.ctypes.data_as(_F64P))
    for n in ("sigma_c", "sigma_m", "sigma_pop", "sigma_measure_noise_national",
              "sigma_measure_noise_state", "sigma_e_bias", "random_walk_scale", "mu_b_T_scale",
              "polling_bias_scale"):
        setattr(d, n, float(data.get(n, 0.0) or 0.0))
    cov = np.asarray(data["state_covariance_0"], dtype=np.float64)
    cov = np.ascontiguousarray(cov.T).reshape(-1)  # column-major
    keep.append(cov)
    d.state_covariance_0 = cov.ctypes.data_as(_F64P)
    d.variant = v
    return d, keep


def num_params(data: dict, variant: str | int = "full") -> int:
    """Unconstrained dimension D (parameters block, poll_model_2020.stan:56-69)."""
    v = VARIANTS[variant] if isinstance(variant, str) else int(variant)
    S, T = int(data["S"]), int(data["T"])
    D = 2 * S + S * T + int(data["P"]) + int(data["N_national_polls"]) + int(data["N_state_polls"])
    if v == VARIANT_FULL:
        D += int(data["M"]) + int(data["Pop"]) + 2 + T
    return D


def column_layout(data: dict, variant: str | int = "full"):
    """Ordered (name, dims) of every block of one CmdStan output row after the 7 sampler
    columns: parameters (stan:56-69), transformed parameters (stan:72-83), generated
    quantities (stan:135).  Matrices are flattened column-major, as CmdStan does."""
    v = VARIANTS[variant] if isinstance(variant, str) else int(variant)
    S, T, P = int(data["S"]), int(data["T"]), int(data["P"])
    Nn, Ns = int(data["N_national_polls"]), int(data["N_state_polls"])
    full = v == VARIANT_FULL
    M, Pop = (int(data["M"]), int(data["Pop"])) if full else (0, 0)
    blocks = [("raw_mu_b_T", (S,)), ("raw_mu_b", (S, T)), ("raw_mu_c", (P,))]
    if full:
        blocks += [("raw_mu_m", (M,)), ("raw_mu_pop", (Pop,)), ("mu_e_bias", ()), ("rho_e_bias", ()),
                   ("raw_e_bias", (T,))]
    blocks += [("raw_measure_noise_national", (Nn,)), ("raw_measure_noise_state", (Ns,)),
               ("raw_polling_bias", (S,))]
    blocks += [("mu_b", (S, T)), ("mu_c", (P,))]
    if full:
        blocks += [("mu_m", (M,)), ("mu_pop", (Pop,)), ("e_bias", (T,))]
    blocks += [("polling_bias", (S,)), ("national_mu_b_average", (T,)),
               ("national_polling_bias_average", ())]
    if full:
        blocks += [("sigma_rho", ())]
    blocks += [("logit_pi_democrat_state", (Ns,)), ("logit_pi_democrat_national", (Nn,)),
               ("predicted_score", (T, S))]
    out, col = {}, N_SAMPLER_COLS
    for name, dims in blocks:
        n = int(np.prod(dims)) if dims else 1
        out[name] = (col, col + n, dims)
        col += n
    return out, col

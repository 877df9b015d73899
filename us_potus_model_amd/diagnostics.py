"""R-hat / ESS for pooled chains (Vehtari, Gelman, Simpson, Carpenter, Buerkner 2021).

The reference never inspects sampler diagnostics; the ESS/s metric of
BASELINE.json is therefore defined here: bulk-ESS = ESS of the rank-normalised split chains,
minimum over {lp__, mu_b[:,T], predicted_score[T,:]}.  Pure numpy, runs on the pooled draws
after the all-gather.
"""
from __future__ import annotations

import numpy as np
from scipy import special


def _split(x: np.ndarray) -> np.ndarray:
    """[chains, draws] -> [2*chains, draws//2]."""
    c, n = x.shape
    h = n // 2
    return np.concatenate([x[:, :h], x[:, n - h:]], axis=0)


def _rank_normalise(x: np.ndarray) -> np.ndarray:
    r = np.argsort(np.argsort(x.reshape(-1), kind="stable"), kind="stable").reshape(x.shape) + 1.0
    return special.ndtri((r - 0.375) / (x.size + 0.25))


def _autocov(x: np.ndarray) -> np.ndarray:
    """Biased autocovariance per chain via FFT. x: [chains, n]."""
    c, n = x.shape
    m = 1 << int(np.ceil(np.log2(2 * n)))
    xc = x - x.mean(axis=1, keepdims=True)
    f = np.fft.rfft(xc, m, axis=1)
    ac = np.fft.irfft(f * np.conj(f), m, axis=1)[:, :n]
    return ac / n


def rhat_basic(x: np.ndarray) -> float:
    c, n = x.shape
    if n < 2:
        return np.nan
    w = x.var(axis=1, ddof=1).mean()
    b = n * x.mean(axis=1).var(ddof=1) if c > 1 else 0.0
    if w == 0:
        return np.nan
    return float(np.sqrt(((n - 1) / n * w + b / n) / w))


def ess_basic(x: np.ndarray) -> float:
    """Geyer initial-monotone-sequence ESS over chains. x: [chains, n]."""
    c, n = x.shape
    if n < 4:
        return np.nan
    acov = _autocov(x)
    chain_var = acov[:, 0] * n / (n - 1.0)
    mean_var = chain_var.mean()
    var_plus = mean_var * (n - 1.0) / n
    if c > 1:
        var_plus += x.mean(axis=1).var(ddof=1)
    if not np.isfinite(var_plus) or var_plus <= 0:
        return np.nan
    rho = 1.0 - (mean_var - acov.mean(axis=0)) / var_plus
    rho[0] = 1.0
    # Geyer: sums of adjacent pairs must be positive and non-increasing
    t, tau, prev = 0, 0.0, np.inf
    while t + 1 < n:
        pair = rho[t] + rho[t + 1]
        if pair < 0:
            break
        pair = min(pair, prev)
        tau += 2.0 * pair
        prev = pair
        t += 2
    tau -= 1.0
    tau = max(tau, 1.0 / np.log10(c * n))
    return float(c * n / tau)


def rhat(x: np.ndarray) -> float:
    """Rank-normalised split R-hat (max of bulk and folded)."""
    xs = _split(np.asarray(x, dtype=np.float64))
    bulk = rhat_basic(_rank_normalise(xs))
    folded = rhat_basic(_rank_normalise(np.abs(xs - np.median(xs))))
    return float(np.nanmax([bulk, folded]))


def ess_bulk(x: np.ndarray) -> float:
    return ess_basic(_rank_normalise(_split(np.asarray(x, dtype=np.float64))))


def ess_mean(x: np.ndarray) -> float:
    return ess_basic(_split(np.asarray(x, dtype=np.float64)))


def summarise(draws: np.ndarray) -> dict:
    """draws: [chains, n, k] -> per-column rhat, bulk ESS, mean, mcse."""
    c, n, k = draws.shape
    out = dict(rhat=np.zeros(k), ess_bulk=np.zeros(k), ess_mean=np.zeros(k), mean=np.zeros(k), sd=np.zeros(k))
    for j in range(k):
        x = draws[:, :, j]
        out["rhat"][j] = rhat(x)
        out["ess_bulk"][j] = ess_bulk(x)
        out["ess_mean"][j] = ess_mean(x)
        out["mean"][j] = x.mean()
        out["sd"][j] = x.std(ddof=1)
    out["mcse"] = out["sd"] / np.sqrt(out["ess_mean"])
    return out

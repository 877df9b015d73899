"""Synthetic `data` lists with the schema of poll_model_2020.stan:1-41.

Used for small parity cases, the no-mode variant and the BASELINE config-5 stress shape
(51 states x 600 days x 10 000 polls).  Polls are simulated
forward from the model itself so the posterior is well behaved.
"""
from __future__ import annotations

import numpy as np


def _corr(S: int, rng) -> np.ndarray:
    f = rng.normal(size=(S, 9))
    C = np.corrcoef(f)
    C = 0.75 * np.maximum(C, 0.0) + 0.25
    np.fill_diagonal(C, 1.0)
    w, v = np.linalg.eigh(C)
    return (v * np.maximum(w, 1e-6)) @ v.T


def make(S=51, T=254, N_state=1258, N_national=361, P=161, M=3, Pop=3, seed=20201103,
         variant="full") -> dict:
    rng = np.random.default_rng(seed)
    w = rng.dirichlet(np.full(S, 3.0))
    cov0 = 0.0049 * (0.9 * _corr(S, rng) + 0.1 * np.eye(S)) if S > 1 else np.array([[0.0049]])
    cov0 = 0.5 * (cov0 + cov0.T)
    scales = dict(polling_bias_scale=0.052, mu_b_T_scale=0.12, random_walk_scale=0.05 / np.sqrt(300.0) * 4)
    sig = dict(sigma_c=0.06, sigma_m=0.04, sigma_pop=0.04, sigma_measure_noise_national=0.04,
               sigma_measure_noise_state=0.04, sigma_e_bias=0.02)
    prior = rng.normal(0.0, 0.4, size=S)

    def days(n):
        d = T - np.floor(T * rng.beta(1.0, 2.0, size=n)).astype(int)
        return np.clip(d, 1, T)

    def pollsters(n):
        p = rng.zipf(1.3, size=n)
        return ((p - 1) % P) + 1

    state = rng.integers(1, S + 1, size=N_state)
    day_state, day_nat = days(N_state), days(N_national)
    poll_state, poll_nat = pollsters(N_state), pollsters(N_national)
    pm = np.array([0.5, 0.37, 0.13][:M]); pm = pm / pm.sum() if M <= 3 else np.full(M, 1.0 / M)
    pp = np.array([0.8, 0.196, 0.004][:Pop]); pp = pp / pp.sum() if Pop <= 3 else np.full(Pop, 1.0 / Pop)
    mode_state = rng.choice(M, size=N_state, p=pm) + 1
    mode_nat = rng.choice(M, size=N_national, p=pm) + 1
    pop_state = rng.choice(Pop, size=N_state, p=pp) + 1
    pop_nat = rng.choice(Pop, size=N_national, p=pp) + 1
    unadj_state = (rng.random(N_state) < 0.78).astype(float)
    unadj_nat = (rng.random(N_national) < 0.78).astype(float)
    n2_state = np.clip(np.rint(rng.lognormal(np.log(660.0), 0.6, size=N_state)), 100, 60000).astype(int)
    n2_nat = np.clip(np.rint(rng.lognormal(np.log(900.0), 0.6, size=N_national)), 100, 60000).astype(int)

    # forward simulation of the model (stan:70-113)
    nsd = np.sqrt(w @ cov0 @ w)
    Ls = {k: np.linalg.cholesky(cov0 * (v / nsd) ** 2) for k, v in scales.items()}
    mu_b = np.zeros((S, T))
    mu_b[:, T - 1] = Ls["mu_b_T_scale"] @ rng.normal(size=S) + prior
    for t in range(T - 2, -1, -1):
        mu_b[:, t] = Ls["random_walk_scale"] @ rng.normal(size=S) + mu_b[:, t + 1]
    pb = Ls["polling_bias_scale"] @ rng.normal(size=S)
    mu_c = rng.normal(size=P) * sig["sigma_c"]
    full = variant == "full"
    mu_m = rng.normal(size=M) * sig["sigma_m"] if full else np.zeros(M)
    mu_pop = rng.normal(size=Pop) * sig["sigma_pop"] if full else np.zeros(Pop)
    e = np.zeros(T)
    if full:
        rho, mue = 0.7, 0.0
        e[0] = rng.normal() * sig["sigma_e_bias"]
        for t in range(1, T):
            e[t] = mue + rho * (e[t - 1] - mue) + rng.normal() * np.sqrt(1 - rho ** 2) * sig["sigma_e_bias"]
    eta_s = (mu_b[state - 1, day_state - 1] + mu_c[poll_state - 1] + mu_m[mode_state - 1] + mu_pop[pop_state - 1]
             + unadj_state * e[day_state - 1] + rng.normal(size=N_state) * sig["sigma_measure_noise_state"]
             + pb[state - 1])
    eta_n = ((w @ mu_b)[day_nat - 1] + mu_c[poll_nat - 1] + mu_m[mode_nat - 1] + mu_pop[pop_nat - 1]
             + unadj_nat * e[day_nat - 1] + rng.normal(size=N_national) * sig["sigma_measure_noise_national"]
             + pb @ w)
    y_state = rng.binomial(n2_state, 1.0 / (1.0 + np.exp(-eta_s)))
    y_nat = rng.binomial(n2_nat, 1.0 / (1.0 + np.exp(-eta_n)))

    i32 = lambda a: np.asarray(a, dtype=np.int32)
    data = dict(
        N_national_polls=int(N_national), N_state_polls=int(N_state), T=int(T), S=int(S), P=int(P),
        M=int(M), Pop=int(Pop),
        state=i32(state), day_state=i32(day_state), day_national=i32(day_nat),
        poll_state=i32(poll_state), poll_national=i32(poll_nat),
        poll_mode_state=i32(mode_state), poll_mode_national=i32(mode_nat),
        poll_pop_state=i32(pop_state), poll_pop_national=i32(pop_nat),
        n_democrat_national=i32(y_nat), n_two_share_national=i32(n2_nat),
        n_democrat_state=i32(y_state), n_two_share_state=i32(n2_state),
        unadjusted_national=unadj_nat, unadjusted_state=unadj_state,
        mu_b_prior=prior, state_weights=w, state_covariance_0=cov0,
        **sig, **scales,
    )
    if not full:  # what the 2008/2012 scripts pass (final_2012.R:500-542)
        for k in ("poll_mode_state", "poll_mode_national", "poll_pop_state", "poll_pop_national",
                  "unadjusted_national", "unadjusted_state", "sigma_m", "sigma_pop", "sigma_e_bias"):
            data.pop(k)
        data["M"] = 0
        data["Pop"] = 0
        data["sigma_a"] = 1.0  # unused extra entry, must be ignored
    return data


def small(variant="full", seed=7) -> dict:
    """A case the CPU oracle samples in seconds."""
    return make(S=6, T=24, N_state=70, N_national=25, P=9, M=3, Pop=3, seed=seed, variant=variant)


def stress() -> dict:
    """BASELINE.json configs[4]: 51 x 600 days x 10 000 polls."""
    return make(S=51, T=600, N_state=8000, N_national=2000, P=300, M=3, Pop=3, seed=20201103)

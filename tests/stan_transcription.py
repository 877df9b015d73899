"""Independent literal transcription of scripts/model/poll_model_2020.stan:42-132 in torch fp64.

Written statement by statement from the Stan source (not from oracle/potus_oracle.c) so that
torch autograd provides a second opinion on both the value and the gradient the oracle
computes.  `~` statements drop constants; Jacobians follow the Stan reference manual
(offset/multiplier: log|multiplier|; lower=0,upper=1: log(inv_logit(x)) + log(1-inv_logit(x))).
"""
import numpy as np
import torch


def log_prob(data: dict, q: torch.Tensor, variant="full") -> torch.Tensor:
    f64 = torch.float64
    full = variant == "full"
    S, T, P = int(data["S"]), int(data["T"]), int(data["P"])
    Nn, Ns = int(data["N_national_polls"]), int(data["N_state_polls"])
    tt = lambda a: torch.as_tensor(np.asarray(a), dtype=f64)
    ti = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.long) - 1
    w = tt(data["state_weights"])
    cov0 = tt(data["state_covariance_0"])
    # transformed data, stan:42-55
    nsd = torch.sqrt(w @ cov0 @ w)
    chol = lambda scale: torch.linalg.cholesky(cov0 * (scale / nsd) ** 2)
    L_pb, L_T, L_W = chol(data["polling_bias_scale"]), chol(data["mu_b_T_scale"]), chol(data["random_walk_scale"])
    # parameters, stan:56-69 (declaration order, matrices column-major)
    o = 0
    def take(n):
        nonlocal o
        v = q[o:o + n]
        o += n
        return v
    raw_mu_b_T = take(S)
    raw_mu_b = take(S * T).reshape(T, S).T  # column-major S x T
    raw_mu_c = take(P)
    lp = torch.zeros((), dtype=f64)
    if full:
        M, Pop = int(data["M"]), int(data["Pop"])
        raw_mu_m, raw_mu_pop = take(M), take(Pop)
        x_mu_e, x_rho = take(1)[0], take(1)[0]
        raw_e_bias = take(T)
        mu_e_bias = 0.0 + 0.02 * x_mu_e
        lp = lp + np.log(0.02)
        rho_e_bias = torch.sigmoid(x_rho)
        lp = lp + torch.log(rho_e_bias) + torch.log1p(-rho_e_bias)
    raw_nn, raw_ns, raw_pb = take(Nn), take(Ns), take(S)
    assert o == q.numel()
    # transformed parameters, stan:70-113
    polling_bias = L_pb @ raw_pb
    nat_pb = polling_bias @ w
    cols = [None] * T
    cols[T - 1] = L_T @ raw_mu_b_T + tt(data["mu_b_prior"])
    for i in range(1, T):
        cols[T - 1 - i] = L_W @ raw_mu_b[:, T - 1 - i] + cols[T - i]
    mu_b = torch.stack(cols, dim=1)
    nat_avg = mu_b.T @ w
    mu_c = raw_mu_c * data["sigma_c"]
    state, day_s, day_n = ti(data["state"]), ti(data["day_state"]), ti(data["day_national"])
    eta_s = mu_b[state, day_s] + mu_c[ti(data["poll_state"])]
    eta_n = nat_avg[day_n] + mu_c[ti(data["poll_national"])]
    if full:
        mu_m = raw_mu_m * data["sigma_m"]
        mu_pop = raw_mu_pop * data["sigma_pop"]
        sigma_rho = torch.sqrt(1 - rho_e_bias ** 2) * data["sigma_e_bias"]
        e = [raw_e_bias[0] * data["sigma_e_bias"]]
        for t in range(1, T):
            e.append(mu_e_bias + rho_e_bias * (e[t - 1] - mu_e_bias) + raw_e_bias[t] * sigma_rho)
        e_bias = torch.stack(e)
        eta_s = eta_s + mu_m[ti(data["poll_mode_state"])] + mu_pop[ti(data["poll_pop_state"])] \
            + tt(data["unadjusted_state"]) * e_bias[day_s]
        eta_n = eta_n + mu_m[ti(data["poll_mode_national"])] + mu_pop[ti(data["poll_pop_national"])] \
            + tt(data["unadjusted_national"]) * e_bias[day_n]
    eta_s = eta_s + raw_ns * data["sigma_measure_noise_state"] + polling_bias[state]
    eta_n = eta_n + raw_nn * data["sigma_measure_noise_national"] + nat_pb
    # model, stan:115-132
    sn = lambda v: -0.5 * (v ** 2).sum()
    lp = lp + sn(raw_mu_b_T) + sn(raw_mu_b) + sn(raw_mu_c)
    if full:
        lp = lp + sn(raw_mu_m) + sn(raw_mu_pop)
        lp = lp - 0.5 * (mu_e_bias / 0.02) ** 2
        lp = lp - 0.5 * ((rho_e_bias - 0.7) / 0.1) ** 2
        lp = lp + sn(raw_e_bias)
    lp = lp + sn(raw_nn) + sn(raw_ns) + sn(raw_pb)
    ls = torch.nn.functional.logsigmoid
    ys, Nst = tt(data["n_democrat_state"]), tt(data["n_two_share_state"])
    yn, Nnt = tt(data["n_democrat_national"]), tt(data["n_two_share_national"])
    lp = lp + (ys * ls(eta_s) + (Nst - ys) * ls(-eta_s)).sum()
    lp = lp + (yn * ls(eta_n) + (Nnt - yn) * ls(-eta_n)).sum()
    aux = dict(mu_b=mu_b, eta_s=eta_s, eta_n=eta_n, polling_bias=polling_bias, nat_avg=nat_avg)
    return lp, aux


def log_prob_grad(data, q, variant="full"):
    qt = torch.tensor(np.asarray(q, dtype=np.float64), requires_grad=True)
    lp, aux = log_prob(data, qt, variant)
    lp.backward()
    return float(lp.detach()), qt.grad.numpy().copy(), {k: v.detach().numpy() for k, v in aux.items()}

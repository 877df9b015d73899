"""Stan 2.24's warm-up adaptation restated from a device chain's OWN saved rows, one transition at a time (test infrastructure).

Two fp64 chains whose gradients differ in the 13th digit drift apart after some tens of transitions, so a device chain cannot be
compared with the oracle's chain over a whole warm-up.  Here nothing can drift: every quantity is recomputed from the device's own
previous row.

  * the step size of every transition = exp(x) of stepsize_adaptation::learn_stepsize fed with the accept_stat__ of the rows
    before it (restarted after a metric update with mu = log(10 eps));
  * at the end of a window the inverse metric = var_adaptation's (diag_e) or covar_adaptation's (dense_e) regularised estimate from
    the very draws saved for that window, and the step size that follows = base_hmc::init_stepsize run by the ORACLE from the
    device's draw, that metric and the step size learn_stepsize had just proposed (same Philox momenta);
  * the window schedule is windowed_adaptation's: compute_next_window doubles the window and stretches the last one to the start of
    the terminal buffer (oracle/potus_oracle.c:1006-1065 restates the same upstream code) -- `run_through_the_windows` below stops
    the device after every window end, so the metric the device holds after EVERY update is compared, not just the last one;
  * the step size kept after the warm-up = exp(x_bar);
  * the transitions named in replay_rows are the oracle's transitions from the device's previous draw, step size and metric:
    same tree depth, leapfrog count and divergence flag, values to 1e-6.

With potus_opts.pooled_metric (a declared deviation from Stan: one dense metric for all chains of a handle) the estimate at a window end is
covar_adaptation's formula applied to the window draws of ALL chains of the handle, chain after chain -- N = chains x n draws, N/(N+5) cov +
1e-3 5/(N+5) I; everything else is per chain as before.

Covers warm-ups whose rows are all saved (save_warmup = 1).  Reference schedule: scripts/model/final_2016.R:6-11,533-541 runs 500
warm-up iterations = 75 | 25, 50, 100, 200 | 50."""
import numpy as np

from oracle_lib import OracleModel
from us_potus_model_amd import _abi


def window_schedule(nw, ib, tb, bw):
    """[(first row, last row)] of the variance windows of windowed_adaptation for nw warm-up iterations (0-based rows; the metric
    is updated after the window's last row)."""
    if nw < 20:
        return []
    if ib + bw + tb > nw:                                            # windowed_adaptation's constructor
        ib, tb = int(0.15 * nw), int(0.1 * nw)
        bw = nw - (ib + tb)
    out, start, size, nxt, last = [], ib, bw, ib + bw - 1, nw - tb - 1
    while True:
        out.append((start, nxt))
        if nxt == last:
            return out
        size *= 2                                                    # compute_next_window
        start, nxt = nxt + 1, nxt + size
        if nxt != last and nxt + 2 * size >= nw - tb:
            nxt = last


def run_through_the_windows(h, n_total):
    """Run n_total iterations, stopping after every window end: {row of the window end: (step sizes, metrics) the device holds right
    after it} -- metrics[chain] is the diagonal (diag_e) or the D x D matrix (dense_e)."""
    o = h.opts
    dense = o.metric == _abi.METRIC_DENSE
    held, done = {}, 0
    for _, end in window_schedule(o.num_warmup, o.init_buffer, o.term_buffer, o.window):
        if end + 1 > n_total:
            break
        h.run(end + 1 - done)
        done = end + 1
        eps, diag = h.adaptation()
        held[end] = (np.array(eps), [h.dense_metric(c) for c in range(o.chains)] if dense else np.array(diag))
    if n_total > done:
        h.run(n_total - done)
    return held


def adaptation_replayed_from_the_device_rows(data, variant, h, chain, seed, replay_rows, held=None):
    """See the module docstring.  held: what run_through_the_windows returned (optional: without it only the last metric update is
    compared with the device's, through potus_get_adaptation at the end)."""
    o_ = h.opts
    nw = o_.num_warmup
    dense = o_.metric == _abi.METRIC_DENSE
    d_all = h.draws()
    d = d_all[chain]
    pooled = bool(getattr(o_, "pooled_metric", 0))                   # potus_opts.pooled_metric: ONE estimate from the window draws of ALL chains of the handle
    eps_final, minv_final = h.adaptation()
    eps_final = eps_final[chain]
    minv_final = h.dense_metric(chain) if dense else np.asarray(minv_final[chain])
    D = h.D
    m = OracleModel(data, variant)
    o = m.default_opts(num_warmup=nw, num_samples=o_.num_samples, seed=seed, fast_grad=1, max_depth=o_.max_depth, dense_metric=1 if dense else 0)
    chain_id = o_.chain_id_offset + chain + 1
    windows = window_schedule(nw, o_.init_buffer, o_.term_buffer, o_.window)
    ends = {e: (s, i) for i, (s, e) in enumerate(windows)}
    delta, gamma, kappa, t0 = o_.delta, o_.gamma, o_.kappa, o_.t0
    chol_of = (lambda M_: np.linalg.cholesky(M_)) if dense else (lambda M_: None)

    def estimate(w):
        n = float(len(w))
        if dense:                                                    # covar_adaptation::learn_covariance
            return (n / (n + 5.0)) * np.cov(np.array(w).T) + 1e-3 * (5.0 / (n + 5.0)) * np.eye(D)
        return (n / (n + 5.0)) * np.var(np.array(w), axis=0, ddof=1) + 1e-3 * (5.0 / (n + 5.0))   # var_adaptation::learn_variance

    # the very first search: from the initial point (U(-2,2), Philox index = Stan index, first attempt) with the unit metric
    q_init = np.array([o_.init_radius * (2.0 * m.L.oracle_rng_uniform(seed, chain_id, 0xFFFFFFFF, 5, 0, i) - 1.0) for i in range(D)])
    minv = np.eye(D) if dense else np.ones(D)
    eps0 = m.init_stepsize_from(chain_id, o, 0xFFFFFFFF, q_init, o_.stepsize, minv, chol_of(minv))
    assert d[0, 2] == eps0, (d[0, 2], eps0)
    mu, s_bar, x_bar, cnt = np.log(10.0 * o_.stepsize), 0.0, 0.0, 0.0   # (services: set_mu(log(10 * stepsize)) precedes the first search)
    metric_at = {}                                                   # row -> metric it ran under
    next_eps = eps0
    updates = 0
    for it in range(min(nw, len(d))):
        assert abs(d[it, 2] / next_eps - 1.0) < 1e-12, (it, d[it, 2], next_eps)
        metric_at[it] = minv
        cnt += 1.0                                                   # learn_stepsize
        a = min(1.0, d[it, 1])
        eta = 1.0 / (cnt + t0)
        s_bar = (1.0 - eta) * s_bar + eta * (delta - a)
        x = mu - s_bar * np.sqrt(cnt) / gamma
        x_eta = cnt ** (-kappa)
        x_bar = (1.0 - x_eta) * x_bar + x_eta * x
        next_eps = np.exp(x)
        if it in ends:
            start, idx = ends[it]
            want = estimate(np.concatenate([dc[start:it + 1, 7:] for dc in d_all]) if pooled else d[start:it + 1, 7:])
            scale = np.abs(want).max()
            if held is not None and it in held:                      # what the device held right after this update
                got = np.asarray(held[it][1][chain])
                assert np.allclose(got, want, rtol=1e-9, atol=1e-12 * scale), (it, np.abs(got - want).max())
                minv = got
            elif idx == len(windows) - 1:                            # the last metric update: the device still holds it
                assert np.allclose(minv_final, want, rtol=1e-9, atol=1e-12 * scale), (it, np.abs(minv_final - want).max())
                minv = minv_final
            else:
                minv = want
            next_eps = m.init_stepsize_from(chain_id, o, it, d[it, 7:], next_eps, minv, chol_of(minv))
            if held is not None and it in held:
                assert abs(held[it][0][chain] / next_eps - 1.0) < 1e-12, (it, held[it][0][chain], next_eps)
            mu, s_bar, x_bar, cnt = np.log(10.0 * next_eps), 0.0, 0.0, 0.0
            updates += 1
    if len(d) >= nw:
        assert abs(eps_final / np.exp(x_bar) - 1.0) < 1e-12, (eps_final, np.exp(x_bar))   # complete_adaptation
        assert updates == len(windows)
    for it in range(nw, len(d)):                                     # sampling: the adapted step size and metric
        assert d[it, 2] == eps_final
        metric_at[it] = minv_final
    chols = {}
    for first, count in replay_rows:
        for it in range(first, first + count):
            M_ = metric_at[it]
            if dense and id(M_) not in chols:
                chols[id(M_)] = np.linalg.cholesky(M_)
            ref = m.transitions_from(chain_id, o, it, d[it - 1, 7:], d[it, 2], M_, chols.get(id(M_)))[0]
            assert np.array_equal(d[it, 3:6], ref[3:6]), (it, d[it, :7], ref[:7])                  # treedepth__, n_leapfrog__, divergent__
            assert np.allclose(d[it, [0, 1, 6]], ref[[0, 1, 6]], rtol=1e-6, atol=1e-8), (it, d[it, :7], ref[:7])
            assert np.allclose(d[it, 7:], ref[7:], rtol=1e-6, atol=1e-7), (it, np.abs(d[it, 7:] - ref[7:]).max())
    return updates


def rows_around(ends, n=3):
    """[(first, count)]: n transitions on either side of every window end."""
    return [r for e in ends for r in ((e - n + 1, n), (e + 1, n))]

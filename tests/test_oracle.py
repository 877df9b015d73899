"""CPU tests of the oracle (the checker itself): known answers, autograd, finite differences."""
import ctypes as C

import numpy as np
import pytest

from conftest import GOLD
from oracle_lib import OracleModel, lib
from us_potus_model_amd import _abi


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    L = lib()
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, want in kat:
        o = (C.c_uint32 * 4)()
        L.oracle_philox((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), o)
        assert list(o) == want


def test_rng_moments():
    L = lib()
    u = np.array([L.oracle_rng_uniform(1843, 1, 0, 1, 0, i) for i in range(20000)])
    assert 0 < u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.01
    a, b = C.c_double(), C.c_double()
    z = []
    for i in range(10000):
        L.oracle_rng_normal_pair(1843, 1, 0, 0, 0, i, C.byref(a), C.byref(b))
        z += [a.value, b.value]
    z = np.array(z)
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1) < 0.03


@pytest.mark.parametrize("name", ["2016", "2012", "2008", "small_full", "small_nomode"])
def test_oracle_matches_golden_and_autograd(cases, name):
    data, variant = cases[name]
    m = OracleModel(data, variant)
    g = np.load(GOLD / f"logprob_{name}.npz")
    assert m.D == g["q"].shape[1] == _abi.num_params(data, variant)
    for i, q in enumerate(g["q"]):
        lp, grad = m.log_prob_grad(q)
        lpf, gradf = m.log_prob_grad(q, fast=True)
        assert lp == pytest.approx(g["lp"][i], rel=1e-13)
        scale = np.abs(g["grad"][i]).max()
        assert np.abs(grad - g["grad"][i]).max() <= 1e-13 * scale
        assert abs(lpf - lp) <= 1e-12 * abs(lp) and np.abs(gradf - grad).max() <= 1e-12 * scale
    # independent torch transcription (autograd) on a fresh point
    q = np.random.default_rng(99).uniform(-1.5, 1.5, m.D)
    lp, grad = m.log_prob_grad(q)
    import stan_transcription as st      # (torch: imported only by the tests that differentiate the transcription)
    lpt, gt, aux = st.log_prob_grad(data, q, variant)
    assert abs(lp - lpt) <= 1e-12 * abs(lpt)
    assert np.abs(grad - gt).max() <= 1e-12 * np.abs(gt).max()
    # write_array against the transcription's intermediates
    layout, ncols = _abi.column_layout(data, variant)
    assert ncols == m.n_cols
    wa = m.write_array(q)
    S, T = int(data["S"]), int(data["T"])
    a, b, _ = layout["mu_b"]
    assert np.allclose(wa[a - 7:b - 7].reshape(T, S).T, aux["mu_b"], rtol=1e-12, atol=1e-14)
    a, b, _ = layout["predicted_score"]
    ps = wa[a - 7:b - 7].reshape(S, T).T          # [T, S]
    assert np.allclose(ps, 1 / (1 + np.exp(-aux["mu_b"].T)), rtol=1e-12)
    a, b, _ = layout["logit_pi_democrat_state"]
    assert np.allclose(wa[a - 7:b - 7], aux["eta_s"], rtol=1e-12, atol=1e-13)
    a, b, _ = layout["logit_pi_democrat_national"]
    assert np.allclose(wa[a - 7:b - 7], aux["eta_n"], rtol=1e-12, atol=1e-13)


def test_oracle_matches_autograd_at_the_stress_shape():
    """BASELINE configs[4] sizes (51 x 600 days x 10 000 polls, D = 41 610): the oracle the GPU test at that size is
    checked against is itself pinned to the independent torch transcription."""
    from us_potus_model_amd import synthetic
    data = synthetic.stress()
    m = OracleModel(data, "full")
    assert m.D == 41610
    q = np.random.default_rng(7).uniform(-1.5, 1.5, m.D)
    lp, grad = m.log_prob_grad(q)
    lpf, gradf = m.log_prob_grad(q, fast=True)
    import stan_transcription as st
    lpt, gt, _ = st.log_prob_grad(data, q, "full")
    assert abs(lp - lpt) <= 1e-12 * abs(lpt) and abs(lpf - lpt) <= 1e-12 * abs(lpt)
    scale = np.abs(gt).max()
    assert np.abs(grad - gt).max() <= 1e-12 * scale and np.abs(gradf - gt).max() <= 1e-12 * scale


@pytest.mark.parametrize("name", ["small_full", "small_nomode"])
def test_finite_differences(cases, name):
    data, variant = cases[name]
    m = OracleModel(data, variant)
    rng = np.random.default_rng(5)
    q = rng.uniform(-1, 1, m.D)
    _, g = m.log_prob_grad(q)
    for i in rng.choice(m.D, 40, replace=False):
        h = 1e-5
        qp, qm = q.copy(), q.copy()
        qp[i] += h
        qm[i] -= h
        fd = (m.log_prob_grad(qp)[0] - m.log_prob_grad(qm)[0]) / (2 * h)
        assert fd == pytest.approx(g[i], rel=2e-5, abs=2e-4)


def test_data_block_validation(cases):
    data, variant = cases["small_full"]
    bad = dict(data)
    bad["state"] = np.array(data["state"]).copy()
    bad["state"][0] = int(data["S"]) + 1          # declared legal (stan:9) but out of range at stan:97
    with pytest.raises(ValueError):
        OracleModel(bad, variant)
    bad = dict(data)
    bad["unadjusted_state"] = np.array(data["unadjusted_state"]).copy()
    bad["unadjusted_state"][3] = 1.5
    with pytest.raises(ValueError):
        OracleModel(bad, variant)
    bad = dict(data)
    cov = np.array(data["state_covariance_0"]).copy()
    cov[0, 0] = -1.0
    bad["state_covariance_0"] = cov
    with pytest.raises(ValueError):
        OracleModel(bad, variant)


def test_cholesky_factors(cases):
    data, variant = cases["2016"]
    m = OracleModel(data, variant)
    LB, LT, LW = m.cholesky()
    w, cov = np.asarray(data["state_weights"]), np.asarray(data["state_covariance_0"])
    nsd = np.sqrt(w @ cov @ w)
    for L, sc in ((LB, data["polling_bias_scale"]), (LT, data["mu_b_T_scale"]), (LW, data["random_walk_scale"])):
        assert np.allclose(L @ L.T, cov * (sc / nsd) ** 2, rtol=1e-11, atol=1e-16)
        assert np.allclose(L, np.tril(L))
        # the scaling is what the R script prints as a check (final_2016.R:344-353)
        assert np.sqrt(w @ (L @ L.T) @ w) == pytest.approx(sc, rel=1e-10)


@pytest.mark.parametrize("name", ["2016", "2012", "2008", "small_full"])
def test_the_three_factors_are_one_matrix_times_three_scalars(cases, name):
    """stan:42-55 derives the three covariances from state_covariance_0 and three scales, so cholesky_ss_cov_mu_b_T = aT * cholesky_ss_cov_mu_b_walk and
    cholesky_ss_cov_poll_bias = aB * cholesky_ss_cov_mu_b_walk with aT = mu_b_T_scale / random_walk_scale, aB = polling_bias_scale / random_walk_scale.
    The cluster pass of the device path relies on it ("One matrix, three scalars": no 51 x 51 product besides the walk's); the oracle keeps Stan's
    three factorisations, so this pins the identity on the factors the parity tests compare against."""
    data, variant = cases[name]
    if name == "small_full":
        # Stan puts no bound on the scales and only uses their squares (stan:50-52): a negative scale gives the same factors, and the
        # multiples are |aT|, |aB| (the cluster pass used the signed ratio)
        data = dict(data, mu_b_T_scale=-data["mu_b_T_scale"], random_walk_scale=-data["random_walk_scale"], polling_bias_scale=-data["polling_bias_scale"])
        pos = [np.array(L_) for L_ in OracleModel(cases[name][0], variant).cholesky()]
        assert all(np.array_equal(a, b) for a, b in zip(pos, OracleModel(data, variant).cholesky()))
        data = dict(data, random_walk_scale=-data["random_walk_scale"])         # now the ratios are negative
    LB, LT, LW = OracleModel(data, variant).cholesky()
    aT, aB = abs(data["mu_b_T_scale"] / data["random_walk_scale"]), abs(data["polling_bias_scale"] / data["random_walk_scale"])
    scale = np.abs(LW).max()
    assert np.abs(LT - aT * LW).max() <= 4e-15 * aT * scale
    assert np.abs(LB - aB * LW).max() <= 4e-15 * aB * scale
    # ... and what the device does with it: L_T z_T + L_B z_b + L_W c = L_W (aT z_T + aB z_b + c), to rounding
    rng = np.random.default_rng(7)
    zT, zb, c = rng.standard_normal((3, LW.shape[0]))
    lhs, rhs = LT @ zT + LB @ zb + LW @ c, LW @ (aT * zT + aB * zb + c)
    assert np.abs(lhs - rhs).max() <= 1e-14 * np.abs(lhs).max()


def test_sampler_small_posterior(cases):
    """Adaptive NUTS on the small case: sane adaptation, R-hat ~ 1, literal == fast trajectories."""
    from us_potus_model_amd import diagnostics as dg
    data, variant = cases["small_full"]
    m = OracleModel(data, variant)
    o = m.default_opts(num_warmup=200, num_samples=200, fast_grad=1, seed=1843)
    res = [m.sample_chain(c, o) for c in (1, 2, 3, 4)]
    x = np.stack([r[0] for r in res])
    assert np.isfinite(x).all()
    assert (x[:, :, 5] == 0).mean() > 0.98                      # hardly any divergences
    assert 0.6 < x[:, :, 1].mean() < 0.99                       # accept_stat near the 0.8 target
    assert all(0.01 < r[1][0] < 1.0 for r in res)               # adapted step size
    rh = [dg.rhat(x[:, :, 7 + j]) for j in range(0, m.D, 7)]
    assert np.nanmax(rh) < 1.06
    # n_leapfrog is 2^depth - 1 unless the last doubling was cut short
    assert (x[:, :, 4] <= 2 ** x[:, :, 3] * 2 - 1).all() and (x[:, :, 4] >= 2 ** (x[:, :, 3] - 1)).all()
    # energy = H at the sample: energy + lp = kinetic >= 0
    assert (x[:, :, 6] + x[:, :, 0] >= 0).all()
    o_lit = m.default_opts(num_warmup=10, num_samples=0, fast_grad=0, seed=7, save_warmup=1)
    o_fast = m.default_opts(num_warmup=10, num_samples=0, fast_grad=1, seed=7, save_warmup=1)
    a, b = m.sample_chain(1, o_lit)[0], m.sample_chain(1, o_fast)[0]
    assert np.allclose(a[:5], b[:5], rtol=1e-7, atol=1e-7)


def test_dense_metric_oracle(cases):
    """stan::mcmc::dense_e_metric + covar_adaptation restated (groundwork for BASELINE configs[4]; the device path
    has the diagonal metric only so far).  Checks: the adapted metric is symmetric positive definite and its diagonal
    agrees with the diagonal-metric adaptation; the posterior agrees with the diagonal-metric chains within MCSE;
    with unit metrics (no adaptation windows) both samplers make the same transitions."""
    from us_potus_model_amd import diagnostics as dg
    data, variant = cases["small_full"]
    m = OracleModel(data, variant)
    # no windows (num_warmup < 20): identity metric both ways -> identical chains
    o_d = m.default_opts(num_warmup=10, num_samples=10, seed=3, fast_grad=1, dense_metric=0)
    o_f = m.default_opts(num_warmup=10, num_samples=10, seed=3, fast_grad=1, dense_metric=1)
    a, b = m.sample_chain(1, o_d)[0], m.sample_chain(1, o_f)[0]
    assert np.array_equal(a[:, 3:6], b[:, 3:6]) and np.allclose(a[:, 7:], b[:, 7:], rtol=1e-9, atol=1e-12)
    nw, ns = 150, 100
    diag, dense, metrics = [], [], []
    for c in (1, 2):
        diag.append(m.sample_chain(c, m.default_opts(num_warmup=nw, num_samples=ns, seed=21, fast_grad=1))[0][:, 7:])
        dr, ad, nl, M = m.sample_chain_metric(c, m.default_opts(num_warmup=nw, num_samples=ns, seed=21, fast_grad=1, dense_metric=1))
        dense.append(dr[:, 7:]); metrics.append((ad, M))
    ad, M = metrics[0]
    assert np.allclose(M, M.T, rtol=0, atol=0) and np.linalg.eigvalsh(M).min() > 0
    assert np.allclose(np.diag(M), ad[1:])
    ad_diag = m.sample_chain(1, m.default_opts(num_warmup=nw, num_samples=0, seed=21, fast_grad=1))[1]
    assert abs(np.median(np.log(np.diag(M) / ad_diag[1:]))) < 0.25        # same scale as the diagonal adaptation
    diag, dense = np.stack(diag), np.stack(dense)                           # [chain, draw, D]
    for j in range(0, m.D, 7):
        mcse = np.sqrt(diag[:, :, j].var() / max(dg.ess_bulk(diag[:, :, j]), 10) + dense[:, :, j].var() / max(dg.ess_bulk(dense[:, :, j]), 10))
        assert abs(diag[:, :, j].mean() - dense[:, :, j].mean()) < 5 * mcse, j


def test_oracle_chains_are_independent_streams(cases):
    data, variant = cases["small_full"]
    m = OracleModel(data, variant)
    o = m.default_opts(num_warmup=5, num_samples=0, seed=1843, save_warmup=1)
    a, b = m.sample_chain(1, o)[0], m.sample_chain(2, o)[0]
    a2 = m.sample_chain(1, o)[0]
    assert np.array_equal(a, a2) and not np.allclose(a[:, 7:], b[:, 7:])


@pytest.mark.parametrize("year", ["2008", "2012", "2016"])
def test_oracle_posterior_against_everything_the_reference_published(year):
    """The oracle's pin to the reference: the only outputs of this path the reference publishes are the README tables
    (election-day predicted_score of 51 states + the nation, README.md:83-136 / 179-232 / 279-332) and the Brier
    scores (README.md:75,169,260) -- tests/golden/readme_<year>.csv, made by scripts/make_readme_golden.py.  The
    committed oracle run of the same configuration (8 chains x 1000 + 1000, seed 1843; tests/golden/posterior_<year>.npz,
    made by scripts/make_golden.py posterior) reproduces all 3 x 52 rows to the third decimal the tables are rounded to
    (observed worst: mean 0.0019, interval ends 0.0043, P(win) 0.031), the three scores and the RMSE against the certified results."""
    from conftest import readme_golden, rmse_ex_dc
    from us_potus_model_amd import dataprep
    g = np.load(GOLD / f"posterior_{year}.npz")
    meta = dataprep.load_npz(GOLD / f"data_{year}.npz")["meta"]
    pub, rows = readme_golden(year)
    assert len(rows) == 52
    states = list(meta["states"])
    for r in rows:
        if r["state"] == "--":
            got = (g["national__mean"], g["national__q025"], g["national__q975"], g["national__p_win"])
        else:
            i = states.index(r["state"])
            got = (g["predicted_score_T__mean"][i], g["predicted_score_T__q025"][i], g["predicted_score_T__q975"][i], g["predicted_score_T__p_win"][i])
        assert abs(got[0] - float(r["mean"])) <= 0.004, (year, r, got)
        assert abs(got[1] - float(r["low"])) <= 0.007 and abs(got[2] - float(r["high"])) <= 0.007, (year, r, got)
        assert abs(got[3] - float(r["prob"])) <= 0.045, (year, r, got)
    ev = np.asarray(meta["ev_state"], dtype=float)
    p = g["predicted_score_T__p_win"]
    won = np.array([int(next(r for r in rows if r["state"] == s)["won_readme"]) for s in states])
    assert abs(np.sum(ev / ev.sum() * (won - p) ** 2) - pub["ev_wtd_brier"]) <= 0.003
    assert abs(np.mean((won - p) ** 2) - pub["unwtd_brier"]) <= 0.002
    assert int(np.sum(np.round(p) == won)) == int(pub["states_correct"])
    # README.md:79,175,275 -- the only published figures with seven digits: RMSE of the election-day means against the certified
    # results over the 50 states (observed: 1.3e-4 / 0.8e-4 / 0.7e-4 below the published 0.02318035 / 0.02247233 / 0.02724916)
    assert abs(rmse_ex_dc(states, g["predicted_score_T__mean"], rows) - pub["rmse_ex_dc"]) <= 3e-4


@pytest.mark.parametrize("name", ["small_full", "small_nomode"])
def test_pooled_tree_gives_the_same_draws(cases, name):
    """oracle_opts.pooled (what bench.py's cpu_baseline times: per-level buffers, fused loops) is the same sampler as the
    recursion written the way upstream writes it: same draws, sampler columns and adapted metric, bit for bit -- across
    adaptation windows, divergent leaves and both gradient forms."""
    data, variant = cases[name]
    m = OracleModel(data, variant)
    for fast in (0, 1):
        kw = dict(num_warmup=120, num_samples=30, seed=11, fast_grad=fast, save_warmup=1)
        d0, a0, n0 = m.sample_chain(3, m.default_opts(pooled=0, **kw))
        d1, a1, n1 = m.sample_chain(3, m.default_opts(pooled=1, **kw))
        assert n0 == n1 and np.array_equal(d0, d1) and np.array_equal(a0, a1)
        assert d0[:, 3].max() >= 5            # real trees


def test_transitions_from_replays_a_chain(cases):
    """oracle_transitions_from (single transitions from given states under a given metric: the checker of the dense sampler
    at sizes where a whole oracle run is out of reach) reproduces the rows of the chain it is fed from, diagonal and dense."""
    data, variant = cases["small_full"]
    m = OracleModel(data, variant)
    for dense in (0, 1):
        o = m.default_opts(num_warmup=40, num_samples=6, seed=5, save_warmup=1, dense_metric=dense)
        d, adapt, _, metric = m.sample_chain_metric(2, o)
        # the sampling phase: fixed metric and step size; row t starts from the draw before it
        qs, eps = d[39:45, 7:], d[40:46, 2]
        minv = metric if dense else adapt[1:]
        chol = np.linalg.cholesky(metric) if dense else None
        rows = m.transitions_from(2, o, 40, qs, eps, minv, chol)
        assert np.array_equal(rows[:, 3:6], d[40:46, 3:6])
        np.testing.assert_allclose(rows[:, [0, 1, 6]], d[40:46][:, [0, 1, 6]], rtol=1e-9)
        np.testing.assert_allclose(rows[:, 7:], d[40:46, 7:], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("nw,windows,dense", [(500, [(75, 99), (100, 149), (150, 249), (250, 449)], 0), (400, [(75, 99), (100, 149), (150, 349)], 0),
                                              (280, [(75, 99), (100, 229)], 0), (30, [(4, 26)], 0),
                                              (280, [(75, 99), (100, 229)], 1), (30, [(4, 26)], 1)])   # (the oracle's dense_e chain costs 25 s per 280 iterations)
def test_the_replay_harness_accepts_the_oracles_own_chain(cases, nw, windows, dense):
    """tests/adaptation_replay.py is what holds the device samplers to Stan's adaptation window by window (GPU suite).  Here the harness
    itself is checked on a chain that must satisfy it: the oracle's own (its window schedule, oracle/potus_oracle.c:1006-1065, against the
    Python restatement of windowed_adaptation::compute_next_window -- doubling, the stretched last window, the constructor's rescaling for
    short warm-ups), diagonal and dense."""
    from types import SimpleNamespace
    from adaptation_replay import adaptation_replayed_from_the_device_rows, rows_around, window_schedule
    data, variant = cases["small_full"]
    assert window_schedule(nw, 75, 50, 25) == windows
    m = OracleModel(data, variant)
    o = m.default_opts(num_warmup=nw, num_samples=3, seed=11, save_warmup=1, dense_metric=dense, fast_grad=1)
    d, adapt, _, metric = m.sample_chain_metric(1, o)
    opts = SimpleNamespace(num_warmup=nw, num_samples=3, max_depth=o.max_depth, init_buffer=75, term_buffer=50, window=25, chain_id_offset=0, chains=1,
                           delta=o.delta, gamma=o.gamma, kappa=o.kappa, t0=o.t0, stepsize=o.stepsize, init_radius=o.init_radius,
                           metric=_abi.METRIC_DENSE if dense else _abi.METRIC_DIAG)
    fake = SimpleNamespace(opts=opts, D=m.D, draws=lambda: d[None], adaptation=lambda: (np.array([adapt[0]]), adapt[None, 1:]),
                           dense_metric=lambda c: metric)
    n = adaptation_replayed_from_the_device_rows(data, variant, fake, 0, 11, [(1, 2)] + rows_around([e for _, e in windows], 2) + [(nw, 3)])
    assert n == len(windows)

"""Dense-metric NUTS (metric = dense_e: stan::mcmc::dense_e_metric + covar_adaptation, BASELINE configs[4]) on the
device, potus_dense.hpp: its pieces against numpy, the sampler against the oracle's dense_e restatement
(oracle_opts.dense_metric) and against its own saved draws."""
import ctypes as C

import numpy as np
import pytest

from oracle_lib import OracleModel
from us_potus_model_amd import Handle, _abi, diagnostics as dg, sampler

pytestmark = pytest.mark.gpu
DP = C.POINTER(C.c_double)


def _lib():
    L = sampler.load_library()
    L.potus_dense_matvec_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, DP, C.POINTER(C.c_longlong)]
    L.potus_dense_factor_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, DP]
    L.potus_dense_pool_matvec_probe.argtypes = L.potus_dense_matvec_probe.argtypes
    L.potus_dense_pool_factor_probe.argtypes = L.potus_dense_factor_probe.argtypes
    return L


@pytest.mark.parametrize("chains,D,nrhs", [(3, 1000, 1), (2, 2049, 2), (1, 16500, 3), (2, 1027, 3), (1, 71, 2), (2, 640, 2), (1, 129, 1)])
def test_dense_matvec_against_numpy(chains, D, nrhs):
    """[y_r] = M^-1 [x_r] for up to three right-hand sides out of the strict upper triangle + diagonal vector
    (k_dn_symv + k_dn_symv_finish): partial row blocks, paired blocks with and without a middle one, several column
    tiles, odd D (padded rows); the lower triangle of the uploaded matrix (where the sampler keeps the Cholesky factor)
    must not matter; the fused x_0 . M^-1 x_0; reproducible bit for bit."""
    L = _lib()
    rng = np.random.default_rng(4)
    B = rng.standard_normal((chains, D, 8))
    M = np.einsum("cik,cjk->cij", B, B) / 8 + np.eye(D)[None]          # symmetric positive definite
    Mdev = M + np.tril(rng.standard_normal((chains, D, D)), -1)        # garbage below the diagonal: it is not the metric's
    x = rng.standard_normal((chains, nrhs, D))
    out = []
    for _ in range(2):
        y, dot, ms = np.zeros((chains, nrhs, D)), np.zeros(chains), C.c_double()
        nb = C.c_longlong()
        assert L.potus_dense_matvec_probe(0, chains, D, nrhs, Mdev.ctypes.data, x.ctypes.data, y.ctypes.data, dot.ctypes.data, 2, C.byref(ms), C.byref(nb)) == 0
        assert 4 * D * D * (2 if nrhs == 3 else 1) <= nb.value <= (4 * D * D + 8 * 704 * D) * (2 if nrhs == 3 else 1)   # the upper triangle + at most a band of tile width
        out.append((y, dot))
    ref = np.einsum("cij,crj->cri", M, x)
    assert np.abs(out[0][0] - ref).max() <= 1e-12 * np.abs(ref).max() * np.sqrt(D)
    assert np.allclose(out[0][1], np.einsum("ci,ci->c", x[:, 0], ref[:, 0]), rtol=1e-11)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("chains,D,n", [(2, 300, 40), (1, 1000, 25), (1, 2051, 150)])
def test_dense_covariance_cholesky_and_solve_against_numpy(chains, D, n):
    """The window-end pieces on caller data: M^-1 = n/(n+5) cov + 1e-3 5/(n+5) I (covar_adaptation::learn_covariance),
    its blocked Cholesky factor, and the momentum draw's triangular solve p = L^-T u."""
    import scipy.linalg as sl
    L = _lib()
    rng = np.random.default_rng(7)
    draws = rng.standard_normal((chains, n, D)) * rng.uniform(0.2, 3.0, (1, 1, D)) + rng.standard_normal((chains, 1, D))
    u = rng.standard_normal((chains, D))
    Mi, Lc, p, ms = np.zeros((chains, D, D)), np.zeros((chains, D, D)), np.zeros((chains, D)), (C.c_double * 3)()
    assert L.potus_dense_factor_probe(0, chains, D, n, draws.ctypes.data, u.ctypes.data, Mi.ctypes.data, Lc.ctypes.data, p.ctypes.data, ms) == 0
    for c in range(chains):
        ref = (n / (n + 5.0)) * np.cov(draws[c].T) + 1e-3 * (5.0 / (n + 5.0)) * np.eye(D)
        assert np.allclose(Mi[c], ref, rtol=1e-10, atol=1e-12 * np.abs(ref).max())
        assert np.array_equal(Mi[c], Mi[c].T)                                 # exactly symmetric
        Lref = np.linalg.cholesky(Mi[c])
        Lg = np.tril(Lc[c])
        assert np.allclose(Lg, Lref, rtol=1e-8, atol=1e-10 * np.abs(Lref).max())
        assert np.allclose(Lg @ Lg.T, Mi[c], rtol=1e-10, atol=1e-12 * np.abs(ref).max())
        pref = sl.solve_triangular(Lg.T, u[c], lower=False)
        assert np.allclose(p[c], pref, rtol=1e-8, atol=1e-9 * np.abs(pref).max())


def _window_metric(draws_window):
    n = draws_window.shape[0]
    return (n / (n + 5.0)) * np.cov(draws_window.T) + 1e-3 * (5.0 / (n + 5.0)) * np.eye(draws_window.shape[1])


@pytest.mark.parametrize("cus", [1, 8])
def test_dense_sampler_follows_the_oracle_and_adapts_its_metric(cases, cus):
    """30 warm-up iterations of the small model (windowed_adaptation rescales to init 4 / window 23 / term 3: the metric
    is updated after iteration 27): the transitions before the update follow the oracle's dense_e chain (same Philox
    streams; the unit metric makes M^-1 p exact), the adapted M^-1 is the regularised covariance of the very draws the
    sampler saved for that window, and it agrees with the oracle's."""
    data, variant = cases["small_full"]
    nw = 30
    h = Handle(data, variant, chains=2, num_warmup=nw, num_samples=0, save_warmup=1, seed=1843, metric=_abi.METRIC_DENSE, cus_per_chain=cus)
    h.init(); h.run(17); h.run(13)
    d = h.draws()
    m = OracleModel(data, variant)
    o = m.default_opts(num_warmup=nw, num_samples=0, save_warmup=1, seed=1843, fast_grad=1, dense_metric=1)
    for c in (0, 1):
        ref, ad, nl, Mref = m.sample_chain_metric(c + 1, o)
        k = 12
        assert np.array_equal(d[c][:k, 3:6], ref[:k, 3:6]), (c, d[c][:k, :7], ref[:k, :7])      # depth, n_leapfrog, divergent
        assert np.allclose(d[c][:k, :3], ref[:k, :3], rtol=1e-6, atol=1e-9)
        assert np.allclose(d[c][:k, 7:], ref[:k, 7:], rtol=1e-6, atol=1e-7)
        Mi = h.dense_metric(c)
        want = _window_metric(d[c][4:27, 7:])                # draws of iterations 4 .. 26 (0-based): the window
        assert np.allclose(Mi, want, rtol=1e-9, atol=1e-12), np.abs(Mi - want).max()
        assert np.array_equal(Mi, Mi.T) and np.linalg.eigvalsh(Mi).min() > 0
        eps, diag = h.adaptation()
        assert np.allclose(diag[c], np.diag(Mi)) and eps[c] > 0
        if np.allclose(d[c][:27, 7:], ref[:27, 7:], rtol=1e-5, atol=1e-6):          # still in step with the oracle at the window's end
            assert np.allclose(Mi, Mref, rtol=1e-3, atol=1e-6 * np.abs(Mref).max())
            assert np.array_equal(d[c][27:, 3:6], ref[27:, 3:6]) and np.allclose(d[c][27:, 2], ref[27:, 2], rtol=1e-4)   # dense-metric transitions
    assert np.isfinite(d).all()
    ms, passes, nbytes, rounds = h.dense_timing()
    assert passes > 0 and nbytes > 0 and rounds > 0 and ms > 0
    h.close()


@pytest.mark.parametrize("cus", [1, 8])
@pytest.mark.parametrize("nw,windows", [(280, [(75, 99), (100, 229)]), (200, [(75, 99), (100, 149)])])
def test_dense_adaptation_replayed_through_every_window_end(cases, nw, windows, cus):
    """The dense sampler's adaptation lives on the HOST (potus_hmc.hip: the window counters of dense_transition_end, a mirror of
    windowed_adaptation), so its schedule needs its own replay: 280 warm-up iterations = 75 | 25, then the second
    window stretched to 130 | 50; 200 = 75 | 25, 50 (doubled) | 50.  Every step size from the device's own accept_stat__ column, the
    D x D metric the device holds after EVERY update against covar_adaptation's regularised covariance of the very draws of that
    window, init_stepsize under that matrix, and two transitions on either side of every window end replayed by the oracle."""
    from adaptation_replay import adaptation_replayed_from_the_device_rows, rows_around, run_through_the_windows, window_schedule
    data, variant = cases["small_full"]
    assert window_schedule(nw, 75, 50, 25) == windows
    h = Handle(data, variant, chains=2, num_warmup=nw, num_samples=2, save_warmup=1, seed=1843, metric=_abi.METRIC_DENSE, cus_per_chain=cus)
    h.init()
    held = run_through_the_windows(h, nw + 2)
    assert sorted(held) == [e for _, e in windows]
    for c in (0, 1):
        assert adaptation_replayed_from_the_device_rows(data, variant, h, c, 1843, [(1, 2)] + rows_around([e for _, e in windows], 2) + [(nw, 2)], held) == len(windows)
    h.close()


def test_dense_sampler_is_reproducible_and_chunk_invariant(cases):
    data, variant = cases["small_nomode"]
    kw = dict(chains=3, num_warmup=40, num_samples=10, seed=7, metric=_abi.METRIC_DENSE)
    a = Handle(data, variant, **kw); a.init(); a.run(50); da = a.draws(); a.close()
    b = Handle(data, variant, **kw); b.init(); b.run(33); b.run(3); b.run(14); db = b.draws(); b.close()
    assert np.array_equal(da, db)
    c = Handle(data, variant, **{**kw, "chains": 1, "chain_id_offset": 2}); c.init(); c.run(50); dc = c.draws(); c.close()
    assert np.array_equal(da[2], dc[0])


def test_dense_posterior_parity_small(cases):
    """Statistical parity with the oracle's dense_e sampler: pooled means of every unconstrained coordinate within
    5 combined MCSE; the adapted sampler accepts at the target rate."""
    from us_potus_model_amd import synthetic
    # (D = 117 < the 200 draws of the last adaptation window: with fewer draws than dimensions the regularised covariance
    #  is so ill-conditioned that every transition runs to the maximum tree depth -- in Stan as here)
    variant = "full"
    data = synthetic.make(S=4, T=12, N_state=30, N_national=8, P=3, seed=3, variant=variant)
    nw = ns = 400
    h = Handle(data, variant, chains=4, num_warmup=nw, num_samples=ns, seed=1843, metric=_abi.METRIC_DENSE)
    h.init(); h.run(nw + ns)
    d = h.draws()
    x = d[:, :, 7:]
    st, _ = h.chain_status()
    assert st == [0, 0, 0, 0] and d[:, :, 5].mean() < 0.02
    assert 0.6 < d[:, :, 1].mean() < 0.97                              # accept_stat__ around delta = 0.8
    m = OracleModel(data, variant)
    o = m.default_opts(num_warmup=nw, num_samples=ns, seed=4242, fast_grad=1, dense_metric=1)
    y = np.stack([m.sample_chain(c, o)[0][:, 7:] for c in (1, 2, 3, 4)])
    worst = 0.0
    for j in range(h.D):
        a, b = x[:, :, j], y[:, :, j]
        se = np.hypot(a.std() / np.sqrt(dg.ess_mean(a)), b.std() / np.sqrt(dg.ess_mean(b)))
        worst = max(worst, abs(a.mean() - b.mean()) / se)
        assert dg.rhat(a) < 1.08
    assert worst < 5.0, worst
    h.close()


@pytest.mark.parametrize("cus", [0, 1])
def test_dense_sampler_on_the_2016_posterior(cases, cus):
    """D = 15 098: 1.8 GB of inverse metric per chain.  40 warm-up iterations cross one window end (covariance of 30
    draws, 236-block Cholesky, init_stepsize); the metric equals the regularised covariance of the saved window draws."""
    data, variant = cases["2016"]
    nw = 40
    h = Handle(data, variant, chains=2, num_warmup=nw, num_samples=0, save_warmup=1, seed=1843, metric=_abi.METRIC_DENSE, cus_per_chain=cus)
    assert h.cus_per_chain == (16 if cus == 0 else 1)            # the gradient of a round: the cluster pass or one workgroup per chain
    h.init(); h.run(nw)
    d = h.draws()
    assert np.isfinite(d).all() and (d[:, :, 4] >= 1).all()
    ib, bw = int(0.15 * nw), nw - int(0.15 * nw) - int(0.1 * nw)
    Mi = h.dense_metric(1)
    want = _window_metric(d[1][ib:ib + bw, 7:])
    assert np.allclose(Mi, want, rtol=1e-9, atol=1e-12), np.abs(Mi - want).max()
    ms, passes, nbytes, rounds = h.dense_timing()
    print(f"2016 dense (cus_per_chain {h.cus_per_chain}): {passes} matrix passes, {nbytes / 1e9:.1f} GB in {ms:.1f} ms = {nbytes / ms / 1e9 * 1e3 / 1e3:.2f} TB/s; {rounds} leaf rounds")
    assert d[:, -3:, 1].mean() > 0.3                                   # transitions under the adapted metric accept
    h.close()


def test_dense_sixteen_chains_of_the_stress_shape_fit_one_gpu():
    """BASELINE configs[4] as specified: 16 chains per GPU, 51 states x 600 days x 10 000 polls, dense metric.  One
    13.85 GB matrix per chain (M^-1 above the diagonal, its Cholesky factor below) = 222 GB of the 288; the first
    transition runs (unit metric: it must be the diagonal sampler's first transition)."""
    from us_potus_model_amd import synthetic
    data = synthetic.stress()
    kw = dict(chains=16, num_warmup=1, num_samples=0, save_warmup=1, seed=5)
    h = Handle(data, "full", metric=_abi.METRIC_DENSE, **kw)
    assert h.cus_per_chain == 16 and h.D == 41610
    h.init(); h.run(1)
    d = h.draws()
    ms, passes, nbytes, rounds = h.dense_timing()
    print(f"stress shape, 16 dense chains: {passes} passes, {nbytes / 1e12:.2f} TB in {ms:.0f} ms = {nbytes / ms / 1e9:.2f} TB/s")
    h.close()
    g = Handle(data, "full", **kw)
    g.init(); g.run(1)
    e = g.draws()
    g.close()
    assert np.array_equal(d[:, :, 3:6], e[:, :, 3:6])                    # depth, n_leapfrog, divergent
    assert np.allclose(d[:, :, 7:], e[:, :, 7:], rtol=1e-9, atol=1e-10) and np.allclose(d[:, :, 0], e[:, :, 0], rtol=1e-10)


def test_dense_product_does_not_depend_on_the_launch_shape():
    """How the column tiles are dealt to workgroups depends on the number of chains taking part in a launch; a chain's
    M^-1 x must not (its draws would otherwise depend on what its companions are doing): one chain alone and the same
    chain among sixteen give the same bytes."""
    L = _lib()
    D, nrhs = 4100, 2
    rng = np.random.default_rng(11)
    B = rng.standard_normal((D, 8))
    M1 = (B @ B.T) / 8 + np.eye(D)
    x1 = rng.standard_normal((nrhs, D))
    outs = []
    for chains in (1, 16):
        M = np.ascontiguousarray(np.broadcast_to(M1, (chains, D, D)))
        x = np.ascontiguousarray(np.broadcast_to(x1, (chains, nrhs, D)))
        y, dot, ms, nb = np.zeros((chains, nrhs, D)), np.zeros(chains), C.c_double(), C.c_longlong()
        assert L.potus_dense_matvec_probe(0, chains, D, nrhs, M.ctypes.data, x.ctypes.data, y.ctypes.data, dot.ctypes.data, 1, C.byref(ms), C.byref(nb)) == 0
        outs.append((y, dot))
    assert np.array_equal(outs[0][0][0], outs[1][0][0]) and np.array_equal(outs[1][0][0], outs[1][0][15])
    assert outs[0][1][0] == outs[1][1][7]
    assert np.allclose(outs[0][0][0], x1 @ M1, rtol=1e-11, atol=1e-11)


# ---------------------------------------------------------------------------------------------------- round 3
def _post_window_rows_against_the_oracle(data, variant, h, d, chain, first, count, max_depth, metric, chain_id):
    """Transitions first .. first + count - 1 of the device chain, each replayed by the oracle FROM THE DEVICE'S OWN STATE: the
    draw before it, the step size it used (stepsize__ of its own row), the device's inverse metric (potus_get_dense_metric) and
    that matrix's Cholesky factor.  Nothing can drift: every row is an independent comparison."""
    m = OracleModel(data, variant)
    o = m.default_opts(num_warmup=h.opts.num_warmup, num_samples=h.opts.num_samples, seed=int(h.opts.seed), fast_grad=1, dense_metric=1,
                       max_depth=max_depth)
    rows = d[chain][first:first + count]
    qs, eps = d[chain][first - 1:first + count - 1, 7:], rows[:, 2]
    Lc = np.linalg.cholesky(metric)
    ref = m.transitions_from(chain_id, o, first, qs, eps, metric, Lc)
    assert np.array_equal(rows[:, 3:6], ref[:, 3:6]), (rows[:, :7], ref[:, :7])                # treedepth__, n_leapfrog__, divergent__
    assert np.allclose(rows[:, [0, 1, 6]], ref[:, [0, 1, 6]], rtol=1e-6, atol=1e-8), (rows[:, :7], ref[:, :7])
    assert np.allclose(rows[:, 7:], ref[:, 7:], rtol=1e-6, atol=1e-7)
    return ref


@pytest.mark.parametrize("storage", [_abi.STORAGE_F64, _abi.STORAGE_F32])
def test_dense_transitions_after_the_window_match_the_oracle_at_2016_size(cases, storage):
    """D = 15 098, after the metric update (30 warm-up iterations: window of 23 draws, update after iteration 26; 236-block
    Cholesky).  The three transitions that follow run under the adapted dense metric -- momenta from the blocked back
    substitution L' p = u, p# = M^-1 p out of the upper triangle -- and must be the oracle's transitions from the same state
    under the same matrix: same tree depth, leapfrog count and divergence flag, values to 1e-6.  A wrong k_dn_trsv_*, a wrong
    p# slot or a wrong tile of the symmetric product at this size cannot pass.  (max_depth 6 keeps the oracle's 63 dense
    products per transition within a test's time.)  With fp32 storage the device's metric is the rounded matrix and the oracle
    is handed exactly that."""
    data, variant = cases["2016"]
    nw, md = 30, 6
    h = Handle(data, variant, chains=2, num_warmup=nw, num_samples=0, save_warmup=1, seed=1843, metric=_abi.METRIC_DENSE, max_depth=md,
               metric_storage=storage)
    h.init(); h.run(nw)
    d = h.draws()
    assert np.isfinite(d).all()
    Mi = h.dense_metric(1)
    if storage == _abi.STORAGE_F32:
        assert np.array_equal(Mi, Mi.astype(np.float32).astype(np.float64))                  # every element is an fp32 number
        want = (lambda w: (len(w) / (len(w) + 5.0)) * np.cov(w.T) + 1e-3 * (5.0 / (len(w) + 5.0)) * np.eye(w.shape[1]))(d[1][4:27, 7:])
        assert np.abs(Mi - want).max() <= 6.1e-8 * np.abs(want).max()                        # the window's covariance, rounded
    assert np.array_equal(Mi, Mi.T)
    res, solve = h.dense_check(1, 2)
    assert res < 1e-12 and solve < 1e-9, (res, solve)
    _post_window_rows_against_the_oracle(data, variant, h, d, 1, 27, 3, md, Mi, 2)
    h.close()


def test_dense_factor_is_the_factor_of_the_metric_at_the_stress_size():
    """D = 41 610 (BASELINE configs[4]'s shape): 20 warm-up iterations cross a window end (15 draws, 651-block in-place Cholesky
    of a 13.85 GB matrix).  On the device: L L' x against M^-1 x from the sampler's own matrix pass, and the momentum draw's
    blocked back substitution multiplied back (potus_dense_check)."""
    from us_potus_model_amd import synthetic
    data = synthetic.stress()
    h = Handle(data, "full", chains=1, num_warmup=20, num_samples=0, save_warmup=1, seed=3, metric=_abi.METRIC_DENSE, max_depth=3)
    assert h.D == 41610
    h.init(); h.run(20)
    t = h.dense_adapt_timing()
    assert t["window_ends"] == 1 and t["chol_ms"] > 0
    res, solve = h.dense_check(0, 2)
    print(f"D = 41610: ||L L' x - M^-1 x|| / ||M^-1 x|| = {res:.2e}, ||L' p - u|| / ||u|| = {solve:.2e}; window end: {t}")
    assert res < 1e-12 and solve < 1e-9, (res, solve)
    d = h.draws()
    assert np.isfinite(d).all() and (d[0, :, 4] >= 1).all()       # (trees are cut at depth 3 here: the acceptance rate says nothing)
    h.close()


def test_dense_fp32_storage_samples_the_same_posterior(cases):
    """potus_opts.metric_storage = f32 (M^-1 rounded to fp32 IS the metric; its factor, the momenta and the accumulation stay
    fp64): the posterior of the small model within 5 combined MCSE of the fp64 dense sampler, and a second potus_init puts a
    dense handle back at the unit metric and the start of its window schedule (same bytes as a fresh handle)."""
    from us_potus_model_amd import synthetic
    data = synthetic.make(S=4, T=12, N_state=30, N_national=8, P=3, seed=3, variant="full")
    nw = ns = 400
    runs = {}
    for st in (_abi.STORAGE_F64, _abi.STORAGE_F32):
        h = Handle(data, "full", chains=4, num_warmup=nw, num_samples=ns, seed=1843, metric=_abi.METRIC_DENSE, metric_storage=st)
        h.init(); h.run(nw + ns)
        runs[st] = h.draws()
        if st == _abi.STORAGE_F32:
            assert h.dense_check(0, 2)[0] < 1e-13
            h.init(); h.run(nw + ns)                           # again from the start
            assert np.array_equal(runs[st], h.draws())
        h.close()
    x, y = runs[_abi.STORAGE_F64][:, :, 7:], runs[_abi.STORAGE_F32][:, :, 7:]
    assert not np.array_equal(x, y)
    worst = 0.0
    for j in range(x.shape[2]):
        a, b = x[:, :, j], y[:, :, j]
        se = np.hypot(a.std() / np.sqrt(dg.ess_mean(a)), b.std() / np.sqrt(dg.ess_mean(b)))
        worst = max(worst, abs(a.mean() - b.mean()) / se)
    assert worst < 5.0, worst
    assert abs(runs[_abi.STORAGE_F64][:, :, 1].mean() - runs[_abi.STORAGE_F32][:, :, 1].mean()) < 0.05


# ---------------------------------------------------------------- potus_opts.pooled_metric (csrc/potus_dense_pool.hpp; round 6)
@pytest.mark.parametrize("chains,D,nrhs,f32", [(3, 1000, 1, 0), (2, 2049, 2, 0), (16, 700, 2, 0), (17, 333, 3, 0), (1, 71, 2, 0), (5, 257, 3, 0), (2, 16500, 3, 0),
                                               (4, 4097, 2, 0), (3, 1000, 1, 1), (16, 700, 2, 1), (17, 333, 3, 1), (1, 71, 2, 1), (2, 16500, 3, 1), (4, 4097, 2, 1)])
def test_pooled_product_against_numpy(chains, D, nrhs, f32, monkeypatch):
    """Y = M^-1 X for ONE full symmetric matrix and the right-hand sides of every chain in one pass on the fp64 matrix cores (k_dn_pool_mm +
    k_dn_pool_finish): one to three operand tiles of sixteen right-hand sides, more than 48 of them (several launches), column panels and row
    splits with ragged ends, odd D (padded rows), D below one panel; the fused x_0 . M^-1 x_0 per chain; reproducible bit for bit; and a chain's
    numbers do not depend on which other chains take part in the launch.  f32: the same with metric_storage = f32 -- the matrix rounded to fp32 IS the
    matrix (four columns per 16-byte load, panels of 512 columns), the arithmetic stays fp64: the same tolerance against the rounded matrix, half the bytes."""
    L = _lib()
    monkeypatch.setenv("POTUS_PROBE_F32", str(f32))
    rng = np.random.default_rng(4)
    B = rng.standard_normal((D, 8))
    M = B @ B.T / 8 + np.eye(D)
    Mref = M.astype(np.float32).astype(np.float64) if f32 else M
    x = rng.standard_normal((chains, nrhs, D))
    out = []
    for _ in range(2):
        y, dot, ms, nb = np.zeros((chains, nrhs, D)), np.zeros(chains), C.c_double(), C.c_longlong()
        assert L.potus_dense_pool_matvec_probe(0, chains, D, nrhs, M.ctypes.data, x.ctypes.data, y.ctypes.data, dot.ctypes.data, 2, C.byref(ms), C.byref(nb)) == 0
        assert nb.value == (4 if f32 else 8) * D * ((D + 7) // 8 * 8) * -(-chains * nrhs // 48)   # the whole matrix once per launch of up to 48 right-hand sides
        out.append((y, dot))
    ref = np.einsum("ij,crj->cri", Mref, x)
    assert np.abs(out[0][0] - ref).max() <= 1e-12 * np.abs(ref).max() * np.sqrt(D)
    assert np.allclose(out[0][1], np.einsum("ci,ci->c", x[:, 0], ref[:, 0]), rtol=1e-11)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    if chains >= 3:                                                          # idle companions (compacted list of active chains): same bytes for those that run
        monkeypatch.setenv("POTUS_PROBE_ACTIVE", f"0,{chains - 1}")
        y2, dot2, ms, nb = np.zeros((chains, nrhs, D)), np.zeros(chains), C.c_double(), C.c_longlong()
        assert L.potus_dense_pool_matvec_probe(0, chains, D, nrhs, M.ctypes.data, x.ctypes.data, y2.ctypes.data, dot2.ctypes.data, 1, C.byref(ms), C.byref(nb)) == 0
        for c in (0, chains - 1):
            assert np.array_equal(y2[c], out[0][0][c]) and dot2[c] == out[0][1][c]


@pytest.mark.parametrize("chains,D,n,f32", [(4, 300, 10, 0), (2, 1000, 25, 0), (3, 2051, 40, 0), (4, 300, 10, 1), (3, 2051, 40, 1)])
def test_pooled_window_end_against_numpy(chains, D, n, f32, monkeypatch):
    """The pooled window end on caller data: ONE M^-1 = N/(N+5) cov + 1e-3 5/(N+5) I over the N = chains x n draws of all chains
    (covar_adaptation::learn_covariance applied to the pooled sample), stored as the full symmetric matrix; ONE blocked Cholesky factor in
    its own buffer; every chain's momentum draw p = L^-T u out of that one factor.  f32 (metric_storage = f32): the matrix is that estimate rounded to
    fp32 -- exactly representable, symmetric -- and the factor is the factor OF THE ROUNDED matrix to fp64 accuracy."""
    import scipy.linalg as sl
    L = _lib()
    monkeypatch.setenv("POTUS_PROBE_F32", str(f32))
    rng = np.random.default_rng(11)
    draws = rng.standard_normal((chains, n, D)) * rng.uniform(0.2, 3.0, (1, 1, D)) + rng.standard_normal((chains, 1, D))
    u = rng.standard_normal((chains, D))
    Mi, Lc, p, ms = np.zeros((D, D)), np.zeros((D, D)), np.zeros((chains, D)), (C.c_double * 3)()
    assert L.potus_dense_pool_factor_probe(0, chains, D, n, draws.ctypes.data, u.ctypes.data, Mi.ctypes.data, Lc.ctypes.data, p.ctypes.data, ms) == 0
    N = chains * n
    ref = (N / (N + 5.0)) * np.cov(draws.reshape(N, D).T) + 1e-3 * (5.0 / (N + 5.0)) * np.eye(D)
    if f32:
        assert np.array_equal(Mi, Mi.astype(np.float32).astype(np.float64)) and np.allclose(Mi, ref, rtol=1e-7, atol=1e-7 * np.abs(ref).max()) and np.array_equal(Mi, Mi.T)
    else:
        assert np.allclose(Mi, ref, rtol=1e-10, atol=1e-12 * np.abs(ref).max()) and np.array_equal(Mi, Mi.T)
    Lg = np.tril(Lc)
    assert np.allclose(Lg @ Lg.T, Mi, rtol=1e-10, atol=1e-12 * np.abs(ref).max())
    for c in range(chains):
        pref = sl.solve_triangular(Lg.T, u[c], lower=False)
        assert np.allclose(p[c], pref, rtol=1e-8, atol=1e-9 * np.abs(pref).max())


@pytest.mark.parametrize("cus", [1, 8])
def test_pooled_sampler_replayed_through_every_window_end(cases, cus):
    """potus_opts.pooled_metric on the small model, 3 chains, 200 warm-up iterations = 75 | 25, 50 (doubled) | 50: up to the first window end
    the chains are the per-chain sampler's (unit metric: same bytes as pooled_metric = 0); after EVERY window end all chains hold the SAME
    matrix = the regularised covariance of the window draws of all three (tests/adaptation_replay.py, pooled estimate), every chain's step
    size is init_stepsize's under that matrix, and two transitions on either side of every window end are the oracle's from the device's
    own state, that matrix and its factor.  The matrix pass streams ONE matrix per round."""
    from adaptation_replay import adaptation_replayed_from_the_device_rows, rows_around, run_through_the_windows, window_schedule
    data, variant = cases["small_full"]
    nw, windows = 200, [(75, 99), (100, 149)]
    assert window_schedule(nw, 75, 50, 25) == windows
    kw = dict(chains=3, num_warmup=nw, num_samples=2, save_warmup=1, seed=1843, metric=_abi.METRIC_DENSE, cus_per_chain=cus)
    h = Handle(data, variant, pooled_metric=1, **kw)
    h.init()
    held = run_through_the_windows(h, nw + 2)
    assert sorted(held) == [e for _, e in windows]
    for e in held:
        for c in (1, 2):
            assert np.array_equal(held[e][1][c], held[e][1][0])                 # one matrix for everybody
    for c in range(3):
        assert adaptation_replayed_from_the_device_rows(data, variant, h, c, 1843, [(1, 2)] + rows_around([e for _, e in windows], 2) + [(nw, 2)], held) == len(windows)
    d = h.draws()
    ms, passes, nbytes, rounds = h.dense_timing()
    assert passes > 0 and nbytes == passes * 8 * h.D * ((h.D + 7) // 8 * 8)      # bytes = passes x ONE full matrix, whatever the number of chains
    res, solve = h.dense_check(2, 2)
    assert res < 1e-12 and solve < 1e-9, (res, solve)                            # L L' x = M^-1 x by the pooled pass; L' p = u
    h.close()
    g = Handle(data, variant, pooled_metric=0, **kw)
    g.init(); g.run(75 + 25)
    assert np.array_equal(g.draws()[:, :100], d[:, :100])                       # rows 0 .. 99: before anybody's first metric update
    g.close()


@pytest.mark.parametrize("storage", [_abi.STORAGE_F64, _abi.STORAGE_F32])
def test_pooled_sampler_is_reproducible_and_chunk_invariant(cases, storage):
    data, variant = cases["small_nomode"]
    kw = dict(chains=3, num_warmup=40, num_samples=10, seed=7, metric=_abi.METRIC_DENSE, pooled_metric=1, metric_storage=storage)
    a = Handle(data, variant, **kw); a.init(); a.run(50); da = a.draws(); a.close()
    b = Handle(data, variant, **kw); b.init(); b.run(33); b.run(3); b.run(14); db = b.draws(); b.close()
    assert np.array_equal(da, db) and np.isfinite(da).all()
    a = Handle(data, variant, **kw); a.init(); a.run(20); a.init(); a.run(50); dc = a.draws(); a.close()   # a second potus_init: back to the unit metric
    assert np.array_equal(da, dc)


def test_pooled_posterior_parity_small(cases):
    """Statistical parity of the pooled-metric sampler with the oracle's (per-chain) dense_e sampler: pooled means of every unconstrained
    coordinate within 5 combined MCSE; accept rate around delta."""
    from us_potus_model_amd import synthetic
    variant = "full"
    data = synthetic.make(S=4, T=12, N_state=30, N_national=8, P=3, seed=3, variant=variant)
    nw = ns = 400
    h = Handle(data, variant, chains=4, num_warmup=nw, num_samples=ns, seed=1843, metric=_abi.METRIC_DENSE, pooled_metric=1)
    h.init(); h.run(nw + ns)
    d = h.draws()
    x = d[:, :, 7:]
    st, _ = h.chain_status()
    assert st == [0, 0, 0, 0] and d[:, :, 5].mean() < 0.02 and 0.6 < d[:, :, 1].mean() < 0.97
    m = OracleModel(data, variant)
    o = m.default_opts(num_warmup=nw, num_samples=ns, seed=4242, fast_grad=1, dense_metric=1)
    y = np.stack([m.sample_chain(c, o)[0][:, 7:] for c in (1, 2, 3, 4)])
    worst = 0.0
    for j in range(h.D):
        a, b = x[:, :, j], y[:, :, j]
        se = np.hypot(a.std() / np.sqrt(dg.ess_mean(a)), b.std() / np.sqrt(dg.ess_mean(b)))
        worst = max(worst, abs(a.mean() - b.mean()) / se)
        assert dg.rhat(a) < 1.08
    assert worst < 5.0, worst
    h.close()


@pytest.mark.parametrize("storage", [_abi.STORAGE_F64, _abi.STORAGE_F32])
def test_pooled_transitions_after_the_window_match_the_oracle_at_2016_size(cases, storage):
    """D = 15 098 with ONE 1.8 GB inverse metric for 4 chains (30 warm-up iterations: one window of 4 x 23 pooled draws, update after iteration 26;
    236-block Cholesky in the factor's own buffer).  The matrix every chain reports is the pooled estimate; the three transitions that follow are the
    oracle's from the device's own state under that matrix and its factor -- momenta from the one factor, p# = M^-1 p out of the pooled pass
    (k_dn_pool_mm): same tree depth, leapfrog count and divergence flag, values to 1e-6, for the first and the last chain.  (A posterior-level
    comparison at this size would say nothing: a dense metric adapted on a few hundred draws in 15 098 dimensions leaves every tree at its depth
    limit -- DESIGN 4c; the small-model test above carries the statistical parity.)  With metric_storage = f32 the matrix every chain reports is that
    estimate rounded to fp32, and the oracle's transitions under THAT matrix are the device's: one 0.9 GB matrix streamed per round."""
    data, variant = cases["2016"]
    nw, md, chains = 30, 6, 4
    f32 = storage == _abi.STORAGE_F32
    h = Handle(data, variant, chains=chains, num_warmup=nw, num_samples=0, save_warmup=1, seed=1843, metric=_abi.METRIC_DENSE, max_depth=md, pooled_metric=1,
               metric_storage=storage)
    h.init(); h.run(nw)
    d = h.draws()
    assert np.isfinite(d).all()
    w = np.concatenate([d[c][4:27, 7:] for c in range(chains)])
    N = len(w)
    want = (N / (N + 5.0)) * np.cov(w.T) + 1e-3 * (5.0 / (N + 5.0)) * np.eye(h.D)
    Mi = h.dense_metric(chains - 1)
    if f32:
        assert np.array_equal(Mi, Mi.astype(np.float32).astype(np.float64)) and np.allclose(Mi, want, rtol=1e-6, atol=1e-7 * np.abs(want).max()) and np.array_equal(Mi, Mi.T)
    else:
        assert np.allclose(Mi, want, rtol=1e-9, atol=1e-12 * np.abs(want).max()) and np.array_equal(Mi, Mi.T)
    assert np.array_equal(h.dense_metric(0), Mi)
    res, solve = h.dense_check(chains - 1, 2)
    assert res < 1e-12 and solve < 1e-9, (res, solve)
    for c in (0, chains - 1):
        _post_window_rows_against_the_oracle(data, variant, h, d, c, 27, 3, md, Mi, c + 1)
    ms, passes, nbytes, rounds = h.dense_timing()
    assert nbytes == passes * (4 if f32 else 8) * h.D * ((h.D + 7) // 8 * 8)
    print(f"2016, pooled dense metric{' (fp32 storage)' if f32 else ''}, {chains} chains: {passes} matrix passes of {nbytes / max(passes, 1) / 1e9:.2f} GB in {ms:.0f} ms = {nbytes / ms / 1e9:.2f} TB/s; {rounds} leaf rounds")
    h.close()


def test_pooled_over_two_handles_equals_one_handle_with_all_chains(cases):
    """pooled_metric = 2 (the window end in two halves, potus_dense_pool_window / _finish): four chains of one posterior as TWO handles of two -- the second on
    `second_device()` where the box has one (peer copies), else on the same GPU -- driven by sampler.run_pooled, which adds the handles' moments (Chan's
    update, what parallel.pool_window_moments does across ranks) before every window end is finished.  Against ONE handle that pools the four chains itself
    (pooled_metric = 1): the same matrix on both handles after every window end, equal to the one handle's to rounding (another order of summation), draws
    before the first window end identical, and the rows after it the oracle's transitions from the device's own state under the device's own matrix.
    With one handle and nothing in between, the two halves ARE pooled_metric = 1: same bytes."""
    from conftest import second_device
    from adaptation_replay import window_schedule
    data, variant = cases["small_full"]
    nw = 60                                                     # 15 % / 10 %: init buffer 9, one window of 45 draws (rows 9 .. 53), terminal buffer 6
    (start, end), = window_schedule(nw, 75, 50, 25)
    kw = dict(num_warmup=nw, num_samples=4, save_warmup=1, seed=1843, metric=_abi.METRIC_DENSE)
    one = Handle(data, variant, chains=4, pooled_metric=1, **kw)
    one.init(); one.run(nw + 4)
    d1, M1 = one.draws(), one.dense_metric(0)
    one.close()
    hs = [Handle(data, variant, chains=2, chain_id_offset=0, pooled_metric=2, **kw),
          Handle(data, variant, chains=2, chain_id_offset=2, pooled_metric=2, device=second_device(), **kw)]
    for h in hs:
        h.init()
    sampler.run_pooled(hs, 20)                                  # (stops and resumes: chunks that do not end at the window end)
    sampler.run_pooled(hs, nw + 4 - 20)
    d2 = np.concatenate([h.draws() for h in hs])
    Ma, Mb = hs[0].dense_metric(1), hs[1].dense_metric(0)
    assert np.array_equal(Ma, Mb) and np.array_equal(Ma, Ma.T)
    assert np.allclose(Ma, M1, rtol=1e-10, atol=1e-13 * np.abs(M1).max()), np.abs(Ma - M1).max()
    assert np.array_equal(d2[:, :end + 1], d1[:, :end + 1])     # up to and including the window's last row: the unit metric
    w = np.concatenate([d2[c][start:end + 1, 7:] for c in range(4)])
    N = len(w)
    want = (N / (N + 5.0)) * np.cov(w.T) + 1e-3 * (5.0 / (N + 5.0)) * np.eye(hs[0].D)
    assert np.allclose(Ma, want, rtol=1e-9, atol=1e-12 * np.abs(want).max())
    for hi, h in enumerate(hs):
        res, solve = h.dense_check(0, 2)
        assert res < 1e-12 and solve < 1e-9, (res, solve)
        for c in range(2):
            _post_window_rows_against_the_oracle(data, variant, h, h.draws(), c, end + 1, 3, 10, Ma, 2 * hi + c + 1)
    with pytest.raises(sampler.PotusError):
        hs[0].pool_finish(10.0)                                 # nothing pending
    for h in hs:
        h.close()
    solo = Handle(data, variant, chains=4, pooled_metric=2, **kw)
    solo.init()
    sampler.run_pooled([solo], nw + 4)
    assert np.array_equal(solo.draws(), d1) and np.array_equal(solo.dense_metric(3), M1)
    with pytest.raises(sampler.PotusError):                     # a pending window end blocks the next run
        t = Handle(data, variant, chains=2, pooled_metric=2, **kw)
        t.init(); t.run(nw); t.run(1)
    solo.close()


def test_pooled_sampler_with_more_right_hand_sides_than_one_launch_holds(cases):
    """Twenty chains on one pooled metric: the first pass of a transition carries 60 right-hand sides (two launches of the product, whole chains at a time), a
    leaf round 40; reproducible bit for bit and chunk-invariant, every chain reports the one matrix, the factor is its factor, and a chain's draws are those it
    has in a handle of eight (the product's order of summation does not depend on the companions) as long as the metrics agree -- i.e. up to the first window end."""
    data, variant = cases["small_nomode"]
    kw = dict(num_warmup=40, num_samples=6, seed=11, metric=_abi.METRIC_DENSE, pooled_metric=1, save_warmup=1, cus_per_chain=1)   # (one gradient kernel for both chain counts)
    a = Handle(data, variant, chains=20, **kw); a.init(); a.run(46); da = a.draws()
    assert np.isfinite(da).all() and np.array_equal(a.dense_metric(19), a.dense_metric(0))
    res, solve = a.dense_check(19, 2)
    assert res < 1e-12 and solve < 1e-9, (res, solve)
    a.close()
    b = Handle(data, variant, chains=20, **kw); b.init(); b.run(17); b.run(29); db = b.draws(); b.close()
    assert np.array_equal(da, db)
    from adaptation_replay import window_schedule
    first_end = window_schedule(40, 75, 50, 25)[0][1]
    c = Handle(data, variant, chains=8, chain_id_offset=5, **kw); c.init(); c.run(first_end + 1); dc = c.draws(); c.close()
    assert np.array_equal(dc[:, :first_end + 1], da[5:13, :first_end + 1])

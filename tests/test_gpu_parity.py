"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the golden
fixtures.  Floating point, so tolerances are stated: log density 1e-11 relative; gradient
1e-10 of its max-norm; early NUTS trajectories 1e-6; posterior summaries in units of MCSE.

Both device paths are covered: cus_per_chain = 1 (one workgroup per chain, potus_model.hpp /
potus_nuts.hpp) and clusters of 8 and 16 workgroups per chain (potus_cluster.hpp)."""
from pathlib import Path

import numpy as np
import pytest

from adaptation_replay import adaptation_replayed_from_the_device_rows as _adaptation_replayed_from_the_device_rows, rows_around, run_through_the_windows, window_schedule
from conftest import GOLD
from oracle_lib import OracleModel
from us_potus_model_amd import Handle, PotusModel, _abi, diagnostics as dg, sampler

pytestmark = pytest.mark.gpu

LP_RTOL, GRAD_RTOL = 1e-11, 1e-10


def _blocks(data, variant):
    layout, _ = _abi.column_layout(data, variant)
    D = _abi.num_params(data, variant)
    return {k: (a - 7, b - 7) for k, (a, b, _) in layout.items() if b - 7 <= D}


CUS = [1, 4, 8, 16]


@pytest.mark.parametrize("cus", CUS)
@pytest.mark.parametrize("name", ["small_full", "small_nomode", "2016", "2012", "2008"])
def test_log_prob_grad_matches_oracle_and_golden(cases, name, cus):
    data, variant = cases[name]
    h = Handle(data, variant, chains=1, cus_per_chain=cus)
    m = OracleModel(data, variant)
    g = np.load(GOLD / f"logprob_{name}.npz")
    rng = np.random.default_rng(2024)
    q = np.vstack([g["q"], rng.uniform(-2, 2, (5, h.D)), 0.2 * rng.standard_normal((4, h.D))])
    lp, grad = h.log_prob_grad(q)
    for i in range(q.shape[0]):
        lpo, go = (g["lp"][i], g["grad"][i]) if i < 3 else m.log_prob_grad(q[i])
        scale = np.abs(go).max()
        err = np.abs(grad[i] - go)
        if err.max() > GRAD_RTOL * scale or abs(lp[i] - lpo) > LP_RTOL * abs(lpo):
            per_block = {k: float(err[a:b].max() / scale) for k, (a, b) in _blocks(data, variant).items()}
            pytest.fail(f"{name} point {i}: lp {lp[i]!r} vs {lpo!r}; grad rel err by block {per_block}")
    h.close()


EDGE_SHAPES = {
    "no_national_polls": dict(S=6, T=24, N_state=70, N_national=0, P=9),
    "no_state_polls": dict(S=6, T=24, N_state=0, N_national=25, P=9),
    "one_state": dict(S=1, T=24, N_state=30, N_national=10, P=4),
    "two_days": dict(S=6, T=2, N_state=20, N_national=5, P=3),
    "one_pollster": dict(S=6, T=24, N_state=70, N_national=25, P=1),
    "mostly_unpolled_days": dict(S=6, T=200, N_state=12, N_national=3, P=3),
    "63_states": dict(S=63, T=30, N_state=200, N_national=20, P=9),
}


@pytest.mark.parametrize("cus", [1, 8, 16])
@pytest.mark.parametrize("variant", ["full", "no_mode_adjustment"])
@pytest.mark.parametrize("shape", list(EDGE_SHAPES))
def test_edge_shapes_match_the_oracle(shape, variant, cus):
    """Empty and ragged inputs: no national / no state polls, a single state, two days (members without days),
    a single pollster, campaigns where most days have no poll, the largest S a wave holds."""
    from us_potus_model_amd import synthetic
    data = synthetic.make(seed=3, variant=variant, **EDGE_SHAPES[shape])
    h = Handle(data, variant, chains=2, num_warmup=10, num_samples=0, save_warmup=1, seed=3, cus_per_chain=cus)
    m = OracleModel(data, variant)
    q = np.random.default_rng(1).uniform(-2, 2, (3, h.D))
    lp, g = h.log_prob_grad(q)
    for i in range(3):
        lpo, go = m.log_prob_grad(q[i])
        assert abs(lp[i] - lpo) <= LP_RTOL * max(1.0, abs(lpo)), (i, lp[i], lpo)
        assert np.abs(g[i] - go).max() <= GRAD_RTOL * np.abs(go).max(), i
    h.init(); h.run(4)
    d = h.draws()
    o = m.default_opts(num_warmup=10, num_samples=0, save_warmup=1, seed=3, fast_grad=1)
    for c in (0, 1):
        ref = m.sample_chain(c + 1, o)[0][:4]
        assert np.array_equal(d[c][:4, 3:6], ref[:, 3:6]), (c, d[c][:4, :7], ref[:, :7])
        assert np.allclose(d[c][:4, 7:], ref[:, 7:], rtol=1e-6, atol=1e-7)
    h.close()


def test_stress_shape_log_prob_grad_and_first_transitions():
    """BASELINE configs[4] sizes (51 states x 600 days x 10 000 polls, D = 41 610; diagonal metric): beyond the
    one-workgroup kernels (T > 256), so the library must pick a cluster by itself and refuse cus_per_chain = 1."""
    from us_potus_model_amd import synthetic
    data = synthetic.stress()
    with pytest.raises(sampler.PotusError, match="T = 600"):
        Handle(data, "full", chains=1, cus_per_chain=1)
    iters = 3
    h = Handle(data, "full", chains=2, num_warmup=iters, num_samples=0, save_warmup=1, seed=5)
    assert h.D == 41610 and h.cus_per_chain in (16, 32)     # chosen by the library (LDS of the poll-heavy members)
    m = OracleModel(data, "full")
    rng = np.random.default_rng(5)
    q = np.vstack([np.zeros((1, h.D)), rng.uniform(-2, 2, (2, h.D)), 0.2 * rng.standard_normal((2, h.D))])
    lp, grad = h.log_prob_grad(q)
    for i in range(q.shape[0]):
        lpo, go = m.log_prob_grad(q[i])
        assert abs(lp[i] - lpo) <= LP_RTOL * abs(lpo), (i, lp[i], lpo)
        assert np.abs(grad[i] - go).max() <= GRAD_RTOL * np.abs(go).max(), i
    h.init(); h.run(iters)
    d = h.draws()
    o = m.default_opts(num_warmup=iters, num_samples=0, save_warmup=1, seed=5, fast_grad=1)
    for c in (0, 1):
        ref = m.sample_chain(c + 1, o)[0]
        assert np.array_equal(d[c][:, 3:6], ref[:, 3:6]), (c, d[c][:, :7], ref[:, :7])   # depth, n_leapfrog, divergent
        assert np.allclose(d[c][:, 7:], ref[:, 7:], rtol=1e-6, atol=1e-7)
    h.close()


def test_stress_shape_with_the_adjoint_on_the_matrix_cores():
    """The same posterior on clusters of 32 (19 days per member: the 4-days-per-wave build): 16.7 polls per day, so the library
    takes the build whose adjoint product runs on the fp64 matrix cores (v_mfma_f64_16x16x4_f64 over the member's
    pseudo-state x day matrix of residual sums, prefix over days on the accumulators) instead of the walk over the polls.  Log
    density, gradient and the first transitions against the oracle; the walk (POTUS_CL_MFMA = 0) gives the same trees."""
    import os
    from us_potus_model_amd import synthetic
    data = synthetic.stress()
    iters = 3
    m = OracleModel(data, "full")
    rng = np.random.default_rng(6)
    q = np.vstack([np.zeros((1, 41610)), rng.uniform(-2, 2, (2, 41610)), 0.2 * rng.standard_normal((1, 41610))])
    ref_lp = [m.log_prob_grad(qi) for qi in q]
    o = m.default_opts(num_warmup=iters, num_samples=0, save_warmup=1, seed=5, fast_grad=1)
    ref = m.sample_chain(1, o)[0]
    old = os.environ.get("POTUS_CL_MFMA")
    try:
        for flag in ("1", "0"):
            os.environ["POTUS_CL_MFMA"] = flag
            h = Handle(data, "full", chains=1, num_warmup=iters, num_samples=0, save_warmup=1, seed=5, cus_per_chain=32)
            lp, grad = h.log_prob_grad(q)
            for i, (lpo, go) in enumerate(ref_lp):
                assert abs(lp[i] - lpo) <= LP_RTOL * abs(lpo), (flag, i, lp[i], lpo)
                assert np.abs(grad[i] - go).max() <= GRAD_RTOL * np.abs(go).max(), (flag, i)
            h.init(); h.run(iters)
            d = h.draws()
            assert np.array_equal(d[0][:, 3:6], ref[:, 3:6]), (flag, d[0][:, :7], ref[:, :7])
            assert np.allclose(d[0][:, 7:], ref[:, 7:], rtol=1e-6, atol=1e-7)
            h.close()
    finally:
        if old is None:
            os.environ.pop("POTUS_CL_MFMA", None)
        else:
            os.environ["POTUS_CL_MFMA"] = old


def test_stress_shape_sixteen_chains_on_one_gpu():
    """BASELINE configs[4] asks for 16 chains per GPU at 51 states x 600 days x 10 000 polls: 16 clusters of 16 compute
    units (the members keep the two 51 x 51 factors as packed triangles, the poll counts as int32 pairs and the AR(1)
    tangents of their own days only, which is what makes 600 days / 706 polls per member fit 160 KB of LDS).  First
    and last chain against the oracle; the rate of the run is printed."""
    from us_potus_model_amd import synthetic
    data = synthetic.stress()
    iters = 3
    h = Handle(data, "full", chains=16, num_warmup=iters, num_samples=0, save_warmup=1, seed=5)
    assert h.cus_per_chain == 16 and h.D == 41610
    h.init(); h.run(iters)
    d = h.draws()
    ms, lf = h.last_run_timing()
    print(f"stress shape, 16 chains x 16 CUs: {lf} leapfrogs in {ms:.1f} ms = {lf / ms * 1e3:.0f} leapfrogs/s, {ms * 1e3 * 16 / lf:.1f} us per leapfrog per chain")
    m = OracleModel(data, "full")
    o = m.default_opts(num_warmup=iters, num_samples=0, save_warmup=1, seed=5, fast_grad=1)
    for c in (0, 15):
        ref = m.sample_chain(c + 1, o)[0]
        assert np.array_equal(d[c][:, 3:6], ref[:, 3:6]), (c, d[c][:, :7], ref[:, :7])
        assert np.allclose(d[c][:, 7:], ref[:, 7:], rtol=1e-6, atol=1e-7)
    h.close()


@pytest.mark.parametrize("cus", [1, 16])
def test_log_prob_grad_is_deterministic_and_batched(cases, cus):
    data, variant = cases["2016"]
    h = Handle(data, variant, chains=1, cus_per_chain=cus)
    q = np.random.default_rng(1).uniform(-2, 2, (40, h.D))
    lp1, g1 = h.log_prob_grad(q)
    lp2, g2 = h.log_prob_grad(q[::-1].copy())
    assert np.array_equal(lp1, lp2[::-1]) and np.array_equal(g1, g2[::-1])   # same bytes, any batch slot
    h.close()


@pytest.mark.parametrize("name,tag", [("2016", 16), ("2012", 17), ("2008", 17), ("small_full", 4)])
def test_fixed_layout_builds_serve_both_variants_and_agree_with_the_dynamic_one(cases, name, tag, monkeypatch):
    """The reference's posteriors on clusters of 16 take the fixed-layout builds of the pass: tag 16 for poll_model_2020.stan (2016), tag 17
    for poll_model_2020_no_mode_adjustment.stan (final_2012.R:558, final_2008.R:562; round 5 -- until then they ran the dynamic build).
    Same arithmetic in the same order as the dynamic build (POTUS_CL_DYNAMIC=1): log density, gradient and the first transitions bit for bit."""
    data, variant = cases[name]
    K = 16 if tag >= 16 else 4
    kw = dict(chains=2, num_warmup=20, num_samples=0, save_warmup=1, seed=1843, cus_per_chain=K)
    q = np.random.default_rng(2).uniform(-2, 2, (3, Handle(data, variant, chains=1, cus_per_chain=1).D))
    out = []
    for dyn in ("0", "1"):
        monkeypatch.setenv("POTUS_CL_DYNAMIC", dyn)
        h = Handle(data, variant, **kw)
        assert h.L.potus_debug_build_tag(h.h) == (tag if dyn == "0" else 4)
        lp, g = h.log_prob_grad(q)
        h.init(); h.run(4)
        out.append((lp, g, h.draws()[:, :4].copy()))
        h.close()
    m = OracleModel(data, variant)
    for i in range(3):
        lpo, go = m.log_prob_grad(q[i])
        assert abs(out[0][0][i] - lpo) <= 1e-11 * abs(lpo) and np.abs(out[0][1][i] - go).max() <= 1e-10 * np.abs(go).max()
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("cus", [1, 4, 16])
def test_negative_scales_are_their_absolute_values(cases, cus):
    """Stan bounds none of the three scales and only uses their squares (stan:50-52), so a negative scale gives the same posterior.  The
    cluster pass folds L_T = aT L_W, L_B = aB L_W with aT, aB = ratios of the scales (signed until round 5, which flipped the
    prior deviation of mu_b_T / polling_bias on clusters only): log density and gradient against the oracle, and against the positive
    scales bit for bit."""
    data, variant = cases["2016" if cus == 16 else "small_full"]
    neg = dict(data, mu_b_T_scale=-data["mu_b_T_scale"], polling_bias_scale=-data["polling_bias_scale"])
    q = np.random.default_rng(5).uniform(-2, 2, (2, Handle(data, variant, chains=1, cus_per_chain=cus).D))
    hp, hn = Handle(data, variant, chains=1, cus_per_chain=cus), Handle(neg, variant, chains=1, cus_per_chain=cus)
    (lp_p, g_p), (lp_n, g_n) = hp.log_prob_grad(q), hn.log_prob_grad(q)
    assert np.array_equal(lp_p, lp_n) and np.array_equal(g_p, g_n)
    m = OracleModel(neg, variant)
    for i in range(2):
        lpo, go = m.log_prob_grad(q[i])
        assert abs(lp_n[i] - lpo) <= 1e-11 * abs(lpo) and np.abs(g_n[i] - go).max() <= 1e-10 * np.abs(go).max()
    hp.close(); hn.close()


@pytest.mark.parametrize("cus", CUS)
@pytest.mark.parametrize("name,iters", [("small_full", 8), ("small_nomode", 8), ("2016", 3), ("2012", 3), ("2008", 3)])
def test_nuts_follows_the_oracle_chain(cases, name, iters, cus):
    """Same Philox streams + same algorithm => the first transitions agree to rounding."""
    data, variant = cases[name]
    h = Handle(data, variant, chains=2, num_warmup=30, num_samples=0, save_warmup=1, seed=1843, cus_per_chain=cus)
    h.init()
    h.run(iters)
    d = h.draws()[:, :iters]
    m = OracleModel(data, variant)
    o = m.default_opts(num_warmup=30, num_samples=0, save_warmup=1, seed=1843, fast_grad=1)
    for c in (0, 1):
        ref = m.sample_chain(c + 1, o)[0][:iters]
        assert np.array_equal(d[c][:, 3:6], ref[:, 3:6]), (name, c, d[c][:, :7], ref[:, :7])   # depth, n_leapfrog, divergent
        assert np.allclose(d[c][:, :3], ref[:, :3], rtol=1e-6, atol=1e-9), (d[c][:, :3], ref[:, :3])
        assert np.allclose(d[c][:, 6], ref[:, 6], rtol=1e-8)
        assert np.allclose(d[c][:, 7:], ref[:, 7:], rtol=1e-6, atol=1e-7)
    h.close()


@pytest.mark.parametrize("cus,twin", [(1, 0), (1, 1), (16, 0), (16, 1)])
def test_adaptation_matches_oracle_through_a_metric_update(cases, cus, twin):
    """150 warm-up iterations of the small model: init buffer, one window (draws 75 .. 99), metric update + init_stepsize, term
    buffer -- every step size, the metric and the transitions around the window end replayed from the device's own rows."""
    data, variant = cases["small_full"]
    nw = 150
    h = Handle(data, variant, chains=2, num_warmup=nw, num_samples=0, save_warmup=1, seed=11, cus_per_chain=cus, twin=twin)
    h.init()
    h.run(nw)
    d = h.draws()
    # against the oracle's own chain while rounding differences have not been amplified: the first transitions are the same
    m = OracleModel(data, variant)
    ref, ad, nl = m.sample_chain(1, m.default_opts(num_warmup=nw, num_samples=0, save_warmup=1, seed=11, fast_grad=1))
    k = 20
    assert np.array_equal(d[0][:k, 3:6], ref[:k, 3:6]) and np.allclose(d[0][:k, 2], ref[:k, 2], rtol=1e-6)
    for c in (0, 1):
        _adaptation_replayed_from_the_device_rows(data, variant, h, c, 11, [(1, 3), (74, 3), (98, 8), (147, 3)])
    h.close()


@pytest.mark.parametrize("cus,twin", [(1, 0), (1, 1), (16, 0), (16, 1)])
@pytest.mark.parametrize("name", ["2016", "2012"])
def test_diag_transitions_after_the_window_match_the_oracle_at_2016_size(cases, name, cus, twin):
    """The headline path, pinned per transition: 40 warm-up iterations of the 2016 / 2012 posteriors (init buffer
    6, one window of 30 draws ending with iteration 35, metric update, init_stepsize, term buffer) on every form of the diagonal
    sampler -- k_run, k_run_twin, k_cl_run<., false>, k_cl_run<., true>.  The three transitions after the window end (and three
    before it) are the oracle's transitions from the device's own draw, step size and diagonal inverse metric; every step size of
    the warm-up is dual averaging's / init_stepsize's from the device's own accept_stat__ column."""
    data, variant = cases[name]
    nw = 40
    h = Handle(data, variant, chains=2, num_warmup=nw, num_samples=2, save_warmup=1, seed=1843, cus_per_chain=cus, twin=twin)
    h.init()
    h.run(nw + 2)
    assert np.isfinite(h.draws()).all()
    _adaptation_replayed_from_the_device_rows(data, variant, h, 1, 1843, [(33, 3), (36, 3), (40, 2)])
    h.close()


@pytest.mark.parametrize("cus,twin", [(1, 0), (1, 1), (16, 0), (16, 1)])
@pytest.mark.parametrize("nw,windows", [(500, [(75, 99), (100, 149), (150, 249), (250, 449)]), (400, [(75, 99), (100, 149), (150, 349)])])
def test_adaptation_replayed_through_every_window_end(cases, nw, windows, cus, twin):
    """The schedules the reference runs.  500 warm-up iterations are final_2016.R:6-11,533-541's: 75 | 25, 50, 100, 200 | 50
    -- four metric updates through the DOUBLING branch of windowed_adaptation::compute_next_window; 400 give 75 | 25, 50, 200 | 50 -- the
    third window STRETCHED to the start of the terminal buffer.  Small model, every form of the diagonal sampler, both chains: every
    step size of the warm-up from the device's own accept_stat__ column, the metric the device holds after EVERY update against the
    regularised variance of the very draws of that window, init_stepsize at every window end, and three transitions on either side
    of every window end (plus the first and last of the warm-up) replayed by the oracle from the device's own rows."""
    data, variant = cases["small_full"]
    assert window_schedule(nw, 75, 50, 25) == windows
    h = Handle(data, variant, chains=2, num_warmup=nw, num_samples=3, save_warmup=1, seed=11, cus_per_chain=cus, twin=twin)
    h.init()
    held = run_through_the_windows(h, nw + 3)
    assert sorted(held) == [e for _, e in windows]
    for c in (0, 1):
        n = _adaptation_replayed_from_the_device_rows(data, variant, h, c, 11, [(1, 3)] + rows_around([e for _, e in windows]) + [(nw - 2, 5)], held)
        assert n == len(windows)
    h.close()


@pytest.mark.parametrize("name,cus,twin", [("2016", 16, 1), ("2008", 16, 0), ("2012", 1, 1)])
def test_two_window_ends_at_2016_size(cases, name, cus, twin):
    """The headline kernels through two metric updates at full size: init_buffer = window = term_buffer = 10 of 60 warm-up iterations
    -> windows of 10 (rows 10 .. 19) and, stretched, 30 (rows 20 .. 49).  Both chains; k_cl_run<16, true> on 2016,
    k_cl_run<17, false> on 2008 (the fixed build of the no-mode variant, round 5), two workgroups per chain on 2012."""
    data, variant = cases[name]
    nw = 60
    assert window_schedule(nw, 10, 10, 10) == [(10, 19), (20, 49)]
    h = Handle(data, variant, chains=2, num_warmup=nw, num_samples=2, save_warmup=1, seed=1843, cus_per_chain=cus, twin=twin,
               init_buffer=10, term_buffer=10, window=10)
    h.init()
    held = run_through_the_windows(h, nw + 2)
    assert sorted(held) == [19, 49] and np.isfinite(h.draws()).all()
    for c in (0, 1):
        assert _adaptation_replayed_from_the_device_rows(data, variant, h, c, 1843, [(18, 2), (20, 2), (48, 2), (50, 2), (60, 2)], held) == 2
    h.close()


@pytest.mark.parametrize("twin", [0, 1])
@pytest.mark.parametrize("name", ["2016", "2012", "small_nomode"])
def test_the_gradient_kept_at_a_window_end_is_the_gradient_of_the_point(cases, name, twin):
    """One workgroup (or two) per chain: at a window end the sampler re-evaluates log density and gradient at the chain's point
    before init_stepsize.  That pass is a separately inlined copy of the model pass (cold_transition_end); in round 3 it was found
    to hand garbage to the S x T block of the gradient on the 2016 posterior (a register spilled under an empty EXEC mask,
    scripts/check_spill_exec.py), which collapsed the step size after every metric update.  Checked directly on the device state:
    the vector kept as the gradient equals potus_log_prob_grad of the vector kept as the point, and the step size that
    init_stepsize finds from it is a sane one."""
    import ctypes
    data, variant = cases[name]
    nw = 40                                            # windows: init buffer 6, one window ending with iteration 35
    h = Handle(data, variant, chains=3, num_warmup=nw, num_samples=0, save_warmup=1, seed=99, cus_per_chain=1, twin=twin)
    h.init(); h.run(35)
    eps_before = np.array(h.adaptation()[0])
    h.run(1)                                           # iteration 35: metric update, gradient refresh, init_stepsize
    lib = h.L
    lib.potus_debug_state.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.potus_debug_state.restype = ctypes.c_int
    sz = np.zeros(3)
    assert lib.potus_debug_state(h.h, 0, sz.ctypes.data_as(ctypes.c_void_p), None)
    st = np.zeros((3 * (1 + twin), int(sz[0]), int(sz[1])))
    assert lib.potus_debug_state(h.h, 1, st.ctypes.data_as(ctypes.c_void_p), None)
    for b in range(st.shape[0]):                       # every chain, both sides
        q, g_kept = st[b, 0, :h.D], st[b, 1, :h.D]     # V_QC, V_GC
        lp, g = h.log_prob_grad(q)
        assert np.array_equal(g[0], g_kept), (b, np.abs(g[0] - g_kept).max())
    eps_after = np.array(h.adaptation()[0])
    assert (np.abs(np.log(eps_after / eps_before)) < np.log(64.0)).all(), (eps_before, eps_after)
    h.close()


@pytest.mark.parametrize("cus,twin", [(16, 0), (16, 1), (8, 0)])
def test_the_gradient_kept_at_a_window_end_in_cluster_mode(cases, cus, twin):
    """The same check on the cluster kernels (their window-end pass is inlined into cl_cold_transition_end / cl_cold_twin_end):
    the chain's vectors are kept in internal order there, so the permutation is recovered from the inverse metric, which
    potus_get_adaptation hands out in Stan's order and whose entries are all different after a metric update."""
    import ctypes
    data, variant = cases["2016"]
    h = Handle(data, variant, chains=2, num_warmup=40, num_samples=0, save_warmup=1, seed=99, cus_per_chain=cus, twin=twin)
    h.init(); h.run(36)                                # iteration 35 ends the first window
    lib = h.L
    lib.potus_debug_state.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.potus_debug_state.restype = ctypes.c_int
    sz = np.zeros(3)
    assert lib.potus_debug_state(h.h, 0, sz.ctypes.data_as(ctypes.c_void_p), None)
    st = np.zeros((2 * (1 + twin), int(sz[0]), int(sz[1])))
    assert lib.potus_debug_state(h.h, 1, st.ctypes.data_as(ctypes.c_void_p), None)
    minv = np.asarray(h.adaptation()[1])
    for b in range(st.shape[0]):
        c = b % 2
        where = {v: i for i, v in enumerate(st[b, 11])}          # V_MINV, internal order (with padding)
        assert len(set(minv[c])) == h.D
        perm = np.array([where[v] for v in minv[c]])             # Stan index -> internal position
        q, g_kept = st[b, 0, perm], st[b, 1, perm]               # V_QC, V_GC
        lp, g = h.log_prob_grad(q)
        assert np.allclose(g[0], g_kept, rtol=1e-12, atol=1e-12), (b, np.abs(g[0] - g_kept).max())
    h.close()


@pytest.mark.parametrize("cus", [1, 8, 16])
@pytest.mark.parametrize("name,nw,total,splits", [("small_full", 150, 104, ([104], [99, 5], [100, 4], [99, 1, 4], [50, 49, 2, 3])),
                                                  ("2016", 60, 58, ([58], [53, 5], [54, 4], [53, 1, 4]))])
def test_launch_boundaries_do_not_change_the_draws_across_a_window_end(cases, name, nw, total, splits, cus):
    """The metric update + init_stepsize at the end of the first adaptation window and the transition after it, inside
    one launch or split over launches in every way, give the same bytes (a once-seen miscompilation of the oversized
    kernel broke this)."""
    data, variant = cases[name]
    kw = dict(chains=1, num_warmup=nw, num_samples=0, save_warmup=1, seed=11, cus_per_chain=cus)
    out = []
    for chunks in splits:
        h = Handle(data, variant, **kw); h.init()
        for n in chunks:
            h.run(n)
        out.append(h.draws()[0][:total].copy()); h.close()
    for d in out[1:]:
        assert np.array_equal(out[0], d)


def test_sample_over_several_devices_equals_one_device(cases, tmp_path):
    """PotusModel.sample(devices=[...]): the chains are dealt to the devices in blocks and run together; extract(),
    as_array() and the CSV files are those of a single-device fit.  The second block goes to device 1 when the box has two GPUs (the
    driver's 8-GPU node: potus_run_many across devices, peer copies), else to device 0 again."""
    data, variant = cases["small_full"]
    kw = dict(seed=5, chains=5, iter_warmup=30, iter_sampling=12, refresh=10)
    one = PotusModel(variant).sample(data, **kw)
    from conftest import second_device
    two = PotusModel(variant).sample(data, devices=[0, second_device()], **kw)       # two GPUs wherever the box has them
    assert two.chains == 5 and len(two._hs) == 2
    for name in ("mu_b", "predicted_score", "lp__", "raw_polling_bias"):
        assert np.array_equal(one.extract(name), two.extract(name)), name
    assert np.array_equal(one.as_array("mu_c"), two.as_array("mu_c"))
    f1, f2 = one.output_files(tmp_path / "a"), two.output_files(tmp_path / "b")
    assert [Path(f).name for f in f1] == [Path(f).name for f in f2] and len(f2) == 5
    for a_, b_ in zip(f1, f2):
        la = [ln for ln in open(a_) if not ln.startswith("#")]
        lb = [ln for ln in open(b_) if not ln.startswith("#")]
        assert la == lb


@pytest.mark.parametrize("cus", [0, 1])
def test_three_backtests_concurrently_give_the_bytes_of_separate_runs(cases, cus):
    """BASELINE configs[3] on one GPU: the 2008, 2012 and 2016 posteriors advance together under potus_run_many
    (one host thread, as R has) and every chain is the chain it would have been alone."""
    from us_potus_model_amd import run_many
    kw = dict(chains=4, num_warmup=12, num_samples=4, seed=77, cus_per_chain=cus)
    alone, together, hs = [], [], []
    t_alone = 0.0
    for name in ("2008", "2012", "2016"):
        data, variant = cases[name]
        h = Handle(data, variant, **kw); h.init(); h.run(7); h.run(9)
        t_alone += h.last_run_timing()[0]
        alone.append(h.draws().copy()); h.close()
    for name in ("2008", "2012", "2016"):
        data, variant = cases[name]
        h = Handle(data, variant, **kw); h.init(); hs.append(h)
    run_many(hs, 7); run_many(hs, 9)
    t_together = max(h.last_run_timing()[0] for h in hs)
    with pytest.raises(sampler.PotusError, match="twice"):
        run_many([hs[0], hs[0]], 1)
    k16 = hs[0].cus_per_chain
    for h in hs:
        together.append(h.draws().copy()); h.close()
    for a, b in zip(alone, together):
        assert np.array_equal(a, b)
    if cus == 0:
        assert k16 == 16                                 # 3 x 4 chains x 16 CUs = 192 of 256: one group
        assert t_together < 0.9 * t_alone, (t_together, t_alone)   # concurrent, not one after the other


@pytest.mark.parametrize("cus", [1, 16])
def test_same_seed_same_bytes_and_chain_ids(cases, cus):
    data, variant = cases["small_full"]
    kw = dict(num_warmup=40, num_samples=20, seed=99, cus_per_chain=cus)
    a = Handle(data, variant, chains=3, **kw); a.init(); a.run(60); da = a.draws(); a.close()
    b = Handle(data, variant, chains=3, **kw); b.init(); b.run(25); b.run(35); db = b.draws(); b.close()
    assert np.array_equal(da, db)                      # deterministic, independent of chunking
    c = Handle(data, variant, chains=1, chain_id_offset=2, **kw); c.init(); c.run(60); dc = c.draws(); c.close()
    assert np.array_equal(da[2], dc[0])                # chain 3 is chain 3 wherever it runs
    assert not np.array_equal(da[0], da[1])


@pytest.mark.parametrize("name", ["small_full", "small_nomode", "2016"])
def test_write_array_matches_oracle(cases, name):
    data, variant = cases[name]
    h = Handle(data, variant, chains=2, num_warmup=6, num_samples=0, save_warmup=1, seed=3)
    h.init(); h.run(6)
    d = h.draws()
    m = OracleModel(data, variant)
    full = h.write_array(0, h.n_cols, 6)               # [iter, chain, ncols]
    assert np.array_equal(full[:, :, :7], np.transpose(d[:, :, :7], (1, 0, 2)))
    for it in (0, 5):
        for c in (0, 1):
            want = m.write_array(d[c, it, 7:])
            assert np.allclose(full[it, c, 7:], want, rtol=1e-11, atol=1e-12), name
    a, b, dims = h.layout["predicted_score"]
    part = h.write_array(a, b, 6)
    assert np.array_equal(part, full[:, :, a:b])
    h.close()


def test_extract_shapes_and_transpose(cases):
    data, variant = cases["small_full"]
    model = PotusModel("scripts/model/poll_model_2020.stan")
    fit = model.sample(data, seed=5, chains=2, iter_warmup=20, iter_sampling=10, refresh=10)
    S, T = int(data["S"]), int(data["T"])
    mu_b = fit.extract("mu_b")
    ps = fit.extract("predicted_score")
    assert mu_b.shape == (20, S, T) and ps.shape == (20, T, S)
    assert np.allclose(ps, 1 / (1 + np.exp(-np.transpose(mu_b, (0, 2, 1)))), rtol=1e-12)   # stan:135-139
    assert fit.extract("mu_e_bias").shape == (20,) and fit.model_name == "poll_model_2020_model"
    sp = fit.sampler_params()
    assert sp["n_leapfrog__"].shape == (2, 10) and (sp["stepsize__"] > 0).all()
    model2 = PotusModel("scripts/model/poll_model_2020_no_mode_adjustment.stan")
    with pytest.raises(KeyError):
        model2.sample(cases["small_nomode"][0], chains=1, iter_warmup=5, iter_sampling=2, refresh=0).extract("mu_m")


def test_stan_csv_round_trip(cases, tmp_path):
    data, variant = cases["small_nomode"]
    h = Handle(data, variant, chains=2, num_warmup=10, num_samples=4, seed=8)
    h.init(); h.run(14)
    files = h.write_stan_csv(tmp_path, "poll")
    full = h.write_array(0, h.n_cols, 4)
    for c, f in enumerate(files):
        lines = open(f).read().splitlines()
        header = [l for l in lines if l and not l.startswith("#")][0].split(",")
        assert header[:7] == list(_abi.SAMPLER_COLS) and len(header) == h.n_cols
        assert header[7] == "raw_mu_b_T.1" and header[-1].startswith("predicted_score.")
        rows = np.array([[float(x) for x in l.split(",")] for l in lines if l and not l.startswith("#") and not l.startswith("lp__")])
        assert rows.shape == (4, h.n_cols)
        assert np.allclose(rows, full[:, c, :], rtol=2e-5, atol=1e-12)       # %.6g text, as CmdStan writes
        assert any(l.startswith("# Step size") for l in lines) and any("Elapsed Time" in l for l in lines)
        k_adapt = next(i for i, l in enumerate(lines) if l.startswith("# Adaptation terminated"))
        k_first = next(i for i, l in enumerate(lines) if l and l[0] in "-0123456789")
        assert k_adapt < k_first                                       # no warm-up rows saved: the block follows the header
    h.close()
    # with save_warmup CmdStan writes the adaptation block when warm-up ends, i.e. after the warm-up rows
    h = Handle(data, variant, chains=1, num_warmup=10, num_samples=4, seed=8, save_warmup=1)
    h.init(); h.run(14)
    lines = open(h.write_stan_csv(tmp_path, "pollw")[0]).read().splitlines()
    rows_before = sum(1 for l in lines[:next(i for i, l in enumerate(lines) if l.startswith("# Adaptation terminated"))] if l and l[0] in "-0123456789")
    assert rows_before == 10 and sum(1 for l in lines if l and l[0] in "-0123456789") == 14
    assert "save_warmup = 1" in "\n".join(lines)
    h.close()


@pytest.mark.parametrize("kind", ["diag", "diag_save_warmup", "dense", "dense_save_warmup", "dense_2016"])
def test_stan_csv_passes_the_strict_reader(cases, tmp_path, kind):
    """potus_write_stan_csv against tests/stan_csv_reader.py -- what rstan::read_stan_csv (final_2016.R:543) and CmdStan's own
    grammar of the adaptation block require, restated as assertions (R is not installed here): comment grammar, header, row
    counts against num_samples / num_warmup / save_warmup, the adaptation block where CmdStan puts it, the inverse metric (one
    numeric line for diag_e, D lines of D values for dense_e; left out WITHOUT its header line above 2048 parameters), the three
    Elapsed Time lines; then extract() through the reader equals write_array through the ABI."""
    from stan_csv_reader import read_stan_csv
    big = kind == "dense_2016"
    data, variant = cases["2016" if big else "small_full"]
    dense, sw = kind.startswith("dense"), kind.endswith("save_warmup")
    nw, ns, chains = (6, 2, 1) if big else (30, 6, 3)
    h = Handle(data, variant, chains=chains, num_warmup=nw, num_samples=ns, seed=8, save_warmup=int(sw), metric=_abi.METRIC_DENSE if dense else _abi.METRIC_DIAG,
               max_depth=4 if big else 10, chain_id_offset=2)
    h.init(); h.run(nw + ns)
    files = h.write_stan_csv(tmp_path, "poll_model_2020")
    fit = read_stan_csv(files)
    assert fit.model_name == "poll_model_2020" and len(fit.chains) == chains and fit.n_kept == ns and fit.warmup2 == (nw if sw else 0)
    assert [c.values["chain_id"] for c in fit.chains] == [3 + c for c in range(chains)]
    assert len(fit.fnames) == h.n_cols and fit.chains[0].values["sampler_t"] == ("NUTS(dense_e)" if dense else "NUTS(diag_e)")
    S, T = int(data["S"]), int(data["T"])
    assert fit.dims["raw_mu_b"] == (S, T) and fit.dims["predicted_score"] == (T, S) and fit.dims["mu_e_bias"] == ()
    eps, minv = h.adaptation()
    n_saved = ns + (nw if sw else 0)
    full = h.write_array(0, h.n_cols, n_saved)                                   # [iter, chain, col]
    for c, ch in enumerate(fit.chains):
        assert np.isclose(ch.stepsize, eps[c], rtol=1e-5)
        if dense and not big:
            assert np.allclose(ch.inv_metric, h.dense_metric(c), rtol=2e-5, atol=1e-12)
        elif dense:
            assert ch.inv_metric is None                                          # 15 098 x 15 098 numbers are not text
        else:
            assert np.allclose(ch.inv_metric, minv[c], rtol=2e-5)
        assert np.allclose(ch.rows, full[:, c, :], rtol=2e-5, atol=1e-12)         # %.6g text, as CmdStan writes
        assert ch.elapsed[0] >= 0 and abs(ch.elapsed[0] + ch.elapsed[1] - ch.elapsed[2]) < 2e-3
    a, b, _ = h.layout["predicted_score"]
    ps = fit.extract("predicted_score")                                          # [draws, T, S], chains merged
    want = np.transpose(full[(nw if sw else 0):, :, a:b], (1, 0, 2)).reshape(chains * ns, S, T).transpose(0, 2, 1)
    assert ps.shape == (chains * ns, T, S) and np.allclose(ps, want, rtol=2e-5, atol=1e-12)
    h.close()


def test_error_paths(cases):
    data, variant = cases["small_full"]
    h = Handle(data, variant, chains=1, num_warmup=5, num_samples=5)
    with pytest.raises(sampler.PotusError, match="potus_init"):
        h.run(1)
    with pytest.raises(sampler.PotusError):
        h.write_array(5, 3, 1)
    h.close()
    with pytest.raises(sampler.PotusError, match="max_depth"):
        Handle(data, variant, chains=1, max_depth=40)
    from us_potus_model_amd import synthetic
    with pytest.raises(sampler.PotusError, match="at least two"):
        Handle(synthetic.make(S=6, T=1, N_state=20, N_national=5, P=3, seed=3), "full", chains=1)
    bad = dict(data); bad["day_state"] = data["day_state"].copy(); bad["day_state"][3] = data["T"] + 1
    with pytest.raises(sampler.PotusError, match="day_state"):
        Handle(bad, variant, chains=1)


_ORACLE_POSTERIOR = {}


@pytest.mark.parametrize("cus", [1, 16])
def test_posterior_parity_small(cases, cus):
    """Statistical parity: pooled means of every unconstrained coordinate within 5 combined MCSE."""
    data, variant = cases["small_full"]
    nw = ns = 400
    h = Handle(data, variant, chains=4, num_warmup=nw, num_samples=ns, seed=1843, cus_per_chain=cus)
    h.init(); h.run(nw + ns)
    d = h.draws()
    x = d[:, :, 7:]
    st, _ = h.chain_status()
    assert st == [0, 0, 0, 0] and d[:, :, 5].mean() < 0.02        # divergent__ among the saved draws
    if "y" not in _ORACLE_POSTERIOR:
        m = OracleModel(data, variant)
        o = m.default_opts(num_warmup=nw, num_samples=ns, seed=4242, fast_grad=1)      # different seed: independent run
        _ORACLE_POSTERIOR["y"] = np.stack([m.sample_chain(c, o)[0][:, 7:] for c in (1, 2, 3, 4)])
    y = _ORACLE_POSTERIOR["y"]
    worst = 0.0
    for j in range(h.D):
        a, b = x[:, :, j], y[:, :, j]
        se = np.hypot(a.std() / np.sqrt(dg.ess_mean(a)), b.std() / np.sqrt(dg.ess_mean(b)))
        worst = max(worst, abs(a.mean() - b.mean()) / se)
        assert dg.rhat(a) < 1.08
    assert worst < 5.0, worst
    h.close()


def test_posterior_2016_against_golden_and_readme(data_2016):
    """BASELINE configs[1]: 2016 backtest, 8 chains, 1000/1000, seed 1843 -- against the oracle's
    committed posterior summary (tests/golden/posterior_2016.npz) and, softly (0.01), against the
    reference's published table (README.md:279-332: national 0.512 / 0.485 / 0.540)."""
    g = np.load(GOLD / "posterior_2016.npz")
    model = PotusModel("full")
    fit = model.sample(data_2016, seed=1843, chains=8, iter_warmup=1000, iter_sampling=1000, refresh=500)
    S, T = 51, 254
    mu_b_T = fit.extract("mu_b")[:, :, T - 1].reshape(8, 1000, S)
    ps_T = fit.extract("predicted_score")[:, T - 1, :].reshape(8, 1000, S)
    from conftest import assert_posterior_within_mcse
    assert_posterior_within_mcse(mu_b_T, ps_T, g)
    nat = ps_T @ np.asarray(data_2016["state_weights"])
    assert abs(nat.mean() - 0.512) < 0.01 and abs(np.quantile(nat, 0.025) - 0.485) < 0.012 and abs(np.quantile(nat, 0.975) - 0.540) < 0.012
    sp = fit.sampler_params()
    assert sp["divergent__"].mean() < 0.01 and 0.6 < sp["accept_stat__"].mean() < 0.97


def test_the_references_scripted_call_against_the_oracles_run_of_it(data_2016):
    """BASELINE configs[0] as the reference scripts it -- final_2016.R:6-11 (n_chains 6, n_warmup 500, n_sampling 500) and :533-541 (seed 1843,
    refresh 50): the same call on the device against the CPU oracle's run of that very call (tests/golden/posterior_2016_scripted.npz,
    `scripts/make_golden.py posterior --call scripted`): means of mu_b[:, T] and predicted_score[T, :] within 5 combined MCSE, interval ends
    within 6 MCSE + 0.004.  (`bench.py --config 0` times this call; until round 6 nothing compared its draws with anything but their own R-hat.)"""
    from conftest import assert_posterior_within_mcse
    g = np.load(GOLD / "posterior_2016_scripted.npz")
    assert list(g["config"]) == [6, 500, 500, 1843]
    fit = PotusModel("full").sample(data_2016, seed=1843, chains=6, iter_warmup=500, iter_sampling=500, refresh=50)
    S, T = 51, 254
    mu_b_T = fit.extract("mu_b")[:, :, T - 1].reshape(6, 500, S)
    ps_T = fit.extract("predicted_score")[:, T - 1, :].reshape(6, 500, S)
    z = assert_posterior_within_mcse(mu_b_T, ps_T, g, rhat_max=1.08)
    sp = fit.sampler_params()
    assert sp["divergent__"].mean() < 0.01 and 0.6 < sp["accept_stat__"].mean() < 0.97, z


@pytest.mark.parametrize("name", ["small_full", "2016"])
def test_posterior_summary_matches_numpy_restatement(cases, name):
    """potus_posterior_summary (device: sort per cell in LDS) against oracle/posterior_summary_ref.py on the same
    draws; quantiles/means to 1e-12, exceedance probabilities exactly."""
    import sys
    sys.path.insert(0, str(GOLD.parent.parent / "oracle"))
    from posterior_summary_ref import posterior_summary
    data, variant = cases[name]
    S, T = int(data["S"]), int(data["T"])
    ns = 150 if name == "small_full" else 40
    h = Handle(data, variant, chains=4, num_warmup=60, num_samples=ns, seed=5)
    h.init(); h.run(60 + ns)
    ev = np.arange(3, 3 + S, dtype=np.float64) * (538.0 / np.arange(3, 3 + S).sum())   # any weights; sums to 538
    got = h.posterior_summary(ev)
    a, b, _ = h.layout["predicted_score"]
    ps = h.write_array(a, b, ns).reshape(ns * 4, S, T).transpose(0, 2, 1)             # [draw, T, S]
    ref = posterior_summary(ps, data["state_weights"], ev)
    for k in ("state", "national", "electoral_votes"):
        assert got[k].shape == ref[k].shape
        assert np.allclose(got[k], ref[k], rtol=1e-12, atol=1e-12), (k, np.abs(got[k] - ref[k]).max())
    assert np.array_equal(got["state"][..., 3], ref["state"][..., 3])
    h.close()

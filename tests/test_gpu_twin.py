"""Twin mode (potus_opts.twin, potus_cluster.hpp): two clusters per chain, one per end of the NUTS trajectory.  The
forward and the backward doublings of a transition are integrated at the same time; the trajectory-level bookkeeping
(accept step, rho, U-turn checks across the trajectory) is taken in doubling order by whichever side built the subtree.
Same algorithm, same RNG streams: the chains must follow the oracle's exactly as the one-cluster sampler does."""
import os

import numpy as np
import pytest

from oracle_lib import OracleModel
from us_potus_model_amd import Handle, PotusModel, sampler

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cus", [8, 16])
@pytest.mark.parametrize("name,iters", [("small_full", 10), ("small_nomode", 8), ("2016", 3), ("2008", 3)])
def test_twin_follows_the_oracle_chain(cases, name, iters, cus):
    data, variant = cases[name]
    h = Handle(data, variant, chains=2, num_warmup=30, num_samples=0, save_warmup=1, seed=1843, cus_per_chain=cus, twin=1)
    assert h.clusters_per_chain == 2 and h.cus_per_chain == cus
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


@pytest.mark.parametrize("max_depth", [1, 3, 6])
def test_twin_with_short_trees(cases, max_depth):
    """Trees cut by max_depth (every transition ends at the depth limit early in warm-up): the combine that reaches the
    limit ends the trajectory, the side that is speculating on a later doubling drops it."""
    data, variant = cases["small_full"]
    iters = 25
    kw = dict(num_warmup=30, num_samples=0, save_warmup=1, seed=5, max_depth=max_depth)
    h = Handle(data, variant, chains=2, cus_per_chain=8, twin=1, **kw)
    h.init(); h.run(iters)
    d = h.draws()[:, :iters]
    m = OracleModel(data, variant)
    o = m.default_opts(fast_grad=1, **kw)
    for c in (0, 1):
        ref = m.sample_chain(c + 1, o)[0][:iters]
        assert d[c][:, 3].max() <= max_depth
        k = 12                                      # in step with the oracle at least this long
        assert np.array_equal(d[c][:k, 3:6], ref[:k, 3:6]), (c, d[c][:k, :7], ref[:k, :7])
        assert np.allclose(d[c][:k, 7:], ref[:k, 7:], rtol=1e-6, atol=1e-7)
    h.close()


def test_twin_on_the_stress_shape():
    """configs[4]'s shape (51 states x 600 days x 10 000 polls, D = 41 610) under the diagonal metric: 8 chains take two
    clusters of 16 each by default; first transitions against the oracle."""
    from us_potus_model_amd import synthetic
    data = synthetic.stress()
    iters = 3
    h = Handle(data, "full", chains=8, num_warmup=iters, num_samples=0, save_warmup=1, seed=5)
    assert h.cus_per_chain == 16 and h.clusters_per_chain == 2 and h.D == 41610
    h.init(); h.run(iters)
    d = h.draws()
    ms, lf = h.last_run_timing()
    print(f"stress shape, 8 chains x 2 x 16 CUs: {lf} leapfrogs in {ms:.1f} ms = {lf / ms * 1e3:.0f} leapfrogs/s, {ms * 1e3 * 8 / lf:.1f} us per leapfrog per chain")
    m = OracleModel(data, "full")
    o = m.default_opts(num_warmup=iters, num_samples=0, save_warmup=1, seed=5, fast_grad=1)
    for c in (0, 7):
        ref = m.sample_chain(c + 1, o)[0]
        assert np.array_equal(d[c][:, 3:6], ref[:, 3:6]), (c, d[c][:, :7], ref[:, :7])
        assert np.allclose(d[c][:, 7:], ref[:, 7:], rtol=1e-6, atol=1e-7)
    h.close()


@pytest.mark.parametrize("name,chains,nw,cus", [("2016", 3, 40, 16), ("small_full", 2, 150, 8), ("2012", 2, 30, 16)])
def test_twin_gives_the_bytes_of_one_cluster(cases, name, chains, nw, cus):
    """Same arithmetic in the same order (the metropolis terms are summed per doubling in both modes): the draws of a whole
    warm-up, adaptation windows and metric updates included, are the same bytes with one cluster per chain and with two."""
    data, variant = cases[name]
    kw = dict(chains=chains, num_warmup=nw, num_samples=5, save_warmup=1, seed=99, cus_per_chain=cus)
    out = []
    for twin in (0, 1):
        h = Handle(data, variant, twin=twin, **kw)
        assert h.clusters_per_chain == 1 + twin
        h.init(); h.run(nw + 5)
        out.append((h.draws().copy(), h.adaptation()))
        print(name, "twin" if twin else "one cluster", "%d leapfrogs in %.1f ms" % tuple(reversed(h.last_run_timing())))
        if twin:
            cnt, rb, rf = h.twin_stats()
            assert cnt == h.total_leapfrogs() and rb + rf >= cnt and max(rb, rf) < cnt      # both ends work, and at the same time
            print(f"  counted {cnt}, run by the backward side {rb}, by the forward side {rf}")
        h.close()
    (a, ada), (b, adb) = out
    assert np.array_equal(a, b), np.argwhere(a != b)[:5]
    assert np.array_equal(ada[0], adb[0]) and np.array_equal(ada[1], adb[1])


@pytest.mark.parametrize("name", ["small_full", "small_nomode"])
def test_twin_many_seeds_give_the_bytes_of_one_cluster(cases, name):
    """Early warm-up is where the odd cases live (divergent leaves, subtrees that fail their own U-turn check, trees of
    every depth up to the limit): many seeds, both samplers, the same bytes."""
    data, variant = cases[name]
    n_div = depths = 0
    for seed in range(1, 11):
        kw = dict(chains=4, num_warmup=40, num_samples=0, save_warmup=1, seed=seed, cus_per_chain=8)
        out = []
        for twin in (0, 1):
            h = Handle(data, variant, twin=twin, **kw)
            h.init(); h.run(14)
            out.append(h.draws()[:, :14].copy())
            h.close()
        a, b = out
        assert np.array_equal(a, b), (seed, np.argwhere(a != b)[:5])
        n_div += int(a[:, :, 5].sum()); depths |= sum(1 << int(v) for v in np.unique(a[:, :, 3]))
    print(f"{name}: {n_div} divergent transitions, tree depths seen: {[d for d in range(12) if depths >> d & 1]}")
    assert n_div > 0 and bin(depths).count("1") >= 4


def test_twin_through_a_metric_update_and_across_launch_boundaries(cases):
    """150 warm-up iterations: init buffer, the first window end (metric update + init_stepsize, run redundantly by both
    sides), and the same bytes however the iterations are split over launches."""
    data, variant = cases["small_full"]
    nw, total = 150, 110
    kw = dict(chains=2, num_warmup=nw, num_samples=0, save_warmup=1, seed=11, cus_per_chain=8, twin=1)
    out = []
    for chunks in ([110], [99, 1, 10], [50, 49, 2, 9]):
        h = Handle(data, variant, **kw); h.init()
        for n in chunks:
            h.run(n)
        out.append(h.draws()[:, :total].copy())
        eps, minv = h.adaptation()
        h.close()
    for d in out[1:]:
        assert np.array_equal(out[0], d)
    d = out[0][0]
    m = OracleModel(data, variant)
    ref, ad, nl = m.sample_chain(1, m.default_opts(num_warmup=nw, num_samples=0, save_warmup=1, seed=11, fast_grad=1))
    assert np.array_equal(d[:20, 3:6], ref[:20, 3:6]) and np.allclose(d[:20, 2], ref[:20, 2], rtol=1e-6)
    w = d[75:100, 7:]
    want = (25 / 30.0) * w.var(axis=0, ddof=1) + 1e-3 * (5 / 30.0)
    assert np.allclose(minv[0], want, rtol=1e-10, atol=0), np.abs(minv[0] / want - 1).max()
    if np.allclose(d[:100, 7:], ref[:100, 7:], rtol=1e-5, atol=1e-6):
        assert np.allclose(minv[0], ad[1:], rtol=1e-4) and np.array_equal(d[100:110, 3:6], ref[100:110, 3:6])


@pytest.mark.parametrize("chains,k", [(9, 14), (10, 12), (11, 11), (12, 10)])
def test_nine_to_twelve_chains_take_two_smaller_clusters(cases, chains, k):
    """The library's choice for 9-12 chains: two clusters of 14 / 12 / 11 / 10 workgroups per chain instead of one of 16; first
    transitions of the first and the last chain against the oracle."""
    data, variant = cases["2016"]
    iters = 3
    h = Handle(data, variant, chains=chains, num_warmup=30, num_samples=0, save_warmup=1, seed=1843)
    assert h.cus_per_chain == k and h.clusters_per_chain == 2
    h.init(); h.run(iters)
    d = h.draws()[:, :iters]
    m = OracleModel(data, variant)
    o = m.default_opts(num_warmup=30, num_samples=0, save_warmup=1, seed=1843, fast_grad=1)
    for c in (0, chains - 1):
        ref = m.sample_chain(c + 1, o)[0][:iters]
        assert np.array_equal(d[c][:, 3:6], ref[:, 3:6]), (c, d[c][:, :7], ref[:, :7])
        assert np.allclose(d[c][:, 7:], ref[:, 7:], rtol=1e-6, atol=1e-7)
    h.close()


def test_twin_is_the_default_when_it_fits_and_only_then(cases):
    data, variant = cases["small_full"]
    h = Handle(data, variant, chains=8, num_warmup=10, num_samples=0)
    assert h.cus_per_chain == 16 and h.clusters_per_chain == 2
    h.close()
    h = Handle(data, variant, chains=16, num_warmup=10, num_samples=0)
    assert h.cus_per_chain == 16 and h.clusters_per_chain == 1
    h.close()
    h = Handle(data, variant, chains=2, num_warmup=10, num_samples=0, cus_per_chain=16)       # an explicit size: one cluster unless asked
    assert h.clusters_per_chain == 1
    h.close()
    with pytest.raises(sampler.PotusError):
        Handle(data, variant, chains=16, num_warmup=10, num_samples=0, cus_per_chain=16, twin=1)
    with pytest.raises(sampler.PotusError):
        Handle(data, variant, chains=2, num_warmup=30, num_samples=0, metric=1, cus_per_chain=8, twin=1)


def test_twin_posterior_parity(cases):
    """A full small run in twin mode against an independent run of the oracle: pooled means of every unconstrained
    coordinate within 5 combined MCSE (as test_posterior_parity_small)."""
    from us_potus_model_amd import diagnostics as dg
    data, variant = cases["small_full"]
    nw = ns = 400
    h = Handle(data, variant, chains=4, num_warmup=nw, num_samples=ns, seed=1843, cus_per_chain=8, twin=1)
    assert h.clusters_per_chain == 2
    h.init(); h.run(nw + ns)
    d = h.draws()
    x = d[:, :, 7:]
    st, _ = h.chain_status()
    assert st == [0, 0, 0, 0] and d[:, :, 5].mean() < 0.02 and 0.7 < d[:, :, 1].mean() < 0.95
    m = OracleModel(data, variant)
    o = m.default_opts(num_warmup=nw, num_samples=ns, seed=4242, fast_grad=1)
    y = np.stack([m.sample_chain(c, o)[0][:, 7:] for c in (1, 2, 3, 4)])
    worst = 0.0
    for j in range(h.D):
        a, b = x[:, :, j], y[:, :, j]
        se = np.hypot(a.std() / np.sqrt(dg.ess_mean(a)), b.std() / np.sqrt(dg.ess_mean(b)))
        worst = max(worst, abs(a.mean() - b.mean()) / se)
        assert dg.rhat(a) < 1.08
    assert worst < 5.0, worst
    h.close()


def test_twin_watchdog(cases):
    """A member of one side that never shows up: both sides give up, potus_run reports the watchdog error."""
    data, variant = cases["small_full"]
    os.environ["POTUS_DEBUG_DROP_MEMBER"] = "4"
    try:
        h = Handle(data, variant, chains=2, num_warmup=10, num_samples=0, cus_per_chain=8, twin=1, seed=3)
    finally:
        del os.environ["POTUS_DEBUG_DROP_MEMBER"]
    h.init()
    with pytest.raises(sampler.PotusError, match="error 8"):
        h.run(3)
    h.close()
    g = Handle(data, variant, chains=2, num_warmup=10, num_samples=0, cus_per_chain=8, twin=1, seed=3)
    g.init(); g.run(3)
    assert g.total_leapfrogs() > 0
    g.close()


def _xcd_local(h):
    import ctypes as C
    L = h.L
    L.potus_debug_xcd_local.argtypes = [C.c_int, C.POINTER(C.c_int)]
    L.potus_debug_xcd_local.restype = C.c_int
    out = (C.c_int * 64)()
    n = L.potus_debug_xcd_local(h.h, out)
    return [int(out[i]) for i in range(max(n, 0))]


@pytest.mark.parametrize("chains,twin", [(8, 1), (8, 0), (4, 1), (6, 1)])
def test_exchange_stores_plain_inside_an_xcd_and_write_through_across_give_the_same_bytes(cases, chains, twin, monkeypatch):
    """Round 6: a launch finds out whether every member of a cluster runs on one XCD (cl_find_local: the hardware XCC ids, all-reduced once per launch) and
    then publishes its exchange words with PLAIN stores, which only that XCD's L2 sees -- 552 instead of 1 304 cycles one way (scripts/micro/pingpong.hip).
    With a multiple of eight clusters in the launch k_cl_run deals the blocks of one XCD residue to whole clusters, so 8 chains (one or two clusters each) and
    4 chains x 2 clusters are XCD-local; 6 chains x 2 clusters are not and keep the write-through stores.  Either way the draws are the bytes of the run with the
    write-through path forced (POTUS_DEBUG_DROP_MEMBER=-1)."""
    data, variant = cases["2016"]
    kw = dict(chains=chains, num_warmup=12, num_samples=4, seed=1843, cus_per_chain=16, twin=twin)
    h = Handle(data, variant, **kw)
    h.init(); h.run(16)
    a, loc = h.draws(), _xcd_local(h)
    h.close()
    assert len(loc) == chains * (2 if twin else 1)
    assert all(v == (0 if chains == 6 else 1) for v in loc), loc
    monkeypatch.setenv("POTUS_DEBUG_DROP_MEMBER", "-1")
    g = Handle(data, variant, **kw)
    monkeypatch.delenv("POTUS_DEBUG_DROP_MEMBER")
    g.init(); g.run(16)
    b, locb = g.draws(), _xcd_local(g)
    g.close()
    assert all(v == 0 for v in locb), locb
    assert np.array_equal(a, b) and np.isfinite(a).all()

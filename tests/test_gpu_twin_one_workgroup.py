"""Two workgroups per chain, one per end of the NUTS trajectory (potus_nuts_twin.hpp): the form the library picks for 65-128
chains on 256 compute units, where clusters no longer fit and one workgroup per chain leaves half of the chip idle.  Same
protocol as the two-cluster form (test_gpu_twin.py), on top of the one-workgroup kernels: same algorithm, same RNG streams,
the chains follow the oracle's and the draws are the bytes of the one-workgroup sampler."""
import numpy as np
import pytest

from oracle_lib import OracleModel
from us_potus_model_amd import Handle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,iters", [("small_full", 10), ("small_nomode", 8), ("2016", 3), ("2012", 3)])
def test_two_workgroups_follow_the_oracle_chain(cases, name, iters):
    data, variant = cases[name]
    h = Handle(data, variant, chains=2, num_warmup=30, num_samples=0, save_warmup=1, seed=1843, cus_per_chain=1, twin=1)
    assert h.clusters_per_chain == 2 and h.cus_per_chain == 1
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


@pytest.mark.parametrize("name,chains,nw", [("2016", 3, 40), ("small_full", 2, 150), ("2008", 2, 30)])
def test_two_workgroups_give_the_bytes_of_one(cases, name, chains, nw):
    """The draws of a whole warm-up, adaptation windows and metric updates included, are the same bytes with one workgroup
    per chain and with two (the metropolis terms are summed per doubling in both)."""
    data, variant = cases[name]
    kw = dict(chains=chains, num_warmup=nw, num_samples=5, save_warmup=1, seed=99, cus_per_chain=1)
    out = []
    for twin in (0, 1):
        h = Handle(data, variant, twin=twin, **kw)
        assert h.clusters_per_chain == 1 + twin
        h.init(); h.run(nw + 5)
        out.append((h.draws().copy(), h.adaptation()))
        print(name, "two workgroups" if twin else "one workgroup", "%d leapfrogs in %.1f ms" % tuple(reversed(h.last_run_timing())))
        if twin:
            cnt, rb, rf = h.twin_stats()
            assert cnt == h.total_leapfrogs() and rb + rf >= cnt and max(rb, rf) < cnt      # both ends work, and at the same time
            print(f"  counted {cnt}, run by the backward side {rb}, by the forward side {rf}")
        h.close()
    (a, ada), (b, adb) = out
    assert np.array_equal(a, b), np.argwhere(a != b)[:5]
    assert np.array_equal(ada[0], adb[0]) and np.array_equal(ada[1], adb[1])


@pytest.mark.parametrize("name", ["small_full", "small_nomode"])
def test_two_workgroups_many_seeds(cases, name):
    """Early warm-up is where the odd cases live (divergent leaves, subtrees that fail their own U-turn check, trees of
    every depth up to the limit): many seeds, both samplers, the same bytes; short trees (max_depth) among them."""
    data, variant = cases[name]
    n_div = depths = 0
    for seed in range(1, 9):
        kw = dict(chains=4, num_warmup=40, num_samples=0, save_warmup=1, seed=seed, cus_per_chain=1, max_depth=(3 if seed == 8 else 10))
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


def test_two_workgroups_across_launch_boundaries(cases):
    data, variant = cases["small_full"]
    kw = dict(chains=2, num_warmup=150, num_samples=0, save_warmup=1, seed=11, cus_per_chain=1, twin=1)
    out = []
    for chunks in ([110], [99, 1, 10]):
        h = Handle(data, variant, **kw); h.init()
        for n in chunks:
            h.run(n)
        out.append(h.draws()[:, :110].copy())
        h.close()
    assert np.array_equal(out[0], out[1])


def test_the_library_picks_two_workgroups_for_65_to_128_chains(cases):
    data, variant = cases["2016"]
    for chains, sides in ((64, 1), (65, 2), (128, 2), (129, 1)):
        h = Handle(data, variant, chains=chains, num_warmup=10, num_samples=0)
        assert h.clusters_per_chain == sides and h.cus_per_chain == (4 if chains == 64 else 1), (chains, h.cus_per_chain, h.clusters_per_chain)
        h.close()
    h = Handle(data, variant, chains=100, num_warmup=10, num_samples=0, cus_per_chain=1)      # an explicit size: one workgroup unless asked
    assert h.clusters_per_chain == 1
    h.close()
    # 128 chains, all 256 compute units: first transitions of the first and the last chain against the oracle
    iters = 2
    h = Handle(data, variant, chains=128, num_warmup=30, num_samples=0, save_warmup=1, seed=1843)
    h.init(); h.run(iters)
    d = h.draws()[:, :iters]
    ms, lf = h.last_run_timing()
    print(f"128 chains x 2 workgroups: {lf} leapfrogs in {ms:.1f} ms = {lf / ms * 1e3:.0f} leapfrogs/s")
    m = OracleModel(data, variant)
    o = m.default_opts(num_warmup=30, num_samples=0, save_warmup=1, seed=1843, fast_grad=1)
    for c in (0, 127):
        ref = m.sample_chain(c + 1, o)[0][:iters]
        assert np.array_equal(d[c][:, 3:6], ref[:, 3:6]), (c, d[c][:, :7], ref[:, :7])
        assert np.allclose(d[c][:, 7:], ref[:, 7:], rtol=1e-6, atol=1e-7)
    h.close()


def test_two_workgroup_handles_run_beside_a_cluster_handle(cases):
    """potus_run_many: a handle with two workgroups per chain holds its 2 x chains workgroups for the whole launch, as a cluster
    does -- it is grouped by the compute units it needs (here 70 x 2 workgroups + 2 clusters of 16 fit one device together)
    and gives the bytes of a run on its own."""
    from us_potus_model_amd import sampler
    data, variant = cases["small_full"]
    kw1 = dict(chains=70, num_warmup=20, num_samples=4, seed=31, save_warmup=1)
    kw2 = dict(chains=2, num_warmup=20, num_samples=4, seed=32, save_warmup=1, cus_per_chain=16, twin=0)
    alone = []
    for kw in (kw1, kw2):
        h = Handle(data, variant, **kw); h.init(); h.run(24)
        alone.append(h.draws().copy()); h.close()
    a, b = Handle(data, variant, **kw1), Handle(data, variant, **kw2)
    assert a.cus_per_chain == 1 and a.clusters_per_chain == 2 and b.cus_per_chain == 16
    a.init(); b.init()
    for n in (10, 14):
        sampler.run_many([a, b], n)
    assert np.array_equal(a.draws(), alone[0]) and np.array_equal(b.draws(), alone[1])
    a.close(); b.close()


@pytest.mark.parametrize("side", [0, 1])
def test_two_workgroups_watchdog(cases, side):
    """A side that never shows up (the test hook of the cluster sampler, here per side): the other side gives up waiting for its
    combines after a few seconds, potus_run reports the watchdog error instead of hanging, and the device is usable afterwards."""
    import os
    from us_potus_model_amd import sampler
    data, variant = cases["small_full"]
    os.environ["POTUS_DEBUG_DROP_MEMBER"] = str(side + 1)
    try:
        h = Handle(data, variant, chains=2, num_warmup=10, num_samples=0, cus_per_chain=1, twin=1, seed=3)
    finally:
        del os.environ["POTUS_DEBUG_DROP_MEMBER"]
    h.init()
    with pytest.raises(sampler.PotusError, match="error 8"):
        h.run(3)
    h.close()
    g = Handle(data, variant, chains=2, num_warmup=10, num_samples=0, cus_per_chain=1, twin=1, seed=3)
    g.init(); g.run(3)
    assert g.total_leapfrogs() > 0
    g.close()

import os, sys, subprocess, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
if len(sys.argv) > 1:
    sys.path.insert(0, str(ROOT))
    from us_potus_model_amd import Handle, dataprep
    from us_potus_model_amd import synthetic
    data = synthetic.small("full")
    h = Handle(data, "full", chains=2, num_warmup=150, num_samples=0, seed=11, save_warmup=1, cus_per_chain=16)
    h.init(); h.run(150)
    np.save(sys.argv[1], h.draws())
else:
    for tag, lib in (("a", "libpotus_hmc_step1.so"), ("b", os.environ.get("LIBB", "libpotus_hmc.so"))):
        env = dict(os.environ, POTUS_LIB=str(ROOT / "us_potus_model_amd" / lib))
        subprocess.run([sys.executable, __file__, f"/tmp/cmp_{tag}.npy"], env=env, check=True)
    a, b = np.load("/tmp/cmp_a.npy"), np.load("/tmp/cmp_b.npy")
    same = [np.array_equal(a[:, i], b[:, i]) for i in range(a.shape[1])]
    first = same.index(False) if False in same else -1
    print("first differing iteration:", first)
    if first >= 0:
        np.set_printoptions(linewidth=200, precision=6)
        for c in (0, 1):
            for i in range(max(first - 1, 0), min(first + 3, a.shape[1])):
                print(f"chain {c} iter {i} a:", a[c, i, :7])
                print(f"chain {c} iter {i} b:", b[c, i, :7])

"""Debug aid (GPU box): the dense-metric sampler next to the diagonal one on the 2016 posterior, first transitions."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from us_potus_model_amd import Handle, _abi, dataprep, sampler  # noqa: E402

L = sampler.load_library()
DP = C.POINTER(C.c_double)
L.potus_dense_matvec_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, DP, C.POINTER(C.c_longlong)]
for D, nrhs in ((15098, 2), (15098, 3), (700, 3)):
    M = np.eye(D)[None].copy()
    x = np.random.default_rng(1).standard_normal((1, nrhs, D))
    y, dot, ms, nb = np.zeros((1, nrhs, D)), np.zeros(1), C.c_double(), C.c_longlong()
    rc = L.potus_dense_matvec_probe(0, 1, D, nrhs, M.ctypes.data, x.ctypes.data, y.ctypes.data, dot.ctypes.data, 1, C.byref(ms), C.byref(nb))
    print("identity probe", D, nrhs, rc, "max |y - x|", np.abs(y - x).max(), "dot", dot[0], (x[0, 0] ** 2).sum(), flush=True)
data = dataprep.load_npz(ROOT / "tests" / "golden" / "data_2016.npz")["data"]
for cus in (1, 16):
    for metric in (_abi.METRIC_DIAG, _abi.METRIC_DENSE):
        h = Handle(data, "full", chains=2, num_warmup=5, num_samples=0, save_warmup=1, seed=1843, metric=metric, cus_per_chain=cus)
        h.init(); h.run(5)
        d = h.draws()
        print("cus", cus, "metric", metric, "\n", np.array2string(d[0][:, :7], precision=6, max_line_width=200), flush=True)
        h.close()

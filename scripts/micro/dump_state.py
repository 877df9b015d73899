"""Dump the whole device state after n iterations (development aid): python dump_state.py out.npz n_iter"""
import sys, ctypes, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from us_potus_model_amd import Handle, synthetic
data = synthetic.small("full")
h = Handle(data, "full", chains=1, num_warmup=150, num_samples=0, seed=11, save_warmup=1, cus_per_chain=16)
h.init()
lib = h.L
lib.potus_debug_state.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
lib.potus_debug_state.restype = ctypes.c_int
out = {}
for n in [int(a) for a in sys.argv[2:]]:
    h.run(n)
    sz = np.zeros(3)
    lib.potus_debug_state(h.h, 0, sz.ctypes.data_as(ctypes.c_void_p), None)
    st = np.zeros((int(sz[0]), int(sz[1])))
    sc = np.zeros(int(sz[2]), dtype=np.uint8)
    lib.potus_debug_state(h.h, 1, st.ctypes.data_as(ctypes.c_void_p), sc.ctypes.data_as(ctypes.c_void_p))
    k = "it%d" % sum(int(a) for a in sys.argv[2:][: len(out) // 2 + 1])
    out[k + "_state"] = st; out[k + "_scal"] = sc
np.savez(sys.argv[1], **out)

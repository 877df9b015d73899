import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from us_potus_model_amd import Handle, dataprep
data = dataprep.load_npz(ROOT / "tests" / "golden" / "data_2016.npz")["data"]
import os
NW = int(os.environ.get("NW", "1000")); NS = int(os.environ.get("NS", "1000"))
h = Handle(data, "full", chains=8, num_warmup=NW, num_samples=NS, seed=1843, save_warmup=int(os.environ.get("SW", "0")))
h.init()
import ctypes as C, numpy as np
from us_potus_model_amd import sampler
L = sampler.load_library()
names = {0: "A", 1: "B", 2: "carry", 3: "C", 4: "D", 5: "E", 6: "E2", 7: "X2", 18: "F", 10: "scalar", 11: "merge", 12: "copyq", 13: "pnear", 9: "begin", 15: "end",
         20: "ar:shuffle", 21: "ar:drain", 22: "ar:bar1", 23: "ar:payload", 24: "ar:poll", 26: "ar:gather"}
def prof():
    if not hasattr(L, "potus_debug_profile"):
        return None
    out = np.zeros((8 * h.cus_per_chain, 64))
    L.potus_debug_profile.argtypes = [C.c_int, C.POINTER(C.c_double)]
    L.potus_debug_profile(h.h, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out[0].copy()
prev = prof()
for i in range(20):
    h.run(100)
    ms, lf = h.last_run_timing()
    print(f"chunk {i:2d}: {lf:7d} leapfrogs {ms:8.1f} ms -> {ms*1e3*8/lf:6.2f} us/leapfrog/chain", flush=True)
    if i in (9, 19) and prev is not None:
        cur = prof()
        d = cur - prev
        prev = cur
        print("  phase", "warmup" if i == 9 else "sampling", "leaves", d[16], "cycles per leaf:", {nm: int(d[k] / max(d[16], 1)) for k, nm in names.items()},
              "sum", int(sum(d[k] for k in names) / max(d[16], 1)))

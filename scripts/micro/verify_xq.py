import sys, ctypes as C, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from us_potus_model_amd import Handle, synthetic, sampler
data = synthetic.small("full")
h = Handle(data, "full", chains=1, num_warmup=150, num_samples=0, seed=11, save_warmup=1, cus_per_chain=16)
h.init(); h.run(150)
L = sampler.load_library()
out = np.zeros((16, 64))
L.potus_debug_profile.argtypes = [C.c_int, C.POINTER(C.c_double)]
L.potus_debug_profile(h.h, out.ctypes.data_as(C.POINTER(C.c_double)))
for m in range(16):
    if out[m][56] != 0:
        print("member", m, "first mismatch: rep index", int(out[m][56] - 1), "state", out[m][57], "sent", out[m][58], "x1e", out[m][59])
print("S", data["S"], "T", data["T"], "P", data["P"], "NR", 2 * data["S"] + data["P"] + data["M"] + data["Pop"] + 2)
d = h.draws()[0]
print(d[95:115, 4]); print(d[95:115, 5]); print("eps", h.adaptation()[0])

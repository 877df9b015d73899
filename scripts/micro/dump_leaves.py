import sys, ctypes as C, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from us_potus_model_amd import Handle, synthetic, sampler
data = synthetic.small("full")
h = Handle(data, "full", chains=1, num_warmup=150, num_samples=0, seed=11, save_warmup=1, cus_per_chain=16)
h.init(); h.run(101)
L = sampler.load_library()
out = np.zeros((16, 64))
L.potus_debug_profile.argtypes = [C.c_int, C.POINTER(C.c_double)]
L.potus_debug_profile(h.h, out.ctypes.data_as(C.POINTER(C.c_double)))
np.set_printoptions(linewidth=200, precision=6)
for mm in range(16):
    print("member", mm, "H0", out[mm][35], "kin0, lp0, eps, epoch", out[mm][56:60], "leaf0", out[mm][44:46], "part0, part1, lp0b", out[mm][46:49])
print(h.draws()[0, 100, :7])

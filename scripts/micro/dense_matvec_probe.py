"""Rate of the dense-metric mat-vec building block (potus_dense.hpp) at the sizes of the 2016 posterior and of
BASELINE configs[4]: matrices generated on the device, HIP-event time per launch, bytes = 8 D^2 per chain."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from us_potus_model_amd import sampler  # noqa: E402

L = sampler.load_library()
L.potus_dense_matvec_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
for chains, D in ((1, 15098), (8, 15098), (1, 41610), (8, 41610)):
    p = np.random.default_rng(1).standard_normal((chains, D))
    y = np.zeros((chains, D))
    ms = C.c_double()
    rc = L.potus_dense_matvec_probe(0, chains, D, None, p.ctypes.data, y.ctypes.data, 5, C.byref(ms))
    if rc:
        buf = C.create_string_buffer(512); L.potus_last_error(buf, 512)
        print(f"chains={chains} D={D}: error {rc}: {buf.value.decode()}")
        continue
    # spot check of a few rows against the generator's formula
    ok = True
    for c in (0, chains - 1):
        for i in (0, D // 3, D - 1):
            j = np.arange(D)
            row = np.exp(-np.abs(i - j) / 50.0) * (1.0 + 0.1 * c) + (j == i)
            ok &= abs(row @ p[c] - y[c, i]) <= 1e-10 * max(1.0, abs(y[c, i]))
    gb = chains * D * D * 8 / 1e9
    print(f"chains={chains} D={D}: {ms.value:.3f} ms per mat-vec, {gb / (ms.value * 1e-3):.0f} GB/s = {gb / (ms.value * 1e-3) / 8000:.2f} of 8 TB/s; rows check {'ok' if ok else 'MISMATCH'}")

L.potus_dense_welford_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
for chains, D in ((8, 15098), (4, 41610)):
    a = np.random.default_rng(2).standard_normal((chains, D)); d = np.random.default_rng(3).standard_normal((chains, D))
    ms = C.c_double()
    rc = L.potus_dense_welford_probe(0, chains, D, a.ctypes.data, d.ctypes.data, None, 5, C.byref(ms))
    gb = chains * D * D * 16 / 1e9
    print(f"welford update chains={chains} D={D}: rc={rc}, {ms.value:.3f} ms, {gb / (ms.value * 1e-3):.0f} GB/s read+write = {gb / (ms.value * 1e-3) / 8000:.2f} of 8 TB/s")

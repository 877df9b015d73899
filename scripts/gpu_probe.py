#!/usr/bin/env python3
"""Development probe (GPU box): NUTS timings at several chain counts; with POTUS_LIB pointing at
the -DPOTUS_PROF build also the in-kernel cycle breakdown per phase."""
import ctypes as C
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from us_potus_model_amd import Handle, dataprep, sampler  # noqa: E402

YEAR = os.environ.get("POTUS_YEAR", "2016")             # 2012 / 2008: the no_mode_adjustment posteriors (final_2012.R:558, final_2008.R:562)
VARIANT = "full" if YEAR == "2016" else "no_mode_adjustment"
data = dataprep.load_npz(ROOT / "tests" / "golden" / f"data_{YEAR}.npz")["data"]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
chain_counts = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 8, 64, 256]
KCL = int(os.environ.get("POTUS_K", "1"))
TWIN = int(os.environ.get("POTUS_TWIN", "0"))          # 1: two clusters per chain (profile rows: side 0's members, then side 1's)
NAMESK = {0: "A load+suffix", 1: "B matvec/AR + X1", 2: "(carry: in B)", 3: "C polls", 4: "D gathers/seg1", 5: "E prefix/seg2", 6: "E2 payload parts",
          7: "X2 publish+wait", 18: "F finish grads", 19: "X3 allreduce", 10: "leaf scalar", 11: "merge", 12: "copy q", 13: "p_near",
          9: "begin (all transitions)", 15: "end (all transitions)", 20: "ar:shuffle", 21: "ar:drain", 22: "ar:barrier1", 23: "ar:payload st",
          24: "ar:signal+poll", 25: "ar:barrier2", 26: "ar:gather"}
NAMES1 = {0: "A load+suffix", 1: "B carry/C/AR1", 2: "C polls", 3: "D gathers", 4: "E prefix/seg2", 5: "F dZ/adjoint", 6: "G reduce",
         8: "momentum", 9: "init copy", 10: "leaf scalar", 11: "merge", 12: "copy q", 13: "p_near", 14: "adapt", 15: "save"}
NAMES = NAMES1 if KCL == 1 else NAMESK
for chains in chain_counts:
    h = Handle(data, VARIANT, chains=chains, num_warmup=iters, num_samples=0, seed=1843, cus_per_chain=KCL, twin=TWIN)
    h.init()
    ms_tot, lf_tot = 0.0, 0
    for _ in range(3):
        h.run(iters // 3)
        ms, lf = h.last_run_timing()
        ms_tot += ms
        lf_tot += lf
    print(f"NUTS chains={chains}: {lf_tot} leapfrogs in {ms_tot:.1f} ms -> {lf_tot/ms_tot*1e3:.0f} leapfrogs/s, "
          f"{ms_tot*1e3*chains/lf_tot:.2f} us/leapfrog/chain", flush=True)
    L = sampler.load_library()
    if hasattr(L, "potus_debug_profile"):
        out = np.zeros((chains * KCL * (2 if TWIN else 1), 64))
        L.potus_debug_profile.argtypes = [C.c_int, C.POINTER(C.c_double)]
        if L.potus_debug_profile(h.h, out.ctypes.data_as(C.POINTER(C.c_double))):
            p = out[0]
            leaves, merges = p[16], p[17]
            tot = sum(p[k] for k in NAMES)
            print(f"  chain 0: leaves {leaves:.0f} merges {merges:.0f}; cycles per leaf by phase (clock64 ticks):")
            for k, nm in NAMES.items():
                print(f"    {nm:16s} {p[k]/max(leaves,1):10.0f}  ({100*p[k]/tot:4.1f}%)")
            sub = {56: "C: thread 0 after the 51-term dot", 57: "C: ... its noise element arrived", 58: "C: ... binomial term done", 59: "C: ... epilogue of the element issued"}
            if KCL > 1:
                print("  per member (cycles per leaf):  " + " ".join(f"{nm[:9]:>9s}" for nm in NAMES.values()))
                for mm in range(KCL):
                    print(f"    m={mm:2d}                        " + " ".join(f"{out[mm][k]/max(leaves,1):9.0f}" for k in NAMES))
            if KCL > 1:
                names = ["start", "A done (bar1)", "B done/X1 in (bar2)", "C written (bar3)", "polls done", "D done", "E done", "E2 done (X2 out)",
                         "X2 prefix in (w0)", "F done", "sweeps done (bar)", "X3 in"]
                st = np.array([out[mm][40:52] for mm in range(KCL)])
                t0 = st[:, 0].min()
                print("  timeline of leaf 3000 (us since the first member entered the pass), members as columns:")
                for k, nm in enumerate(names):
                    print(f"    {nm:22s}" + " ".join(f"{(st[mm, k] - t0) / 100.0:6.2f}" for mm in range(KCL)))
                for ph, nm in ((0, {"1": "B", "3": "C", "4": "D", "5": "E", "6": "E2", "7": "F to its first barrier"}[os.environ.get("POTUS_PROF_WAVES", "1")] + " per wave"),):
                    for mm in (0, KCL - 4, KCL - 1):
                        print(f"  {nm} (cycles per leaf, member {mm}): ", [int(out[mm][32 + 8 * ph + w] / max(leaves, 1)) for w in range(8)])
            if KCL > 1:
                for mm in (0, KCL // 2, KCL - 1):
                    print(f"  phase B, wave 0 of member {mm} (cycles per leaf since barrier 1): own part {out[mm][27]/leaves:.0f}, X1 in {out[mm][28]/leaves:.0f}, "
                          f"; phase C, wave 6: totals of the previous leaf in {out[mm][29]/leaves:.0f}, verdicts {out[mm][30]/leaves:.0f}")
            if KCL > 1:
                for mm in (0, KCL // 2, KCL - 1):
                    print(f"  phase F, thread 0 of member {mm} (cycles per leaf since its X2 wait ended): own slots done {out[mm][52]/leaves:.0f}, barrier {out[mm][53]/leaves:.0f}, "
                          f"day block stored {out[mm][54]/leaves:.0f}, totals of the next position summed {out[mm][55]/leaves:.0f}")
            PK = os.environ.get("POTUS_PROF_KIND", "")          # which optional counters the build carries in slots 56-61: "fetch" or "c"
            if KCL > 1 and PK == "fetch":
                for mm in (0, KCL - 1):
                    print(f"  previous leaf's totals, member {mm}: first fetch of 16 words {out[mm][56]/leaves:.0f} cycles, needed a re-fetch in {100*out[mm][57]/leaves:.0f} % of the leaves, "
                          f"{out[mm][58]/leaves:.2f} re-fetch rounds per leaf")
            if KCL > 1 and PK == "fetch":
                for mm in (0, 1, KCL // 2, KCL - 1):
                    print(f"  X1 fetch (phase B, wave 0), member {mm}: first fetch {out[mm][56]/leaves:.0f} cycles, had to wait in {100*out[mm][57]/leaves:.0f} % of the leaves, "
                          f"{out[mm][58]/leaves:.2f} re-fetch rounds per leaf;  X2 prefix fetch (phase F, wave 0): first fetch {out[mm][59]/leaves:.0f} cycles, waited in "
                          f"{100*out[mm][60]/leaves:.0f} %, {out[mm][61]/leaves:.2f} re-fetch rounds per leaf")
            passes = leaves + 1e-9
            for mm in ((0, KCL // 2, KCL - 1) if KCL > 1 and PK == "c" else ()):
                print(f"  phase C, thread 0 of member {mm} (cycles per leaf since the phase began): " + ", ".join(f"{nm[3:]} {out[mm][k]/passes:.0f}" for k, nm in sub.items()))
    h.close()

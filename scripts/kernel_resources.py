#!/usr/bin/env python3
"""Register / scratch / LDS usage of every kernel in the built library (code-object metadata), e.g. to check spills:
    python scripts/kernel_resources.py [path/to/libpotus_hmc.so]
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")


def resources(lib):
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={tmp / 'fat.bin'}", str(lib), str(tmp / "copy.so")], check=True)
        subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={tmp / 'fat.bin'}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={tmp / 'dev.co'}"], check=True)
        txt = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(tmp / "dev.co")], check=True, capture_output=True, text=True).stdout
    out = []
    for blk in txt.split("- .agpr_count")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        out.append(dict(name=g("name"), vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), vspill=g("vgpr_spill_count"), sspill=g("sgpr_spill_count"),
                        scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size")))
    return out


if __name__ == "__main__":
    lib = Path(sys.argv[1]) if len(sys.argv) > 1 else Path(__file__).resolve().parent.parent / "us_potus_model_amd" / "libpotus_hmc.so"
    for r in resources(lib):
        print(f"{r['name'][:64]:64s} vgpr {r['vgpr']:>4s} sgpr {r['sgpr']:>4s} vspill {r['vspill']:>4s} sspill {r['sspill']:>4s} scratch {r['scratch']:>5s} lds {r['lds']:>6s}")

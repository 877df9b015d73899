#!/usr/bin/env python3
"""Scan the gfx950 code object of the library for the 16-byte-store data hazard.

A VMEM store of more than 64 bits reads its data registers one cycle after it issues: a VALU instruction that writes one of
them in the very next slot corrupts the stored data (observed on MI355X in round 4: dword 0 of an exchange word replaced by an
LDS address).  hipcc 7.2 pads the pair only for buffer stores WITHOUT a register soffset (LLVM's createsVALUHazard), so a
`raw_buffer_store_b128(..., soffset = SGPR, ...)` followed by a VALU write of its data registers goes out unpadded.  The sampler's
16-byte stores (xst, tw_st in potus_cluster.hpp) therefore carry an `s_nop 1` that names the data registers; this script checks
the result: every buffer_store_dwordx3/x4 (and _format_xyz/xyzw) must not be followed directly by a VALU write of its data.

    python scripts/check_store_hazard.py [path/to/libpotus_hmc.so]      exit status 1 when a hazard is found
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")
STORE = re.compile(r"^\s*buffer_store_(?:dwordx[34]|format_xyzw?|format_d16_xyzw)\s+v\[(\d+):(\d+)\]")
VALU = re.compile(r"^\s*(v_\w+)\s+(v\[(\d+):(\d+)\]|v(\d+))")
NOT_VALU_WRITE = ("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane", "v_nop")


def disassemble(lib):
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={tmp / 'fat.bin'}", str(lib), str(tmp / "copy.so")], check=True)
        subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={tmp / 'fat.bin'}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={tmp / 'dev.co'}"], check=True)
        return subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--no-show-raw-insn", str(tmp / "dev.co")], check=True, capture_output=True, text=True).stdout


def scan(text):
    hits, stores, func = [], 0, "?"
    lines = text.splitlines()
    for i, ln in enumerate(lines):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            func = m.group(1)
            continue
        s = STORE.match(ln)
        if not s:
            continue
        stores += 1
        lo, hi = int(s.group(1)), int(s.group(2))
        nxt = lines[i + 1] if i + 1 < len(lines) else ""
        v = VALU.match(nxt)
        if not v or v.group(1).startswith(NOT_VALU_WRITE):
            continue
        a, b = (int(v.group(3)), int(v.group(4))) if v.group(3) else (int(v.group(5)), int(v.group(5)))
        if a <= hi and b >= lo:
            hits.append((func, ln.split("//")[0].strip(), nxt.split("//")[0].strip()))
    return stores, hits


if __name__ == "__main__":
    lib = Path(sys.argv[1]) if len(sys.argv) > 1 else Path(__file__).resolve().parent.parent / "us_potus_model_amd" / "libpotus_hmc.so"
    stores, hits = scan(disassemble(lib))
    print(f"{lib}: {stores} buffer stores of more than 64 bits, {len(hits)} followed directly by a VALU write of their data registers")
    for f, a, b in hits:
        print(f"  {f[:60]}\n      {a}\n      {b}")
    sys.exit(1 if hits else 0)

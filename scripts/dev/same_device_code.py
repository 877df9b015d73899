#!/usr/bin/env python3
"""Development: are the gfx950 device functions of two builds of libpotus_hmc.so the same instructions?

    python scripts/dev/same_device_code.py build/variants/libpotus_a.so us_potus_model_amd/libpotus_hmc.so

Disassembles both code objects (as scripts/check_store_hazard.py does), strips addresses, and compares function by function.  A
refactoring behind a build flag that is off must leave every function identical -- then the build that was validated on the GPU and
the one being shipped are the same device code, whatever the source diff looks like (used at the end of round 4, when the GPU budget
was spent: the variants CL_F_LATE / CL_G_EXCHANGE / CL_DOT_PIPE / CL_SKIP_OOB were added around a default path that had to stay put).
Exit status 1 when a function differs or is missing.

Branch displacements and the literal of a `s_getpc_b64` / `s_add_u32` pair (the address of a global or a callee relative to the
instruction) change whenever ANY function of the code object changes size, so they are masked: what is compared is the instruction
stream.  `--diff N` prints the first N differing lines of every function that differs."""
import difflib
import re
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from check_store_hazard import disassemble  # noqa: E402


def functions(path):
    out, name, buf = {}, None, []
    for ln in disassemble(path).splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            if name:
                out[name] = buf
            name, buf = m.group(1), []
        elif name:
            ins = re.sub(r"^\s*[0-9a-f]+:\s*", "", ln).split("//")[0].strip()
            if re.match(r"s_c?branch\w*\s", ins):
                ins = ins.split()[0] + " <rel>"
            elif buf and re.match(r"s_add_u32 (s\d+), \1, 0x[0-9a-f]+$", ins) and (buf[-1].startswith("s_getpc_b64") or (len(buf) > 1 and buf[-2].startswith("s_getpc_b64"))):
                ins = re.sub(r"0x[0-9a-f]+$", "<pcrel>", ins)
            buf.append(ins)
    if name:
        out[name] = buf
    return out


def main():
    nd = 0
    if "--diff" in sys.argv:
        i = sys.argv.index("--diff")
        nd = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    a, b = functions(sys.argv[1]), functions(sys.argv[2])
    bad = 0
    for nm in sorted(set(a) | set(b)):
        if nm not in a or nm not in b:
            print(f"only in {'the first' if nm in a else 'the second'}: {nm}")
            bad += 1
        elif a[nm] != b[nm]:
            print(f"differs ({len(a[nm])} / {len(b[nm])} lines): {nm}")
            bad += 1
            if nd:
                d = [ln for ln in difflib.unified_diff(a[nm], b[nm], lineterm="", n=0) if not ln.startswith(("@@", "---", "+++"))]
                print("\n".join("    " + ln for ln in d[:nd]))
    print(f"{len(a)} / {len(b)} device functions, {bad} different")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

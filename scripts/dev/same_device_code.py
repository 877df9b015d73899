#!/usr/bin/env python3
"""Development: are the gfx950 device functions of two builds of libpotus_hmc.so the same instructions?

    python scripts/dev/same_device_code.py build/variants/libpotus_a.so us_potus_model_amd/libpotus_hmc.so

Disassembles both code objects (as scripts/check_store_hazard.py does), strips addresses, and compares function by function.  A
refactoring behind a build flag that is off must leave every function identical -- then the build that was validated on the GPU and
the one being shipped are the same device code, whatever the source diff looks like (used at the end of round 4, when the GPU budget
was spent: the variants CL_F_LATE / CL_G_EXCHANGE / CL_DOT_PIPE / CL_SKIP_OOB were added around a default path that had to stay put).
Exit status 1 when a function differs or is missing."""
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
            buf.append(re.sub(r"^\s*[0-9a-f]+:\s*", "", ln).split("//")[0].strip())
    if name:
        out[name] = buf
    return out


def main():
    a, b = functions(sys.argv[1]), functions(sys.argv[2])
    bad = 0
    for nm in sorted(set(a) | set(b)):
        if nm not in a or nm not in b:
            print(f"only in {'the first' if nm in a else 'the second'}: {nm}")
            bad += 1
        elif a[nm] != b[nm]:
            print(f"differs ({len(a[nm])} / {len(b[nm])} lines): {nm}")
            bad += 1
    print(f"{len(a)} / {len(b)} device functions, {bad} different")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

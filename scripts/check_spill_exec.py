#!/usr/bin/env python3
"""Scan the gfx950 code object of the built library for vector-register spills that execute under a narrowed EXEC mask.

Why: hipcc 7.2 (LLVM's greedy allocator) likes to put the spill of a long-lived VGPR into a cold block.  The exit block of
a rotated divergent loop restores EXEC twice (loop-exit mask, then the mask saved by the loop's guard); a
`scratch_store_dword` that lands between the two restores runs with the lanes that passed the guard only -- with none at
all if the guard failed for the whole wave -- and the reload further down, under the full mask, hands garbage to the
others.  That is how the day table of the one-workgroup pass (PassStatic::d_t) was lost in cold_transition_end: the
gradient of the S x T block was wrong at every window end of the 2016 posterior (found in round 3 by the test that
compares one workgroup per chain with two).

What is flagged: a `scratch_store` / `scratch_load` followed, inside the same basic block, by an instruction that widens EXEC
(`s_or_b64 exec, exec, ...`), with no `s_and_saveexec` / `s_andn2 exec` in between (a reload whose registers are all
redefined before the mask widens served the narrowed region itself and is let through).  Block boundaries are branch targets and
the instructions after branches.  Prologue / epilogue saves of callee-saved registers are plain stores at full EXEC and do
not match.  Exit status 1 if anything is flagged:

    python scripts/check_spill_exec.py [path/to/libpotus_hmc.so]
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")
INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):")
FUNC = re.compile(r"^[0-9a-f]+ <(\S+)>:")


def disassemble(lib):
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={tmp / 'fat.bin'}", str(lib), str(tmp / "copy.so")], check=True)
        subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={tmp / 'fat.bin'}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={tmp / 'dev.co'}"], check=True)
        return subprocess.run([str(LLVM / "llvm-objdump"), "-d", str(tmp / "dev.co")], check=True, capture_output=True, text=True).stdout


def functions(text):
    name, body = None, []
    for line in text.splitlines():
        m = FUNC.match(line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
            continue
        m = INS.match(line)
        if m and name:
            body.append((int(m.group(3), 16), m.group(1), m.group(2)))
    if name:
        yield name, body


def branch_target(addr, ops):
    m = re.match(r"(\d+)", ops)
    if not m:
        return None
    off = int(m.group(1))
    if off >= 0x8000:
        off -= 0x10000
    return addr + 4 + 4 * off


def regs(operand):
    """Vector registers named by one operand: v7 -> {7}, v[8:9] -> {8, 9}."""
    m = re.match(r"\s*v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"\s*v(\d+)\b", operand)
    return {int(m.group(1))} if m else set()


def scan(body):
    targets = set()
    for addr, op, ops in body:
        if op.startswith("s_cbranch") or op == "s_branch":
            t = branch_target(addr, ops)
            if t is not None:
                targets.add(t)
    hits = []
    pending = []           # spill instructions of the current block that no EXEC-narrowing instruction has followed yet
    wwm = False            # between `s_or_saveexec_b64 sX, -1` and `s_mov_b64 exec, sX`: whole-wave save of a register that holds
                           # spilled SGPRs in its lanes -- every lane is stored, nothing to flag
    for addr, op, ops in body:
        if addr in targets:
            pending = []
        if op.startswith("s_or_saveexec") and ops.rstrip().endswith("-1"):
            wwm, pending = True, []
        elif wwm and op == "s_mov_b64" and ops.startswith("exec"):
            wwm = False
        elif (op.startswith("scratch_store") or op.startswith("scratch_load")) and not wwm:
            pending.append((addr, op, ops, regs(ops.split(",")[0]) if op.startswith("scratch_load") else None))
        elif op == "s_or_b64" and ops.startswith("exec"):
            hits += [(p[:3], (addr, op, ops)) for p in pending]
            pending = []
        elif (op.startswith("s_and_saveexec") or (re.match(r"s_andn2_b64|s_and_b64|s_mov_b64|s_xor_b64", op) and ops.startswith("exec"))):
            pending = []
        elif pending and not re.search(r"store|write|v_cmp|s_waitcnt|s_nop", op):
            # a reload whose registers are all redefined before EXEC widens was meant for the narrowed region itself
            d = regs(ops.split(",")[0])
            for p in pending:
                if p[3]:
                    p[3].difference_update(d)
            pending = [p for p in pending if p[3] is None or p[3]]
        if op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64", "s_swappc_b64", "s_endpgm"):
            pending = []
    return hits


def main():
    lib = Path(sys.argv[1]) if len(sys.argv) > 1 else Path(__file__).resolve().parent.parent / "us_potus_model_amd" / "libpotus_hmc.so"
    bad = 0
    for name, body in functions(disassemble(lib)):
        for (a, op, ops), (a2, op2, ops2) in scan(body):
            print(f"{name[:72]}: {op} {ops}  at {a:#x}, EXEC widened at {a2:#x} by {op2} {ops2}")
            bad += 1
    print(f"{bad} spill instruction(s) under a narrowed EXEC mask")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

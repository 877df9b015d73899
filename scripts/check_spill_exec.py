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

    python scripts/check_spill_exec.py [path/to/libpotus_hmc.so] [--deep]

--deep adds a flow-based check (deep_scan): the regions of narrowed EXEC open at every spill instruction are tracked along
the code, and a reload is reported when no store to its slot ran in a context with at least the reload's lanes.  Both
checks report the round-2 library's cold_transition_end and nothing in the current one.
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


def slot(op, ops):
    """(base, offset) of a scratch access."""
    parts = [x.strip() for x in ops.split(",")]
    addr = parts[1:] if op.startswith("scratch_load") else [parts[0]] + parts[2:]
    m = re.search(r"offset:(\d+)", ops)
    base = " ".join(a.split(" offset")[0] for a in addr)
    return base, int(m.group(1)) if m else 0


def deep_scan(body):
    """Second, flow-based check.  Every narrowing of EXEC that saves the old mask (s_*_saveexec, or s_mov sX, exec followed by
    s_mov exec, sY) opens a region, the instruction that ORs the saved mask back closes it; the regions open at an instruction
    form its context (a tuple of region ids), carried along fall-through and forward branches.  A reload of a spill slot is
    covered if some store to that slot ran in a context that is a prefix of the reload's, i.e. with at least its lanes; a
    reload with no such store is reported.  Lanes that leave a loop (s_andn2 exec, exec, sX) do not open a region: they have
    been through the loop's first trip.  Whole-wave saves (s_or_saveexec sX, -1) are skipped."""
    state_at = {}                     # forward branch target -> (context, saved)
    ctx, saved, wwm, dead = (), {}, False, False
    stores, loads = {}, []
    for addr, op, ops in body:
        if addr in state_at:
            c2, s2 = state_at.pop(addr)
            if dead or len(c2) < len(ctx):
                ctx, saved = c2, dict(s2) if dead else {**saved, **s2}
            dead = False
        if dead:
            continue
        o = [x.strip() for x in ops.split(",")]
        if op.startswith("s_or_saveexec") and o[-1] == "-1":
            wwm = True
        elif wwm and op == "s_mov_b64" and o[0] == "exec":
            wwm = False
        elif wwm:
            pass
        elif op.startswith("s_and_saveexec"):
            saved[o[0]] = ctx
            ctx = ctx + (addr,)
        elif op.startswith("s_or_saveexec") or op.startswith("s_andn2_saveexec"):      # the switch to the else side of a region
            saved[o[0]] = saved.get(o[1], ctx[:-1])
        elif op == "s_mov_b64" and o[1:] == ["exec"]:
            saved[o[0]] = ctx
        elif op == "s_mov_b64" and o[0] == "exec":
            ctx = saved[o[1]] if o[1] in saved and len(saved[o[1]]) <= len(ctx) else ctx + (addr,)
        elif op == "s_or_b64" and o[0] == "exec" and len(o) == 3 and o[1] == "exec":
            if o[2] in saved and len(saved[o[2]]) < len(ctx):
                ctx = saved[o[2]]
        elif op.startswith("scratch_store"):
            stores.setdefault(slot(op, ops), []).append((addr, ctx))
        elif op.startswith("scratch_load"):
            loads.append((addr, slot(op, ops), ctx, ops))
        elif re.match(r"s_(mov|and|or|xor|andn2|orn2|cselect|not)_b64", op) and o[0].startswith("s[") and o[0] in saved and not (op == "s_xor_b64" and "exec" in o):
            del saved[o[0]]                                                             # the register no longer holds that mask
        if op.startswith("s_cbranch") or op == "s_branch":
            t = branch_target(addr, ops)
            if t is not None and t > addr and t not in state_at:
                state_at[t] = (ctx, dict(saved))
        if op in ("s_branch", "s_setpc_b64", "s_endpgm"):
            dead = True
    hits = []
    for addr, sl, ctx, ops in loads:
        st = stores.get(sl, [])
        if st and not any(c == ctx[:len(c)] for _, c in st):
            hits.append((addr, sl, len(ctx), [(hex(a), len(c)) for a, c in st][:4]))
    return hits


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = Path(args[0]) if args else Path(__file__).resolve().parent.parent / "us_potus_model_amd" / "libpotus_hmc.so"
    bad = 0
    text = disassemble(lib)
    for name, body in functions(text):
        for (a, op, ops), (a2, op2, ops2) in scan(body):
            print(f"{name[:72]}: {op} {ops}  at {a:#x}, EXEC widened at {a2:#x} by {op2} {ops2}")
            bad += 1
    print(f"{bad} spill instruction(s) under a narrowed EXEC mask")
    if "--deep" in sys.argv:
        n = 0
        for name, body in functions(text):
            for addr, sl, depth, st in deep_scan(body):
                print(f"{name[:72]}: reload of slot {sl} at {addr:#x} (depth {depth}) has no store in a context at least as wide; stores (address, depth): {st}")
                n += 1
        print(f"{n} reload(s) not covered by a store with at least their lanes (flow-based check; review by hand)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

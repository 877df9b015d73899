"""Checks on the built gfx950 code object that need no GPU.

scripts/check_spill_exec.py: no vector-register spill may execute under an EXEC mask that is widened later in the same
block -- the shape in which hipcc 7.2 lost PassStatic::d_t in cold_transition_end (round 3: the S x T block of the gradient
kept at a window end was wrong for the 2016 posterior on the one-workgroup path, and k_run stored a spilled pair with an
empty mask in every leaf)."""
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _load(name):
    spec = importlib.util.spec_from_file_location(name, ROOT / "scripts" / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def test_no_spill_runs_under_a_narrowed_exec_mask():
    import __graft_entry__ as g
    g.build()
    chk = _load("check_spill_exec")
    lib = ROOT / "us_potus_model_amd" / "libpotus_hmc.so"
    text = chk.disassemble(lib)
    hits = [(name, a, b) for name, body in chk.functions(text) for a, b in chk.scan(body)]
    assert not hits, hits[:5]
    # the flow-based check: every reload of a spill slot has a store to that slot that ran with at least its lanes
    deep = [(name, h) for name, body in chk.functions(text) for h in chk.deep_scan(body)]
    assert not deep, deep[:5]


def test_the_scanner_sees_the_shape_it_is_looking_for():
    chk = _load("check_spill_exec")
    # the exit block of a rotated divergent loop: loop-exit mask restored, spill, guard mask restored
    body = [(0x00, "s_and_saveexec_b64", "s[6:7], vcc"), (0x04, "s_cbranch_execz", "3"), (0x08, "v_add_f64", "v[0:1], v[0:1], v[2:3]"),
            (0x0c, "s_andn2_b64", "exec, exec, s[8:9]"), (0x10, "s_cbranch_execnz", "65533"),
            (0x14, "s_or_b64", "exec, exec, s[8:9]"), (0x18, "scratch_store_dword", "off, v53, s32 offset:728"), (0x20, "s_or_b64", "exec, exec, s[6:7]"),
            (0x24, "s_endpgm", "")]
    assert [h[0][0] for h in chk.scan(body)] == [0x18]
    # a whole-wave save of a register that carries spilled SGPRs, and a reload consumed inside the narrowed region: let through
    ok = [(0x00, "s_or_saveexec_b64", "s[100:101], -1"), (0x04, "scratch_store_dword", "off, v251, off"), (0x0c, "s_mov_b64", "exec, s[100:101]"),
          (0x10, "s_or_b64", "exec, exec, s[4:5]"),
          (0x14, "s_and_saveexec_b64", "s[4:5], vcc"), (0x18, "scratch_load_dwordx2", "v[8:9], off, off offset:76"),
          (0x20, "global_load_dwordx2", "v[8:9], v9, s[8:9] offset:448"), (0x28, "s_or_b64", "exec, exec, s[4:5]"), (0x2c, "s_endpgm", "")]
    assert chk.scan(ok) == []
    # the flow-based check on the same shapes: the reload at full EXEC of a slot stored inside the guard's region is reported,
    # a reload inside the region of a slot stored before it is not
    bad = body[:-1] + [(0x24, "scratch_load_dword", "v212, off, s32 offset:728"), (0x2c, "s_endpgm", "")]
    assert [h[0] for h in chk.deep_scan(bad)] == [0x24]
    fine = [(0x00, "scratch_store_dword", "off, v53, s32 offset:728")] + [(a + 8, op, ops) for a, op, ops in body[:3]] + \
           [(0x14, "scratch_load_dword", "v212, off, s32 offset:728"), (0x1c, "s_or_b64", "exec, exec, s[6:7]"), (0x20, "s_endpgm", "")]
    assert chk.deep_scan(fine) == []


def test_no_valu_write_follows_a_wide_buffer_store_onto_its_data_registers():
    """scripts/check_store_hazard.py: a buffer store of more than 64 bits reads its data registers after it has issued, and hipcc
    7.2 pads the pair only for stores without a register soffset.  Round 4 lost the low dword of exchange words that way (one word
    in ~10^5: gradients wrong in the ninth digit, runs not reproducible); xst / tw_st now carry their own wait states."""
    import __graft_entry__ as g
    g.build()
    chk = _load("check_store_hazard")
    stores, hits = chk.scan(chk.disassemble(ROOT / "us_potus_model_amd" / "libpotus_hmc.so"))
    assert stores > 100 and not hits, hits[:5]


def test_the_store_hazard_scanner_sees_the_shape():
    chk = _load("check_store_hazard")
    bad = "0000000000001000 <k>:\n\tbuffer_store_dwordx4 v[2:5], v6, s[88:91], s1 offen sc1  // 0\n\tv_cndmask_b32_e32 v2, v159, v132, vcc  // 8\n"
    assert len(chk.scan(bad)[1]) == 1
    padded = "0000000000001000 <k>:\n\tbuffer_store_dwordx4 v[2:5], v6, s[88:91], s1 offen sc1  // 0\n\ts_nop 1  // 8\n\tv_cndmask_b32_e32 v2, v159, v132, vcc  // c\n"
    other = "0000000000001000 <k>:\n\tbuffer_store_dwordx4 v[2:5], v6, s[88:91], s1 offen sc1  // 0\n\tv_mov_b32_e32 v7, v1  // 8\n"
    narrow = "0000000000001000 <k>:\n\tbuffer_store_dwordx2 v[2:3], v6, s[88:91], s1 offen sc1  // 0\n\tv_mov_b32_e32 v2, v1  // 8\n"
    assert chk.scan(padded)[1] == [] and chk.scan(other)[1] == [] and chk.scan(narrow)[1] == []

"""CPU tests of the C-ABI boundary: the library loads, exports every declared symbol, and the
host-only entry points (layout arithmetic, column names, validation, error path) behave.
No compute call is made here -- that needs an MI355X (tests/test_gpu_*.py)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from us_potus_model_amd import _abi, sampler

ROOT = Path(__file__).resolve().parent.parent


def _has_gpu():
    from conftest import _has_gpu as has
    return has()


def test_library_exports_every_declared_symbol():
    hdr = (ROOT / "include" / "potus_hmc.h").read_text()
    declared = set(re.findall(r"\b(potus_[A-Za-z_0-9]+)\s*\(", hdr))
    assert declared == set(sampler.EXPORTS), declared ^ set(sampler.EXPORTS)
    L = sampler.load_library()
    for name in sorted(declared):
        assert hasattr(L, name), f"libpotus_hmc.so does not export {name}"
    assert b"gfx950" in L.potus_version()
    # ... and the other way round: every potus_* symbol the library exports is declared, in the boundary header or -- the
    # development / verification hooks the tests use -- in include/potus_hmc_debug.h
    import subprocess
    dbg = set(re.findall(r"\b(potus_[A-Za-z_0-9]+)\s*\(", (ROOT / "include" / "potus_hmc_debug.h").read_text()))
    assert not (dbg & declared), dbg & declared
    nm = subprocess.run(["nm", "-D", "--defined-only", str(sampler.lib_path())], check=True, capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if ln.split()[-1].startswith("potus_") and ln.split()[-2] in ("T", "t")}
    assert exported == declared | dbg, exported ^ (declared | dbg)


def test_struct_layout_matches_header():
    # potus_data: 7 int32 + pad, 13 + 4 pointers, 6 doubles, pointer, 3 doubles, int32 (+pad)
    assert C.sizeof(_abi.PotusData) == 32 + 17 * 8 + 6 * 8 + 8 + 3 * 8 + 8
    assert C.sizeof(_abi.PotusOpts) == 8 * 4 + 6 * 8 + 8 + 8 * 4             # ... device, save_warmup, cus_per_chain, metric, twin, metric_storage, pooled_metric, reserved_


@pytest.mark.parametrize("name,D,ncols", [("2016", 15098, 43360), ("small_full", 292, 753), ("small_nomode", 260, 690)])
def test_num_params_and_columns(cases, name, D, ncols):
    data, variant = cases[name]
    L = sampler.load_library()
    d, keep = _abi.make_data(data, variant)
    a, b = C.c_int(), C.c_int()
    assert L.potus_num_params(C.byref(d), C.byref(a)) == 0 and a.value == D == _abi.num_params(data, variant)
    assert L.potus_num_columns(C.byref(d), C.byref(b)) == 0 and b.value == ncols
    layout, n2 = _abi.column_layout(data, variant)
    assert n2 == ncols


def test_column_names_follow_cmdstan(cases):
    data, variant = cases["2016"]
    L = sampler.load_library()
    d, keep = _abi.make_data(data, variant)
    buf = C.create_string_buffer(96)

    def name(k):
        assert L.potus_column_name(C.byref(d), k, buf, 96) == 0
        return buf.value.decode()

    layout, ncols = _abi.column_layout(data, variant)
    assert [name(k) for k in range(7)] == list(_abi.SAMPLER_COLS)
    assert name(7) == "raw_mu_b_T.1"
    a, b, _ = layout["raw_mu_b"]
    assert name(a) == "raw_mu_b.1.1" and name(a + 1) == "raw_mu_b.2.1" and name(a + 51) == "raw_mu_b.1.2"
    assert name(b - 1) == "raw_mu_b.51.254"
    assert name(layout["mu_e_bias"][0]) == "mu_e_bias" and name(layout["rho_e_bias"][0]) == "rho_e_bias"
    a, b, _ = layout["mu_b"]
    assert name(a) == "mu_b.1.1" and name(b - 1) == "mu_b.51.254"
    a, b, _ = layout["predicted_score"]
    assert name(a) == "predicted_score.1.1" and name(a + 1) == "predicted_score.2.1" and name(b - 1) == "predicted_score.254.51"
    assert b == ncols
    assert L.potus_column_name(C.byref(d), ncols, buf, 96) != 0


def test_create_rejects_bad_data_before_touching_the_device(cases):
    data, variant = cases["small_full"]
    L = sampler.load_library()
    o = _abi.PotusOpts()
    L.potus_default_opts(C.byref(o))
    assert (o.num_warmup, o.num_samples, o.max_depth, o.seed) == (1000, 1000, 10, 1843)
    assert (o.delta, o.gamma, o.kappa, o.t0) == (0.8, 0.05, 0.75, 10.0)
    bad = dict(data)
    bad["day_state"] = np.array(data["day_state"]).copy()
    bad["day_state"][0] = int(data["T"]) + 1
    d, keep = _abi.make_data(bad, variant)
    h = C.c_int(-1)
    rc = L.potus_create(C.byref(d), C.byref(o), C.byref(h))
    assert rc == 1
    buf = C.create_string_buffer(256)
    L.potus_last_error(buf, 256)
    assert b"day_state" in buf.value


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure path")
def test_no_silent_cpu_fallback(cases):
    """Without an MI355X the product path must fail loudly, never compute on the CPU."""
    data, variant = cases["small_full"]
    with pytest.raises(sampler.PotusError, match="HIP device|MI355X|gfx950|failed"):
        sampler.Handle(data, variant, chains=1)


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing in the package may import/link/dlopen it."""
    for p in (ROOT / "us_potus_model_amd").rglob("*"):
        if p.is_file() and p.suffix in {".py", ".hip", ".hpp", ".h"}:
            txt = p.read_text()
            assert "potus_oracle" not in txt and "oracle_lib" not in txt, p
    import subprocess
    out = subprocess.run(["ldd", str(sampler.lib_path())], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_device_code_has_no_waterfall_loops(tmp_path):
    """Every buffer access takes its resource and scalar offset from scalar registers.  When the compiler cannot prove
    such an operand wave-uniform it wraps the access in a `v_readfirstlane ... s_cbranch_execnz` (waterfall) loop;
    that once cost 13 % of the time per leapfrog.  Checked on the disassembly of the built library."""
    import shutil
    import subprocess
    llvm = Path("/opt/rocm/lib/llvm/bin")
    tools = [llvm / "llvm-objcopy", llvm / "clang-offload-bundler", llvm / "llvm-objdump"]
    if not all(t.exists() for t in tools) or not sampler.lib_path().exists():
        pytest.skip("ROCm LLVM tools or the built library not present")
    fat, co = tmp_path / "fat.bin", tmp_path / "dev.co"
    subprocess.run([str(tools[0]), f"--dump-section=.hip_fatbin={fat}", str(sampler.lib_path()), str(tmp_path / "copy.so")], check=True)
    subprocess.run([str(tools[1]), "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--output={co}"], check=True)
    asm = subprocess.run([str(tools[2]), "-d", str(co)], check=True, capture_output=True, text=True).stdout.splitlines()
    assert sum("s_endpgm" in ln for ln in asm) >= 8            # the kernels are there
    mem = re.compile(r"\b(buffer|global|scratch)_(load|store|atomic)")
    bad = [asm[i].strip() for i in range(len(asm) - 1) if mem.search(asm[i]) and "s_xor_b64 exec, exec" in asm[i + 1]]
    assert not bad, bad[:5]
    shutil.rmtree(tmp_path, ignore_errors=True)


def test_the_library_was_built_from_these_sources():
    """libpotus_hmc.so is git-ignored and travels prebuilt to the GPU box: the digest __graft_entry__.build() leaves beside it must be that of the
    sources in the tree (a source edited after -- or during -- the last build would otherwise ship a library that is not the code under review)."""
    import __graft_entry__ as g
    src = g.PKG / "csrc"
    deps = [src / "potus_hmc.hip", *sorted(src.glob("*.hpp")), *sorted((g.ROOT / "include").glob("*.h"))]
    stamp = g.PKG / "libpotus_hmc.so.src"
    assert stamp.exists(), "run `python __graft_entry__.py` (build())"
    assert stamp.read_text().strip() == g._digest(deps), "libpotus_hmc.so is older than its sources: run `python __graft_entry__.py`"


def test_dot_call_wrappers_compile_and_fail_like_r_errors(tmp_path):
    """R/src/potus_call.c -- the .Call() entry points the R shim prefers when they are loaded (north_star: "thin C-ABI .Call()/dyn.load FFI") -- compiles
    warning-free against the stub R API of tests/r_stub/ (R is not in the image), exports the four entry points R/potus_sampling.R names, and turns a
    library error into an R error carrying the library's message, with the PROTECT stack balanced (without a GPU every call ends that way:
    tests/test_gpu_boundary.py drives the wrappers for real)."""
    import ctypes as C
    from conftest import build_call_wrapper, call_wrapper_int
    W = build_call_wrapper(tmp_path)
    shim = open(ROOT / "R" / "potus_sampling.R").read()
    for name in ("potus_call_extract", "potus_call_diagnostics", "potus_call_sampler_params", "potus_call_version"):
        assert hasattr(W, name)
    assert '.Call("potus_call_extract"' in shim and '.Call("potus_call_diagnostics"' in shim and 'is.loaded("potus_call_extract")' in shim
    v = W.stub_call0(C.cast(W.potus_call_version, C.c_void_p))
    assert W.stub_string(v).decode().startswith("potus_hmc 0.5") and W.stub_last_error() == b""
    h = call_wrapper_int(W, [12345])                                   # no such handle
    for fn in (W.potus_call_extract, W.potus_call_diagnostics):
        r = W.stub_call3(C.cast(fn, C.c_void_p), h, call_wrapper_int(W, [0]), call_wrapper_int(W, [7]))
        assert b"libpotus_hmc error" in W.stub_last_error() and b"handle" in W.stub_last_error(), W.stub_last_error()
        assert W.XLENGTH(r) == 0 and W.stub_protect_depth() == 0       # R_NilValue came back through the error handler; nothing left protected
    W.stub_call1(C.cast(W.potus_call_sampler_params, C.c_void_p), call_wrapper_int(W, [12345]))
    assert b"libpotus_hmc error" in W.stub_last_error() and W.stub_protect_depth() == 0
    W.stub_release_all()
    assert W.stub_live_bytes() == 0

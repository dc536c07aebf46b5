"""CPU: the C-ABI library loads, exports every symbol include/ssdn_hip.h declares, and the ctypes struct mirrors of
ssdn/hip/lib.py have the sizes the compiler gave the C structs.  No compute call (there is no GPU here)."""
import ctypes as C
import os
import re

from ssdn.hip import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_header_symbols():
    lib = L.load()
    hdr = open(os.path.join(ROOT, "include", "ssdn_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*|void\*?)\s+(ssdn_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert declared == set(L.SYMBOLS), (declared ^ set(L.SYMBOLS))
    for s in declared:
        assert hasattr(lib, s), s
    # ... and the other direction: every ssdn_* symbol the library exports is declared in the header
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("ssdn_") and " T " in ln}
    assert exported == declared, (exported ^ declared)
    m = re.search(r"#define SSDN_ABI_VERSION (\d+)", hdr)
    assert lib.ssdn_abi_version() == L.ABI_VERSION == int(m.group(1))


def test_struct_mirrors_match_compiler_layout():
    lib = L.load()
    assert lib.ssdn_struct_size(0) == C.sizeof(L.OpRec)
    for name, ty in L.ARG_TYPES.items():
        assert lib.ssdn_struct_size(L.OP[name]) == C.sizeof(ty), name
    # enum values in the header == the binding's table
    hdr = open(os.path.join(ROOT, "include", "ssdn_hip.h")).read()
    for m in re.finditer(r"SSDN_OP_([A-Z_]+)\s*=\s*(\d+)", hdr):
        key = m.group(1).lower()
        assert L.OP[key] == int(m.group(2)), key


def test_invalid_ops_fail_loudly_without_a_gpu():
    lib = L.load()
    bad = L.ConvArgs()
    bad.ntaps = 0
    assert lib.ssdn_conv_lds_bytes(C.byref(bad)) < 0
    assert b"ntaps" in lib.ssdn_last_error()
    rec = (L.OpRec * 1)()
    rec[0].type = 999
    rec[0].args = C.cast(C.pointer(bad), C.c_void_p)
    assert lib.ssdn_run_ops(rec, 1, None) != 0
    assert b"unknown type" in lib.ssdn_last_error()


def test_missing_library_is_an_error_not_a_fallback(monkeypatch, tmp_path):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        L.load()
        assert False, "expected SsdnHipError"
    except L.SsdnHipError as e:
        assert "no CPU fallback" in str(e)


def test_graft_entry_build_is_consistent():
    """the driver's build check: __graft_entry__.build() (make + import + ABI assertion) succeeds on a CPU-only host"""
    import importlib
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    g = importlib.import_module("__graft_entry__")
    g.build()

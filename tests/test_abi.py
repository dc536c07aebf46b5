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
    declared = set(re.findall(r"^\s*(?:int|int64_t|const char\*|void\*?)\s+(ssdn_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
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


def test_plan_blob_loader_on_the_host():
    """The step-level entry points' loader (csrc/plan.hip) needs no GPU: a hand-made blob with two tensors (one an alias) and no ops loads,
    reports its arena size and description, resolves names after a bind; a truncated blob, a foreign ABI number and an argument struct of
    the wrong size are refused with a message."""
    import struct
    lib = L.load()

    def blob(abi=L.ABI_VERSION, ops=b"", nops=0, cut=0):
        t = b"".join(n.encode().ljust(56, b"\0") + struct.pack("<Qii", nb, al, 0) for n, nb, al in (("params", 1000, -1), ("m/in32", 4096, -1), ("inp", 4096, 1)))
        meta = b'{"B": 2}'
        b = b"SSDNPLAN" + struct.pack("<IIII", 1, abi, 3, 4) + t + struct.pack("<I", nops) + ops + struct.pack("<I", 0) * 3 + struct.pack("<I", len(meta)) + meta
        return b[:len(b) - cut]
    plan = C.c_void_p()
    good = blob()
    assert lib.ssdn_plan_load(good, len(good), C.byref(plan)) == 0, lib.ssdn_last_error()
    assert lib.ssdn_plan_arena_bytes(plan) == 1024 + 4096 and lib.ssdn_plan_meta(plan) == b'{"B": 2}'
    assert lib.ssdn_plan_run(plan, 1, None) != 0 and b"not bound" in lib.ssdn_last_error()
    assert lib.ssdn_plan_bind(plan, C.c_void_p(0x10000)) == 0
    p, n = C.c_void_p(), C.c_int64()
    assert lib.ssdn_plan_tensor(plan, b"inp", C.byref(p), C.byref(n)) == 0 and p.value == 0x10000 + 1024 and n.value == 4096
    assert lib.ssdn_plan_tensor(plan, b"nope", C.byref(p), C.byref(n)) != 0
    assert lib.ssdn_plan_set_lr(plan, 3e-4, 1, 1.0) != 0            # (no optimiser in this plan)
    lib.ssdn_plan_destroy(plan)
    for bad, msg in ((blob(cut=5), b"truncated"), (blob(abi=L.ABI_VERSION + 1), b"ABI"),
                     (blob(ops=struct.pack("<iiII", L.OP["adam"], 0, 8, 0) + b"\0" * 8, nops=1), b"argument bytes")):
        plan = C.c_void_p()
        assert lib.ssdn_plan_load(bad, len(bad), C.byref(plan)) != 0 and msg in lib.ssdn_last_error(), lib.ssdn_last_error()

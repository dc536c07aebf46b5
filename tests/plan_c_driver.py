"""Standalone driver of a plan blob through the C ABI of libssdn_hip.so (tests/test_hip_plan_c.py runs it in a subprocess).
Imports NOTHING of this repository: ctypes for the library, torch only as the owner of device memory and the stream.
usage: python plan_c_driver.py <libssdn_hip.so> <plan blob> <inputs.pt> <out.pt>"""
import ctypes as C
import json
import sys

import torch

lib = C.CDLL(sys.argv[1])
lib.ssdn_last_error.restype = C.c_char_p
lib.ssdn_plan_load.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
lib.ssdn_plan_arena_bytes.argtypes = [C.c_void_p]
lib.ssdn_plan_arena_bytes.restype = C.c_int64
lib.ssdn_plan_meta.argtypes = [C.c_void_p]
lib.ssdn_plan_meta.restype = C.c_char_p
lib.ssdn_plan_bind.argtypes = [C.c_void_p, C.c_void_p]
lib.ssdn_plan_tensor.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
lib.ssdn_plan_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
lib.ssdn_train_step.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_void_p]


def ok(rc):
    assert rc == 0, lib.ssdn_last_error().decode()


blob = open(sys.argv[2], "rb").read()
inp = torch.load(sys.argv[3])
plan = C.c_void_p()
ok(lib.ssdn_plan_load(blob, len(blob), C.byref(plan)))
meta = json.loads(lib.ssdn_plan_meta(plan).decode())
arena = torch.zeros(lib.ssdn_plan_arena_bytes(plan) + 256, dtype=torch.uint8, device="cuda:0")
base = (arena.data_ptr() + 255) & ~255
ok(lib.ssdn_plan_bind(plan, C.c_void_p(base)))


def view(name, dtype):
    p, n = C.c_void_p(), C.c_int64()
    ok(lib.ssdn_plan_tensor(plan, name.encode(), C.byref(p), C.byref(n)))
    off = p.value - arena.data_ptr()
    return arena[off:off + n.value].view(dtype)


view("params", torch.float32)[:inp["params"].numel()].copy_(inp["params"])
view("m/in32", torch.float32).copy_(inp["noisy"].reshape(-1))
if inp.get("noise_param") is not None:          # ssdn with a known noise parameter
    view("noise_param", torch.float32).copy_(inp["noise_param"].reshape(-1))
if inp.get("ref") is not None:                  # mse / mask_mse: the reference image
    view("ref", torch.float32).copy_(inp["ref"].reshape(-1))
if inp.get("coords") is not None:               # mask_mse: the masked coordinates (int64 [ncoords, 2])
    view("coords", torch.int64).copy_(inp["coords"].reshape(-1))
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ok(lib.ssdn_plan_run(plan, 0, stream))                       # SSDN_PLAN_REPACK: MFMA shadows of the parameters just written
out = {"meta": meta, "loss": [], "params": None}
for step in range(1, inp["steps"] + 1):
    ok(lib.ssdn_train_step(plan, C.c_float(inp["lr"]), step, stream))
    torch.cuda.synchronize()
    out["loss"].append(view("loss", torch.float32).cpu().clone())
out["params"] = view("params", torch.float32).cpu().clone()
if meta.get("pipeline") == "ssdn":
    out["pme"] = view("pme", torch.float32).cpu().clone()
torch.save(out, sys.argv[4])
print("plan_c_driver: %d steps, loss[0][:3] = %s" % (inp["steps"], out["loss"][0][:3].tolist()))

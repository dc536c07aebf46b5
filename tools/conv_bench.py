"""Micro-benchmark of single SSDN_OP_CONV / SSDN_OP_WGRAD launches with overridden tilings (tuning aid, GPU only).
usage: python tools/conv_bench.py [layer-name ...]"""
import os, sys, itertools, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "selfsupervised-denoising_amd")]
import torch
from ssdn.hip import lib as L
from ssdn.hip.engine import DeviceNet, OpList, current_stream
from ssdn.hip.graph import NetPlan, Op


def time_op(dn, op, iters=20):
    rec = dn._mat(op)
    ol = OpList([rec])
    s = current_stream()
    for _ in range(3):
        ol.run(s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        ol.run(s)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    B, P = 32, 64
    cus = L.load().ssdn_device_cus()
    plan = NetPlan("m/", 3, 9, True, B, P, P, cus=cus)
    dev = torch.device("cuda:0")
    flat = torch.randn(plan.nparams, device=dev) * 0.05
    dn = DeviceNet(plan, dev, flat, torch.zeros_like(flat))
    for name, t in dn.t.items():
        if t.dtype in (torch.float16, torch.bfloat16):
            t.copy_(torch.randn(t.shape, device=dev) * 0.5)
    dn.pack.run(current_stream())
    want = sys.argv[1:] or ["decode_block_1.2", "decode_block_2.2", "output_block.0", "encode_block_1.2", "decode_block_5.0"]
    for op in plan.fwd + plan.bwd:
        if op.type != "conv" or op.a["layer"] not in want:
            continue
        a = op.a
        flops = 2.0 * a["N"] * a["H"] * a["W"] * a["M"] * a["Ktot"] * len(a["taps"])
        base = time_op(dn, op)
        print("%-18s %-5s H=%3d K=%3d M=%3d default tile (%d,%d,%d) kc=%d : %8.1f us  %7.1f TF" % (
            a["layer"], a["role"], a["H"], a["Ktot"], a["M"], 1 << a["ltw"], 1 << a["lth"], 1 << a["ltn"], a["kc"], base, flops / base / 1e6))
        if os.environ.get("CONV_BENCH_ONLY_DEFAULT"):
            continue
        kcs = [kc for kc in (16, 32, 48, 64, 96) if a["Ktot"] % kc == 0]
        tiles = [(5, 3, 0), (4, 4, 0), (3, 5, 0)] if a["H"] >= 32 else [(a["ltw"], a["lth"], a["ltn"])]
        for (ltw, lth, ltn), kc in itertools.product(tiles, kcs):
            o = Op("conv", dict(a))
            o.a.update(ltw=ltw, lth=lth, ltn=ltn, kc=kc)
            try:
                t = time_op(dn, o)
            except Exception as e:
                print("      tile (%d,%d,%d) kc=%d: %s" % (1 << ltw, 1 << lth, 1 << ltn, kc, str(e)[:60]))
                continue
            print("      tile (%2d,%2d,%d) kc=%3d : %8.1f us  %7.1f TF" % (1 << ltw, 1 << lth, 1 << ltn, kc, t, flops / t / 1e6))


if __name__ == "__main__":
    main()

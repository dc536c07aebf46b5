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
    wrep = int(os.environ.get("SSDN_CDMA_WREP", "0"))
    if wrep > 1:      # experiment: every packed weight tensor in `wrep` consecutive copies
        for name in list(dn.t):
            if "/wf/" in name or "/wd/" in name:
                dn.t[name] = dn.t[name].repeat(wrep).contiguous()
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
        kcs = [kc for kc in (32, 48, 64) if a["Ktot"] % kc == 0]
        tiles = [(4, 4, 0), (5, 4, 0), (4, 5, 0)] if a["H"] >= 32 else [(a["ltw"], a["lth"], a["ltn"])]
        if len(a["taps"]) == 1:
            tiles = [(5, 0, 3), (5, 0, 4)]
        for (ltw, lth, ltn), kc in itertools.product(tiles, kcs):
            o = Op("conv", dict(a))
            o.a.update(ltw=ltw, lth=lth, ltn=ltn, kc=kc)
            try:
                t = time_op(dn, o)
            except Exception as e:
                print("      tile (%d,%d,%d) kc=%d: %s" % (1 << ltw, 1 << lth, 1 << ltn, kc, str(e)[:60]))
                continue
            print("      tile (%2d,%2d,%d) kc=%3d : %8.1f us  %7.1f TF" % (1 << ltw, 1 << lth, 1 << ltn, kc, t, flops / t / 1e6))




def all_ops(kind="wgrad"):
    """time every op of one type of the training plan in isolation (us, TF/s)"""
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
    tot = 0.0
    for op in plan.fwd + plan.bwd:
        if op.type != kind:
            continue
        a = op.a
        t = time_op(dn, op)
        tot += t
        if kind in ("wgrad", "conv"):
            flops = 2.0 * a["N"] * a["H"] * a["W"] * a["M"] * (a.get("kreal") or a["Ktot"]) * len(a["taps"])   # real channels
            print("%-18s %-5s N=%3d H=%3d K=%3d M=%3d taps=%d tile (%d,%d,%d) : %8.1f us  %7.1f TF" % (
                a["layer"], a.get("role", kind), a["N"], a["H"], a["Ktot"], a["M"], len(a["taps"]), 1 << a["ltw"], 1 << a["lth"], 1 << a["ltn"],
                t, flops / t / 1e6))
        else:
            print("%-18s %8.1f us" % (a.get("layer", ""), t))
    print("total %.1f us" % tot)


def pair(*layers):
    """run the forward convs of the given layers back to back (launch-gap experiments under rocprofv3)"""
    B, P = 32, 64
    cus = L.load().ssdn_device_cus()
    plan = NetPlan("m/", 3, 9, True, B, P, P, cus=cus)
    dev = torch.device("cuda:0")
    flat = torch.randn(plan.nparams, device=dev) * 0.05
    dn = DeviceNet(plan, dev, flat, torch.zeros_like(flat))
    dn.pack.run(current_stream())
    recs = []
    for name in layers:
        for op in plan.fwd:
            if op.type == "conv" and op.a["layer"] == name:
                recs.append(dn._mat(op))
    ol = OpList(recs)
    for _ in range(30):
        ol.run(current_stream())
    torch.cuda.synchronize()


def trace(layer="decode_block_1.2", role="fwd"):
    """per-workgroup phase timeline from s_memtime stamps (ssdn_debug_set_trace); needs a `make TUNING=1` build of the library"""
    import ctypes as C
    import numpy as np
    B, P = 32, 64
    lib = L.load()
    plan = NetPlan("m/", 3, 9, True, B, P, P, cus=lib.ssdn_device_cus())
    dev = torch.device("cuda:0")
    flat = torch.randn(plan.nparams, device=dev) * 0.05
    dn = DeviceNet(plan, dev, flat, torch.zeros_like(flat))
    for name, t in dn.t.items():
        if t.dtype in (torch.float16, torch.bfloat16):
            t.copy_(torch.randn(t.shape, device=dev) * 0.5)
    dn.pack.run(current_stream())
    if role == "wgrad":
        op = [o for o in plan.bwd if o.type == "wgrad" and o.a["layer"] == layer][0]
    else:
        op = [o for o in plan.fwd + plan.bwd if o.type == "conv" and o.a["layer"] == layer and o.a["role"] == role][0]
    ol = OpList([dn._mat(op)])
    for _ in range(3):
        ol.run(current_stream())
    buf = torch.zeros(4096 * 32, dtype=torch.int64, device=dev)
    lib.ssdn_debug_set_trace.argtypes = [C.c_void_p]
    lib.ssdn_debug_set_trace(C.c_void_p(buf.data_ptr()))
    ol.run(current_stream())
    torch.cuda.synchronize()
    lib.ssdn_debug_set_trace(None)
    t = buf.cpu().view(-1, 32).numpy()
    t = t[t[:, 0] > 0]
    if (t[:, 31] >> 32).any():          # k_cdma tuning builds: HW_ID of wave 0 in the last slot
        hw = (t[:, 31] & 0xffffffff).astype(np.int64)
        t = t.copy(); t[:, 31] = 0
        f = lambda lo, n: (hw >> lo) & ((1 << n) - 1)
        import collections
        print("HW_ID fields: wave_id", sorted(collections.Counter(f(0, 4)).items()), " simd", sorted(collections.Counter(f(4, 2)).items()),
              " cu", len(set(f(8, 4))), " sh", sorted(collections.Counter(f(12, 1)).items()), " se", sorted(collections.Counter(f(13, 3)).items()),
              " tg_id", sorted(collections.Counter(f(16, 4)).items()))
        # start offsets inside one XCD (its own clock): workgroups b % 8 == 0
        g = t[0::8]
        off = np.sort(g[:, 0] - g[:, 0].min())
        print("XCD 0: start offsets of its %d workgroups (ticks):" % len(g), [int(v) for v in off[::max(1, len(off) // 16)]])
        tg = ((hw[0::8] >> 16) & 1)
        print("XCD 0: mean start offset by tg_id parity:", [float((g[:, 0] - g[:, 0].min())[tg == k].mean()) if (tg == k).any() else None for k in (0, 1)])
    t0 = t[:, 0].min()
    nst = int((t[0] > 0).sum())
    d = np.diff(t[:, :nst], axis=1)
    print("workgroups traced", len(t), "stamps per WG", nst)
    print("median phase durations (ticks):", [int(v) for v in np.median(d, axis=0)])
    st = np.sort(t[:, 0]) - t0
    print("WG start times: 0..15:", [int(v) for v in st[:16]], " #512:", int(st[min(512, len(st) - 1)]), " #1024:", int(st[min(1024, len(st) - 1)]), " last:", int(st[-1]), " last end:", int(t[:, nst - 1].max() - t0))
    tot = t[:, nst - 1] - t[:, 0]
    print("per-WG total ticks: median %d min %d max %d" % (np.median(tot), tot.min(), tot.max()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pair":
        pair(*sys.argv[2:])
    elif len(sys.argv) > 1 and sys.argv[1] == "all":
        all_ops(*sys.argv[2:3])
    elif len(sys.argv) > 1 and sys.argv[1] == "trace":
        trace(*(sys.argv[2:4]))
    else:
        main()

"""Calibration of the planner's weight-gradient cost model (graph.MEGA_COST): every SSDN_OP_WGRAD op of the BASELINE config-2 plan
launched ALONE with the grid the chip-wide plan gave it (one block per CU, so the launch time is the time of one block) next to
the model's figure, then the chip-wide launch itself.  GPU only.  usage: python tools/wgrad_calib.py [ghz]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "selfsupervised-denoising_amd")]
import torch
from ssdn.hip import lib as L
from ssdn.hip import graph as G
from ssdn.hip.engine import DeviceNet, OpList, current_stream
from ssdn.hip.graph import NetPlan

# the cost model is calibrated on ONE launch holding every op (the production plan, "split", runs two launches: WGRAD_MEGA=split)
G.WGRAD_MEGA = os.environ.get("WGRAD_MEGA", "all")


def time_list(ol, iters=10):
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
    ghz = float(sys.argv[1]) if len(sys.argv) > 1 else 2.1
    B, P = 32, 64
    cus = L.load().ssdn_device_cus()
    plan = NetPlan("m/", 3, 9, True, B, P, P, cus=cus)
    dev = torch.device("cuda:0")
    flat = torch.randn(plan.nparams, device=dev) * 0.05
    dn = DeviceNet(plan, dev, flat, torch.zeros_like(flat))
    for name, t in dn.t.items():
        if t.dtype in (torch.float16, torch.bfloat16):
            t.copy_(torch.randn(t.shape, device=dev) * 0.5)
    ops = [op for op in plan.bwd if op.type == "wgrad"]
    recs = [dn._mat(op) for op in ops]
    import ctypes as C
    tot_model = 0.0
    for op, rec in zip(ops, recs):
        a = op.a
        t = time_list(OpList([rec]))
        v = (C.c_int32 * 9)()
        inst = L.load().ssdn_wgrad_variant(C.byref(rec[1]), v)
        tile, ntiles, c_tile, fixed = plan._mega_candidates(a)
        rounds = -(-ntiles // a["nslabs"])
        print("%-18s H=%3d K=%3d M=%3d taps=%d tile=%s ns=%3d mb=%d rounds=%3d inst=%3d %s : measured %8.1f us  model %8.1f us (per tile %.2f us, fixed %.1f us)  -> measured per tile %.2f us" % (
            a["layer"], a["H"], a["Ktot"], a["M"], len(a["taps"]), tile, a["nslabs"], a["mblocks"], rounds, inst, list(v), t, a["cost"] / ghz / 1e3,
            c_tile / ghz / 1e3, fixed / ghz / 1e3, (t - fixed / ghz / 1e3) / rounds))
    print("planned makespan (max, mean) us:", [(round(a / ghz / 1e3, 1), round(b / ghz / 1e3, 1)) for a, b in plan.mega_makespan.values()])
    print("chip-wide launch (all ops): %.1f us" % time_list(OpList(recs)))
    red = OpList([dn._mat(op) for op in plan.bwd if op.type == "wreduce"])
    print("slab reductions: %.1f us" % time_list(red))


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] in ("subsets", "trace", "load", "mega_only")):
    main()


def subsets():
    """chip-wide launches of subsets of the ops (diagnosis aid)"""
    B, P = 32, 64
    cus = L.load().ssdn_device_cus()
    plan = NetPlan("m/", 3, 9, True, B, P, P, cus=cus)
    dev = torch.device("cuda:0")
    flat = torch.randn(plan.nparams, device=dev) * 0.05
    dn = DeviceNet(plan, dev, flat, torch.zeros_like(flat))
    ops = [op for op in plan.bwd if op.type == "wgrad"]
    recs = [dn._mat(op) for op in ops]
    name = lambda i: "%s/K%d" % (ops[i].a["layer"].replace("code_block_", "").replace("output_block", "out"), ops[i].a["Ktot"])  # noqa: E731
    sets = {"dec1.2+dec1.0": [3, 4], "out0+out2": [1, 2], "thin x2": [5, 24], "dec2.2+dec2.0": [6, 7], "enc1.2+enc2.0": [22, 23],
            "static only": [i for i in range(len(ops)) if i in (3, 4, 6, 7, 8, 9, 10, 11, 21, 22, 23)],
            "generic only": [12, 13, 14, 15, 16, 17, 18, 19, 20], "out4+out2": [0, 1], "all but generic": [i for i in range(len(ops)) if i not in (12, 13, 14, 15, 16, 17, 18, 19, 20)]}
    for k, idx in sets.items():
        t = time_list(OpList([recs[i] for i in idx]))
        alone = [time_list(OpList([recs[i]])) for i in idx]
        print("%-16s %s: merged %.1f us; alone %s" % (k, [name(i) for i in idx], t, [round(x, 1) for x in alone]), flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "subsets":
    subsets()


def trace():
    """per-block timeline of the chip-wide launch from the s_memrealtime stamps of k_wgrad_mega (100 MHz)"""
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
    ops = [op for op in plan.bwd if op.type == "wgrad"]
    if G.WGRAD_MEGA == "split":          # the production plan: trace the chip-wide launch (group 1) alone
        ops = [op for op in ops if plan.wgrad_group_of(op.a["layer"], plan.is_skip_half(op)) == 1]
    ol = OpList([dn._mat(op) for op in ops])
    for _ in range(3):
        ol.run(current_stream())
    buf = torch.zeros(3 * 4096, dtype=torch.int64, device=dev)
    lib.ssdn_debug_set_trace.argtypes = [C.c_void_p]
    lib.ssdn_debug_set_trace(C.c_void_p(buf.data_ptr()))
    ol.run(current_stream())
    torch.cuda.synchronize()
    lib.ssdn_debug_set_trace(None)
    t = buf.cpu().view(-1, 3).numpy()
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    tick = 0.01        # us per s_memtime tick (100 MHz constant clock)
    print("blocks %d; launch span %.1f us; last start %.1f us" % (len(t), (t[:, 1].max() - t0) * tick, (t[:, 0].max() - t0) * tick))
    ends = np.sort((t[:, 1] - t0) * tick)
    print("CU-time used %.0f us x CU of %.0f available (%.1f %%); block end times: 10%% %.0f, 50%% %.0f, 90%% %.0f, max %.0f us" % (
        ((t[:, 1] - t[:, 0]) * tick).sum(), 256 * ends[-1], 100 * ((t[:, 1] - t[:, 0]) * tick).sum() / (256 * ends[-1]),
        ends[len(ends) // 10], ends[len(ends) // 2], ends[9 * len(ends) // 10], ends[-1]))
    for e, op in enumerate(ops):
        sel = t[(t[:, 2] & 0xffff) == e]
        if not len(sel):
            continue
        d = (sel[:, 1] - sel[:, 0]) * tick
        a = op.a
        tile, ntiles, c_tile, fixed = plan._mega_candidates(a)
        rounds = -(-ntiles // a["nslabs"])
        print("%-18s K=%3d tile=%s ns=%3d mb=%d rounds=%3d: block time median %7.1f us (min %7.1f max %7.1f)  start %6.1f..%6.1f  end max %6.1f | model cost %7.1f us, per tile: model %.2f measured %.2f us" % (
            a["layer"], a["Ktot"], tile, a["nslabs"], a["mblocks"], rounds, np.median(d), d.min(), d.max(), (sel[:, 0].min() - t0) * tick,
            (sel[:, 0].max() - t0) * tick, (sel[:, 1].max() - t0) * tick, a["cost"] / 2.1e3, c_tile / 2.1e3, (np.median(d) - fixed / 2.1e3) / rounds))


if len(sys.argv) > 1 and sys.argv[1] == "trace":
    trace()


def load():
    """every op as a launch of its own that fills the chip (one block per CU): per-tile time under homogeneous full load"""
    from ssdn.hip.graph import Op
    B, P = 32, 64
    lib = L.load()
    cus = lib.ssdn_device_cus()
    plan = NetPlan("m/", 3, 9, True, B, P, P, cus=cus)
    dev = torch.device("cuda:0")
    flat = torch.randn(plan.nparams, device=dev) * 0.05
    dn = DeviceNet(plan, dev, flat, torch.zeros_like(flat))
    for name, t in dn.t.items():
        if t.dtype in (torch.float16, torch.bfloat16):
            t.copy_(torch.randn(t.shape, device=dev) * 0.5)
    seen = set()
    for op in [o for o in plan.bwd if o.type == "wgrad"]:
        a = dict(op.a)
        tile, ntiles, c_tile, fixed = plan._mega_candidates(a)
        key = (a["H"], a["Ktot"], a["M"], len(a["taps"]), a["mblocks"])
        if key in seen:
            continue
        seen.add(key)
        mb, G = max(1, a["mblocks"]), max(1, a["csplit"])
        for ns in sorted({min(ntiles, cus // (mb * G)), min(ntiles, cus // (2 * mb * G))}, reverse=True):
            if False:
                continue
            a2 = dict(a)
            a2["nslabs"] = ns
            a2["mega"] = 0
            nt = len(a["taps"])
            dn.t["tmp_slab"] = torch.zeros(mb * ns * nt * a["Mpad"] * a["Kpad"], device=dev)
            dn.t["tmp_bslab"] = torch.zeros(mb * ns * a["Mpad"], device=dev)
            a2["slab"], a2["bslab"] = "tmp_slab", "tmp_bslab"
            t = time_list(OpList([dn._mat(Op("wgrad", a2))]))
            rounds = -(-ntiles // ns)
            print("%-18s H=%3d K=%3d M=%3d taps=%d tile=%s mb=%d G=%d ns=%3d rounds=%3d: %7.1f us; per tile %.2f us (model %.2f), fixed model %.1f us" % (
                a["layer"], a["H"], a["Ktot"], a["M"], nt, tile, mb, G, ns, rounds, t, (t - fixed / 2.1e3) / rounds, c_tile / 2.1e3, fixed / 2.1e3), flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "load":
    load()


def mega_only():
    """a few chip-wide launches + reductions, nothing else (for rocprofv3 --pmc passes: tools/pmc_mega.sh)"""
    B, P = 32, 64
    cus = L.load().ssdn_device_cus()
    plan = NetPlan("m/", 3, 9, True, B, P, P, cus=cus)
    dev = torch.device("cuda:0")
    flat = torch.randn(plan.nparams, device=dev) * 0.05
    dn = DeviceNet(plan, dev, flat, torch.zeros_like(flat))
    for name, t in dn.t.items():
        if t.dtype in (torch.float16, torch.bfloat16):
            t.copy_(torch.randn(t.shape, device=dev) * 0.5)
    ops = [op for op in plan.bwd if op.type == "wgrad"]
    ol = OpList([dn._mat(op) for op in ops])
    red = OpList([dn._mat(op) for op in plan.bwd if op.type == "wreduce"])
    big = [OpList([dn._mat(op)]) for op in ops if op.a["layer"] in ("decode_block_1.2", "output_block.0", "encode_block_1.2")]
    for _ in range(4):
        ol.run(current_stream())
        red.run(current_stream())
    for o in big:
        o.run(current_stream())
    torch.cuda.synchronize()


if len(sys.argv) > 1 and sys.argv[1] == "mega_only":
    mega_only()

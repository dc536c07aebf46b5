"""Isolated timing of the chained small forward layers (k_conv_chain) against the separate launches, BASELINE config 2 shapes.
usage: python tools/chain_bench.py [B P]   (TUNING build: SSDN_CHAIN_ABLATE = 1 no MFMA loop, 2 no HBM stores, 4 no pool, 8 no weight stream)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "selfsupervised-denoising_amd"), ROOT]
import torch
from ssdn.hip import lib as L
from ssdn.hip.engine import DeviceNet, OpList, current_stream
from ssdn.hip.graph import NetPlan

B, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 64)
if os.environ.get("SSDN_LIB"):            # a -DSSDN_TUNING build of the library (ablation bits)
    L.LIB_PATH = os.environ["SSDN_LIB"]
lib = L.load()
dev = torch.device("cuda:0")
plan = NetPlan("m/", 3, 9, True, B, P, P, cus=lib.ssdn_device_cus())
g = torch.Generator().manual_seed(5)
flat = (torch.randn(plan.nparams, generator=g) * 0.08).to(dev)
dn = DeviceNet(plan, dev, flat, torch.zeros_like(flat))
dn.t["m/in32"].copy_(torch.rand(dn.t["m/in32"].shape, generator=g).to(dev))
dn.pack.run(current_stream())
DIR = sys.argv[3] if len(sys.argv) > 3 else "fwd"
if DIR == "fwd":
    full, ops = dn.fwd, plan.fwd
else:
    dn.fwd.run(current_stream())
    dn.t["m/g32"].copy_((torch.randn(dn.t["m/g32"].shape, generator=g) * 1e-3).to(dev))
    full, ops = dn.bwd, None
full.run(current_stream())
best, at = 0, 0
for i in range(full.n):
    n = lib.ssdn_chain_len(C.byref(full.arr, i * C.sizeof(L.OpRec)), full.n - i)
    if n > best:
        best, at = n, i
print("chain of %d ops at op %d" % (best, at))
ol = OpList([(next(k for k, v in L.OP.items() if v == full.arr[at + j].type), full.args[at + j]) for j in range(best)])
for on in (0, 1, 0, 1):
    lib.ssdn_conv_set_chain(on)
    for _ in range(20):
        ol.run(current_stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ol.run(current_stream())
    e1.record()
    torch.cuda.synchronize()
    print("chain %d: %.2f us per pass" % (on, e0.elapsed_time(e1) * 1e3 / 200))

# cold caches: 768 MB of writes between the passes evict the weights (and everything else) from L2 and the Infinity Cache -- in a training
# step the chain's weights were written ~0.3 ms and ~0.5 GB of traffic earlier (CHAIN_BENCH_COLD=1)
if os.environ.get("CHAIN_BENCH_COLD"):
    junk = torch.empty(768 << 20, dtype=torch.uint8, device=dev)
    lib.ssdn_conv_set_chain(1)
    for cold in (0, 1, 0, 1):
        ts = []
        for _ in range(30):
            if cold:
                junk.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ol.run(current_stream())
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print("chained, %s caches: median %.2f us per pass (single bracketed launches: includes ~10 us of event overhead)" % ("COLD" if cold else "warm", ts[len(ts) // 2]))

if os.environ.get("SSDN_LIB"):
    lib.ssdn_debug_set_trace.argtypes = [C.c_void_p]
    N = 4 * B
    tr = torch.zeros(N * 96, dtype=torch.int64, device=dev)      # CH_TRACE stamps per workgroup (csrc/conv_chain.hip)
    lib.ssdn_conv_set_chain(1)
    lib.ssdn_debug_set_trace(tr.data_ptr())
    ol.run(current_stream())
    torch.cuda.synchronize()
    lib.ssdn_debug_set_trace(None)
    t = tr.cpu().view(N, 96)
    nst = int((t[0] != 0).sum())
    d = (t[:, 1:nst] - t[:, :nst - 1]).double()
    # the stamps of wave 0 (csrc/conv_chain.hip): a convolution -- per work item of the wave (items wave, wave + 4, ..) and 48-channel chunk
    # one after the B-fragment pipeline is filled and one after the 27 K-steps, then four behind the items; a POOL_BWD op -- one
    names = ["start: first weights + warm-up requested", "start: arena cleared, barrier", "start: planes loaded", "start: warm-up arrived, barrier"]
    for i in range(best):
        ar = ol.args[i]
        if not hasattr(ar, "Ktot"):
            names += ["L%d pool_bwd" % i]
            continue
        nch = ar.Ktot // 48
        nitems = (ar.Mpad // 32) * (2 if ar.H * ar.W > 128 else 1)
        for it in range((nitems + 3) // 4):
            names += sum([["L%d item %d chunk %d: way to the first MFMA" % (i, it, k), "L%d item %d chunk %d: 27 K-steps" % (i, it, k)] for k in range(nch)], [])
        names += ["L%d epilogue of the last item" % i, "L%d barrier" % i, "L%d second pass + store" % i, "L%d pool / upsum" % i]
    for i in range(best):
        ar = ol.args[i]
        print("  op %2d %-10s %s" % (i, next(k for k, v in L.OP.items() if v == full.arr[at + i].type),
              " ".join("%s=%s" % (f, getattr(ar, f)) for f in ("N", "H", "W", "Ktot", "M", "Mpad", "ntaps", "shifted") if hasattr(ar, f))))
    print("stamps per workgroup: %d; s_memtime ticks (mean over workgroups, min, max):" % nst)
    for i in range(nst - 1):
        print("  %-44s %8.1f %6d %6d" % (names[i] if i < len(names) else "?", float(d[:, i].mean()), int(d[:, i].min()), int(d[:, i].max())))
    print("  total %.1f ticks" % float((t[:, nst - 1] - t[:, 0]).double().mean()))

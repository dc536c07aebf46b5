"""Two builds of the library on the same inputs: which tensors do the conv launches of the training plan write differently, and where?
(bring-up aid, GPU only).  usage: SSDN_HIP_LIB=<lib> python tools/cmp_libs.py dump <file> [B] [P] [layer ...]   |   python tools/cmp_libs.py diff <fileA> <fileB>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "selfsupervised-denoising_amd")]
import torch


def dump(path, B=32, P=64, want=()):
    from ssdn.hip import lib as L
    from ssdn.hip.engine import DeviceNet, OpList, current_stream
    from ssdn.hip.graph import NetPlan
    lib = L.load()
    plan = NetPlan("m/", 3, 9, True, B, P, P, cus=lib.ssdn_device_cus())
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    flat = torch.randn(plan.nparams, device=dev) * 0.05
    dn = DeviceNet(plan, dev, flat, torch.zeros_like(flat))
    g = torch.Generator(device=dev).manual_seed(7)
    for name, t in dn.t.items():
        if t.dtype in (torch.float16, torch.bfloat16):
            t.copy_(torch.randn(t.shape, device=dev, generator=g) * 0.5)
        elif t.dtype == torch.uint8:
            t.copy_(torch.randint(0, 256, t.shape, device=dev, dtype=torch.uint8, generator=g))
    dn.pack.run(current_stream())
    torch.cuda.synchronize()
    init = {k: v.clone() for k, v in dn.t.items()}
    out = {}
    for i, op in enumerate(plan.fwd + plan.bwd):
        a = op.a
        if op.type != "conv" or len(a["taps"]) != 9 or a["Mpad"] < 96 or a["H"] < 32 or (want and a["layer"] not in want):
            continue
        for k, v in init.items():
            dn.t[k].copy_(v)
        OpList([dn._mat(op)]).run(current_stream())
        torch.cuda.synchronize()
        for k in init:
            if not torch.equal(dn.t[k], init[k]):
                out["%s/%s/%s" % (a["layer"], a["role"], k)] = dn.t[k].cpu()
    torch.save(out, path)
    print("dumped", len(out), "tensors to", path)


def diff(pa, pb):
    A, Bd = torch.load(pa), torch.load(pb)
    bad = 0
    for k in sorted(set(A) | set(Bd)):
        if k not in A or k not in Bd:
            print(k, "only in one dump"); bad += 1; continue
        a, b = A[k], Bd[k]
        if torch.equal(a.view(torch.uint8), b.view(torch.uint8)):
            print("%-50s identical" % k); continue
        bad += 1
        af, bf = a.float(), b.float()
        ne = af != bf
        print("%-50s %d of %d differ, max abs diff %g" % (k, int(ne.sum()), ne.numel(), float((af - bf).abs().max())))
        if af.dim() == 4:
            px = ne.any(dim=3)
            cnt_r = [int(px[:, r::16, :].sum()) for r in range(16)]
            cnt_c = [int(px[:, :, c::16].sum()) for c in range(16)]
            cnt_ch = [int(ne[..., 8 * j:8 * j + 8].sum()) for j in range(ne.shape[3] // 8)][:16]
            print("      differing pixels by row %% 16: %s\n      by col %% 16: %s\n      differing elements by 8-channel piece: %s" % (cnt_r, cnt_c, cnt_ch))
            print("      images:", px.any(dim=2).any(dim=1).nonzero().view(-1).tolist()[:24], " rows:", px.any(dim=2).any(dim=0).nonzero().view(-1).tolist()[:48],
                  " cols:", px.any(dim=1).any(dim=0).nonzero().view(-1).tolist()[:48], " channels:", ne.any(dim=0).any(dim=0).any(dim=0).nonzero().view(-1).tolist()[:24])
    print("cmp_libs:", "identical" if not bad else "%d tensors differ" % bad)


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        a = sys.argv[3:]
        dump(sys.argv[2], int(a[0]) if a else 32, int(a[1]) if len(a) > 1 else 64, a[2:])
    else:
        diff(sys.argv[2], sys.argv[3])

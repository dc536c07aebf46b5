"""GPU parity tests, called through the C-ABI (libssdn_hip.so):

* hardware probes pin the MFMA / LDS-transpose lane maps the kernels are written against;
* every planned op of a whole forward+backward is checked IN ISOLATION ("teacher forcing"): its inputs are uploaded from
  the CPU interpreter (oracle/interp.py, fp16-storage emulation), the single op runs on the device, and its output must
  match the interpreter's to fp16 rounding -- so one wrong kernel cannot hide behind, or be blamed for, another;
* end-to-end forward / parameter gradients are compared with the fp32 oracle (oracle/restate.py) and with the golden
  vectors generated from the reference (tests/golden).
Tolerances are written next to each assertion.
"""
import os

import numpy as np
import pytest
import torch

import restate as R
from interp import Interp
from test_lowering_cpu import flat_params

pytestmark = pytest.mark.gpu

OUTDIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def dev():
    return torch.device("cuda:0")


# ------------------------------------------------------------------------------------------------------------
# probes
# ------------------------------------------------------------------------------------------------------------
def test_probe_mfma_layout():
    """v_mfma_f32_32x32x16_f16: A[i][k] lane = i + 32*(k//8), B[k][j] lane = j + 32*(k//8), D[i][j] in lane j + 32*((i//4)%2),
    register 4*(i//8) + i%4.  Asymmetric operands so a transposed result cannot pass."""
    import ctypes as C
    from ssdn.hip import lib as L
    lib = L.load()
    rng = np.random.RandomState(0)
    A = rng.randint(-3, 4, size=(32, 16)).astype(np.float32)
    Bm = rng.randint(-3, 4, size=(16, 32)).astype(np.float32)
    a_frag = np.zeros((64, 8), np.float16)
    b_frag = np.zeros((64, 8), np.float16)
    for lane in range(64):
        for e in range(8):
            a_frag[lane, e] = A[lane & 31, (lane >> 5) * 8 + e]
            b_frag[lane, e] = Bm[(lane >> 5) * 8 + e, lane & 31]
    da, db = torch.from_numpy(a_frag).to(dev()), torch.from_numpy(b_frag).to(dev())
    dd = torch.zeros(64, 16, device=dev())
    L.check(lib.ssdn_probe_mfma(C.c_void_p(da.data_ptr()), C.c_void_p(db.data_ptr()), C.c_void_p(dd.data_ptr()), None))
    torch.cuda.synchronize()
    d = dd.cpu().numpy()
    D = A @ Bm
    got = np.zeros((32, 32), np.float32)
    for lane in range(64):
        for r in range(16):
            got[8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), lane & 31] = d[lane, r]
    np.testing.assert_array_equal(got, D)


def test_probe_tr16_mapping():
    """ds_read_b64_tr_b16: in each 16-lane group lane i supplies row i>>2, column chunk i&3 of a 4x16 matrix and receives
    column i.  The raw result is dumped to gpurun_out/tr16_probe.txt for diagnosis."""
    import ctypes as C
    from ssdn.hip import lib as L
    lib = L.load()
    # LDS image: 64 rows x 16 halves (32 B per row); value = row*16 + col
    img = (np.arange(64 * 16).reshape(64, 16)).astype(np.float16)
    addr = np.zeros(64, np.int32)
    for lane in range(64):
        g, i = lane >> 4, lane & 15
        row = g * 4 + (i >> 2)
        addr[lane] = row * 32 + (i & 3) * 8
    dimg = torch.from_numpy(img).to(dev())
    daddr = torch.from_numpy(addr).to(dev())
    dout = torch.zeros(64, 4, dtype=torch.float16, device=dev())
    L.check(lib.ssdn_probe_tr16(C.c_void_p(dimg.data_ptr()), img.nbytes, C.c_void_p(daddr.data_ptr()), C.c_void_p(dout.data_ptr()), None))
    torch.cuda.synchronize()
    out = dout.cpu().numpy().astype(np.int32)
    os.makedirs(OUTDIR, exist_ok=True)
    with open(os.path.join(OUTDIR, "tr16_probe.txt"), "w") as f:
        for lane in range(64):
            f.write("lane %2d addr row %2d chunk %d -> %s\n" % (lane, addr[lane] // 32, (addr[lane] % 32) // 8,
                                                              ["(r%d,c%d)" % (v // 16, v % 16) for v in out[lane]]))
    exp = np.zeros((64, 4), np.int32)
    for lane in range(64):
        g, i = lane >> 4, lane & 15
        for j in range(4):
            exp[lane, j] = (g * 4 + j) * 16 + i
    np.testing.assert_array_equal(out, exp)


# ------------------------------------------------------------------------------------------------------------
# per-op teacher-forced parity of a whole planned forward + backward
# ------------------------------------------------------------------------------------------------------------
def _upload(dn, it):
    for name, t in it.t.items():
        if name in dn.t:
            dn.t[name].copy_(t.to(dn.t[name].dtype).reshape(dn.t[name].shape))


def _out_of(op):
    a = op.a
    if op.type == "conv":
        if a.get("unrot") is not None:          # fused UNROT_BWD: the launch's only output
            return ("act", a["unrot"], a["M"] // 4)
        if a.get("urot") is not None:           # fused UNROT_FWD: the launch's only 16-bit output (+ sign bytes)
            return ("act", a["urot"], 4 * a["M"])
        return ("f32", a["dst32"], None) if a["dst32"] is not None else ("act", a["dst"], a["M"])
    if op.type == "pool_fwd":
        return ("act", a["pooled"], a["C"])
    if op.type == "pool_bwd":
        return ("act", a["dz"], a["C"])
    if op.type in ("upsum_bwd",):
        return ("act", a["dst"], a["C"])
    if op.type == "unrot_fwd":
        return ("act", a["dst"], 4 * a["C"])
    if op.type == "unrot_bwd":
        return ("act", a["dst"], a["C"])
    if op.type in ("pack_input", "grad_pack"):
        return ("act", a["dst"], a["cpad"])
    return None


# the (3, 9, True, 8, 32) case makes the planner pick multi-image tiles with one-row halos (TH = 1, TN > 1) for the 1x1 head
# (cin, cout, blindspot, B, P, cus): cus = 0 plans for the device's CU count; a small value plans persistent grids of that
# many workgroups, so that the multi-tile paths of the weight-gradient kernel (double-buffered prefetch, compile-time K-step
# schedule of the hot shapes) run at test sizes too -- at cus = 256 they only trigger from BASELINE config 2 upwards
# conv_mode (last field): 1 = the library's default kernel choice; 2 = the persistent LDS-DMA kernel k_cdma for every 3x3
# layer of its shape class (>= 16x16 pixels), which by default only serves layers with >= 1 tile per CU (BASELINE sizes)
# (3, 9, True, 4, 64, 0, 1): 256 16x16 tiles at full resolution = one per CU of an MI355X, from where on decode_block_1.2 stores its output
# un-rotated (fused SSDN_OP_UNROT_FWD in k_cdma) and the data gradients of the 64x64 stage fuse SSDN_OP_UPSUM_BWD; k_conv_thin and k_cdma
# leave LeakyReLU sign bytes of e0 / d1a and the data gradients of encode_block_1.2 / decode_block_1.2 read them (ssdn_conv_args.sign_out /
# mask_sign).  (3, 9, True, 16, 64, 0, 1): one tile per CU at the 32x32 stage too -- sign bytes of d2a / d2b, the fused UPSUM_BWD of
# decode_block_1.0's data gradient reads its mask as sign bytes (upsum_mask_sign)
CASES = [(3, 9, True, 2, 32, 0, 1), (1, 2, True, 1, 32, 0, 1), (3, 9, True, 4, 64, 0, 1), (3, 9, True, 16, 64, 0, 1), (3, 3, False, 2, 32, 0, 1), (3, 9, True, 1, 64, 0, 1), (3, 1, False, 2, 64, 0, 1),
         (3, 9, True, 8, 32, 0, 1), (3, 9, True, 2, 32, 8, 1), (3, 3, False, 2, 64, 6, 1),
         (3, 9, True, 2, 32, 0, 2), (1, 2, True, 1, 32, 0, 2), (3, 3, False, 2, 64, 0, 2), (3, 9, True, 3, 64, 0, 2), (3, 1, False, 5, 32, 0, 2)]


@pytest.fixture
def conv_mode_reset():
    yield
    from ssdn.hip import lib as L
    L.load().ssdn_conv_set_mode(1)


@pytest.mark.parametrize("cin,cout,bs,B,P,cus_plan,conv_mode", CASES)
def test_every_op_teacher_forced(cin, cout, bs, B, P, cus_plan, conv_mode, conv_mode_reset):
    from ssdn.hip.engine import DeviceNet, OpList, current_stream
    from ssdn.hip.graph import NetPlan
    from ssdn.hip import lib as L
    L.check(L.load().ssdn_conv_set_mode(conv_mode))
    cus = cus_plan or L.load().ssdn_device_cus()
    p = R.make_params(cin, cout, bs, seed=7)
    plan = NetPlan("m/", cin, cout, bs, B, P, P, cus=cus, dev_cus=L.load().ssdn_device_cus())
    flat = flat_params(plan, p)
    it = Interp(plan, flat, fp16=True)
    it.t["m/in32"] = R.hash_tensor((B, cin, P, P), 91, 0, 1)
    it.run(plan.pack)
    it.run(plan.fwd)
    it.t["m/g32"] = R.hash_tensor((B, cout, P, P), 92, -1, 1) * 1e-3
    it.run(plan.bwd)

    dparams = flat.to(dev())
    dgrads = torch.zeros_like(dparams)
    dn = DeviceNet(plan, dev(), dparams, dgrads)
    dn.t["m/gmax"][0] = int(np.float32(it.t["m/g32"].abs().max()).view(np.int32))
    dn.pack.run(current_stream())
    torch.cuda.synchronize()
    failures = []
    # packed weights first
    for l in plan.layers:
        for kind in ("wf/", "wd/"):
            name = "m/" + kind + l.name
            if name in it.t:
                got = dn.t[name].float().cpu().reshape(it.t[name].shape)
                if not torch.equal(got, it.t[name]):
                    failures.append("%s: packed weights differ (max %g)" % (name, float((got - it.t[name]).abs().max())))
    _upload(dn, it)
    ops = plan.fwd + plan.bwd
    i = 0
    while i < len(ops):
        op = ops[i]
        if op.type == "wgrad":
            nred = op.a.get("mblocks", 1)            # a merged launch is followed by one reduction per block of output channels
            group = [dn._mat(op)] + [dn._mat(ops[i + 1 + k]) for k in range(nred)]
            l = next(x for x in plan.layers if x.name == ops[i + 1].a["layer"])
            dgrads.fill_(float("nan"))
            dn.t["m/scale"][0] = it.scale
            dn.t["m/scale"][1] = 1.0 / it.scale
            OpList(group).run(current_stream())
            torch.cuda.synchronize()
            gw = dgrads[l.w_off:l.w_off + l.M * l.cin * l.ntaps].cpu().reshape(l.M, l.cin, l.ntaps)
            rw = it.grads[l.w_off:l.w_off + l.M * l.cin * l.ntaps].reshape(l.M, l.cin, l.ntaps)
            for k in range(nred):
                a2 = ops[i + 1 + k].a
                sl = (slice(a2["m_off"], a2["m_off"] + a2["M"]), slice(a2["c_off"], a2["c_off"] + a2["cin"]))
                g, r = gw[sl], rw[sl]
                # fp32 accumulation over up to 2^17 fp16 products in a different order: 1e-3 of the block's scale
                tol = 1e-3 * float(r.abs().max()) + 1e-12
                err = float((g - r).abs().max()) if torch.isfinite(g).all() else float("inf")
                if not err <= tol:
                    failures.append("op %d wgrad %s m_off %d c_off %d: max err %.3e (tol %.3e)" % (i, l.name, a2["m_off"], a2["c_off"], err, tol))
                if a2["with_bias"]:
                    gb = dgrads[l.b_off + a2["m_off"]: l.b_off + a2["m_off"] + a2["M"]].cpu()
                    rb = it.grads[l.b_off + a2["m_off"]: l.b_off + a2["m_off"] + a2["M"]]
                    tolb = 1e-3 * float(rb.abs().max()) + 1e-12
                    errb = float((gb - rb).abs().max()) if torch.isfinite(gb).all() else float("inf")
                    if not errb <= tolb:
                        failures.append("op %d bias-grad %s: max err %.3e (tol %.3e)" % (i, l.name, errb, tolb))
            i += 1 + nred
            continue
        kind, dst, ch = _out_of(op)
        rec = dn._mat(op)
        if kind == "f32":
            dn.t[dst].fill_(float("nan"))
        else:
            dn.t[dst.t][..., dst.co:dst.co + ch] = float("nan")
        uv = op.a.get("upsum") if op.type == "conv" else None
        if uv is not None:
            dn.t[uv.t][..., uv.co:uv.co + op.a["upsum_c"]] = float("nan")
        pv = op.a.get("pool") if op.type == "conv" else None
        if pv is not None:
            dn.t[pv.t][..., pv.co:pv.co + ch] = float("nan")
        smk = op.a.get("smask") if op.type == "unrot_fwd" else ((op.a.get("urot_smask") or op.a.get("sign_out")) if op.type == "conv" else None)
        if smk is not None:
            dn.t[smk].fill_(0xA5)
        if op.type == "pool_fwd" and op.a.get("route"):
            dn.t[op.a["route"]].fill_(-1)
        OpList([rec]).run(current_stream())
        torch.cuda.synchronize()
        if op.type == "conv" and op.a.get("urot") is not None:
            # fused UNROT_FWD: the rows the one-row shift leaves empty are not the launch's to write (they stay as they were: zero
            # in the zero-initialised tensor of a real run, NaN here); everything else is compared below
            t, Cq, Pq = dn.t[dst.t], op.a["M"], op.a["H"]
            empty = [t[:, 0, :, 0:Cq], t[:, :, Pq - 1, Cq:2 * Cq], t[:, Pq - 1, :, 2 * Cq:3 * Cq], t[:, :, 0, 3 * Cq:4 * Cq]]
            if not all(bool(torch.isnan(e.float()).all()) for e in empty):
                failures.append("op %d conv %s: fused UNROT_FWD wrote into the rows the shift leaves empty" % (i, op.a["layer"]))
            for e in empty:
                e.zero_()
        if smk is not None:
            # LeakyReLU sign bytes of the un-rotated tensor; row P-1 is not the kernel's to write.  SSDN_OP_UNROT_FWD: exact (its input
            # is the forced one).  Fused into the conv: exact against the signs of what the launch itself stored (un-rotated back)
            gots, wants = dn.t[smk].cpu(), it.t[smk]
            if op.type == "conv" and op.a.get("sign_out"):
                # sign_out: exactly the signs of the output the launch itself stored, every pixel
                from interp import _pack_signs
                own = _pack_signs(dn.t[dst.t][..., dst.co:dst.co + ch].float().cpu())
                if not torch.equal(gots, own):
                    failures.append("op %d conv %s: %d sign bytes differ from the signs of the stored output" % (i, op.a["layer"], int((gots != own).sum())))
                gots = wants = torch.full_like(gots, 0xA5)      # (nothing more to check below)
            elif op.type == "conv":
                Cq = op.a["M"]
                own = dn.t[dst.t][..., dst.co:dst.co + 4 * Cq].float().cpu()
                rows = torch.cat([it._rot(own[..., r * Cq:(r + 1) * Cq], ang)[:, 1:] for r, ang in enumerate((0, 90, 180, 270))], 0)
                pos = (rows > 0).to(torch.int64).reshape(*rows.shape[:3], Cq // 8, 8)
                wants = torch.cat([(pos << torch.arange(8)).sum(-1).to(torch.uint8), wants[:, -1:]], 1)
            if not torch.equal(gots[:, :-1], wants[:, :-1]):
                failures.append("op %d %s: %d sign bytes differ" % (i, op.type, int((gots[:, :-1] != wants[:, :-1]).sum())))
            if not bool((gots[:, -1] == 0xA5).all()):
                failures.append("op %d %s: sign bytes of the cut-off row were written" % (i, op.type))
            dn.t[smk].copy_(it.t[smk])
        if op.type == "pool_fwd" and op.a.get("route"):
            # route words of the max-pool (ssdn_pool_args.route): exact -- the op's input is the forced one
            gotr = dn.t[op.a["route"]].cpu().to(torch.int64) & 0xffffffff
            if not torch.equal(gotr, it.t[op.a["route"]]):
                failures.append("op %d pool_fwd: %d route words differ" % (i, int((gotr != it.t[op.a["route"]]).sum())))
        if pv is not None:
            # fused max-pool: exactly SSDN_OP_POOL_FWD of what the launch itself stored (max is exact on the rounded values)
            mine = dn.t[dst.t][..., dst.co:dst.co + ch].float().cpu()
            want = it._windows(mine, op.a["pool_shifted"]).max(3).values
            gotp = dn.t[pv.t][..., pv.co:pv.co + ch].float().cpu()
            if not torch.equal(gotp, want):
                failures.append("op %d conv %s: fused max-pool differs from pool(dst) in %d elements" % (
                    i, op.a["layer"], int((gotp != want).sum())))
            dn.t[pv.t][..., pv.co:pv.co + ch] = it.t[pv.t][..., pv.co:pv.co + ch].to(dev()).to(dn.t[pv.t].dtype)
        c_lo = 0
        if uv is not None:
            # fused UPSUM_BWD: the up-sampled-input channels only exist as 2x2 sums at half resolution (bf16); the rest in dst
            uc = op.a["upsum_c"]
            gotu, refu = dn.t[uv.t][..., uv.co:uv.co + uc].float().cpu(), it.t[uv.t][..., uv.co:uv.co + uc]
            # each of the 4 bf16 addends may differ by one ulp (2^-8 of ITS size, not of the possibly cancelling sum's) + the
            # sum's own rounding
            tolu = 1.6e-2 * refu.abs() + 2e-2 * float(refu.abs().mean()) + 1e-9
            badu = ~((gotu - refu).abs() <= tolu)
            if badu.any():
                failures.append("op %d conv %s: fused upsum: %d/%d elements off, max err %.3e" % (
                    i, op.a["layer"], int(badu.sum()), badu.numel(), float(torch.nan_to_num((gotu - refu).abs(), nan=9e9).max())))
            dn.t[uv.t][..., uv.co:uv.co + uc] = refu.to(dev()).to(dn.t[uv.t].dtype)
            c_lo = uc
            if c_lo >= ch:
                i += 1
                continue
        if kind == "f32":
            got, ref = dn.t[dst].cpu(), it.t[dst]
        else:
            got, ref = dn.t[dst.t][..., dst.co + c_lo:dst.co + ch].float().cpu(), it.t[dst.t][..., dst.co + c_lo:dst.co + ch]
        # one ulp of the storage type (fp16: 2^-10, bf16 gradients: 2^-7 relative) + accumulation-order noise
        ulp = 1.6e-2 if (kind == "act" and plan.tensors[dst.t].kind == "actb") else 2e-3
        tol = ulp * ref.abs() + ulp * float(ref.abs().max()) * 1e-2 + 1e-9
        if op.type == "conv" and op.a.get("add") is not None:
            # the kernel rounds the convolution result to bf16 BEFORE the skip gradient is added (the tile passes through
            # LDS as 16-bit): one extra rounding, relative to the addends rather than to their (possibly cancelling) sum
            ad = op.a["add"]
            addv = it.t[ad.t][..., ad.co:ad.co + ch].abs()
            tol = tol + ulp * (addv + (ref - it.t[ad.t][..., ad.co:ad.co + ch]).abs())
        bad = ~((got - ref).abs() <= tol)
        if bad.any():
            idx = bad.nonzero()[0].tolist()
            failures.append("op %d %s %s: %d/%d elements off, max err %.3e, first at %s got %g want %g" % (
                i, op.type, op.a.get("layer", ""), int(bad.sum()), bad.numel(),
                float(torch.nan_to_num((got - ref).abs(), nan=9e9).max()), idx, float(got[tuple(idx)]), float(ref[tuple(idx)])))
        # restore the forced value
        if kind == "f32":
            dn.t[dst].copy_(ref)
        else:
            dn.t[dst.t][..., dst.co + c_lo:dst.co + ch] = ref.to(dev()).to(dn.t[dst.t].dtype)
        i += 1
    os.makedirs(OUTDIR, exist_ok=True)
    with open(os.path.join(OUTDIR, "teacher_forced_%d_%d_%d_%d_%d_cus%d_mode%d.txt" % (cin, cout, int(bs), B, P, cus_plan, conv_mode)), "w") as f:
        f.write("\n".join(failures) if failures else "all %d ops OK\n" % len(ops))
    assert not failures, "\n".join(failures[:40])


# ------------------------------------------------------------------------------------------------------------
# end-to-end network forward / backward vs the fp32 oracle and the reference's golden vectors
# ------------------------------------------------------------------------------------------------------------
def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("tag,cin,cout,bs", [("bs_rgb", 3, 9, True), ("bs_mono", 1, 2, True), ("plain_rgb", 3, 3, False), ("sigma", 3, 1, False)])
def test_net_forward_vs_reference_golden(golden_dir, tag, cin, cout, bs):
    """fp16 storage / fp32 accumulation against the reference's fp32 output: relative L2 error <= 5e-3."""
    from ssdn.hip.engine import DeviceNet, current_stream
    from ssdn.hip.graph import NetPlan
    from ssdn.hip import lib as L
    g = np.load(os.path.join(golden_dir, "g_net_%s.npz" % tag))
    plan = NetPlan("m/", cin, cout, bs, 2, 32, 32, cus=L.load().ssdn_device_cus(), train=False)
    flat = flat_params(plan, R.make_params(cin, cout, bs, seed=3)).to(dev())
    dn = DeviceNet(plan, dev(), flat, None)
    dn.t["m/in32"].copy_(R.hash_tensor((2, cin, 32, 32), 31, 0, 1))
    dn.pack.run(current_stream())
    dn.fwd.run(current_stream())
    torch.cuda.synchronize()
    out = dn.t["m/out32"].cpu()
    assert _rel(out, torch.from_numpy(g["out"])) <= 5e-3


def _run_device_fwd_bwd(plan, p, x, g):
    from ssdn.hip.engine import DeviceNet, current_stream
    flat = flat_params(plan, p).to(dev())
    grads = torch.zeros_like(flat)
    dn = DeviceNet(plan, dev(), flat, grads)
    dn.t["m/in32"].copy_(x)
    dn.pack.run(current_stream())
    dn.fwd.run(current_stream())
    dn.t["m/g32"].copy_(g)
    dn.t["m/gmax"][0] = int(np.float32(g.abs().max()).view(np.int32))
    dn.bwd.run(current_stream())
    torch.cuda.synchronize()
    return dn, grads.cpu()


@pytest.mark.parametrize("cin,cout,bs,B,P", [(3, 9, True, 2, 32), (3, 3, False, 2, 64)])
def test_net_backward_end_to_end(cin, cout, bs, B, P):
    """Whole HIP forward+backward (no forcing) vs autograd of the fp32 oracle AND vs the CPU interpreter with fp16 storage:
    per-tensor cosine >= 0.99 and relative L2 <= 0.15 against both.  The floor is NOT arithmetic error but branch flips: an
    activation within one fp16 ulp of zero takes the other LeakyReLU slope (1 vs 0.1) / max-pool winner in ~1e-3 of the
    elements, each flip changes that element's gradient by 90% => sqrt(1e-3)*0.9 ~ 3e-2 per layer (measured 3e-2..9e-2),
    zero-mean noise far below minibatch gradient noise.  Even two fp16 executions that differ only in fp32 summation order
    (device vs interpreter) diverge this way once errors propagate, which is why the TIGHT statement about the kernels is
    the teacher-forced per-op test above, and this one only bounds the end-to-end effect."""
    from ssdn.hip.graph import NetPlan
    from ssdn.hip import lib as L
    p = R.make_params(cin, cout, bs, seed=7)
    plan = NetPlan("m/", cin, cout, bs, B, P, P, cus=L.load().ssdn_device_cus())
    x = R.hash_tensor((B, cin, P, P), 91, 0, 1)
    g = R.hash_tensor((B, cout, P, P), 92, -1, 1) * 1e-3
    dn, gh = _run_device_fwd_bwd(plan, p, x, g)
    it = Interp(plan, flat_params(plan, p), fp16=True)
    it.t["m/in32"] = x
    it.run(plan.pack)
    it.run(plan.fwd)
    it.t["m/g32"] = g
    it.run(plan.bwd)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ref = R.net_forward(leaves, x, bs)
    (ref * g).sum().backward()
    assert _rel(dn.t["m/out32"].cpu(), it.t["m/out32"]) <= 2e-3
    assert _rel(dn.t["m/out32"].cpu(), ref.detach()) <= 5e-3
    bad = []
    for l in plan.layers:
        for nm, sl, rg in (("w", slice(l.w_off, l.w_off + l.M * l.cin * l.ntaps), leaves[l.name + ".weight"].grad.reshape(-1)),
                           ("b", slice(l.b_off, l.b_off + l.M), leaves[l.name + ".bias"].grad)):
            a, ai = gh[sl], it.grads[sl]
            cos = float((a * rg).sum() / (a.norm() * rg.norm() + 1e-30))
            # bounds = 1.5 x the worst layer measured on the device (round 5: 0.058 vs the fp16 interpreter, 0.087 / cosine 0.9965 vs the fp32
            # oracle -- bf16 gradients of 1e-3-sized seeds through 20 layers at batch 2; the tight statements are the teacher-forced
            # per-op cases above and the full-size fixtures of tests/test_hip_fullsize.py)
            if not _rel(a, ai) <= 0.09:
                bad.append("%s.%s vs fp16 interpreter: rel %.3e" % (l.name, nm, _rel(a, ai)))
            if not (_rel(a, rg) <= 0.13 and cos >= 0.995):
                bad.append("%s.%s vs fp32 oracle: rel %.3e cos %.5f" % (l.name, nm, _rel(a, rg), cos))
    assert not bad, "\n".join(bad)


def test_full_size_config2_properties():
    """BASELINE config 2 at full size (B = 32, 64x64 RGB, blind-spot, 128 images inside the U-Net): the size at which the
    persistent multi-tile kernels and the compile-time-scheduled weight-gradient path run for real.  Checked through
    size-independent properties and against the fp32 oracle:
      * determinism: two executions give BIT-IDENTICAL parameter gradients (no atomics, fixed-order slab reduction);
      * linearity: doubling the upstream gradient doubles every parameter gradient EXACTLY (a power-of-two scale commutes
        with every bf16 / fp32 rounding on the way);
      * the forward output and every parameter gradient agree with autograd of the fp32 oracle within the end-to-end
        bounds of test_net_backward_end_to_end (cosine >= 0.99, relative L2 <= 0.15; forward 5e-3)."""
    from ssdn.hip.engine import current_stream
    from ssdn.hip.graph import NetPlan
    from ssdn.hip import lib as L
    cin, cout, bs, B, P = 3, 9, True, 32, 64
    p = R.make_params(cin, cout, bs, seed=11)
    plan = NetPlan("m/", cin, cout, bs, B, P, P, cus=L.load().ssdn_device_cus())
    x = R.hash_tensor((B, cin, P, P), 191, 0, 1)
    g = R.hash_tensor((B, cout, P, P), 192, -1, 1) * 1e-3
    dn, g1 = _run_device_fwd_bwd(plan, p, x, g)
    out1 = dn.t["m/out32"].cpu().clone()

    def rerun(gg):
        dn.grads.zero_()
        dn.fwd.run(current_stream())
        dn.t["m/g32"].copy_(gg)
        dn.t["m/gmax"][0] = int(np.float32(gg.abs().max()).view(np.int32))
        dn.bwd.run(current_stream())
        torch.cuda.synchronize()
        return dn.grads.cpu().clone()

    g1b = rerun(g)
    assert torch.equal(g1, g1b), "parameter gradients are not bit-reproducible"
    g2 = rerun(2.0 * g)
    assert torch.equal(g2, 2.0 * g1), "backward pass is not exactly linear in the upstream gradient"

    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ref = R.net_forward(leaves, x, bs)
    (ref * g).sum().backward()
    assert _rel(out1, ref.detach()) <= 5e-3
    bad = []
    for l in plan.layers:
        for nm, sl, rg in (("w", slice(l.w_off, l.w_off + l.M * l.cin * l.ntaps), leaves[l.name + ".weight"].grad.reshape(-1)),
                           ("b", slice(l.b_off, l.b_off + l.M), leaves[l.name + ".bias"].grad)):
            a = g1[sl]
            cos = float((a * rg).sum() / (a.norm() * rg.norm() + 1e-30))
            if not (_rel(a, rg) <= 0.15 and cos >= 0.99):
                bad.append("%s.%s vs fp32 oracle: rel %.3e cos %.5f" % (l.name, nm, _rel(a, rg), cos))
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("B,P,mode", [(2, 32, "all"), (32, 64, "all"), (2, 32, None), (4, 64, "buckets"), (32, 64, "split"), (4, 128, "split")])
def test_merged_weight_gradient_launch_is_bit_identical(B, P, mode, monkeypatch):
    """A run of consecutive SSDN_OP_WGRAD ops executes as ONE launch -- the chip-wide k_wgrad_mega (csrc/wgrad_mega.hip: one workgroup
    per CU works through a list of blocks of several ops' grids) or, for round 3's per-layer plans, k_wgrad_multi for the small layers;
    every block runs the code of its own op's launch, so the gradients must equal those of one-op-at-a-time execution bit for bit."""
    import ctypes as C
    from ssdn.hip import lib as L
    from ssdn.hip import graph as G
    from ssdn.hip.engine import DeviceNet, OpList, current_stream
    from ssdn.hip.graph import NetPlan
    monkeypatch.setattr(G, "WGRAD_MEGA", mode)
    monkeypatch.setattr(G, "MEGA_MIN_PX", 0)                  # (the small fixture too)
    dev = torch.device("cuda:0")
    plan = NetPlan("m/", 3, 9, True, B, P, P, cus=L.load().ssdn_device_cus())
    g = torch.Generator(device="cpu").manual_seed(11)
    flat = (torch.randn(plan.nparams, generator=g) * 0.05).to(dev)
    dn = DeviceNet(plan, dev, flat, torch.zeros_like(flat))
    for name, t in dn.t.items():
        if t.dtype in (torch.float16, torch.bfloat16):
            t.copy_((torch.randn(t.shape, generator=g) * 0.5).to(dev))
    ops = [op for op in plan.bwd if op.type == "wgrad"]
    assert len(ops) > 8
    recs = [dn._mat(op) for op in ops]
    red = OpList([dn._mat(op) for op in plan.bwd if op.type == "wreduce"])
    dn.t["m/scale"].fill_(1.0)                                # (written by SSDN_OP_GRAD_PACK in a real backward pass)

    def run(lists):
        # (rows of padded output channels hold whatever the LDS held: compare what the reductions read -- the real rows)
        dn.grads.fill_(float("nan"))
        for ol in lists:
            ol.run(current_stream())
        red.run(current_stream())
        torch.cuda.synchronize()
        return dn.grads.clone()

    one_by_one = run([OpList([r]) for r in recs])           # a run of one op is never merged
    merged = run([OpList(recs)])
    if mode:
        nmerge = sum(1 for r in recs if L.load().ssdn_wgrad_mega_ok(C.byref(r[1])))
        assert nmerge == len(recs), "every op of the BASELINE plan must have an instance in the chip-wide launch (%d of %d)" % (nmerge, len(recs))
    else:
        nmerge = sum(1 for r in recs if L.load().ssdn_wgrad_mergeable(C.byref(r[1])))
        assert nmerge >= 8, "the fixture must exercise the merged launch (%d mergeable ops)" % nmerge
    n = plan.nparams
    assert torch.isfinite(one_by_one[:n]).all()
    assert torch.equal(one_by_one[:n], merged[:n])
    # ... and against fp64 on the CPU for a layer with many pixels and one with few (the lowering itself: teacher-forced tests)
    again = run([OpList(recs)])
    assert torch.equal(merged[:n], again[:n]), "the merged launch must be reproducible bit for bit"


def _longest_chain(lib, ol):
    """longest run of ops the library executes as one k_conv_chain launch, over all start positions of the list"""
    import ctypes as C
    from ssdn.hip import lib as L
    best = 0
    for i in range(ol.n):
        n = lib.ssdn_chain_len(C.byref(ol.arr, i * C.sizeof(L.OpRec)), ol.n - i)
        assert n >= 0, lib.ssdn_last_error().decode()
        best = max(best, n)
    return best


@pytest.fixture
def conv_chain_reset():
    yield
    from ssdn.hip import lib as L
    L.load().ssdn_conv_set_chain(1)


# (cin, cout, blindspot, B, P, layers the chain must cover): 4 x B images; the chain starts at the first layer whose images have
# <= 64 pixels and whose max-pool the planner fused into the conv (whole-image tiles: N a multiple of 256 / pixels per image)
@pytest.mark.parametrize("cin,cout,bs,B,P,min_chain", [(3, 9, True, 4, 64, 7), (3, 9, True, 8, 32, 2), (3, 3, False, 16, 64, 7), (1, 2, True, 32, 64, 7)])
def test_conv_chain_is_bit_identical(cin, cout, bs, B, P, min_chain, conv_chain_reset):
    """A run of consecutive small forward layers executes as ONE launch with the activations resident in LDS (k_conv_chain,
    csrc/conv_chain.hip); every tensor the separate launches would have written -- each layer's output and each fused max-pool
    output -- must come out bit for bit the same."""
    import ctypes as C
    from ssdn.hip import lib as L
    from ssdn.hip.engine import DeviceNet, OpList, current_stream
    from ssdn.hip.graph import NetPlan
    lib = L.load()
    dev = torch.device("cuda:0")
    plan = NetPlan("m/", cin, cout, bs, B, P, P, cus=lib.ssdn_device_cus())
    g = torch.Generator(device="cpu").manual_seed(5)
    flat = (torch.randn(plan.nparams, generator=g) * 0.08).to(dev)
    dn = DeviceNet(plan, dev, flat, torch.zeros_like(flat))
    dn.t["m/in32"].copy_(torch.rand(dn.t["m/in32"].shape, generator=g).to(dev))
    dn.pack.run(current_stream())
    recs = [dn._mat(op) for op in plan.fwd]
    assert _longest_chain(lib, OpList(recs)) >= min_chain, "the fixture must exercise the chained launch"

    def run_fwd(chain):
        L.check(lib.ssdn_conv_set_chain(int(chain)))
        for name, t in dn.t.items():
            if t.dtype == torch.float16 and "/w" not in name:
                t.fill_(float("nan"))
        if any(op.type == "conv" and op.a.get("urot") is not None for op in plan.fwd):
            dn.t["m/u"].zero_()       # (fused UNROT_FWD relies on the rows the shift leaves empty being zero from the allocation on)
        OpList(recs).run(current_stream())
        torch.cuda.synchronize()
        return {name: t.clone() for name, t in dn.t.items() if t.dtype in (torch.float16, torch.float32) and "/w" not in name}

    separate = run_fwd(False)
    chained = run_fwd(True)
    assert torch.isfinite(separate["m/out32"]).all()
    bad = [name for name in separate if not torch.equal(separate[name].view(torch.int16 if separate[name].dtype == torch.float16 else torch.int32),
                                                        chained[name].view(torch.int16 if chained[name].dtype == torch.float16 else torch.int32))]
    assert not bad, "tensors that differ between the chained and the separate launches: %s" % bad


@pytest.mark.parametrize("cin,cout,bs,B,P,min_chain", [(3, 9, True, 4, 64, 9), (3, 9, True, 8, 32, 3), (3, 3, False, 16, 64, 9), (1, 2, True, 32, 64, 9)])
def test_backward_chain_is_bit_identical(cin, cout, bs, B, P, min_chain, conv_chain_reset):
    """The data gradients of the small layers -- with their fused epilogues (LeakyReLU' mask, skip-gradient add, fused up-sampling
    adjoint) and the max-pool backward ops between them -- execute as ONE launch (k_conv_chain<true>); every gradient tensor of the
    backward pass and the flat parameter gradient must come out bit for bit as from the separate launches."""
    from ssdn.hip import lib as L
    from ssdn.hip.engine import DeviceNet, current_stream
    from ssdn.hip.graph import NetPlan
    lib = L.load()
    dev = torch.device("cuda:0")
    plan = NetPlan("m/", cin, cout, bs, B, P, P, cus=lib.ssdn_device_cus())
    g = torch.Generator(device="cpu").manual_seed(6)
    flat = (torch.randn(plan.nparams, generator=g) * 0.08).to(dev)
    dn = DeviceNet(plan, dev, flat, torch.zeros_like(flat))
    dn.t["m/in32"].copy_(torch.rand(dn.t["m/in32"].shape, generator=g).to(dev))
    dn.pack.run(current_stream())
    dn.fwd.run(current_stream())
    dn.t["m/g32"].copy_((torch.randn(dn.t["m/g32"].shape, generator=g) * 1e-3).to(dev))
    dn.t["m/gmax"][0] = int(np.float32(dn.t["m/g32"].abs().max().item()).view(np.int32))
    assert _longest_chain(lib, dn.bwd) >= min_chain, "the fixture must exercise the chained launch"

    def run_bwd(chain):
        L.check(lib.ssdn_conv_set_chain(int(chain)))
        for name, t in dn.t.items():
            if t.dtype == torch.bfloat16 and "/w" not in name:
                t.fill_(float("nan"))
        dn.grads.fill_(float("nan"))
        dn.bwd.run(current_stream())
        torch.cuda.synchronize()
        out = {name: t.clone() for name, t in dn.t.items() if t.dtype == torch.bfloat16 and "/w" not in name}
        out["grads"] = dn.grads[:plan.nparams].clone()
        return out

    separate = run_bwd(False)
    chained = run_bwd(True)
    assert torch.isfinite(separate["grads"]).all()
    as_int = lambda t: t.view(torch.int16 if t.dtype == torch.bfloat16 else torch.int32)  # noqa: E731
    bad = [name for name in separate if not torch.equal(as_int(separate[name]), as_int(chained[name]))]
    assert not bad, "tensors that differ between the chained and the separate launches: %s" % bad

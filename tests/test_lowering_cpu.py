"""CPU check of the LOWERING (ssdn.hip.graph.NetPlan) with the test-only interpreter (oracle/interp.py):
the planned op list, executed naively in fp32, must reproduce the oracle's forward and autograd gradients."""
import numpy as np
import pytest
import torch

import restate as R
from interp import Interp
from ssdn.hip.graph import NetPlan


def flat_params(plan, p):
    flat = torch.zeros(plan.nparams, dtype=next(iter(p.values())).dtype)
    for l in plan.layers:
        flat[l.w_off:l.w_off + l.M * l.cin * l.ntaps] = p[l.name + ".weight"].reshape(-1)
        flat[l.b_off:l.b_off + l.M] = p[l.name + ".bias"]
    return flat


# dev_cus = 8: the plan of a device with 8 CUs, on which the full-resolution layers of these small cases already have >= one tile per CU --
# the fusions of the BASELINE sizes (decode_block_1.2 storing un-rotated + sign bytes, UPSUM_BWD in the data gradients) are lowered
@pytest.mark.parametrize("cin,cout,bs,B,P,dev_cus", [(3, 9, True, 2, 32, None), (1, 2, True, 1, 32, None), (3, 3, False, 2, 32, None),
                                                     (3, 1, False, 1, 64, None), (3, 9, True, 2, 32, 8), (1, 2, True, 1, 64, 8)])
def test_forward_backward_lowering(cin, cout, bs, B, P, dev_cus):
    """float64 on both sides: in fp32 a handful of activations within 1e-7 of zero flip the LeakyReLU branch between two
    summation orders, which hides real bugs behind a 1e-3 noise floor; in fp64 the lowering must agree to ~1e-12."""
    torch.set_default_dtype(torch.float64)
    try:
        _check_lowering(cin, cout, bs, B, P, dev_cus)
    finally:
        torch.set_default_dtype(torch.float32)


def _check_lowering(cin, cout, bs, B, P, dev_cus=None):
    p = {k: v.double() for k, v in R.make_params(cin, cout, bs, seed=7).items()}
    plan = NetPlan("m/", cin, cout, bs, B, P, P, cus=8, dev_cus=dev_cus)
    if dev_cus and bs:
        assert not any(op.type == "unrot_fwd" for op in plan.fwd) and any(op.a.get("urot") is not None for op in plan.fwd if op.type == "conv")
        assert "m/d1b" not in plan.tensors
    it = Interp(plan, flat_params(plan, p), fp16=False)
    x = R.hash_tensor((B, cin, P, P), 91, 0, 1).double()
    it.t["m/in32"] = x.clone()
    it.run(plan.pack)
    it.run(plan.fwd)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ref = R.net_forward(leaves, x, bs)
    np.testing.assert_allclose(it.t["m/out32"].numpy(), ref.detach().numpy(), rtol=1e-9, atol=1e-11)
    # backward: arbitrary upstream gradient
    g = R.hash_tensor((B, cout, P, P), 92, -1, 1).double() * 1e-3
    (ref * g).sum().backward()
    it.t["m/g32"] = g.clone()
    it.run(plan.bwd)
    for l in plan.layers:
        gw = it.grads[l.w_off:l.w_off + l.M * l.cin * l.ntaps].reshape(l.M, l.cin, l.k, l.k)
        gb = it.grads[l.b_off:l.b_off + l.M]
        rw, rb = leaves[l.name + ".weight"].grad, leaves[l.name + ".bias"].grad
        sc = float(rw.abs().max()) + 1e-12
        np.testing.assert_allclose(gw.numpy(), rw.numpy(), rtol=1e-8, atol=1e-10 * sc, err_msg=l.name)
        np.testing.assert_allclose(gb.numpy(), rb.numpy(), rtol=1e-8, atol=1e-10 * (float(rb.abs().max()) + 1e-12), err_msg=l.name)


def _validate_with_library(op):
    """ask libssdn_hip.so itself (no GPU needed) whether it accepts the planned tiling"""
    import ctypes as C
    from ssdn.hip import lib as L
    a = op.a
    fake = L.View(0x1000, 128, 0)
    if op.type == "conv":
        s = L.ConvArgs()
        s.src0, s.src1 = fake, fake
        s.c0, s.c1, s.up0, s.N, s.H, s.W, s.ntaps = a["c0"], a["c1"], a["up0"], a["N"], a["H"], a["W"], len(a["taps"])
        for i, (dy, dx) in enumerate(a["taps"]):
            s.dy[i], s.dx[i] = dy, dx
        s.M, s.Mpad, s.Ktot, s.kc = a["M"], a["Mpad"], a["Ktot"], a["kc"]
        s.ltw, s.lth, s.ltn, s.bf16 = a["ltw"], a["lth"], a["ltn"], a["bf16"]
        s.dst32 = 0x1000 if a["dst32"] is not None else None
        rc = L.load().ssdn_conv_lds_bytes(C.byref(s))
    else:
        s = L.WgradArgs()
        s.dz, s.src0, s.src1 = fake, fake, fake
        s.c0, s.c1, s.up0, s.N, s.H, s.W, s.ntaps = a["c0"], a["c1"], a["up0"], a["N"], a["H"], a["W"], len(a["taps"])
        for i, (dy, dx) in enumerate(a["taps"]):
            s.dy[i], s.dx[i], s.coff[i] = dy, dx, a["coff"][i]
        s.M, s.Mpad, s.Ktot, s.Kpad, s.nslabs = a["M"], a["Mpad"], a["Ktot"], a["Kpad"], a["nslabs"]
        s.ltw, s.lth, s.ltn, s.csplit = a["ltw"], a["lth"], a["ltn"], a.get("csplit", 0)
        rc = L.load().ssdn_wgrad_lds_bytes(C.byref(s))
    assert 0 <= rc <= 160 * 1024, (op.type, a["layer"], rc, L.load().ssdn_last_error())


def test_plan_tilings_fit_lds():
    """every conv / wgrad tiling of the BASELINE configurations is accepted by the library and fits the 160 KiB LDS of a CU"""
    from ssdn.hip.graph import LDS_LIMIT
    # (the small CU counts make workgroups own several tiles at small sizes: multi-tile prefetch constraints, output split)
    for (cin, cout, bs, B, P, cus) in [(3, 9, True, 32, 64, 256), (3, 9, True, 16, 128, 256), (3, 3, False, 32, 64, 256),
                                       (1, 1, False, 4, 32, 256), (3, 9, True, 2, 768, 256), (3, 9, True, 2, 512, 256),
                                       (3, 9, True, 2, 32, 8), (3, 3, False, 2, 64, 6), (1, 2, True, 1, 32, 3), (3, 9, True, 1, 32, 1),
                                       (3, 9, True, 4, 64, 16), (3, 9, True, 2, 96, 256), (3, 9, True, 8, 64, 64)]:
        plan = NetPlan("m/", cin, cout, bs, B, P, P, cus=cus, train=P <= 128)
        for op in plan.fwd + plan.bwd:
            a = op.a
            if op.type in ("conv", "wgrad"):
                _validate_with_library(op)
            if op.type == "conv":
                padT = max(0, -min(t[0] for t in a["taps"])); padB = max(0, max(t[0] for t in a["taps"]))
                padL = max(0, -min(t[1] for t in a["taps"])); padR = max(0, max(t[1] for t in a["taps"]))
                NP = (1 << a["ltn"]) * ((1 << a["lth"]) + padT + padB) * ((1 << a["ltw"]) + padL + padR)
                lds = NP * (a["kc"] * 2 + 16) + 2 * min(3, a["Mpad"] // 32) * 32 * (a["kc"] * 2 + 16)
                assert lds <= LDS_LIMIT
                assert a["ltw"] + a["lth"] + a["ltn"] <= 8


def test_every_baseline_weight_gradient_op_has_an_instance_in_the_chip_wide_launch():
    """graph.WGRAD_MEGA plans: every SSDN_OP_WGRAD of the BASELINE configurations (and of the sigma-estimation net) must be runnable as
    an entry of k_wgrad_mega (csrc/wgrad_mega.hip instantiates a curated set of kernel variants; an op without one would silently fall
    back to a launch of its own), the planner's cost-model class must be the variant the library picks, and the blocks of a launch
    group must add up to what the plan was made for."""
    import ctypes as C
    from ssdn.hip import graph as G
    from ssdn.hip import lib as L
    from ssdn.hip.engine import DeviceNet
    lib = L.load()
    for (cin, cout, bs, B, P) in [(3, 9, True, 32, 64), (3, 9, True, 16, 128), (3, 1, False, 32, 64), (3, 3, False, 32, 64), (1, 2, True, 32, 64)]:
        plan = NetPlan("m/", cin, cout, bs, B, P, P, cus=256)
        assert getattr(plan, "_mega_ops", None), "the BASELINE sizes use the chip-wide plan"
        flat = torch.zeros(plan.nparams)
        dn = DeviceNet(plan, torch.device("cpu"), flat, torch.zeros_like(flat))
        blocks = {}
        for op in plan.bwd:
            if op.type != "wgrad":
                continue
            rec = dn._mat(op)
            v = (C.c_int32 * 9)()
            inst = lib.ssdn_wgrad_variant(C.byref(rec[1]), v)
            assert inst >= 0 and lib.ssdn_wgrad_mega_ok(C.byref(rec[1])) == 1, (op.a["layer"], op.a["Ktot"], list(v), lib.ssdn_last_error())
            thin, mt, cpw, nl, both, ps, ks = list(v)[:7]
            tile, ntiles, c_tile, fixed = plan._mega_candidates(dict(op.a))
            if not thin and ks == 8 and not both:           # compile-time schedule: the planner must have priced it as one
                base = G.MEGA_COST["static2" if nl == 2 else "static"]
                assert c_tile == pytest.approx(8 * (base[0] + base[1] * mt * cpw)), (op.a["layer"], list(v))
            g = plan.wgrad_group_of(op.a["layer"])
            blocks[g] = blocks.get(g, 0) + op.a["nslabs"] * max(1, op.a["mblocks"]) * max(1, op.a["csplit"])
            assert op.a["mega"] == plan.wgrad_group_info(g)[0]
        for g, n in blocks.items():
            W = plan.wgrad_group_info(g)[0]
            assert n >= W // 2, "a launch group with far fewer blocks than CUs wastes the chip (%d blocks for %d)" % (n, W)


def test_baseline_plan_keeps_sign_bytes_and_route_words_instead_of_activation_reads():
    """BASELINE config 2's training plan: the producers k_cdma / k_conv_thin / k_gdma serve leave LeakyReLU sign bytes and the
    data gradients that only need the sign read them (ssdn_conv_args.sign_out / mask_sign / upsum_mask_sign / unrot_smask); the
    64x64-stage max-pool leaves route words for its stand-alone backward op (ssdn_pool_args.route); an evaluation plan has none."""
    plan = NetPlan("m/", 3, 9, True, 32, 64, 64, cus=256)
    conv = lambda ops, role: {op.a["layer"]: op.a for op in ops if op.type == "conv" and op.a["role"] == role}  # noqa: E731
    f, b = conv(plan.fwd, "fwd"), conv(plan.bwd, "dgrad")
    for layer, t in (("encode_block_1.0", "m/smk_e0"), ("decode_block_2.0", "m/smk_d2a"), ("decode_block_2.2", "m/smk_d2b"),
                     ("decode_block_1.0", "m/smk_d1a"), ("output_block.0", "m/smk_na")):
        assert f[layer]["sign_out"] == t, layer
        assert plan.tensors[t].kind == "u8"
    assert f["decode_block_1.2"]["urot"] is not None and f["decode_block_1.2"]["urot_smask"] == "m/smk_d1b"
    assert b["encode_block_1.2"]["mask_sign"] == "m/smk_e0" and b["decode_block_1.2"]["mask_sign"] == "m/smk_d1a"
    assert b["decode_block_2.2"]["mask_sign"] == "m/smk_d2a" and b["output_block.2"]["mask_sign"] == "m/smk_na"
    assert b["decode_block_1.0"]["upsum_mask_sign"] == "m/smk_d2b" and b["decode_block_2.0"].get("upsum_mask_sign") is None
    # round 6: k_cdma also serves a stage with one tile per TWO CUs where an image is more than one tile (graph.cdma_fills) -- the plain
    # network's 32x32 stage at batch 32 (config 4), not the blind-spot network's 16x16 stage
    from ssdn.hip import graph as G
    assert G.cdma_fills(128, 256, 32, 32) and not G.cdma_fills(128, 256, 16, 16) and G.cdma_fills(512, 256, 16, 16)
    assert b["output_block.0"]["unrot_smask"] == "m/smk_d1b"
    pools = [op.a for op in plan.fwd if op.type == "pool_fwd"]
    assert [bool(a.get("route")) for a in pools] == [True, False]          # 64x64 -> 32x32 stand-alone; 32x32 -> 16x16: its backward is chained
    pb = [op.a for op in plan.bwd if op.type == "pool_bwd" and op.a.get("route")]
    assert len(pb) == 1 and pb[0]["route"] == pools[0]["route"] and pb[0]["H"] == 64
    ev = NetPlan("m/", 3, 9, True, 2, 512, 512, cus=256, train=False)
    assert not any(k.startswith("m/smk_") or k.startswith("m/route_") for k in ev.tensors)
    assert not any(op.a.get("sign_out") or op.a.get("route") for op in ev.fwd)

"""CPU test of the chained-launch planner (csrc/conv_chain.hip::chain_build, reached through ssdn_chain_len -- host code only): which
runs of the materialised forward / backward lists execute as ONE k_conv_chain launch.  (Numerics of the chained launches: -m gpu,
tests/test_hip_ops.py::test_conv_chain_is_bit_identical / ::test_backward_chain_is_bit_identical.)"""
import ctypes as C

import pytest
import torch

from ssdn.hip import lib as L
from ssdn.hip.engine import DeviceNet
from ssdn.hip.graph import NetPlan


def _chains(lib, ol):
    """[(first op, ops)] of the chained runs the executor would form walking the list"""
    out, i = [], 0
    while i < ol.n:
        n = lib.ssdn_chain_len(C.byref(ol.arr, i * C.sizeof(L.OpRec)), ol.n - i)
        assert n >= 0, lib.ssdn_last_error().decode()
        if n > 1:
            out.append((i, n))
            i += n
        else:
            i += 1
    return out


@pytest.mark.parametrize("cin,cout,bs,B,P,fwd_len,bwd_lens", [
    (3, 9, True, 4, 64, 8, [3, 9]),        # BASELINE config 2's per-image shapes (16 images): 16x16 .. 2x2
    (3, 3, False, 16, 64, 8, [3, 9]),      # plain (un-shifted) taps
    (3, 9, True, 2, 128, 0, []),           # 128x128 patches: the small end of the U is 4x4 and the planner's pools are not all fused
])
def test_chained_runs_of_the_materialised_lists(cin, cout, bs, B, P, fwd_len, bwd_lens):
    lib = L.load()
    plan = NetPlan("m/", cin, cout, bs, B, P, P, cus=256, dev_cus=256)
    flat = torch.zeros(plan.nparams)
    dn = DeviceNet(plan, torch.device("cpu"), flat, torch.zeros_like(flat))
    try:
        L.check(lib.ssdn_conv_set_chain(1))
        fwd = _chains(lib, dn.fwd)
        bwd = _chains(lib, dn.bwd)
        if fwd_len:
            assert [n for _, n in fwd] == [fwd_len]
            first = plan.fwd[fwd[0][0]]
            assert first.type == "conv" and first.a["H"] * first.a["W"] == 256 and first.a["M"] == 48      # encode_block_3: the thin 16x16 layer
            # (chip-wide weight-gradient launches wait behind the run of chainable ops: ONE backward chain; with round 3's per-layer
            #  launches the decoder bucket's merged launch stays inside it: two)
            from ssdn.hip import graph as G
            assert [n for _, n in bwd] == ([sum(bwd_lens)] if G.WGRAD_MEGA else bwd_lens)
        else:
            assert all(n <= 8 for _, n in fwd) and all(n <= 12 for _, n in bwd)
        # every chained op is a main-lane conv / max-pool backward on small images, and the runs do not overlap
        for ol, ops, runs in ((dn.fwd, plan.fwd, fwd), (dn.bwd, None, bwd)):
            for i, n in runs:
                for k in range(i, i + n):
                    assert ol.arr[k].lane == ol.arr[i].lane
                    assert ol.arr[k].type in (L.OP["conv"], L.OP["pool_bwd"])
        L.check(lib.ssdn_conv_set_chain(0))
        assert _chains(lib, dn.fwd) == [] and _chains(lib, dn.bwd) == []
    finally:
        lib.ssdn_conv_set_chain(1)

"""The 16x16-pixel 96-channel layers (decode_block_3.*): k_conv's flat path (what a training step runs) against k_cdma forced onto them
(ssdn_conv_set_mode(2): one 256-pixel tile per workgroup, 128 workgroups).  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "selfsupervised-denoising_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch
from ssdn.hip import lib as L
from ssdn.hip.engine import DeviceNet, current_stream
from ssdn.hip.graph import NetPlan
from conv_bench import time_op

lib = L.load()
plan = NetPlan("m/", 3, 9, True, 32, 64, 64, cus=lib.ssdn_device_cus())
dev = torch.device("cuda:0")
flat = torch.randn(plan.nparams, device=dev) * 0.05
dn = DeviceNet(plan, dev, flat, torch.zeros_like(flat))
for name, t in dn.t.items():
    if t.dtype in (torch.float16, torch.bfloat16):
        t.copy_(torch.randn(t.shape, device=dev) * 0.5)
dn.pack.run(current_stream())
for mode in (1, 2, 1, 2):
    lib.ssdn_conv_set_mode(mode)
    for op in plan.fwd + plan.bwd:
        if op.type == "conv" and op.a["layer"] in ("decode_block_3.0", "decode_block_3.2", "encode_block_3.0"):
            a = op.a
            try:
                t = time_op(dn, op)
                print("mode %d %-18s %-5s H=%3d K=%3d M=%3d : %7.1f us" % (mode, a["layer"], a["role"], a["H"], a["Ktot"], a["M"], t))
            except Exception as e:
                print("mode %d %-18s %-5s: %s" % (mode, a["layer"], a["role"], str(e)[:100]))
lib.ssdn_conv_set_mode(1)

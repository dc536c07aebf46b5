"""Measurement aid: kernel times of one inference step at the evaluation sizes (batch 1, 512 x 512 = padded BSD300, 768 x 768 = padded Kodak;
Denoiser.run_pipeline in eval mode).  usage: python tools/eval_trace.py [size ...]   (run under rocprofv3 --kernel-trace --stats for per-kernel times)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "selfsupervised-denoising_amd"), ROOT]
import torch
import bench as B
from ssdn.denoiser import Denoiser
from ssdn.datasets import NoisyDataset

d = Denoiser(B.make_cfg(), device="cuda:0")
d.eval()
for P in [int(x) for x in sys.argv[1:]] or [512, 768]:
    g = torch.Generator().manual_seed(P)
    x = torch.rand((1, 3, P, P), generator=g).cuda()
    meta = {NoisyDataset.Metadata.INPUT_NOISE_VALUES: torch.full((1, 1, 1, 1), 25 / 255.0, device="cuda:0")}
    with torch.no_grad():
        for _ in range(3):
            d.run_pipeline([x, None, meta])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            d.run_pipeline([x, None, meta])
        torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 10
    flops = 10.237e9 * (P * P) / 4096.0        # forward GFLOP of one 64 x 64 patch, scaled by area
    print("eval %d x %d, batch 1: %.3f ms per image = %.1f images/s, %.0f TFLOP/s algorithmic" % (P, P, ms, 1e3 / ms, flops / ms / 1e9))

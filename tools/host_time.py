"""tuning aid: host time of the enqueue calls of one training step vs its GPU time (is the step host-bound?)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "selfsupervised-denoising_amd"), os.path.join(ROOT, "oracle")]
import torch
import bench
from ssdn.denoiser import Denoiser
from ssdn.datasets import NoisyDataset
d = Denoiser(bench.make_cfg(), device="cuda:0"); d.train()
noisy, clean = bench.synth_batch(32, 64, 1, "cuda:0")
meta = {NoisyDataset.Metadata.INPUT_NOISE_VALUES: torch.full((32, 1, 1, 1), 25 / 255.0, device="cuda:0"), NoisyDataset.Metadata.CLEAN: clean}
data = [noisy, clean, meta]
for _ in range(10): d.train_step(data, 1e-4)
torch.cuda.synchronize()
eng = d._last_train_engine
def timed(fn, n=30):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    ts.sort(); return ts[len(ts) // 2]
h, g = timed(lambda: eng.forward()); print("forward  : host enqueue %.0f us, until done %.0f us" % (h * 1e6, g * 1e6))
h, g = timed(lambda: eng.backward()); print("backward : host enqueue %.0f us, until done %.0f us" % (h * 1e6, g * 1e6))
h, g = timed(lambda: eng.adam(1e-4, 5)); print("adam+pack: host enqueue %.0f us, until done %.0f us" % (h * 1e6, g * 1e6))
h, g = timed(lambda: d.train_step(data, 1e-4)); print("train_step: host %.0f us, until done %.0f us" % (h * 1e6, g * 1e6))
os.environ["X"] = "1"

"""Upper bound of what LeakyReLU sign bit-masks could save: the same training step planned WITHOUT the mask operand of the data-gradient
epilogues (wrong gradients -- timing only).  Measurement aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "selfsupervised-denoising_amd"), ROOT]
import torch
import bench as B
from ssdn.hip import graph as G
from ssdn.denoiser import Denoiser
from ssdn.datasets import DevicePatchStream, NoisyDataset
from ssdn.params import NoiseAlgorithm

dev = torch.device("cuda", 0)
nd = NoisyDataset(None, "gauss25", NoiseAlgorithm.SELFSUPERVISED_DENOISING, pad_uniform=False, pad_multiple=32, square=True, training_mode=True)
g = torch.Generator().manual_seed(1)
u8 = [torch.randint(0, 256, (32, 3, 64, 64), generator=g, dtype=torch.uint8).pin_memory() for _ in range(4)]
idx = torch.arange(32)
orig = G.NetPlan._conv
runs = {}
for v in ("masks", "no masks (timing only)"):
    if v != "masks":
        def nomask(self, lst, layer, role, *a, **kw):
            if role == "dgrad":
                kw["mask"] = None
            return orig(self, lst, layer, role, *a, **kw)
        G.NetPlan._conv = nomask
    torch.manual_seed(0)
    d = Denoiser(B.make_cfg(), device=str(dev))
    d.train()
    stream = DevicePatchStream(None, nd, dev, seed=1).attach(d)
    state = {"pending": stream.upload(u8[0])}

    def step(i, d=d, stream=stream, state=state):
        cur, state["pending"] = state["pending"], stream.upload(u8[(i + 1) % 4])
        d.train_step(stream.prepare(cur, idx), 3e-4, None)
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    runs[v] = step
G.NetPlan._conv = orig
res = {v: [] for v in runs}
for rnd in range(3):
    for v, step in runs.items():
        for i in range(10):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(150):
            step(i)
        torch.cuda.synchronize()
        res[v].append(1e3 * (time.perf_counter() - t0) / 150)
for v in runs:
    print("%s: ms/step %s  median %.4f" % (v, [round(x, 4) for x in res[v]], sorted(res[v])[1]))

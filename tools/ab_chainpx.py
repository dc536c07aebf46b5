"""Same-process A/B: chained launches with / without the 16x16 (256-pixel) thin layers (ssdn_conv_set_chain(1|2)); measurement aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "selfsupervised-denoising_amd"), ROOT]
import torch
import bench as B
from ssdn.hip import lib as L
from ssdn.denoiser import Denoiser
from ssdn.datasets import DevicePatchStream, NoisyDataset
from ssdn.params import NoiseAlgorithm

dev = torch.device("cuda", 0)
nd = NoisyDataset(None, "gauss25", NoiseAlgorithm.SELFSUPERVISED_DENOISING, pad_uniform=False, pad_multiple=32, square=True, training_mode=True)
g = torch.Generator().manual_seed(1)
u8 = [torch.randint(0, 256, (32, 3, 64, 64), generator=g, dtype=torch.uint8).pin_memory() for _ in range(4)]
idx = torch.arange(32)
torch.manual_seed(0)
d = Denoiser(B.make_cfg(), device=str(dev))
d.train()
stream = DevicePatchStream(None, nd, dev, seed=1).attach(d)
state = {"pending": stream.upload(u8[0])}


def step(i):
    cur, state["pending"] = state["pending"], stream.upload(u8[(i + 1) % 4])
    d.train_step(stream.prepare(cur, idx), 3e-4, None)


res = {1: [], 2: []}
for rnd in range(3):
    for mode in (1, 2):
        L.load().ssdn_conv_set_chain(mode)
        for i in range(15):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(150):
            step(i)
        torch.cuda.synchronize()
        res[mode].append(1e3 * (time.perf_counter() - t0) / 150)
L.load().ssdn_conv_set_chain(1)
for mode, name in ((1, "chains take the 16x16 thin layers"), (2, "chains up to 8x8 only")):
    print("%s: ms/step %s  median %.4f" % (name, [round(x, 4) for x in res[mode]], sorted(res[mode])[1]))

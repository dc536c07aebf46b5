"""Same-process A/B of graph.SIGN_BYTES_CONV (LeakyReLU sign bytes vs the saved activations as masks of the k_cdma data gradients); measurement aid."""
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
runs = {}
for v in (True, False):
    G.SIGN_BYTES_CONV = v
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
    runs[v] = (step, d)
res = {v: [] for v in runs}
for rnd in range(3):
    for v, (step, d) in runs.items():
        for i in range(10):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(150):
            step(i)
        torch.cuda.synchronize()
        res[v].append(1e3 * (time.perf_counter() - t0) / 150)
for v in runs:
    print("conv sign bytes %s: ms/step %s  median %.4f" % (v, [round(x, 4) for x in res[v]], sorted(res[v])[1]))
w = [runs[v][1].flat.clone() for v in runs]
print("weights identical across variants:", all(torch.equal(w[0], x) for x in w[1:]))

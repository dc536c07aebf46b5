"""Same-process A/B: two-lane backward pass vs every weight-gradient op on the main lane (serial), with the persistent
weight-gradient grids planned for 3/4 or all of the CUs (measurement aid, round 4)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "selfsupervised-denoising_amd"), ROOT]
import torch
import bench as B
from ssdn.hip import engine as E, graph as G
from ssdn.denoiser import Denoiser
from ssdn.datasets import DevicePatchStream, NoisyDataset
from ssdn.params import NoiseAlgorithm

dev = torch.device("cuda", 0)
VARIANTS = {"two_lanes_192": ({"wgrad": (1,), "wreduce": (1,)}, (3, 4)),
            "serial_192": ({"wgrad": (0,), "wreduce": (0,)}, (3, 4)),
            "serial_256": ({"wgrad": (0,), "wreduce": (0,)}, (1, 1)),
            "two_lanes_256": ({"wgrad": (1,), "wreduce": (1,)}, (1, 1))}
nd = NoisyDataset(None, "gauss25", NoiseAlgorithm.SELFSUPERVISED_DENOISING, pad_uniform=False, pad_multiple=32, square=True, training_mode=True)
g = torch.Generator().manual_seed(1)
u8 = [torch.randint(0, 256, (32, 3, 64, 64), generator=g, dtype=torch.uint8).pin_memory() for _ in range(4)]
idx = torch.arange(32)
runs = {}
for v, (lane, frac) in VARIANTS.items():
    torch.manual_seed(0)
    E.OpList.LANE = lane
    G.WGRAD_CU_FRAC = frac
    E.MAIN_LANE_WGRADS = () if lane["wgrad"] == (0,) else ("encode_block_1.0", "encode_block_1.2")
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
N = 150
res = {v: [] for v in VARIANTS}
for rnd in range(3):
    for v in VARIANTS:
        step = runs[v][0]
        for i in range(10):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            step(i)
        torch.cuda.synchronize()
        res[v].append(1e3 * (time.perf_counter() - t0) / N)
for v in VARIANTS:
    print("%s: ms/step %s  median %.4f" % (v, [round(x, 4) for x in res[v]], sorted(res[v])[len(res[v]) // 2]))

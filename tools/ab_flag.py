"""Same-process A/B of planner constants of ssdn.hip.graph (e.g. POOL_ROUTE, SIGN_BYTES_CONV, SPLIT_GROUP0): the BASELINE config-2 training
step under each variant, interleaved rounds; also reports whether the weights after the same steps are bit-identical.  Measurement aid.
usage: python tools/ab_flag.py FLAG [FLAG2 ...]                       (boolean constants, switched together: True vs False)
       python tools/ab_flag.py "NAME=expr" ["NAME2=expr"] --vs "NAME=expr" ... [--vs ...]   (any constants; one variant per --vs group)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "selfsupervised-denoising_amd"), ROOT]
import torch
import bench as B
from ssdn.hip import graph as G
from ssdn.denoiser import Denoiser
from ssdn.datasets import DevicePatchStream, NoisyDataset
from ssdn.params import NoiseAlgorithm

args = sys.argv[1:]
if any("=" in x for x in args):
    groups, cur = [], []
    for x in args:
        if x == "--vs":
            groups.append(cur); cur = []
        else:
            cur.append(x)
    groups.append(cur)
    variants = {" ".join(gr): [(x.split("=", 1)[0], eval(x.split("=", 1)[1])) for x in gr] for gr in groups}
else:
    assert args and all(isinstance(getattr(G, f), bool) for f in args), "name boolean constants of ssdn.hip.graph"
    variants = {"+".join(args) + " = %s" % v: [(f, v) for f in args] for v in (True, False)}
dev = torch.device("cuda", 0)
nd = NoisyDataset(None, "gauss25", NoiseAlgorithm.SELFSUPERVISED_DENOISING, pad_uniform=False, pad_multiple=32, square=True, training_mode=True)
g = torch.Generator().manual_seed(1)
u8 = [torch.randint(0, 256, (32, 3, 64, 64), generator=g, dtype=torch.uint8).pin_memory() for _ in range(4)]
idx = torch.arange(32)
runs = {}
for v, sets in variants.items():
    for f, val in sets:
        assert hasattr(G, f), f
        setattr(G, f, val)
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
    print("%s: ms/step %s  median %.4f" % (v, [round(x, 4) for x in res[v]], sorted(res[v])[1]))
w = [runs[v][1].flat.clone() for v in runs]
print("weights identical across variants:", all(torch.equal(w[0], x) for x in w[1:]))

"""Same-process A/B of the weight-gradient plans: round 3's per-layer launches on the side lane vs the chip-wide launch
(graph.WGRAD_MEGA = "all" / "buckets") on the main lane or the side lane (measurement aid, round 4)."""
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
VARIANTS = {"per_layer_lanes": (None, 1), "mega_all_lane0": ("all", 0), "mega_split_half": ("split", 0, (1, 2)), "mega_split_3_8": ("split", 0, (3, 8)),
            "mega_split_5_8": ("split", 0, (5, 8)), "split_half_dec2": ("split", 0, (1, 2), ("output_block", "decode_block_2")),
            "split_5_8_dec2": ("split", 0, (5, 8), ("output_block", "decode_block_2")), "split_half_dec22": ("split", 0, (1, 2), ("output_block", "decode_block_2.2")),
            "split_dec22_thin": ("split", 0, (1, 2), ("output_block", "decode_block_2.2", "decode_block_1.0/skip")),
            "split_dec22_thin_d20s": ("split", 0, (1, 2), ("output_block", "decode_block_2.2", "decode_block_1.0/skip", "decode_block_2.0/skip")), "mega_buckets_lane0": ("buckets", 0), "mega_buckets_lane1": ("buckets", 1),
            "mega_all_lane1": ("all", 1)}
if len(sys.argv) > 1:
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in sys.argv[1:]}
nd = NoisyDataset(None, "gauss25", NoiseAlgorithm.SELFSUPERVISED_DENOISING, pad_uniform=False, pad_multiple=32, square=True, training_mode=True)
g = torch.Generator().manual_seed(1)
u8 = [torch.randint(0, 256, (32, 3, 64, 64), generator=g, dtype=torch.uint8).pin_memory() for _ in range(4)]
idx = torch.arange(32)
runs = {}
for v, spec in VARIANTS.items():
    mode, lane = spec[0], spec[1]
    torch.manual_seed(0)
    G.WGRAD_MEGA = mode
    E.MEGA_LANE = lane
    G.SPLIT_HEAD_CUS = spec[2] if len(spec) > 2 else (1, 2)
    G.SPLIT_GROUP0 = spec[3] if len(spec) > 3 else ("output_block",)
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
    eng = d._last_train_engine
    print(v, "planned makespan (cycles):", getattr(eng.main.plan, "mega_makespan", None), flush=True)
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

"""One GPU's shard of every BASELINE configuration through `Denoiser.train_step` (device patch stream, metrics on): ms per step and
patches/s -- a sanity sweep that no configuration falls off the fast paths (measurement aid; the headline number is bench.py's).
usage: python tools/cfg_bench.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "selfsupervised-denoising_amd"), ROOT]
import torch
import ssdn
from ssdn.denoiser import Denoiser
from ssdn.datasets import DevicePatchStream, NoisyDataset
from ssdn.params import ConfigValue, NoiseAlgorithm, NoiseValue

CONFIGS = [  # tag, algorithm, style, noise value, channels, per-GPU batch, patch
    ("config 2: ssdn gauss25 sigma_known, 64x64, batch 32", "ssdn", "gauss25", "known", 3, 32, 64),
    ("config 3: ssdn gauss25 sigma_var (+ sigma-estimation net), 64x64, batch 256 / 8 GPUs", "ssdn", "gauss25", "var", 3, 32, 64),
    ("config 4: n2v gauss25, 64x64, batch 256 / 8 GPUs", "n2v", "gauss25", "known", 3, 32, 64),
    ("config 5: ssdn poisson30 sigma_const, 128x128, batch 128 / 8 GPUs", "ssdn", "poisson30", "const", 3, 16, 128),
    ("config 1 shape on the device: n2c gauss25 mono, 32x32, batch 4", "n2c", "gauss25", "known", 1, 4, 32),
]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
if os.environ.get("WGRAD_MEGA"):          # A/B aid: plan variant ("all", "split", "buckets", "none")
    from ssdn.hip import graph as _G
    _G.WGRAD_MEGA = None if os.environ["WGRAD_MEGA"] == "none" else os.environ["WGRAD_MEGA"]
    print("WGRAD_MEGA =", _G.WGRAD_MEGA)
if os.environ.get("SIGMA_CONCURRENT") is not None:          # A/B aid: the sigma-estimation network's lists on a second stream (1) or behind the main net's (0)
    from ssdn.hip.engine import DenoiserEngine as _DE
    _DE.SIGMA_CONCURRENT = int(os.environ["SIGMA_CONCURRENT"])
    print("SIGMA_CONCURRENT =", _DE.SIGMA_CONCURRENT)
only = os.environ.get("CFG_ONLY")                            # e.g. CFG_ONLY="config 3"
dev = torch.device("cuda", 0)
for tag, alg, style, mode, ch, B, P in CONFIGS:
    if only and not tag.startswith(only):
        continue
    cfg = ssdn.cfg.base()
    cfg[ConfigValue.ALGORITHM] = NoiseAlgorithm(alg)
    cfg[ConfigValue.NOISE_STYLE] = style
    cfg[ConfigValue.NOISE_VALUE] = NoiseValue(mode)
    cfg[ConfigValue.IMAGE_CHANNELS] = ch
    ssdn.cfg.infer(cfg, model_only=True)
    torch.manual_seed(0)
    d = Denoiser(cfg, device=str(dev))
    d.train()
    nd = NoisyDataset(None, style, NoiseAlgorithm(alg), pad_uniform=False, pad_multiple=32, square=cfg[ConfigValue.BLINDSPOT], training_mode=True)
    stream = DevicePatchStream(None, nd, dev, seed=1).attach(d)
    g = torch.Generator().manual_seed(1)
    u8 = [torch.randint(0, 256, (B, ch, P, P), generator=g, dtype=torch.uint8).pin_memory() for _ in range(4)]
    idx = torch.arange(B)
    pending = stream.upload(u8[0])

    def step(i):
        global pending
        cur, pending = pending, stream.upload(u8[(i + 1) % 4])
        d.train_step(stream.prepare(cur, idx), 3e-4, None, metrics=True)
    for i in range(15):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    m = d.read_metrics("train")
    print("%-86s %8.3f ms/step %9.0f patches/s   loss %.4f" % (tag, 1e3 * dt, B / dt, m["loss"][0] / max(1, m["loss"][1])), flush=True)

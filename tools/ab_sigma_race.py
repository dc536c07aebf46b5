"""Config 3 (sigma-estimation network next to the main network): is the concurrent schedule (DenoiserEngine.SIGMA_CONCURRENT) bit-identical to
the sequential one, and to itself run after run?  (bring-up aid, GPU only)   usage: python tools/ab_sigma_race.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "selfsupervised-denoising_amd"), ROOT]
import torch
import ssdn
from ssdn.denoiser import Denoiser
from ssdn.datasets import NoisyDataset
from ssdn.hip.engine import DenoiserEngine
from ssdn.params import ConfigValue, NoiseAlgorithm, NoiseValue

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 3
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
B, P = 32, 64
g = torch.Generator().manual_seed(3)
clean = torch.rand(B, 3, P, P, generator=g)
noisy = (clean + torch.randn(B, 3, P, P, generator=g) * 0.1).clamp(0, 1)
MD = NoisyDataset.Metadata


def run(conc):
    DenoiserEngine.SIGMA_CONCURRENT = conc
    cfg = ssdn.cfg.base()
    cfg[ConfigValue.ALGORITHM] = NoiseAlgorithm("ssdn")
    cfg[ConfigValue.NOISE_STYLE] = "gauss25"
    cfg[ConfigValue.NOISE_VALUE] = NoiseValue("var")
    cfg[ConfigValue.IMAGE_CHANNELS] = 3
    ssdn.cfg.infer(cfg, model_only=True)
    torch.manual_seed(0)
    d = Denoiser(cfg, device=str(dev))
    d.train()
    grads = []
    for i in range(steps):
        meta = {MD.INPUT_NOISE_VALUES: torch.full((B, 1, 1, 1), 0.1), MD.CLEAN: clean.to(dev)}
        d.train_step([noisy.to(dev), None, meta], 3e-4)
        torch.cuda.synchronize()
        grads.append(d.flat_grad.clone())
    return d.flat.clone(), grads, d


seq, gseq, d0 = run(0)
print("mode", mode)
for rep in range(reps):
    con, gcon, d1 = run(mode)
    n_main, n_sig = d1._n_main, d1._n_sig
    for i in range(steps):
        a, b = gseq[i], gcon[i]
        ne = (a != b)
        if ne.any():
            idx = ne.nonzero().view(-1)
            print("rep %d step %d: gradients differ in %d elements (main net: %d, sigma net: %d), first at %d, max abs diff %.3e" % (
                rep, i, len(idx), int((idx < n_main).sum()), int(((idx >= n_main) & (idx < n_main + n_sig)).sum()), int(idx[0]), float((a - b).abs().max())))
            bad = {}
            for nn, base in ((d1.get_model(Denoiser.MODEL, False), 0), (d1.get_model(Denoiser.SIGMA_ESTIMATOR, False), n_main)):
                for l in nn.layers:
                    lo, hi = base + l.w_off, base + l.w_off + l.M * l.cin * l.k * l.k + l.M
                    c = int(ne[lo:hi].sum())
                    if c:
                        bad[("sigma/" if base else "") + l.name] = c
            print("    layers:", bad)
            break
    else:
        print("rep %d: %d steps bit-identical to the sequential schedule; parameters equal: %s" % (rep, steps, bool(torch.equal(seq, con))))

"""Write the plan blob of one configuration and input shape (DenoiserEngine.export_plan) -- what ssdn_plan_load() of libssdn_hip.so takes
(include/ssdn_hip.h, "step-level entry points").  Needs an MI355X (the plan is made for the device's CU count).
usage: python tools/export_plan.py out.bin [--algorithm ssdn] [--style gauss25] [--mode known] [--batch 32] [--patch 64] [--eval]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "selfsupervised-denoising_amd")]
import torch  # noqa: E402
import ssdn  # noqa: E402
from ssdn.denoiser import Denoiser  # noqa: E402
from ssdn.params import ConfigValue, NoiseAlgorithm, NoiseValue  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--algorithm", default="ssdn")
    ap.add_argument("--style", default="gauss25")
    ap.add_argument("--mode", default="known")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--patch", type=int, default=64)
    ap.add_argument("--eval", action="store_true", help="an inference plan (no backward / optimiser phase)")
    a = ap.parse_args()
    cfg = ssdn.cfg.base()
    cfg[ConfigValue.ALGORITHM] = NoiseAlgorithm(a.algorithm)
    cfg[ConfigValue.NOISE_STYLE] = a.style
    cfg[ConfigValue.NOISE_VALUE] = NoiseValue(a.mode)
    cfg[ConfigValue.IMAGE_CHANNELS] = 3
    ssdn.cfg.infer(cfg, model_only=True)
    d = Denoiser(cfg, device="cuda:0")
    d.train(not a.eval)
    with torch.set_grad_enabled(not a.eval):
        eng = d._engine(a.batch, a.patch, a.patch, not a.eval)
    blob = eng.export_plan(dict(config=d.config_name(), algorithm=a.algorithm))
    with open(a.out, "wb") as f:
        f.write(blob)
    print("wrote %s: %d bytes, configuration %s, batch %d at %dx%d, %s" % (a.out, len(blob), d.config_name(), a.batch, a.patch, a.patch,
                                                                           "inference" if a.eval else "training"))


if __name__ == "__main__":
    main()

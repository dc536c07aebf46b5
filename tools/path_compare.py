"""Forward of one BASELINE config-5 shard through the drop-in Denoiser; saves the raw network output and the loss so that two
kernel-selection settings (environment toggles, separate processes) can be compared element-wise (tuning / debugging aid).
usage: python tools/path_compare.py out.pt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "selfsupervised-denoising_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import restate as R
from test_hip_denoiser import make_denoiser, _flat_of
from test_hip_fullsize import _inputs
from ssdn.denoiser import Denoiser
from ssdn.datasets import NoisyDataset
from ssdn.params import PipelineOutput

B, P = 16, 128
d = make_denoiser("ssdn", "poisson30", "const", 3)
d.train()
p0 = R.make_params(3, 9, True, seed=5)
tr = R.CpuTrainer("ssdn", 3, "poisson30", "const", params={k: v.clone() for k, v in p0.items()})
nets = [(d.get_model(Denoiser.MODEL, False), 0, tr.p)]
d.flat.copy_(_flat_of(d, nets, tr))
d.mark_dirty()
clean, noisy, npar = _inputs(B, 3, P, "poisson30", 301)
MD = NoisyDataset.Metadata
out = d.run_pipeline([noisy, clean, {MD.INPUT_NOISE_VALUES: npar, MD.CLEAN: clean}])
d.backward()
torch.cuda.synchronize()
eng = d._last_train_engine
res = {"loss": out[PipelineOutput.LOSS].detach().cpu(), "grad": d.flat_grad.cpu().clone()}
for name in ("m/out32", "m/e3", "m/p3", "m/e4", "m/e5", "m/e6", "m/d5b", "m/d4b", "m/d3b", "m/d2b", "m/d1b", "m/na", "m/nb"):
    if name in eng.main.t:
        res[name] = eng.main.t[name].float().cpu()
torch.save(res, sys.argv[1])
print("saved", sys.argv[1], [k for k in res])

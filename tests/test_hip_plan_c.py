"""GPU: the step-level C entry points (include/ssdn_hip.h: ssdn_plan_load / ssdn_plan_bind / ssdn_train_step; csrc/plan.hip).
BASELINE config 2 (ssdn gauss25 sigma_known, batch 32, 64x64 RGB) is planned once by the Python package and exported as a blob
(DenoiserEngine.export_plan); a SEPARATE process that imports nothing of this repository (tests/plan_c_driver.py: ctypes + a torch
tensor as the device arena) loads the blob into libssdn_hip.so and runs two training steps.  Loss and updated parameters must be
BIT-identical to the same two steps through `Denoiser.train_step`."""
import os
import subprocess
import sys

import pytest
import torch

from test_hip_denoiser import make_denoiser

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_step_through_the_c_abi_alone_is_bit_identical(tmp_path):
    import fullsize as F
    from ssdn.datasets import NoisyDataset
    from ssdn.hip import lib as L
    from ssdn.params import PipelineOutput
    alg, style, mode, B, P = F.CASES["cfg2"]
    clean, noisy, npar = F.inputs("cfg2")
    MD = NoisyDataset.Metadata
    torch.manual_seed(21)
    d = make_denoiser(alg, style, mode, 3)
    d.train()
    params0 = d.flat.detach().cpu().clone()
    lr, steps = 3e-4, 2
    losses = []
    for _ in range(steps):
        out = d.train_step([noisy, clean, {MD.INPUT_NOISE_VALUES: npar, MD.CLEAN: clean}], lr)
        torch.cuda.synchronize()
        losses.append(out[PipelineOutput.LOSS].detach().cpu().reshape(-1).clone())
    eng = d._last_train_engine
    blob = eng.export_plan(dict(config="BASELINE config 2: ssdn gauss25 sigma_known"))
    (tmp_path / "plan.bin").write_bytes(blob)
    torch.save(dict(params=params0, noisy=noisy, noise_param=npar.reshape(-1), lr=lr, steps=steps), tmp_path / "in.pt")
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "plan_c_driver.py"), L.LIB_PATH, str(tmp_path / "plan.bin"),
                        str(tmp_path / "in.pt"), str(tmp_path / "out.pt")], capture_output=True, text=True, timeout=600, cwd=str(tmp_path), env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    got = torch.load(tmp_path / "out.pt")
    assert got["meta"]["B"] == B and got["meta"]["pipeline"] == "ssdn" and len(got["meta"]["layers"]) == 20
    for a, b in zip(losses, got["loss"]):
        assert torch.equal(a, b.reshape(-1)), (a[:4], b.reshape(-1)[:4])
    n = d._n_main
    assert torch.equal(d.flat.detach().cpu()[:n], got["params"][:n]), "parameters after two steps through the C ABI differ"
    assert torch.equal(d._last_train_engine.pme.cpu().reshape(-1), got["pme"])
    assert len(blob) < 2 << 20

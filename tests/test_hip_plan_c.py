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


@pytest.mark.parametrize("tag", ["cfg2", "cfg3", "cfg4"])
def test_train_step_through_the_c_abi_alone_is_bit_identical(tmp_path, tag):
    """cfg2: ssdn, sigma known (the bench workload).  cfg3: + the sigma-estimation network -- the blob concatenates the two networks' lists of
    a phase into ONE ssdn_run_ops list (the Python path runs them as separate calls on two streams): same results, bit for bit.  cfg4:
    Noise2Void -- the plain network, masked MSE at the exported coordinates against the reference image."""
    import fullsize as F
    from ssdn.datasets import NoisyDataset
    from ssdn.hip import lib as L
    from ssdn.params import PipelineOutput
    alg, style, mode, B, P = F.CASES[tag]
    clean, noisy, npar = F.inputs(tag)
    MD = NoisyDataset.Metadata
    meta = {MD.INPUT_NOISE_VALUES: npar, MD.CLEAN: clean}
    ref = clean
    if alg == "n2v":
        ref, coords = F.n2v_extras(tag)
        meta[MD.MASK_COORDS] = coords
    torch.manual_seed(21)
    d = make_denoiser(alg, style, mode, 3)
    d.train()
    if F.sigma_params(tag) is not None:           # (the reference zero-initialises the estimator's last layer: give the sigma path signal)
        import restate as R
        from ssdn.denoiser import Denoiser
        d.get_model(Denoiser.SIGMA_ESTIMATOR, False).load_state_dict(R.reference_state_dict(F.sigma_params(tag)))
    params0 = d.flat.detach().cpu().clone()
    lr, steps = 3e-4, 2
    losses = []
    for _ in range(steps):
        out = d.train_step([noisy, ref, meta], lr)
        torch.cuda.synchronize()
        losses.append(out[PipelineOutput.LOSS].detach().cpu().reshape(-1).clone())
    eng = d._last_train_engine
    blob = eng.export_plan(dict(config="BASELINE shard " + tag))
    (tmp_path / "plan.bin").write_bytes(blob)
    torch.save(dict(params=params0, noisy=noisy, noise_param=npar.reshape(-1) if (alg == "ssdn" and mode == "known") else None,
                    ref=eng.ref.detach().cpu() if alg != "ssdn" else None, coords=eng.coords.detach().cpu() if alg == "n2v" else None,
                    lr=lr, steps=steps), tmp_path / "in.pt")
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "plan_c_driver.py"), L.LIB_PATH, str(tmp_path / "plan.bin"),
                        str(tmp_path / "in.pt"), str(tmp_path / "out.pt")], capture_output=True, text=True, timeout=600, cwd=str(tmp_path), env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    got = torch.load(tmp_path / "out.pt")
    assert got["meta"]["B"] == B and got["meta"]["pipeline"] == {"ssdn": "ssdn", "n2v": "mask_mse"}[alg]
    assert len(got["meta"]["layers"]) == {"cfg2": 20, "cfg3": 40, "cfg4": 20}[tag]
    for a, b in zip(losses, got["loss"]):
        assert torch.equal(a, b.reshape(-1)), (a[:4], b.reshape(-1)[:4])
    n = d.flat.numel()
    assert torch.equal(d.flat.detach().cpu(), got["params"][:n]), "parameters after two steps through the C ABI differ"
    if alg == "ssdn":
        assert torch.equal(d._last_train_engine.pme.cpu().reshape(-1), got["pme"])
    assert len(blob) < 4 << 20

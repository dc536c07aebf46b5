"""GPU: the drop-in `Denoiser` façade end to end (run_pipeline -> backward -> fused Adam) against
(a) the golden training trajectories generated from the reference (tests/golden/g_train_*.npz) and
(b) the oracle trainer run side by side on the host (full tensors).
fp16 activations vs the fp32 reference: losses within 1e-2 relative; parameter UPDATES (what Adam did in 3 steps) must point
the same way: cosine >= 0.9 per network (Adam's first steps are ~lr*sign(g), so an element whose tiny gradient flips sign
under fp16 noise moves the opposite way -- the cosine, not an element-wise bound, is the meaningful statement)."""
import os

import numpy as np
import pytest
import torch

import restate as R
from test_oracle_golden import TRAIN_CASES, train_inputs

pytestmark = pytest.mark.gpu


def make_denoiser(alg, style, mode, ch):
    import ssdn
    from ssdn.denoiser import Denoiser
    from ssdn.params import ConfigValue, NoiseAlgorithm, NoiseValue
    cfg = ssdn.cfg.base()
    cfg[ConfigValue.ALGORITHM] = NoiseAlgorithm(alg)
    cfg[ConfigValue.NOISE_STYLE] = style
    cfg[ConfigValue.NOISE_VALUE] = NoiseValue(mode)
    cfg[ConfigValue.IMAGE_CHANNELS] = ch
    ssdn.cfg.infer(cfg, model_only=True)
    return Denoiser(cfg, device="cuda:0")


@pytest.mark.parametrize("tag,alg,style,mode,ch", TRAIN_CASES)
def test_training_trajectory(golden_dir, tag, alg, style, mode, ch):
    from ssdn.denoiser import Denoiser
    from ssdn.datasets import NoisyDataset
    from ssdn.params import PipelineOutput
    g = np.load(os.path.join(golden_dir, "g_train_%s.npz" % tag))
    bs = alg == "ssdn"
    cout = ch + ch * (ch + 1) // 2 if alg == "ssdn" else ch
    p0 = R.make_params(ch, cout, bs, seed=5)
    sp0 = R.make_params(ch, 1, False, seed=6) if (mode == "var" and alg == "ssdn") else None
    d = make_denoiser(alg, style, mode, ch)
    d.get_model(Denoiser.MODEL, False).load_state_dict(R.reference_state_dict(p0))
    if sp0 is not None:
        d.get_model(Denoiser.SIGMA_ESTIMATOR, False).load_state_dict(R.reference_state_dict(sp0))
    d.mark_dirty()
    d.train()
    tr = R.CpuTrainer(alg, ch, style, mode, params={k: v.clone() for k, v in p0.items()},
                      sigma_params={k: v.clone() for k, v in sp0.items()} if sp0 is not None else None)
    clean, noisy, ref, coords, npar = train_inputs(alg, style, ch)
    MD = NoisyDataset.Metadata
    meta = {MD.INPUT_NOISE_VALUES: npar, MD.CLEAN: clean}
    if alg == "n2v":
        meta[MD.MASK_COORDS] = coords
    flat0 = d.flat.clone()
    for it in range(3):
        lr = R.trainer_lr((it + 1) * 40, 1000)
        out = d.train_step([noisy, ref, meta], lr)
        r = tr.step(lr, noisy, ref, npar, coords)
        loss = out[PipelineOutput.LOSS].detach().cpu().numpy()
        np.testing.assert_allclose(loss, g["loss_it%d" % it], rtol=1e-2, atol=2e-3, err_msg="loss at iteration %d vs reference" % it)
        np.testing.assert_allclose(loss, r["loss"].detach().numpy(), rtol=1e-2, atol=2e-3)
        if it == 0:
            o = out[PipelineOutput.IMG_DENOISED].detach().cpu()
            # posterior mean vs the reference's: 1e-2 (with an ESTIMATED, still tiny sigma the PME weights amplify fp16 error)
            assert float((o - torch.from_numpy(g["out0"])).norm() / torch.from_numpy(g["out0"]).norm()) <= 1e-2
    torch.cuda.synchronize()
    # parameter updates: device vs oracle
    upd = (d.flat - flat0).cpu()
    nets = [(d.get_model(Denoiser.MODEL, False), 0, p0, tr.p)]
    if sp0 is not None:
        nets.append((d.get_model(Denoiser.SIGMA_ESTIMATOR, False), d._n_main, sp0, tr.ps))
    for net, base, start, cur in nets:
        du, ru = [], []
        for l in net.layers:
            for suffix, off, n in ((".weight", l.w_off, l.M * l.cin * l.k * l.k), (".bias", l.b_off, l.M)):
                du.append(upd[base + off: base + off + n])
                ru.append((cur[l.name + suffix].detach() - start[l.name + suffix]).reshape(-1))
        du, ru = torch.cat(du), torch.cat(ru)
        cos = float((du * ru).sum() / (du.norm() * ru.norm() + 1e-30))
        assert cos >= 0.9, "update direction cosine %.4f" % cos
        assert 0.8 <= float(du.norm() / ru.norm()) <= 1.25
    if mode == "const" and alg == "ssdn":
        est = float(d.l_params[Denoiser.ESTIMATED_SIGMA].detach().cpu().reshape(-1)[0])
        assert est == pytest.approx(float(tr.est.detach().reshape(-1)[0]), abs=5e-5)


def test_reference_training_idiom_backward_bridge():
    """`torch.mean(outputs[LOSS]).backward()` (train.py:201) drives the HIP backward; `.grad` of the parameters is a view of
    the flat gradient buffer and torch.optim.Adam over denoiser.parameters() works as in the reference."""
    from ssdn.denoiser import Denoiser
    from ssdn.datasets import NoisyDataset
    from ssdn.params import PipelineOutput
    d = make_denoiser("ssdn", "gauss25", "known", 3)
    d.train()
    clean, noisy, ref, coords, npar = train_inputs("ssdn", "gauss25", 3)
    meta = {NoisyDataset.Metadata.INPUT_NOISE_VALUES: npar, NoisyDataset.Metadata.CLEAN: clean}
    opt = torch.optim.Adam(d.parameters(), betas=[0.9, 0.99])
    opt.zero_grad()
    out = d.run_pipeline([noisy, ref, meta])
    torch.mean(out[PipelineOutput.LOSS]).backward()
    w = d.get_model(Denoiser.MODEL, False).get_submodule("decode_block_1.2").weight
    assert w.grad is not None and float(w.grad.abs().sum()) > 0
    assert w.grad.data_ptr() >= d.flat_grad.data_ptr()
    before = w.detach().clone()
    opt.step()
    assert float((w.detach() - before).abs().max()) > 0


def test_eval_forward_any_square_size():
    """Inference at a non-training size (eval pads Kodak to 768x768, BSD300 to 512x512; here 96x96 to stay small):
    Denoiser.forward == IMG_DENOISED of the pipeline, vs the oracle."""
    from ssdn.denoiser import Denoiser
    d = make_denoiser("ssdn", "gauss25", "known", 3)
    p = R.make_params(3, 9, True, seed=5)
    d.get_model(Denoiser.MODEL, False).load_state_dict(R.reference_state_dict(p))
    d.mark_dirty()
    d.eval()
    from ssdn.datasets import NoisyDataset
    from ssdn.params import PipelineOutput
    x = R.hash_tensor((1, 3, 96, 96), 77, 0, 1)
    meta = {NoisyDataset.Metadata.INPUT_NOISE_VALUES: torch.full((1, 1, 1, 1), 25 / 255.0)}
    with torch.no_grad():
        out = d.run_pipeline([x, None, meta])
    tr = R.CpuTrainer("ssdn", 3, "gauss25", "known", params=p)
    with torch.no_grad():
        r = tr.forward(x, None, torch.full((1, 1, 1, 1), 25 / 255.0))
    o = out[PipelineOutput.IMG_DENOISED].cpu()
    assert float((o - r["out"]).norm() / r["out"].norm()) <= 5e-3
    # PSNR criterion of BASELINE.json: same weights => within 0.05 dB of the reference path
    clean = R.hash_tensor((1, 3, 96, 96), 78, 0, 1)
    assert abs(float(R.psnr(o, clean) - R.psnr(r["out"], clean))) <= 0.05

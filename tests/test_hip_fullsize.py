"""GPU: the per-GPU shards of BASELINE.json configs 3, 4 and 5 at FULL size, through the drop-in `Denoiser`, plus the
evaluation image sizes (Kodak pads to 768x768, BSD300 to 512x512; reference train.py:814-862, noise_wrapper.py:183-269).

Full sizes are where the persistent kernels run for real (k_cdma from one 256-pixel tile per CU upwards; the persistent,
statically scheduled weight-gradient path; 128x128 tiles).  Checks are the size-independent ones plus oracle bounds:
  * determinism: two executions of the same step give BIT-IDENTICAL losses and parameter gradients (no atomics, fixed-order
    reductions, fixed tile -> workgroup assignment);
  * the oracle (fp32, torch-CPU) on the same inputs: loss within 1e-2 relative; whole-gradient cosine >= 0.985 and sign
    agreement >= 0.95 (the end-to-end bounds of tests/test_hip_denoiser.py::test_training_trajectory -- fp16/bf16 storage
    flips LeakyReLU branches of near-zero activations, a zero-mean per-layer gradient noise);
  * eval: posterior-mean image within 5e-3 relative L2 of the oracle's and PSNR within 0.05 dB (BASELINE.json's criterion).
"""
import os

import pytest
import torch

import restate as R
from test_hip_denoiser import make_denoiser, _flat_grad_of

pytestmark = pytest.mark.gpu


def _inputs(B, C, P, style, seed):
    clean = R.hash_tensor((B, C, P, P), seed, 0, 1)
    if style.startswith("gauss"):
        noisy = torch.clamp(clean + R.hash_tensor((B, C, P, P), seed + 1, -1, 1) * 0.17, 0, 1)
        npar = torch.full((B, 1, 1, 1), 25 / 255.0)
    else:                                   # poisson30 (reference semantics incl. the rate-1 quirk are the data layer's business;
        lam = 30.0                          # here: any non-negative image with a plausible spread)
        noisy = torch.clamp(clean + R.hash_tensor((B, C, P, P), seed + 1, -1, 1) * torch.sqrt(clean / lam + 1e-3), 0, 1)
        npar = torch.full((B, 1, 1, 1), lam)
    return clean, noisy, npar


def _run_config(alg, style, mode, B, P, ncoords=0, seed=301, loss_rtol=1e-2, min_cos=0.985, min_agree=0.95):
    from ssdn.denoiser import Denoiser
    from ssdn.datasets import NoisyDataset
    from ssdn.params import PipelineOutput
    C = 3
    bs = alg == "ssdn"
    cout = C + C * (C + 1) // 2 if alg == "ssdn" else C
    p0 = R.make_params(C, cout, bs, seed=5)
    sp0 = R.make_params(C, 1, False, seed=6, zero_output_weights=True) if (mode == "var" and alg == "ssdn") else None
    d = make_denoiser(alg, style, mode, C)
    d.train()
    tr = R.CpuTrainer(alg, C, style, mode, params={k: v.clone() for k, v in p0.items()},
                      sigma_params={k: v.clone() for k, v in sp0.items()} if sp0 is not None else None)
    nets = [(d.get_model(Denoiser.MODEL, False), 0, tr.p)]
    if sp0 is not None:
        nets.append((d.get_model(Denoiser.SIGMA_ESTIMATOR, False), d._n_main, tr.ps))
    from test_hip_denoiser import _flat_of
    d.flat.copy_(_flat_of(d, nets, tr))
    d.mark_dirty()
    clean, noisy, npar = _inputs(B, C, P, style, seed)
    MD = NoisyDataset.Metadata
    meta = {MD.INPUT_NOISE_VALUES: npar, MD.CLEAN: clean}
    ref, coords = clean, None
    if alg == "n2v":
        ref = torch.clamp(clean + R.hash_tensor((B, C, P, P), seed + 2, -1, 1) * 0.17, 0, 1)
        coords = R.hash_tensor((B, ncoords, 2), seed + 3, 0, P).long()
        meta[MD.MASK_COORDS] = coords

    def once():
        d.flat_grad.fill_(float("nan"))
        out = d.run_pipeline([noisy, ref, meta])
        d.backward()
        torch.cuda.synchronize()
        return out[PipelineOutput.LOSS].detach().cpu().clone(), d.flat_grad.cpu().clone()

    n = d._n_main + d._n_sig + (1 if tr.est is not None else 0)       # (the flat buffer is padded to a multiple of 4)
    loss1, g1 = once()
    loss2, g2 = once()
    assert torch.isfinite(g1[:n]).all()
    assert torch.equal(loss1, loss2) and torch.equal(g1[:n], g2[:n]), "step is not bit-reproducible"

    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    r = tr.forward(noisy, ref, npar, coords)
    r["loss"].mean().backward()
    rl = r["loss"].detach()
    assert float((loss1 - rl).abs().max()) <= loss_rtol * float(rl.abs().max()) + 2e-3, (loss1.view(-1)[:4], rl.view(-1)[:4])
    gr = _flat_grad_of(d, nets, tr)
    cos = float((g1[:n] * gr[:n]).sum() / (g1[:n].norm() * gr[:n].norm() + 1e-30))
    agree = float(((g1[:n] > 0) == (gr[:n] > 0)).float().mean())
    assert cos >= min_cos and agree >= min_agree, "gradient cosine %.4f, sign agreement %.4f" % (cos, agree)
    return d


def test_config3_shard_ssdn_sigma_var_with_sigma_net_b32_64():
    """BASELINE config 3, one rank's shard: ssdn gauss25 sigma_var + sigma-estimation network, batch 32, 64x64 RGB."""
    _run_config("ssdn", "gauss25", "var", 32, 64)


def test_config4_shard_n2v_plain_net_b32_64_with_64_mask_coordinates():
    """BASELINE config 4, one rank's shard: n2v (plain net, masked MSE), batch 32, 64x64, 64 mask coordinates per patch."""
    _run_config("n2v", "gauss25", "known", 32, 64, ncoords=64)


def test_config5_shard_ssdn_poisson_sigma_const_b16_128():
    """BASELINE config 5, one rank's shard: ssdn poisson30 sigma_const, batch 16, 128x128 RGB ("large-tile LDS stress").

    At random-init weights this loss is ill-conditioned in the network output (the Poisson variance mu / lambda sits near its
    clamp for many pixels): measured with tools/path_compare.py, two kernel selections of THIS library whose activations
    differ by at most ONE fp16 ulp per element (relative L2 5e-4 at the network output) differ by 5e-3 in the loss and 9e-2
    (relative L2) in the gradient.  The bounds are therefore twice / slightly below those of the well-conditioned configs."""
    _run_config("ssdn", "poisson30", "const", 16, 128, loss_rtol=2e-2, min_cos=0.98, min_agree=0.94)


@pytest.mark.parametrize("P", [512, 768])
def test_eval_sizes_forward_vs_oracle(P):
    """Evaluation shapes: batch 2 at 512x512 (BSD300) and 768x768 (Kodak), forward only, ssdn gauss25 sigma_known."""
    from ssdn.denoiser import Denoiser
    from ssdn.datasets import NoisyDataset
    from ssdn.params import PipelineOutput
    d = make_denoiser("ssdn", "gauss25", "known", 3)
    p = R.make_params(3, 9, True, seed=5)
    d.get_model(Denoiser.MODEL, False).load_state_dict(R.reference_state_dict(p))
    d.eval()
    B = 2
    clean, noisy, npar = _inputs(B, 3, P, "gauss25", 401)
    meta = {NoisyDataset.Metadata.INPUT_NOISE_VALUES: npar}
    with torch.no_grad():
        out = d.run_pipeline([noisy, None, meta])
        again = d.run_pipeline([noisy, None, meta])
    torch.cuda.synchronize()
    o = out[PipelineOutput.IMG_DENOISED].cpu()
    assert torch.equal(o, again[PipelineOutput.IMG_DENOISED].cpu())
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    tr = R.CpuTrainer("ssdn", 3, "gauss25", "known", params=p)
    with torch.no_grad():
        r = tr.forward(noisy, None, npar)
    assert float((o - r["out"]).norm() / r["out"].norm()) <= 5e-3
    for b in range(B):
        assert abs(float(R.psnr(o[b:b + 1], clean[b:b + 1]) - R.psnr(r["out"][b:b + 1], clean[b:b + 1]))) <= 0.05

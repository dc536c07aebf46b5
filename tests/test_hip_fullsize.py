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

import numpy as np
import pytest
import torch

import restate as R
from test_hip_denoiser import make_denoiser, _flat_grad_of

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


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


# Bounds = 2-4x what this test measured on MI355X (profiles/r04_parity_fullsize.txt is its own output, from a PASSING run).  Config 2
# (the bench workload) is well conditioned: loss 4e-4, PSNR 0.002 dB, gradient norms 6e-3, per-layer cosine >= 0.9999.  Config 5 at the
# raw random initialisation is not: the network's output puts the model covariance's eigenvalues between 1e-5 and 5 against a
# Poisson variance mu * est that sits at its clamp (1.9e-5) for most pixels, the posterior mean lies in [-5, 3] (PSNR 5 dB), and a
# difference of 7.6e-4 (relative L2) in the network output -- fp16 activation storage -- moves it by > 1e-2 at 7 % of the pixels
# (tools/pme_analysis.py, profiles/r04_pme_analysis_cfg5.txt).  For that case the test does not bound the raw difference of the
# posterior mean; it DECOMPOSES it: (i) the head kernel against an fp64 evaluation of the reference's formula on the device's own
# network output (`pme_kernel`), (ii) what is left of |device - fixture| after subtracting the fp64-propagated effect of the network
# output difference, |f64(device net_out) - f64(oracle net_out)| (`pme_resid`), (iii) quantiles of the raw difference.  "cfg5b" is
# config 5 with the last layer where training takes it (oracle/fullsize.py): conditioned, tight bounds.
FULL_BOUNDS = {
    #        loss rel (vs reference)  PSNR dB   per-tensor |g| rel, first entries / |g|   per-layer / whole-gradient cosine (vs oracle)
    "cfg2": dict(loss=1.5e-3, psnr=0.01, gnorm=2e-2, ghead=2e-2, layer_cos=0.9998, cos=0.99995,
                 netout=2e-3, pme_kernel=5e-5, pme_resid=5e-4, pme_q50=6e-4, pme_q99=8e-3, pme_max=1.5e-2),
    "cfg5": dict(loss=1.2e-2, psnr=0.04, gnorm=0.09, ghead=0.15, layer_cos=0.978, cos=0.995,
                 netout=2e-3, pme_kernel=4e-3, pme_resid=3e-3, pme_q50=5e-5, pme_q99=6e-2, pme_max=None),
    "cfg5b": dict(loss=1.5e-3, psnr=0.01, gnorm=2e-2, ghead=2e-2, layer_cos=0.9995, cos=0.99995,
                  netout=1.5e-3, pme_kernel=5e-6, pme_resid=5e-4, pme_q50=3e-4, pme_q99=3e-3, pme_max=5e-3),
}


@pytest.mark.parametrize("tag", ["cfg2", "cfg5", "cfg5b"])
def test_full_size_vs_live_reference_fixture(golden_dir, tag):
    """The sizes the bench and the config-5 shard RUN, against what the LIVE REFERENCE produced there (tests/golden/g_full_*.npz,
    oracle/gen_golden_fullsize.py: its own Denoiser.run_pipeline + mean(LOSS).backward() on oracle/fullsize.py's inputs): per-sample
    loss, per-image PSNR of the posterior mean, probes of the denoised image and of mu, per-tensor gradient norms -- and, next to
    the oracle (which tests/test_oracle_golden.py pins to the same fixture at 2e-4 / 2e-3), per-layer gradient cosines."""
    import numpy as np
    import fullsize as F
    from ssdn.denoiser import Denoiser
    from ssdn.datasets import NoisyDataset
    from ssdn.params import PipelineOutput
    from test_hip_denoiser import _flat_of
    from test_oracle_golden import param_name_map
    g = np.load(os.path.join(golden_dir, "g_full_%s.npz" % tag), allow_pickle=False)
    alg, style, mode, B, P = F.CASES[tag]
    bnd = FULL_BOUNDS[tag]
    d = make_denoiser(alg, style, mode, 3)
    d.train()
    tr = R.CpuTrainer(alg, 3, style, mode, params=F.params(tag))
    net = d.get_model(Denoiser.MODEL, False)
    nets = [(net, 0, tr.p)]
    d.flat.copy_(_flat_of(d, nets, tr))
    d.mark_dirty()
    clean, noisy, npar = F.inputs(tag)
    MD = NoisyDataset.Metadata
    out = d.run_pipeline([noisy, clean, {MD.INPUT_NOISE_VALUES: npar, MD.CLEAN: clean}])
    d.backward()
    torch.cuda.synchronize()
    lines = ["%s: %s %s sigma_%s, batch %d, %dx%d" % (tag, alg, style, mode, B, P, P)]
    # ---- against the reference fixture ----
    loss, ref_loss = out[PipelineOutput.LOSS].detach().cpu().reshape(-1), torch.from_numpy(g["loss"]).reshape(-1)
    loss_rel = float((loss - ref_loss).abs().max() / ref_loss.abs().max())
    lines.append("  loss: max |dev - ref| / max |ref| = %.3e   (bound %.1e)" % (loss_rel, bnd["loss"]))
    pme, mu = out[PipelineOutput.IMG_DENOISED].cpu(), out[PipelineOutput.IMG_MU].cpu()
    probe = float((pme[:, :, 3::16, 5::16] - torch.from_numpy(g["out_probe"])).abs().max())
    probe_mu = float((mu[:, :, 3::16, 5::16] - torch.from_numpy(g["mu_probe"])).abs().max())
    dps = max(abs(float(R.psnr(pme[b:b + 1], clean[b:b + 1])) - float(g["psnr_out"][b])) for b in range(B))
    lines.append("  posterior mean probe: max abs diff %.3e; mu probe %.3e; per-image PSNR max |diff| %.4f dB (bound %.2f)" % (probe, probe_mu, dps, bnd["psnr"]))
    no_dev = d._last_engine.main.tensor("out32").detach().cpu().clone()
    gd = d.flat_grad.cpu()
    worst_gn = worst_head = 0.0
    for name, (which, key) in param_name_map(g["names"]).items():
        if which == "est":
            o = d._n_main + d._n_sig
            mine, headm = float(gd[o].abs()), gd[o:o + 1]
        else:
            l = next(x for x in net.layers if key.startswith(x.name + "."))
            sl = slice(l.w_off, l.w_off + l.M * l.cin * l.k * l.k) if key.endswith("weight") else slice(l.b_off, l.b_off + l.M)
            mine, headm = float(gd[sl].double().norm()), gd[sl][:16]
        want = float(g["gnorm/" + name])
        rel = abs(mine - want) / (want + 1e-12)
        worst_gn = max(worst_gn, rel)
        href = torch.from_numpy(g["ghead/" + name]).reshape(-1)
        worst_head = max(worst_head, float((headm[:href.numel()] - href).abs().max()) / (want + 1e-12))
    lines.append("  per-tensor gradient norm: max relative difference %.3e (bound %.1e); first 16 entries of every tensor: max |diff| / |g| %.3e (bound %.1e)" % (
        worst_gn, bnd["gnorm"], worst_head, bnd["ghead"]))
    # ---- against the oracle (pinned to the same fixture by the CPU suite): direction of every layer's gradient ----
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    r = tr.forward(noisy, clean, npar)
    r["loss"].mean().backward()
    gr = _flat_grad_of(d, nets, tr)
    n = d._n_main + (1 if tr.est is not None else 0)
    gd64, gr64 = gd.double(), gr.double()
    cos_all = float((gd64[:n] * gr64[:n]).sum() / (gd64[:n].norm() * gr64[:n].norm() + 1e-300))
    worst_cos, worst_layer = 1.0, ""
    for l in net.layers:
        sl = slice(l.w_off, l.w_off + l.M * l.cin * l.k * l.k)
        c = float((gd64[sl] * gr64[sl]).sum() / (gd64[sl].norm() * gr64[sl].norm() + 1e-300))
        if c < worst_cos:
            worst_cos, worst_layer = c, l.name
    agree = float(((gd[:n] > 0) == (gr[:n] > 0)).float().mean())
    lines.append("  gradient vs fp32 oracle: whole cosine %.6f (bound %.4f), worst layer %s %.6f (bound %.4f), sign agreement %.4f" % (
        cos_all, bnd["cos"], worst_layer, worst_cos, bnd["layer_cos"], agree))
    # ---- the posterior mean, decomposed ----
    with torch.no_grad():
        no_ref = r["net_out"].detach()
        est64 = tr.est.detach().double() if tr.est is not None else None
        f64 = lambda no: R.ssdn_head(no.double(), noisy.double(), npar.double(), style, mode, est64)["out"]   # noqa: E731
        p64_dev, p64_ref = f64(no_dev), f64(no_ref)
    pr = lambda t: t[:, :, 3::16, 5::16]   # noqa: E731
    netout_rel = float((no_dev - no_ref).norm() / no_ref.norm())
    pme_kernel = float((pr(pme).double() - pr(p64_dev)).abs().max())
    raw = (pr(pme).double() - torch.from_numpy(g["out_probe"]).double()).abs()
    resid = float((raw - (pr(p64_dev) - pr(p64_ref)).abs()).clamp(min=0).max())
    q50, q99 = float(raw.median()), float(np.quantile(raw.numpy().ravel(), 0.99))
    lines.append("  network output vs oracle: rel-L2 %.3e (bound %.1e); head kernel vs fp64 reference formula on the device's own net_out: max %.3e (bound %.1e)" % (
        netout_rel, bnd["netout"], pme_kernel, bnd["pme_kernel"]))
    lines.append("  posterior mean vs fixture at %d probes: median %.3e (bound %.1e), 99%% %.3e (bound %.1e), max %.3e (bound %s), %d probes > 1e-2; "
                 "not explained by |f64(device net_out) - f64(oracle net_out)|: max %.3e (bound %.1e)" % (
                     raw.numel(), q50, bnd["pme_q50"], q99, bnd["pme_q99"], float(raw.max()), bnd["pme_max"], int((raw > 1e-2).sum()), resid, bnd["pme_resid"]))
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_fullsize_%s.txt" % tag), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    assert loss_rel <= bnd["loss"] and dps <= bnd["psnr"] and worst_gn <= bnd["gnorm"] and worst_head <= bnd["ghead"], lines
    assert cos_all >= bnd["cos"] and worst_cos >= bnd["layer_cos"], lines
    assert netout_rel <= bnd["netout"] and pme_kernel <= bnd["pme_kernel"] and resid <= bnd["pme_resid"], lines
    assert q50 <= bnd["pme_q50"] and q99 <= bnd["pme_q99"] and (bnd["pme_max"] is None or float(raw.max()) <= bnd["pme_max"]), lines


# configs 3 and 4 (round 5): one rank's shard at full size against the LIVE reference's fixture -- bounds of the
# config-2 class (both are well conditioned); measured values: profiles/r05_parity_fullsize.txt (this test's own output, from a passing run)
FULL_BOUNDS_34 = {
    "cfg3": dict(loss=1.5e-3, psnr=0.01, gnorm=2e-2, ghead=2e-2, layer_cos=0.9995, cos=0.99995, probe=2e-2),
    "cfg4": dict(loss=1.5e-3, psnr=0.01, gnorm=2e-2, ghead=2e-2, layer_cos=0.9995, cos=0.99995, probe=5e-3),
}


@pytest.mark.parametrize("tag", ["cfg3", "cfg4"])
def test_full_size_configs_3_and_4_vs_live_reference_fixture(golden_dir, tag):
    """BASELINE configs 3 (ssdn gauss25 sigma_var: blind-spot network + sigma-estimation network; reference denoiser.py:76-88,261-265)
    and 4 (Noise2Void: plain network, masked MSE at 64 coordinates per patch against a second noisy realisation; denoiser.py:159-180,
    utils/n2v_loss.py:6-17), one rank's shard (batch 32, 64x64), against tests/golden/g_full_cfg3|cfg4.npz -- what the live reference's
    Denoiser.run_pipeline + mean(LOSS).backward() produced on oracle/fullsize.py's inputs: per-sample loss, per-image PSNR, probes of
    the output image, per-tensor gradient norms and first entries of BOTH networks; and, next to the oracle (pinned to the same fixtures
    by tests/test_oracle_golden.py), the cosine of every layer's gradient."""
    import numpy as np
    import fullsize as F
    from ssdn.denoiser import Denoiser
    from ssdn.datasets import NoisyDataset
    from ssdn.params import PipelineOutput
    from test_hip_denoiser import _flat_of
    from test_oracle_golden import param_name_map
    g = np.load(os.path.join(golden_dir, "g_full_%s.npz" % tag), allow_pickle=False)
    alg, style, mode, B, P = F.CASES[tag]
    bnd = FULL_BOUNDS_34[tag]
    d = make_denoiser(alg, style, mode, 3)
    d.train()
    tr = R.CpuTrainer(alg, 3, style, mode, params=F.params(tag), sigma_params=F.sigma_params(tag))
    net = d.get_model(Denoiser.MODEL, False)
    nets = [(net, 0, tr.p)]
    snet = None
    if F.sigma_params(tag) is not None:
        snet = d.get_model(Denoiser.SIGMA_ESTIMATOR, False)
        nets.append((snet, d._n_main, tr.ps))
    d.flat.copy_(_flat_of(d, nets, tr))
    d.mark_dirty()
    clean, noisy, npar = F.inputs(tag)
    MD = NoisyDataset.Metadata
    meta = {MD.INPUT_NOISE_VALUES: npar, MD.CLEAN: clean}
    ref, coords = clean, None
    if alg == "n2v":
        ref, coords = F.n2v_extras(tag)
        meta[MD.MASK_COORDS] = coords
    out = d.run_pipeline([noisy, ref, meta])
    d.backward()
    torch.cuda.synchronize()
    lines = ["%s: %s %s sigma_%s, batch %d, %dx%d" % (tag, alg, style, mode, B, P, P)]
    loss, ref_loss = out[PipelineOutput.LOSS].detach().cpu().reshape(-1), torch.from_numpy(g["loss"]).reshape(-1)
    loss_rel = float((loss - ref_loss).abs().max() / ref_loss.abs().max())
    img = out[PipelineOutput.IMG_DENOISED].cpu()
    probe = float((img[:, :, 3::16, 5::16] - torch.from_numpy(g["out_probe"])).abs().max())
    dps = max(abs(float(R.psnr(img[b:b + 1], clean[b:b + 1])) - float(g["psnr_out"][b])) for b in range(B))
    lines.append("  loss: max |dev - ref| / max |ref| = %.3e (bound %.1e); output image probe max abs diff %.3e (bound %.1e); per-image PSNR max |diff| %.4f dB (bound %.2f)" % (
        loss_rel, bnd["loss"], probe, bnd["probe"], dps, bnd["psnr"]))
    gd = d.flat_grad.cpu()
    worst_gn = worst_head = 0.0
    worst_name = ""
    for name, (which, key) in param_name_map(g["names"]).items():
        nn, base = (snet, d._n_main) if which == "sigma" else (net, 0)
        l = next(x for x in nn.layers if key.startswith(x.name + "."))
        sl = slice(base + l.w_off, base + l.w_off + l.M * l.cin * l.k * l.k) if key.endswith("weight") else slice(base + l.b_off, base + l.b_off + l.M)
        mine, headm = float(gd[sl].double().norm()), gd[sl][:16]
        want = float(g["gnorm/" + name])
        rel = abs(mine - want) / (want + 1e-12)
        if rel > worst_gn:
            worst_gn, worst_name = rel, name
        href = torch.from_numpy(g["ghead/" + name]).reshape(-1)
        worst_head = max(worst_head, float((headm[:href.numel()] - href).abs().max()) / (want + 1e-12))
    lines.append("  per-tensor gradient norm (%d tensors): max relative difference %.3e at %s (bound %.1e); first 16 entries: max |diff| / |g| %.3e (bound %.1e)" % (
        len(g["names"]), worst_gn, worst_name, bnd["gnorm"], worst_head, bnd["ghead"]))
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    r = tr.forward(noisy, ref, npar, coords)
    r["loss"].mean().backward()
    gr = _flat_grad_of(d, nets, tr)
    n = d._n_main + d._n_sig
    gd64, gr64 = gd.double(), gr.double()
    cos_all = float((gd64[:n] * gr64[:n]).sum() / (gd64[:n].norm() * gr64[:n].norm() + 1e-300))
    worst_cos, worst_layer = 1.0, ""
    for nn, base in ((net, 0),) + (((snet, d._n_main),) if snet is not None else ()):
        for l in nn.layers:
            sl = slice(base + l.w_off, base + l.w_off + l.M * l.cin * l.k * l.k)
            c = float((gd64[sl] * gr64[sl]).sum() / (gd64[sl].norm() * gr64[sl].norm() + 1e-300))
            if c < worst_cos:
                worst_cos, worst_layer = c, ("sigma/" if base else "") + l.name
    agree = float(((gd[:n] > 0) == (gr[:n] > 0)).float().mean())
    lines.append("  gradient vs fp32 oracle: whole cosine %.6f (bound %.5f), worst layer %s %.6f (bound %.4f), sign agreement %.4f" % (
        cos_all, bnd["cos"], worst_layer, worst_cos, bnd["layer_cos"], agree))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "parity_fullsize_%s.txt" % tag), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    assert loss_rel <= bnd["loss"] and dps <= bnd["psnr"] and probe <= bnd["probe"] and worst_gn <= bnd["gnorm"] and worst_head <= bnd["ghead"], lines
    assert cos_all >= bnd["cos"] and worst_cos >= bnd["layer_cos"], lines


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
    if P == 512:
        # ... and against the LIVE reference: tests/golden/g_eval_512.npz holds what the reference's own Denoiser (eval mode) returns for the
        # first image of this batch (oracle/gen_golden_eval.py): strided probe of the denoised image and of mu, the PSNR, the output's norm.
        # Same bounds as against the oracle: 5e-3 relative on the probe, 0.05 dB.
        g = np.load(os.path.join(GOLDEN, "g_eval_512.npz"))
        probe = o[:1, :, 3::16, 5::16].double()
        want = torch.from_numpy(g["out_probe"]).double()
        assert float((probe - want).norm() / want.norm()) <= 5e-3
        mu = out[PipelineOutput.IMG_MU].cpu()[:1, :, 3::16, 5::16].double()
        wmu = torch.from_numpy(g["mu_probe"]).double()
        assert float((mu - wmu).norm() / wmu.norm()) <= 5e-3
        assert abs(float(R.psnr(o[:1], clean[:1])) - float(g["psnr_out"][0])) <= 0.05
        assert abs(float(o[:1].double().norm()) / float(g["out_norm"]) - 1.0) <= 5e-3


def test_full_size_config2_properties():
    """Size-independent properties at the bench workload's size (config 2, batch 32, 64x64): the whole step is BIT-reproducible -- two
    runs of forward + loss + backward from the same weights give identical loss and identical gradients (no floating-point atomics
    anywhere; the chip-wide weight-gradient launch hands blocks to CUs in whatever order they free up, every block owns its slab, the
    slabs are summed in a fixed order) -- and a second Denoiser built from the same seed reproduces them."""
    import fullsize as F
    from ssdn.datasets import NoisyDataset
    from ssdn.params import PipelineOutput
    alg, style, mode, B, P = F.CASES["cfg2"]
    clean, noisy, npar = F.inputs("cfg2")
    MD = NoisyDataset.Metadata
    res = []
    for rep in range(2):
        torch.manual_seed(21)
        d = make_denoiser(alg, style, mode, 3)
        d.train()
        for again in range(2):
            out = d.run_pipeline([noisy, clean, {MD.INPUT_NOISE_VALUES: npar, MD.CLEAN: clean}])
            d.backward()
            torch.cuda.synchronize()
            res.append((out[PipelineOutput.LOSS].detach().cpu().clone(), d.flat_grad.detach().cpu().clone()))
    assert torch.isfinite(res[0][1]).all() and float(res[0][1].abs().max()) > 0
    for loss, grad in res[1:]:
        assert torch.equal(loss, res[0][0]) and torch.equal(grad, res[0][1])

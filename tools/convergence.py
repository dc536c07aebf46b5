#!/usr/bin/env python
"""Convergence / PSNR evidence for the fp16-forward / bf16-gradient arithmetic of the HIP path (BASELINE.json:
"PSNR ... matches the reference within +-0.05 dB").

Two FREE-RUNNING trainings from the same initial weights on the same stream of structured synthetic images (smooth random
textures; gauss25 noise, sigma known, 64x64 RGB, batch 8):
    device : Denoiser.train_step  (libssdn_hip.so: fp16 activations, bf16 gradients, fused Adam)
    oracle : oracle/restate.CpuTrainer.step (fp32 torch-CPU autograd + Adam -- the reference's operator family)
Nothing is re-synchronised between them.  Reported:
  * the two loss curves (raw + exponentially smoothed) and their gap;
  * eval PSNR of BOTH resulting weight sets on a held-out synthetic set, each evaluated by the SAME evaluator (the fp32 oracle
    forward), plus the device-evaluated PSNR of the device weights;  |delta PSNR| is the number BASELINE's criterion is about;
  * per-layer cosine between the device's and the oracle's parameter gradients at step 0 (identical weights) -- the measured
    values behind the bounds in tests/test_hip_denoiser.py / test_hip_ops.py.
Usage (GPU box):  python tools/convergence.py [--steps 300] [--out gpurun_out/convergence.json]
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "selfsupervised-denoising_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import torch  # noqa: E402


def textures(n, P, seed):
    """smooth random RGB textures in [0,1]: sums of a few oriented sinusoids + a soft blob, different per image"""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(P, dtype=torch.float32), torch.arange(P, dtype=torch.float32), indexing="ij")
    out = torch.zeros(n, 3, P, P)
    for i in range(n):
        img = torch.zeros(3, P, P)
        for _ in range(4):
            fx, fy, ph = (torch.rand(3, generator=g) * torch.tensor([0.5, 0.5, 6.28])).tolist()
            amp = torch.rand(3, generator=g).view(3, 1, 1) * 0.25
            img += amp * torch.sin(xx * fx + yy * fy + ph)
        cx, cy, r = (torch.rand(3, generator=g) * torch.tensor([P, P, P / 3]) + torch.tensor([0, 0, 4.0])).tolist()
        img += 0.3 * torch.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r)) * (torch.rand(3, generator=g).view(3, 1, 1) - 0.5)
        out[i] = (img + 0.5).clamp(0, 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--patch", type=int, default=64)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "convergence.json"))
    args = ap.parse_args()
    import restate as R
    import ssdn
    from ssdn.datasets import NoisyDataset
    from ssdn.denoiser import Denoiser
    from ssdn.params import ConfigValue, NoiseAlgorithm, NoiseValue, PipelineOutput
    from test_hip_denoiser import _flat_grad_of, _flat_of

    B, P, N = args.batch, args.patch, args.steps
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    cfg = ssdn.cfg.base()
    cfg[ConfigValue.ALGORITHM] = NoiseAlgorithm.SELFSUPERVISED_DENOISING
    cfg[ConfigValue.NOISE_STYLE] = "gauss25"
    cfg[ConfigValue.NOISE_VALUE] = NoiseValue.KNOWN
    ssdn.cfg.infer(cfg, model_only=True)
    torch.manual_seed(0)
    d = Denoiser(cfg, device="cuda:0")                     # reference-style random init (He-normal), zero biases
    net = d.get_model(Denoiser.MODEL, False)
    p0 = {k.replace("output_conv", "output_block.4"): v.detach().cpu().clone() for k, v in net.state_dict().items() if not k.startswith("output_conv")}
    tr = R.CpuTrainer("ssdn", 3, "gauss25", "known", params={k: v.clone() for k, v in p0.items()})
    nets = [(net, 0, tr.p)]
    assert torch.equal(_flat_of(d, nets, tr)[:d._n_main], d.flat.cpu()[:d._n_main])
    MD = NoisyDataset.Metadata
    sigma = 25 / 255.0
    npar = torch.full((B, 1, 1, 1), sigma)
    total_images = N * B
    gnoise = torch.Generator().manual_seed(99)
    dev_loss, ora_loss, layer_cos = [], [], {}
    t0 = time.time()
    d.train()
    for it in range(N):
        clean = textures(B, P, 1000 + it)
        noisy = (clean + torch.randn(clean.shape, generator=gnoise) * sigma).clamp(0, 1)
        lr = R.trainer_lr(it * B, total_images)
        meta = {MD.INPUT_NOISE_VALUES: npar, MD.CLEAN: clean}
        if it == 0:                # gradients at identical weights
            out = d.run_pipeline([noisy, None, meta])
            d.backward()
            torch.cuda.synchronize()
            r = tr.forward(noisy, None, npar)
            for t in tr.leaves:
                t.grad = None
            r["loss"].mean().backward()
            gd, gr = d.flat_grad.cpu(), _flat_grad_of(d, nets, tr)
            for l in net.layers:
                sl = slice(l.w_off, l.w_off + l.M * l.cin * l.k * l.k)
                a, b = gd[sl], gr[sl]
                layer_cos[l.name] = {"cos": float((a * b).sum() / (a.norm() * b.norm() + 1e-30)),
                                     "rel_l2": float((a - b).norm() / (b.norm() + 1e-30))}
            n = d._n_main
            layer_cos["ALL"] = {"cos": float((gd[:n] * gr[:n]).sum() / (gd[:n].norm() * gr[:n].norm() + 1e-30)),
                                "sign_agreement": float(((gd[:n] > 0) == (gr[:n] > 0)).float().mean())}
        out = d.train_step([noisy, None, meta], lr)
        dev_loss.append(float(out[PipelineOutput.LOSS].mean()))
        r = tr.step(lr, noisy, None, npar)
        ora_loss.append(float(r["loss"].mean()))
        if it % 25 == 0:
            print("step %4d  lr %.2e  loss device %.4f  oracle %.4f   (%.0f s)" % (it, lr, dev_loss[-1], ora_loss[-1], time.time() - t0), flush=True)

    def smooth(xs, a=0.9):
        s, out = xs[0], []
        for x in xs:
            s = a * s + (1 - a) * x
            out.append(s)
        return out

    # held-out evaluation: both weight sets through the SAME fp32 evaluator; device weights also through the device
    Ne = 16
    eclean = textures(Ne, P, 777000)
    enoisy = (eclean + torch.randn(eclean.shape, generator=torch.Generator().manual_seed(5)) * sigma).clamp(0, 1)
    enpar = torch.full((Ne, 1, 1, 1), sigma)
    dev_params = {k.replace("output_conv", "output_block.4"): v.detach().cpu().clone() for k, v in net.state_dict().items() if not k.startswith("output_conv")}
    with torch.no_grad():
        ev_dev = R.CpuTrainer("ssdn", 3, "gauss25", "known", params=dev_params).forward(enoisy, None, enpar)
        ev_ora = tr.forward(enoisy, None, enpar)
        d.eval()
        dout = d.run_pipeline([enoisy, None, {MD.INPUT_NOISE_VALUES: enpar}])[PipelineOutput.IMG_DENOISED].cpu()
    psnr = lambda a: float(R.psnr(a, eclean).mean())      # noqa: E731
    res = {
        "setup": "ssdn gauss25 sigma_known, %dx%d RGB, batch %d, %d free-running steps, reference LR schedule over %d images, structured synthetic textures" % (P, P, B, N, total_images),
        "loss_device": dev_loss, "loss_oracle": ora_loss,
        "smoothed_gap_last50": float(sum(abs(a - b) for a, b in zip(smooth(dev_loss)[-50:], smooth(ora_loss)[-50:])) / 50),
        "loss_mean_last50": {"device": sum(dev_loss[-50:]) / 50, "oracle": sum(ora_loss[-50:]) / 50},
        "psnr_noisy_input": psnr(enoisy),
        "psnr_oracle_weights_fp32_eval": psnr(ev_ora["out"]),
        "psnr_device_weights_fp32_eval": psnr(ev_dev["out"]),
        "psnr_device_weights_device_eval": psnr(dout),
        "delta_psnr_training_arithmetic_dB": psnr(ev_dev["out"]) - psnr(ev_ora["out"]),
        "delta_psnr_eval_arithmetic_dB": psnr(dout) - psnr(ev_dev["out"]),
        "gradient_agreement_step0": layer_cos,
        "seconds": time.time() - t0,
    }
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if not k.startswith("loss_") or k == "loss_mean_last50"}, indent=1))


if __name__ == "__main__":
    main()

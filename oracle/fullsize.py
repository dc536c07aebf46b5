"""TEST INFRASTRUCTURE -- deterministic inputs of the FULL-SIZE parity fixtures (BASELINE.json config 2: ssdn gauss25 sigma_known,
batch 32, 64x64 RGB -- the workload bench.py times; one rank's shard of configs 3 (sigma_var + sigma-estimation network), 4 (Noise2Void)
and 5 (ssdn poisson30 sigma_const, batch 16, 128x128)).

Used by oracle/gen_golden_fullsize.py (live reference -> tests/golden/g_full_*.npz), tests/test_oracle_golden.py (restatement vs
those fixtures, CPU) and tests/test_hip_fullsize.py (device vs those fixtures, GPU).  Images are smooth synthetic textures
(sums of oriented sinusoids + a blob: the regime the network trains in; hash-noise images make LeakyReLU branches of near-zero
activations flip under 16-bit storage and say little about training); noise follows the reference's styles, drawn with fixed
torch CPU generators (same torch build here and on the GPU box)."""
import torch

CASES = {
    #  tag      algorithm style       mode     B   P
    "cfg2": ("ssdn", "gauss25", "known", 32, 64),
    # BASELINE configs 3 and 4, one rank's shard (round 5): the sigma-estimation network next to the blind-spot network; Noise2Void --
    # the plain network, masked MSE at 64 coordinates per patch against a second noisy realisation
    "cfg3": ("ssdn", "gauss25", "var", 32, 64),
    "cfg4": ("n2v", "gauss25", "known", 32, 64),
    "cfg5": ("ssdn", "poisson30", "const", 16, 128),
    # config 5 again with the network's LAST layer in the regime a few hundred optimisation steps bring it to (output mean near the
    # image mean, small model covariance): at the raw random initialisation the Poisson variance mu * est sits at its 1e-3 clamp for
    # most pixels and the posterior mean is ill-conditioned in the network output (tools/pme_analysis.py) -- "cfg5" keeps that case,
    # this one measures the kernels where the loss is conditioned
    "cfg5b": ("ssdn", "poisson30", "const", 16, 128),
}


def sigma_params(tag):
    """weights of the sigma-estimation network ("cfg3"): restate.make_params(seed=6) with a NON-zero last layer, so that the sigma path
    carries signal (the reference initialises that layer to zero, noise_network.py:180-183); None for the other cases"""
    import restate as R
    return R.make_params(3, 1, False, seed=6) if CASES[tag][2] == "var" and CASES[tag][0] == "ssdn" else None


def params(tag):
    """the closed-form weights of restate.make_params(seed=5); "cfg5b": last layer scaled by 1/4, bias of the three mean channels 0.5"""
    import restate as R
    if CASES[tag][0] != "ssdn":
        return R.make_params(3, 3, False, seed=5)
    p = R.make_params(3, 9, True, seed=5)
    if tag == "cfg5b":
        with torch.no_grad():
            p["output_block.4.weight"] *= 0.25
            p["output_block.4.bias"][:3] = 0.5
    return p


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


def inputs(tag):
    """-> (clean [B,3,P,P], noisy, noise parameter [B,1,1,1]); 8-bit clean images like the data layer's"""
    alg, style, mode, B, P = CASES[tag]
    clean = (textures(B, P, 4242 + P + (17 if tag == "cfg5b" else 0)) * 255).round() / 255
    g = torch.Generator().manual_seed(7 + P)
    if style.startswith("gauss"):
        sigma = 25 / 255.0
        noisy = (clean + torch.randn(clean.shape, generator=g) * sigma).clamp(0, 1)
        npar = torch.full((B, 1, 1, 1), sigma)
    else:                       # the reference's Poisson style: rate-1 noise on lambda x (utils/noise.py:101-104)
        lam = 30.0
        noisy = ((clean * lam + torch.poisson(torch.ones(clean.shape), generator=g)) / lam).clamp(0, 1)
        npar = torch.full((B, 1, 1, 1), lam)
    return clean, noisy, npar


def n2v_extras(tag):
    """Noise2Void ("cfg4"): the reference image = a second noisy realisation of the clean image, and 64 mask coordinates per patch
    (the reference's sampler draws one per 8x8 box, utils/n2v_ups.py:65-88; the loss applies patch 0's to every patch, n2v_loss.py:12)"""
    alg, style, mode, B, P = CASES[tag]
    clean, _, _ = inputs(tag)
    g = torch.Generator().manual_seed(99 + P)
    ref = (clean + torch.randn(clean.shape, generator=g) * (25 / 255.0)).clamp(0, 1)
    box = P // 8
    by, bx = torch.meshgrid(torch.arange(8), torch.arange(8), indexing="ij")
    off = torch.randint(0, box, (B, 64, 2), generator=g)
    coords = torch.stack([by.reshape(-1) * box, bx.reshape(-1) * box], dim=1).unsqueeze(0) + off
    return ref, coords.long()

"""where do the outlier pixels of the config-5 posterior mean come from?
Device (fp16 activations) vs fp32 oracle (= the reference's arithmetic, pinned to the live-reference fixture) vs fp64 evaluations of the
SAME closed form on each side's own network output.  GPU only.  usage: python tools/pme_analysis.py [cfg5|cfg2]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "selfsupervised-denoising_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import restate as R
import fullsize as F
from ssdn.denoiser import Denoiser
from ssdn.datasets import NoisyDataset
from ssdn.params import PipelineOutput
from test_hip_fullsize import make_denoiser
from test_hip_denoiser import _flat_of

tag = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
alg, style, mode, B, P = F.CASES[tag]
g = np.load(os.path.join(ROOT, "tests", "golden", "g_full_%s.npz" % tag))
d = make_denoiser(alg, style, mode, 3)
d.eval()
tr = R.CpuTrainer(alg, 3, style, mode, params=F.params(tag))
net = d.get_model(Denoiser.MODEL, False)
d.flat.copy_(_flat_of(d, [(net, 0, tr.p)], tr))
d.mark_dirty()
clean, noisy, npar = F.inputs(tag)
MD = NoisyDataset.Metadata
with torch.no_grad():
    out = d.run_pipeline([noisy, clean, {MD.INPUT_NOISE_VALUES: npar, MD.CLEAN: clean}])
    torch.cuda.synchronize()
    pme_dev = out[PipelineOutput.IMG_DENOISED].cpu()
    no_dev = d._last_engine.main.tensor("out32").cpu().clone()
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    r = tr.forward(noisy, clean, npar)
    pme_ref, no_ref = r["out"], r["net_out"]
    est = tr.est.double() if tr.est is not None else None
    h64 = lambda no: R.ssdn_head(no.double(), noisy.double(), npar.double(), style, mode, est)["out"]   # noqa: E731
    pme64_dev, pme64_ref = h64(no_dev), h64(no_ref)
    # the same fp64 closed form in the well-conditioned arrangement the kernel uses: mu + Sx (Sx + Sn)^-1 (y - mu)
    def stable64(no):
        no = no.double()
        mu, a = no[:, :3], no[:, 3:].permute(0, 2, 3, 1)
        z = torch.zeros_like(a[..., 0])
        A = torch.stack([torch.stack([a[..., 0], a[..., 1], a[..., 2]], -1), torch.stack([z, a[..., 3], a[..., 4]], -1), torch.stack([z, z, a[..., 5]], -1)], -1)
        sx = A.transpose(-1, -2) @ A
        e = (torch.nn.functional.softplus(est - 4.0) + 1e-3) if est is not None else None
        m = mu.clamp(min=1e-3)
        if style.startswith("gauss"):
            var = (npar.double().clamp(min=1e-3) ** 2 if mode == "known" else e ** 2).expand_as(mu)
        else:
            var = m / npar.double() if mode == "known" else m * e
        sn = torch.diag_embed(var.permute(0, 2, 3, 1))
        dd = (noisy.double() - mu).permute(0, 2, 3, 1)[..., None]
        return (mu.permute(0, 2, 3, 1)[..., None] + sx @ torch.linalg.solve(sx + sn, dd))[..., 0].permute(0, 3, 1, 2), sx, sn
    st_dev, sx_dev, sn_dev = stable64(no_dev)
    st_ref, sx_ref, _ = stable64(no_ref)

def stat(name, a, b):
    dlt = (a.double() - b.double()).abs()
    print("%-64s max %.3e  99.9%% %.3e  99%% %.3e  median %.3e  #>1e-2: %d of %d" % (
        name, float(dlt.max()), float(dlt.flatten().quantile(0.999)) if dlt.numel() < 2 ** 24 else float(np.quantile(dlt.numpy().ravel(), 0.999)),
        float(np.quantile(dlt.numpy().ravel(), 0.99)), float(dlt.median()), int((dlt > 1e-2).sum()), dlt.numel()))
    return dlt

print("%s: %s %s sigma_%s, batch %d, %dx%d; network output: rel-L2(dev - oracle) = %.3e, max abs %.3e" % (
    tag, alg, style, mode, B, P, P, float((no_dev - no_ref).norm() / no_ref.norm()), float((no_dev - no_ref).abs().max())))
pr = lambda t: t[:, :, 3::16, 5::16]   # noqa: E731
fx = torch.from_numpy(g["out_probe"])
print("-- at the fixture's probe pixels (every 16th) --")
stat("device PME vs live-reference fixture", pr(pme_dev), fx)
stat("fp32 oracle PME vs live-reference fixture", pr(pme_ref), fx)
stat("fp64 closed form on the ORACLE's net_out vs fixture", pr(st_ref), fx)
stat("fp64 closed form on the DEVICE's net_out vs fixture", pr(st_dev), fx)
print("-- probes, against the reference's OWN formula (1e-6 regularisers included) in fp64 --")
stat("head kernel vs fp64 reference formula, both on the device's net_out", pr(pme_dev), pr(pme64_dev))
stat("fixture (reference fp32) vs fp64 reference formula on the oracle's net_out", fx, pr(pme64_ref))
prop = (pr(pme64_dev) - pr(pme64_ref)).abs()
resid = ((pr(pme_dev).double() - fx.double()).abs() - prop).clamp(min=0)
print("device-vs-fixture deviation NOT explained by the propagated net_out difference |f64(dev net_out) - f64(oracle net_out)|: max %.3e, #>1e-3: %d" % (float(resid.max()), int((resid > 1e-3).sum())))
print("-- all pixels --")
stat("head kernel vs fp64 reference formula, both on the device's net_out (all pixels)", pme_dev, pme64_dev)
stat("fp32 oracle vs fp64 reference formula on the oracle's net_out (all pixels)", pme_ref, pme64_ref)
stat("device kernel vs fp64 closed form, both on the device's net_out", pme_dev, st_dev)
stat("fp32 oracle (reference's inverse-of-inverses) vs fp64, oracle net_out", pme_ref, st_ref)
stat("reference's formula in fp64 vs stable form in fp64 (oracle net_out)", pme64_ref, st_ref)
dd = stat("fp64 PME: device's net_out vs oracle's net_out", st_dev, st_ref)
stat("device PME vs fp32 oracle PME", pme_dev, pme_ref)
# where are the pixels that move, and how are they conditioned?
idx = (dd.amax(1) > 1e-2).nonzero()
ev = torch.linalg.eigvalsh(sx_ref)
snd = torch.diagonal(sn_dev, dim1=-2, dim2=-1)
print("pixels whose fp64 posterior mean moves by > 1e-2 between the two network outputs: %d of %d" % (len(idx), B * P * P))
for (b, y, x) in idx[:12].tolist():
    a_d, a_r = no_dev[b, 3:, y, x], no_ref[b, 3:, y, x]
    print("  (b=%d,y=%d,x=%d): |y-mu| %.3f  eig(Sx) %s  noise var %s  d(net_out A) max %.2e  PME dev %s oracle %s" % (
        b, y, x, float((noisy[b, :, y, x] - no_ref[b, :3, y, x]).abs().max()), ["%.1e" % v for v in ev[b, y, x].tolist()],
        ["%.1e" % v for v in snd[b, y, x].tolist()], float((a_d - a_r).abs().max()),
        ["%.3f" % v for v in st_dev[b, :, y, x].tolist()], ["%.3f" % v for v in st_ref[b, :, y, x].tolist()]))
gain = (ev[..., -1] / (ev[..., -1] + snd.amin(-1)))
print("largest eigenvalue of Sx / (that + smallest noise variance): median %.3f, 1%% %.3f, 99%% %.3f" % (float(gain.median()), float(gain.flatten().quantile(0.01)), float(gain.flatten().quantile(0.99))))
for b in range(min(B, 4)):
    print("  image %d: PSNR device %.4f, oracle %.4f, fp64(dev net_out) %.4f dB" % (b, float(R.psnr(pme_dev[b:b + 1], clean[b:b + 1])), float(R.psnr(pme_ref[b:b + 1], clean[b:b + 1])), float(R.psnr(st_dev[b:b + 1].float(), clean[b:b + 1]))))

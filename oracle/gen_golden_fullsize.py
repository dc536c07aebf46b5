"""TEST INFRASTRUCTURE -- full-size parity fixtures from the LIVE reference (build container only: needs /root/reference).

    python oracle/gen_golden_fullsize.py [tag ...] -> tests/golden/g_full_cfg2|cfg3|cfg4|cfg5|cfg5b.npz   (a few KB each)

The small fixtures of oracle/gen_golden.py are all batch 2, <= 64x64.  These two drive the reference's own Denoiser
(run_pipeline + torch.mean(LOSS).backward(), train.py:200-201) at the sizes the bench and the config-5 shard run: batch 32 at
64x64 (BASELINE config 2) and batch 16 at 128x128 (config 5), on the inputs of oracle/fullsize.py with the closed-form weights of
restate.make_params(seed=5).  Stored: per-sample loss, per-tensor gradient norm + its first 16 entries, a strided probe of the
denoised image and of mu.  Only outputs are stored; inputs and weights regenerate from their seeds."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
import restate as R  # noqa: E402
import fullsize as F  # noqa: E402

OUT = os.environ.get("SSDN_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")      # (override: tests/test_oracle_golden.py regenerates into a scratch directory)
torch.set_num_threads(8)


def main():
    ref = ref_shim.import_reference()
    with ref_shim.reference_modules(ref):
        import ssdn
        from ssdn.denoiser import Denoiser
        from ssdn.datasets import NoisyDataset
        from ssdn.params import ConfigValue, NoiseAlgorithm, NoiseValue, PipelineOutput
        MD = NoisyDataset.Metadata
        only = sys.argv[1:]
        for tag, (alg, style, mode, B, P) in F.CASES.items():
            if only and tag not in only:
                continue
            cfg = ssdn.cfg.base()
            cfg[ConfigValue.ALGORITHM] = NoiseAlgorithm(alg)
            cfg[ConfigValue.NOISE_STYLE] = style
            cfg[ConfigValue.NOISE_VALUE] = NoiseValue(mode)
            cfg[ConfigValue.IMAGE_CHANNELS] = 3
            ssdn.cfg.infer(cfg, model_only=True)
            d = Denoiser(cfg, device="cpu")
            d._models[Denoiser.MODEL].load_state_dict(R.reference_state_dict(F.params(tag)))
            if F.sigma_params(tag) is not None:
                d._models[Denoiser.SIGMA_ESTIMATOR].load_state_dict(R.reference_state_dict(F.sigma_params(tag)))
            clean, noisy, npar = F.inputs(tag)
            meta = {MD.INPUT_NOISE_VALUES: npar, MD.IMAGE_SHAPE: None, MD.CLEAN: clean}
            refimg = clean
            if alg == "n2v":
                refimg, coords = F.n2v_extras(tag)
                meta[MD.MASK_COORDS] = coords
            o = d.run_pipeline([noisy, refimg, meta])
            torch.mean(o[PipelineOutput.LOSS]).backward()
            arrs = {"names": np.array([n for n, _ in d.named_parameters()]), "loss": o[PipelineOutput.LOSS].detach().numpy(),
                    "out_probe": o[PipelineOutput.IMG_DENOISED].detach()[:, :, 3::16, 5::16].numpy(),
                    "psnr_out": np.array([float(ssdn.utils.calculate_psnr(o[PipelineOutput.IMG_DENOISED].detach()[b:b + 1], clean[b:b + 1])) for b in range(B)])}
            if PipelineOutput.IMG_MU in o:
                arrs["mu_probe"] = o[PipelineOutput.IMG_MU].detach()[:, :, 3::16, 5::16].numpy()
            if PipelineOutput.NOISE_STD_DEV in o:
                arrs["noise_std"] = o[PipelineOutput.NOISE_STD_DEV].detach().reshape(B, -1)[:, :4].numpy()
            for n, prm in d.named_parameters():
                f = prm.grad.reshape(-1)
                arrs["gnorm/" + n] = np.float64(f.double().norm())
                arrs["ghead/" + n] = f[:16].clone().numpy()
            np.savez_compressed(os.path.join(OUT, "g_full_%s.npz" % tag), **arrs)
            print("wrote g_full_%s" % tag, "loss[:3]", arrs["loss"].reshape(-1)[:3], "tensors", len(arrs["names"]))


if __name__ == "__main__":
    main()

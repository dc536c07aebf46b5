"""TEST INFRASTRUCTURE -- evaluation-size parity fixture from the LIVE reference (build container only: needs /root/reference).

    python oracle/gen_golden_eval.py -> tests/golden/g_eval_512.npz   (a few KB)

The reference's own Denoiser in eval mode (run_pipeline under torch.no_grad(), ssdn/ssdn/eval.py / denoiser.py:112-126) on ONE 512x512 RGB
image (the BSD300 shape class; gauss25, sigma known) with the closed-form weights of restate.make_params(seed=5).  The input is the
FIRST image of the batch tests/test_hip_fullsize.py::test_eval_sizes_forward_vs_oracle builds (restate.hash_tensor, seed 401): the
device test compares its own result for that image against the numbers stored here.  Stored: a strided probe of the denoised image
(every 16th pixel), its PSNR against the clean image, the probe of mu."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
import restate as R  # noqa: E402

OUT = os.environ.get("SSDN_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")      # (override: tests/test_oracle_golden.py regenerates into a scratch directory)
torch.set_num_threads(8)
P, SEED = 512, 401


def eval_inputs(B, P=P, seed=SEED):
    """the batch of test_eval_sizes_forward_vs_oracle (tests/test_hip_fullsize.py::_inputs, gauss25)"""
    clean = R.hash_tensor((B, 3, P, P), seed, 0, 1)
    noisy = torch.clamp(clean + R.hash_tensor((B, 3, P, P), seed + 1, -1, 1) * 0.17, 0, 1)
    return clean, noisy, torch.full((B, 1, 1, 1), 25 / 255.0)


def main():
    ref = ref_shim.import_reference()
    with ref_shim.reference_modules(ref):
        import ssdn
        from ssdn.denoiser import Denoiser
        from ssdn.datasets import NoisyDataset
        from ssdn.params import ConfigValue, NoiseAlgorithm, NoiseValue, PipelineOutput
        MD = NoisyDataset.Metadata
        cfg = ssdn.cfg.base()
        cfg[ConfigValue.ALGORITHM] = NoiseAlgorithm("ssdn")
        cfg[ConfigValue.NOISE_STYLE] = "gauss25"
        cfg[ConfigValue.NOISE_VALUE] = NoiseValue("known")
        cfg[ConfigValue.IMAGE_CHANNELS] = 3
        ssdn.cfg.infer(cfg, model_only=True)
        d = Denoiser(cfg, device="cpu")
        d._models[Denoiser.MODEL].load_state_dict(R.reference_state_dict(R.make_params(3, 9, True, seed=5)))
        d.eval()
        clean, noisy, npar = eval_inputs(2)
        clean, noisy, npar = clean[:1], noisy[:1], npar[:1]
        with torch.no_grad():
            o = d.run_pipeline([noisy, None, {MD.INPUT_NOISE_VALUES: npar, MD.IMAGE_SHAPE: None, MD.CLEAN: clean}])
        out = o[PipelineOutput.IMG_DENOISED]
        arrs = {"out_probe": out[:, :, 3::16, 5::16].numpy(), "mu_probe": o[PipelineOutput.IMG_MU][:, :, 3::16, 5::16].numpy(),
                "psnr_out": np.array([float(ssdn.utils.calculate_psnr(out, clean))]), "out_norm": np.float64(out.double().norm())}
        np.savez_compressed(os.path.join(OUT, "g_eval_%d.npz" % P), **arrs)
        print("wrote g_eval_%d" % P, "psnr", arrs["psnr_out"], "probe", arrs["out_probe"].shape)


if __name__ == "__main__":
    main()

"""`DenoiserEvaluator` -- evaluation driver (drop-in for /root/reference/ssdn/ssdn/eval.py:18-125): load a `.wt` or `.training`
file, pad every test image by reflection to a x32 multiple / square / uniform size (Kodak 768x768, BSD300 512x512), run the
forward kernels, un-pad, per-image PSNR -> `psnrs.csv`, images -> `eval_imgs/`.  Reference defect fixed: `runs_dir` is honoured
(eval.py:32-35 drops it)."""
import logging
import os
from typing import Callable, Dict

import torch

import ssdn
from ssdn.cfg import DEFAULT_RUN_DIR
from ssdn.datasets import NoisyDataset
from ssdn.denoiser import Denoiser
from ssdn.params import PipelineOutput
from ssdn.train import DenoiserTrainer

logger = logging.getLogger("ssdn.eval")


class DenoiserEvaluator(DenoiserTrainer):
    RUN_KIND = "eval"

    def __init__(self, target_path: str, runs_dir: str = DEFAULT_RUN_DIR, run_dir: str = None):
        super().__init__({}, runs_dir=runs_dir)
        sd = torch.load(target_path, map_location="cpu", weights_only=False)
        if "denoiser" in sd:
            self.load_state_dict(sd)
        else:
            self.denoiser = Denoiser.from_state_dict(sd)
        self.cfg = self.denoiser.cfg
        self._run_dir = run_dir
        self.init_state()

    def evaluate(self):
        self.reset_metrics(train=False)
        if self.denoiser is None:
            raise RuntimeError("Denoiser not initialised for evaluation")
        _ = self.writer
        ssdn.logging_helper.setup(self.run_dir_path, "log.txt")
        logger.info("Loading Test Dataset...")
        self.testloader, self.testset, self.test_sampler = self.test_data()
        logger.info(ssdn.utils.separator())
        logger.info("EVALUATION STARTED")
        self._evaluate(self.testloader, self.evaluation_output_callback(self.testset))
        logger.info(self.eval_state_str("EVALUATION RESULT"))
        logger.info("EVALUATION FINISHED")
        logger.info(ssdn.utils.separator())
        return {k: float(torch.as_tensor(m.accumulated()).float().mean())
                for k, m in self.state[ssdn.params.StateValue.HISTORY][ssdn.params.HistoryValue.EVAL].items()
                if isinstance(m, ssdn.utils.Metric) and not m.empty()}

    def evaluation_output_callback(self, dataset) -> Callable[[int, Dict], None]:
        """psnrs.csv gets a row for EVERY evaluated instance; images are saved for the first pass over the dataset only
        (the test sets are evaluated several times with fresh noise, cfg.test_length)."""
        def callback(output_0_index: int, outputs: Dict):
            inp = outputs[PipelineOutput.INPUTS][NoisyDataset.INPUT]
            metadata = outputs[PipelineOutput.INPUTS][NoisyDataset.METADATA]
            n = inp.shape[0]
            remaining = len(dataset) - output_0_index
            if remaining > 0:
                self.save_image_outputs(outputs, os.path.join(self.run_dir_path, "eval_imgs"), "img_{index:05}_{desc}.png",
                                        batch_indexes=range(min(remaining, n)))
            with open(os.path.join(self.run_dir_path, "psnrs.csv"), "a") as f:
                if output_0_index == 0:
                    f.write(",".join(["id", "psnr_nsy"] + list(self.img_outputs(prefix="psnr").values())) + "\n")
                clean = metadata[NoisyDataset.Metadata.CLEAN]
                pairs = zip(NoisyDataset.unpad(inp.cpu(), metadata), NoisyDataset.unpad(clean, metadata))
                values = [torch.stack([ssdn.utils.calculate_psnr(a, b) for a, b in pairs])]
                values += [self.calculate_psnr(outputs, key, unpad=True).cpu() for key in self.img_outputs(prefix="psnr")]
                for i in range(n):
                    f.write(",".join(["{:04d}".format(output_0_index + i)] + ["{:.4f}".format(float(v[i])) for v in values]) + "\n")
        return callback

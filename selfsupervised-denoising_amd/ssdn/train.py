"""`DenoiserTrainer` -- the training driver of the `ssdn` package, MI355X edition (drop-in for
/root/reference/ssdn/ssdn/train.py:46-909: same constructor, `train()`, `evaluate()`, `snapshot()`, `state_dict()` /
`load_state_dict()`, `set_train_data` / `set_test_data`, `resume_run()`, run-directory naming, metric names, `.wt` and
`.training` file layouts).

What is kept: iterations are counted in IMAGES; the evaluation / print / snapshot intervals are checked BEFORE the step
(train.py:159-190); the learning rate follows `compute_ramped_lrate` with the two fractions passed swapped (train.py:276-282 ->
10 % ramp-up, 30 % ramp-down, lr(0) = 0: the original paper's schedule); per-step metrics (loss, PSNR of every image output,
noise / model std x 255) are averaged over a print interval.

What is different underneath: a step is `Denoiser.train_step` (planned HIP op lists + fused Adam; no autograd graph, no
torch.optim); data parallelism is one process per GPU (`torch.distributed` env of torchrun): every rank draws the SAME global
sampling order and takes its rows of each global minibatch (ssdn.hip.dp.shard_rows), gradients are all-reduced overlapped with
the backward pass, rank 0 owns the run directory.  The optimiser state is stored in torch.optim.Adam's state-dict layout, so
`.training` files interchange with the reference.

Reference defects decided here: the `self.test_dataself.test_data()` typo that breaks any run with a
validation set (train.py:142) and the `len(None)` crash of `update_eta` without one (train.py:673) are FIXED; `timings["eta"]`
is a number from the start; `torch.load(..., weights_only=False)` for the pickled enums.
"""
from __future__ import annotations

import glob
import logging
import math
import os
import re
from collections import defaultdict
from typing import Callable, Dict, Optional, Tuple, Union

import torch
from torch import Tensor
from torch.utils.data import DataLoader

import ssdn
from ssdn.cfg import DEFAULT_RUN_DIR
from ssdn.datasets import CleanPatches, DevicePatchStream, FixedLengthSampler, HDF5Dataset, NoisyDataset, SamplingOrder, UnlabelledImageFolderDataset
from ssdn.datasets.transforms import RandomCrop
from ssdn.denoiser import Denoiser
from ssdn.models import NoiseNetwork
from ssdn.params import ConfigValue, DatasetType, HistoryValue, Pipeline, PipelineOutput, StateValue
from ssdn.utils import Metric, MetricDict, TrackedTime, separator

logger = logging.getLogger("ssdn.train")
_PSNR_CACHE = "_psnr_per_sample"          # key of the per-sample PSNR values the device metric kernel produced for an output dict


class _RankShard(torch.utils.data.Sampler):
    """Batches of dataset indexes for ONE rank: the global order is cut into global minibatches and rank r takes rows
    [r*B/W, (r+1)*B/W) of each (ssdn.hip.dp.shard_rows) -- a W-GPU run consumes the order of a 1-GPU run.

    Every rank yields the SAME number of batches (a rank that ran out early would leave the others blocked in the gradient
    all-reduce): a final partial global minibatch -- TRAIN_ITERATIONS not a multiple of the global batch -- is NOT sharded,
    every rank processes all of it (identical rows on every rank: the 1/W-averaged all-reduce is then the gradient of that
    batch).  `counts` carries, in order, the number of GLOBAL samples each yielded batch stands for; the trainer advances
    ITERATION (LR schedule, sampler index in checkpoints) by that, not by rows x world."""

    def __init__(self, sampler: FixedLengthSampler, global_batch: int, rank: int, world: int):
        from collections import deque
        from ssdn.hip.dp import shard_rows
        self.sampler, self.global_batch = sampler, global_batch
        self.lo, self.hi = shard_rows(global_batch, rank, world)
        self.counts = deque()

    def __iter__(self):
        batch = []
        self.counts.clear()          # (counts of an abandoned earlier pass -- prefetched, never trained on -- must not leak into this one)
        for idx in self.sampler:
            batch.append(idx)
            if len(batch) == self.global_batch:
                self.counts.append(self.global_batch)
                yield batch[self.lo:self.hi]
                batch = []
        if batch:
            self.counts.append(len(batch))
            yield batch

    def __len__(self):
        return -(-len(self.sampler) // self.global_batch)


class DenoiserTrainer:
    def __init__(self, cfg: Dict, state: Optional[Dict] = None, runs_dir: str = DEFAULT_RUN_DIR, run_dir: str = None):
        self.runs_dir = os.path.abspath(runs_dir)
        self._run_dir = run_dir
        self._writer = None
        self.cfg = cfg
        if self.cfg:
            ssdn.cfg.infer(self.cfg)
        self.state = {} if state is None else state
        self._denoiser: Optional[Denoiser] = None
        self._train_iter = None
        self.trainloader = self.trainset = self.train_sampler = None
        self.testloader = self.testset = self.test_sampler = None
        # one process per GPU (torchrun environment): join the job and select this rank's device BEFORE anything is allocated
        # (world 1: selects the device only; no GPU: gloo)
        from ssdn.hip import dp
        self.rank, self.world, self.local_rank = dp.init_from_env()
        self._exchange = None
        self._shard: Optional[_RankShard] = None
        self.device_data = None          # None: device-side patch preparation whenever a GPU is present (see train_data)

    # ---- target -----------------------------------------------------------------------------------------------------------
    @property
    def denoiser(self) -> Denoiser:
        return self._denoiser

    @denoiser.setter
    def denoiser(self, denoiser: Denoiser):
        self._denoiser = denoiser
        self.init_optimiser()

    def init_optimiser(self):
        """The optimiser is the Denoiser's fused Adam (betas 0.9 / 0.99, train.py:100-107); a fresh target starts from zero moments."""
        self._exchange = None

    def new_target(self):
        # (a single-process program keeps the device its caller selected; a rank of a job uses its own GPU)
        device = "cuda:%d" % (self.local_rank if self.world > 1 else torch.cuda.current_device()) if torch.cuda.is_available() else None
        self.denoiser = Denoiser(self.cfg, device=device)
        self.init_state()

    def _sync_replicas(self):
        """Identical replicas: rank 0's parameters (and optimiser moments) go to every rank -- the user's seed, whatever it
        was, decides the initial weights, and the ranks' RNG states stay different (noise realisations and Noise2Void masks must
        not be correlated across the shards of a global minibatch)."""
        if self.world <= 1:
            return
        import torch.distributed as dist
        d = self._denoiser
        for t in (d.flat, d.adam_m, d.adam_v):
            dist.broadcast(t, src=0)
        steps = torch.tensor([d.adam_steps], dtype=torch.int64, device=d.flat.device)
        dist.broadcast(steps, src=0)
        d.adam_steps = int(steps.item())
        d.mark_dirty()

    def init_state(self):
        self.state[StateValue.INITIALISED] = True
        self.state[StateValue.ITERATION] = 0
        self.state[StateValue.HISTORY] = {HistoryValue.TRAIN: MetricDict(), HistoryValue.EVAL: MetricDict(),
                                          HistoryValue.TIMINGS: defaultdict(TrackedTime)}
        self.reset_metrics()

    # ---- the loop ---------------------------------------------------------------------------------------------------------
    @property
    def learning_rate(self) -> float:
        c = self.cfg
        return ssdn.utils.compute_ramped_lrate(self.state[StateValue.ITERATION], c[ConfigValue.TRAIN_ITERATIONS],
                                               c[ConfigValue.LR_RAMPDOWN_FRACTION], c[ConfigValue.LR_RAMPUP_FRACTION],
                                               c[ConfigValue.LEARNING_RATE])

    def train(self):
        if self.denoiser is None:
            self.new_target()
        denoiser = self.denoiser
        self._sync_replicas()                # whatever built or loaded the target: every rank starts from rank 0's weights
        if self.rank == 0:
            _ = self.writer
            ssdn.logging_helper.setup(self.run_dir_path, "log.txt")
        logger.info(separator())
        logger.info("Loading Training Dataset...")
        self.trainloader, self.trainset, self.train_sampler = self.train_data()
        logger.info("Loaded Training Dataset.")
        if self.cfg[ConfigValue.TEST_DATA_PATH]:
            logger.info("Loading Validation Dataset...")
            self.testloader, self.testset, self.test_sampler = self.test_data()
            logger.info("Loaded Validation Dataset.")
        if self.world > 1 and self._exchange is None:
            self._exchange = denoiser.gradient_exchange(self.world)
        logger.info(separator())
        logger.info("TRAINING STARTED")
        logger.info(separator())
        history = self.state[StateValue.HISTORY]
        train_history = history[HistoryValue.TRAIN]
        MD = NoisyDataset.Metadata
        data_itr = iter(self.trainloader)
        # The collector and a 1.7 ms step: a generation-2 pass over this process's heap (torch, the planned engines, the dataset) takes longer
        # than a training step.  Everything alive now goes to the permanent generation (gc.freeze), so the automatic passes inside the loop
        # only see what the loop itself allocated; a full collection runs where the loop stops anyway, at PRINT_INTERVAL.  bench.py times
        # its steps under the same policy.
        import gc
        gc.collect()
        gc.freeze()
        while True:
            iteration = self.state[StateValue.ITERATION]
            if iteration % self.cfg[ConfigValue.EVAL_INTERVAL] == 0 and self.testloader is not None:
                self._evaluate(self.testloader, output_callback=self.validation_output_callback(0))
            if iteration % self.cfg[ConfigValue.PRINT_INTERVAL] == 0:
                self._flush_device_metrics()
                history[HistoryValue.TIMINGS]["total"].update()
                last_print = history[HistoryValue.TIMINGS]["last_print"]
                last_print.update()
                self.update_eta(history[HistoryValue.EVAL]["n"] + train_history["n"], last_print.total)
                if self.rank == 0:
                    logger.info(self.state_str(eval_prefix="VALID"))
                    self.write_metrics(eval_prefix="valid")
                last_print.total = 0
                self.reset_metrics()
                gc.collect()
            if iteration % self.cfg[ConfigValue.SNAPSHOT_INTERVAL] == 0 and self.rank == 0:
                self.snapshot()
            if iteration >= self.cfg[ConfigValue.TRAIN_ITERATIONS]:
                break
            data = next(data_itr)
            image_count = data[NoisyDataset.INPUT].shape[0]
            denoiser.train()
            # H11: on a GPU the step's metric sums stay on the device (one kernel inside train_step, SSDN_OP_METRICS) and come to the
            # host when the trainer prints; the reference's per-step tensor arithmetic (train.py:205-218) is the CPU path
            on_device = getattr(denoiser, "device", torch.device("cpu")).type == "cuda" and torch.is_tensor(data[NoisyDataset.METADATA].get(MD.CLEAN))
            outputs = denoiser.train_step(data, self.learning_rate, self._exchange, **({"metrics": True} if on_device else {}))
            train_history["n"] += image_count
            if not on_device:
                with torch.no_grad():
                    train_history["loss"] += outputs[PipelineOutput.LOSS]
                    for key, name in self.img_outputs(prefix="psnr").items():
                        train_history[name] += self.calculate_psnr(outputs, key, False)
                    for key in (PipelineOutput.NOISE_STD_DEV, PipelineOutput.MODEL_STD_DEV):
                        if key in outputs:
                            train_history[key.value] += outputs[key] * 255
            # images consumed by the whole job: rows x world for a sharded minibatch, the true count for the un-sharded tail
            self.state[StateValue.ITERATION] += self._shard.counts.popleft() if self._shard is not None else image_count
        logger.info(separator())
        logger.info("TRAINING FINISHED")
        logger.info(separator())
        if self.rank == 0:
            self.snapshot()
            self.snapshot(output_name="final-{}.wt".format(self.denoiser.config_name()), subdir="", model_only=True)

    def _flush_device_metrics(self):
        """Bring the sums the training steps left on the device (Denoiser.accumulate_metrics) into the history's Metric objects:
        one 64-byte read per PRINT_INTERVAL instead of ~20 launches and a `Metric +=` per step."""
        d = self._denoiser
        if d is None or getattr(d, "device", torch.device("cpu")).type != "cuda" or not hasattr(d, "read_metrics"):
            return
        train_history = self.state[StateValue.HISTORY][HistoryValue.TRAIN]
        for name, (total, count) in d.read_metrics("train", reset=True).items():
            m = train_history[name]
            m.total = total if m.total is None else m.total + total
            m.n += count

    def evaluate(self, dataloader: DataLoader, output_callback: Callable = None):
        self.reset_metrics(train=False)
        return self._evaluate(dataloader, output_callback)

    def _evaluate(self, dataloader, output_callback: Optional[Callable]):
        self.denoiser.eval()
        with torch.no_grad():
            eval_history = self.state[StateValue.HISTORY][HistoryValue.EVAL]
            idx = 0
            for data in dataloader:
                image_count = data[NoisyDataset.INPUT].shape[0]
                outputs = self.denoiser.run_pipeline(data)
                eval_history["n"] += image_count
                if getattr(self.denoiser, "device", torch.device("cpu")).type == "cuda":
                    # per-image PSNR over the un-padded extent from ONE kernel (the values are needed on the host: psnrs.csv)
                    per = self.denoiser.accumulate_metrics(data, "eval", with_loss=False, per_sample=True)
                    outputs[_PSNR_CACHE] = {key: per[name] for key, name in self.img_outputs(prefix="psnr").items() if name in per}
                for key, name in self.img_outputs(prefix="psnr").items():
                    eval_history[name] += self.calculate_psnr(outputs, key, unpad=True)
                if output_callback:
                    output_callback(idx, outputs)
                idx += image_count

    # ---- outputs ----------------------------------------------------------------------------------------------------------
    def validation_output_callback(self, output_index: int) -> Callable:
        def callback(output_0_index: int, outputs: Dict):
            n = outputs[PipelineOutput.INPUTS][NoisyDataset.INPUT].shape[0]
            bi = output_index - output_0_index
            if 0 <= bi < n and self.rank == 0:
                self._save_image_outputs(outputs, os.path.join(self.run_dir_path, "val_imgs"), "{iter:08}_{desc}.png", bi)
        return callback

    def save_image_outputs(self, outputs: Dict, output_dir: str, fileformat: str, batch_indexes=None):
        if batch_indexes is None:
            clean = outputs[PipelineOutput.INPUTS][NoisyDataset.METADATA][NoisyDataset.Metadata.CLEAN]
            batch_indexes = range(clean.shape[0])
        for bi in batch_indexes:
            self._save_image_outputs(outputs, output_dir, fileformat, bi)

    def _save_image_outputs(self, outputs: Dict, output_dir: str, fileformat: str, batch_index: int):
        os.makedirs(output_dir, exist_ok=True)
        metadata = outputs[PipelineOutput.INPUTS][NoisyDataset.METADATA]
        MD = NoisyDataset.Metadata

        def save(img: Tensor, desc: str):
            name = fileformat.format(iter=self.state[StateValue.ITERATION], index=int(metadata[MD.INDEXES][batch_index]), desc=desc)
            ssdn.utils.save_tensor_image(NoisyDataset.unpad(img.cpu(), metadata, batch_index), os.path.join(output_dir, name))

        if MD.CLEAN in metadata:
            save(metadata[MD.CLEAN], "cln")
        save(outputs[PipelineOutput.INPUTS][NoisyDataset.INPUT], "nsy")
        if PipelineOutput.IMG_DENOISED in outputs:
            save(outputs[PipelineOutput.IMG_DENOISED], "out")
        if PipelineOutput.IMG_MU in outputs:
            save(outputs[PipelineOutput.IMG_MU], "out-mu")
        if PipelineOutput.MODEL_STD_DEV in outputs:
            save(outputs[PipelineOutput.MODEL_STD_DEV][:, None, ...] / (10.0 / 255), "out-std")

    def snapshot(self, output_name: str = None, subdir: str = None, model_only: bool = False):
        """`models/model_XXXXXXXX.wt` (Denoiser.state_dict()) or `training/model_XXXXXXXX.training` (everything to resume)."""
        if not model_only:
            self._flush_device_metrics()         # (the history in a `.training` file holds everything trained on so far)
        if subdir is None:
            subdir = "models" if model_only else "training"
        out_dir = os.path.join(self.run_dir_path, subdir)
        os.makedirs(out_dir, exist_ok=True)
        it = self.state[StateValue.ITERATION]
        if output_name is None:
            output_name = ("model_{:08d}.wt" if model_only else "model_{:08d}.training").format(it)
        obj = self.denoiser.state_dict() if model_only else self.state_dict()
        torch.save(_to_cpu(obj), os.path.join(out_dir, output_name))

    def write_metrics(self, eval_prefix: str = "eval"):
        it = self.state[StateValue.ITERATION]
        hist = self.state[StateValue.HISTORY]
        for prefix, md in (("train", hist[HistoryValue.TRAIN]), (eval_prefix, hist[HistoryValue.EVAL])):
            for key, metric in md.items():
                if isinstance(metric, Metric) and not metric.empty():
                    self.writer.add_scalar(prefix + "/" + key, float(torch.as_tensor(metric.accumulated()).float().mean()), it)
            if prefix == "train":
                self.writer.add_scalar("train/learning_rate", self.learning_rate, it)

    def state_str(self, eval_prefix: str = "EVAL") -> str:
        s = self.train_state_str()
        if self.state[StateValue.HISTORY][HistoryValue.EVAL]["n"] > 0:
            s = os.linesep.join([s, self.eval_state_str(prefix="{:10} {:>5}".format("", eval_prefix))])
        return s

    @staticmethod
    def _metric_strs(md) -> list:
        return ["{}={:8.2f}".format(k, float(torch.as_tensor(m.accumulated()).float().mean()))
                for k, m in md.items() if isinstance(m, Metric) and not m.empty()]

    def train_state_str(self) -> str:
        hist = self.state[StateValue.HISTORY]
        eta = hist[HistoryValue.TIMINGS].get("eta", None)
        eta_s = "???" if not isinstance(eta, (int, float)) else ("<1s" if eta < 1 else ssdn.utils.seconds_to_dhms(eta))
        parts = self._metric_strs(hist[HistoryValue.TRAIN])
        s = "[{:08d}] {:>5} | ".format(self.state[StateValue.ITERATION], "TRAIN") + ", ".join(parts)
        if parts:
            s += " | "
        return s + "[{} ~ ETA: {}]".format(ssdn.utils.seconds_to_dhms(hist[HistoryValue.TIMINGS]["total"].total, trim=False), eta_s)

    def eval_state_str(self, prefix: str = "EVAL") -> str:
        return "{} | ".format(prefix) + ", ".join(self._metric_strs(self.state[StateValue.HISTORY][HistoryValue.EVAL]))

    def reset_metrics(self, eval: bool = True, train: bool = True):
        if train and hasattr(getattr(self, "denoiser", None), "reset_device_metrics"):
            self.denoiser.reset_device_metrics("train")     # (sums the steps since the last flush left on the device go with the host Metrics)
        hist = self.state[StateValue.HISTORY]
        for on, key in ((train, HistoryValue.TRAIN), (eval, HistoryValue.EVAL)):
            if on:
                hist[key]["n"] = 0
                for m in hist[key].values():
                    if isinstance(m, Metric):
                        m.reset()

    def img_outputs(self, prefix: str = None) -> Dict:
        outs = {PipelineOutput.IMG_DENOISED: "out"}
        if self.cfg[ConfigValue.PIPELINE] == Pipeline.SSDN:
            outs[PipelineOutput.IMG_MU] = "mu_out"
        return {k: ("_".join((prefix, v)) if prefix else v) for k, v in outs.items()}

    @staticmethod
    def calculate_psnr(outputs: Dict, output: PipelineOutput, unpad: bool = True) -> Tensor:
        cached = outputs.get(_PSNR_CACHE)
        if cached is not None and output in cached and unpad:
            return cached[output]
        metadata = outputs[PipelineOutput.INPUTS][NoisyDataset.METADATA]
        clean = metadata[NoisyDataset.Metadata.CLEAN]
        img = outputs[output]
        if unpad:
            pairs = zip(NoisyDataset.unpad(img, metadata), NoisyDataset.unpad(clean, metadata))
            return torch.stack([ssdn.utils.calculate_psnr(a, b.to(a.device)) for a, b in pairs])
        return ssdn.utils.calculate_psnr(img, clean.to(img.device))

    # ---- run directory ----------------------------------------------------------------------------------------------------
    @property
    def writer(self):
        os.makedirs(self.run_dir_path, exist_ok=True)
        if self._writer is None:
            self._writer = ssdn.logging_helper.ScalarWriter(self.run_dir_path, purge_step=self.state[StateValue.ITERATION] + 1)
        return self._writer

    @property
    def run_dir_path(self) -> str:
        return os.path.join(self.runs_dir, self.run_dir)

    RUN_KIND = "train"

    @property
    def run_dir(self) -> str:
        if self._run_dir is None:
            self._run_dir = "{:05d}-{}-{}".format(self.next_run_id(), self.RUN_KIND, self.config_name())
        return self._run_dir

    def next_run_id(self) -> int:
        ids = []
        if os.path.exists(self.runs_dir):
            for path, _, _ in os.walk(self.runs_dir):
                head = os.path.basename(path).split("-")[0]
                if head.isdigit():
                    ids.append(int(head))
        return max(ids) + 1 if ids else 0

    def update_eta(self, samples: int, elapsed: float, smoothing_factor: float = 0.95) -> float:
        timings = self.state[StateValue.HISTORY][HistoryValue.TIMINGS]
        prev = timings.get("eta", None)
        if samples <= 0:
            return prev
        remaining = self.cfg[ConfigValue.TRAIN_ITERATIONS] - self.state[StateValue.ITERATION]
        if self.testloader is not None:
            remaining += len(self.testloader) * math.ceil(remaining / self.cfg[ConfigValue.EVAL_INTERVAL])
        new = elapsed / samples * remaining
        timings["eta"] = new if not isinstance(prev, (int, float)) else smoothing_factor * new + (1 - smoothing_factor) * prev
        return timings["eta"]

    def config_name(self) -> str:
        it = self.state.get(StateValue.ITERATION, 0) or self.cfg[ConfigValue.TRAIN_ITERATIONS]
        it_s = "iter%dm" % (it // 1000000) if it >= 1000000 else ("iter%dk" % (it // 1000) if it >= 1000 else "iter%d" % it)
        parts = [ssdn.cfg.config_name(self.cfg), it_s]
        if self.cfg.get(ConfigValue.TEST_DATASET_NAME) is not None:
            parts.insert(0, self.cfg[ConfigValue.TEST_DATASET_NAME])
        if self.cfg.get(ConfigValue.TRAIN_DATASET_NAME) is not None:
            parts.insert(0, self.cfg[ConfigValue.TRAIN_DATASET_NAME])
        return "-".join(parts)

    # ---- checkpoint (.training) -------------------------------------------------------------------------------------------
    def state_dict(self) -> Dict:
        order = self.train_sampler.last_iter().state_dict() if self.train_sampler is not None and self.train_sampler.last_iter() is not None \
            else {"order": [], "index": 0}
        order = dict(order)
        order["index"] = self.state[StateValue.ITERATION]      # what was PROCESSED, not what the loader has prefetched
        sd = {"denoiser": self.denoiser.state_dict(), "state": self.state, "train_order_iter": order,
              "optimizer": self.denoiser.optimizer_state_dict(self.learning_rate), "rng": torch.get_rng_state()}
        # (an extra key the reference's loader ignores) the device patch stream's Philox key and minibatch counter: a resumed run
        # continues the noise stream instead of restarting it at offset 0 under a re-drawn key
        if isinstance(self.trainloader, DevicePatchStream):
            sd["device_stream"] = self.trainloader.state_dict()
        return sd

    def load_state_dict(self, state_dict: Union[Dict, str]):
        if isinstance(state_dict, str):
            state_dict = torch.load(state_dict, map_location="cpu", weights_only=False)
        self.denoiser = Denoiser.from_state_dict(state_dict["denoiser"])
        self.cfg = self.denoiser.cfg
        self.state = state_dict["state"]
        self._train_iter = SamplingOrder.from_state_dict(state_dict["train_order_iter"])
        self.denoiser.load_optimizer_state_dict(state_dict["optimizer"])
        self._stream_state = state_dict.get("device_stream")
        torch.set_rng_state(state_dict["rng"])
        if self.world > 1 and self.rank > 0:
            # the file holds rank 0's generator state: the other ranks continue from a state derived from it, not from a copy
            torch.manual_seed(int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) + self.rank)

    # ---- data -------------------------------------------------------------------------------------------------------------
    def _open(self, path, dtype, transform):
        ch = self.cfg[ConfigValue.IMAGE_CHANNELS]
        if dtype == DatasetType.FOLDER:
            return UnlabelledImageFolderDataset(path, channels=ch, transform=transform, recursive=True)
        if dtype == DatasetType.HDF5:
            return HDF5Dataset(path, transform=transform, channels=ch)
        raise NotImplementedError("Dataset type not implemented")

    def train_data(self) -> Tuple[DataLoader, NoisyDataset, FixedLengthSampler]:
        cfg = self.cfg
        crop = RandomCrop(cfg[ConfigValue.TRAIN_PATCH_SIZE], pad_if_needed=True, padding_mode="reflect")
        dataset = NoisyDataset(self._open(cfg[ConfigValue.TRAIN_DATA_PATH], cfg[ConfigValue.TRAIN_DATASET_TYPE], crop),
                               cfg[ConfigValue.NOISE_STYLE], cfg[ConfigValue.ALGORITHM], pad_uniform=False,
                               pad_multiple=NoiseNetwork.input_wh_mul(), square=cfg[ConfigValue.BLINDSPOT], training_mode=True)
        _ = dataset[0]
        if self.world > 1:
            g = torch.get_rng_state()
            torch.manual_seed(1234 + self.state.get(StateValue.ITERATION, 0))      # the same global order on every rank
        sampler = FixedLengthSampler(dataset, num_samples=cfg[ConfigValue.TRAIN_ITERATIONS], shuffled=True)
        if self._train_iter is not None:
            order = self._train_iter
            missing = cfg[ConfigValue.TRAIN_ITERATIONS] - len(order.order)
            if missing > 0:      # resumed with MORE iterations than the stored order holds (the reference would stop short with
                extra = FixedLengthSampler(dataset, num_samples=missing, shuffled=True)      # StopIteration): extend it
                order = SamplingOrder(list(order.order) + list(extra.sampler()), order.index)
            sampler.for_next_iter(order)
            self._train_iter = None
        kw = dict(num_workers=cfg[ConfigValue.DATALOADER_WORKERS], pin_memory=cfg[ConfigValue.PIN_DATA_MEMORY] or torch.cuda.is_available())
        # N2: on a GPU the workers ship the clean patch as uint8 and the per-sample noise / Noise2Void manipulation / metadata
        # are produced for the whole minibatch on the device (ssdn.datasets.device_stream); SSDN_HOST_DATA=1 keeps the
        # reference's host-side preparation
        device_stream = (self.device_data if self.device_data is not None else
                         (torch.cuda.is_available() and not os.environ.get("SSDN_HOST_DATA"))) and \
            cfg[ConfigValue.TRAIN_PATCH_SIZE] % NoiseNetwork.input_wh_mul() == 0
        source = CleanPatches(dataset) if device_stream else dataset
        if device_stream:
            kw["collate_fn"] = CleanPatches.collate          # (the workers deliver whole minibatches: CleanPatches.__getitems__)
        if self.world > 1:
            _ = iter(sampler)                        # materialise the order under the common seed, then reuse it
            sampler.for_next_iter(sampler.last_iter())
            torch.set_rng_state(g)
            self._shard = _RankShard(sampler, cfg[ConfigValue.TRAIN_MINIBATCH_SIZE], self.rank, self.world)
            loader = DataLoader(source, batch_sampler=self._shard, **kw)
        else:
            self._shard = None
            loader = DataLoader(source, sampler=sampler, batch_size=cfg[ConfigValue.TRAIN_MINIBATCH_SIZE], **kw)
        if device_stream:
            dev = self.denoiser.device if self.denoiser is not None and hasattr(self.denoiser, "device") else \
                torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
            loader = DevicePatchStream(loader, dataset, dev, rank=self.rank)
            if getattr(self, "_stream_state", None):
                loader.load_state_dict(self._stream_state, rank=self.rank)
                self._stream_state = None
            if self.denoiser is not None and self.denoiser.device.type == "cuda":
                loader.attach(self.denoiser)
        return loader, dataset, sampler

    def set_train_data(self, path: str):
        self.cfg[ConfigValue.TRAIN_DATA_PATH] = path
        self.cfg[ConfigValue.TRAIN_DATASET_TYPE] = self.cfg[ConfigValue.TRAIN_DATASET_NAME] = None
        ssdn.cfg.infer_datasets(self.cfg)

    def test_data(self) -> Tuple[DataLoader, NoisyDataset, FixedLengthSampler]:
        cfg = self.cfg
        dataset = NoisyDataset(self._open(cfg[ConfigValue.TEST_DATA_PATH], cfg[ConfigValue.TEST_DATASET_TYPE], None),
                               cfg[ConfigValue.NOISE_STYLE], cfg[ConfigValue.ALGORITHM], pad_uniform=True,
                               pad_multiple=NoiseNetwork.input_wh_mul(), square=cfg[ConfigValue.BLINDSPOT], training_mode=False)
        _ = dataset[0]
        name = cfg[ConfigValue.TEST_DATASET_NAME]
        n = ssdn.cfg.test_length(name) if name in ("bsd", "kodak", "set14") else len(dataset)
        sampler = FixedLengthSampler(dataset, num_samples=n, shuffled=False)
        loader = DataLoader(dataset, sampler=sampler, batch_size=cfg[ConfigValue.TEST_MINIBATCH_SIZE],
                            num_workers=cfg[ConfigValue.DATALOADER_WORKERS], pin_memory=cfg[ConfigValue.PIN_DATA_MEMORY])
        return loader, dataset, sampler

    def set_test_data(self, path: str):
        self.cfg[ConfigValue.TEST_DATA_PATH] = path
        self.cfg[ConfigValue.TEST_DATASET_TYPE] = self.cfg[ConfigValue.TEST_DATASET_NAME] = None
        ssdn.cfg.infer_datasets(self.cfg)


def _to_cpu(obj):
    """checkpoints hold CPU tensors (the reference saves from GPU tensors and loads with map_location='cpu')"""
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return type(obj)((k, _to_cpu(v)) for k, v in obj.items()) if not isinstance(obj, defaultdict) else obj
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    if isinstance(obj, Metric) and torch.is_tensor(obj.total):
        obj.total = obj.total.detach().cpu()
    return obj


def resume_run(run_dir: str, iteration: int = None) -> DenoiserTrainer:
    """Newest (or the given) `training/*.training` of a run directory -> a trainer that continues in that directory."""
    run_dir = os.path.abspath(run_dir)
    found = {}
    for path in glob.glob(os.path.join(run_dir, "training", "*.training")):
        m = re.findall(r"\d+", os.path.basename(path))
        if m:
            found[int(m[0])] = path
    if iteration is None:
        if not found:
            raise ValueError("Run directory contains no training files.")
        iteration = max(found)
    path = found[iteration]
    logger.info("Loading from '{}'...".format(path))
    trainer = DenoiserTrainer(None, runs_dir=os.path.abspath(os.path.join(run_dir, "..")), run_dir=os.path.basename(run_dir))
    trainer.load_state_dict(path)
    logger.info("Loaded training state.")
    for t in trainer.state[StateValue.HISTORY][HistoryValue.TIMINGS].values():
        if isinstance(t, TrackedTime):
            t.forget()                  # absolute times of the old process mean nothing now
    return trainer

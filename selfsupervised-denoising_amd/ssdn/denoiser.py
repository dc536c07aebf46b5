"""`Denoiser` -- network(s) + loss head of one ssdn configuration, MI355X edition.

Drop-in for /root/reference/ssdn/ssdn/denoiser.py:23-412: same constructor (`cfg`, `device`), `run_pipeline`, `forward`,
`get_model`, `state_dict` / `from_state_dict` (same key layout incl. the DataParallel `module.` prefix and the `_models`
aliases) and `config_name`.  What differs is everything underneath: the nets' parameters are views
into ONE flat fp32 device buffer [main net | sigma estimator | learnable sigma scalar]; a forward / loss / backward /
optimiser step is four calls into libssdn_hip.so (`ssdn.hip.engine.DenoiserEngine`); nn.DataParallel is gone -- data
parallelism is one process per GPU with an RCCL all-reduce of the flat gradient (`ssdn.hip.dp`).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

import ssdn
from ssdn.datasets import NoisyDataset
from ssdn.models import NoiseNetwork
from ssdn.params import ConfigValue, NoiseValue, Pipeline, PipelineOutput


import os as _os

_CHECK_LOSS_GRAD = _os.environ.get("SSDN_CHECK_LOSS_GRAD", "0") == "1"
_MAX_ENGINES = 4        # cached (batch, size, mode) plans per Denoiser; the least recently used one is dropped beyond this


class _ParallelShim(nn.Module):
    """Occupies the place of nn.DataParallel in the module tree so that checkpoints keep the
    `models.<id>.module.<param>` key layout (denoiser.py:102-110).  It parallelises nothing."""

    def __init__(self, module: nn.Module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


class _LossBridge(torch.autograd.Function):
    """Lets the reference's training idiom `torch.mean(outputs[LOSS]).backward()` drive the HIP backward pass:
    the returned LOSS tensor carries this node; its backward runs the planned backward op list and exposes the flat
    gradient buffer through every parameter's `.grad`."""

    @staticmethod
    def forward(ctx, anchor: Tensor, denoiser: "Denoiser", engine, loss: Tensor):
        ctx.denoiser, ctx.engine = denoiser, engine        # the engine that PRODUCED this loss, not "the last one used"
        return loss.clone()

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        d = ctx.denoiser
        if grad_out.dim() != 2 or grad_out.shape[1] != 1:
            raise NotImplementedError("LOSS is [B, 1]; the fused loss head differentiates mean(LOSS) over the batch (train.py:201)")
        if _CHECK_LOSS_GRAD:      # debug aid (SSDN_CHECK_LOSS_GRAD=1): costs a host-device sync per step
            B = grad_out.shape[0]
            if not torch.allclose(grad_out, torch.full_like(grad_out, 1.0 / B), rtol=1e-5, atol=0):
                raise NotImplementedError("the fused loss head differentiates mean(LOSS) over the batch (train.py:201); "
                                          "other reductions of LOSS are not supported")
        d._backward_engine(ctx.engine)
        return None, None, None, None


class Denoiser(nn.Module):
    MODEL = "denoiser_model"
    SIGMA_ESTIMATOR = "sigma_estimation_model"
    ESTIMATED_SIGMA = "estimated_sigma"

    def __init__(self, cfg: Dict, device: str = None):
        super().__init__()
        self.device = torch.device(device) if device else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.cfg = cfg
        C = cfg[ConfigValue.IMAGE_CHANNELS]
        self._pipeline = cfg[ConfigValue.PIPELINE]
        ssdn_pipe = self._pipeline == Pipeline.SSDN
        if ssdn_pipe and cfg.get(ConfigValue.DIAGONAL_COVARIANCE):
            # the reference's diagonal branch raises (`c00.shape()`, denoiser.py:240) and is unreachable from its CLI
            raise NotImplementedError("DIAGONAL_COVARIANCE is broken in the reference (denoiser.py:240) and not supported")
        cout = C + (C * (C + 1)) // 2 if ssdn_pipe else C               # means + triangular A (denoiser.py:55-64)
        self._var = ssdn_pipe and cfg[ConfigValue.NOISE_VALUE] == NoiseValue.UNKNOWN_VARIABLE
        self._const = ssdn_pipe and cfg[ConfigValue.NOISE_VALUE] == NoiseValue.UNKNOWN_CONSTANT
        blind = cfg[ConfigValue.BLINDSPOT]
        from ssdn.hip.graph import net_layers, net_param_count
        n_main = net_param_count(net_layers(C, cout, blind))
        n_sig = net_param_count(net_layers(C, 1, False)) if self._var else 0
        n_tot = n_main + n_sig + (1 if self._const else 0)
        n_pad = (n_tot + 3) // 4 * 4
        self._n_main, self._n_sig = n_main, n_sig
        self.flat = torch.zeros(n_pad, device=self.device)
        self.flat_grad = torch.zeros(n_pad, device=self.device)
        self.adam_m = torch.zeros(n_pad, device=self.device)
        self.adam_v = torch.zeros(n_pad, device=self.device)
        self.adam_steps = 0

        self.models = nn.ModuleDict()    # "parallelised" handles (reference: nn.DataParallel wrappers)
        self._models = nn.ModuleDict()   # plain handles to the same modules
        self._add(Denoiser.MODEL, NoiseNetwork(C, cout, blindspot=blind, device=self.device,
                                               flat=(self.flat[:n_main], self.flat_grad[:n_main])))
        if self._var:
            self._add(Denoiser.SIGMA_ESTIMATOR, NoiseNetwork(C, 1, blindspot=False, zero_output_weights=True, device=self.device,
                                                            flat=(self.flat[n_main:n_main + n_sig], self.flat_grad[n_main:n_main + n_sig])))
        self.l_params = nn.ParameterDict()
        if self._const:
            self.l_params[Denoiser.ESTIMATED_SIGMA] = nn.Parameter(self.flat[n_main + n_sig:n_main + n_sig + 1].view(1, 1, 1, 1))
        self._engines: Dict[Tuple, list] = {}       # key -> [engine, parameter version its 16-bit weight shadows hold]
        self._version = 0
        self._last_engine = None                     # engine of the last run_pipeline (any mode)
        self._last_train_engine = None               # engine of the last TRAINING-mode run_pipeline (backward / optimiser)
        self._exchange = None                        # ssdn.hip.dp.GradExchange of train_step (data parallel)
        self._anchor = torch.zeros((), requires_grad=True)

    def _add(self, model_id: str, model: nn.Module):
        self._models[model_id] = model
        self.models[model_id] = _ParallelShim(model)

    # ---- reference surface --------------------------------------------------------------------------------------
    def get_model(self, model_id: str, parallelised: bool = True) -> nn.Module:
        return (self.models if parallelised else self._models)[model_id]

    def config_name(self) -> str:
        return ssdn.cfg.config_name(self.cfg)

    def state_dict(self, params_only: bool = False, **kw) -> Dict:
        sd = super().state_dict(**kw)
        if not params_only:
            sd["cfg"] = self.cfg
        return sd

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        sd = {k: v for k, v in state_dict.items() if k != "cfg"}
        r = super().load_state_dict(sd, strict=strict, **kw)
        self.mark_dirty()
        return r

    @staticmethod
    def from_state_dict(state_dict: Dict) -> "Denoiser":
        d = Denoiser(state_dict["cfg"])
        d.load_state_dict(state_dict, strict=False)
        return d

    def mark_dirty(self):
        """The fp32 parameters changed: the fp16/bf16 MFMA shadows are re-packed before the next run.  In-place updates of the
        parameters themselves (a `torch.optim` step, `load_state_dict`, `p.copy_()` under no_grad) are detected through the version
        counter of the flat buffer, which all parameter views share.  Writes through `p.data` (its own version counter) or a raw
        pointer are NOT: call this after them.  Evaluation-mode runs re-pack unconditionally (23 us), so a stale shadow can only
        ever be seen by a training-mode forward that follows an undeclared `.data` write."""
        self._version += 1
        for m in self._models.values():
            if hasattr(m, "mark_dirty"):
                m.mark_dirty()

    def _param_version(self):
        return (self._version, self.flat._version)

    # ---- engines ---------------------------------------------------------------------------------------------------
    def _engine(self, B: int, H: int, W: int, train: bool, ncoords: int = 64):
        from ssdn.hip import lib as L
        from ssdn.hip.engine import DenoiserEngine
        if self.device.type != "cuda":
            raise L.SsdnHipError("Denoiser.run_pipeline needs an MI355X: device is %s and the ssdn hot path has no CPU fallback" % self.device)
        key = (B, H, W, train, ncoords)
        if key not in self._engines:
            cfg = self.cfg
            eng = DenoiserEngine(self._pipeline.value, cfg[ConfigValue.IMAGE_CHANNELS], cfg[ConfigValue.BLINDSPOT],
                                 cfg.get(ConfigValue.NOISE_STYLE) or "gauss", cfg[ConfigValue.NOISE_VALUE].value if self._pipeline == Pipeline.SSDN else "known",
                                 B, H, W, self.device, self.flat, self.flat_grad, self.adam_m, self.adam_v,
                                 self._n_main, self._n_sig, self._const, train=train, ncoords=ncoords)
            self._engines[key] = [eng, None]
            while len(self._engines) > _MAX_ENGINES:         # LRU: a plan owns ~1 GB of buffers at BASELINE sizes
                old = next(k for k in self._engines if k != key)
                dead = self._engines.pop(old)[0]
                if dead is self._last_engine:
                    self._last_engine = None
                if dead is self._last_train_engine:
                    self._last_train_engine = None
        slot = self._engines.pop(key)
        self._engines[key] = slot                            # most recently used last
        if slot[1] != self._param_version() or not train:
            slot[0].repack()
            slot[1] = self._param_version()
        return slot[0]

    # ---- pipelines -------------------------------------------------------------------------------------------------
    def forward(self, data: Tensor) -> Tensor:
        """Inference (denoiser.py:112-126): denoise a BCHW batch with the configured pipeline."""
        return self.run_pipeline([data])[PipelineOutput.IMG_DENOISED]

    def run_pipeline(self, data: List, **kwargs) -> Dict:
        """Reference surface (denoiser.py:128-138).  Outputs are fresh tensors, as in the reference: the caller may keep
        them across calls.  (`train_step` uses the engine's buffers directly.)"""
        return self._run(data, clone=True)

    def input_buffer(self, B: int, H: int, W: int, ncoords: int = 64) -> Optional[Tensor]:
        """The fp32 [B, C, H, W] input buffer of the TRAINING engine for this shape.  A producer that writes the noisy minibatch
        straight into it (ssdn.datasets.DevicePatchStream does, when attached) and passes that very tensor as the pipeline
        input saves `run_pipeline` its device-to-device copy; the buffer is overwritten by the next minibatch.
        Never BUILDS an engine (a plan owns ~1 GB) and never touches the LRU order or the weight shadows: None until the first
        training step of this shape has created it (the producer then uses a buffer of its own, once)."""
        slot = self._engines.get((B, H, W, True, ncoords))
        return slot[0].inp if slot is not None else None

    def _run(self, data: List, clone: bool, bridge: bool = True) -> Dict:
        if self._pipeline not in (Pipeline.MSE, Pipeline.SSDN, Pipeline.MASK_MSE):
            raise NotImplementedError("Unsupported processing pipeline")
        inp = data[NoisyDataset.INPUT]
        B, C, H, W = inp.shape
        ref = data[NoisyDataset.REFERENCE] if len(data) > NoisyDataset.REFERENCE else None
        meta = data[NoisyDataset.METADATA] if len(data) > NoisyDataset.METADATA else {}
        MD = NoisyDataset.Metadata
        coords = meta.get(MD.MASK_COORDS) if isinstance(meta, dict) else None
        if coords is not None and coords.device.type == "cpu" and coords.numel():
            c0 = coords[0]          # the reference indexes with batch element 0's coordinates and raises IndexError out of range
            if int(c0[:, 0].min()) < -H or int(c0[:, 0].max()) >= H or int(c0[:, 1].min()) < -W or int(c0[:, 1].max()) >= W:
                raise IndexError("mask coordinates out of range for a %dx%d image" % (H, W))
        train = self.training and torch.is_grad_enabled()
        eng = self._engine(B, H, W, train, ncoords=(coords.shape[1] if coords is not None else 64))
        if inp.data_ptr() != eng.inp.data_ptr():                          # (a producer may have written the engine's buffer itself)
            eng.inp.copy_(inp.to(torch.float32), non_blocking=True)      # device boundary (denoiser.py:143,186)
        have_loss = True
        if self._pipeline == Pipeline.SSDN:
            if self.cfg[ConfigValue.NOISE_VALUE] == NoiseValue.KNOWN:
                npv = meta[MD.INPUT_NOISE_VALUES]
                # A producer may mark a parameter tensor it never rewrites (DevicePatchStream's cached constant of a fixed-sigma style:
                # `_ssdn_const`); that very OBJECT (held by reference, so its address cannot be handed out again) is uploaded once.
                # Everything else -- e.g. the per-minibatch tensor of a ranged style, filled through a raw pointer, whose
                # (address, _version) can repeat with different contents -- is copied every step (B floats, device to device).
                if not (getattr(npv, "_ssdn_const", False) and npv is getattr(eng, "_np_src", None)):
                    eng.noise_param.copy_(npv.reshape(B).to(torch.float32), non_blocking=True)
                    eng._np_src = npv if getattr(npv, "_ssdn_const", False) else None
        else:
            have_loss = ref is not None and (self._pipeline == Pipeline.MSE or coords is not None)
            if have_loss:
                eng.ref.copy_(ref.to(torch.float32), non_blocking=True)
                if self._pipeline == Pipeline.MASK_MSE:
                    c0 = coords[0].to(torch.int64)
                    c0 = torch.stack([c0[:, 0] % H, c0[:, 1] % W], 1)              # python-style negative indices wrap
                    eng.coords.copy_(c0, non_blocking=True)   # element 0's mask for everyone (n2v_loss.py:12)
        if have_loss:
            eng.forward()
        else:
            eng.net_forward_only()
        self._last_engine = eng
        self._has_loss_last = have_loss
        if train:
            self._last_train_engine = eng
        own = (lambda t: t.clone()) if clone else (lambda t: t)
        out = {PipelineOutput.INPUTS: data}
        net_out = eng.main.tensor("out32")
        if self._pipeline == Pipeline.SSDN:
            out[PipelineOutput.IMG_MU] = own(eng.mu)
            out[PipelineOutput.IMG_DENOISED] = own(eng.pme)
            gauss = eng.style == "gauss"
            nstd = eng.noise_std
            if gauss:
                nstd = nstd[:1].view(1, 1, 1) if self._const else nstd.view(B, 1, 1)
            out[PipelineOutput.NOISE_STD_DEV] = own(nstd)
            out[PipelineOutput.MODEL_STD_DEV] = own(eng.model_std)
        else:
            out[PipelineOutput.IMG_DENOISED] = own(net_out)
        if have_loss:
            loss = eng.loss
            # (_LossBridge.forward clones; the eval branch clones here)
            # (train_step drives the backward list itself: no autograd bridge, no copy of the loss)
            out[PipelineOutput.LOSS] = (_LossBridge.apply(self._anchor, self, eng, loss) if bridge else loss) if train else own(loss)
        return out

    def backward(self):
        """Run the planned backward pass of the last training-mode run_pipeline; gradients land in `flat_grad` and are
        visible as `.grad` of every parameter."""
        eng = self._last_train_engine
        if eng is None or not eng.train:
            raise RuntimeError("backward() needs a preceding training-mode run_pipeline()")
        self._backward_engine(eng)

    def _backward_engine(self, eng):
        if eng is None or not eng.train:
            raise RuntimeError("this LOSS was not produced by a training-mode run_pipeline()")
        eng.backward()
        self._expose_grads()

    def _expose_grads(self):
        for net, base in ((self._models[Denoiser.MODEL], 0),) + (((self._models[Denoiser.SIGMA_ESTIMATOR], self._n_main),) if self._var else ()):
            for l in net.layers:
                h = net.get_submodule(l.name)
                h.weight.grad = self.flat_grad[base + l.w_off: base + l.w_off + l.M * l.cin * l.k * l.k].view(l.M, l.cin, l.k, l.k)
                h.bias.grad = self.flat_grad[base + l.b_off: base + l.b_off + l.M]
        if self._const:
            o = self._n_main + self._n_sig
            self.l_params[Denoiser.ESTIMATED_SIGMA].grad = self.flat_grad[o:o + 1].view(1, 1, 1, 1)

    def optimizer_step(self, lr: float, grad_scale: float = 1.0):
        """Fused Adam (betas 0.9/0.99, eps 1e-8; train.py:100-107) over the flat buffer + re-pack of the fp16 MFMA shadows.
        Acts on the engine of the last TRAINING-mode run_pipeline (an eval / snapshot call in between does not matter)."""
        eng = self._last_train_engine
        if eng is None:
            raise RuntimeError("optimizer_step() needs a preceding training-mode run_pipeline()")
        self.adam_steps += 1
        eng.adam(lr, self.adam_steps, grad_scale)
        # the kernel wrote the flat buffer through a raw pointer: bump our own counter (torch's did not move); the shadows of
        # THIS engine are fresh, other cached shapes and the nets' own forward engines re-pack lazily
        self.mark_dirty()
        for slot in self._engines.values():
            if slot[0] is eng:
                slot[1] = self._param_version()

    def gradient_exchange(self, world: int):
        """The bucketed, backward-overlapped all-reduce of this model's flat gradient (ssdn.hip.dp.GradExchange)."""
        from ssdn.hip import dp
        net = self._models[Denoiser.MODEL]
        return dp.GradExchange(world, dp.bucket_ranges(net.layers, self._n_main, self.flat.numel()), self.device)

    # ---- H11: per-step metrics on the device ---------------------------------------------------------------------------
    METRIC_NAMES = ("loss", "psnr_out", "psnr_mu_out", PipelineOutput.NOISE_STD_DEV.value, PipelineOutput.MODEL_STD_DEV.value)
    assert len(METRIC_NAMES) <= 7       # the accumulator is 16 floats: (sum, count) per metric in [0, 14), the kernel's arrival counter in [15]

    def _metrics_acc(self, kind: str) -> Tensor:
        accs = self.__dict__.setdefault("_macc", {})
        if kind not in accs:
            accs[kind] = torch.zeros(16, dtype=torch.float32, device=self.device)
        return accs[kind]

    def accumulate_metrics(self, data: List, kind: str = "train", with_loss: bool = True, per_sample: bool = False):
        """Add the metrics of the LAST `run_pipeline` / `train_step` batch (`data` = its inputs) to the device-resident accumulator
        `kind` with one kernel launch (SSDN_OP_METRICS: loss, PSNR of IMG_DENOISED and IMG_MU against the clean image over each
        sample's un-padded extent, noise / model std-dev x 255 -- what the reference trainer accumulates with ~20 ATen launches per
        step, train.py:205-218, utils/data.py:94-105).  Nothing is copied to the host; `read_metrics` does that when the trainer
        prints.  per_sample=True returns {"psnr_out": [B], "psnr_mu_out": [B]} of this batch (host tensors: synchronises)."""
        eng = self._last_engine
        if eng is None:
            raise RuntimeError("accumulate_metrics() needs a preceding run_pipeline()")
        meta = data[NoisyDataset.METADATA]
        MD = NoisyDataset.Metadata
        clean = meta[MD.CLEAN]
        B = clean.shape[0]
        if clean.device != self.device or clean.dtype != torch.float32 or not clean.is_contiguous():
            clean = clean.to(self.device, torch.float32, non_blocking=True).contiguous()
        ext = None
        shp = meta.get(MD.IMAGE_SHAPE)
        if shp is not None:
            shp = torch.as_tensor(shp).reshape(B, -1)[:, -2:].to(torch.int32)
            if bool((shp != torch.tensor(list(clean.shape[-2:]), dtype=torch.int32)).any()):
                ext = shp.to(self.device, non_blocking=True).contiguous()
        eng.accumulate_metrics(self._metrics_acc(kind), clean, ext, with_loss=with_loss and self._has_loss_last)
        self._metrics_keep = (clean, ext)               # alive until the launch has run
        if per_sample:
            per = eng.metrics_per.cpu()                      # (synchronises: the launch has run)
            if kind != "train":
                self._metrics_acc(kind).zero_()              # only the per-sample values of such a batch are used: its sums must not pile up
            res = {"psnr_out": per[:, 1].clone()}
            if self._pipeline == Pipeline.SSDN:
                res["psnr_mu_out"] = per[:, 2].clone()
            return res
        return None

    def reset_device_metrics(self, kind: str = "train"):
        """forget what `accumulate_metrics` has summed on the device since the last read (the trainer's reset_metrics)"""
        if kind in self.__dict__.get("_macc", {}):
            self._macc[kind].zero_()

    def read_metrics(self, kind: str = "train", reset: bool = True) -> Dict[str, Tuple[float, int]]:
        """{metric name: (sum over samples, sample count)} accumulated since the last reset -- ONE 64-byte copy to the host."""
        acc = self._metrics_acc(kind)
        v = acc.cpu().tolist()
        if reset:
            acc.zero_()
        return {name: (v[2 * k], int(round(v[2 * k + 1]))) for k, name in enumerate(self.METRIC_NAMES) if v[2 * k + 1] > 0}

    def train_step(self, data: List, lr: float, exchange=None, metrics: bool = False) -> Dict:
        """One whole optimisation step on this GPU: forward + loss + backward + Adam.  exchange: `gradient_exchange(world)`
        for data parallelism -- the per-bucket all-reduces are issued behind events recorded inside the backward list, so
        they overlap the rest of the backward pass; Adam waits for them and folds in 1 / world."""
        from ssdn.hip import dp
        out = self._run(data, clone=False, bridge=False)
        eng = self._last_train_engine
        if metrics:                         # (reads the forward pass's outputs: enqueued in front of the backward pass)
            self.accumulate_metrics(data, "train")
        from ssdn.hip import engine as _engine
        scale = dp.exchange_step(lambda ex: eng.backward(exchange=ex, defer_tail=ex is None and _engine.DEFER_TAIL), self.flat_grad, exchange)
        self.optimizer_step(lr, scale)
        return out

    # ---- optimiser state in the reference's torch.optim.Adam layout (train.py:725,744: `.training` checkpoints) ----------
    def _param_slices(self):
        """(flat offset, numel, shape) of every parameter in `self.parameters()` order == the reference's order (the module
        tree mirrors it, incl. output_conv registered before output_block and the de-duplicated `_models` aliases)."""
        base = self.flat.data_ptr()
        res = []
        for p in self.parameters():
            off = (p.data_ptr() - base) // 4
            res.append((off, p.numel(), tuple(p.shape)))
        return res

    def optimizer_state_dict(self, lr: float = 3e-4) -> Dict:
        """State of the fused Adam as `torch.optim.Adam(denoiser.parameters(), betas=[0.9, 0.99]).state_dict()` would have it."""
        state = {}
        sl = self._param_slices()
        if self.adam_steps > 0:
            for i, (off, n, shape) in enumerate(sl):
                state[i] = {"step": torch.tensor(float(self.adam_steps)),
                            "exp_avg": self.adam_m[off:off + n].view(shape).clone(),
                            "exp_avg_sq": self.adam_v[off:off + n].view(shape).clone()}
        group = {"lr": lr, "betas": (0.9, 0.99), "eps": 1e-8, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(sl)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd: Dict):
        sl = self._param_slices()
        self.adam_m.zero_()
        self.adam_v.zero_()
        steps = 0
        for i, (off, n, shape) in enumerate(sl):
            st = sd.get("state", {}).get(i)
            if st is None:
                continue
            self.adam_m[off:off + n].copy_(st["exp_avg"].reshape(-1).to(self.adam_m))
            self.adam_v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(self.adam_v))
            steps = max(steps, int(float(st["step"])))
        self.adam_steps = steps

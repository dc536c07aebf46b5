"""N2: the training patch stream with the per-sample work on the DEVICE.

The reference prepares every training sample on the host (ssdn/ssdn/datasets/noise_wrapper.py:50-92: noise synthesis,
Noise2Void manipulation, metadata) and ships fp32 noisy + clean patches through the DataLoader: 2 x 49 KB per 64x64 RGB patch over
PCIe, and at > 10 k patches/s the host cores -- not the GPU -- set the pace.  Here the DataLoader workers only read, crop and
return the CLEAN patch as uint8 (12 KB); one pinned upload per minibatch; noise, the Noise2Void manipulation and the
per-sample noise parameters are produced for the whole batch by ONE kernel of libssdn_hip.so (SSDN_OP_NOISE: in-kernel
Philox4x32-10, the reference's distributions including its quirks -- rate-1 Poisson, ranged parameters per sample AND channel,
the [0, x + r) Noise2Void window).  The random STREAM is the kernel's, so host and device pipelines draw different samples of the
same distributions (statistical parity: tests/test_hip_noise.py against the reference-generated moments).  With CPU tensors (no
GPU: tests) the same preparation runs through the two functions of the host path.

Yields `[inp, ref, metadata]` exactly as a DataLoader over `NoisyDataset` does (same keys, shapes and dtypes, batch-stacked),
with device tensors.  Training patches only (all samples of a batch have the same, already valid, size)."""
from __future__ import annotations

from typing import Dict, Iterator, Optional

import torch
from torch.utils.data import Dataset

from ssdn.params import NoiseAlgorithm
from ssdn.datasets.noise_wrapper import NULL_IMAGE, NoisyDataset


class CleanPatches(Dataset):
    """(uint8 CHW clean patch, index) of a NoisyDataset's child -- what a DataLoader worker produces for the device stream.
    The child's images come from 8-bit files, so float -> uint8 is exact."""

    def __init__(self, noisy: NoisyDataset):
        self.noisy = noisy

    def __len__(self) -> int:
        return len(self.noisy)

    def __getitem__(self, index: int):
        img = self.noisy.child[index][0]
        return (img * 255.0).round().clamp_(0, 255).to(torch.uint8), index

    def __getitems__(self, indexes):
        """One worker call per MINIBATCH (torch's fetcher uses it when present).  An HDF5 child under the training crop reads only
        the crop windows (HDF5Dataset.patches_u8: tens of thousands of patches/s per worker instead of 450 through PIL + float).
        Returns the per-sample list any collate function expects -- the samples are views of ONE uint8 [B, C, P, P] array, which
        `CleanPatches.collate` hands over as it is."""
        from ssdn.datasets.transforms import RandomCrop
        child = self.noisy.child
        tf = getattr(child, "transform", None)
        if hasattr(child, "patches_u8") and isinstance(tf, RandomCrop) and tf.pad_if_needed and tf.padding_mode == "reflect" and \
                getattr(child, "output_format", None) is not None:
            base = torch.from_numpy(child.patches_u8(list(indexes), tf.size))
            out = _PreBatched((base[k], int(i)) for k, i in enumerate(indexes))
            out.base, out.indexes = base, torch.as_tensor(list(indexes), dtype=torch.int64)
            return out
        return [self[i] for i in indexes]

    @staticmethod
    def collate(batch):
        """collate_fn of the training DataLoader: a minibatch `__getitems__` built in one piece is passed through, anything else
        is stacked the default way"""
        if isinstance(batch, _PreBatched):
            return batch.base, batch.indexes
        from torch.utils.data import default_collate
        return default_collate(batch)


class _PreBatched(list):
    """list of (patch, index) samples that are views of `base` (uint8 [B, C, P, P]); `indexes` int64 [B]"""
    base = None
    indexes = None


class _Uploaded:
    """a minibatch on its way to the device (DevicePatchStream.upload): device tensor, completion event, the pinned source (kept
    alive until the copy is consumed)"""

    def __init__(self, dev: torch.Tensor, event, host: torch.Tensor):
        self.dev, self.event, self.host = dev, event, host


class DevicePatchStream:
    def __init__(self, loader, noisy: NoisyDataset, device, seed: Optional[int] = None, rank: int = 0):
        self.loader, self.noisy, self.device = loader, noisy, torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:         # ("cuda" != "cuda:0" for torch.device comparisons)
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.generator = torch.Generator(device=self.device)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())      # drawn from the (checkpointed) host RNG
        self.rank = rank
        self.seed = seed + rank            # ranks never share a noise stream, whatever their host RNG states are
        self.generator.manual_seed(self.seed)
        self._calls = 0                    # Philox counter of SSDN_OP_NOISE: a fresh offset per minibatch
        self._static: Dict = {}            # metadata tensors that only depend on the batch shape
        self._denoiser = None

    def state_dict(self) -> Dict:
        """what a `.training` file needs to continue this stream: the Philox key (without the rank offset) and the number of minibatches
        prepared so far (the counter part of the kernel's random-number offsets)"""
        return {"seed": int(self.seed - self.rank), "calls": int(self._calls)}

    def load_state_dict(self, sd: Dict, rank: Optional[int] = None):
        self.rank = self.rank if rank is None else rank
        self.seed = int(sd["seed"]) + self.rank
        self._calls = int(sd["calls"])
        self.generator.manual_seed(self.seed)

    def attach(self, denoiser):
        """Write every noisy minibatch straight into `denoiser`'s training input buffer (Denoiser.input_buffer): the pipeline
        then finds its input in place.  The yielded input tensor IS that buffer -- valid until the next minibatch is prepared."""
        self._denoiser = denoiser
        return self

    def __len__(self) -> int:
        return len(self.loader)

    def __iter__(self) -> Iterator:
        # one minibatch ahead: the upload of batch i + 1 runs on a copy stream while batch i is being trained on
        # (a hipMemcpyAsync on the compute stream costs that stream ~0.1 ms of latency per step, measured by bench.py)
        it = iter(self.loader)
        try:
            clean_u8, indexes = next(it)
        except StopIteration:
            return
        pending = (self.upload(clean_u8), indexes)
        for clean_u8, indexes in it:
            cur, pending = pending, (self.upload(clean_u8), indexes)
            yield self.prepare(*cur)
        yield self.prepare(*pending)

    def upload(self, clean_u8: torch.Tensor):
        """Start the host -> device copy of a minibatch of clean uint8 patches on the stream's own copy stream; the returned
        handle goes to `prepare`, which makes the compute stream wait for the copy.  (CPU device: returns the tensor itself.)"""
        if self.device.type != "cuda" or clean_u8.device == self.device:
            return clean_u8
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        if not clean_u8.is_pinned():
            clean_u8 = clean_u8.pin_memory()
        with torch.cuda.stream(self._copy_stream):
            dev = clean_u8.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        return _Uploaded(dev, ev, clean_u8)

    def prepare(self, clean_u8: torch.Tensor, indexes: torch.Tensor):
        """the batch form of NoisyDataset.prepare_input (noise_wrapper.py:56-88 here; reference noise_wrapper.py:66-135).
        On a GPU: ONE upload + ONE kernel (SSDN_OP_NOISE of libssdn_hip.so); on CPU tensors: the same arithmetic through the two
        functions of the host path (`ssdn.utils.noise.add_style`, `ssdn.utils.n2v_ups.manipulate_batch`)."""
        if self.device.type == "cuda":
            return self._prepare_hip(clean_u8, indexes)
        return self._prepare_torch(clean_u8, indexes)

    # ---- device path -------------------------------------------------------------------------------------------------------
    def _prepare_hip(self, clean_u8: torch.Tensor, indexes: torch.Tensor):
        import ctypes as C
        from ssdn.hip import lib as L
        from ssdn.utils import n2v_ups, noise
        ds, MD, dev = self.noisy, NoisyDataset.Metadata, self.device
        if isinstance(clean_u8, _Uploaded):                     # started by `upload`: wait for the copy, keep the memory alive
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(clean_u8.event)
            clean_u8.dev.record_stream(cur)
            clean_u8 = clean_u8.dev
        if clean_u8.dtype != torch.uint8:
            raise TypeError("device patch stream: clean patches must arrive as uint8")
        if clean_u8.device != dev:
            if not clean_u8.is_pinned():
                clean_u8 = clean_u8.pin_memory()
            clean_u8 = clean_u8.to(dev, non_blocking=True)
        clean_u8 = clean_u8.contiguous()
        B, Cn, H, W = clean_u8.shape
        key = (B, Cn, H, W)
        st = self._static.get(key)
        if st is None:
            if tuple(ds.get_output_size(torch.empty((Cn, H, W))).tolist()) != (Cn, H, W):
                raise ValueError("device patch stream: patches must already have a valid network input size")
            st = {"shape": torch.tensor([Cn, H, W]).repeat(B, 1), "null": torch.stack([NULL_IMAGE] * B).to(dev)}
            self._static = {key: st}
        kind, params, clip = noise.parse_style(ds.noise_style)
        if kind not in ("gauss", "poisson"):
            raise NotImplementedError("Noise type not supported")
        vals = [(p / 255.0 if (kind == "gauss" and isinstance(p, int)) else float(p)) for p in params]
        lo, hi = (vals[0], vals[0]) if len(vals) == 1 else (vals[0], vals[1])
        ranged = len(vals) > 1
        algo = ds.algorithm
        n2v = algo == NoiseAlgorithm.NOISE_TO_VOID and ds.training_mode
        want_ref = algo in (NoiseAlgorithm.NOISE_TO_NOISE, NoiseAlgorithm.NOISE_TO_VOID)
        f32 = dict(dtype=torch.float32, device=dev)
        clean = torch.empty((B, Cn, H, W), **f32)
        inp = None
        if self._denoiser is not None and getattr(self._denoiser, "training", False):
            box_ = n2v_ups._box_size() if n2v else 0
            buf = self._denoiser.input_buffer(B, H, W, ncoords=((W // box_) * (H // box_) if n2v else 64))
            if buf is not None and tuple(buf.shape) == (B, Cn, H, W) and buf.device == dev:
                inp = buf
        if inp is None:
            inp = torch.empty((B, Cn, H, W), **f32)
        ref = torch.empty((B, Cn, H, W), **f32) if want_ref else None
        par = torch.empty((B, Cn, 1, 1), **f32) if ranged else None
        par_ref = torch.empty((B, Cn, 1, 1), **f32) if (ranged and want_ref) else None
        box = n2v_ups._box_size() if n2v else 0
        coords = torch.empty((B, (W // box) * (H // box), 2), dtype=torch.int64, device=dev) if n2v else None
        a = L.NoiseArgs()
        a.clean_u8, a.clean32, a.noisy32 = clean_u8.data_ptr(), clean.data_ptr(), inp.data_ptr()
        a.ref32 = ref.data_ptr() if ref is not None else None
        a.param = par.data_ptr() if par is not None else None
        a.param_ref = par_ref.data_ptr() if par_ref is not None else None
        a.coords = coords.data_ptr() if coords is not None else None
        a.B, a.C, a.H, a.W = B, Cn, H, W
        a.style, a.clip, a.p_lo, a.p_hi = (0 if kind == "gauss" else 1), int(clip), lo, hi
        a.n2v_box, a.n2v_radius = box, 5 // 2
        a.seed, a.offset = self.seed, self._calls
        self._calls += 1
        rec = (L.OpRec * 1)()
        rec[0].type, rec[0].lane, rec[0].args = L.OP["noise"], 0, C.cast(C.pointer(a), C.c_void_p)
        L.check(L.load().ssdn_run_ops(rec, 1, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))

        def coeff(t, fixed):
            if t is not None:
                return t
            k = ("coeff", fixed)
            if k not in st:
                st[k] = torch.full((B, 1, 1, 1), float(fixed), **f32)
                st[k]._ssdn_const = True        # never rewritten: Denoiser uploads this very object once (denoiser.py, KNOWN sigma)
            return st[k]
        metadata: Dict = {}
        if n2v:
            metadata[MD.MASK_COORDS] = coords
        if algo == NoiseAlgorithm.NOISE_TO_CLEAN:
            ref_t, ref_c = clean, coeff(None, 0)
        elif want_ref:
            ref_t, ref_c = ref, coeff(par_ref, lo)
        elif algo == NoiseAlgorithm.SELFSUPERVISED_DENOISING:
            ref_t, ref_c = st["null"], coeff(None, 0)
        elif algo == NoiseAlgorithm.SELFSUPERVISED_DENOISING_MEAN_ONLY:
            ref_t, ref_c = inp, coeff(par, lo)
        else:
            raise NotImplementedError("Denoising algorithm not supported")
        metadata[MD.INDEXES] = torch.as_tensor(indexes)
        metadata[MD.CLEAN] = clean
        metadata[MD.IMAGE_SHAPE] = st["shape"]
        metadata[MD.INPUT_NOISE_VALUES] = coeff(par, lo)
        metadata[MD.REFERENCE_NOISE_VALUES] = ref_c
        return [inp, ref_t, metadata]

    # ---- host path (CPU tensors) -------------------------------------------------------------------------------------------
    def _prepare_torch(self, clean_u8: torch.Tensor, indexes: torch.Tensor):
        from ssdn.utils import n2v_ups, noise
        ds, MD, g = self.noisy, NoisyDataset.Metadata, self.generator
        if clean_u8.device != self.device:
            clean_u8 = clean_u8.to(self.device)
        clean = clean_u8.to(torch.float32).div_(255.0)
        B = clean.shape[0]
        if tuple(ds.get_output_size(clean[0]).tolist()) != tuple(clean.shape[1:]):
            raise ValueError("device patch stream: patches must already have a valid network input size")
        C, H, W = clean.shape[1:]

        def styled(x):
            # the reference draws a ranged noise parameter per leading index of a CHW sample, i.e. per CHANNEL (noise.py:
            # `_range_param`; reference utils/noise.py:34-39): fold the batch into that axis to get one draw per (sample, channel)
            y, c = noise.add_style(x.reshape(B * C, H, W), ds.noise_style, generator=g)
            return y.reshape(B, C, H, W), (c.reshape(B, C, 1, 1) if torch.is_tensor(c) else c)
        inp, inp_coeff = styled(clean)
        metadata: Dict = {}
        if ds.algorithm == NoiseAlgorithm.NOISE_TO_VOID and ds.training_mode:
            inp, coords = n2v_ups.manipulate_batch(inp, 5, generator=g)
            metadata[MD.MASK_COORDS] = coords
        if ds.algorithm == NoiseAlgorithm.NOISE_TO_CLEAN:
            ref, ref_coeff = clean, 0
        elif ds.algorithm in (NoiseAlgorithm.NOISE_TO_NOISE, NoiseAlgorithm.NOISE_TO_VOID):
            ref, ref_coeff = styled(clean)
        elif ds.algorithm == NoiseAlgorithm.SELFSUPERVISED_DENOISING:
            ref, ref_coeff = NULL_IMAGE, 0
        elif ds.algorithm == NoiseAlgorithm.SELFSUPERVISED_DENOISING_MEAN_ONLY:
            ref, ref_coeff = inp, inp_coeff
        else:
            raise NotImplementedError("Denoising algorithm not supported")
        if ref is NULL_IMAGE:          # what default_collate makes of B copies of the null image
            ref = torch.stack([NULL_IMAGE] * B).to(self.device)

        def coeff(c):
            if torch.is_tensor(c):
                return c.to(torch.float32)
            return torch.full((B, 1, 1, 1), float(c), device=self.device)
        metadata[MD.INDEXES] = torch.as_tensor(indexes)
        metadata[MD.CLEAN] = clean
        metadata[MD.IMAGE_SHAPE] = torch.tensor(list(clean.shape[1:])).repeat(B, 1)
        metadata[MD.INPUT_NOISE_VALUES] = coeff(inp_coeff)
        metadata[MD.REFERENCE_NOISE_VALUES] = coeff(ref_coeff)
        return [inp, ref, metadata]

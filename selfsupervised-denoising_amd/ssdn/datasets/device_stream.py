"""N2: the training patch stream with the per-sample work on the DEVICE.

The reference prepares every training sample on the host (ssdn/ssdn/datasets/noise_wrapper.py:50-92: noise synthesis,
Noise2Void manipulation, metadata) and ships fp32 noisy + clean patches through the DataLoader: 2 x 49 KB per 64x64 RGB patch over
PCIe, and at > 10 k patches/s the host cores -- not the GPU -- set the pace.  Here the DataLoader workers only read, crop and
return the CLEAN patch as uint8 (12 KB); one pinned upload per minibatch; noise (`ssdn.utils.noise.add_style`), the Noise2Void
manipulation (`ssdn.utils.n2v_ups.manipulate_batch`) and the metadata tensors are produced for the whole batch on the device
with torch's device RNG (device memory / RNG plumbing; the arithmetic is a handful of elementwise ops).  The distributions are
the reference's -- including its quirks (rate-1 Poisson, per-sample range parameters, N2V window) because the same two functions
implement both paths -- but the random STREAM is the device generator's, so host and device pipelines draw different samples.

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


class DevicePatchStream:
    def __init__(self, loader, noisy: NoisyDataset, device, seed: Optional[int] = None, rank: int = 0):
        self.loader, self.noisy, self.device = loader, noisy, torch.device(device)
        self.generator = torch.Generator(device=self.device)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())      # drawn from the (checkpointed) host RNG
        self.seed = seed + rank            # ranks never share a noise stream, whatever their host RNG states are
        self.generator.manual_seed(self.seed)

    def __len__(self) -> int:
        return len(self.loader)

    def __iter__(self) -> Iterator:
        for clean_u8, indexes in self.loader:
            yield self.prepare(clean_u8, indexes)

    def prepare(self, clean_u8: torch.Tensor, indexes: torch.Tensor):
        """the batch form of NoisyDataset.prepare_input (noise_wrapper.py:56-88 here; reference noise_wrapper.py:66-135)"""
        from ssdn.utils import n2v_ups, noise
        ds, MD, g = self.noisy, NoisyDataset.Metadata, self.generator
        if clean_u8.device != self.device:
            if self.device.type == "cuda" and not clean_u8.is_pinned():
                clean_u8 = clean_u8.pin_memory()
            clean_u8 = clean_u8.to(self.device, non_blocking=True)
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

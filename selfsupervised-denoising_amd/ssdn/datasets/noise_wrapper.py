"""`NoisyDataset`: wraps a clean-image dataset and hands out (noisy input, reference, metadata) for the algorithm being
trained (drop-in for /root/reference/ssdn/ssdn/datasets/noise_wrapper.py:17-280).  Batch layout contract with
`Denoiser.run_pipeline`: data[INPUT], data[REFERENCE], data[METADATA] (a dict keyed by `NoisyDataset.Metadata`).

  algorithm      input              reference
  n2c            noisy              clean
  n2n, n2v       noisy (+UPS, n2v)  a second, independent noisy draw
  ssdn           noisy              empty tensor (nothing)
  ssdn_u_only    noisy              the same noisy image
Padding (evaluation): reflect-pad bottom/right up to a multiple of `pad_multiple`, to a square when `square`, to the largest
image of the set when `pad_uniform` (Kodak -> 768x768, BSD300 -> 512x512); `unpad` undoes it with IMAGE_SHAPE.
"""
from enum import Enum
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch
from torch import Tensor
from torch.utils.data import Dataset

from ssdn.params import NoiseAlgorithm

NULL_IMAGE = torch.zeros(0)


class NoisyDataset(Dataset):
    INPUT = 0
    REFERENCE = 1
    METADATA = 2

    Metadata = Enum("Metadata", [(n, i + 1) for i, n in enumerate(
        "CLEAN IMAGE_SHAPE INDEXES INPUT_NOISE_VALUES REFERENCE_NOISE_VALUES MASK_COORDS".split())],
        module=__name__, qualname="NoisyDataset.Metadata")

    def __init__(self, child: Dataset, noise_style: str, algorithm: NoiseAlgorithm, enable_metadata: bool = True,
                 pad_uniform: bool = False, pad_multiple: Optional[int] = None, square: bool = False,
                 data_format: str = "CHW", training_mode: bool = False):
        if data_format not in ("CHW", "CWH", "BCHW", "BCWH"):
            raise NotImplementedError("Padding not supported by data format")
        self.child, self.noise_style, self.algorithm = child, noise_style, algorithm
        self.enable_metadata, self.pad_uniform, self.pad_multiple, self.square = enable_metadata, pad_uniform, pad_multiple, square
        self.data_format, self.training_mode = data_format, training_mode
        self._max_image_size = None
        if pad_uniform:
            _ = self.max_image_size

    def __len__(self) -> int:
        return len(self.child)

    def __getitem__(self, index: int):
        img = self.child[index][0]
        metadata = {NoisyDataset.Metadata.INDEXES: index} if self.enable_metadata else None
        inp, ref, metadata = self.prepare_input(img, metadata)
        return (inp, ref, metadata) if self.enable_metadata else (inp, ref)

    def prepare_input(self, clean: Tensor, metadata: Optional[Dict] = None) -> Tuple[Tensor, Tensor, Dict]:
        from ssdn.utils import n2v_ups, noise
        MD = NoisyDataset.Metadata
        if metadata is None and self.enable_metadata:
            metadata = {}
        image_shape = clean.shape
        inp, inp_coeff = noise.add_style(clean, self.noise_style)
        if self.algorithm == NoiseAlgorithm.NOISE_TO_VOID and self.training_mode:
            inp, mask_coords = n2v_ups.manipulate(inp, 5)
            if metadata is not None:
                metadata[MD.MASK_COORDS] = mask_coords
        if self.algorithm == NoiseAlgorithm.NOISE_TO_CLEAN:
            ref, ref_coeff = clean, 0
        elif self.algorithm in (NoiseAlgorithm.NOISE_TO_NOISE, NoiseAlgorithm.NOISE_TO_VOID):
            ref, ref_coeff = noise.add_style(clean, self.noise_style)
        elif self.algorithm == NoiseAlgorithm.SELFSUPERVISED_DENOISING:
            ref, ref_coeff = NULL_IMAGE, 0
        elif self.algorithm == NoiseAlgorithm.SELFSUPERVISED_DENOISING_MEAN_ONLY:
            ref, ref_coeff = inp, inp_coeff
        else:
            raise NotImplementedError("Denoising algorithm not supported")
        # (noise first, padding second: reflected noise is not structured noise across the whole padded image)
        inp = self.pad_to_output_size(inp)
        if ref is not NULL_IMAGE:
            ref = self.pad_to_output_size(ref)
        if metadata is not None:
            one = torch.zeros((1, 1, 1))
            metadata[MD.CLEAN] = self.pad_to_output_size(clean)
            metadata[MD.IMAGE_SHAPE] = torch.tensor(image_shape)
            metadata[MD.INPUT_NOISE_VALUES] = one + inp_coeff
            metadata[MD.REFERENCE_NOISE_VALUES] = one + ref_coeff
        return inp, ref, metadata

    # ---- padding ---------------------------------------------------------------------------------------------------------
    def _hw_axes(self) -> Tuple[int, int]:
        f = self.data_format
        return f.index("H"), f.index("W")

    @property
    def max_image_size(self):
        if self._max_image_size is None:
            try:
                sizes = [self.child.image_size(i) for i in range(len(self.child))]
            except AttributeError:
                sizes = [torch.tensor(d[0].shape) for d in self.child]
            self._max_image_size = torch.stack(sizes).max(dim=0).values
        return self._max_image_size

    def get_output_size(self, image: Tensor) -> Tensor:
        h_ax, w_ax = self._hw_axes()
        size = [int(v) for v in (self.max_image_size if self.pad_uniform else image.shape)]
        if self.pad_multiple:
            m = self.pad_multiple
            for ax in (h_ax, w_ax):
                size[ax] = (size[ax] + m - 1) // m * m
        if self.square:
            size[h_ax] = size[w_ax] = max(size[h_ax], size[w_ax])
        return torch.tensor(size)

    def pad_to_output_size(self, image: Tensor) -> Tensor:
        out = self.get_output_size(image)
        if all(int(o) == int(s) for o, s in zip(out, image.shape)):
            return image
        h_ax, w_ax = self._hw_axes()
        pads = [[0, 0] for _ in image.shape]
        pads[w_ax] = [0, int(out[w_ax]) - image.shape[w_ax]]
        pads[h_ax] = [0, int(out[h_ax]) - image.shape[h_ax]]
        return torch.tensor(np.pad(image.numpy() if image.device.type == "cpu" else image.cpu().numpy(), pads, mode="reflect"),
                            device=image.device)

    @staticmethod
    def _unpad_single(image: Tensor, shape) -> Tensor:
        return image[tuple(slice(0, int(s)) for s in shape)]

    @staticmethod
    def unpad(image: Tensor, metadata: Dict, batch_index: Optional[int] = None) -> Union[Tensor, List[Tensor]]:
        """Cut padded image(s) back to their original extent (stored top-left); a batch returns a list."""
        shape = metadata[NoisyDataset.Metadata.IMAGE_SHAPE]
        if batch_index is not None:
            image, shape = image[batch_index], shape[batch_index]
        if image.dim() <= shape.shape[-1]:
            return NoisyDataset._unpad_single(image, shape)
        return [NoisyDataset._unpad_single(i, s) for i, s in zip(image, shape)]

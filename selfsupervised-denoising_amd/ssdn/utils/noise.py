"""Noise models of the data layer (drop-in for /root/reference/ssdn/ssdn/utils/noise.py:14-153): `add_style(images, style)`
with the style grammar 'gauss{SD}', 'gauss{MIN}_{MAX}', 'poisson{LAMBDA}', 'poisson{MIN}_{MAX}', optional '_nc' (no clip);
integer parameters of gauss styles are /255.  Works on CPU tensors (the reference's per-item dataset path) AND on device
tensors (the batched patch stream, ssdn.datasets.patch_stream: noise is drawn where the batch already lives).

Reference quirks kept: Poisson noise is RATE-1 noise added to lambda*x (utils/noise.py:101-104), not
Poisson(lambda*x); a range style draws one parameter per leading-axis entry of whatever it is called on -- per CHANNEL when
called on an unbatched CHW image (noise.py:55-56), per SAMPLE when called on a batch."""
import re
from numbers import Number
from typing import Tuple, Union

import torch
from torch import Tensor

from ssdn.utils.data import clip_img


def _range_param(tensor: Tensor, lo, hi, generator=None):
    shape = [tensor.shape[0]] + [1] * (tensor.dim() - 1)
    return lo + (hi - lo) * torch.rand(shape, device=tensor.device, generator=generator)


def add_gaussian(tensor: Tensor, std_dev, mean: Number = 0, inplace: bool = False, clip: bool = True, generator=None):
    if not inplace:
        tensor = tensor.clone()
    if isinstance(std_dev, (list, tuple)):
        if len(std_dev) == 1:
            std_dev = std_dev[0]
        else:
            lo, hi = std_dev
            lo = lo / 255 if isinstance(lo, int) else lo
            hi = hi / 255 if isinstance(hi, int) else hi
            std_dev = _range_param(tensor, lo, hi, generator)
    if isinstance(std_dev, int):
        std_dev = std_dev / 255
    tensor.add_(torch.randn(tensor.shape, device=tensor.device, generator=generator) * std_dev + mean)
    if clip:
        tensor = clip_img(tensor, inplace=True)
    return tensor, std_dev


def add_poisson(tensor: Tensor, lam, inplace: bool = False, clip: bool = True, generator=None):
    if not inplace:
        tensor = tensor.clone()
    if isinstance(lam, (list, tuple)):
        lam = lam[0] if len(lam) == 1 else _range_param(tensor, lam[0], lam[1], generator)
    tensor.mul_(lam)
    noise = torch.poisson(torch.ones(tensor.shape, device=tensor.device, dtype=torch.float64), generator=generator)
    tensor.add_(noise.to(tensor.dtype))
    tensor.div_(lam)
    if clip:
        tensor = clip_img(tensor, inplace=True)
    return tensor, lam


def parse_style(style: str):
    """'gauss5_50_nc' -> ('gauss', [5, 50], clip=False)"""
    noise_type = re.findall(r"[a-zA-Z]+", style)[0]
    params = style.replace(noise_type, "").split("_")
    clip = "nc" not in params
    params = [p for p in params if p not in ("nc", "")]
    params = [float(p) for p in params] if any("." in p for p in params) else [int(p) for p in params]
    return noise_type, params, clip


def add_style(images: Tensor, style: str, inplace: bool = False, generator=None) -> Tuple[Tensor, Union[Number, Tensor]]:
    noise_type, params, clip = parse_style(style)
    if noise_type == "gauss":
        return add_gaussian(images, params, inplace=inplace, clip=clip, generator=generator)
    if noise_type == "poisson":
        return add_poisson(images, params, inplace=inplace, clip=clip, generator=generator)
    raise NotImplementedError("Noise type not supported")

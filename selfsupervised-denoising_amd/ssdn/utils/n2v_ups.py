"""Noise2Void uniform pixel selection (drop-in for /root/reference/ssdn/ssdn/utils/n2v_ups.py:7-97): per patch, one blind-spot
coordinate per stratified box (box = round(sqrt(100 / 1.5)) = 8 pixels => 64 coordinates for 64x64), the pixel there is replaced
by a random other pixel of its window.  Reference quirks kept: the window's lower bound is
min(x - r, 0), i.e. [0, x + r] clipped to the image rather than the 5x5 neighbourhood (n2v_ups.py:40-43), and `torch.randint`'s
exclusive upper bound; the removed numpy alias `np.int` (n2v_ups.py:73) is replaced by int.

`manipulate` is the per-image CPU routine of the reference; `manipulate_batch` does the same selection for a whole [B,C,H,W]
batch on whatever device the batch lives (vectorised: rejection sampling replaced by "draw from the window without the centre")."""
import math

import numpy as np
import torch
from torch import Tensor

PERC_PIX = 1.5


def _box_size() -> int:
    return int(np.round(np.sqrt(100 / PERC_PIX)))


def rand_num_exclude(_min: int, _max: int, exclude: list):
    rand = torch.randint(_min, _max, (1,))[0]
    return rand_num_exclude(_min, _max, exclude) if rand in exclude else rand


def get_random_coords(box_size):
    while True:
        yield (torch.rand(1) * box_size, torch.rand(1) * box_size)


def get_stratified_coords(shape):
    box = _box_size()
    gen = get_random_coords(box)
    ys, xs = [], []
    for i in range(int(np.ceil(shape[0] / box))):
        for j in range(int(np.ceil(shape[1] / box))):
            y, x = next(gen)
            y, x = int(i * box + y), int(j * box + x)
            if y < shape[0] and x < shape[1]:
                ys.append(y)
                xs.append(x)
    return ys, xs


def manipulate(image: Tensor, subpatch_size: int = 5, inplace: bool = False):
    if subpatch_size % 2 == 0:
        raise ValueError("subpatch_size must be odd")
    if not inplace:
        image = image.clone()
    image_x, image_y = image.shape[2], image.shape[1]
    r = math.floor(subpatch_size / 2)
    coords = get_stratified_coords((image_x, image_y))
    mask_coords = []
    for x, y in zip(*coords):
        mask_coords.append((x, y))
        min_x, max_x = min(x - r, 0), min(x + r, image_x - 1)
        min_y, max_y = min(y - r, 0), min(y + r, image_y - 1)
        rx = rand_num_exclude(min_x, max_x, [x])
        ry = rand_num_exclude(min_y, max_y, [y])
        image[:, y, x] = image[:, ry, rx]
    return image, torch.tensor(mask_coords)


def manipulate_batch(images: Tensor, subpatch_size: int = 5, generator=None):
    """[B,C,H,W] on any device -> (manipulated copy, coords [B, n, 2] int64 on the same device) with the reference's
    selection rule applied per sample.  Coordinates are (first, second) exactly as `manipulate` returns them."""
    B, C, H, W = images.shape
    dev = images.device
    box, r = _box_size(), subpatch_size // 2
    # `manipulate` calls get_stratified_coords((image_x, image_y)) = (W, H): the FIRST coordinate is stratified over W
    n0, n1 = int(np.ceil(W / box)), int(np.ceil(H / box))
    u = torch.rand((B, n0, n1, 2), device=dev, generator=generator) * box
    c0 = (torch.arange(n0, device=dev).view(1, n0, 1) * box + u[..., 0]).long().reshape(B, -1)
    c1 = (torch.arange(n1, device=dev).view(1, 1, n1) * box + u[..., 1]).long().reshape(B, -1)
    c0.clamp_(max=W - 1)
    c1.clamp_(max=H - 1)           # (for sizes that are multiples of the box nothing is clipped / dropped)
    x, y = c0, c1                  # the reference names them (x, y) and indexes image[:, y, x]
    def pick(c, size):             # uniform over [min(c - r, 0), min(c + r, size - 1)) without c
        lo = torch.clamp(c - r, max=0)
        hi = torch.clamp(c + r, max=size - 1)
        span = (hi - lo - 1).clamp(min=1)                     # candidates in [lo, hi) minus the centre
        k = (torch.rand(c.shape, device=dev, generator=generator) * span).long()
        v = lo + k
        v = torch.where(v >= c, v + 1, v).clamp(max=size - 1)
        return torch.where(v < 0, v + size, v)                # a negative draw indexes from the far edge, like the reference's Python indexing
    rx, ry = pick(x, W), pick(y, H)
    out = images.clone()
    bi = torch.arange(B, device=dev).view(B, 1).expand_as(x)
    out[bi, :, y, x] = images[bi, :, ry, rx]
    return out, torch.stack([x, y], -1)

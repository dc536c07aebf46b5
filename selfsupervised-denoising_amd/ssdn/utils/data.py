"""Tensor helpers used on / next to the hot path (drop-in for the used part of /root/reference/ssdn/ssdn/utils/data.py)."""
import torch
from torch import Tensor


def clip_img(img: Tensor, inplace: bool = False) -> Tensor:
    """Clamp to the valid image range: [0,1] for float images, [0,255] for integer images (utils/data.py:19-39)."""
    if not inplace:
        img = img.clone()
    hi = 1 if img.is_floating_point() else 255
    return img.clamp_(0, hi)


def rotate(x: Tensor, angle: int) -> Tensor:
    """BCHW rotation in multiples of 90 degrees with the reference's NUMERIC convention (utils/data.py:42-67):
    rotate(x, 90)[i, j] = x[j, W-1-i] (counter-clockwise although documented clockwise).  On the device the rotations are
    never materialised by this function -- SSDN_OP_PACK_INPUT / SSDN_OP_UNROT_* fold them into loads; this host version
    exists for API parity and tests."""
    if angle == 0:
        return x
    if angle == 90:
        return x.flip(-1).transpose(-2, -1)
    if angle == 180:
        return x.flip(-1).flip(-2)
    if angle == 270:
        return x.flip(-2).transpose(-2, -1)
    raise NotImplementedError("Must be rotation divisible by 90 degrees")


def mse2psnr(mse: Tensor, float_imgs: bool = True) -> Tensor:
    peak = 1.0 if float_imgs else 255.0
    return 20 * torch.log10(torch.tensor(peak, device=mse.device)) - 10 * torch.log10(mse)


def calculate_psnr(img: Tensor, ref: Tensor) -> Tensor:
    """Per-sample PSNR for BCHW (or a single CHW image): -10 log10(mean_chw (img-ref)^2) (utils/data.py:94-105)."""
    se = (img - ref) ** 2
    mse = se.reshape(se.shape[0], -1).mean(1) if se.dim() == 4 else se.mean()
    return mse2psnr(mse, img.is_floating_point())


# ---- image I/O helpers of the trainer / evaluator (utils/data.py:69-125) ------------------------------------------------------
def tensor2image(img: Tensor):
    """CHW float tensor in [0,1] -> PIL image.  The dataset classes hand out tensors with H and W SWAPPED (PIL data
    labelled "CWH" and permuted, datasets/hdf5.py:62-72, folder.py:83-84 -- reproduced in ssdn.datasets), and this function
    swaps them back on the way out (utils/data.py:80), so saved PNGs are upright.  A BCHW batch becomes a horizontal strip."""
    import numpy as np
    from PIL import Image
    img = img.detach().cpu().float()
    if img.dim() == 4:
        img = torch.cat(list(img), dim=1)           # stack along the (swapped) first spatial axis = image x axis
    a = np.clip(img.numpy(), 0, 1).transpose(2, 1, 0)         # C, W', H' -> H', W', C
    if a.shape[-1] == 3:
        return Image.fromarray(np.uint8(a * 255), mode="RGB")
    if a.shape[-1] == 1:
        return Image.fromarray(np.uint8(a[..., 0] * 255), mode="L")
    raise NotImplementedError("Cannot convert image with {} channels to PIL image.".format(a.shape[-1]))


def save_tensor_image(img: Tensor, path: str):
    tensor2image(img).save(path)


def set_color_channels(img, channels: int):
    """PIL image -> 1 (weighted RGB->L) or 3 (replicated) channels (utils/data.py:114-125)."""
    cur = len(img.getbands())
    if cur != channels:
        if channels == 1:
            return img.convert("L")
        if channels == 3:
            return img.convert("RGB")
    return img

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

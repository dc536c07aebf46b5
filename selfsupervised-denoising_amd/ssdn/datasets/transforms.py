"""PIL-level transforms of the training loader.  The reference uses torchvision.transforms.RandomCrop(patch,
pad_if_needed=True, padding_mode="reflect") (train.py:756-760); torchvision is an un-pinned dependency that is absent here,
so the crop positions are not bit-pinned to it ("parity unpinned", distribution only): uniform top-left
corner, reflection padding on all sides when the image is smaller than the patch."""
import numpy as np
import torch
from PIL import Image


class RandomCrop:
    def __init__(self, size: int, pad_if_needed: bool = True, padding_mode: str = "reflect"):
        self.size, self.pad_if_needed, self.padding_mode = int(size), pad_if_needed, padding_mode

    def __call__(self, img: Image.Image) -> Image.Image:
        w, h = img.size
        if self.pad_if_needed and (w < self.size or h < self.size):
            a = np.asarray(img)
            pw, ph = max(0, self.size - w), max(0, self.size - h)
            pad = [(ph, ph), (pw, pw)] + ([(0, 0)] if a.ndim == 3 else [])
            # np.pad's reflect needs pad < size on that axis: pad repeatedly for very small images
            while any(p[0] >= a.shape[i] for i, p in enumerate(pad[:2]) if p[0] > 0):
                step = [(min(p[0], a.shape[i] - 1), min(p[1], a.shape[i] - 1)) for i, p in enumerate(pad[:2])] + pad[2:]
                a = np.pad(a, step, mode=self.padding_mode)
                pad = [(p[0] - s[0], p[1] - s[1]) for p, s in zip(pad, step)]
            a = np.pad(a, pad, mode=self.padding_mode)
            img = Image.fromarray(a)
            w, h = img.size
        top = int(torch.randint(0, h - self.size + 1, (1,)))
        left = int(torch.randint(0, w - self.size + 1, (1,)))
        return img.crop((left, top, left + self.size, top + self.size))


def to_tensor(img: Image.Image) -> torch.Tensor:
    """PIL -> float32 CHW in [0,1] (what torchvision.transforms.functional.to_tensor does for 8-bit images)."""
    a = np.asarray(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(a.transpose(2, 0, 1).copy()).float().div_(255.0)

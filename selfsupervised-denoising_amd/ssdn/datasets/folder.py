"""Unlabelled image-folder dataset (drop-in for /root/reference/ssdn/ssdn/datasets/folder.py:18-188): every image file under
a directory (optionally recursive), loaded with PIL, converted to 1 or 3 channels, optional PIL transform, returned as
(tensor, index).  Like the reference (folder.py:83-84, data_format.py:55) the tensor comes out with H and W SWAPPED: PIL data
is labelled "CWH" there and permuted to "CHW"; `ssdn.utils.tensor2image` swaps back when saving.  Kept, because padding /
un-padding metadata and stored evaluation images depend on it."""
import os
from typing import List, Optional

import torch
from PIL import Image
from torch.utils.data import Dataset

from ssdn.datasets.transforms import to_tensor
from ssdn.utils.data import set_color_channels

IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


def is_image_file(path: str) -> bool:
    return path.lower().endswith(IMG_EXTENSIONS)


class UnlabelledImageFolderDataset(Dataset):
    def __init__(self, dir_path: str, extensions: Optional[List[str]] = None, transform=None, recursive: bool = False,
                 output_format: str = "CHW", channels: int = 3):
        self.dir_path, self.transform, self.channels, self.output_format = dir_path, transform, channels, output_format
        exts = tuple(e.lower() for e in extensions) if extensions else IMG_EXTENSIONS
        files = []
        for root, _, names in sorted(os.walk(os.path.expanduser(dir_path))):
            for n in sorted(names):
                if n.lower().endswith(exts):
                    files.append(os.path.join(root, n))
            if not recursive:
                break
        if not files:
            raise RuntimeError("Found 0 files in: " + dir_path + "\nSupported extensions are: " + ",".join(exts))
        self.files = files

    def __getitem__(self, index: int):
        with open(self.files[index], "rb") as f:
            img = Image.open(f)
            img.load()
        img = set_color_channels(img, self.channels)
        if self.transform:
            img = self.transform(img)
        if not isinstance(img, torch.Tensor):
            img = to_tensor(img)
        if self.output_format is not None:
            img = img.permute(0, 2, 1)            # the reference's "CWH" -> "CHW" relabelling: a transpose of H and W
        return img, index

    def image_size(self, index: int, ignore_transform: bool = False) -> torch.Tensor:
        if self.transform is not None and not ignore_transform:
            return torch.tensor(self[index][0].shape)
        with Image.open(self.files[index]) as im:
            w, h = im.size
        return torch.tensor([self.channels, w, h])

    def __len__(self) -> int:
        return len(self.files)

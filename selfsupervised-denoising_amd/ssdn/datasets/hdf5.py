"""Dataset over a file written by `external/dataset_tool_h5.py` (drop-in for /root/reference/ssdn/ssdn/datasets/hdf5.py:19-100):
`images` = variable-length uint8 (raw CHW bytes), `shapes` = int32 [N,3].  h5py is used when it is importable; otherwise the
dependency-free reader `ssdn.datasets.h5lite` parses the one layout directly.  Unlike the reference (which re-opens the file
for every item, hdf5.py:57-59) the file stays open per PROCESS: a handle is never shared across fork (re-opened when the pid
changes) and the dependency-free reader only uses positioned reads.  Output tensors carry the reference's swapped H/W
(hdf5.py:62,71-72), see ssdn.datasets.folder."""
import os

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

from ssdn.datasets.transforms import to_tensor
from ssdn.utils.data import set_color_channels


class HDF5Dataset(Dataset):
    def __init__(self, file_path: str, transform=None, h5_format: str = "CWH", output_format: str = "CHW", channels: int = 3):
        self.file_path, self.transform, self.output_format, self.channels, self.h5_format = file_path, transform, output_format, channels, h5_format
        self._h = None
        self._pid = -1
        self.img_count = len(self._open())
        self._h = None                          # no handle survives the constructor: every process (the parent too) opens its own

    def _open(self):
        if self._h is None or self._pid != os.getpid():      # forked DataLoader workers inherit __dict__, not the handle
            self._pid = os.getpid()
            try:
                import h5py
                real = isinstance(getattr(h5py, "__version__", None), str)      # (not an import stub left in sys.modules)
            except ImportError:
                real = False
            if real:
                self._h = _H5pyFile(self.file_path)
            else:
                from ssdn.datasets.h5lite import ImageFile
                self._h = ImageFile(self.file_path)
        return self._h

    def __getstate__(self):                      # DataLoader workers re-open their own handle
        d = dict(self.__dict__)
        d["_h"] = None
        return d

    def __getitem__(self, index: int):
        chw = self._open().image(index)                       # uint8 [3, h, w] as stored
        img = Image.fromarray(np.ascontiguousarray(chw.transpose(1, 2, 0)))
        img = set_color_channels(img, self.channels)
        if self.transform:
            img = self.transform(img)
        if not isinstance(img, torch.Tensor):
            img = to_tensor(img)
        if self.output_format is not None:
            img = img.permute(0, 2, 1)
        return img, index

    def image_size(self, index: int, ignore_transform: bool = False) -> torch.Tensor:
        if self.transform is not None and not ignore_transform:
            return torch.tensor(self[index][0].shape)
        c, h, w = (int(v) for v in self._open().shapes[index])
        return torch.tensor([self.channels, w, h])

    def __len__(self) -> int:
        return self.img_count


class _H5pyFile:
    def __init__(self, path):
        import h5py
        self.f = h5py.File(path, "r")
        self.shapes = self.f["shapes"][...]

    def __len__(self):
        return self.f["images"].shape[0]

    def image(self, i):
        return np.reshape(self.f["images"][i], self.shapes[i])

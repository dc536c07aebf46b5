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

    # ---- crop at read (the training loader's fast path, ssdn.datasets.device_stream.CleanPatches) -------------------------------
    def patches_u8(self, indexes, size: int) -> np.ndarray:
        """uint8 [len(indexes), channels, size, size]: one uniformly placed size x size crop of every listed image, exactly what
        `self[i]` yields under `RandomCrop(size, pad_if_needed=True, padding_mode="reflect")` as bytes -- same value mapping
        (PIL's RGB -> L weights for one channel), same swapped H/W (out[c, i, j] = stored[c, top + j, left + i]) -- but only the
        crop window is read (memory-mapped file), no PIL image, no float round trip, one call per MINIBATCH.  Crop positions come
        from torch's generator, like RandomCrop's (parity of positions is distributional, transforms.py).  Images smaller than the
        patch, or a non-h5lite backend, take the per-item path."""
        h = self._open()
        n = len(indexes)
        out = np.empty((n, self.channels, size, size), dtype=np.uint8)
        u = torch.rand(2 * n).numpy()
        fast = hasattr(h, "view")
        for k, idx in enumerate(indexes):
            c, ih, iw = (int(v) for v in h.shapes[idx])
            if not fast or ih < size or iw < size or c not in (1, 3):
                img = self[idx][0]                                   # (reflect padding etc.: the generic path; needs a RandomCrop transform)
                out[k] = (img * 255.0).round().clamp_(0, 255).to(torch.uint8).numpy()
                continue
            top = min(int(u[2 * k] * (ih - size + 1)), ih - size)
            left = min(int(u[2 * k + 1] * (iw - size + 1)), iw - size)
            win = h.view(idx)[:, top:top + size, left:left + size]      # (c, y, x) view of the mapped file
            if c == self.channels:
                out[k] = win.transpose(0, 2, 1)
            elif self.channels == 1:                                    # PIL "L": (R * 19595 + G * 38470 + B * 7471 + 0x8000) >> 16
                w32 = win.astype(np.uint32)
                out[k, 0] = ((w32[0] * 19595 + w32[1] * 38470 + w32[2] * 7471 + 0x8000) >> 16).astype(np.uint8).T
            else:                                                       # one stored channel, three wanted: replicated
                out[k] = np.broadcast_to(win[0].T, (3, size, size))
        return out

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

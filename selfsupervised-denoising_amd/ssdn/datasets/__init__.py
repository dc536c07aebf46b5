from ssdn.datasets.noise_wrapper import NoisyDataset  # noqa: F401
from ssdn.datasets.sampler import FixedLengthSampler, SamplingOrder  # noqa: F401
from ssdn.datasets.folder import UnlabelledImageFolderDataset  # noqa: F401
from ssdn.datasets.hdf5 import HDF5Dataset  # noqa: F401
from ssdn.datasets.device_stream import CleanPatches, DevicePatchStream  # noqa: F401

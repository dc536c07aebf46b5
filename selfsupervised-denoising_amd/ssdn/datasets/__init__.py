from ssdn.datasets.noise_wrapper import NoisyDataset  # noqa: F401

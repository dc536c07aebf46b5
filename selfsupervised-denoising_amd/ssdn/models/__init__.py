from ssdn.models.noise_network import NoiseNetwork  # noqa: F401

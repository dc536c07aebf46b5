"""Batch layout contract between the data layer and `Denoiser.run_pipeline` (reference:
/root/reference/ssdn/ssdn/datasets/noise_wrapper.py:46-49,271-280).  Only the index constants and the metadata
vocabulary live here for now; the full noisy-patch dataset is SURVEY.md section 8(f) row N2."""
from enum import Enum


class NoisyDataset:
    INPUT = 0
    REFERENCE = 1
    METADATA = 2

    Metadata = Enum("Metadata", [(n, i + 1) for i, n in enumerate(
        "CLEAN IMAGE_SHAPE INDEXES INPUT_NOISE_VALUES REFERENCE_NOISE_VALUES MASK_COORDS".split())],
        module=__name__, qualname="NoisyDataset.Metadata")

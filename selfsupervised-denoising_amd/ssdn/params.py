"""Enumerations of the `ssdn` configuration / state vocabulary.

CHECKPOINT CONTRACT: `.wt` / `.training` files written by the reference pickle members of these
enums by (module path, class name, value), e.g. `ssdn.params.ConfigValue(4)`.  The classes therefore must live at
`ssdn.params`, carry the same member names and the same values as /root/reference/ssdn/ssdn/params.py:7-93 (string
values for the user-facing enums, 1-based declaration order for the rest).  Declared with the functional Enum API;
tests/test_dropin_surface.py pins names and values against the reference-generated contract fixture.
"""
from enum import Enum

_M = __name__


def _ordered(name, members):
    """Enum whose values are 1..n in declaration order (what `auto()` yields in the reference)."""
    return Enum(name, [(m, i + 1) for i, m in enumerate(members.split())], module=_M, qualname=name)


NoiseAlgorithm = Enum("NoiseAlgorithm", [
    ("SELFSUPERVISED_DENOISING", "ssdn"),
    ("SELFSUPERVISED_DENOISING_MEAN_ONLY", "ssdn_u_only"),
    ("NOISE_TO_NOISE", "n2n"),
    ("NOISE_TO_CLEAN", "n2c"),
    ("NOISE_TO_VOID", "n2v"),
], module=_M, qualname="NoiseAlgorithm")

NoiseValue = Enum("NoiseValue", [("UNKNOWN_CONSTANT", "const"), ("UNKNOWN_VARIABLE", "var"), ("KNOWN", "known")],
                  module=_M, qualname="NoiseValue")

Pipeline = Enum("Pipeline", [("MSE", "mse"), ("SSDN", "ssdn"), ("MASK_MSE", "mask_mse")], module=_M, qualname="Pipeline")

Blindspot = Enum("Blindspot", [("ENABLED", "blindspot"), ("DISABLED", "normal")], module=_M, qualname="Blindspot")

ConfigValue = _ordered("ConfigValue", """
    INFER_CFG ALGORITHM BLINDSPOT PIPELINE IMAGE_CHANNELS
    NOISE_STYLE
    LEARNING_RATE LR_RAMPUP_FRACTION LR_RAMPDOWN_FRACTION
    NOISE_VALUE DIAGONAL_COVARIANCE
    EVAL_INTERVAL PRINT_INTERVAL SNAPSHOT_INTERVAL TRAIN_ITERATIONS
    DATALOADER_WORKERS TRAIN_DATASET_NAME TRAIN_DATASET_TYPE TRAIN_DATA_PATH TRAIN_PATCH_SIZE TRAIN_MINIBATCH_SIZE
    TEST_DATASET_NAME TEST_DATASET_TYPE TEST_DATA_PATH TEST_MINIBATCH_SIZE PIN_DATA_MEMORY
""")

DatasetType = _ordered("DatasetType", "HDF5 FOLDER")

StateValue = _ordered("StateValue", "INITIALISED MODE ITERATION REFERENCE HISTORY")

HistoryValue = _ordered("HistoryValue", "TRAIN EVAL TIMINGS")

PipelineOutput = Enum("PipelineOutput", [
    ("INPUTS", 1),
    ("LOSS", "loss"),
    ("IMG_DENOISED", "out"),
    ("IMG_MU", "out_mu"),
    ("NOISE_STD_DEV", "noise_std"),
    ("MODEL_STD_DEV", "model_std"),
], module=_M, qualname="PipelineOutput")

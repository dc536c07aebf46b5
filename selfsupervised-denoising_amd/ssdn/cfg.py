"""Configuration defaults and inference helpers (drop-in for /root/reference/ssdn/ssdn/cfg.py:10-184).

A configuration is a plain dict keyed by `ssdn.params.ConfigValue`; it is pickled into every checkpoint, so the keys and
the default values below are part of the file format.
"""
import os
from typing import Dict

from ssdn.params import ConfigValue as CV
from ssdn.params import DatasetType, NoiseAlgorithm, Pipeline

DEFAULT_RUN_DIR = "runs"


def base() -> Dict:
    """Defaults of `ssdn train start` (cfg.py:10-32)."""
    cfg = {
        CV.TRAIN_ITERATIONS: 2000000, CV.TRAIN_MINIBATCH_SIZE: 4, CV.TEST_MINIBATCH_SIZE: 2,
        CV.IMAGE_CHANNELS: 3, CV.TRAIN_PATCH_SIZE: 64,
        CV.LEARNING_RATE: 3e-4, CV.LR_RAMPDOWN_FRACTION: 0.1, CV.LR_RAMPUP_FRACTION: 0.3,
        CV.EVAL_INTERVAL: 10000, CV.PRINT_INTERVAL: 1000, CV.SNAPSHOT_INTERVAL: 10000,
        CV.DATALOADER_WORKERS: 4, CV.PIN_DATA_MEMORY: False, CV.DIAGONAL_COVARIANCE: False,
    }
    for k in (CV.TRAIN_DATA_PATH, CV.TRAIN_DATASET_TYPE, CV.TRAIN_DATASET_NAME,
              CV.TEST_DATA_PATH, CV.TEST_DATASET_TYPE, CV.TEST_DATASET_NAME):
        cfg[k] = None
    return cfg


class DatasetName:
    BSD = "bsd"
    IMAGE_NET = "ilsvrc"
    KODAK = "kodak"
    SET14 = "set14"


_PATH_HINTS = (("BSDS300", DatasetName.BSD), ("ILSVRC", DatasetName.IMAGE_NET), ("KODAK", DatasetName.KODAK), ("SET14", DatasetName.SET14))


def infer_datasets(cfg: Dict) -> None:
    """Guess dataset name (substring of the path, case-insensitive; first hit wins when several match, None when
    none does) and container type (file -> HDF5, directory -> FOLDER) for entries left unset (cfg.py:42-93)."""
    def name_of(path):
        hits = [n for key, n in _PATH_HINTS if key.lower() in path.lower()]
        return hits[0] if hits else None

    def type_of(path):
        return DatasetType.FOLDER if os.path.isdir(path) else DatasetType.HDF5

    for path_k, name_k, type_k in ((CV.TRAIN_DATA_PATH, CV.TRAIN_DATASET_NAME, CV.TRAIN_DATASET_TYPE),
                                   (CV.TEST_DATA_PATH, CV.TEST_DATASET_NAME, CV.TEST_DATASET_TYPE)):
        path = cfg.get(path_k)
        if path is None:
            continue
        if cfg.get(name_k) is None:
            cfg[name_k] = name_of(path)
        if cfg.get(type_k) is None:
            cfg[type_k] = type_of(path)


_TEST_LENGTH = {DatasetName.BSD: 300, DatasetName.KODAK: 240, DatasetName.SET14: 280}


def test_length(dataset_name: str) -> int:
    """Noisy instances evaluated per test set: 3x BSD300, 10x Kodak, 20x Set14 (cfg.py:96-113)."""
    return _TEST_LENGTH[dataset_name]


_PIPELINE = {NoiseAlgorithm.SELFSUPERVISED_DENOISING: Pipeline.SSDN,
             NoiseAlgorithm.SELFSUPERVISED_DENOISING_MEAN_ONLY: Pipeline.MSE,
             NoiseAlgorithm.NOISE_TO_NOISE: Pipeline.MSE,
             NoiseAlgorithm.NOISE_TO_CLEAN: Pipeline.MSE,
             NoiseAlgorithm.NOISE_TO_VOID: Pipeline.MASK_MSE}
_BLINDSPOT = {NoiseAlgorithm.SELFSUPERVISED_DENOISING: True,
              NoiseAlgorithm.SELFSUPERVISED_DENOISING_MEAN_ONLY: True,
              NoiseAlgorithm.NOISE_TO_NOISE: False,
              NoiseAlgorithm.NOISE_TO_CLEAN: False,
              NoiseAlgorithm.NOISE_TO_VOID: False}


def infer_pipeline(algorithm: NoiseAlgorithm) -> Pipeline:
    if algorithm not in _PIPELINE:
        raise NotImplementedError("Algorithm does not have a default pipeline.")
    return _PIPELINE[algorithm]


def infer_blindspot(algorithm: NoiseAlgorithm) -> bool:
    if algorithm not in _BLINDSPOT:
        raise NotImplementedError("Not known if algorithm requires blindspot.")
    return _BLINDSPOT[algorithm]


def infer(cfg: Dict, model_only: bool = False) -> Dict:
    if cfg.get(CV.PIPELINE) is None:
        cfg[CV.PIPELINE] = infer_pipeline(cfg[CV.ALGORITHM])
    if cfg.get(CV.BLINDSPOT) is None:
        cfg[CV.BLINDSPOT] = infer_blindspot(cfg[CV.ALGORITHM])
    if not model_only:
        infer_datasets(cfg)
    return cfg


def config_name(cfg: Dict) -> str:
    """`ssdn-gauss25-sigma_known[-mono][-diag]` grammar of cfg.py:158-184."""
    cfg = infer(cfg)
    alg = cfg[CV.ALGORITHM]
    parts = [alg.value]
    if cfg[CV.PIPELINE] != infer_pipeline(alg):
        parts.append(cfg[CV.PIPELINE].value + "_pipeline")
    if cfg[CV.BLINDSPOT] != infer_blindspot(alg):
        parts.append("blindspot" if cfg[CV.BLINDSPOT] else "blindspot_disabled")
    parts.append(cfg[CV.NOISE_STYLE])
    ssdn_pipe = cfg[CV.PIPELINE] == Pipeline.SSDN
    if ssdn_pipe:
        parts.append("sigma_" + cfg[CV.NOISE_VALUE].value)
    if cfg[CV.IMAGE_CHANNELS] == 1:
        parts.append("mono")
    if ssdn_pipe and cfg[CV.DIAGONAL_COVARIANCE]:
        parts.append("diag")
    return "-".join(parts)

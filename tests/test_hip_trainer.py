"""GPU: the drop-in CLI end to end on a tiny synthetic image folder -- `ssdn train start`, `ssdn train resume`, `ssdn eval` --
checking the run-directory artefacts of the reference (train.py:128-233,397-408; eval.py:45-125): log.txt, scalars, `.training`
and `.wt` snapshots, final weights, validation PNGs, psnrs.csv; and that training actually trains."""
import csv
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make_folder(path, n=6, seed=0):
    from PIL import Image
    os.makedirs(path, exist_ok=True)
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:80, 0:112]
    for i in range(n):
        base = 0.5 + 0.35 * np.sin(xx / (6.0 + i) + i) * np.cos(yy / (9.0 - i))       # smooth structure a denoiser can learn
        img = np.stack([base, np.roll(base, 5 * i, 1), 1 - base], -1) + rng.rand(80, 112, 3) * 0.05
        Image.fromarray(np.uint8(np.clip(img, 0, 1) * 255)).save(os.path.join(path, "img%02d.png" % i))


def _scalars(run):
    rows = list(csv.DictReader(open(os.path.join(run, "scalars.csv"))))
    out = {}
    for r in rows:
        out.setdefault(r["tag"], []).append((int(r["step"]), float(r["value"])))
    return out


def test_cli_train_resume_eval_round_trip(tmp_path):
    from ssdn.__main__ import start_cli
    from ssdn.params import ConfigValue, StateValue
    data = str(tmp_path / "ILSVRC_like")
    val = str(tmp_path / "kodak_like")
    _make_folder(data, 6, 0)
    _make_folder(val, 2, 1)
    runs = str(tmp_path / "runs")
    torch.manual_seed(20260929)          # weights, sampling order and the device noise stream all derive from the host RNG
    trainer = start_cli(["train", "start", "-a", "ssdn", "-n", "gauss25", "--noise_value", "known", "-t", data, "-v", val,
                         "-i", "192", "--train_batch_size", "8", "--validation_batch_size", "2", "--patch_size", "32",
                         "--eval_interval", "96", "--print_interval", "48", "--checkpoint_interval", "96", "--runs_dir", runs])
    run = trainer.run_dir_path
    assert os.path.basename(run) == "00000-train-ilsvrc-kodak-ssdn-gauss25-sigma_known-iter192"
    assert trainer.state[StateValue.ITERATION] == 192
    assert os.path.exists(os.path.join(run, "log.txt")) and "TRAINING FINISHED" in open(os.path.join(run, "log.txt")).read()
    tfiles = sorted(os.path.basename(p) for p in glob.glob(os.path.join(run, "training", "*.training")))
    assert tfiles == ["model_00000000.training", "model_00000096.training", "model_00000192.training"]
    assert os.path.exists(os.path.join(run, "final-ssdn-gauss25-sigma_known.wt"))
    assert len(glob.glob(os.path.join(run, "val_imgs", "*_out.png"))) >= 2            # one validation image per evaluation
    sc = _scalars(run)
    assert [s for s, _ in sc["train/loss"]] == [48, 96, 144, 192]
    assert all(np.isfinite(v) for _, v in sc["train/loss"]) and sc["train/loss"][-1][1] < sc["train/loss"][0][1]
    assert "valid/psnr_out" in sc and "train/psnr_mu_out" in sc and "train/learning_rate" in sc
    # the kodak-named validation set is evaluated cfg.test_length("kodak") = 240 instances per evaluation
    assert trainer.cfg[ConfigValue.TEST_DATASET_NAME] == "kodak"

    # resume: continue the same directory to 256 images
    t2 = start_cli(["train", "resume", run, "-i", "256"])
    assert t2.run_dir_path == run and t2.state[StateValue.ITERATION] == 256
    assert os.path.exists(os.path.join(run, "training", "model_00000256.training"))
    assert float(torch.load(os.path.join(run, "training", "model_00000256.training"), weights_only=False)["optimizer"]["state"][0]["step"]) == 32.0

    # evaluate the final weights on a (non-named) folder: every image once, padded to x32 squares and un-padded again
    ev = start_cli(["eval", "-m", os.path.join(run, "final-ssdn-gauss25-sigma_known.wt"), "-d", data, "--runs_dir", runs, "--batch_size", "2"])
    erun = ev.run_dir_path
    assert os.path.basename(erun).startswith("00001-eval-")
    rows = list(csv.DictReader(open(os.path.join(erun, "psnrs.csv"))))
    assert len(rows) == 6 and set(rows[0]) == {"id", "psnr_nsy", "psnr_out", "psnr_mu_out"}
    assert all(np.isfinite(float(r["psnr_out"])) and float(r["psnr_out"]) > 5.0 for r in rows)   # a 24-step model is not good, but sane
    from PIL import Image
    im = Image.open(sorted(glob.glob(os.path.join(erun, "eval_imgs", "*_out.png")))[0])
    assert im.size == (112, 80)                                                      # un-padded, upright


def test_cli_train_n2v_mask_coordinates_flow_through_the_loader(tmp_path):
    from ssdn.__main__ import start_cli
    data = str(tmp_path / "set14_like")
    _make_folder(data, 4, 2)
    t = start_cli(["train", "start", "-a", "n2v", "-n", "gauss25", "-t", data, "-i", "32", "--train_batch_size", "8",
                   "--patch_size", "64", "--print_interval", "16", "--checkpoint_interval", "32", "--runs_dir", str(tmp_path / "r")])
    sc = _scalars(t.run_dir_path)
    assert [s for s, _ in sc["train/loss"]] == [16, 32] and all(np.isfinite(v) for _, v in sc["train/loss"])

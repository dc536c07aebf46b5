"""CPU: the drop-in surface of the trainer / CLI / checkpoints against the contract captured from
the LIVE reference (tests/golden/g_ckpt_contract.json "training_file" + a reference-written `.training` file,
oracle/gen_golden.py).  No compute call: the step itself is covered by the GPU tests."""
import gzip
import io
import json
import os

import pytest
import torch

import ssdn
from ssdn.params import ConfigValue, HistoryValue, NoiseAlgorithm, NoiseValue, StateValue
from ssdn.train import DenoiserTrainer, resume_run


@pytest.fixture(scope="module")
def contract(golden_dir):
    return json.load(open(os.path.join(golden_dir, "g_ckpt_contract.json")))["training_file"]


def _cfg():
    cfg = ssdn.cfg.base()
    cfg[ConfigValue.ALGORITHM] = NoiseAlgorithm.SELFSUPERVISED_DENOISING
    cfg[ConfigValue.NOISE_STYLE] = "gauss25"
    cfg[ConfigValue.NOISE_VALUE] = NoiseValue.KNOWN
    return cfg


def _trainer(tmp_path, iters=4):
    cfg = _cfg()
    cfg[ConfigValue.TRAIN_ITERATIONS] = iters
    tr = DenoiserTrainer(cfg, runs_dir=str(tmp_path / "runs"))
    tr.denoiser = ssdn.denoiser.Denoiser(tr.cfg, device="cpu")
    tr.init_state()
    from ssdn.datasets import FixedLengthSampler
    tr.train_sampler = FixedLengthSampler(list(range(7)), num_samples=20, shuffled=True)
    _ = iter(tr.train_sampler)
    return tr


def test_training_file_written_here_has_the_reference_layout(tmp_path, contract):
    tr = _trainer(tmp_path)
    d = tr.denoiser
    d.adam_steps = 2
    d.adam_m.uniform_(-1, 1)
    d.adam_v.uniform_(0, 1)
    tr.state[StateValue.ITERATION] = 4
    tr.state[StateValue.HISTORY][HistoryValue.TRAIN]["loss"] += torch.ones(2, 1)
    tr.state[StateValue.HISTORY][HistoryValue.TIMINGS]["total"].update()
    sd = tr.state_dict()
    assert sorted(sd.keys()) == contract["keys"]
    assert sorted(k.name for k in sd["state"].keys()) == contract["state_keys"]
    assert sorted(k.name for k in sd["state"][StateValue.HISTORY].keys()) == contract["history_keys"]
    assert sorted(sd["train_order_iter"].keys()) == contract["train_order_iter_keys"] and sd["train_order_iter"]["index"] == 4
    osd = sd["optimizer"]
    assert sorted(osd.keys()) == contract["optimizer_keys"]
    assert sorted(osd["state"][0].keys()) == contract["optimizer_state_entry_keys"]
    assert len(osd["param_groups"][0]["params"]) == contract["optimizer_n_params"]
    assert [list(osd["state"][i]["exp_avg"].shape) for i in range(len(osd["state"]))] == contract["optimizer_param_shapes"]
    assert list(osd["param_groups"][0]["betas"]) == contract["optimizer_betas"]
    torch.optim.Adam(d.parameters(), betas=[0.9, 0.99]).load_state_dict(osd)            # accepted by the real optimiser
    # pickled globals: nothing outside the reference's vocabulary (+ torch internals)
    import pickletools
    import zipfile
    tr.snapshot()
    path = os.path.join(tr.run_dir_path, "training", "model_00000004.training")
    zf = zipfile.ZipFile(path)
    pk = [n for n in zf.namelist() if n.endswith("data.pkl")][0]
    globs = {"%s.%s" % tuple(a.split(" ")[:2]) for op, a, _ in pickletools.genops(zf.read(pk)) if op.name == "GLOBAL"}
    ours = {g for g in globs if g.startswith("ssdn.")}
    assert ours <= set(contract["globals"]), ours - set(contract["globals"])
    assert tr.run_dir == contract["run_dir"] and tr.config_name() == contract["config_name"]


def test_reference_written_training_file_resumes_here(tmp_path, golden_dir, contract):
    """a `.training` file written by the reference's DenoiserTrainer loads, resumes and is re-written equivalently"""
    run = tmp_path / "runs" / "00003-train-ssdn-gauss25-sigma_known-iter4"
    (run / "training").mkdir(parents=True)
    raw = gzip.open(os.path.join(golden_dir, "g_training_file.training.gz")).read()
    (run / "training" / "model_00000004.training").write_bytes(raw)
    (run / "training" / "model_00000002.training").write_bytes(b"older, never opened")
    tr = resume_run(str(run))
    assert tr.state[StateValue.ITERATION] == 4 and tr.run_dir == run.name and tr.runs_dir == str(tmp_path / "runs")
    assert tr.cfg[ConfigValue.ALGORITHM] == NoiseAlgorithm.SELFSUPERVISED_DENOISING
    d = tr.denoiser
    bias = d.get_model(d.MODEL, False).get_submodule("output_block.4").bias.detach().cpu()
    assert torch.allclose(bias, torch.arange(9.0) * 0.25 - 1)                       # the recognisable values of the fixture
    assert d.adam_steps == 2
    m = d.optimizer_state_dict()["state"][contract["optimizer_param_shapes"].index([9])]["exp_avg"]   # output_conv.bias keeps its moments
    assert m.shape == (9,) and float(m.abs().sum()) > 0
    assert tr._train_iter is not None and tr._train_iter.index == contract["train_order_index"]
    hist = tr.state[StateValue.HISTORY]
    assert hist[HistoryValue.TRAIN]["n"] == 4 and float(hist[HistoryValue.TRAIN]["loss"].accumulated()) > 0
    assert hist[HistoryValue.TIMINGS]["total"].last_time is None                       # absolute times were forgotten
    # evaluator accepts both file kinds
    from ssdn.eval import DenoiserEvaluator
    ev = DenoiserEvaluator(str(run / "training" / "model_00000004.training"), runs_dir=str(tmp_path / "evals"))
    assert ev.run_dir.startswith("00000-eval-") and ev.runs_dir == str(tmp_path / "evals")
    tr.snapshot(model_only=True)
    ev2 = DenoiserEvaluator(os.path.join(tr.run_dir_path, "models", "model_00000004.wt"), runs_dir=str(tmp_path / "evals"))
    assert ev2.cfg[ConfigValue.NOISE_STYLE] == "gauss25"


def test_learning_rate_schedule_and_run_directories(tmp_path, golden_dir):
    import numpy as np
    g = np.load(os.path.join(golden_dir, "g_lr.npz"))
    cfg = _cfg()                                                     # defaults: 2 M iterations, the table's setting
    tr = DenoiserTrainer(cfg, runs_dir=str(tmp_path / "r"))
    tr.state[StateValue.ITERATION] = 0
    assert tr.learning_rate == 0.0                                   # lr(0) = 0: the swapped fractions (train.py:276-282)
    for it, lr in zip(g["iters"], g["lr"]):
        tr.state[StateValue.ITERATION] = int(it)
        assert tr.learning_rate == pytest.approx(float(lr), rel=1e-12, abs=1e-18)
    tr.state[StateValue.ITERATION] = 0
    cfg[ConfigValue.TRAIN_ITERATIONS] = 1000
    assert tr.run_dir == "00000-train-ssdn-gauss25-sigma_known-iter1k"
    os.makedirs(os.path.join(tr.runs_dir, "00007-train-x"))
    assert DenoiserTrainer(cfg, runs_dir=str(tmp_path / "r")).run_dir.startswith("00008-train-")
    cfg2 = dict(cfg)
    cfg2[ConfigValue.TRAIN_ITERATIONS] = 2000000
    cfg2[ConfigValue.TRAIN_DATASET_NAME], cfg2[ConfigValue.TEST_DATASET_NAME] = "ilsvrc", "kodak"
    assert DenoiserTrainer(cfg2, runs_dir=str(tmp_path / "q")).config_name() == "ilsvrc-kodak-ssdn-gauss25-sigma_known-iter2m"


def test_cli_surface(tmp_path):
    from ssdn.cli.cli import build_parser
    parser, cmds = build_parser()
    a = vars(parser.parse_args(["train", "start", "-a", "ssdn", "-n", "gauss25", "--noise_value", "known", "-t", "x.h5", "-i", "1000",
                                "--train_batch_size", "32", "--runs_dir", str(tmp_path)]))
    assert a["command"] == "train" and a["train_cmd"] == "start" and a["algorithm"] == "ssdn" and a["iterations"] == 1000
    assert a["mono"] is False and a["diagonal"] is False
    a = vars(parser.parse_args(["train", "resume", "runs/00001-train-x", "-i", "50"]))
    assert a["train_cmd"] == "resume" and a["run_dir"] == "runs/00001-train-x" and a["train_dataset"] is None
    a = vars(parser.parse_args(["eval", "-m", "m.wt", "-d", "kodak", "--batch_size", "2"]))
    assert a["command"] == "eval" and a["model"] == "m.wt" and a["batch_size"] == 2
    with pytest.raises(SystemExit):
        parser.parse_args(["train", "start", "-a", "nonsense", "-n", "gauss25", "-t", "x", "-i", "1"])
    with pytest.raises(SystemExit):                                   # ssdn needs --noise_value (cmds/train.py:143-144)
        args = vars(parser.parse_args(["train", "start", "-a", "ssdn", "-n", "gauss25", "-t", "x", "-i", "1"]))
        args["PARSER"] = parser
        cmds["train"].execute(args)


def test_native_tensorboard_event_file(tmp_path):
    """N4: the trainer's scalars also go to a TensorBoard event file written without the tensorboard package: TFRecord framing with
    masked CRC-32C, Event { wall_time, step, file_version | summary { value { tag, simple_value } } }."""
    import struct
    from ssdn import logging_helper as lh
    assert lh.crc32c(b"123456789") == 0xE3069283                      # the CRC-32C check value
    w = lh.ScalarWriter(str(tmp_path))
    w.add_scalar("train/loss", 1.5, 64)
    w.add_scalar("valid/psnr_out", 31.25, 128)
    w.close()
    files = [f for f in os.listdir(tmp_path) if f.startswith("events.out.tfevents.")]
    assert len(files) == 1
    raw = open(os.path.join(tmp_path, files[0]), "rb").read()
    recs, pos = [], 0
    while pos < len(raw):
        (n,) = struct.unpack_from("<Q", raw, pos)
        assert struct.unpack_from("<I", raw, pos + 8)[0] == lh._masked_crc(raw[pos:pos + 8])
        data = raw[pos + 12:pos + 12 + n]
        assert struct.unpack_from("<I", raw, pos + 12 + n)[0] == lh._masked_crc(data)
        recs.append(data)
        pos += 16 + n
    assert len(recs) == 3 and b"brain.Event:2" in recs[0]
    assert recs[1][0] == 0x09 and b"train/loss" in recs[1] and struct.pack("<f", 1.5) in recs[1] and recs[1][9:11] == b"\x10\x40"
    assert b"valid/psnr_out" in recs[2] and struct.pack("<f", 31.25) in recs[2] and recs[2][9:12] == b"\x10\x80\x01"
    assert open(os.path.join(tmp_path, "scalars.csv")).read().count("\n") == 3

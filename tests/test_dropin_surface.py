"""CPU: the drop-in surface the hot path sits behind -- enum vocabulary, cfg defaults / inference / config_name, LR ramp,
Denoiser / NoiseNetwork state_dict layout and pickle globals -- against the contract captured from the live reference
(tests/golden/g_ckpt_contract.json, g_lr.npz) and, when /root/reference is present, against the reference itself."""
import io
import json
import os
import pickletools
import zipfile

import numpy as np
import pytest
import torch

import ssdn
from ssdn.denoiser import Denoiser
from ssdn.models import NoiseNetwork
from ssdn.params import ConfigValue, NoiseAlgorithm, NoiseValue, Pipeline, PipelineOutput, StateValue, HistoryValue, DatasetType


def make_cfg(alg, style="gauss25", mode="known", ch=3):
    cfg = ssdn.cfg.base()
    cfg[ConfigValue.ALGORITHM] = NoiseAlgorithm(alg)
    cfg[ConfigValue.NOISE_STYLE] = style
    cfg[ConfigValue.NOISE_VALUE] = NoiseValue(mode)
    cfg[ConfigValue.IMAGE_CHANNELS] = ch
    return ssdn.cfg.infer(cfg, model_only=True)


@pytest.fixture(scope="module")
def contract(golden_dir):
    return json.load(open(os.path.join(golden_dir, "g_ckpt_contract.json")))


@pytest.mark.parametrize("tag,alg,mode", [("ssdn_known", "ssdn", "known"), ("ssdn_var", "ssdn", "var"), ("ssdn_const", "ssdn", "const"), ("n2c", "n2c", "known")])
def test_state_dict_layout_and_pickle_globals(contract, tag, alg, mode):
    d = Denoiser(make_cfg(alg, mode=mode), device="cpu")
    sd = d.state_dict()
    want = contract[tag]
    assert list(sd.keys()) == want["keys"]
    for k, shp in want["shapes"].items():
        assert list(sd[k].shape) == shp, k
    assert d.config_name() == want["config_name"]
    # alias pair shares storage, as in the reference (noise_network.py:149-156)
    assert sd["_models.denoiser_model.output_conv.weight"].data_ptr() == sd["_models.denoiser_model.output_block.4.weight"].data_ptr()
    assert sd["models.denoiser_model.module.output_conv.weight"].data_ptr() == sd["_models.denoiser_model.output_conv.weight"].data_ptr()
    buf = io.BytesIO()
    torch.save(sd, buf)
    zf = zipfile.ZipFile(io.BytesIO(buf.getvalue()))
    pk = [n for n in zf.namelist() if n.endswith("data.pkl")][0]
    globs = {"%s.%s" % tuple(a.split(" ")[:2]) for op, a, _ in pickletools.genops(zf.read(pk)) if op.name == "GLOBAL"}
    assert {g for g in globs if g.startswith("ssdn.")} <= set(want["globals"])
    # round trip through a file, cfg enums keep their identity
    back = torch.load(io.BytesIO(buf.getvalue()), weights_only=False)
    assert back["cfg"][ConfigValue.ALGORITHM] is NoiseAlgorithm(alg)
    d2 = Denoiser.from_state_dict(back) if False else Denoiser(back["cfg"], device="cpu")
    d2.load_state_dict(back, strict=False)
    assert torch.equal(d2.flat[: d._n_main], d.flat[: d._n_main])
    assert {k.name: (v.name if hasattr(v, "name") else v) for k, v in sd["cfg"].items()} == want["cfg"]


def test_enum_vocabulary_matches_reference_values():
    assert [m.name for m in ConfigValue][:5] == ["INFER_CFG", "ALGORITHM", "BLINDSPOT", "PIPELINE", "IMAGE_CHANNELS"]
    assert ConfigValue.PIN_DATA_MEMORY.value == 26 and ConfigValue.NOISE_STYLE.value == 6
    assert StateValue.ITERATION.value == 3 and HistoryValue.TIMINGS.value == 3 and DatasetType.FOLDER.value == 2
    assert PipelineOutput.INPUTS.value == 1 and PipelineOutput.IMG_DENOISED.value == "out"
    assert NoiseAlgorithm("n2v") is NoiseAlgorithm.NOISE_TO_VOID and Pipeline("mask_mse") is Pipeline.MASK_MSE
    assert ConfigValue.__module__ == "ssdn.params" and ConfigValue.__qualname__ == "ConfigValue"


def test_cfg_defaults_and_inference():
    b = ssdn.cfg.base()
    assert b[ConfigValue.TRAIN_ITERATIONS] == 2000000 and b[ConfigValue.TRAIN_MINIBATCH_SIZE] == 4
    assert b[ConfigValue.LEARNING_RATE] == 3e-4 and b[ConfigValue.LR_RAMPDOWN_FRACTION] == 0.1 and b[ConfigValue.LR_RAMPUP_FRACTION] == 0.3
    c = make_cfg("n2v")
    assert c[ConfigValue.PIPELINE] is Pipeline.MASK_MSE and c[ConfigValue.BLINDSPOT] is False
    assert ssdn.cfg.config_name(make_cfg("ssdn", "poisson30", "const", 1)) == "ssdn-poisson30-sigma_const-mono"
    assert ssdn.cfg.config_name(make_cfg("n2c")) == "n2c-gauss25"
    assert ssdn.cfg.test_length("kodak") == 240
    c = ssdn.cfg.base()
    c[ConfigValue.TRAIN_DATA_PATH] = "/data/ILSVRC2012_val.h5"
    ssdn.cfg.infer_datasets(c)
    assert c[ConfigValue.TRAIN_DATASET_NAME] == "ilsvrc" and c[ConfigValue.TRAIN_DATASET_TYPE] is DatasetType.HDF5


def test_lr_ramp(golden_dir):
    g = np.load(os.path.join(golden_dir, "g_lr.npz"))
    for i, lr in zip(g["iters"], g["lr"]):
        # the trainer passes (rampdown, rampup) into (ramp_up, ramp_down): train.py:276-282
        assert ssdn.utils.compute_ramped_lrate(int(i), 2000000, 0.1, 0.3, 3e-4) == pytest.approx(float(lr), rel=1e-12, abs=0)


def test_noise_network_surface():
    net = NoiseNetwork(3, 9, blindspot=True, device="cpu")
    assert net.blindspot is True and NoiseNetwork.input_wh_mul() == 32
    assert len(net.state_dict()) == 42
    assert sum(p.numel() for p in net.parameters()) == 1269129
    assert sum(p.numel() for p in NoiseNetwork(3, 1, device="cpu").parameters()) == 1102177
    w = net.get_submodule("decode_block_1.2").weight
    assert float(w.std()) == pytest.approx((2 / 1.01 / 864) ** 0.5, rel=0.03)
    assert float(NoiseNetwork(3, 1, zero_output_weights=True, device="cpu").output_conv.weight.abs().max()) == 0.0
    with pytest.raises(Exception, match="no CPU fallback"):
        net(torch.zeros(1, 3, 32, 32))


def test_same_seed_same_weights_as_reference():
    """H8: init_weights draws in the reference's module order -> bit-identical initial weights for the same torch seed."""
    import ref_shim
    if not ref_shim.available():
        pytest.skip("reference tree not present on this machine")
    ref = ref_shim.import_reference()
    with ref_shim.reference_modules(ref):
        from ssdn.models import NoiseNetwork as RefNet
        torch.manual_seed(123)
        r = RefNet(3, 9, blindspot=True)
        rsd = {k: v.clone() for k, v in r.state_dict().items()}
    torch.manual_seed(123)
    ours = NoiseNetwork(3, 9, blindspot=True, device="cpu").state_dict()
    assert list(ours.keys()) == list(rsd.keys())
    for k in rsd:
        assert torch.equal(ours[k], rsd[k]), k

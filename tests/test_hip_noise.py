"""SSDN_OP_NOISE (csrc/elementwise.hip::k_noise) -- the training patch stream's per-sample work on the device (SURVEY.md section 8f
N2; reference: datasets/noise_wrapper.py:66-135, utils/noise.py:54-107, utils/n2v_ups.py:40-88).

The random stream is the kernel's own (Philox), so parity with the reference is STATISTICAL (SURVEY.md section 8c "parity
unpinned"): the moments of every noise style are compared with the values the LIVE reference produced on a constant 0.5 image
(tests/golden/g_ckpt_contract.json "data_layer"/"styles", written by oracle/gen_golden.py), the Noise2Void geometry with
"data_layer"/"n2v", and everything deterministic (clean / 255, clipping, parameter ranges, the replacement rule) exactly."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from ssdn.hip import lib as L


@pytest.fixture(scope="module")
def contract(golden_dir):
    return json.load(open(os.path.join(golden_dir, "g_ckpt_contract.json")))["data_layer"]


def _run(u8, style, clip, lo, hi, seed=7, offset=0, ref=False, n2v=False, device="cuda:0"):
    B, Cn, H, W = u8.shape
    dev = torch.device(device)
    u8 = u8.to(dev).contiguous()
    f32 = dict(dtype=torch.float32, device=dev)
    out = {"clean": torch.full((B, Cn, H, W), -7.0, **f32), "noisy": torch.full((B, Cn, H, W), -7.0, **f32),
           "ref": torch.full((B, Cn, H, W), -7.0, **f32) if ref else None, "param": torch.full((B, Cn), -7.0, **f32),
           "param_ref": torch.full((B, Cn), -7.0, **f32) if ref else None,
           "coords": torch.full((B, (W // 8) * (H // 8), 2), -7, dtype=torch.int64, device=dev) if n2v else None}
    a = L.NoiseArgs()
    a.clean_u8, a.clean32, a.noisy32, a.param = u8.data_ptr(), out["clean"].data_ptr(), out["noisy"].data_ptr(), out["param"].data_ptr()
    a.ref32 = out["ref"].data_ptr() if ref else None
    a.param_ref = out["param_ref"].data_ptr() if ref else None
    a.coords = out["coords"].data_ptr() if n2v else None
    a.B, a.C, a.H, a.W = B, Cn, H, W
    a.style, a.clip, a.p_lo, a.p_hi = style, int(clip), lo, hi
    a.n2v_box, a.n2v_radius = (8 if n2v else 0), 2
    a.seed, a.offset = seed, offset
    rec = (L.OpRec * 1)()
    rec[0].type, rec[0].lane, rec[0].args = L.OP["noise"], 0, C.cast(C.pointer(a), C.c_void_p)
    L.check(L.load().ssdn_run_ops(rec, 1, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    torch.cuda.synchronize()
    return {k: (v.cpu() if v is not None else None) for k, v in out.items()}


def test_noise_op_validates_its_arguments_without_a_gpu():
    """host-side checks come before any launch"""
    lib = L.load()
    a = L.NoiseArgs()
    rec = (L.OpRec * 1)()
    rec[0].type, rec[0].args = L.OP["noise"], C.cast(C.pointer(a), C.c_void_p)
    assert lib.ssdn_run_ops(rec, 1, None) != 0 and b"clean_u8" in lib.ssdn_last_error()
    buf = (C.c_float * 16)()
    a.clean_u8, a.noisy32 = C.cast(buf, C.c_void_p), C.cast(buf, C.c_void_p)
    a.B = a.C = 1
    a.H = a.W = 2
    a.style = 3
    assert lib.ssdn_run_ops(rec, 1, None) != 0 and b"style" in lib.ssdn_last_error()
    a.style, a.p_lo, a.p_hi = 1, 0.0, 0.0
    assert lib.ssdn_run_ops(rec, 1, None) != 0 and b"lambda" in lib.ssdn_last_error()
    a.style, a.p_lo, a.p_hi, a.n2v_box = 0, 0.1, 0.1, 8
    assert lib.ssdn_run_ops(rec, 1, None) != 0 and b"coords" in lib.ssdn_last_error()


@pytest.mark.gpu
def test_noise_moments_match_the_reference_styles(contract):
    """constant image 128 / 255 (the contract used 0.5): mean, std dev, range and parameter of every style"""
    from ssdn.utils import noise
    u8 = torch.full((4, 3, 64, 64), 128, dtype=torch.uint8)
    c0 = 128 / 255.0
    for st, want in contract["styles"].items():
        kind, params, clip = noise.parse_style(st)
        vals = [(p / 255.0 if (kind == "gauss" and isinstance(p, int)) else float(p)) for p in params]
        lo, hi = (vals[0], vals[0]) if len(vals) == 1 else (vals[0], vals[1])
        o = _run(u8, 0 if kind == "gauss" else 1, clip, lo, hi, seed=11)
        y, par = o["noisy"], o["param"]
        assert torch.equal(o["clean"], torch.full_like(o["clean"], float(np.float32(128) / np.float32(255))))
        if lo == hi:
            assert float((par - np.float32(lo)).abs().max()) == 0.0 and want["coeff"] == pytest.approx(lo, rel=1e-6)
        else:                                            # a parameter per sample AND channel, uniform over the range
            assert float(par.min()) >= lo and float(par.max()) < hi and len(torch.unique(par)) == par.numel()
            assert lo < min(want["coeff"]) and max(want["coeff"]) < hi
        for b in range(4):
            for c in range(3):
                p, plane = float(par[b, c]), y[b, c].double()
                if kind == "gauss":
                    assert float(plane.mean()) == pytest.approx(c0, abs=4 * p / 64 + 1e-4)
                    assert float(plane.std()) == pytest.approx(p, rel=0.06)
                else:                                    # rate-1 noise on lambda x: mean shift 1 / lambda, std dev 1 / lambda
                    assert float(plane.mean()) == pytest.approx(c0 + 1 / p, abs=4 / p / 64 + 1e-4)
                    assert float(plane.std()) == pytest.approx(1 / p, rel=0.06)
                    k = ((plane - c0) * p)[plane < 1.0]  # ... and the noise is integer counts (where the clip did not cut it)
                    assert float((k - k.round()).abs().max()) < 1e-3 and float(k.min()) >= -1e-3
        if lo == hi:                                     # whole tensor against the reference's numbers (its image was 0.5)
            assert float(y.mean()) - c0 == pytest.approx(want["mean"] - 0.5, abs=1.5e-3)
            assert float(y.std()) == pytest.approx(want["std"], rel=0.03)
        if clip:
            assert float(y.min()) >= 0.0 and float(y.max()) <= 1.0


@pytest.mark.gpu
def test_gaussian_shape_clipping_and_determinism():
    u8 = torch.zeros((2, 3, 64, 64), dtype=torch.uint8)
    u8[1] = 255
    a = _run(u8, 0, False, 0.1, 0.1, seed=3, offset=5)
    z = (a["noisy"][0] / 0.1).double().flatten()        # clean = 0: pure N(0, 1)
    assert abs(float(z.mean())) < 0.03 and float(z.std()) == pytest.approx(1.0, abs=0.03)
    assert float((z ** 4).mean()) == pytest.approx(3.0, abs=0.25)                       # kurtosis of a normal
    assert float((z.abs() > 2).double().mean()) == pytest.approx(0.0455, abs=0.006)     # two-sided 2 sigma tail
    assert float(a["noisy"][0].min()) < -0.25 and float(a["noisy"][1].max()) > 1.25     # `_nc`: nothing clipped
    b = _run(u8, 0, True, 0.1, 0.1, seed=3, offset=5)
    assert torch.equal(b["noisy"], a["noisy"].clamp(0, 1))                               # same stream, clipped
    assert torch.equal(_run(u8, 0, False, 0.1, 0.1, seed=3, offset=5)["noisy"], a["noisy"])   # deterministic in (seed, offset)
    c = _run(u8, 0, False, 0.1, 0.1, seed=3, offset=6)["noisy"]
    d = _run(u8, 0, False, 0.1, 0.1, seed=4, offset=5)["noisy"]
    for other in (c, d):
        corr = float(torch.corrcoef(torch.stack([a["noisy"][0].flatten(), other[0].flatten()]))[0, 1])
        assert abs(corr) < 0.03                                                          # fresh offset / seed: an independent draw
    r = _run(u8, 0, False, 0.1, 0.1, seed=3, offset=5, ref=True)
    assert torch.equal(r["noisy"], a["noisy"])
    corr = float(torch.corrcoef(torch.stack([r["noisy"][0].flatten(), r["ref"][0].flatten()]))[0, 1])
    assert abs(corr) < 0.03 and float(r["ref"][0].std()) == pytest.approx(0.1, rel=0.05)   # the reference image is a second draw


@pytest.mark.gpu
def test_noise2void_manipulation_geometry(contract):
    torch.manual_seed(0)
    u8 = torch.randint(0, 256, (5, 3, 64, 64), dtype=torch.uint8)
    plain = _run(u8, 0, True, 25 / 255.0, 25 / 255.0, seed=9, offset=2, ref=True)
    o = _run(u8, 0, True, 25 / 255.0, 25 / 255.0, seed=9, offset=2, ref=True, n2v=True)
    want = contract["n2v"]
    assert list(o["coords"].shape[1:]) == want["coords_shape"] and o["coords"].dtype == torch.int64
    assert int(o["coords"].min()) >= want["coord_min"] and int(o["coords"].max()) <= want["coord_max"]
    assert torch.equal(o["ref"], plain["ref"]) and torch.equal(o["clean"], plain["clean"])      # only the input is manipulated
    total_changed = 0
    for b in range(5):
        cs = o["coords"][b].tolist()
        assert len({(x // 8, y // 8) for x, y in cs}) == 64                                  # one coordinate per 8 x 8 box
        assert [(x // 8) * 8 + (y // 8) for x, y in cs] == list(range(64))                   # in the reference's box order
        diff = (o["noisy"][b] != plain["noisy"][b]).any(0)
        ys, xs = diff.nonzero(as_tuple=True)
        assert set(zip(xs.tolist(), ys.tolist())) <= {(x, y) for x, y in cs}                 # nothing else changed
        total_changed += int(diff.sum())
        for x, y in cs:        # image[:, y, x] = image[:, ry, rx], (rx, ry) from [min(c - 2, 0), min(c + 2, 63)) without c, negatives wrap
            cand_x = sorted({v % 64 for v in range(min(x - 2, 0), min(x + 2, 63)) if v != x})
            cand_y = sorted({v % 64 for v in range(min(y - 2, 0), min(y + 2, 63)) if v != y})
            src = (plain["noisy"][b][:, cand_y][:, :, cand_x] == o["noisy"][b][:, y, x].view(3, 1, 1)).all(0)
            assert bool(src.any()), (b, x, y)
    assert total_changed >= 5 * 56                                                           # (a copied value may coincide with the old one)
    # the interior window really is [0, c + 2): far-away sources occur
    far = 0
    for b in range(5):
        for x, y in o["coords"][b].tolist():
            if x >= 16 and y >= 16:
                near = (plain["noisy"][b][:, max(y - 2, 0):y + 3, max(x - 2, 0):x + 3] == o["noisy"][b][:, y, x].view(3, 1, 1)).all(0)
                far += int(not bool(near.any()))
    assert far > 20


@pytest.mark.gpu
def test_device_patch_stream_uses_the_kernel_and_keeps_the_batch_format():
    """DevicePatchStream on a GPU == the CPU preparation in structure (keys, shapes, dtypes), for every algorithm"""
    from ssdn.datasets import DevicePatchStream, NoisyDataset
    from ssdn.params import NoiseAlgorithm
    MD = NoisyDataset.Metadata
    u8 = torch.randint(0, 256, (6, 3, 64, 64), dtype=torch.uint8)
    idx = torch.arange(6)
    for algo, style in ((NoiseAlgorithm.SELFSUPERVISED_DENOISING, "gauss25"), (NoiseAlgorithm.NOISE_TO_VOID, "gauss5_50"),
                        (NoiseAlgorithm.NOISE_TO_NOISE, "poisson30"), (NoiseAlgorithm.NOISE_TO_CLEAN, "poisson5_50"),
                        (NoiseAlgorithm.SELFSUPERVISED_DENOISING_MEAN_ONLY, "gauss25_nc")):
        nd = NoisyDataset(None, style, algo, pad_uniform=False, pad_multiple=32, square=True, training_mode=True)
        gpu = DevicePatchStream(None, nd, "cuda:0", seed=1).prepare(u8.pin_memory(), idx)
        cpu = DevicePatchStream(None, nd, "cpu", seed=1).prepare(u8, idx)
        assert gpu[0].shape == cpu[0].shape and gpu[0].dtype == cpu[0].dtype and gpu[0].device.type == "cuda"
        assert gpu[1].shape == cpu[1].shape and gpu[1].dtype == cpu[1].dtype
        assert set(gpu[2]) == set(cpu[2])
        for k in cpu[2]:
            assert tuple(gpu[2][k].shape) == tuple(cpu[2][k].shape) and gpu[2][k].dtype == cpu[2][k].dtype, (algo, k)
        assert torch.equal(gpu[2][MD.CLEAN].cpu(), cpu[2][MD.CLEAN])
        assert torch.equal(gpu[2][MD.IMAGE_SHAPE].cpu(), cpu[2][MD.IMAGE_SHAPE])
        if algo == NoiseAlgorithm.NOISE_TO_CLEAN:
            assert torch.equal(gpu[1].cpu(), cpu[2][MD.CLEAN])
        if algo == NoiseAlgorithm.SELFSUPERVISED_DENOISING_MEAN_ONLY:
            assert gpu[1] is gpu[0]
        # two consecutive minibatches never share a noise realisation
        again = DevicePatchStream(None, nd, "cuda:0", seed=1)
        first, second = again.prepare(u8.pin_memory(), idx)[0], again.prepare(u8.pin_memory(), idx)[0]
        assert torch.equal(first, gpu[0]) and not torch.equal(first, second)


@pytest.mark.gpu
def test_device_patch_stream_state_continues_the_noise_stream():
    """the `.training` state of the stream (Philox key + minibatch counter): a stream restored after minibatch 1 produces minibatches
    2 and 3 of the original run bit for bit (the counter was not checkpointed)"""
    from ssdn.datasets import DevicePatchStream, NoisyDataset
    from ssdn.params import NoiseAlgorithm
    nd = NoisyDataset(None, "gauss5_50", NoiseAlgorithm.NOISE_TO_VOID, pad_uniform=False, pad_multiple=32, square=True, training_mode=True)
    u8 = torch.randint(0, 256, (4, 3, 64, 64), dtype=torch.uint8)
    idx = torch.arange(4)
    a = DevicePatchStream(None, nd, "cuda:0", seed=11, rank=2)
    first = a.prepare(u8.pin_memory(), idx)[0].clone()
    sd = a.state_dict()
    assert sd == {"seed": 11, "calls": 1}
    want = [a.prepare(u8.pin_memory(), idx)[0].clone() for _ in range(2)]
    b = DevicePatchStream(None, nd, "cuda:0", seed=999, rank=2)
    b.load_state_dict(sd)
    got = [b.prepare(u8.pin_memory(), idx)[0].clone() for _ in range(2)]
    assert all(torch.equal(w, g) for w, g in zip(want, got)) and not torch.equal(first, got[0])
    other_rank = DevicePatchStream(None, nd, "cuda:0", seed=999, rank=0)
    other_rank.load_state_dict(sd, rank=3)
    assert not torch.equal(other_rank.prepare(u8.pin_memory(), idx)[0], want[0])       # ranks never share a stream

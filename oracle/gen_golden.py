"""TEST INFRASTRUCTURE -- generate tests/golden/*.npz from the LIVE reference.

Run in the build container only (needs /root/reference):   python oracle/gen_golden.py
The reference ships no golden vectors for this path, so the pins are
made here by importing the reference itself (oracle/ref_shim.py) and driving ITS
NoiseNetwork / Denoiser / optimizer on deterministic inputs (oracle/restate.hash_tensor,
restate.make_params -- regenerable anywhere, so only outputs are stored).
Fixtures are data (inputs' seeds + expected outputs); no reference source is copied.
"""
import json
import os
import pickle
import pickletools
import sys
import io

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
import restate as R  # noqa: E402

OUT = os.environ.get("SSDN_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")      # (override: tests/test_oracle_golden.py regenerates into a scratch directory)
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()})
    print("wrote", name, {k: tuple(np.asarray(v.detach() if isinstance(v, torch.Tensor) else v).shape) for k, v in arrs.items()})


def RefNoisy_null():
    return torch.zeros(0)


def main():
    ref = ref_shim.import_reference()
    with ref_shim.reference_modules(ref):
        import ssdn
        from ssdn.models import NoiseNetwork
        from ssdn.models.noise_network import ShiftConv2d
        from ssdn.models.utility import Shift2d
        from ssdn.denoiser import Denoiser
        from ssdn.datasets import NoisyDataset
        from ssdn.params import ConfigValue, NoiseAlgorithm, NoiseValue, PipelineOutput
        MD = NoisyDataset.Metadata

        # ---- G9 rotate ------------------------------------------------------------------
        x = torch.arange(2 * 1 * 4 * 4, dtype=torch.float32).reshape(2, 1, 4, 4)
        save("g_rotate", x=x, **{"r%d" % a: ssdn.utils.rotate(x, a) for a in (0, 90, 180, 270)})

        # ---- G1 ShiftConv2d ---------------------------------------------------------------
        for cin, cout in ((3, 5), (48, 48)):
            conv = ShiftConv2d(cin, cout, 3, stride=1, padding=1)
            w = R.hash_tensor((cout, cin, 3, 3), 11 + cin, -0.3, 0.3)
            b = R.hash_tensor((cout,), 12 + cin, -0.1, 0.1)
            with torch.no_grad():
                conv.weight.copy_(w)
                conv.bias.copy_(b)
                x = R.hash_tensor((2, cin, 8, 8), 13 + cin, -1, 1)
                save("g_shiftconv_c%d" % cin, out=conv(x))

        # ---- G2 shifted max pool (negative inputs so the zero row matters) ----------------
        import torch.nn as nn
        x = R.hash_tensor((2, 3, 8, 8), 21, -1.0, 0.5)
        save("g_pool", shifted=nn.Sequential(Shift2d((1, 0)), nn.MaxPool2d(2))(x), plain=nn.MaxPool2d(2)(x))

        # ---- G3 NoiseNetwork.forward ---------------------------------------------------------
        for tag, cin, cout, bs, zero in (("bs_rgb", 3, 9, True, False), ("bs_mono", 1, 2, True, False),
                                         ("plain_rgb", 3, 3, False, False), ("sigma", 3, 1, False, False)):
            net = NoiseNetwork(cin, cout, blindspot=bs, zero_output_weights=zero)
            p = R.make_params(cin, cout, bs, seed=3)
            net.load_state_dict(R.reference_state_dict(p))
            x = R.hash_tensor((2, cin, 32, 32), 31, 0, 1).requires_grad_(True)
            y = net(x)
            extra = {}
            if bs:
                # blind-spot property: d out[0,:,13,17] / d x -- exactly zero at (13,17)
                g, = torch.autograd.grad(y[0, :, 13, 17].sum(), x)
                extra["gin"] = g[0]
            save("g_net_" + tag, out=y, keys=np.array(list(net.state_dict().keys())), **extra)
        # one 64x64 case (config-2 patch size)
        net = NoiseNetwork(3, 9, blindspot=True)
        net.load_state_dict(R.reference_state_dict(R.make_params(3, 9, True, seed=4)))
        with torch.no_grad():
            save("g_net_bs_rgb64", out=net(R.hash_tensor((1, 3, 64, 64), 32, 0, 1)))

        # ---- G4 _ssdn_pipeline head: outputs + gradients --------------------------------------
        def make_cfg(alg, style, mode, ch):
            cfg = ssdn.cfg.base()
            cfg[ConfigValue.ALGORITHM] = NoiseAlgorithm(alg)
            cfg[ConfigValue.NOISE_STYLE] = style
            cfg[ConfigValue.NOISE_VALUE] = NoiseValue(mode)
            cfg[ConfigValue.IMAGE_CHANNELS] = ch
            ssdn.cfg.infer(cfg, model_only=True)
            return cfg

        class FakeNet(torch.nn.Module):
            def __init__(self, out):
                super().__init__()
                self.out = out

            def forward(self, x):
                return self.out

        B, H = 2, 8
        for style, npar in (("gauss25", 25 / 255.0), ("poisson30", 30.0)):
            for mode in ("known", "const", "var"):
                for ch in (1, 3):
                    d = Denoiser(make_cfg("ssdn", style, mode, ch), device="cpu")
                    ncomp = ch + ch * (ch + 1) // 2
                    net_out = R.hash_tensor((B, ncomp, H, H), 41 + ch, -0.4, 0.6)
                    net_out[:, :ch] = R.hash_tensor((B, ch, H, H), 42, 0.05, 0.95)
                    net_out = net_out.clone().requires_grad_(True)
                    noisy = R.hash_tensor((B, ch, H, H), 43, 0.0, 1.0)
                    d.models[Denoiser.MODEL] = FakeNet(net_out)
                    raw = None
                    if mode == "var":
                        raw = R.hash_tensor((B, 1, H, H), 44, 1.0, 3.0).requires_grad_(True)
                        d.models[Denoiser.SIGMA_ESTIMATOR] = FakeNet(raw)
                    if mode == "const":
                        with torch.no_grad():
                            d.l_params[Denoiser.ESTIMATED_SIGMA].fill_(1.7)
                        raw = d.l_params[Denoiser.ESTIMATED_SIGMA]
                    meta = {MD.INPUT_NOISE_VALUES: torch.full((B, 1, 1, 1), npar), MD.IMAGE_SHAPE: None}
                    o = d.run_pipeline([noisy, None, meta])
                    o[PipelineOutput.LOSS].mean().backward()
                    arrs = dict(loss=o[PipelineOutput.LOSS], out=o[PipelineOutput.IMG_DENOISED],
                                out_mu=o[PipelineOutput.IMG_MU], noise_std=o[PipelineOutput.NOISE_STD_DEV],
                                model_std=o[PipelineOutput.MODEL_STD_DEV], g_net_out=net_out.grad)
                    if raw is not None:
                        arrs["g_raw"] = raw.grad
                    save("g_head_%s_%s_c%d" % (style, mode, ch), **arrs)

        # ---- G5 MSE / masked MSE -------------------------------------------------------------
        out = R.hash_tensor((3, 3, 16, 16), 51, 0, 1).requires_grad_(True)
        tgt = R.hash_tensor((3, 3, 16, 16), 52, 0, 1)
        d = Denoiser(make_cfg("n2c", "gauss25", "known", 3), device="cpu")
        d.models[Denoiser.MODEL] = FakeNet(out)
        o = d.run_pipeline([out.detach(), tgt, {}])
        o[PipelineOutput.LOSS].mean().backward()
        save("g_mse", loss=o[PipelineOutput.LOSS], g_out=out.grad)
        out = R.hash_tensor((3, 3, 16, 16), 51, 0, 1).requires_grad_(True)
        coords = (R.hash_tensor((3, 64, 2), 53, 0, 16)).long()
        d = Denoiser(make_cfg("n2v", "gauss25", "known", 3), device="cpu")
        d.models[Denoiser.MODEL] = FakeNet(out)
        o = d.run_pipeline([out.detach(), tgt, {MD.MASK_COORDS: coords}])
        o[PipelineOutput.LOSS].mean().backward()
        save("g_maskmse", loss=o[PipelineOutput.LOSS], g_out=out.grad, coords=coords)

        # ---- G6/G7 full Denoiser: parameter grads after one backward + 3 Adam steps --------------
        from ssdn.train import DenoiserTrainer  # noqa: F401  (import check only)
        for tag, alg, style, mode, ch, P in (("ssdn_known_rgb", "ssdn", "gauss25", "known", 3, 32),
                                             ("ssdn_var_rgb", "ssdn", "gauss25", "var", 3, 32),
                                             ("ssdn_const_poisson_rgb", "ssdn", "poisson30", "const", 3, 32),
                                             ("ssdn_known_mono", "ssdn", "gauss25", "known", 1, 32),
                                             ("n2c_mono", "n2c", "gauss25", "known", 1, 32),
                                             ("n2v_rgb", "n2v", "gauss25", "known", 3, 32)):
            cfg = make_cfg(alg, style, mode, ch)
            d = Denoiser(cfg, device="cpu")
            bs = cfg[ConfigValue.BLINDSPOT]
            cout = ch + ch * (ch + 1) // 2 if alg == "ssdn" else ch
            d._models[Denoiser.MODEL].load_state_dict(R.reference_state_dict(R.make_params(ch, cout, bs, seed=5)))
            if mode == "var" and alg == "ssdn":
                # non-zero last layer so the sigma path carries signal
                d._models[Denoiser.SIGMA_ESTIMATOR].load_state_dict(R.reference_state_dict(R.make_params(ch, 1, False, seed=6)))
            Bn = 2
            clean = R.hash_tensor((Bn, ch, P, P), 61, 0, 1)
            npar = 25 / 255.0 if style.startswith("gauss") else 30.0
            noisy = torch.clamp(clean + (R.hash_tensor((Bn, ch, P, P), 62, -1, 1)) * 0.17, 0, 1)
            refimg = clean if alg != "n2v" else torch.clamp(clean + R.hash_tensor((Bn, ch, P, P), 63, -1, 1) * 0.17, 0, 1)
            coords = R.hash_tensor((Bn, 64, 2), 64, 0, P).long()      # 64 coordinates per patch, like the reference sampler (n2v_ups.py:65-88)
            meta = {MD.INPUT_NOISE_VALUES: torch.full((Bn, 1, 1, 1), npar), MD.IMAGE_SHAPE: None, MD.CLEAN: clean}
            if alg == "n2v":
                meta[MD.MASK_COORDS] = coords
            opt = torch.optim.Adam(d.parameters(), betas=[0.9, 0.99])   # train.py:107
            names = [n for n, _ in d.named_parameters()]
            arrs = {"names": np.array(names)}
            N_IT = 1000
            lrs = []
            for it in range(3):
                # train.py:261-282 -- LR by images seen, fractions swapped at the call site
                lr = ssdn.utils.compute_ramped_lrate((it + 1) * 40, N_IT, cfg[ConfigValue.LR_RAMPDOWN_FRACTION],
                                                     cfg[ConfigValue.LR_RAMPUP_FRACTION], cfg[ConfigValue.LEARNING_RATE])
                lrs.append(lr)
                for g in opt.param_groups:
                    g["lr"] = lr
                opt.zero_grad()
                o = d.run_pipeline([noisy, refimg, meta])
                torch.mean(o[PipelineOutput.LOSS]).backward()
                if it == 0:
                    arrs["loss0"] = o[PipelineOutput.LOSS]
                    arrs["out0"] = o[PipelineOutput.IMG_DENOISED]
                    for n, prm in d.named_parameters():
                        arrs["grad/" + n] = prm.grad.clone()
                opt.step()
                arrs["loss_it%d" % it] = o[PipelineOutput.LOSS]
            arrs["lrs"] = np.array(lrs, dtype=np.float64)
            for n, prm in d.named_parameters():
                flat = prm.detach().reshape(-1)
                arrs["sum/" + n] = flat.double().sum()
                arrs["head/" + n] = flat[:8].clone()
            # keep the file small: only norm + first 16 entries of each gradient
            small = {}
            for k, v in arrs.items():
                if k.startswith("grad/"):
                    f = v.reshape(-1)
                    small["gnorm/" + k[5:]] = f.double().norm()
                    small["ghead/" + k[5:]] = f[:16].clone()
                else:
                    small[k] = v
            save("g_train_" + tag, **small)

        # ---- G8 LR ramp table --------------------------------------------------------------------
        N = 2000000
        iters = np.array([0, 1, 4, 1000, 100000, 199999, 200000, 200001, 1000000, 1399999, 1400000, 1400001, 1700000, 1999996, 2000000])
        lr_direct = [ssdn.utils.compute_ramped_lrate(int(i), N, 0.1, 0.3, 3e-4) for i in iters]
        save("g_lr", iters=iters, lr=np.array(lr_direct, dtype=np.float64))

        # ---- G11 PSNR ------------------------------------------------------------------------------
        a = R.hash_tensor((3, 3, 16, 16), 71, 0, 1)
        b = torch.clamp(a + R.hash_tensor((3, 3, 16, 16), 72, -0.1, 0.1), 0, 1)
        save("g_psnr", psnr=ssdn.utils.calculate_psnr(a, b))

        # ---- G10 checkpoint contract: key lists + pickled globals ---------------------------------
        ck = {}
        for tag, alg, mode in (("ssdn_known", "ssdn", "known"), ("ssdn_var", "ssdn", "var"), ("ssdn_const", "ssdn", "const"), ("n2c", "n2c", "known")):
            d = Denoiser(make_cfg(alg, "gauss25", mode, 3), device="cpu")
            sd = d.state_dict()
            buf = io.BytesIO()
            torch.save(sd, buf)
            # list the globals the pickle references
            import zipfile
            zf = zipfile.ZipFile(io.BytesIO(buf.getvalue()))
            pk = [n for n in zf.namelist() if n.endswith("data.pkl")][0]
            globs = sorted({"%s.%s" % (a.split(" ")[0], a.split(" ")[1]) for op, a, _ in pickletools.genops(zf.read(pk))
                            if op.name == "GLOBAL"})
            ck[tag] = {"keys": [k for k in sd.keys()], "shapes": {k: list(v.shape) for k, v in sd.items() if hasattr(v, "shape")},
                       "globals": globs, "config_name": d.config_name(),
                       "cfg": {k.name: (v.name if hasattr(v, "name") else v) for k, v in sd["cfg"].items()}}
        # ---- G10b `.training` contract: a state_dict written by the reference's OWN DenoiserTrainer -------------------
        from ssdn.train import DenoiserTrainer
        from ssdn.datasets import FixedLengthSampler, NoisyDataset as RefNoisy
        from ssdn.params import StateValue, HistoryValue
        tcfg = make_cfg("ssdn", "gauss25", "known", 3)
        tr = DenoiserTrainer(tcfg, runs_dir="/tmp/ssdn_ref_runs")
        tr.new_target()
        tr.train_sampler = FixedLengthSampler(list(range(7)), num_samples=20, shuffled=True)
        _ = iter(tr.train_sampler)
        # two real optimiser steps so that the Adam state exists
        Bn, P = 2, 32
        noisy = torch.clamp(R.hash_tensor((Bn, 3, P, P), 61, 0, 1) + R.hash_tensor((Bn, 3, P, P), 62, -1, 1) * 0.17, 0, 1)
        meta = {MD.INPUT_NOISE_VALUES: torch.full((Bn, 1, 1, 1), 25 / 255.0), MD.CLEAN: noisy, MD.IMAGE_SHAPE: None}
        for _i in range(2):
            opt = tr.optimizer
            opt.zero_grad()
            o = tr.denoiser.run_pipeline([noisy, RefNoisy_null(), meta]) if False else tr.denoiser.run_pipeline([noisy, noisy, meta])
            torch.mean(o[PipelineOutput.LOSS]).backward()
            opt.step()
            tr.state[StateValue.HISTORY][HistoryValue.TRAIN]["loss"] += o[PipelineOutput.LOSS].detach()
            tr.state[StateValue.HISTORY][HistoryValue.TRAIN]["n"] += Bn
            tr.state[StateValue.ITERATION] += Bn
        tr.state[StateValue.HISTORY][HistoryValue.TIMINGS]["total"].update()
        tsd = tr.state_dict()
        buf = io.BytesIO()
        torch.save(tsd, buf)
        import zipfile
        zf = zipfile.ZipFile(io.BytesIO(buf.getvalue()))
        pk = [n for n in zf.namelist() if n.endswith("data.pkl")][0]
        tglobs = sorted({"%s.%s" % (a.split(" ")[0], a.split(" ")[1]) for op, a, _ in pickletools.genops(zf.read(pk))
                         if op.name == "GLOBAL"})
        osd = tsd["optimizer"]
        ck["training_file"] = {
            "keys": sorted(tsd.keys()), "globals": tglobs,
            "state_keys": sorted(k.name for k in tsd["state"].keys()),
            "history_keys": sorted(k.name for k in tsd["state"][StateValue.HISTORY].keys()),
            "train_order_iter_keys": sorted(tsd["train_order_iter"].keys()),
            "train_order_index": int(tsd["train_order_iter"]["index"]),
            "optimizer_keys": sorted(osd.keys()),
            "optimizer_group_keys": sorted(osd["param_groups"][0].keys()),
            "optimizer_state_entry_keys": sorted(osd["state"][0].keys()),
            "optimizer_n_params": len(osd["param_groups"][0]["params"]),
            "optimizer_param_shapes": [list(osd["state"][i]["exp_avg"].shape) for i in range(len(osd["state"]))],
            "optimizer_betas": list(osd["param_groups"][0]["betas"]),
            "run_dir": tr.run_dir, "config_name": tr.config_name(),
        }
        # a reference-written `.training` file as a data fixture.  To keep it small every large tensor is zeroed first (zeros
        # deflate to nothing); the 9 output biases and their Adam moments keep recognisable values for a round-trip check.
        import gzip
        with torch.no_grad():
            for k, v in tsd["denoiser"].items():
                if torch.is_tensor(v) and v.numel() > 16:
                    v.zero_()
            tsd["denoiser"]["models.denoiser_model.module.output_conv.bias"].copy_(torch.arange(9.0) * 0.25 - 1)
            for i, st in osd["state"].items():
                if st["exp_avg"].numel() > 16:
                    st["exp_avg"].zero_()
                    st["exp_avg_sq"].zero_()
        buf2 = io.BytesIO()
        torch.save(tsd, buf2)
        with gzip.open(os.path.join(OUT, "g_training_file.training.gz"), "wb", compresslevel=9) as f:
            f.write(buf2.getvalue())

        # ---- G12 data layer: padding shapes, style parsing, n2v coordinate statistics --------------------------------
        import ssdn.utils.noise as ref_noise
        import ssdn.utils.n2v_ups as ref_ups
        dl = {}
        class _Imgs(torch.utils.data.Dataset):
            def __init__(self, shapes): self.shapes = shapes
            def __len__(self): return len(self.shapes)
            def __getitem__(self, i): return (R.hash_tensor(self.shapes[i], 900 + i, 0, 1), i)
        for tag, shapes, kw in (("kodak_like", [(3, 48, 72), (3, 72, 48)], dict(pad_uniform=True, pad_multiple=32, square=True)),
                                ("bsd_like", [(3, 33, 50), (3, 50, 33)], dict(pad_uniform=True, pad_multiple=32, square=False)),
                                ("train_like", [(3, 64, 64)], dict(pad_uniform=False, pad_multiple=32, square=True))):
            nd = RefNoisy(_Imgs(shapes), "gauss25", NoiseAlgorithm.SELFSUPERVISED_DENOISING, training_mode=False, **kw)
            inp, ref_, md_ = nd[0]
            dl[tag] = {"out_shape": list(inp.shape), "image_shape": [int(v) for v in md_[MD.IMAGE_SHAPE]],
                       "clean_padded_sum": float(md_[MD.CLEAN].double().sum()),
                       "noise_values_shape": list(md_[MD.INPUT_NOISE_VALUES].shape), "ref_numel": int(ref_.numel())}
        torch.manual_seed(7)
        styles = {}
        for st in ("gauss25", "gauss5_50", "gauss0.1", "gauss25_nc", "poisson30", "poisson5_50"):
            x = torch.full((4, 3, 64, 64), 0.5)
            y, coeff = ref_noise.add_style(x, st)
            styles[st] = {"coeff": (coeff.reshape(-1).tolist() if torch.is_tensor(coeff) else float(coeff)),
                          "coeff_shape": (list(coeff.shape) if torch.is_tensor(coeff) else []),
                          "mean": float(y.mean()), "std": float((y - 0.5).std()), "min": float(y.min()), "max": float(y.max())}
        dl["styles"] = styles
        torch.manual_seed(11)
        img = R.hash_tensor((3, 64, 64), 77, 0, 1)
        out_img, coords = ref_ups.manipulate(img, 5)
        dl["n2v"] = {"ncoords": int(coords.shape[0]), "coords_shape": list(coords.shape),
                     "changed_pixels": int(((out_img != img).any(0)).sum()),
                     "coord_max": int(coords.max()), "coord_min": int(coords.min())}
        ck["data_layer"] = dl
        with open(os.path.join(OUT, "g_ckpt_contract.json"), "w") as f:
            json.dump(ck, f, indent=1, sort_keys=True)
        print("wrote g_ckpt_contract.json")


if __name__ == "__main__":
    main()

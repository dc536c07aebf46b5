"""N > 1 through the TRAINER on CPU: two gloo processes run `DenoiserTrainer.train()` over an HDF5 patch stream (SURVEY.md
section 8e / 8f N2; reference: train.py:127-190 + datasets/hdf5.py + datasets/sampler.py:86-111) with a stub in the place of
the HIP step (`Denoiser.train_step` raises on CPU by design; the step itself is covered by the GPU tests).  Checked:

  * the trainer joins the process group by itself (nothing called init_process_group),
  * rank shards of every global minibatch are disjoint and their union is the single-process sampling order,
  * a final partial global minibatch is processed un-sharded by every rank (no rank runs out of steps early),
  * ITERATION / learning-rate sequence equal the single-process run's, replicas end with identical weights equal to the
    single-process result although the ranks were seeded differently (initial weights come from rank 0),
  * DataLoader workers (fork) read the HDF5 file concurrently without corrupting a single patch.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import ssdn
from ssdn.datasets import HDF5Dataset, NoisyDataset, h5lite
from ssdn.params import ConfigValue, NoiseAlgorithm, NoiseValue, PipelineOutput, StateValue

ITERS, GB, P, NIMG = 44, 8, 32, 23           # 44 = 5 global minibatches of 8 + a tail of 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _images():
    rng = np.random.RandomState(7)
    # every pixel of image i has value i in channel 0: a returned patch tells which image it was cut from
    imgs = []
    for i in range(NIMG):
        im = rng.randint(0, 256, size=(3, 40 + i % 5, 37 + i % 3), dtype=np.uint8)
        im[0] = i
        imgs.append(im)
    return imgs


def _cfg(path, batch):
    cfg = ssdn.cfg.base()
    cfg[ConfigValue.ALGORITHM] = NoiseAlgorithm.SELFSUPERVISED_DENOISING
    cfg[ConfigValue.NOISE_STYLE] = "gauss25"
    cfg[ConfigValue.NOISE_VALUE] = NoiseValue.KNOWN
    cfg[ConfigValue.TRAIN_ITERATIONS] = ITERS
    cfg[ConfigValue.TRAIN_MINIBATCH_SIZE] = batch
    cfg[ConfigValue.TRAIN_PATCH_SIZE] = P
    cfg[ConfigValue.TRAIN_DATA_PATH] = path
    cfg[ConfigValue.DATALOADER_WORKERS] = 2
    cfg[ConfigValue.PRINT_INTERVAL] = 16
    cfg[ConfigValue.EVAL_INTERVAL] = cfg[ConfigValue.SNAPSHOT_INTERVAL] = 10 ** 9
    return cfg


def _g(idx, n):
    return torch.sin(torch.arange(n, dtype=torch.float64) * 0.013 * (idx + 1)).to(torch.float32)


def _make_stub(cfg):
    from ssdn.denoiser import Denoiser
    from ssdn.hip import dp

    class StubDenoiser(Denoiser):
        """train_step = a fake gradient that depends only on WHICH images the batch holds, exchanged and applied through the
        same driver (`dp.exchange_step`) the HIP step uses"""

        def __init__(self, cfg):
            super().__init__(cfg, device="cpu")
            self.seen, self.lrs, self.bad = [], [], 0

        def train_step(self, data, lr, exchange=None):
            MD = NoisyDataset.Metadata
            inp, meta = data[NoisyDataset.INPUT], data[NoisyDataset.METADATA]
            idx = [int(i) for i in meta[MD.INDEXES]]
            clean = meta[MD.CLEAN]
            for b, i in enumerate(idx):          # channel 0 of image i is i / 255 everywhere: the patch belongs to the index
                if not torch.all((clean[b, 0] * 255).round() == i):
                    self.bad += 1
            self.seen.append(idx)
            self.lrs.append(lr)
            n = self.flat.numel()

            def bwd(ex):
                self.flat_grad.copy_(torch.stack([_g(i, n) for i in idx]).mean(0))
            scale = dp.exchange_step(bwd, self.flat_grad, exchange)
            self.flat.sub_(0.5 * scale * self.flat_grad)          # lr-free update: lr(0) = 0 would hide everything
            B = inp.shape[0]
            return {PipelineOutput.INPUTS: data, PipelineOutput.LOSS: torch.ones(B, 1), PipelineOutput.IMG_DENOISED: clean.clone(),
                    PipelineOutput.IMG_MU: inp.clone(), PipelineOutput.NOISE_STD_DEV: torch.ones(B, 1, 1), PipelineOutput.MODEL_STD_DEV: torch.ones(B, P, P)}
    return StubDenoiser(cfg)


def _worker(rank, world, port, path, runs, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    torch.manual_seed(100 + rank)                       # deliberately different: replicas must still agree
    import torch.distributed as dist
    from ssdn.train import DenoiserTrainer
    tr = DenoiserTrainer(_cfg(path, GB), runs_dir=runs)
    assert dist.is_initialized() and dist.get_world_size() == world and (tr.rank, tr.world) == (rank, world)
    tr.denoiser = _make_stub(tr.cfg)
    tr.init_state()
    tr.train()
    d = tr.denoiser
    out.put((rank, d.seen, d.lrs, d.bad, d.flat.numpy().copy(), tr.state[StateValue.ITERATION], len(tr._shard.counts),
             os.path.isdir(os.path.join(tr.run_dir_path, "training")) if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_trainer_over_an_hdf5_stream(tmp_path):
    path = str(tmp_path / "train_set.h5")
    h5lite.write_dataset_file(path, _images())
    assert os.path.getsize(path) > 8192
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, path, str(tmp_path / "runs"), out)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = out.get(timeout=300)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    # the single-process order: DenoiserTrainer.train_data draws it under manual_seed(1234 + ITERATION) when world > 1
    from ssdn.datasets import FixedLengthSampler
    torch.manual_seed(1234)
    order = list(FixedLengthSampler(list(range(NIMG)), num_samples=ITERS, shuffled=True).sampler())
    assert len(order) == ITERS
    nfull = ITERS // GB
    per = GB // world
    for rank in range(world):
        seen = res[rank][0]
        assert len(seen) == nfull + 1                                      # every rank takes the same number of steps
        for k in range(nfull):
            assert seen[k] == order[k * GB + rank * per: k * GB + (rank + 1) * per]
        assert seen[nfull] == order[nfull * GB:]                           # the tail: un-sharded, on every rank
        assert res[rank][2] == 0                                           # no patch came from the wrong image / corrupted bytes
        assert res[rank][4] == ITERS and res[rank][5] == 0                 # ITERATION = images consumed by the job
    for k in range(nfull):                                                 # disjoint shards whose union is the global minibatch
        assert sorted(res[0][0][k] + res[1][0][k]) == sorted(order[k * GB:(k + 1) * GB])
    # learning rates: the single-process schedule at ITERATION 0, 8, 16, ...
    from ssdn.utils.utils import compute_ramped_lrate
    want_lr = [compute_ramped_lrate(k * GB, ITERS, 0.1, 0.3, 3e-4) for k in range(nfull + 1)]
    assert res[0][1] == pytest.approx(want_lr) and res[1][1] == pytest.approx(want_lr)
    # weights: both replicas identical, equal to the single-process run over the same order with batch 8 from rank 0's init
    assert np.array_equal(res[0][3], res[1][3])
    torch.manual_seed(100)
    rcfg = _cfg(path, GB)
    ssdn.cfg.infer(rcfg)
    ref = _make_stub(rcfg)
    n = ref.flat.numel()
    for k in range(nfull + 1):
        ids = order[k * GB:(k + 1) * GB]
        ref.flat.sub_(0.5 * torch.stack([_g(i, n) for i in ids]).mean(0))
    assert np.allclose(res[0][3], ref.flat.numpy(), rtol=0, atol=2e-6)
    assert res[0][6] is True                                               # rank 0 owns the run directory


def test_hdf5_dataset_with_forked_workers_returns_exact_bytes(tmp_path):
    """a handle inherited through fork shared one file offset between workers (seek + read raced: exceptions and
    silently wrong bytes).  400 images, 4 forked workers, several epochs: every image must come back bit-exact."""
    from torch.utils.data import DataLoader
    rng = np.random.RandomState(3)
    imgs = [rng.randint(0, 256, size=(3, 9 + i % 7, 11 + i % 5), dtype=np.uint8) for i in range(400)]
    path = str(tmp_path / "many.h5")
    h5lite.write_dataset_file(path, imgs)
    assert os.path.getsize(path) > 8192
    ds = HDF5Dataset(path, channels=3)
    assert ds._h is None                                                   # the constructor leaves no handle behind
    _ = ds[5]                                                              # ... and a handle opened in the PARENT is not inherited
    loader = DataLoader(ds, batch_size=None, shuffle=True, num_workers=4, multiprocessing_context="fork")
    for _epoch in range(4):
        n = 0
        for t, idx in loader:
            want = torch.from_numpy(imgs[int(idx)]).float().div(255).permute(0, 2, 1)     # the reference's swapped H / W
            assert torch.equal(t, want), int(idx)
            n += 1
        assert n == 400

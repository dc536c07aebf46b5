#!/usr/bin/env python
"""bench.py -- training patches/sec of the ssdn hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): ssdn gauss25 sigma_known, 64x64 RGB
patches, batch 32 PER GPU (weak scaling), blind-spot U-Net + posterior head + SSDN loss + backward + gradient all-reduce
+ fused Adam.  Synthetic data (clean ~ U[0,1), clipped N(0,(25/255)^2) noise), reference-style random-init weights;
batches are resident in HBM when the timed region starts.  One "step" = one optimisation step on one minibatch.

Prints ONE JSON line (rank 0) with `value` = whole-job patches/s, plus
  roofline:     the dominant kernel (k_conv<3>, every 96-output-channel 3x3/1x1 convolution in forward and data-gradient
                role), timed with HIP events on the launch stream inside the timed region (ssdn_profile_*), against the
                dense fp16 MFMA peak of /opt/skills/guides/MI355X_MICROARCH.md;
  cpu_baseline: the CPU oracle (oracle/restate.py, torch-CPU fp32, the operator family the reference runs on) timed on
                this box's host cores on a bounded sample of the same workload (N=1, rank 0 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "selfsupervised-denoising_amd"), os.path.join(ROOT, "oracle")]

import torch  # noqa: E402

MFMA_FP16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
TRAIN_GFLOP_PER_PATCH = 30.7           # 3 x 10.237 GFLOP (algorithmic, blind-spot RGB 64x64)


def make_cfg():
    import ssdn
    from ssdn.params import ConfigValue, NoiseAlgorithm, NoiseValue
    cfg = ssdn.cfg.base()
    cfg[ConfigValue.ALGORITHM] = NoiseAlgorithm.SELFSUPERVISED_DENOISING
    cfg[ConfigValue.NOISE_STYLE] = "gauss25"
    cfg[ConfigValue.NOISE_VALUE] = NoiseValue.KNOWN
    cfg[ConfigValue.IMAGE_CHANNELS] = 3
    cfg[ConfigValue.TRAIN_MINIBATCH_SIZE] = 32
    ssdn.cfg.infer(cfg, model_only=True)
    return cfg


def synth_batch(B, P, seed, device):
    g = torch.Generator().manual_seed(seed)
    clean = torch.rand((B, 3, P, P), generator=g)
    noisy = torch.clamp(clean + torch.randn((B, 3, P, P), generator=g) * (25.0 / 255.0), 0, 1)
    return noisy.to(device), clean.to(device)


def cpu_baseline(P):
    """Oracle step time on the host cores: batch 8, 1 warm-up + 2 timed steps (about 10-20 s of CPU work)."""
    import restate as R
    torch.manual_seed(0)
    threads = torch.get_num_threads()
    Bc = 8
    tr = R.CpuTrainer("ssdn", 3, "gauss25", "known", seed=0)
    noisy, clean = synth_batch(Bc, P, 1234, "cpu")
    npar = torch.full((Bc, 1, 1, 1), 25.0 / 255.0)
    tr.step(1e-6, noisy, clean, npar)
    t0 = time.perf_counter()
    n = 2
    for _ in range(n):
        tr.step(1e-6, noisy, clean, npar)
    dt = time.perf_counter() - t0
    return {"value": round(n * Bc / dt, 3), "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": "%d full training steps (fwd+loss+autograd bwd+Adam) of batch %d, 64x64 RGB ssdn gauss25 sigma_known, torch-CPU fp32, %d threads" % (n, Bc, threads)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="patches per GPU")
    ap.add_argument("--patch", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import ssdn  # noqa: F401
    from ssdn.denoiser import Denoiser
    from ssdn.datasets import NoisyDataset
    from ssdn.hip import dp, lib as L
    from ssdn.utils.utils import compute_ramped_lrate
    import torch.distributed as dist

    rank, world, local = dp.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    lib = L.load()

    torch.manual_seed(0)                      # identical replicas: same init on every rank
    d = Denoiser(make_cfg(), device=str(device))
    d.train()
    B, P = args.batch, args.patch
    MD = NoisyDataset.Metadata
    batches = []
    for i in range(4):
        noisy, clean = synth_batch(B, P, 1000 * (rank + 1) + i, device)
        meta = {MD.INPUT_NOISE_VALUES: torch.full((B, 1, 1, 1), 25.0 / 255.0, device=device), MD.CLEAN: clean}
        batches.append([noisy, clean, meta])
    allreduce = dp.GradAllReduce(world)
    N_IT = 2000000
    seen = 0

    def step(i):
        nonlocal seen
        lr = compute_ramped_lrate(seen + 200000, N_IT, 0.1, 0.3, 3e-4)   # flat part of the schedule
        d.train_step(batches[i % len(batches)], lr, allreduce if world > 1 else None)
        seen += B * world

    for i in range(args.warmup):
        step(i)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # roofline leg: HIP events around k_conv<3> launches, on the launch stream, during the timed steps.  An event pair costs
    # ~10 us of stream time (it serialises what would be back-to-back kernels), so only every 7th launch is bracketed: 45
    # launches per step -> the sample rotates over all layers; bracketing all of them cost 11 % of the step.
    prof_kind = L.PROF["conv_mt3"]
    PROF_STRIDE = 7
    lib.ssdn_profile_enable(prof_kind, 64 * args.steps // PROF_STRIDE + 64)
    lib.ssdn_profile_set_stride(prof_kind, PROF_STRIDE)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms, cnt, fl, by = C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
    L.check(lib.ssdn_profile_read(prof_kind, C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by)))
    lib.ssdn_profile_enable(prof_kind, 0)

    if rank == 0:
        value = args.steps * B * world / dt
        achieved = (fl.value / 1e12) / (ms.value / 1e3) if ms.value > 0 else 0.0
        # HBM-side traffic of the same kernel family: PMC passes cannot run inside this process, so the per-launch figure is
        # the committed result of tools/pmc_traffic.sh (FETCH_SIZE x2 + WRITE_SIZE); null if absent
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_final_traffic.json")) as f:
                traffic = int(json.load(f)["hbm_bytes_per_launch"])
        except (OSError, KeyError, ValueError):
            pass
        res = {
            "metric": "training patches/sec (64x64 gauss25 SSDN)", "value": round(value, 2), "unit": "patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic (clean U[0,1), clipped gauss25 noise; random-init weights), resident in HBM",
            "config": {"workload": "ssdn gauss25 sigma_known, %dx%d RGB patches, batch %d per GPU, blind-spot U-Net fwd+bwd + posterior head + Adam" % (P, P, B),
                       "global_batch": B * world, "parallelism": "dp%d" % world,
                       "achieved_train_tflops_algorithmic": round(value * TRAIN_GFLOP_PER_PATCH / 1e3, 2)},
            "roofline": {"bound": "mfma", "kernel": "k_conv<3> (implicit-GEMM conv, fwd + dgrad roles, 96-wide output tiles)",
                         "achieved": round(achieved, 2), "peak": MFMA_FP16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / MFMA_FP16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": "HBM-side bytes per launch (rocprofv3 FETCH_SIZE*2 + WRITE_SIZE, profiles/r01_final_traffic.json)",
                         "algorithmic_bytes_per_launch": int(by.value / max(1, cnt.value)),
                         "launches": int(cnt.value), "sampling": "every %dth launch of the timed region" % PROF_STRIDE, "avg_launch_us": round(1e3 * ms.value / max(1, cnt.value), 3),
                         "kernel_time_share": round(PROF_STRIDE * ms.value / 1e3 / dt, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(P)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

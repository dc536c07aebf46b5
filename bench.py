#!/usr/bin/env python
"""bench.py -- training patches/sec of the ssdn hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Launched by `torch.distributed.run` (RANK / WORLD_SIZE / LOCAL_RANK in the environment) the
script is a rank; launched plainly with --gpus N > 1 it re-executes itself under `torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1` and relays rank 0's line.  Any failure is reported as ONE JSON line {"error": ...}.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): ssdn gauss25 sigma_known, 64x64 RGB
patches, batch 32 PER GPU (weak scaling), blind-spot U-Net + posterior head + SSDN loss + backward + gradient all-reduce
+ fused Adam.  Synthetic data (clean uint8 patches ~ U{0..255}), reference-style random-init weights.  One "step" = one
optimisation step on one minibatch, INCLUDING the batch's way to the device (SURVEY.md section 8(d) / BASELINE.md section 3: the
metric "includes H2D of the batch"): the minibatch starts every step as clean uint8 patches in pinned HOST memory (what the
DataLoader workers of the trainer deliver, 393 KB), is uploaded, and noise / rotation stack / fp16 NHWC packing happen on the
device (`ssdn.datasets.DevicePatchStream`, the trainer's default data path on a GPU).

Prints ONE JSON line (rank 0) with `value` = whole-job patches/s of that loop, plus
  value_resident / value_with_fp32_h2d: the same step with the prepared batch already in HBM / arriving as fp32 noisy + clean
                from pinned host memory (the reference's DataLoader format); side legs of >= 50 steps each, N = 1 only;
  roofline:     the dominant kernel (k_cdma<3,*>: the 96-output-channel 3x3 convolutions at 32x32 and 64x64 pixels, forward
                and data-gradient role), timed with HIP events on the launch stream inside the timed region
                (ssdn_profile_*), against the dense fp16 MFMA peak of /opt/skills/guides/MI355X_MICROARCH.md;
  families:     the other MFMA kernel families the same way, after the timed region (k_conv by block size, k_gdma, k_wgrad);
  cpu_baseline: the CPU oracle (oracle/restate.py, torch-CPU fp32, the operator family the reference runs on) timed on
                this box's host cores on a bounded sample of the same workload (N=1, rank 0 only).
"""
import argparse
import ctypes as C
import json
import re
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


def trainer_leg(B, P, steps, workers=8, warm=30):
    """patches/s of `ssdn train` itself: DenoiserTrainer.train() on an HDF5 file (h5lite.write_dataset_file: 160 images of
    3 x 375 x 500, the size class of the reference's ImageNet validation crops), `workers` DataLoader processes, metrics on."""
    import logging
    import shutil
    import tempfile
    import numpy as np
    import ssdn
    from ssdn.datasets import h5lite
    from ssdn.params import ConfigValue
    from ssdn.train import DenoiserTrainer
    tmp = tempfile.mkdtemp(prefix="ssdn_bench_")
    try:
        rng = np.random.default_rng(7)
        path = os.path.join(tmp, "train.h5")
        h5lite.write_dataset_file(path, [rng.integers(0, 256, size=(3, 375, 500), dtype=np.uint8) for _ in range(160)])
        cfg = make_cfg()
        cfg[ConfigValue.TRAIN_MINIBATCH_SIZE] = B
        cfg[ConfigValue.TRAIN_PATCH_SIZE] = P
        cfg[ConfigValue.TRAIN_DATA_PATH] = path
        cfg[ConfigValue.TEST_DATA_PATH] = None
        cfg[ConfigValue.TRAIN_ITERATIONS] = (warm + steps) * B
        cfg[ConfigValue.PRINT_INTERVAL] = 50 * B
        cfg[ConfigValue.EVAL_INTERVAL] = cfg[ConfigValue.SNAPSHOT_INTERVAL] = 10 ** 9
        cfg[ConfigValue.DATALOADER_WORKERS] = workers
        ssdn.cfg.infer(cfg)
        tr = DenoiserTrainer(cfg, runs_dir=tmp)
        tr.new_target()
        d = tr.denoiser
        inner = d.train_step
        marks = {"n": 0}

        def timed_step(*a, **kw):
            if marks["n"] == warm:
                torch.cuda.synchronize()
                marks["t0"] = time.perf_counter()
            out = inner(*a, **kw)
            marks["n"] += 1
            if marks["n"] == warm + steps:
                torch.cuda.synchronize()
                marks["t1"] = time.perf_counter()
            return out
        d.train_step = timed_step
        logging.getLogger("ssdn").setLevel(logging.WARNING)
        tr.train()
        dt = marks["t1"] - marks["t0"]
        hist = tr.state[ssdn.params.StateValue.HISTORY][ssdn.params.HistoryValue.TRAIN]
        return round(steps * B / dt, 2), {"steps": steps, "warmup_steps": warm, "dataloader_workers": workers, "ms_per_step": round(1e3 * dt / steps, 4),
                                          "images": "160 x (3, 375, 500) uint8, HDF5 (dataset_tool_h5 layout)", "metrics": "on (device accumulators, read at PRINT_INTERVAL = 50 steps)",
                                          "logged_metrics": sorted(k for k in hist if k != "n")}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def synth_batch(B, P, seed, device):
    g = torch.Generator().manual_seed(seed)
    clean = torch.rand((B, 3, P, P), generator=g)
    noisy = torch.clamp(clean + torch.randn((B, 3, P, P), generator=g) * (25.0 / 255.0), 0, 1)
    return noisy.to(device), clean.to(device)


def _physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(P):
    """the oracle's full training step (fwd + loss + autograd bwd + Adam; torch-CPU fp32, the operator
    family the reference runs on) on this box's host cores, SAME batch as the GPU workload (32), >= 1 warm-up + 5 timed steps,
    MEDIAN, thread count = the better of {physical cores, half of them} (an oversubscribed 128-thread run measured slower than
    the reference on 8 cores).  Bounded: the batch drops to 16 if a warm step takes > 5 s.  About 20-30 s of CPU work."""
    import restate as R
    import statistics
    torch.manual_seed(0)
    phys = _physical_cores()
    Bc = 32
    tr = R.CpuTrainer("ssdn", 3, "gauss25", "known", seed=0)

    def data(b):
        noisy, clean = synth_batch(b, P, 1234, "cpu")
        return noisy, clean, torch.full((b, 1, 1, 1), 25.0 / 255.0)

    def one(b, d):
        t0 = time.perf_counter()
        tr.step(1e-6, d[0], d[1], d[2])
        return time.perf_counter() - t0

    best_t, best_threads = None, phys
    d = data(Bc)
    for th in sorted({phys, max(1, phys // 2)}, reverse=True):
        torch.set_num_threads(th)
        one(Bc, d) if best_t is None else None          # first call also warms the allocator / oneDNN primitives
        t = one(Bc, d)
        if best_t is None or t < best_t:
            best_t, best_threads = t, th
    torch.set_num_threads(best_threads)
    if best_t > 5.0:
        Bc = 16
        d = data(Bc)
        one(Bc, d)
    n = 5
    times = [one(Bc, d) for _ in range(n)]
    med = statistics.median(times)
    return {"value": round(Bc / med, 3), "unit": "patches/s", "cores": best_threads, "kind": "port",
            "sample": "median of %d timed full training steps (fwd+loss+autograd bwd+Adam) after warm-up, batch %d, 64x64 RGB ssdn "
                      "gauss25 sigma_known, torch-CPU fp32, %d threads (of %d physical cores; better of {all, half}). Caveat: the port does "
                      "not scale with cores (oneDNN on 64x64 patches: 64 threads here ~1.5x the reference on 8 cores, BASELINE.md 5.15 patches/s) -- a stated "
                      "baseline, not a tuned CPU implementation"
                      % (n, Bc, best_threads, phys),
            "step_seconds": [round(t, 3) for t in times]}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a torchrun environment: run the N ranks under torch.distributed.run ourselves and relay
    rank 0's JSON line; a job that dies without one is reported as {"error": ...} (e.g. RCCL's duplicate-device error when the
    box has fewer GPUs than ranks), never as a bare traceback."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + \
        [a for a in sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    sys.stderr.write(p.stderr[-20000:])
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and ('"metric"' in ln or '"error"' in ln):
            line = ln
    if line is None:
        tail = [ln for ln in p.stderr.splitlines() if ln.strip()][-12:]
        line = json.dumps({"error": "bench.py --gpus %d: the %d-rank job exited with code %d without a result line" % (args.gpus, args.gpus, p.returncode),
                           "n_gpus": args.gpus, "stderr_tail": tail})
    print(line)
    return 0 if (p.returncode == 0 and '"error"' not in line[:12]) else 1


class _StubDenoiser:
    """SSDN_BENCH_STUB=1 (tests/test_bench_dryrun.py, CPU): stands where `Denoiser` stands so that the launcher, the rank
    sharding, the bucketed exchange driver, the timing protocol and the result line of THIS script run end to end over gloo
    without a GPU.  It computes nothing of the workload; its line carries "stub": true and is never a measurement."""

    def __init__(self, device):
        from ssdn.hip.graph import net_layers, net_param_count
        self.layers = net_layers(3, 9, True)
        self.n = net_param_count(self.layers)
        self.flat = torch.zeros(self.n, device=device)
        self.flat_grad = torch.zeros(self.n, device=device)
        self.device = device

    def train(self):
        pass

    def gradient_exchange(self, world):
        from ssdn.hip import dp
        return dp.GradExchange(world, dp.bucket_ranges(self.layers, self.n, self.n), self.device)

    def train_step(self, data, lr, exchange=None):
        from ssdn.hip import dp
        x = data[0]

        def bwd(ex):
            self.flat_grad.fill_(float(x.float().mean()))
        scale = dp.exchange_step(bwd, self.flat_grad, exchange)
        self.flat.sub_(lr * scale * self.flat_grad)
        return {}


def run_rank(args):
    import ssdn  # noqa: F401
    from ssdn.datasets import DevicePatchStream, NoisyDataset
    from ssdn.hip import dp, lib as L
    from ssdn.params import NoiseAlgorithm
    from ssdn.utils.utils import compute_ramped_lrate
    import torch.distributed as dist

    stub = os.environ.get("SSDN_BENCH_STUB") == "1"
    # SSDN_DP_PLAN=split|buckets|all: which weight-gradient plan the benchmarked engines are built with, so that a
    # driver-side scaling run can compare them without editing source -- "split" (default: the head bucket's weight gradients on the side
    # lane, everything else one chip-wide launch: the fastest single-GPU step; 4.33 MB of the exchange sit behind the backward pass),
    # "buckets" (one launch per gradient bucket: every bucket's all-reduce runs under the next bucket's launch, at ~13 % more single-GPU
    # time).  The package itself never reads the environment (ssdn.hip.graph: planner constants); the choice is echoed in the line.
    dp_plan = os.environ.get("SSDN_DP_PLAN", "split")
    if dp_plan not in ("split", "buckets", "all"):
        raise RuntimeError("SSDN_DP_PLAN must be split, buckets or all (got %r)" % dp_plan)
    if not stub:
        from ssdn.hip import graph as _graph
        _graph.WGRAD_MEGA = dp_plan
    # SSDN_BENCH_BACKEND=gloo: ranks may share a GPU (device = LOCAL_RANK mod visible GPUs) and exchange device tensors through
    # gloo -- how the whole N-rank path is exercised on the 1-GPU test boxes; RCCL itself refuses two ranks on one device
    backend = "gloo" if stub else os.environ.get("SSDN_BENCH_BACKEND")
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    local_env = int(os.environ.get("LOCAL_RANK", 0))
    rank, world, local = dp.init_from_env(backend, device_index=(local_env % ndev) if ndev else None)
    if ndev:
        local = local_env % ndev
    if world != args.gpus:
        raise RuntimeError("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if stub:
        device = torch.device("cpu")
        lib = None
        d = _StubDenoiser(device)
    else:
        if not torch.cuda.is_available():
            raise L.SsdnHipError("bench.py needs an MI355X (no GPU visible); the ssdn hot path has no CPU fallback")
        device = torch.device("cuda", local)
        torch.cuda.set_device(device)
        lib = L.load()
        from ssdn.denoiser import Denoiser
        torch.manual_seed(0)                      # identical replicas: same init on every rank
        d = Denoiser(make_cfg(), device=str(device))
    d.train()
    B, P = args.batch, args.patch
    MD = NoisyDataset.Metadata

    def sync():
        if device.type == "cuda":
            torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        sync()

    # the minibatches as the trainer's DataLoader workers deliver them: clean uint8 patches in pinned host memory
    g = torch.Generator().manual_seed(1000 * (rank + 1))
    u8 = [torch.randint(0, 256, (B, 3, P, P), generator=g, dtype=torch.uint8) for _ in range(4)]
    if device.type == "cuda":
        u8 = [t.pin_memory() for t in u8]
    nd = NoisyDataset(None, "gauss25", NoiseAlgorithm.SELFSUPERVISED_DENOISING, pad_uniform=False, pad_multiple=32, square=True, training_mode=True)
    stream = DevicePatchStream(None, nd, device, seed=1, rank=rank)
    if not stub:
        stream.attach(d)              # the noise kernel writes the engine's input buffer directly (as the trainer's stream does)
    idx = torch.arange(B)
    exchange = d.gradient_exchange(world) if world > 1 else None    # bucketed all-reduce overlapped with backward
    N_IT = 2000000
    seen = 0

    def lr_now():
        return compute_ramped_lrate(seen + 200000, N_IT, 0.1, 0.3, 3e-4)   # flat part of the schedule

    pending = stream.upload(u8[0])

    def step(i):
        """the measured step: pinned uint8 minibatch -> device (EVERY step; the copy of minibatch i + 1 runs on the patch stream's
        copy stream under step i, as in DevicePatchStream.__iter__) -> noise kernel -> forward + loss + backward [+ all-reduce] + Adam"""
        nonlocal seen, pending
        cur, pending = pending, stream.upload(u8[(i + 1) % len(u8)])
        d.train_step(stream.prepare(cur, idx), lr_now(), exchange)
        seen += B * world

    # (the collector: the policy of DenoiserTrainer.train() -- everything alive goes to the permanent generation before the first step, so the
    #  automatic passes inside the timed region only see what the steps themselves allocate (a generation-2 pass over the whole heap takes
    #  longer than a training step, and the driver's 20-step region is 34 ms).  Collected BEFORE the warm-up steps, so that the device does not
    #  sit idle -- and clock down -- between them and the timed region)
    import gc
    gc.collect()
    gc.freeze()
    for i in range(args.warmup):
        step(i)

    # roofline leg: HIP events of launches of the dominant kernel, k_cdma<3,*> (the 96-output-channel 3x3 layers at
    # 32x32 and 64x64 pixels, forward and data-gradient roles: 8 launches per step, 60 % of the step's flops), on the launch
    # stream, during the timed steps.  Since round 5 the two events of a sample ride on the sampled launch's own dispatch
    # (hipExtLaunchKernelGGL start / stop event, csrc/common.h::SSDN_LAUNCH): the kernel's duration, within ~5 % of the rocprofv3
    # kernel trace (two hipEventRecord calls around the launch measured ~10 us more per sample and cost the stream as much).
    # Every 9th launch is sampled (9 is coprime to the 8 launches of a step: the sample rotates over all eight layers).
    PROF_STRIDE = 9
    if lib is not None:
        prof_kind = L.PROF["cdma_mt3"]
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
    if lib is not None:
        L.check(lib.ssdn_profile_read(prof_kind, C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by)))
        lib.ssdn_profile_enable(prof_kind, 0)

    # ---- after the timed region (does not touch `value`) ----------------------------------------------------------------
    # (a) per-family table: every MFMA kernel family bracketed at stride 3 over a few extra steps
    families = {}
    FAM_STEPS, FAM_STRIDE = 30, 3        # (ten samples of a once-per-step launch: with four, one disturbed sample moved the average by 80 %)
    if lib is not None:
        for name, kind in L.PROF.items():
            lib.ssdn_profile_enable(kind, 80 * FAM_STEPS // FAM_STRIDE + 64)
            lib.ssdn_profile_set_stride(kind, FAM_STRIDE)
        for i in range(FAM_STEPS):
            step(i)
        sync()
        for name, kind in L.PROF.items():
            m2, c2, f2, b2 = C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
            L.check(lib.ssdn_profile_read(kind, C.byref(m2), C.byref(c2), C.byref(f2), C.byref(b2)))
            lib.ssdn_profile_enable(kind, 0)
            if c2.value:
                families[name] = {"sampled_launches": int(c2.value), "avg_launch_us": round(1e3 * m2.value / c2.value, 2),
                                  "tflops_algorithmic": round(f2.value / 1e12 / (m2.value / 1e3), 1) if m2.value > 0 else None,
                                  "launches_per_step": round(c2.value * FAM_STRIDE / FAM_STEPS, 1),
                                  "ms_per_step_bracketed_sum": round(m2.value * FAM_STRIDE / FAM_STEPS, 4)}
    # (a1) the dominant kernel's ALGORITHMIC bytes per launch on one basis (the sampled launches' mean depended on which of
    # the eight launches of a step the samples fell on -- 109.9 MB in a 20-step run, 116.9 MB in a 100-step one): ONE extra step with every
    # launch of the family counted, i.e. the exact mean over the step's eight launches (inputs read once + outputs written once)
    alg_bytes_exact = None
    if lib is not None:
        lib.ssdn_profile_enable(prof_kind, 64)
        lib.ssdn_profile_set_stride(prof_kind, 1)
        step(0)
        sync()
        m3, c3, f3, b3 = C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
        L.check(lib.ssdn_profile_read(prof_kind, C.byref(m3), C.byref(c3), C.byref(f3), C.byref(b3)))
        lib.ssdn_profile_enable(prof_kind, 0)
        if c3.value:
            alg_bytes_exact = int(b3.value / c3.value)
    # (a2) N > 1: how long the optimiser stream WAITS for the gradient exchange behind the last slab reduction (events on the compute
    # stream around launch + finish of ssdn.hip.dp.exchange_step: last k_wreduce_multi -> k_adam_pack), median of EXP_STEPS extra steps
    exposed = None
    if exchange is not None:
        EXP_STEPS = 20
        exchange.measure_exposed(True)
        samples = []
        for i in range(EXP_STEPS):
            step(i)
            sync()
            v = exchange.exposed_us()
            if v is not None:
                samples.append(v)
        exchange.measure_exposed(False)
        if samples:
            samples.sort()
            exposed = {"allreduce_exposed_us": round(samples[len(samples) // 2], 1), "min": round(samples[0], 1), "max": round(samples[-1], 1),
                       "steps": len(samples), "collectives_per_step_bytes": exchange.unit_bytes(),
                       "what": "time the optimiser stream waits for the exchange between the last slab reduction and Adam (rank 0)"}
    # (b) side legs (N = 1), >= SIDE_STEPS steps each whatever --steps says: the prepared batch already resident in HBM
    # (kernel-only rate), and the reference's DataLoader format (fp32 noisy + clean from pinned host memory, 2 x 1.6 MB per step)
    resident_value = fp32_value = None
    SIDE_STEPS = max(50, args.steps // 2)
    if world == 1 and not stub:
        side = DevicePatchStream(None, nd, device, seed=2, rank=rank)         # (not attached: four distinct resident batches)
        res_batches = [side.prepare(u8[i], idx) for i in range(len(u8))]

        def timed(fn):
            for i in range(5):
                fn(i)
            sync()
            t1 = time.perf_counter()
            for i in range(SIDE_STEPS):
                fn(i)
            sync()
            return round(SIDE_STEPS * B / (time.perf_counter() - t1), 2)
        resident_value = timed(lambda i: d.train_step(res_batches[i % len(res_batches)], 3e-4, exchange))
        host = [[b[0].cpu().pin_memory(), b[2][MD.CLEAN].cpu().pin_memory(), b[2][MD.INPUT_NOISE_VALUES].cpu().pin_memory()] for b in res_batches]

        def fp32_step(i):
            hb = host[i % len(host)]
            clean = hb[1].to(device, non_blocking=True)
            d.train_step([hb[0].to(device, non_blocking=True), clean,
                          {MD.INPUT_NOISE_VALUES: hb[2].to(device, non_blocking=True), MD.CLEAN: clean}], 3e-4, exchange)
        fp32_value = timed(fp32_step)

    # (c) the REAL trainer end to end (N = 1): `DenoiserTrainer.train()` over a synthetic HDF5 file written in the reference
    # converter's layout -- forked DataLoader workers cropping at read (uint8), pinned upload, device noise, training step, the
    # per-step metrics on the device, console / scalar logging at PRINT_INTERVAL.  Timed from step TR_WARM to the last step.
    trainer_value = trainer_info = None
    if world == 1 and not stub and not args.no_trainer_leg:
        try:
            trainer_value, trainer_info = trainer_leg(B, P, max(200, args.steps), workers=args.trainer_workers)
        except Exception as e:  # noqa: BLE001 -- a side leg must never cost the line its `value`
            import traceback
            traceback.print_exc()
            trainer_value, trainer_info = None, {"error": "%s: %s" % (type(e).__name__, str(e)[:500])}

    if rank == 0:
        value = args.steps * B * world / dt
        achieved = (fl.value / 1e12) / (ms.value / 1e3) if ms.value > 0 else 0.0
        # HBM-side traffic of the same kernel family: PMC passes cannot run inside this process, so the per-launch figure is
        # the committed result of tools/pmc_traffic.sh (FETCH_SIZE x2 + WRITE_SIZE); null if absent
        traffic, traffic_src = None, None
        lib_sha = None
        try:
            import hashlib
            with open(L.LIB_PATH, "rb") as f:
                lib_sha = hashlib.sha256(f.read()).hexdigest()
        except OSError:
            pass
        # the newest committed PMC summary of THIS kernel (tools/pmc_traffic.sh) -- quoted only if it was collected with the library this
        # run loaded (lib_sha256 in the summary): a figure of another build says nothing about the benchmarked binary
        cands = [f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.match(r"r\d+_traffic\.json$", f)]
        for cand in sorted(cands, key=lambda f: int(re.match(r"r(\d+)_", f).group(1)), reverse=True):      # newest round first (r10 > r9)
            if lib_sha is None:
                traffic_src = "the loaded library could not be read back for its sha256: no PMC summary quoted"
                break
            try:
                with open(os.path.join(ROOT, "profiles", cand)) as f:
                    tj = json.load(f)
                if tj.get("lib_sha256") != lib_sha:
                    if traffic_src is None:
                        traffic_src = "profiles/%s describes another build of the library (its lib_sha256 differs): not quoted" % cand
                    continue        # (an older summary may still be the one of this build)
                traffic = int(tj["hbm_bytes_per_launch"])
                traffic_src = "profiles/%s (collected at %s, same library build: sha256 %s...)" % (cand, tj.get("collected_at", "?"), lib_sha[:12])
                break
            except (OSError, KeyError, ValueError):
                continue
        res = {
            "metric": "training patches/sec (64x64 gauss25 SSDN)", "value": round(value, 2), "unit": "patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic (clean uint8 patches U{0..255} in pinned host memory, uploaded EVERY step; clipped gauss25 noise made on the "
                    "device; random-init weights)",
            "config": {"workload": "ssdn gauss25 sigma_known, %dx%d RGB patches, batch %d per GPU, H2D of the minibatch + device noise + blind-spot U-Net fwd+bwd + posterior head + Adam" % (P, P, B),
                       "global_batch": B * world, "parallelism": "dp%d" % world, "dp_plan": dp_plan,
                       "achieved_train_tflops_algorithmic": round(value * TRAIN_GFLOP_PER_PATCH / 1e3, 2)},
        }
        if world > 1:
            res["config"]["backend"] = backend or "nccl (RCCL)"
            res["allreduce_exposed_us"] = exposed["allreduce_exposed_us"] if exposed else None
            res["allreduce"] = exposed
        if stub:
            res["stub"] = True
            res["data"] = "STUB ENGINE (SSDN_BENCH_STUB=1, CPU dry run of the launcher / exchange / timing protocol): not a measurement"
        else:
            res["roofline"] = {"bound": "mfma", "kernel": "k_cdma<3,*>: persistent LDS-DMA 3x3 convolution on 96-output-channel blocks (decode_block_1.*/2.*, forward + data-gradient roles, 8 launches per step); flops counted on REAL channels",
                               "achieved": round(achieved, 2), "peak": MFMA_FP16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(achieved / MFMA_FP16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic,
                               "traffic_unit": "HBM-side bytes per launch, rocprofv3 FETCH_SIZE*2 + WRITE_SIZE from separate --pmc passes over this "
                                               "command; NOT measured in this run: %s" % traffic_src,
                               "algorithmic_bytes_per_launch": alg_bytes_exact if alg_bytes_exact else int(by.value / max(1, cnt.value)),
                               "algorithmic_bytes_basis": "mean over ALL launches of the family in one step (every input read once + every output written once; one extra step after the timed region with every launch counted)",
                               "launches": int(cnt.value), "sampling": "every %dth launch of the timed region; the two HIP events of a sample ride on the sampled launch's own dispatch (hipExtLaunchKernelGGL start / stop event: the kernel's duration as a kernel trace reports it)" % PROF_STRIDE, "avg_launch_us": round(1e3 * ms.value / max(1, cnt.value), 3),
                               "kernel_time_share": round(PROF_STRIDE * ms.value / 1e3 / dt, 4)}
            res["families"] = families
            # the time-dominant family next to the flop-dominant one: the weight gradients of the whole network,
            # ONE chip-wide launch per step (k_wgrad_mega), bracketed in the family leg after the timed region
            wg = families.get("wgrad")
            if wg and wg.get("tflops_algorithmic"):
                res["roofline"]["second_kernel"] = {
                    "kernel": "k_wgrad_mega, the chip-wide launch: weight + bias gradients of every layer but the head's (those: families.wgrad_side, a half-chip launch on the side lane under the latency-bound bottom of the U), one block per (op, pixel partition), a block owns its CU",
                    "bound": "mfma", "achieved": wg["tflops_algorithmic"], "peak": MFMA_FP16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(wg["tflops_algorithmic"] / MFMA_FP16_DENSE_PEAK_TFLOPS, 4), "avg_launch_us": wg["avg_launch_us"],
                    "launches_per_step": wg["launches_per_step"],
                    "kernel_time_share": round(wg["ms_per_step_bracketed_sum"] / (1e3 * dt / args.steps), 4),
                    "sampling": "every %d. launch of %d extra steps after the timed region" % (FAM_STRIDE, FAM_STEPS)}
        if resident_value is not None:
            res["value_resident"] = resident_value
            res["value_with_fp32_h2d"] = fp32_value
            res["side_leg_steps"] = SIDE_STEPS
        if trainer_value is not None or trainer_info is not None:
            res["value_trainer_hdf5"] = trainer_value
            res["trainer_hdf5"] = trainer_info
        if world == 1 and not args.no_cpu_baseline and not stub:
            res["cpu_baseline"] = cpu_baseline(P)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="patches per GPU")
    ap.add_argument("--patch", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-trainer-leg", action="store_true", help="skip the value_trainer_hdf5 side leg (DenoiserTrainer.train() over an HDF5 file)")
    ap.add_argument("--trainer-workers", type=int, default=8)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    try:
        run_rank(args)
    except BaseException as e:  # noqa: BLE001 -- the contract is ONE JSON line, also for failures
        if isinstance(e, (SystemExit, KeyboardInterrupt)):
            raise
        import traceback
        traceback.print_exc()
        if int(os.environ.get("RANK", 0)) == 0:
            print(json.dumps({"error": "%s: %s" % (type(e).__name__, str(e)[:2000]), "n_gpus": args.gpus}))
        sys.exit(1)


if __name__ == "__main__":
    main()

"""Data parallelism for the ssdn hot path: one process per GPU, identical replicas, ONE exchange per step -- a
sum-all-reduce of the flat fp32 gradient buffer over RCCL (torch.distributed backend "nccl" is RCCL on ROCm), averaged by
folding 1/world into the fused Adam pass.  Replaces nn.DataParallel (reference: denoiser.py:102-110):
no per-step parameter broadcast, no scatter/gather of activations, the loss head runs on every rank for its own shard;
the mean over the GLOBAL batch of the reference's `torch.mean(LOSS).backward()` (train.py:201) is reproduced exactly
(every rank differentiates the mean over its shard, the sum over ranks is divided by the world size).

The gradient is exchanged in up to five contiguous buckets of the flat buffer (head | dec1 | dec2..dec5 | encoder | sigma-net +
learnable sigma).  `GradExchange` issues the exchange WITHOUT splitting the backward op list: the list carries
SSDN_OP_EVENT_RECORD marks -- one per (point of the list, lane) behind which a bucket's last slab reduction ON THAT LANE has been
enqueued -- the whole list is enqueued asynchronously, then each bucket's all-reduce is issued on a communication stream that
waits for ALL of the bucket's marks; the optimiser stream waits for the collectives only right before Adam.
What overlaps depends on the plan (graph.WGRAD_MEGA).  Default plan "split": the head bucket (output_block.*, 0.74 MB) is
complete when the side-lane weight-gradient launch and its reductions end, ~0.4 ms before the end of the backward pass: its
all-reduce runs UNDER the rest of the backward pass.  Every other layer's gradient comes out of the chip-wide launch that closes
the backward pass, so those three buckets complete together behind the final reduction run and are exchanged as ONE all-reduce
(4.34 MB) that is EXPOSED in front of Adam (bench.py --gpus N reports it as `allreduce_exposed_us`); with a sigma estimator
(config 3) that network's backward pass follows and covers it.  Plan "buckets" (one launch per bucket, +0.12 ms per step on one
GPU) overlaps every bucket but the last.  With world_size 1 nothing is communicated.

Everything here also runs on CPU tensors with the "gloo" backend (tests/test_dp_gloo.py), where "events" degenerate to
program order.  NOTE: scaling across GPUs has not been measured on hardware by the builder (the driver owns 8-GPU runs).
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Set, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_from_env(backend: Optional[str] = None, device_index: Optional[int] = None) -> Tuple[int, int, int]:
    """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (torchrun sets them) and select
    this rank's GPU (LOCAL_RANK unless `device_index` says otherwise) before anything is allocated.  Process-wide side
    effects happen only when there is a job to join (WORLD_SIZE > 1) or a device is named explicitly: constructing a trainer
    in a single-process program (evaluation included) leaves the caller's current device alone."""
    rank, world, local = env_world()
    have_gpu = torch.cuda.is_available()
    dev = local if device_index is None else device_index
    if have_gpu and (world > 1 or device_index is not None):
        torch.cuda.set_device(dev)
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if have_gpu else "gloo"
        if backend == "nccl":
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_rows(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Rank r of W takes samples [r*B/W, (r+1)*B/W) of every global minibatch, so a W-GPU run consumes the same sample
    order as a single-GPU run with batch B."""
    if global_batch % world:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


def bucket_ranges(layers, n_main: int, n_total: int) -> List[Tuple[int, int]]:
    """Contiguous [lo,hi) ranges of the flat buffer in backward completion order."""
    off = {l.name: l.w_off for l in layers}
    return [(off["output_block.0"], n_main), (off["decode_block_1.0"], off["output_block.0"]),
            (off["decode_block_5.0"], off["decode_block_1.0"]), (0, off["decode_block_5.0"])] + \
        ([(n_main, n_total)] if n_total > n_main else [])


N_MAIN_BUCKETS = 4


def bucket_layers(layers, split_head: bool = True) -> List[Set[str]]:
    """Layer names of the main-net buckets of `bucket_ranges`, same order: head | dec1 | dec2..dec5 | encoder.  split_head=False:
    head and dec1 as one set (the three reduction runs of the per-layer plans, engine._group_reductions_lanes)."""
    off = {l.name: l.w_off for l in layers}
    h, a, b = off["output_block.0"], off["decode_block_1.0"], off["decode_block_5.0"]
    head, dec1 = {l.name for l in layers if l.w_off >= h}, {l.name for l in layers if a <= l.w_off < h}
    rest = [{l.name for l in layers if b <= l.w_off < a}, {l.name for l in layers if l.w_off < b}]
    return ([head, dec1] if split_head else [head | dec1]) + rest


class GradExchange:
    """Bucketed gradient all-reduce overlapped with the backward pass.

        ex = GradExchange(world, bucket_ranges(...))
        ... backward op list enqueued (its SSDN_OP_EVENT_RECORD ops record the marks made with ex.new_event(buckets): one per
            (point of the list, lane) behind which a bucket's last slab reduction on that lane has been enqueued) ...
        ex.launch(flat_grad)        # one asynchronous all-reduce per bucket, each behind its bucket's event
        scale = ex.finish()         # the current stream waits for the collectives; returns 1 / world for Adam

    On CUDA/HIP tensors the collectives are issued from a dedicated communication stream; on CPU tensors (gloo) the
    events are absent and the buckets are reduced in order -- the control flow is the same."""

    def __init__(self, world: int, ranges: Sequence[Tuple[int, int]], device: Optional[torch.device] = None,
                 force_events: bool = False, timing_marks: bool = False):
        self.world, self.ranges = world, [(int(lo), int(hi)) for lo, hi in ranges]
        # force_events with an initialised process group: the collectives are issued even for world 1 (an identity all-reduce
        # through RCCL: the communication stream, the event waits and the asynchronous work handles all run -- the -m gpu test
        # of the exchange on a 1-GPU box)
        self.force = bool(force_events)
        self.timing_marks = bool(timing_marks)              # test aid: marks carry timestamps (when did a bucket complete?)
        self.pending: list = []
        self.device = torch.device(device) if device is not None else None
        self.cuda = self.device is not None and self.device.type == "cuda"
        self.events: list = []                              # every mark of the backward list (torch.cuda.Event)
        self.waits: dict = {}                               # bucket -> indexes into `events` its all-reduce waits for
        self._here: dict = {}                               # bucket -> index of its record_here event
        self.comm_stream = None
        self.exposed_events = None                          # (start, end) timing events around the optimiser stream's wait (measure_exposed)
        if self.cuda and (world > 1 or force_events):      # (force_events: single-GPU test of the event-carrying backward list)
            self.comm_stream = torch.cuda.Stream(device=self.device)

    @property
    def overlapped(self) -> bool:
        return self.comm_stream is not None

    def new_event(self, buckets) -> int:
        """A mark of the backward list: returns the raw hipEvent_t handle for an SSDN_OP_EVENT_RECORD; the all-reduce of every bucket
        in `buckets` waits for it.  (A bucket whose reductions ran on several lanes has one mark per lane.)"""
        e = torch.cuda.Event(enable_timing=self.timing_marks, blocking=False)
        e.record(torch.cuda.current_stream(self.device))            # creates the underlying hipEvent_t
        self.events.append(e)
        for k in buckets:
            self.waits.setdefault(int(k), []).append(len(self.events) - 1)
        return int(e.cuda_event)

    def reset_marks(self):
        """forget the marks of a previous backward list (the engine rebuilds its list for this exchange)"""
        self.events, self.waits, self._here = [], {}, {}

    def record_here(self, k: int):
        """Bucket k is complete at the current position of the current stream (used for buckets whose producer is not
        the main net's op list: the sigma-estimation network)."""
        if self.comm_stream is None:
            return
        if k not in self._here:
            self.new_event([k])
            self._here[k] = len(self.events) - 1
        self.events[self._here[k]].record(torch.cuda.current_stream(self.device))

    def launch(self, flat_grad: torch.Tensor):
        if self.world <= 1 and not (self.force and dist.is_available() and dist.is_initialized()):
            return
        for lo, hi, ks in self._units():
            if self.comm_stream is not None:
                for j in sorted({j for k in ks for j in self.waits.get(k, [])}):
                    self.comm_stream.wait_event(self.events[j])
                with torch.cuda.stream(self.comm_stream):
                    self.pending.append(dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
            else:
                self.pending.append(dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def measure_exposed(self, on: bool = True):
        """Bracket the optimiser stream's WAIT for the collectives with two timing events (`exchange_step` records them on the current
        stream: one behind the backward list, one behind `finish()`, i.e. in front of Adam): the elapsed time between them is the
        time the exchange is exposed after the last slab reduction.  `exposed_us()` reads the last step's bracket (after a
        synchronize)."""
        self.exposed_events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if on and self.cuda else None
        self._host_timing, self._host_exposed_us = bool(on) and not self.cuda, None      # (CPU tensors / gloo: host clock around launch + finish)

    def exposed_us(self) -> Optional[float]:
        if self.exposed_events is None:
            return getattr(self, "_host_exposed_us", None)
        return 1e3 * self.exposed_events[0].elapsed_time(self.exposed_events[1])

    def unit_bytes(self) -> List[int]:
        """sizes of the collectives a step issues, in issue order"""
        return [4 * (hi - lo) for lo, hi, _ in self._units()]

    def _units(self):
        """(lo, hi, bucket indexes) of the collectives to issue.  `groups` (set by the engine that builds the event-carrying backward
        list) names buckets whose completion marks sit at the SAME point of the list -- with the chip-wide weight-gradient launch
        every main-net bucket completes behind the final reduction run -- : such buckets are exchanged as ONE all-reduce when their
        ranges are adjacent in the flat buffer (one latency-bound ring instead of three back to back)."""
        groups = getattr(self, "groups", None) or [[k] for k in range(len(self.ranges))]
        units = []
        for g in groups:
            ks = sorted((k for k in g if self.ranges[k][1] > self.ranges[k][0]), key=lambda k: self.ranges[k][0])
            if not ks:
                continue
            contiguous = all(self.ranges[a][1] == self.ranges[b][0] for a, b in zip(ks, ks[1:]))
            if contiguous:
                units.append((self.ranges[ks[0]][0], self.ranges[ks[-1]][1], ks))
            else:
                units += [(self.ranges[k][0], self.ranges[k][1], [k]) for k in ks]
        return units

    def finish(self) -> float:
        for w in self.pending:
            w.wait()            # NCCL/RCCL: the CURRENT stream waits (no host block); gloo: host wait
        self.pending = []
        return 1.0 / self.world

    # one blocking all-reduce of the whole buffer (kept for callers that do not overlap)
    def __call__(self, flat_grad: torch.Tensor) -> float:
        if self.world > 1:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        return 1.0 / self.world


def exchange_step(run_backward: Callable[[Optional[GradExchange]], None], flat_grad: torch.Tensor,
                  exchange: Optional[GradExchange]) -> float:
    """The gradient half of one data-parallel optimisation step -- the SAME code path for `Denoiser.train_step`,
    `bench.py --gpus N` and the CPU/gloo test: enqueue the backward pass (which marks bucket completion on `exchange`),
    launch the per-bucket all-reduces behind those marks, make the optimiser wait for them.  Returns the gradient scale
    (1 / world) the fused Adam folds in."""
    run_backward(exchange)
    if exchange is None:
        return 1.0
    host_t0 = None
    if exchange.exposed_events is not None:
        exchange.exposed_events[0].record(torch.cuda.current_stream(exchange.device))
    elif getattr(exchange, "_host_timing", False):
        import time
        host_t0 = time.perf_counter()
    exchange.launch(flat_grad)
    scale = exchange.finish()
    if exchange.exposed_events is not None:
        exchange.exposed_events[1].record(torch.cuda.current_stream(exchange.device))
    elif host_t0 is not None:
        exchange._host_exposed_us = 1e6 * (time.perf_counter() - host_t0)
    return scale

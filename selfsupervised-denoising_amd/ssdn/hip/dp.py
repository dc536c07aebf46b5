"""Data parallelism for the ssdn hot path: one process per GPU, identical replicas, ONE exchange per step -- a
sum-all-reduce of the flat fp32 gradient buffer over RCCL (torch.distributed backend "nccl" is RCCL on ROCm), averaged by
folding 1/world into the fused Adam pass.  Replaces nn.DataParallel (reference: denoiser.py:102-110):
no per-step parameter broadcast, no scatter/gather of activations, the loss head runs on every rank for its own shard;
the mean over the GLOBAL batch of the reference's `torch.mean(LOSS).backward()` (train.py:201) is reproduced exactly
(every rank differentiates the mean over its shard, the sum over ranks is divided by the world size).

The gradient is exchanged in up to four contiguous buckets of the flat buffer, in the order the backward pass completes
them (head+dec1 | dec2..dec5 | encoder | sigma-net + learnable sigma).  `GradExchange` overlaps the exchange with the
backward pass WITHOUT splitting the backward op list: the list carries one SSDN_OP_EVENT_RECORD per bucket on the
weight-gradient lane (right after the bucket's last slab reduction); the whole list is enqueued asynchronously, then each
bucket's all-reduce is issued on a communication stream that waits for the bucket's event.  RCCL traffic (5-10 MB, latency
bound on xGMI) therefore runs under the remaining backward kernels; the optimiser stream waits for the collectives only
right before Adam.  With world_size 1 nothing is communicated.

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
    return [(off["decode_block_1.0"], n_main), (off["decode_block_5.0"], off["decode_block_1.0"]), (0, off["decode_block_5.0"])] + \
        ([(n_main, n_total)] if n_total > n_main else [])


def bucket_layers(layers) -> List[Set[str]]:
    """Layer names of the three main-net buckets of `bucket_ranges`, same order."""
    off = {l.name: l.w_off for l in layers}
    a, b = off["decode_block_1.0"], off["decode_block_5.0"]
    return [{l.name for l in layers if l.w_off >= a}, {l.name for l in layers if b <= l.w_off < a},
            {l.name for l in layers if l.w_off < b}]


class GradExchange:
    """Bucketed gradient all-reduce overlapped with the backward pass.

        ex = GradExchange(world, bucket_ranges(...))
        ... backward op list enqueued (it records ex.event_handles()[k] when bucket k is complete) ...
        ex.launch(flat_grad)        # one asynchronous all-reduce per bucket, each behind its bucket's event
        scale = ex.finish()         # the current stream waits for the collectives; returns 1 / world for Adam

    On CUDA/HIP tensors the collectives are issued from a dedicated communication stream; on CPU tensors (gloo) the
    events are absent and the buckets are reduced in order -- the control flow is the same."""

    def __init__(self, world: int, ranges: Sequence[Tuple[int, int]], device: Optional[torch.device] = None,
                 force_events: bool = False):
        self.world, self.ranges = world, [(int(lo), int(hi)) for lo, hi in ranges]
        # force_events with an initialised process group: the collectives are issued even for world 1 (an identity all-reduce
        # through RCCL: the communication stream, the event waits and the asynchronous work handles all run -- the -m gpu test
        # of the exchange on a 1-GPU box)
        self.force = bool(force_events)
        self.pending: list = []
        self.device = torch.device(device) if device is not None else None
        self.cuda = self.device is not None and self.device.type == "cuda"
        self.events: list = []
        self.comm_stream = None
        if self.cuda and (world > 1 or force_events):      # (force_events: single-GPU test of the event-carrying backward list)
            self.comm_stream = torch.cuda.Stream(device=self.device)
            for _ in self.ranges:
                e = torch.cuda.Event(enable_timing=False, blocking=False)
                e.record(torch.cuda.current_stream(self.device))        # creates the underlying hipEvent_t
                self.events.append(e)

    @property
    def overlapped(self) -> bool:
        return bool(self.events)

    def event_handles(self) -> List[int]:
        """raw hipEvent_t handles, one per bucket (for SSDN_OP_EVENT_RECORD)"""
        return [int(e.cuda_event) for e in self.events]

    def record_here(self, k: int):
        """Bucket k is complete at the current position of the current stream (used for buckets whose producer is not
        the main net's op list: the sigma-estimation network)."""
        if self.events:
            self.events[k].record(torch.cuda.current_stream(self.device))

    def launch(self, flat_grad: torch.Tensor):
        if self.world <= 1 and not (self.force and dist.is_available() and dist.is_initialized()):
            return
        for lo, hi, ks in self._units():
            if self.comm_stream is not None:
                for k in ks:
                    self.comm_stream.wait_event(self.events[k])
                with torch.cuda.stream(self.comm_stream):
                    self.pending.append(dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
            else:
                self.pending.append(dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

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
    exchange.launch(flat_grad)
    return exchange.finish()

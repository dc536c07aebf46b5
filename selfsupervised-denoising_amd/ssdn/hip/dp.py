"""Data parallelism for the ssdn hot path: one process per GPU, identical replicas, ONE exchange per step -- a
sum-all-reduce of the flat fp32 gradient buffer over RCCL (torch.distributed backend "nccl" is RCCL on ROCm), averaged by
folding 1/world into the fused Adam pass.  Replaces nn.DataParallel (reference: denoiser.py:102-110):
no per-step parameter broadcast, no scatter/gather of activations, the loss head runs on every rank for its own shard.

The gradient is exchanged in up to three contiguous buckets of the flat buffer in the order the backward pass
completes them (head+dec1 | dec2..dec5 | encoder [| sigma-net, learnable sigma]); every bucket's all-reduce is issued
asynchronously right after the segment of the backward op list that finishes it, so RCCL traffic (5-10 MB, latency
bound on xGMI) overlaps the rest of the backward pass.  With world_size 1 nothing is communicated.

Everything here also runs on CPU tensors with the "gloo" backend (tests/test_dp_gloo.py).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (torchrun sets them)."""
    rank, world, local = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
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


class GradAllReduce:
    """Callable handed to Denoiser.train_step: sums `flat_grad` over ranks, returns the scale (1/world) Adam applies."""

    def __init__(self, world: int):
        self.world = world
        self.pending = []

    def bucket(self, flat_grad: torch.Tensor, lo: int, hi: int):
        """Asynchronously all-reduce one finished bucket (called between backward segments)."""
        if self.world > 1 and hi > lo:
            self.pending.append(dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self) -> float:
        for w in self.pending:
            w.wait()
        self.pending = []
        return 1.0 / self.world

    def __call__(self, flat_grad: torch.Tensor) -> float:
        if self.world > 1:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        return 1.0 / self.world

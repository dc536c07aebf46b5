"""N > 1 path on CPU: two gloo processes, each computing the gradient of ITS shard (compute stand-in = the oracle, this is
a test), all-reduced with ssdn.hip.dp.GradAllReduce and averaged the way the fused Adam does (gscale = 1/world).  The result
must equal the single-process gradient of the whole batch -- i.e. the W-GPU run optimises mean(LOSS) over the global batch
exactly like the reference's DataParallel run (train.py:201)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import restate as R
from ssdn.hip import dp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat_grad(tr):
    return torch.cat([t.grad.reshape(-1) for t in tr.leaves])


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    r, w, _ = dp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    B, P = 4, 32
    noisy = R.hash_tensor((B, 1, P, P), 5, 0, 1)
    clean = R.hash_tensor((B, 1, P, P), 6, 0, 1)
    lo, hi = dp.shard_rows(B, rank, world)
    tr = R.CpuTrainer("n2c", 1, seed=3)
    res = tr.forward(noisy[lo:hi], clean[lo:hi])
    res["loss"].mean().backward()
    flat = _flat_grad(tr)
    ar = dp.GradAllReduce(world)
    # bucketed, asynchronous form (what overlaps with the backward pass on the GPU)
    n = flat.numel()
    for lo_, hi_ in ((n // 2, n), (0, n // 2)):
        ar.bucket(flat, lo_, hi_)
    scale = ar.finish()
    flat2 = _flat_grad(tr)
    scale2 = ar(flat2)                       # monolithic form
    if rank == 0:
        # numpy, not torch tensors: a tensor in a Queue is shared through a file descriptor the consumer must fetch from
        # THIS process, which may already have exited when a loaded host gets round to unpickling (EOFError)
        out.put(((flat * scale).numpy(), (flat2 * scale2).numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _run_world(world):
    """spawn `world` gloo ranks; a rendezvous can lose the race for the probed port on a busy host, so try twice"""
    import queue
    ctx = mp.get_context("spawn")
    last = None
    for attempt in range(2):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            got = q.get(timeout=240)
        except queue.Empty as e:   # pragma: no cover
            got, last = None, e
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        if got is not None and all(p.exitcode == 0 for p in procs):
            return got
        last = last or RuntimeError("rank exit codes %s" % [p.exitcode for p in procs])
    raise last


def test_two_rank_gradient_equals_single_process():
    got_bucketed, got_mono = (torch.from_numpy(a) for a in _run_world(2))
    B, P = 4, 32
    tr = R.CpuTrainer("n2c", 1, seed=3)
    res = tr.forward(R.hash_tensor((B, 1, P, P), 5, 0, 1), R.hash_tensor((B, 1, P, P), 6, 0, 1))
    res["loss"].mean().backward()
    want = _flat_grad(tr)
    # fp32 CPU convolutions sum in a thread-partition dependent order (2 threads per rank vs the parent's pool; the split can
    # change with host load): tolerance relative to the gradient's scale, not to each element
    tol = 2e-5 * float(want.abs().max())
    assert float((got_bucketed - want).abs().max()) <= tol
    assert float((got_mono - want).abs().max()) <= tol
    assert torch.equal(got_bucketed, got_mono)      # the two exchange forms are the same sums


def test_shard_rows_partition():
    for B, W in ((32, 1), (32, 2), (256, 8)):
        rows = [dp.shard_rows(B, r, W) for r in range(W)]
        assert rows[0][0] == 0 and rows[-1][1] == B
        assert all(rows[i][1] == rows[i + 1][0] for i in range(W - 1))
    with pytest.raises(ValueError):
        dp.shard_rows(30, 0, 4)


def test_bucket_ranges_cover_flat_buffer_in_backward_order():
    from ssdn.hip.graph import net_layers, net_param_count
    L = net_layers(3, 9, True)
    n = net_param_count(L)
    r = dp.bucket_ranges(L, n, n + 1)
    assert r[0][1] == n and r[-1] == (n, n + 1) and r[2][0] == 0
    covered = sorted(r)
    assert covered[0][0] == 0 and all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))

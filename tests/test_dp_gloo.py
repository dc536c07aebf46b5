"""N > 1 path on CPU: two gloo processes drive the SAME step-driver code `Denoiser.train_step` / `bench.py --gpus N` use
(`ssdn.hip.dp.exchange_step` + `GradExchange` with the real bucket ranges of a network), with a stub engine whose backward is
the oracle (this is a test) writing the shard's gradient into the flat layout.  The bucketed exchange, scaled by the 1/world
the fused Adam folds in, must equal the single-process gradient of the whole batch -- i.e. the W-GPU run optimises
mean(LOSS) over the global batch exactly like the reference's DataParallel run (train.py:201)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import restate as R
from ssdn.hip import dp
from ssdn.hip.graph import net_layers, net_param_count


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


LAYERS = net_layers(1, 1, False)
NPAR = net_param_count(LAYERS)


def _flat_grad(tr):
    """oracle gradients -> the Denoiser's flat parameter layout"""
    f = torch.zeros(NPAR + 1)                      # + one slot standing for the learnable-sigma scalar (4th bucket)
    for l in LAYERS:
        f[l.w_off:l.w_off + l.M * l.cin * l.k * l.k] = tr.p[l.name + ".weight"].grad.reshape(-1)
        f[l.b_off:l.b_off + l.M] = tr.p[l.name + ".bias"].grad
    f[NPAR] = float(sum(float(t.grad.sum()) for t in tr.leaves[:2]))
    return f


class StubEngine:
    """stands where DenoiserEngine stands in Denoiser.train_step: backward(exchange=...) fills the flat gradient"""

    def __init__(self, rank, world):
        B, P = 4, 32
        lo, hi = dp.shard_rows(B, rank, world)
        self.noisy = R.hash_tensor((B, 1, P, P), 5, 0, 1)[lo:hi]
        self.clean = R.hash_tensor((B, 1, P, P), 6, 0, 1)[lo:hi]
        self.tr = R.CpuTrainer("n2c", 1, seed=3)
        self.flat_grad = torch.zeros(NPAR + 1)
        self.seen_exchange = None

    def backward(self, exchange=None):
        for t in self.tr.leaves:
            t.grad = None
        self.tr.forward(self.noisy, self.clean)["loss"].mean().backward()
        self.flat_grad.copy_(_flat_grad(self.tr))
        self.seen_exchange = exchange


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    r, w, _ = dp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    eng = StubEngine(rank, world)
    ex = dp.GradExchange(world, dp.bucket_ranges(LAYERS, NPAR, NPAR + 1), torch.device("cpu"))
    assert not ex.overlapped and len(ex.ranges) == 5
    # the step driver of Denoiser.train_step / bench.py
    scale = dp.exchange_step(lambda e: eng.backward(exchange=e), eng.flat_grad, ex)
    assert eng.seen_exchange is ex and ex.pending == []
    bucketed = (eng.flat_grad * scale).clone()
    eng.backward()
    scale2 = ex(eng.flat_grad)               # monolithic form
    if rank == 0:
        # numpy, not torch tensors: a tensor in a Queue is shared through a file descriptor the consumer must fetch from
        # THIS process, which may already have exited when a loaded host gets round to unpickling (EOFError)
        out.put((bucketed.numpy(), (eng.flat_grad * scale2).numpy()))
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
    eng = StubEngine(0, 1)                       # world 1: the whole batch
    assert dp.exchange_step(lambda e: eng.backward(exchange=e), eng.flat_grad, None) == 1.0
    want = eng.flat_grad
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
    assert r[0][1] == n and r[-1] == (n, n + 1) and r[3][0] == 0
    covered = sorted(r)
    assert covered[0][0] == 0 and all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))


def test_gradient_buckets_of_the_benchmark_network():
    """the flat gradient of the blind-spot RGB network (BASELINE configs 2 / 3) is exchanged in <= 5 contiguous
    buckets -- head | decode_block_1 | decode_block_2..5 | encoder | sigma estimator -- which tile the whole buffer exactly once;
    sizes 0.74 / 0.67 / 3.15 / 0.50 / 4.41 MB."""
    layers = net_layers(3, 9, True)
    n_main = net_param_count(layers)
    n_sig = net_param_count(net_layers(3, 1, False))
    assert (n_main, n_sig) == (1269129, 1102177)
    r = dp.bucket_ranges(layers, n_main, n_main + n_sig)
    assert r == [(1083456, 1269129), (914784, 1083456), (126048, 914784), (0, 126048), (1269129, 2371306)]
    assert sorted(r)[0][0] == 0 and all(a[1] == b[0] for a, b in zip(sorted(r), sorted(r)[1:])) and sorted(r)[-1][1] == n_main + n_sig
    assert [round((hi - lo) * 4 / 1e6, 2) for lo, hi in r] == [0.74, 0.67, 3.15, 0.50, 4.41]
    assert len(dp.bucket_ranges(layers, n_main, n_main)) == dp.N_MAIN_BUCKETS == 4      # no sigma estimator: four buckets
    names = dp.bucket_layers(layers)
    assert {"output_block.0", "output_block.2", "output_block.4"} == names[0] and {"decode_block_1.0", "decode_block_1.2"} == names[1]
    assert all(n.startswith("decode_block_") for n in names[2]) and len(names[2]) == 8
    assert all(n.startswith("encode_block_") for n in names[3]) and len(names[3]) == 7
    assert dp.bucket_layers(layers, split_head=False) == [names[0] | names[1], names[2], names[3]]
    off = {l.name: l.w_off for l in layers}
    for k, b in enumerate(names):                                          # a bucket's layers are exactly its range of the buffer
        assert all(r[k][0] <= off[n] < r[k][1] for n in b)


def test_coincident_buckets_are_exchanged_as_one_collective():
    """With the chip-wide weight-gradient launch every main-net bucket completes behind the final reduction run: the engine reports the
    buckets as coincident (`GradExchange.groups`) and adjacent ranges become ONE all-reduce; non-adjacent or separate buckets stay."""
    ex = dp.GradExchange(2, [(60, 100), (20, 60), (0, 20), (100, 130)], torch.device("cpu"))
    assert [(lo, hi) for lo, hi, _ in ex._units()] == [(60, 100), (20, 60), (0, 20), (100, 130)]
    ex.groups = [[0, 1, 2], [3]]
    assert [(lo, hi, ks) for lo, hi, ks in ex._units()] == [(0, 100, [2, 1, 0]), (100, 130, [3])]
    ex.groups = [[0, 2], [1], [3]]                      # (not adjacent: not merged)
    assert [(lo, hi) for lo, hi, _ in ex._units()] == [(0, 20), (60, 100), (20, 60), (100, 130)]
    # the engine side: marks of buckets that end in one run of reductions sit behind that run, at the same point
    from ssdn.hip.engine import DeviceNet
    from ssdn.hip.graph import NetPlan
    from ssdn.hip import lib as L
    plan = NetPlan("m/", 3, 9, True, 32, 64, 64, cus=256)
    flat = torch.zeros(plan.nparams)
    dn = DeviceNet(plan, torch.device("cpu"), flat, torch.zeros_like(flat))
    handed = []

    def new_event(ks):
        handed.append(list(ks))
        return 1000 + len(handed)
    ol = dn.bwd_with_events(dp.bucket_layers(plan.layers), new_event)
    types = [int(ol.arr[i].type) for i in range(ol.n)]
    lanes = [int(ol.arr[i].lane) for i in range(ol.n)]
    ev, red = L.OP["event_record"], L.OP["wreduce"]
    # plan "split": the side-lane launch (head layers + decode_block_2.2) reduces on lane 1 in the middle of the backward pass -- the head
    # bucket (0) is complete THERE and is exchanged on its own, early; decode_block_2.2's bucket (2) has reductions on both lanes and
    # gets a mark on each; the three buckets that end in the final run are coincident
    assert handed == [[0, 2], [1, 2, 3]]
    marks = [(i, lanes[i]) for i, t in enumerate(types) if t == ev]
    assert [l for _, l in marks] == [1, 0] and marks[1][0] == ol.n - 1
    assert types[marks[0][0] - 1] == red and lanes[marks[0][0] - 1] == 1 and types[-2] == red and lanes[-2] == 0
    assert ol.coincident == [[0], [1, 2, 3]]
    runs = sum(1 for i, t in enumerate(types) if t == red and (i == 0 or types[i - 1] != red))
    assert runs == 2, "reductions: the side group's run and the final run, not split by the marks"
    # every reduction of a bucket is followed, ON ITS OWN LANE, by a mark of that bucket (the collective must be
    # ordered after every lane that wrote the bucket's range; lanes are only joined at the end of the list)
    buckets = dp.bucket_layers(plan.layers)
    for i, name in enumerate(dn._bwd_layers):
        if name is None:
            continue
        k = next(j for j, b in enumerate(buckets) if name in b)
        lane = dn._bwd_recs[i][2] if len(dn._bwd_recs[i]) > 2 else 1
        assert any(ml == lane and k in ks and dn._bwd_recs.index(dn._bwd_recs[i]) < pos for pos, ml, ks in ol.marks), (name, lane)
    # the exchange waits for ALL marks of the buckets of a unit
    ex = dp.GradExchange(2, dp.bucket_ranges(plan.layers, plan.nparams, plan.nparams), torch.device("cpu"))
    ex.groups = [list(g) for g in ol.coincident]
    assert [(lo, hi, ks) for lo, hi, ks in ex._units()] == [(1083456, 1269129, [0]), (0, 1083456, [3, 2, 1])]

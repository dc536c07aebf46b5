"""CPU test of the backward list the engine emits (ssdn/hip/engine.py::DeviceNet._group_reductions): the planner's list re-ordered for the
merged launches (slab reductions per gradient bucket, small-layer weight gradients per bucket, chained main-lane runs, main-lane weight
gradients).  Legal re-orderings only DELAY side-lane work; these are the invariants the executor's lane semantics rely on."""
import pytest

from ssdn.hip import engine as E
from ssdn.hip.dp import bucket_layers
from ssdn.hip.graph import NetPlan


@pytest.mark.parametrize("cin,cout,bs,B,P", [(3, 9, True, 32, 64), (3, 3, False, 4, 64), (1, 2, True, 2, 32), (3, 9, True, 16, 128)])
def test_backward_list_order_invariants(cin, cout, bs, B, P):
    plan = NetPlan("m/", cin, cout, bs, B, P, P, cus=256)
    recs = [(op.type, i) for i, op in enumerate(plan.bwd)]               # the "argument struct" of record i is its planner index
    out, names = E.DeviceNet._group_reductions(plan, recs)
    assert len(out) == len(recs) == len(names)
    pos = {r[1]: k for k, r in enumerate(out)}
    assert sorted(pos) == list(range(len(recs))), "the emitted list is a permutation of the planner's"
    side = lambda i: plan.bwd[i].type in ("wgrad", "wreduce")            # noqa: E731
    main_lane = lambda k: (len(out[k]) > 2 and out[k][2] == 0) or not side(out[k][1])   # noqa: E731
    # 1. main-lane ops keep their order
    mains = [i for i in range(len(recs)) if not side(i)]
    assert [pos[i] for i in mains] == sorted(pos[i] for i in mains)
    # 2. a side-lane op is only ever delayed: it stays behind every main-lane op that preceded it in the planner's list
    last_main = -1
    for i in range(len(recs)):
        if not side(i):
            last_main = i
        elif last_main >= 0:
            assert pos[i] > pos[last_main], "op %d (%s %s) moved in front of its producer" % (i, plan.bwd[i].type, plan.bwd[i].a["layer"])
    # 3. every reduction follows all weight-gradient launches of its layer, and `names` marks exactly the reductions
    for i, op in enumerate(plan.bwd):
        if op.type == "wreduce":
            assert names[pos[i]] == op.a["layer"]
            for j, oj in enumerate(plan.bwd):
                if oj.type == "wgrad" and oj.a["layer"] == op.a["layer"]:
                    assert pos[j] < pos[i]
        else:
            assert names[pos[i]] is None
    # 4. the reductions of a gradient bucket (per-layer plans) / of a chip-wide launch group (graph.WGRAD_MEGA) are one consecutive run
    #    (the executor merges a run into two launches)
    buckets = bucket_layers(plan.layers)
    if getattr(plan, "_mega_ops", None):
        groups = {}
        for l in plan.layers:
            groups.setdefault(plan.wgrad_group_of(l.name), set()).add(l.name)
        buckets = list(groups.values())
        # ... and a group's weight-gradient records are consecutive too (ONE k_wgrad_mega launch), directly in front of its reductions
        for b in buckets:
            kw = sorted(pos[i] for i, op in enumerate(plan.bwd) if op.type == "wgrad" and op.a["layer"] in b)
            kr = sorted(pos[i] for i, op in enumerate(plan.bwd) if op.type == "wreduce" and op.a["layer"] in b)
            assert kw == list(range(kw[0], kw[0] + len(kw))) and kr[0] == kw[-1] + 1
    for b in buckets:
        ks = sorted(pos[i] for i, op in enumerate(plan.bwd) if op.type == "wreduce" and op.a["layer"] in b)
        if ks:
            assert ks == list(range(ks[0], ks[0] + len(ks)))
    # 5. weight gradients sent to the main lane come after the last data-gradient launch of their bucket, in front of its reductions
    for i, op in enumerate(plan.bwd):
        if op.type == "wgrad" and op.a["layer"] in E.MAIN_LANE_WGRADS:
            assert out[pos[i]][2] == 0
            assert all(pos[j] < pos[i] for j in mains if j < max(k for k, o in enumerate(plan.bwd) if o.type == "wgrad" and o.a["layer"] == op.a["layer"]))
    # 6. the chainable main-lane ops of the encoder end form one run without side-lane records in between
    if P == 64 and B * (4 if bs else 1) >= 16:
        chain = [i for i, op in enumerate(plan.bwd)
                 if (op.type == "conv" and op.a["role"] == "dgrad" and len(op.a["taps"]) == 9 and op.a["H"] * op.a["W"] <= 64) or
                 (op.type == "pool_bwd" and (op.a["H"] // 2) * (op.a["W"] // 2) <= 64)]
        assert len(chain) >= 9
        tail = chain[3:]                                                 # (the decoder bucket's merged launch stays behind the third op)
        ks = [pos[i] for i in tail]
        between = [k for k in range(ks[0], ks[-1] + 1) if k not in ks]
        assert all(main_lane(k) for k in between), "side-lane records inside the chained run"


@pytest.mark.parametrize("cin,cout,bs,B,P", [(3, 9, True, 32, 64), (3, 3, False, 4, 64), (1, 2, True, 2, 32), (3, 9, True, 16, 128)])
def test_bucket_marks_cover_every_lane_that_reduced_the_bucket(cin, cout, bs, B, P):
    """ADVICE round 4 (data-parallel race): the collective of a gradient bucket reads the bucket's range of the flat gradient, which slab
    reductions on SEVERAL lanes may have written (plan "split": decode_block_2.2's on the side lane, the rest of its bucket on the main
    lane) -- and the executor orders no lane after another before the end-of-list join.  So every reduction of a bucket must be followed,
    on its OWN lane, by an SSDN_OP_EVENT_RECORD the bucket's collective waits for."""
    import torch
    from ssdn.hip import lib as L
    plan = NetPlan("m/", cin, cout, bs, B, P, P, cus=256)
    flat = torch.zeros(plan.nparams)
    dn = E.DeviceNet(plan, torch.device("cpu"), flat, torch.zeros_like(flat))
    buckets = bucket_layers(plan.layers)
    waits = {}

    def new_event(ks):
        h = 7000 + len(waits)
        waits[h] = list(ks)
        return h
    ol = dn.bwd_with_events(buckets, new_event)
    ev = L.OP["event_record"]
    recs = [(int(ol.arr[i].type), int(ol.arr[i].lane)) for i in range(ol.n)]
    mark_at = {pos: (lane, ks) for pos, lane, ks in ol.marks}
    assert all(recs[pos] == (ev, lane) for pos, (lane, _) in mark_at.items()) and sum(1 for t, _ in recs if t == ev) == len(mark_at)
    # walk the emitted list: reductions in order of dn._bwd_recs, marks interleaved
    j = -1
    for pos, (t, lane) in enumerate(recs):
        if t == ev:
            continue
        j += 1
        name = dn._bwd_layers[j]
        if name is None:
            continue
        k = next(i for i, b in enumerate(buckets) if name in b)
        assert any(p > pos and ml == lane and k in ks for p, (ml, ks) in mark_at.items()), \
            "reduction of %s (bucket %d, lane %d) has no later mark of its bucket on its lane" % (name, k, lane)
    # every bucket with parameters is complete somewhere
    assert sorted(k for ks in ol.coincident for k in ks) == [k for k, b in enumerate(buckets) if b]

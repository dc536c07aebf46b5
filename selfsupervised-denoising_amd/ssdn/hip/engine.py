"""Device side of the op-list plans: owns the HBM buffers (allocated through torch, used as a plain device allocator),
materialises `ssdn.hip.graph` Op records into the C structs of include/ssdn_hip.h and executes them with ONE call
into libssdn_hip.so per phase (forward / backward / optimiser).  No torch compute op is on this path.
"""
from __future__ import annotations

import os
import ctypes as C
import math
from typing import Dict, List, Optional

import torch

from . import lib as L
from .graph import NetPlan, Op, View


def _ptr(t: torch.Tensor, byte_off: int = 0) -> int:
    return t.data_ptr() + byte_off


class OpList:
    """A materialised op list: a ctypes array of ssdn_op plus the argument structs it points to."""

    # weight-gradient GEMMs / slab reductions run on the library's side streams.  (The executor also offers lane 3, a second
    # weight-gradient lane; alternating the GEMMs between lanes 1 and 3 measured 4 % SLOWER: two of these kernels, each
    # sized to own every CU, only thrash each other's LDS-resident pipelines.)
    # The slab reduction follows its GEMM on the same lane: every cross-lane dependency is a hipEventRecord on the producing
    # stream, which costs that stream ~5-10 us of bubble -- a separate reduction lane (2) measured 4.5 % slower.
    LANE = {"wgrad": (1,), "wreduce": (1,)}

    def __init__(self, recs, lanes: bool = False):
        self.args = [r[1] for r in recs]                      # keep the structs alive
        self.arr = (L.OpRec * max(1, len(recs)))()
        count = {}
        for i, r in enumerate(recs):
            ty, a = r[0], r[1]
            self.arr[i].type = L.OP[ty]
            choices = self.LANE.get(ty, (0,)) if lanes else (0,)
            self.arr[i].lane = choices[count.get(ty, 0) % len(choices)]
            if lanes and len(r) > 2:                      # (type, args, lane): the planner chose the lane of this record itself
                self.arr[i].lane = r[2]
            count[ty] = count.get(ty, 0) + 1
            self.arr[i].args = C.cast(C.pointer(a), C.c_void_p)
        self.n = len(recs)

    def run(self, stream: int = 0):
        if self.n:
            L.check(L.load().ssdn_run_ops(self.arr, self.n, C.c_void_p(stream)))


def current_stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class DeviceNet:
    """One NoiseNetwork instance on the device for a fixed input shape: buffers + materialised fwd/bwd/pack op lists."""

    DT = {"act": torch.float16, "actb": torch.bfloat16, "f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32,
          "u32": torch.int32, "i64": torch.int64, "u8": torch.uint8}

    def __init__(self, plan: NetPlan, device, params: torch.Tensor, grads: Optional[torch.Tensor],
                 shared: Optional[Dict[str, torch.Tensor]] = None):
        self.plan, self.device = plan, device
        self.params, self.grads = params, grads
        self.t: Dict[str, torch.Tensor] = dict(shared or {})
        for name, spec in plan.tensors.items():
            if name in self.t:
                continue
            # zero-initialised once: padding channels / never-written slab corners must be finite, and the rows of `u` that the one-row
            # shift of the un-rotation leaves empty are never written by the fused SSDN_OP_UNROT_FWD (ssdn_conv_args.urot): they ARE these zeros
            self.t[name] = torch.zeros(spec.shape, dtype=self.DT[spec.kind], device=device)
        self.fwd = OpList([self._mat(op) for op in plan.fwd])
        self._bwd_recs, self._bwd_layers = self._group_reductions(plan, [self._mat(op) for op in plan.bwd])
        self.bwd = OpList(self._bwd_recs, lanes=True)
        # the last gradient bucket's slab reductions close the list: a training step may run them in front of its optimiser launch
        # instead (DeviceEngine.backward(defer_tail=True)): [weight-gradient lane: wait for the main lane's last launch, two
        # reduction launches] -> join -> Adam becomes join -> reductions -> Adam on ONE stream (one cross-lane hop less in the serial tail)
        ntail = 0
        while ntail < len(self._bwd_recs) and self._bwd_recs[len(self._bwd_recs) - 1 - ntail][0] == "wreduce":
            ntail += 1
        self.tail_recs = [(r[0], r[1]) for r in self._bwd_recs[len(self._bwd_recs) - ntail:]] if 0 < ntail < len(self._bwd_recs) else []
        self.bwd_head = OpList(self._bwd_recs[:len(self._bwd_recs) - len(self.tail_recs)], lanes=True) if self.tail_recs else None
        self.pack = OpList([self._mat(op) for op in plan.pack])

    def bwd_with_events(self, buckets, new_event):
        """Backward op list with SSDN_OP_EVENT_RECORD marks for the gradient exchange (ssdn.hip.dp).  buckets: list of sets of layer
        names; new_event(bucket indexes) -> raw hipEvent_t handle (int) that the all-reduce of those buckets will wait for.
        A bucket's slab reductions may run on SEVERAL lanes (plan "split": decode_block_2.2's on the side lane, the rest of its bucket
        on the main lane, and the executor orders no lane after another before the end-of-list join): the bucket gets one mark per
        lane, each on that lane right behind the bucket's last reduction THERE, and its collective waits for all of them
        (one mark on the lane of the last reduction in LIST order left the other lane's writes unordered).
        Marks that fall on the same (point of the list, lane) share one event."""
        lane_of = lambda r: r[2] if len(r) > 2 else OpList.LANE.get(r[0], (0,))[0]     # noqa: E731  (what OpList(lanes=True) assigns)
        last = {}                                              # (bucket, lane) -> index of the bucket's last reduction on that lane
        for i, name in enumerate(self._bwd_layers):
            if name is None:
                continue
            for k, b in enumerate(buckets):
                if name in b:
                    last[(k, lane_of(self._bwd_recs[i]))] = i
        # (a mark inside a run of consecutive reductions of one lane would split the run -- the executor merges a run into two
        #  launches --: it moves to the end of the run; buckets that end in the same run are then marked at the same point)
        marks = {}                                             # (index, lane) -> buckets
        for (k, lane), i in last.items():
            while i + 1 < len(self._bwd_recs) and self._bwd_recs[i + 1][0] == "wreduce" and lane_of(self._bwd_recs[i + 1]) == lane:
                i += 1
            marks.setdefault((i, lane), []).append(k)
        recs, self.marks = [], []
        for i, r in enumerate(self._bwd_recs):
            recs.append(r)
            for (idx, lane), ks in sorted(marks.items()):
                if idx == i:
                    recs.append(("event_record", L.EventArgs(new_event(sorted(ks))), lane))
                    self.marks.append((len(recs) - 1, lane, sorted(ks)))
        ol = OpList(recs, lanes=True)
        # buckets whose LAST marks sit at the same point of the list complete together: the exchange may merge their collectives
        done_at = {}
        for (idx, lane), ks in marks.items():
            for k in ks:
                done_at[k] = max(done_at.get(k, -1), idx)
        by_pos = {}
        for k, idx in done_at.items():
            by_pos.setdefault(idx, []).append(k)
        ol.coincident = [sorted(ks) for _, ks in sorted(by_pos.items())]
        ol.marks = list(self.marks)
        return ol

    @staticmethod
    def _order_mega(plan, recs):
        """Backward list of a plan whose weight gradients run as chip-wide launches (graph.WGRAD_MEGA): the main-lane ops in the
        planner's order; the SSDN_OP_WGRAD records of a launch group consecutive (the executor runs such a run as ONE k_wgrad_mega
        launch), placed behind the main-lane op that produces the group's last gradient operand, directly followed by the group's
        slab reductions (a run of reductions is two launches).  A group that would fall inside a run of chainable main-lane ops
        (k_conv_chain) waits behind the run.  Returns (records, layer name of each record if it is a reduction else None)."""
        side = lambda op: op.type in ("wgrad", "wreduce")   # noqa: E731
        gof = lambda op: plan.wgrad_group_of(op.a["layer"], plan.is_skip_half(op))  # noqa: E731
        flush = {}
        for i, op in enumerate(plan.bwd):
            if op.type == "wgrad":
                flush[gof(op)] = i
        lane_of = {}
        for g in list(flush):
            _, lane, after = plan.wgrad_group_info(g)
            lane_of[g] = MEGA_LANE if lane is None else lane
            if after is not None:              # behind that layer's data-gradient launch (never in front of the group's last operand)
                at = [i for i, op in enumerate(plan.bwd) if op.type == "conv" and op.a["role"] == "dgrad" and op.a["layer"] == after]
                if at:
                    flush[g] = max(flush[g], at[0])
        def chainable(op):
            if op.type == "conv":
                px = op.a["H"] * op.a["W"]
                thin = px == 256 and op.a["Mpad"] <= 64 and op.a["Ktot"] <= 48 and op.a.get("upsum") is None
                return op.a["role"] == "dgrad" and len(op.a["taps"]) == 9 and (px <= 64 or thin)
            return op.type == "pool_bwd" and (op.a["H"] // 2) * (op.a["W"] // 2) <= 256
        windows, cur = [], None
        for i, op in enumerate(plan.bwd):
            if chainable(op):
                cur = [i, i] if cur is None else [cur[0], i]
            elif not side(op) and cur is not None:
                windows.append(cur)
                cur = None
        if cur is not None:
            windows.append(cur)
        for first, end in windows:
            for g in flush:
                if first <= flush[g] < end:
                    flush[g] = end
        out, names = [], []
        for i, (op, rec) in enumerate(zip(plan.bwd, recs)):
            if not side(op):
                out.append(rec)
                names.append(None)
            for g in sorted(flush):
                if flush[g] != i:
                    continue
                for op2, rec2 in zip(plan.bwd, recs):
                    if op2.type == "wgrad" and gof(op2) == g:
                        out.append((rec2[0], rec2[1], lane_of[g]))
                        names.append(None)
                for op2, rec2 in zip(plan.bwd, recs):
                    if op2.type == "wreduce" and gof(op2) == g:
                        out.append((rec2[0], rec2[1], lane_of[g]))
                        names.append(op2.a["layer"])
        return out, names

    @staticmethod
    def _group_reductions(plan, recs):
        if getattr(plan, "_mega_ops", None):
            return DeviceNet._order_mega(plan, recs)
        return DeviceNet._group_reductions_lanes(plan, recs)

    @staticmethod
    def _group_reductions_lanes(plan, recs):
        """Move every slab reduction to the end of its reduction run (ssdn.hip.dp.bucket_layers(split_head=False): head+dec1 | dec2..dec5 |
        encoder): the executor merges a run of consecutive SSDN_OP_WREDUCE ops into two launches, instead of two launches per
        layer (51 latency-bound launches, 0.52 ms per step in situ).  Legal: every weight-gradient launch owns its slab and
        the flat gradient is only read after the backward list.  Returns (records, layer name of each record if it is a
        reduction else None)."""
        from .dp import bucket_layers
        if not plan.bwd:
            return recs, [op.a.get("layer") if op.type == "wreduce" else None for op in plan.bwd]
        buckets = bucket_layers(plan.layers, split_head=False)
        bucket_of = {name: k for k, b in enumerate(buckets) for name in b}
        # The weight-gradient GEMMs of the layers at 16x16 pixels and below are issued together, where the last of them
        # stood: the executor runs a run of consecutive small SSDN_OP_WGRAD ops as ONE launch (k_wgrad_multi; same rule as
        # csrc/wgrad_mfma.hip::wgrad_mergeable).  Legal for the same reason: a side-lane op may always be delayed.
        from .graph import WGRAD_SMALL_PX
        small = lambda op: op.type == "wgrad" and op.a["N"] * op.a["H"] * op.a["W"] <= WGRAD_SMALL_PX  # noqa: E731
        # (one run per gradient bucket, so that a bucket still completes where it did)
        flush_at = {}                                   # index of the last small wgrad of each bucket
        for i, op in enumerate(plan.bwd):
            if small(op):
                flush_at[bucket_of[op.a["layer"]]] = i
        nsmall = {k: sum(1 for op in plan.bwd if small(op) and bucket_of[op.a["layer"]] == k) for k in flush_at}
        flush_at = {k: i for k, i in flush_at.items() if nsmall[k] > 1}
        grouped = lambda op: small(op) and bucket_of[op.a["layer"]] in flush_at  # noqa: E731
        last = {}
        for i, op in enumerate(plan.bwd):
            if op.type in ("wgrad", "wreduce"):
                k = bucket_of[op.a["layer"]]
                last[k] = max(last.get(k, -1), flush_at[k] if grouped(op) else i)
        # The main-lane ops on images of <= 64 pixels (data gradients, max-pool backward) run as ONE launch when they are consecutive
        # in the list (k_conv_chain, csrc/conv_chain.hip: the library's rule): side-lane records that would fall between them are
        # emitted behind the last of them (a side-lane op may always be delayed).
        def chainable(op):
            if op.type == "conv":
                px = op.a["H"] * op.a["W"]
                thin = px == 256 and op.a["Mpad"] <= 64 and op.a["Ktot"] <= 48 and op.a.get("upsum") is None
                return op.a["role"] == "dgrad" and len(op.a["taps"]) == 9 and (px <= 64 or thin)
            return op.type == "pool_bwd" and (op.a["H"] // 2) * (op.a["W"] // 2) <= 256
        windows, cur = [], None
        for i, op in enumerate(plan.bwd):
            if chainable(op):
                cur = [i, i] if cur is None else [cur[0], i]
            elif op.type not in ("wgrad", "wreduce") and cur is not None:
                windows.append(cur)
                cur = None
        if cur is not None:
            windows.append(cur)
        for first, end in windows:
            for d in (flush_at, last):
                for k in d:
                    if first <= d[k] < end and (CHAIN_POSTPONES is None or k in CHAIN_POSTPONES):
                        d[k] = end
        out, names, pending = [], [], {k: [] for k in range(len(buckets))}
        pending_small = {k: [] for k in range(len(buckets))}
        pending_main = {k: [] for k in range(len(buckets))}
        # SSDN_OP_GRAD_PACK and the narrow net_out layer's data gradient behind it are ONE launch when adjacent (csrc/gradpack_dgrad.hip): the
        # weight-gradient record the planner puts between them waits behind the pair (it needs the pair's output anyway)
        hold_to, held = -1, []
        if plan.bwd[0].type == "grad_pack":
            j = next((k for k in range(1, len(plan.bwd)) if plan.bwd[k].type not in ("wgrad", "wreduce")), -1)
            if j > 1 and plan.bwd[j].type == "conv" and plan.bwd[j].a["role"] == "dgrad" and len(plan.bwd[j].a["taps"]) == 1 and plan.bwd[j].a["Ktot"] == 16:
                hold_to = j
        for i, (op, rec) in enumerate(zip(plan.bwd, recs)):
            if op.type == "wreduce":
                pending[bucket_of[op.a["layer"]]].append((rec, op.a["layer"]))
            elif op.type == "wgrad" and op.a["layer"] in MAIN_LANE_WGRADS:
                pending_main[bucket_of[op.a["layer"]]].append((rec[0], rec[1], 0))
            elif grouped(op):
                pending_small[bucket_of[op.a["layer"]]].append(rec)
            elif op.type == "wgrad" and i < hold_to:
                held.append(rec)
            else:
                out.append(rec)
                names.append(None)
                if i == hold_to:
                    out += held
                    names += [None] * len(held)
                    held = []
            for k in sorted(set(flush_at) | set(last)):          # bucket by bucket: a bucket's merged launch, then its reductions
                if flush_at.get(k) == i:
                    out += pending_small[k]
                    names += [None] * len(pending_small[k])
                    pending_small[k] = []
                if not SPLIT_SMALL_RUNS:
                    continue
                if last.get(k) == i:
                    out += pending_main[k]
                    names += [None] * len(pending_main[k])
                    pending_main[k] = []
                    for rec_k, name_k in pending[k]:
                        out.append(rec_k)
                        names.append(name_k)
                    pending[k] = []
            for k in sorted(last):
                if last[k] == i and not SPLIT_SMALL_RUNS:
                    out += pending_main[k]
                    names += [None] * len(pending_main[k])
                    pending_main[k] = []
                    for rec_k, name_k in pending[k]:
                        out.append(rec_k)
                        names.append(name_k)
                    pending[k] = []
        return out, names

    # ---- helpers ------------------------------------------------------------------------------------------
    def tensor(self, short: str) -> torch.Tensor:
        return self.t[self.plan.prefix + short]

    def _view(self, v: Optional[View]) -> L.View:
        if v is None:
            return L.View(None, 0, 0)
        t = self.t[v.t]
        return L.View(_ptr(t), t.shape[-1], v.co)

    def _pp(self, off_floats: int) -> int:
        return _ptr(self.params, 4 * (self.plan.param_base + off_floats))

    def _gp(self, off_floats: int) -> int:
        return _ptr(self.grads, 4 * (self.plan.param_base + off_floats))

    def _layer(self, name):
        return next(l for l in self.plan.layers if l.name == name)

    def _mat(self, op: Op):
        a = op.a
        P = self.plan.prefix
        if op.type == "pack_input":
            return op.type, L.PackInputArgs(_ptr(self.t[a["src"]]), self._view(a["dst"]), a["B"], a["C"], a["H"], a["W"], a["R"], a["cpad"])
        if op.type == "conv":
            l = self._layer(a["layer"])
            s = L.ConvArgs()
            s.src0, s.src1 = self._view(a["src0"]), self._view(a["src1"])
            s.c0, s.c1, s.up0 = a["c0"], a["c1"], a["up0"]
            s.N, s.H, s.W = a["N"], a["H"], a["W"]
            s.ntaps = len(a["taps"])
            for i, (dy, dx) in enumerate(a["taps"]):
                s.dy[i], s.dx[i] = dy, dx
            s.w = _ptr(self.t[P + ("wf/" if a["role"] == "fwd" else "wd/") + a["layer"]])
            wc = self.t.get(P + ("wfc/" if a["role"] == "fwd" else "wdc/") + a["layer"])
            s.wc = _ptr(wc) if wc is not None else None
            s.M, s.Mpad, s.Ktot = a["M"], a["Mpad"], a["Ktot"]
            s.bias = self._pp(l.b_off) if a["bias"] else None
            s.act = a["act"]
            s.mask, s.add, s.dst = self._view(a["mask"]), self._view(a["add"]), self._view(a["dst"])
            s.dst32 = _ptr(self.t[a["dst32"]]) if a["dst32"] is not None else None
            s.ltw, s.lth, s.ltn, s.kc = a["ltw"], a["lth"], a["ltn"], a["kc"]
            s.bf16 = a["bf16"]
            s.kreal = a.get("kreal", 0)
            s.pool, s.pool_shifted = self._view(a.get("pool")), a.get("pool_shifted", 0)
            if a.get("pool") is not None and not L.load().ssdn_conv_fuses_pool(C.byref(s)):
                raise L.SsdnHipError("conv %s: the plan fuses the max-pool but the library cannot (planner / library rule mismatch)" % a["layer"])
            s.upsum, s.upsum_mask, s.upsum_c = self._view(a.get("upsum")), self._view(a.get("upsum_mask")), a.get("upsum_c", 0)
            if a.get("upsum") is not None and not L.load().ssdn_conv_fuses_upsum(C.byref(s)):
                raise L.SsdnHipError("conv %s: the plan fuses UPSUM_BWD but the library cannot (planner / library rule mismatch)" % a["layer"])
            s.unrot, s.unrot_mask = self._view(a.get("unrot")), self._view(a.get("unrot_mask"))
            s.unrot_smask = self.t[a["unrot_smask"]].data_ptr() if a.get("unrot_smask") else None
            s.sign_out = self.t[a["sign_out"]].data_ptr() if a.get("sign_out") else None
            s.mask_sign = self.t[a["mask_sign"]].data_ptr() if a.get("mask_sign") else None
            s.upsum_mask_sign = self.t[a["upsum_mask_sign"]].data_ptr() if a.get("upsum_mask_sign") else None
            if (a.get("sign_out") or a.get("mask_sign") or a.get("upsum_mask_sign")) and not L.load().ssdn_conv_signs(C.byref(s)):
                raise L.SsdnHipError("conv %s: the plan uses sign bytes but the library cannot (planner / library rule mismatch)" % a["layer"])
            s.urot = self._view(a.get("urot"))
            s.urot_smask = self.t[a["urot_smask"]].data_ptr() if a.get("urot_smask") else None
            if a.get("urot") is not None and not L.load().ssdn_conv_fuses_urot(C.byref(s)):
                raise L.SsdnHipError("conv %s: the plan fuses UNROT_FWD but the library cannot (planner / library rule mismatch)" % a["layer"])
            if a.get("unrot") is not None and not L.load().ssdn_conv_fuses_unrot(C.byref(s)):
                raise L.SsdnHipError("conv %s: the plan fuses UNROT_BWD but the library cannot (planner / library rule mismatch)" % a["layer"])
            if L.load().ssdn_conv_lds_bytes(C.byref(s)) < 0:
                raise L.SsdnHipError("conv %s/%s: %s" % (a["layer"], a["role"], L.load().ssdn_last_error().decode()))
            return op.type, s
        if op.type in ("pool_fwd", "pool_bwd"):
            return op.type, L.PoolArgs(self._view(a["act"]), self._view(a.get("pooled")), self._view(a.get("dpool")),
                                       self._view(a.get("dz")), a["N"], a["H"], a["W"], a["C"], a["shifted"],
                                       self.t[a["route"]].data_ptr() if a.get("route") else None)
        if op.type == "upsum_bwd":
            return op.type, L.UpsumArgs(self._view(a["src"]), self._view(a["mask"]), self._view(a["dst"]), a["N"], a["H"], a["W"], a["C"])
        if op.type in ("unrot_fwd", "unrot_bwd"):
            return op.type, L.UnrotArgs(self._view(a["src"]), self._view(a["dst"]), self._view(a.get("mask")), a["B"], a["P"], a["C"],
                                        self.t[a["smask"]].data_ptr() if a.get("smask") else None)
        if op.type == "wgrad":
            s = L.WgradArgs()
            s.dz, s.src0, s.src1 = self._view(a["dz"]), self._view(a["src0"]), self._view(a["src1"])
            s.c0, s.c1, s.up0 = a["c0"], a["c1"], a["up0"]
            s.N, s.H, s.W = a["N"], a["H"], a["W"]
            s.ntaps = len(a["taps"])
            for i, (dy, dx) in enumerate(a["taps"]):
                s.dy[i], s.dx[i] = dy, dx
                s.coff[i] = a["coff"][i]
            s.M, s.Mpad, s.Ktot, s.Kpad = a["M"], a["Mpad"], a["Ktot"], a["Kpad"]
            s.slab, s.bslab = _ptr(self.t[a["slab"]]), _ptr(self.t[a["bslab"]])
            s.nslabs = a["nslabs"]
            s.ltw, s.lth, s.ltn = a["ltw"], a["lth"], a["ltn"]
            s.csplit = a.get("csplit", 0)
            s.mblocks = a.get("mblocks", 1)
            s.kreal = a.get("kreal", 0)
            s.mega, s.cost = a.get("mega", 0), a.get("cost", 0.0)
            if L.load().ssdn_wgrad_lds_bytes(C.byref(s)) < 0:
                raise L.SsdnHipError("wgrad %s: %s" % (a["layer"], L.load().ssdn_last_error().decode()))
            return op.type, s
        if op.type == "wreduce":
            l = self._layer(a["layer"])
            # (a merged weight-gradient launch owns mblocks consecutive groups of nslabs slabs: this reduction reads group mblock)
            mb = a.get("mblock", 0)
            soff = 4 * mb * a["nslabs"] * a["ntaps"] * a["Mpad"] * a["Kpad"]
            boff = 4 * mb * a["nslabs"] * a["Mpad"]
            return op.type, L.WreduceArgs(_ptr(self.t[a["slab"]], soff), _ptr(self.t[a["bslab"]], boff), a["nslabs"], a["ntaps"], a["M"],
                                          a["Mpad"], a["Kpad"], a["cin"], a["cin_full"], a["m_off"], a["c_off"], a["tapblock"],
                                          self._gp(l.w_off), self._gp(l.b_off) if a["with_bias"] else None,
                                          _ptr(self.t[P + "scale"], 4))
        if op.type == "wpack":
            l = self._layer(a["layer"])
            return op.type, L.WpackArgs(self._pp(l.w_off), _ptr(self.t[P + "wf/" + l.name]),
                                        _ptr(self.t[P + "wd/" + l.name]) if a["need_d"] else None, a["M"], a["cin"], a["ntaps"],
                                        a["c0"], a["c1_real"], a["Mpad_f"], a["Ktot"], a["Mpad_d"], a["Kd"],
                                        _ptr(self.t[P + "wfc/" + l.name]) if a.get("cm_f") else None,
                                        _ptr(self.t[P + "wdc/" + l.name]) if a.get("cm_d") else None)
        if op.type == "grad_pack":
            return op.type, L.GradPackArgs(_ptr(self.t[a["g"]]), self._view(a["dst"]), a["N"], a["C"], a["H"], a["W"], a["cpad"],
                                           _ptr(self.t[P + "gmax"]), _ptr(self.t[P + "scale"]))
        raise ValueError("unknown op type " + op.type)


# Weight-gradient GEMMs that run on the MAIN lane, after the last data-gradient launch of their gradient bucket, instead of the
# weight-gradient lane: the backward pass ends with lane 1 still working through its queue (150 us) while lane 0 has nothing left.
# Measured on the bench workload (tools/ab_lanes.py, same process, with the chained small layers): encode_block_1.0 alone 2.031 ms
# per step (2.056 with none), with encode_block_1.2 1.985, with encode_block_2.0 instead 2.005, all three 2.034.  Same kernels,
# same slabs: bit-identical.
MAIN_LANE_WGRADS = ("encode_block_1.0", "encode_block_1.2")
# True: when the merged small-layer launches of two gradient buckets end up next to each other in the list, the first bucket's
# reductions go between them (two k_wgrad_multi launches); False: one launch for both, then both buckets' reductions.
SPLIT_SMALL_RUNS = True
# Denoiser.train_step without a gradient exchange leaves the last bucket's slab reductions to the optimiser call (DeviceEngine.backward)
DEFER_TAIL = True
# Gradient buckets whose side-lane records are moved behind a run of chainable main-lane ops (None: all).  Moving the decoder
# bucket's merged weight-gradient launch behind the whole run makes ONE chain of 12 ops, but that launch then waits for the end of the
# chain; leaving it where the bucket's last gradient appears makes two chains (3 + 9 ops) and starts it ~100 us earlier.
CHAIN_POSTPONES = (2,)          # measured (tools/ab_lanes.py, same process): all buckets 1.975 ms per step, the encoder bucket only 1.959, none 1.965

# Pixels of one sample a workgroup of the loss-head kernel walks (256 threads): 1024 = four pixels per thread in sequence.  (One pixel per
# thread used to be slower, 33 vs 20 us, because every WAVE ended with an atomic on the gradient sentinel; the kernel now issues one per block.)
HEAD_PX_PER_BLOCK = 1024

# Lane of the chip-wide weight-gradient launches and their slab reductions (graph.WGRAD_MEGA).  A workgroup of those launches owns its
# CU, one per CU: next to them nothing else runs, so they sit on the main lane (0), behind the data gradients whose results they read.
MEGA_LANE = 0

STYLE = {"gauss": 0, "poisson": 1}
MODE = {"known": 0, "const": 1, "var": 2}


class DenoiserEngine:
    """Everything one Denoiser configuration needs on one GPU for one input shape: the main net, the optional sigma
    estimator, the loss head, the fused Adam pass -- as four op lists (fwd+loss, bwd, optimiser, eval)."""

    def __init__(self, pipeline: str, channels: int, blindspot: bool, style: str, mode: str, B: int, H: int, W: int,
                 device, params: torch.Tensor, grads: torch.Tensor, adam_m: torch.Tensor, adam_v: torch.Tensor,
                 n_main: int, n_sigma: int, has_est: bool, train: bool = True, ncoords: int = 64):
        self.pipeline, self.C, self.blindspot, self.mode = pipeline, channels, blindspot, mode
        self.style = "poisson" if style.startswith("poisson") else "gauss"
        self.B, self.H, self.W, self.device, self.train = B, H, W, device, train
        self.params, self.grads, self.m, self.v = params, grads, adam_m, adam_v
        lib = L.load()
        cus = lib.ssdn_device_cus()
        if cus <= 0:
            raise L.SsdnHipError("no HIP device: " + lib.ssdn_last_error().decode())
        cout = channels + channels * (channels + 1) // 2 if pipeline == "ssdn" else channels
        f32 = dict(dtype=torch.float32, device=device)
        self.inp = torch.zeros((B, channels, H, W), **f32)
        self.main = DeviceNet(NetPlan("m/", channels, cout, blindspot, B, H, W, cus=cus, train=train, param_base=0),
                              device, params, grads, shared={"m/in32": self.inp})
        self.sigma = None
        if pipeline == "ssdn" and mode == "var":
            self.sigma = DeviceNet(NetPlan("s/", channels, 1, False, B, H, W, cus=cus, train=train, param_base=n_main),
                                   device, params, grads, shared={"s/in32": self.inp})
        self.est_off = n_main + n_sigma if has_est else None
        # loss-side buffers
        self.loss = torch.zeros((B, 1), **f32)
        self.ref = torch.zeros((B, channels, H, W), **f32)
        self.noise_param = torch.zeros((B,), **f32)
        self.coords = torch.zeros((ncoords, 2), dtype=torch.int64, device=device)
        self.ncoords = ncoords
        self.nchunks = max(1, min(64, (H * W) // HEAD_PX_PER_BLOCK))
        self.partial = torch.zeros((B, self.nchunks, 2), **f32)
        self.est_raw = torch.zeros((B,), **f32)
        self.g_est_var = torch.zeros((B,), **f32)
        self.mu = torch.zeros((B, channels, H, W), **f32)
        self.pme = torch.zeros((B, channels, H, W), **f32)
        self.model_std = torch.zeros((B, H, W), **f32)
        self.noise_std = torch.zeros((B, H, W) if self.style == "poisson" else (B,), **f32)
        self.zero_buf = torch.zeros((8,), dtype=torch.int32, device=device)   # unused gmax sink for eval
        self._adam_args = None
        # H11: loss / PSNR / std-dev sums of a step go into a device-resident accumulator (SSDN_OP_METRICS); `per` holds the last
        # batch's per-sample values (the evaluator's per-image PSNR)
        self.metrics_per = torch.zeros((B, 8), **f32)
        self._metrics_args = None
        self.ops_loss = OpList(self._loss_ops(want_grad=train))
        opt_recs = self._opt_ops() if train else None          # (both lists share the argument structs adam() updates)
        self.ops_opt = OpList(opt_recs) if train else None
        self.ops_opt_tail = OpList(self.main.tail_recs + opt_recs) if train and self.main.tail_recs else None
        self._tail_pending = False

    # ---- op construction -----------------------------------------------------------------------------------
    def _gmax(self, net: Optional[DeviceNet]):
        if net is not None and self.train:
            return _ptr(net.tensor("gmax"))
        return _ptr(self.zero_buf)

    def _loss_ops(self, want_grad: bool):
        recs = []
        B, Cn, H, W = self.B, self.C, self.H, self.W
        out32 = self.main.tensor("out32")
        g32 = _ptr(self.main.tensor("g32")) if want_grad else None
        # (gmax, the max-|gradient| sentinel the loss kernels fold into with atomicMax, is a RUNNING maximum since the buffers were created:
        #  clearing its 16 bytes every step was a launch of its own in the serial stretch between the forward and the backward pass)
        if self.pipeline == "ssdn":
            est_ptr = None
            if self.mode == "var":
                recs.append(("spatial_mean", L.SpatialMeanArgs(_ptr(self.sigma.tensor("out32")), _ptr(self.est_raw), B, H * W)))
                est_ptr = _ptr(self.est_raw)
            elif self.mode == "const":
                est_ptr = _ptr(self.params, 4 * self.est_off)
            recs.append(("head_ssdn", L.HeadArgs(_ptr(out32), _ptr(self.inp), _ptr(self.noise_param), est_ptr, B, Cn, H, W,
                                                  STYLE[self.style], MODE[self.mode], int(want_grad), _ptr(self.mu), _ptr(self.pme),
                                                  _ptr(self.model_std), _ptr(self.noise_std), g32, _ptr(self.partial),
                                                  self.nchunks, self._gmax(self.main))))
            g_est = None
            if want_grad and self.mode == "const":
                g_est = _ptr(self.grads, 4 * self.est_off)
            elif want_grad and self.mode == "var":
                g_est = _ptr(self.g_est_var)
            recs.append(("head_final", L.HeadFinalArgs(_ptr(self.partial), B, self.nchunks, H, W, MODE[self.mode], _ptr(self.loss), g_est,
                                                        _ptr(self.sigma.tensor("g32")) if (want_grad and self.sigma is not None) else None,
                                                        self._gmax(self.sigma) if self.sigma is not None else None)))
        elif self.pipeline == "mse":
            recs.append(("mse", L.MseArgs(_ptr(out32), _ptr(self.ref), None, 0, B, Cn, H, W, _ptr(self.loss), g32, self._gmax(self.main))))
        elif self.pipeline == "mask_mse":
            recs.append(("mask_mse", L.MseArgs(_ptr(out32), _ptr(self.ref), _ptr(self.coords), self.ncoords, B, Cn, H, W, _ptr(self.loss),
                                               g32, self._gmax(self.main))))
        else:
            raise NotImplementedError("Unsupported processing pipeline")
        return recs

    def _opt_ops(self):
        """Fused Adam over the flat buffer, each range directly followed by the re-packs of the layers inside it: the executor
        runs such a run as ONE launch (k_adam_pack: the thread that updates a weight also writes its fp16 / bf16 shadows)."""
        def adam(lo, hi):
            return L.AdamArgs(_ptr(self.params, 4 * lo), _ptr(self.grads, 4 * lo), _ptr(self.m, 4 * lo), _ptr(self.v, 4 * lo), hi - lo,
                              0.0, 0.9, 0.99, 1e-8, 1.0, 1.0, 1.0)
        n_tot = self.params.numel()
        n_main = self.sigma.plan.param_base if self.sigma is not None else n_tot
        self._adam_args = [adam(0, n_main)]
        recs = [("adam", self._adam_args[0])] + [self.main._mat(op) for op in self.main.plan.pack]
        if self.sigma is not None:
            self._adam_args.append(adam(n_main, n_tot))
            recs += [("adam", self._adam_args[1])] + [self.sigma._mat(op) for op in self.sigma.plan.pack]
        return recs

    # ---- metrics (H11) -------------------------------------------------------------------------------------
    def accumulate_metrics(self, acc: torch.Tensor, clean: torch.Tensor, ext: Optional[torch.Tensor] = None, with_loss: bool = True,
                           stream=None):
        """Enqueue ONE launch that adds this batch's loss / PSNR(out) / PSNR(mu) / noise std / model std sums into `acc` (fp32 [16] on
        the device: sum, count per metric) -- the reference's per-step `Metric +=` lines (train.py:205-218).  clean: fp32 [B,C,H,W] on
        the device; ext: int32 [B,2] valid extents of the two spatial axes (padded evaluation images) or None."""
        a = self._metrics_args
        if a is None:
            a = self._metrics_args = L.MetricsArgs()
            a.B, a.C, a.H, a.W = self.B, self.C, self.H, self.W
            a.per = _ptr(self.metrics_per)
            if self.pipeline == "ssdn":
                a.out, a.mu, a.model_std, a.noise_std = _ptr(self.pme), _ptr(self.mu), _ptr(self.model_std), _ptr(self.noise_std)
                # (gauss: one value per sample -- one for the whole batch when sigma is a learnt constant; poisson: per pixel)
                a.noise_n = self.noise_std.numel() if self.style == "poisson" else (1 if self.mode == "const" else self.B)
            else:
                a.out = _ptr(self.main.tensor("out32"))
            self._metrics_ops = OpList([("metrics", a)])
        if clean.dtype != torch.float32 or not clean.is_contiguous() or tuple(clean.shape) != (self.B, self.C, self.H, self.W):
            raise L.SsdnHipError("metrics: clean must be a contiguous fp32 [B,C,H,W] device tensor of the engine's shape")
        a.clean, a.acc = _ptr(clean), _ptr(acc)
        a.loss = _ptr(self.loss) if with_loss else None
        a.ext = _ptr(ext) if ext is not None else None
        self._metrics_ops.run(current_stream() if stream is None else stream)

    # ---- export: the whole step as a blob libssdn_hip.so runs on its own (csrc/plan.hip, include/ssdn_hip.h "step-level entry points") ----
    def export_plan(self, meta: Optional[dict] = None) -> bytes:
        """The op lists of this engine with every pointer replaced by (tensor, offset), the tensor table, and a JSON description.
        ssdn_plan_load() + ssdn_plan_bind() rebuild them in one caller-owned arena; ssdn_train_step() / ssdn_net_forward() run them --
        a binder needs the library and this blob, not the Python package."""
        import json
        import struct
        tens: List[tuple] = []                       # (name, tensor)
        for name, t in self.main.t.items():
            tens.append((name, t))
        if self.sigma is not None:
            for name, t in self.sigma.t.items():
                tens.append((name, t))
        tens += [("params", self.params), ("grads", self.grads), ("adam_m", self.m), ("adam_v", self.v), ("loss", self.loss), ("ref", self.ref),
                 ("noise_param", self.noise_param), ("coords", self.coords), ("partial", self.partial), ("est_raw", self.est_raw),
                 ("g_est_var", self.g_est_var), ("mu", self.mu), ("pme", self.pme), ("model_std", self.model_std), ("noise_std", self.noise_std),
                 ("zero_buf", self.zero_buf), ("metrics_per", self.metrics_per)]
        tens = [(n, t) for n, t in tens if t is not None]
        table, by_ptr = [], {}
        for name, t in tens:
            nbytes = t.numel() * t.element_size()
            ptr = t.data_ptr()
            alias = by_ptr.get(ptr, -1)
            if alias >= 0 and table[alias][1] < nbytes:
                raise L.SsdnHipError("export_plan: tensor %s aliases a smaller one" % name)
            if alias < 0:
                by_ptr[ptr] = len(table)
            table.append((name, nbytes, alias, ptr))
        spans = sorted((ptr, ptr + nb, i) for i, (n, nb, al, ptr) in enumerate(table) if al < 0)

        def locate(v):
            for lo, hi, i in spans:
                if lo <= v < hi or (v == hi == lo):
                    return i, v - lo
            raise L.SsdnHipError("export_plan: an op argument points outside every tensor of the engine (0x%x)" % v)
        ptr_off = {ty: L.pointer_fields(ty) for ty in set(L.ARG_TYPES.values())}

        def ser(oplists):
            out, n = b"", 0
            for ol in oplists:
                if ol is None:
                    continue
                for i in range(ol.n):
                    a = ol.args[i]
                    raw = bytes(a)
                    rel = b""
                    nr = 0
                    for off in ptr_off[type(a)]:
                        v = struct.unpack_from("<Q", raw, off)[0]
                        if v:
                            ti, d = locate(v)
                            rel += struct.pack("<IIQ", off, ti, d)
                            nr += 1
                    pad = (-len(raw)) % 8
                    out += struct.pack("<iiII", int(ol.arr[i].type), int(ol.arr[i].lane), len(raw), nr) + raw + b"\0" * pad + rel
                    n += 1
            return struct.pack("<I", n) + out
        train = self.train
        phases = [[self.main.pack, self.sigma.pack if self.sigma is not None else None],
                  [self.main.fwd, self.sigma.fwd if self.sigma is not None else None, self.ops_loss],
                  [self.main.bwd, self.sigma.bwd if self.sigma is not None else None] if train else [],
                  [self.ops_opt] if train else []]
        layers = lambda net, base: [dict(name=l.name, w_off=base + l.w_off, b_off=base + l.b_off, M=l.M, cin=l.cin, k=l.k) for l in net.plan.layers]   # noqa: E731
        desc = dict(pipeline=self.pipeline, channels=self.C, blindspot=bool(self.blindspot), style=self.style, mode=self.mode, B=self.B, H=self.H,
                    W=self.W, train=bool(train), nparams=int(self.params.numel()), est_off=self.est_off,
                    layers=layers(self.main, 0) + (layers(self.sigma, self.sigma.plan.param_base) if self.sigma is not None else []))
        desc.update(meta or {})
        mj = json.dumps(desc).encode()
        blob = b"SSDNPLAN" + struct.pack("<IIII", 1, L.ABI_VERSION, len(table), 4)
        for name, nb, al, _ in table:
            nm = name.encode()
            if len(nm) > 55:
                raise L.SsdnHipError("export_plan: tensor name too long: " + name)
            blob += nm + b"\0" * (56 - len(nm)) + struct.pack("<Qii", nb, al, 0)
        for ph in phases:
            blob += ser(ph)
        return blob + struct.pack("<I", len(mj)) + mj

    # ---- execution -----------------------------------------------------------------------------------------
    def repack(self, stream=None):
        s = current_stream() if stream is None else stream
        self.main.pack.run(s)
        if self.sigma is not None:
            self.sigma.pack.run(s)

    # ---- the sigma-estimation network next to the main network (BASELINE config 3) ------------------------------------------------
    # The two networks share only the input and meet in the loss head (reference: ssdn/ssdn/denoiser.py:76-88, 261-265: the estimator
    # reads the noisy image, nothing of the main network).  Enqueued one behind the other on ONE stream (rounds 1-5) the small plain
    # network -- grids of a few dozen workgroups at batch 32 -- ran alone on the chip for 0.86 ms of a 2.5 ms step.  Now its op lists go
    # to a second stream, ordered behind what the caller's stream has enqueued so far and joined back before the consumer (loss head /
    # optimiser): its launches fill the CUs the main network's latency-bound stages leave idle.  Every tensor the two lists write is its
    # own (each DeviceNet owns its buffers; disjoint ranges of the flat gradient), so the results are the sequential order's, bit for bit.
    SIGMA_CONCURRENT = 3          # bit 0: forward lists, bit 1: backward lists (True = both)

    def _fork_sigma(self, s: int, oplist) -> int:
        """run `oplist` (the sigma network's) on the side stream behind everything stream `s` holds; -> the side stream (to join on)"""
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        side = self._side.cuda_stream
        # (the library's own event ring on the RAW handles.  The first version recorded a torch.cuda.Event on torch.cuda.ExternalStream(s) with
        #  s = current_stream().cuda_stream: that wrapper of handle 0 is not ordered behind torch's current stream -- the side stream started
        #  early in ~5 of 12 runs (stale upstream gradients); an event recorded on the torch.cuda.current_stream() OBJECT orders correctly,
        #  12 of 12 (tools/ab_sigma_race.py, profiles/r06_sigma_concurrency.txt).  The engine is handed raw handles, so it orders raw handles.)
        L.check(L.load().ssdn_stream_order(C.c_void_p(s), C.c_void_p(side)))
        oplist.run(side)
        return side

    def _join_sigma(self, s: int, side: int) -> None:
        L.check(L.load().ssdn_stream_order(C.c_void_p(side), C.c_void_p(s)))

    def forward(self, stream=None):
        s = current_stream() if stream is None else stream
        if self.sigma is not None and (int(self.SIGMA_CONCURRENT) & 1):
            ev = self._fork_sigma(s, self.sigma.fwd)
            self.main.fwd.run(s)
            self._join_sigma(s, ev)
        else:
            self.main.fwd.run(s)
            if self.sigma is not None:
                self.sigma.fwd.run(s)
        self.ops_loss.run(s)

    def net_forward_only(self, stream=None):
        s = current_stream() if stream is None else stream
        self.main.fwd.run(s)

    def backward(self, stream=None, exchange=None, defer_tail=False):
        """Enqueue the backward pass.  exchange (ssdn.hip.dp.GradExchange, overlapped): the main net's list then carries one
        event record per gradient bucket; the last bucket (sigma estimator / learnable sigma) is marked after its own list.
        defer_tail (single-GPU training step only): the last bucket's slab reductions are left to the NEXT adam() call, which runs
        them in front of the optimiser launch -- the flat gradient is incomplete until then."""
        s = current_stream() if stream is None else stream
        if defer_tail and exchange is None and self.ops_opt_tail is not None:
            if self.sigma is not None and (int(self.SIGMA_CONCURRENT) & 2):
                ev = self._fork_sigma(s, self.sigma.bwd)
                self.main.bwd_head.run(s)
                self._join_sigma(s, ev)
            else:
                self.main.bwd_head.run(s)
                if self.sigma is not None:
                    self.sigma.bwd.run(s)
            self._tail_pending = True
            return
        self._tail_pending = False
        if exchange is not None and exchange.overlapped:
            from .dp import bucket_layers, N_MAIN_BUCKETS as NB
            if getattr(self, "_bwd_ev_key", None) != id(exchange):
                exchange.reset_marks()
                self._bwd_ev = self.main.bwd_with_events(bucket_layers(self.main.plan.layers), exchange.new_event)
                self._bwd_ev_key = id(exchange)
                if len(exchange.ranges) > NB:
                    exchange.record_here(NB)           # (creates the sigma bucket's mark; re-recorded behind its backward list below)
                # the marks belong to THIS engine's list (an exchange may serve several engines, one per input shape)
                self._bwd_ev_marks = (exchange.events, exchange.waits, exchange._here)
                self._bwd_ev_groups = [list(g) for g in self._bwd_ev.coincident] + [[k] for k in range(NB, len(exchange.ranges))]
            exchange.events, exchange.waits, exchange._here = self._bwd_ev_marks
            exchange.groups = self._bwd_ev_groups
            self._bwd_ev.run(s)
            if self.sigma is not None:
                self.sigma.bwd.run(s)
            if len(exchange.ranges) > NB:
                exchange.record_here(NB)
            return
        if self.sigma is not None and (int(self.SIGMA_CONCURRENT) & 2):
            ev = self._fork_sigma(s, self.sigma.bwd)
            self.main.bwd.run(s)
            self._join_sigma(s, ev)
            return
        self.main.bwd.run(s)
        if self.sigma is not None:
            self.sigma.bwd.run(s)

    def adam(self, lr: float, step: int, gscale: float = 1.0, stream=None):
        for a in self._adam_args:
            a.lr, a.bc1, a.bc2, a.gscale = lr, 1.0 - 0.9 ** step, 1.0 - 0.99 ** step, gscale
        s = current_stream() if stream is None else stream
        if self._tail_pending:                # (the deferred reductions of the last bucket, then the optimiser launch)
            self._tail_pending = False
            self.ops_opt_tail.run(s)
            return
        self.ops_opt.run(s)                   # optimiser step + re-pack of the MFMA shadows (one launch per network)

"""TEST INFRASTRUCTURE -- CPU interpreter of the op list planned by ssdn.hip.graph.NetPlan.

Executes the SAME backend-neutral `Op` records the HIP engine lowers to `ssdn_op` structs, but with plain torch-CPU
float tensors and the op semantics written out naively from include/ssdn_hip.h.  Two uses (tests only):
  * fp16=False: checks the LOWERING (tap offsets, flipped data-gradient taps, packed-weight layouts, pool routing with
    the literal zero row, rotation bookkeeping, skip-gradient accumulation, slab block offsets ...) against the oracle
    (oracle/restate.py + autograd) to ~1e-5, on the CPU, with no GPU involved;
  * fp16=True: rounds every stored activation / gradient / packed weight to fp16 exactly where the device stores fp16,
    which gives the GPU tests a tight comparison target (differences left = fp32 accumulation order only).
Never imported by the product path.
"""
from __future__ import annotations

import math

import numpy as np
import torch

LRELU = 0.1


def _r16(t, on, kind="act"):
    """storage rounding of the device: fp16 for forward activations / forward weights, bf16 for gradients / dgrad weights"""
    if not on:
        return t
    return t.bfloat16().to(t.dtype) if kind in ("actb", "bf16") else t.half().to(t.dtype)


def _lgrad(act):
    return torch.where(act > 0, torch.ones_like(act), torch.full_like(act, LRELU))


def _pack_signs(y):
    """LeakyReLU sign bytes (include/ssdn_hip.h, sign_out / smask): bit q of byte k = (channel 8k+q > 0)"""
    pos = (y > 0).to(torch.int64).reshape(*y.shape[:-1], y.shape[-1] // 8, 8)
    return (pos << torch.arange(8)).sum(-1).to(torch.uint8)


def _unpack_signs(b, C):
    bits = (b.to(torch.int64)[..., None] >> torch.arange(8)) & 1
    return bits.reshape(*b.shape[:-1], -1)[..., :C]


class Interp:
    def __init__(self, plan, flat_params: torch.Tensor, fp16: bool = False):
        self.plan, self.fp16 = plan, fp16
        self.params = flat_params.clone().to(torch.get_default_dtype())
        self.grads = torch.zeros_like(self.params)
        self.t = {}
        self.scale = 1.0
        self.slab = self.bslab = None
        self.L = {l.name: l for l in plan.layers}

    # ---- tensor access ---------------------------------------------------------------------------------
    def view(self, v, c):
        """channels [co, co+c) of an act tensor"""
        return self.t[v.t][..., v.co:v.co + c]

    def alloc(self, name):
        spec = self.plan.tensors[name]
        if name not in self.t:
            self.t[name] = torch.zeros(spec.shape)
        return self.t[name]

    def store(self, v, c, val):
        self.alloc(v.t)[..., v.co:v.co + c] = _r16(val, self.fp16, self.plan.tensors[v.t].kind)

    def weight(self, lname):
        l = self.L[lname]
        w = self.params[self.plan.param_base + l.w_off: self.plan.param_base + l.w_off + l.M * l.cin * l.ntaps]
        b = self.params[self.plan.param_base + l.b_off: self.plan.param_base + l.b_off + l.M]
        return w.reshape(l.M, l.cin, l.ntaps), b

    # ---- ops ---------------------------------------------------------------------------------------------
    def run(self, ops):
        for op in ops:
            getattr(self, "op_" + op.type)(**op.a)

    def op_pack_input(self, src, dst, B, C, H, W, R, cpad):
        x = self.t[src].reshape(B, C, H, W)
        rots = [x]
        if R == 4:
            rots = [x, x.flip(3).transpose(2, 3), x.flip(3).flip(2), x.flip(2).transpose(2, 3)]
        st = torch.cat(rots, 0).permute(0, 2, 3, 1)
        out = torch.zeros(R * B, H, W, cpad)
        out[..., :C] = st
        self.store(dst, cpad, out)

    def op_wpack(self, layer, M, cin, ntaps, c0, c1_real, Mpad_f, Ktot, Mpad_d, Kd, need_d, cm_f=False, cm_d=False):
        # (cm_f / cm_d: the device also writes chunk-major pre-swizzled copies for k_cdma -- a layout detail of the HIP library,
        #  checked through the convolutions that read them)
        w, _ = self.weight(layer)                       # [M, cin, ntaps]
        kmap = [k if k < c0 else (c0 + k - c0 if k - c0 < c1_real else -1) for k in range(Ktot)]
        wf = torch.zeros(ntaps, Mpad_f, Ktot)
        for k, ci in enumerate(kmap):
            if ci >= 0:
                wf[:, :M, k] = w[:, ci, :].t()
        self.t[self.plan.prefix + "wf/" + layer] = _r16(wf, self.fp16)
        if need_d:
            wd = torch.zeros(ntaps, Mpad_d, Kd)
            for c in range(min(Mpad_d, Ktot)):
                ci = kmap[c]
                if ci >= 0:
                    wd[:, c, :M] = w[:, ci, :].t()
            self.t[self.plan.prefix + "wd/" + layer] = _r16(wd, self.fp16, "bf16")

    def _gather(self, src0, src1, c0, c1, up0, N, H, W):
        parts = []
        if c0:
            s = self.view(src0, c0)
            if up0:
                s = s.repeat_interleave(2, 1).repeat_interleave(2, 2)
            parts.append(s)
        if c1:
            parts.append(self.view(src1, c1))
        x = torch.cat(parts, -1)
        assert x.shape[:3] == (N, H, W), (x.shape, N, H, W)
        return x

    @staticmethod
    def _shift(x, dy, dx):
        """y[n,i,j] = x[n,i+dy,j+dx], zero outside"""
        N, H, W, K = x.shape
        out = torch.zeros_like(x)
        ys0, ys1 = max(0, -dy), min(H, H - dy)
        xs0, xs1 = max(0, -dx), min(W, W - dx)
        if ys1 > ys0 and xs1 > xs0:
            out[:, ys0:ys1, xs0:xs1] = x[:, ys0 + dy:ys1 + dy, xs0 + dx:xs1 + dx]
        return out

    def op_conv(self, layer, role, src0, src1, c0, c1, up0, N, H, W, taps, M, Mpad, Ktot, bias, act, mask, add, dst, dst32,
                ltw, lth, ltn, kc, bf16=0, kreal=0, pool=None, pool_shifted=0, upsum=None, upsum_mask=None, upsum_c=0,
                unrot=None, unrot_mask=None, unrot_smask=None, urot=None, urot_smask=None, sign_out=None, mask_sign=None, upsum_mask_sign=None):
        x = self._gather(src0, src1, c0, c1, up0, N, H, W)
        wp = self.t[self.plan.prefix + ("wf/" if role == "fwd" else "wd/") + layer]
        assert wp.shape == (len(taps), Mpad, Ktot), (wp.shape, len(taps), Mpad, Ktot)
        out = torch.zeros(N, H, W, M)
        for t, (dy, dx) in enumerate(taps):
            out += torch.einsum("nhwk,mk->nhwm", self._shift(x, dy, dx), wp[t, :M])
        if bias:
            out += self.weight(layer)[1]
        if act:
            out = torch.where(out > 0, out, LRELU * out)
        if dst32 is not None:
            self.t[dst32] = out.permute(0, 3, 1, 2).contiguous()
            return
        if urot is not None:       # fused SSDN_OP_UNROT_FWD of the (rounded) output; nothing goes to dst
            self._unrot_fwd(_r16(out, self.fp16, self.plan.tensors[urot.t].kind), urot, N // 4, H, M, urot_smask)
            return
        if add is not None:
            out = out + self.view(add, M)
        if mask_sign is not None:      # sign bytes of the tensor `mask` views (written by its producer: sign_out)
            out = out * torch.where(_unpack_signs(self.t[mask_sign], M) > 0, 1.0, LRELU)
        elif mask is not None:
            out = out * _lgrad(self.view(mask, M))
        if unrot is not None:      # fused SSDN_OP_UNROT_BWD of the (rounded) output; nothing goes to dst
            C4 = M // 4
            g = _r16(out, self.fp16, self.plan.tensors[unrot.t].kind)
            outs = []
            for r, ang in enumerate((0, 90, 180, 270)):
                gs = self._rot(g[..., r * C4:(r + 1) * C4], ang)
                outs.append(torch.cat([gs[:, 1:], torch.zeros(N, 1, W, C4)], 1))
            if unrot_smask is not None:      # the sign bytes SSDN_OP_UNROT_FWD left (rows y <= P-2; row P-1 only meets zeros)
                lg = torch.where(_unpack_signs(self.t[unrot_smask], C4) > 0, 1.0, LRELU)
            else:
                lg = _lgrad(self.view(unrot_mask, C4))
            self.store(unrot, C4, torch.cat(outs, 0) * lg)
            return
        if upsum is not None:      # fused SSDN_OP_UPSUM_BWD of the (rounded) channels below upsum_c; the rest goes to dst
            r = _r16(out[..., :upsum_c], self.fp16, self.plan.tensors[dst.t].kind)
            sm = r.reshape(N, H // 2, 2, W // 2, 2, upsum_c).sum((2, 4))
            if upsum_mask_sign is not None:      # sign bytes of the tensor `upsum_mask` views (written by its producer: sign_out)
                lg = torch.where(_unpack_signs(self.t[upsum_mask_sign], upsum_c) > 0, 1.0, LRELU)
            else:
                lg = _lgrad(self.view(upsum_mask, upsum_c))
            self.store(upsum, upsum_c, sm * lg)
            if M > upsum_c:
                self.alloc(dst.t)[..., dst.co + upsum_c:dst.co + M] = _r16(out[..., upsum_c:], self.fp16, self.plan.tensors[dst.t].kind)
            return
        self.store(dst, M, out)
        if sign_out is not None:
            self.t[sign_out] = _pack_signs(self.view(dst, M))
        if pool is not None:       # fused SSDN_OP_POOL_FWD of the stored (rounded) output
            self.store(pool, M, self._windows(self.view(dst, M), pool_shifted).max(3).values)

    def _windows(self, a, shifted):
        """[N,Ho,Wo,4,C] window values in scan order (row-major); the shifted variant sees a zero row on top."""
        N, H, W, C = a.shape
        if shifted:
            a = torch.cat([torch.zeros(N, 1, W, C), a[:, :-1]], 1)
        return a.reshape(N, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(N, H // 2, W // 2, 4, C)

    def op_pool_fwd(self, act, pooled, N, H, W, C, shifted, route=None):
        w = self._windows(self.view(act, C), shifted)
        self.store(pooled, C, w.max(3).values)
        if route is not None:      # include/ssdn_hip.h, ssdn_pool_args.route: nibble = first maximum's scan position | pad row wins << 2 | max > 0 << 3
            mx = w.max(3).values
            hit = (w == mx.unsqueeze(3))
            pos = (hit & (hit.cumsum(3) == 1)).to(torch.int64).argmax(3)            # [N,Ho,Wo,C]
            nib = pos
            if shifted:            # the literal zero row is scanned first: on the top row of windows it holds every maximum <= 0
                top = torch.zeros_like(mx, dtype=torch.bool)
                top[:, 0] = True
                nib = torch.where(top & (mx <= 0), torch.full_like(pos, 4), pos)
            nib = nib | ((mx > 0).to(torch.int64) << 3)
            sh = (4 * torch.arange(8, dtype=torch.int64)).repeat(C // 8)
            words = (nib << sh).reshape(N, H // 2, W // 2, C // 8, 8).sum(-1)
            self.t[route] = words.to(torch.int64)

    def op_pool_bwd(self, act, dpool, dz, N, H, W, C, shifted, route=None):
        a = self.view(act, C)
        w = self._windows(a, shifted)
        mx = w.max(3, keepdim=True).values
        hit = (w == mx)
        first = hit & (hit.cumsum(3) == 1)                       # first position in scan order
        g = first * self.view(dpool, C).unsqueeze(3)           # [N,Ho,Wo,4,C]
        g = g.reshape(N, H // 2, W // 2, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(N, H, W, C)
        if shifted:
            g = torch.cat([g[:, 1:], torch.zeros(N, 1, W, C)], 1)   # undo the shift; what landed on the pad row is dropped
        self.store(dz, C, g * _lgrad(a))

    def op_upsum_bwd(self, src, mask, dst, N, H, W, C):
        s = self.view(src, C).reshape(N, H, 2, W, 2, C).sum((2, 4))
        self.store(dst, C, s * _lgrad(self.view(mask, C)))

    @staticmethod
    def _rot(x, a):  # NHWC version of utils/data.py rotate
        if a == 0:
            return x
        if a == 90:
            return x.flip(2).transpose(1, 2)
        if a == 180:
            return x.flip(2).flip(1)
        return x.flip(1).transpose(1, 2)

    def op_unrot_fwd(self, src, dst, B, P, C, smask=None):
        self._unrot_fwd(self.view(src, C), dst, B, P, C, smask)

    def _unrot_fwd(self, y, dst, B, P, C, smask=None):
        if smask is not None:     # bit q of byte k = (channel 8k+q > 0)
            self.t[smask] = _pack_signs(y)
        s = torch.cat([torch.zeros(4 * B, 1, P, C), y[:, :-1]], 1)
        parts = [self._rot(s[r * B:(r + 1) * B], a) for r, a in enumerate((0, 270, 180, 90))]
        self.store(dst, 4 * C, torch.cat(parts, -1))

    def op_unrot_bwd(self, src, dst, mask, B, P, C):
        g = self.view(src, 4 * C)
        outs = []
        for r, a in enumerate((0, 90, 180, 270)):      # inverse rotations
            gs = self._rot(g[..., r * C:(r + 1) * C], a)
            outs.append(torch.cat([gs[:, 1:], torch.zeros(B, 1, P, C)], 1))   # adjoint of the one-row down shift
        self.store(dst, C, torch.cat(outs, 0) * _lgrad(self.view(mask, C)))

    def op_grad_pack(self, g, dst, N, C, H, W, cpad):
        gg = self.t[g].reshape(N, C, H, W)
        self.scale = 1.0                       # bf16 gradients: no loss scaling
        out = torch.zeros(N, H, W, cpad)
        out[..., :C] = gg.permute(0, 2, 3, 1)
        self.store(dst, cpad, out)

    def op_wgrad(self, layer, dz, src0, src1, c0, c1, up0, N, H, W, taps, coff, M, Mpad, Ktot, Kpad, nslabs, ltw, lth, ltn,
                 slab=None, bslab=None, csplit=0, mblocks=1, kreal=0, mega=0, cost=0.0, **_planner_private):
        x = _r16(self._gather(src0, src1, c0, c1, up0, N, H, W), self.fp16, "bf16")   # staged as bf16 on the device
        self.slabs, self.bslabs = [], []
        for mb in range(mblocks):        # a merged launch covers mblocks blocks of M output channels of the dz view
            g = self.t[dz.t][..., dz.co + mb * M:dz.co + mb * M + M]
            sl = torch.zeros(len(taps), Mpad, Kpad)
            for t, (dy, dx) in enumerate(taps):
                kw = min(Kpad, Ktot - coff[t])
                sl[t, :M, :kw] = torch.einsum("nhwm,nhwk->mk", g, self._shift(x, dy, dx)[..., coff[t]:coff[t] + kw])
            bs = torch.zeros(Mpad)
            bs[:M] = g.sum((0, 1, 2))
            self.slabs.append(sl)
            self.bslabs.append(bs)
        self.slab, self.bslab = self.slabs[0], self.bslabs[0]

    def op_wreduce(self, layer, nslabs, ntaps, M, Mpad, Kpad, cin, cin_full, m_off, c_off, with_bias, tapblock=0,
                   slab=None, bslab=None, mblock=0):
        self.slab, self.bslab = self.slabs[mblock], self.bslabs[mblock]
        l = self.L[layer]
        base = self.plan.param_base
        gw = self.grads[base + l.w_off: base + l.w_off + l.M * l.cin * l.ntaps].reshape(l.M, l.cin, l.ntaps)
        assert cin_full == l.cin
        if tapblock:     # the slab's taps are channel blocks of a 1x1 layer
            blk = self.slab[:, :M, :].permute(1, 0, 2).reshape(M, ntaps * Kpad)[:, :cin]
            gw[m_off:m_off + M, c_off:c_off + cin, 0] = blk / self.scale
        else:
            gw[m_off:m_off + M, c_off:c_off + cin, :] = self.slab[:, :M, :cin].permute(1, 2, 0) / self.scale
        if with_bias:
            self.grads[base + l.b_off + m_off: base + l.b_off + m_off + M] = self.bslab[:M] / self.scale

"""Lowering of the blind-spot U-Net (+ loss heads, optimiser) to the flat op list executed by libssdn_hip.so.

The reference builds an autograd graph of ~200 ATen ops per step (NoiseNetwork.forward,
/root/reference/ssdn/ssdn/models/noise_network.py:186-226; Denoiser.run_pipeline, denoiser.py:128-397).  Here the
same computation is planned ONCE per (batch, patch size) into a static list of `Op` records -- backend neutral:
`ssdn.hip.engine` turns them into `ssdn_op` C structs holding device pointers, and the test-only interpreter
(oracle/interp.py) executes the very same records on the CPU to check the lowering against the oracle.

Nothing here touches a device; this module is pure Python.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

LDS_LIMIT = 160 * 1024
# conv tilings at or below this many bytes of LDS are preferred: two workgroups then share a CU, so one workgroup's
# tile staging / barriers overlap the other's MFMA work (tunable for experiments)
CONV_LDS_PREFERRED = 80 * 1024
# Planner constants of the weight-gradient family (measured on BASELINE config 2; module constants, NOT environment switches: the
# plan a process makes never depends on its environment)
WGRAD_SMALL_PX = 32768        # layers with at most this many pixels (16x16 and below at BASELINE sizes) share merged launches ...
SMALL_WGRAD_G, SMALL_WGRAD_CUS = 2, 64      # ... planned as two column groups on <= 64 / 2 pixel partitions
MID_WGRAD_PX, MID_WGRAD_G = 131072, 2       # the 32x32 stage: two column groups on half the partitions
WGRAD_CU_FRAC = (3, 4)        # share of the CUs the persistent weight-gradient grids are planned for (see _wgrad)
# Round 4: the weight gradients of a whole group of layers as ONE chip-wide launch (k_wgrad_mega, csrc/wgrad_mfma.hip): one
# workgroup per CU, every op's grid sized by the cost model below so that all workgroups finish together.
#   "all": one launch behind the last data gradient; "buckets": one per gradient bucket (ssdn.hip.dp.bucket_layers);
#   None: round 3's per-layer launches on the side lane
WGRAD_MEGA = "split"
FUSE_UNROT_FWD = True         # decode_block_1.2 stores its output un-rotated (no SSDN_OP_UNROT_FWD launch, no d1b tensor) where k_cdma serves it
SIGN_BYTES_HEAD = True        # output_block.0 leaves sign bytes of its 384-channel output for the data gradient of output_block.2
X16_SLOTS = 16                # channel slots of the packed network input (C <= 3 real channels): what its readers address (the first layer and
                              # decode_block_1.0 read a 16-slot chunk, the thin weight gradients the first 8 slots); 32 until round 4: 64-byte pixels
POOL_ROUTE = True             # SSDN_OP_POOL_FWD leaves route words (winner position + LeakyReLU sign); a stand-alone SSDN_OP_POOL_BWD reads them instead of the activation
SIGN_BYTES_CONV = True        # the 3x3 layers k_cdma / k_conv_thin serve leave sign bytes of their outputs; k_cdma's data gradients read them as LeakyReLU' masks
SIGN_BYTES = True             # the fused un-rotation of the backward pass reads LeakyReLU sign bytes (12 B/pixel) instead of d1b (192 B/pixel)
MEGA_MIN_PX = 32768           # networks with fewer pixels (N*H*W) at full resolution keep the per-layer launches: a handful of tiles per
                              # layer is latency, not throughput (config 1's shape, batch 4 at 32x32: 0.66 ms per step vs 0.70 / 0.77)
SPLIT_HEAD_CUS = (1, 2)       # "split": share of the CUs the side-lane launch is planned for
SPLIT_AFTER = "decode_block_2.2"   # "split": the side-lane launch is issued behind this layer's data gradient; None = as soon as the group's last operand exists
                              # (measured, same process: behind decode_block_2.0's data gradient 1.6028 ms per step, behind decode_block_2.2's 1.5892, None 1.5864;
                              #  behind decode_block_2.2 only ONE of the eight k_cdma<3,*> launches of a step runs beside it, with None two)
SPLIT_GROUP0 = ("output_block", "decode_block_2.2")   # "split": layers (name prefixes) of the side-lane launch
# cost model of one weight-gradient block, in cycles (calibrated on BASELINE config 2 with tools/wgrad_calib.py):
#   K-step of 16 pixels = base + per_mfma * MT * CPW;  a block = tiles * ksteps * K-step + fixed + slab bytes / slab_rate
MEGA_COST = {
    # measured INSIDE the chip-wide launch of BASELINE config 2 (every CU streaming; tools/wgrad_calib.py trace: per-block
    # s_memrealtime stamps), in cycles of a nominal 2.1 GHz clock.  On a quiet chip the same blocks run ~1.45x faster.
    "static": (342.0, 38.8),      # compile-time staging schedules (3x3 layers, 16x8-pixel tiles): 4.4 us per (3,7) tile
    "static2": (258.0, 40.2),     # ... of the layers with <= 48 input channels (two loads per input row: csrc/wgrad_body.h::wgrad_nl2)
    "generic": (1350.0, 53.0),    # run-time staging (small layers, odd tiles)
    "head": (1208.0, 0.0),        # 1x1 layers over four 96-channel input blocks (an input AND a dZ row per K-step)
    "head_mb": (1208.0, 0.0),     # ... with the 4 output blocks of a pixel partition side by side (input shared through L2)
    "thin": (2211.0, 963.0),      # k_wgrad_thin: per 256-pixel tile: base + per 32 output channels
    "sync_tile": 9000.0,          # a tile fetched synchronously (unprefetchable tiles of the 4x4 / 2x2 layers), on top of its K-steps
    "fixed": 30000.0,             # prologue (first tile fetched synchronously), launch ramp
    "slab_rate": 10.0,            # bytes per cycle a workgroup writes its slab with
}
TAPS_BLIND = [(ky - 2, kx - 1) for ky in range(3) for kx in range(3)]   # ShiftConv2d: in[y+ky-2, x+kx-1]
TAPS_PLAIN = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]
TAPS_1x1 = [(0, 0)]


def ceil_to(v: int, m: int) -> int:
    return (v + m - 1) // m * m


@dataclass
class TSpec:
    """A tensor of the plan.  kind 'act': NHWC fp16 [N,H,W,C] (forward activation); 'actb': NHWC bf16 (gradient);
    'f32': flat fp32; 'f16' / 'bf16': flat 16-bit; 'u32'; 'i64'."""
    name: str
    kind: str
    shape: Tuple[int, ...]


@dataclass
class View:
    """Channel-slice view of an 'act' tensor (never a copy)."""
    t: str
    co: int = 0


@dataclass
class Op:
    type: str
    a: dict = field(default_factory=dict)


@dataclass
class Layer:
    name: str          # reference state_dict prefix, e.g. 'decode_block_1.0'
    M: int             # output channels
    cin: int           # real input channels
    k: int             # kernel size (3 or 1)
    c0: int            # channels of source 0 in the packed K axis
    c1: int            # channels of source 1 in the packed K axis (padded to a multiple of 16 - c0 % 16 ...)
    c1_real: int
    w_off: int = 0     # offsets (floats) into the flat parameter buffer
    b_off: int = 0

    @property
    def ntaps(self):
        return self.k * self.k

    @property
    def Ktot(self):
        return self.c0 + self.c1

    @property
    def Mpad_f(self):
        return ceil_to(self.M, 32)

    @property
    def Kd(self):
        return ceil_to(self.M, 16)      # reduction length of the data-gradient GEMM


def net_layers(in_channels: int, out_channels: int, blindspot: bool) -> List[Layer]:
    """Same order as the reference modules (noise_network.py:69-156) == oracle/restate.layer_table."""
    c = in_channels
    cpad = 16
    nin = 384 if blindspot else 96
    L = [
        Layer("encode_block_1.0", 48, c, 3, cpad, 0, 0),
        Layer("encode_block_1.2", 48, 48, 3, 48, 0, 0),
        Layer("encode_block_2.0", 48, 48, 3, 48, 0, 0),
        Layer("encode_block_3.0", 48, 48, 3, 48, 0, 0),
        Layer("encode_block_4.0", 48, 48, 3, 48, 0, 0),
        Layer("encode_block_5.0", 48, 48, 3, 48, 0, 0),
        Layer("encode_block_6.0", 48, 48, 3, 48, 0, 0),
        Layer("decode_block_5.0", 96, 96, 3, 48, 48, 48),
        Layer("decode_block_5.2", 96, 96, 3, 96, 0, 0),
        Layer("decode_block_4.0", 96, 144, 3, 96, 48, 48),
        Layer("decode_block_4.2", 96, 96, 3, 96, 0, 0),
        Layer("decode_block_3.0", 96, 144, 3, 96, 48, 48),
        Layer("decode_block_3.2", 96, 96, 3, 96, 0, 0),
        Layer("decode_block_2.0", 96, 144, 3, 96, 48, 48),
        Layer("decode_block_2.2", 96, 96, 3, 96, 0, 0),
        Layer("decode_block_1.0", 96, 96 + c, 3, 96, cpad, c),          # K = 96 + 16: two 48-channel chunks + a 16-channel tail
        Layer("decode_block_1.2", 96, 96, 3, 96, 0, 0),
        Layer("output_block.0", nin, nin, 1, nin, 0, 0),
        Layer("output_block.2", 96, nin, 1, nin, 0, 0),
        Layer("output_block.4", out_channels, 96, 1, 96, 0, 0),
    ]
    # first layer: the only source is the 16-channel padded input, of which `c` channels are real
    L[0].c0, L[0].c1, L[0].c1_real = 0, cpad, c
    off = 0
    for l in L:
        l.w_off = off
        off += l.M * l.cin * l.k * l.k
        l.b_off = off
        off += l.M
    return L


def net_param_count(layers: List[Layer]) -> int:
    return layers[-1].b_off + layers[-1].M


# --------------------------------------------------------------------------------------------------------------
# tiling
# --------------------------------------------------------------------------------------------------------------

def _pads(taps):
    dys = [t[0] for t in taps]
    dxs = [t[1] for t in taps]
    return max(0, -min(dys)), max(0, max(dys)), max(0, -min(dxs)), max(0, max(dxs))


def _pow2ceil_log(v: int) -> int:
    l = 0
    while (1 << l) < v:
        l += 1
    return l


def cdma_fills(tiles: int, cus: int, H: int = 0, W: int = 0) -> bool:
    """csrc/conv_dma.hip::conv_dma_eligible's size rule: k_cdma serves a 3x3 layer from one 16x16-pixel tile per CU upwards -- and from one
    tile per TWO CUs where an image is more than one tile (round 6: the plain network's 32x32 stage at batch 32, 128 tiles: config 4's step
    0.89 -> 0.83 ms).  Not the blind-spot network's 16x16 stage (also 128 tiles, one per image): alone its launches are 4-6 us faster on
    k_cdma than on k_conv's flat path (tools/r6_mode2.py), inside the step -- beside the half-chip weight-gradient launch, whose blocks and
    k_cdma's workgroups cannot share a CU's LDS -- the step was 1.5 % slower."""
    return tiles >= cus or (2 * tiles >= cus and H * W > 256)


def choose_conv_tile(N, H, W, taps, Ktot, Mpad, budget=LDS_LIMIT, out16=True, cus=256):
    """Pick (ltw, lth, ltn, kc) for SSDN_OP_CONV: <= 256 pixels per workgroup, LDS = halo tile + two weight slices.
    Layers with at most one pixel tile per CU run as 32-output-channel blocks (the library's rule, conv_uses_mt1): one
    workgroup per CU at most, so LDS per workgroup does not matter and the whole 96-channel chunk is staged at once."""
    padT, padB, padL, padR = _pads(taps)
    best = None
    for ltw in range(0, min(5, _pow2ceil_log(W)) + 1):
        for lth in range(0, min(8 - ltw, _pow2ceil_log(H)) + 1):
            for ltn in range(0, min(8 - ltw - lth, _pow2ceil_log(N)) + 1):
                TW, TH, TN = 1 << ltw, 1 << lth, 1 << ltn
                tiles = -(-W // TW) * -(-H // TH) * -(-N // TN)
                util = (N * H * W) / (tiles * 256.0)       # MFMA lanes doing useful work
                NP = TN * (TH + padT + padB) * (TW + padL + padR)
                mt1 = out16 and tiles <= cus and Mpad >= 64
                mt = 1 if mt1 else min(3, Mpad // 32)
                # channel chunks the kernel is instantiated for (KS = kc/16 is a template parameter of k_conv)
                kcs = [kc for kc in ((16, 32, 48, 64, 96) if mt1 else (16, 32, 48, 64)) if Ktot % kc == 0]
                for kc in kcs:
                    lds = NP * (kc * 2 + 16) + 2 * mt * 32 * (kc * 2 + 16)
                    if len(taps) == 1 and Ktot // kc >= 2:   # 1x1 layers may run the two-buffer asynchronous tile pipeline
                        lds = max(lds, 2 * NP * kc * 2 + 2 * mt * 32 * (kc * 2 + 16))
                    if out16:      # the output tile is transposed through the same LDS in the epilogue
                        lds = max(lds, TN * TH * TW * (mt * 64 + 16) + mt * 128)
                    if lds > budget:
                        continue
                    # 32-channel-block layers: prefer a tiling whose whole 9-tap weight block fits next to the tile (k_conv ALLW:
                    # one barrier per channel chunk instead of one per tap)
                    allw = mt1 and len(taps) == 9 and NP * (kc * 2 + 16) + 9 * 32 * (kc * 2 + 16) <= LDS_LIMIT
                    # ... and, for whole-image tiles, the flat variant (register prefetch of the next 48/64-channel chunk; the
                    # 96-channel instance has no registers left for it)
                    flat = allw and TW == W and TH == H and TN * TH * TW == 256 and N % TN == 0 and kc <= 64
                    allw = (allw, flat)
                    halo = NP / float(TN * TH * TW)
                    # prefer: high utilisation, then 2 workgroups per CU, then large channel chunks (fewer barriers),
                    # then small halo, then wide tiles
                    key = (round(util, 3), mt1 or (lds <= CONV_LDS_PREFERRED and kc >= min(48, Ktot)), allw, kc, -round(halo, 3), ltw)
                    if best is None or key > best[0]:
                        best = (key, (ltw, lth, ltn, kc))
    if best is None:
        raise ValueError("no conv tiling fits in LDS for N=%d H=%d W=%d Ktot=%d" % (N, H, W, Ktot))
    return best[1]


def conv_fuses_pool(N, H, W, taps, Ktot, M, Mpad, ltw, lth, ltn, kc, cus):
    """The library's rule (csrc/conv_mfma.hip::conv_fuses_pool; the engine cross-checks with ssdn_conv_fuses_pool): the launch
    runs k_conv's flat path -- 32-channel blocks (<= one tile per CU), all nine weight slices next to the tile in LDS, the
    256-pixel tile made of whole images."""
    TW, TH, TN = 1 << ltw, 1 << lth, 1 << ltn
    padT, padB, padL, padR = _pads(taps)
    tiles = -(-W // TW) * -(-H // TH) * -(-N // TN)
    NP = TN * (TH + padT + padB) * (TW + padL + padR)
    m_last = M - (Mpad - 32)
    return (len(taps) == 9 and tiles <= cus and Mpad >= 64 and NP * (kc * 2 + 16) + 9 * 32 * (kc * 2 + 16) <= LDS_LIMIT and
            TW == W and TH == H and TN * TH * TW == 256 and N % TN == 0 and kc <= 64 and m_last in (32, 16, 8) and
            H % 2 == 0 and W % 2 == 0)


def _wg_stride(row_bytes):
    """pixel stride of the weight-gradient kernel's LDS images: == 64 (mod 128) bytes (bank-conflict-free transpose reads)"""
    return ((row_bytes + 63) & ~127) + 64


def choose_wgrad_tile(N, H, W, taps, Kpad, Mpad, Ktot, M, cus, budget=LDS_LIMIT):
    padT, padB, padL, padR = _pads(taps)
    best = None
    for ltw in range(0, min(4, _pow2ceil_log(W)) + 1):
        for lth in range(0, min(8 - ltw, _pow2ceil_log(H)) + 1):
            for ltn in range(max(0, 5 - ltw - lth), 8 - ltw - lth + 1):   # the kernel needs >= 32 pixels per tile
                if ltn > max(_pow2ceil_log(N), 5 - ltw - lth):
                    continue                                             # surplus images would only be masked lanes
                TW, TH, TN = 1 << ltw, 1 << lth, 1 << ltn
                tiles = -(-W // TW) * -(-H // TH) * -(-N // TN)
                util = (N * H * W) / float(tiles * TN * TH * TW)
                NP = TN * (TH + padT + padB) * (TW + padL + padR)
                # two LDS images (the kernel prefetches tile i+1 while tile i is on the matrix cores) ...
                HW_ = TW + padL + padR
                # (each part of an image has one extra dummy row that absorbs void row items)
                lds = 2 * ((NP + HW_) * _wg_stride(Kpad * 2) + (TN * TH * TW + TW) * _wg_stride(Mpad * 2)) + 4096
                if lds > budget:
                    continue
                # ... filled row by row: the rows of a tile are dealt to the kernel's 4 waves, a wave loads one row (<= 4
                # 64-lane 16-byte loads for an input row, <= 3 for a dZ row) per 16-pixel K-step in all but the last few
                # K-steps of a tile -- or, for wide rows (<= 6 loads), an input AND a dZ row per K-step.  Irrelevant when no
                # workgroup gets a second tile.
                HH, HW = TH + padT + padB, TW + padL + padR
                ix, idz = -(-(HW * (Ktot // 8)) // 64), -(-(TW * (M // 8)) // 64)
                rswx, rswd = -(-(TN * HH) // 4), -(-(TN * TH) // 4)
                steps = TN * TH * TW // 16 - 2
                single = tiles <= cus
                if idz > 3:
                    continue
                if not ((ix <= 4 and (single or rswx + rswd <= steps)) or
                        (ix <= 6 and (single or (rswx <= steps and rswd <= steps)))):
                    continue
                # (ties: one image per tile -- the kernel's compile-time schedules need that)
                key = (round(util, 3), TN * TH * TW, -NP, ltw, -ltn)
                if best is None or key > best[0]:
                    best = (key, (ltw, lth, ltn), tiles)
    if best is None:
        raise ValueError("no wgrad tiling fits in LDS")
    return best[1], best[2]


# --------------------------------------------------------------------------------------------------------------
# the plan
# --------------------------------------------------------------------------------------------------------------

class NetPlan:
    """Forward / backward op lists of ONE NoiseNetwork instance for a fixed input shape.

    prefix   unique tensor-name prefix (a Denoiser owns up to two nets)
    B,H,W    input batch / spatial size (H == W when blindspot)
    cus      compute units of the device (sizes the persistent wgrad grids)
    dev_cus  compute units the LIBRARY sees (ssdn_device_cus: its kernel-selection rules use it); only differs from `cus`
             when a test plans small persistent grids on a big device
    """

    def __init__(self, prefix: str, in_channels: int, out_channels: int, blindspot: bool, B: int, H: int, W: int,
                 cus: int = 256, train: bool = True, param_base: int = 0, dev_cus: Optional[int] = None):
        if H % 32 or W % 32:
            raise ValueError("input height/width must be multiples of 32 (NoiseNetwork.input_wh_mul)")
        if blindspot and H != W:
            raise ValueError("blind-spot mode needs square inputs")
        self.prefix, self.C, self.Cout, self.blindspot = prefix, in_channels, out_channels, blindspot
        self.B, self.H, self.W, self.cus, self.train = B, H, W, cus, train
        self.dev_cus = cus if dev_cus is None else dev_cus
        self.R = 4 if blindspot else 1
        self.N = self.R * B
        self.layers = net_layers(in_channels, out_channels, blindspot)
        self.nparams = net_param_count(self.layers)
        self.param_base = param_base          # offset of this net inside the owner's flat parameter buffer
        self.taps3 = TAPS_BLIND if blindspot else TAPS_PLAIN
        self.tensors: Dict[str, TSpec] = {}
        self.fwd: List[Op] = []
        self.bwd: List[Op] = []
        self.pack: List[Op] = []
        self.max_slab = 0
        self.max_bslab = 0
        self._build()

    # ---- helpers -------------------------------------------------------------------------------------------
    def T(self, name, kind, shape):
        full = self.prefix + name
        self.tensors[full] = TSpec(full, kind, tuple(int(s) for s in shape))
        return full

    def act(self, name, N, H, W, C):
        return self.T(name, "act", (N, H, W, C))

    def grad(self, name, N, H, W, C):
        """gradient tensor: bf16 (include/ssdn_hip.h, conventions)"""
        return self.T(name, "actb", (N, H, W, C))

    def _conv(self, lst, layer: Layer, role: str, src0, c0, up0, src1, c1, N, H, W, taps, M, dst=None, dst32=None,
              bias=True, act=True, mask=None, add=None, pool=None, pool_shifted=0, upsum=None, upsum_mask=None, upsum_c=0,
              unrot=None, unrot_mask=None, unrot_smask=None, urot=None, urot_smask=None, sign_out=None, mask_sign=None, upsum_mask_sign=None):
        """pool: view of the pooled tensor -- the conv's epilogue also writes Shift2d + MaxPool2d(2) of its output
        (ssdn_conv_args.pool).  Returns True if the pool was fused (the caller then emits no SSDN_OP_POOL_FWD)."""
        Ktot = c0 + c1
        Mpad = ceil_to(M, 32)
        if upsum is None:
            upsum_mask_sign = None
        ltw, lth, ltn, kc = choose_conv_tile(N, H, W, taps, Ktot, Mpad, out16=dst32 is None, cus=self.cus)
        fused = pool is not None and conv_fuses_pool(N, H, W, taps, Ktot, M, Mpad, ltw, lth, ltn, kc, self.dev_cus)
        if upsum is not None:
            # fused SSDN_OP_UPSUM_BWD: k_cdma (>= one 16x16 tile per CU, whole 96-channel blocks) or k_conv's flat path
            dma = (len(taps) == 9 and H % 16 == 0 and W % 16 == 0 and cdma_fills(N * (H // 16) * (W // 16), self.dev_cus, H, W) and
                   Ktot % 48 in (0, 16) and upsum_c % 96 == 0)
            fused = (dma or conv_fuses_pool(N, H, W, taps, Ktot, M, Mpad, ltw, lth, ltn, kc, self.dev_cus)) and upsum_c % 8 == 0 and \
                mask is None and add is None
            if not fused:
                upsum = upsum_mask = None
            if not (fused and dma):
                upsum_mask_sign = None      # (only k_cdma reads the up-sampled half's mask as sign bytes)
        lst.append(Op("conv", dict(layer=layer.name, role=role, src0=src0, src1=src1, c0=c0, c1=c1, up0=int(up0), N=N, H=H, W=W,
                                   taps=list(taps), M=M, Mpad=Mpad, Ktot=Ktot, bias=bias, act=int(act), mask=mask, add=add,
                                   dst=dst, dst32=dst32, ltw=ltw, lth=lth, ltn=ltn, kc=kc, bf16=int(role == "dgrad"),
                                   kreal=(layer.cin if role == "fwd" else layer.M),
                                   pool=pool if (fused and pool is not None) else None,
                                   pool_shifted=int(pool_shifted) if (fused and pool is not None) else 0,
                                   upsum=upsum, upsum_mask=upsum_mask, upsum_c=int(upsum_c) if upsum is not None else 0,
                                   unrot=unrot, unrot_mask=unrot_mask, unrot_smask=unrot_smask,
                                   urot=urot, urot_smask=urot_smask, sign_out=sign_out, mask_sign=mask_sign,
                                   upsum_mask_sign=upsum_mask_sign)))
        return fused

    def _wgrad(self, layer: Layer, dz: View, Mz: int, src0, c0, up0, src1, c1, cin_real, N, H, W, taps,
               m_off=0, c_off=0, with_bias=True, cblocks=None, mblocks=1):
        """One SSDN_OP_WGRAD + SSDN_OP_WREDUCE pair covering output channels [m_off, m_off+M) x input channels
        [c_off, c_off+cin_real) of `layer` (M <= 96; Ktot <= 96 per tap).
        cblocks (1x1 layers only): list of channel offsets -- the kernel's "taps" become 96-channel blocks of the input,
        so one launch covers M x len(cblocks)*96 weights.
        mblocks > 1: ONE launch covers mblocks consecutive blocks of Mz output channels (dz channels m_off + b*Mz ...), each
        with its own group of slabs and its own reduction; the workgroups of the blocks share the input tiles through L2."""
        Ktot = c0 + c1
        Kpad = ceil_to(Ktot, 32) if cblocks is None else 96
        coff = [0] * len(taps) if cblocks is None else list(cblocks)
        if cblocks is not None:
            taps = [(0, 0)] * len(cblocks)
        Mpad = ceil_to(Mz, 32)
        ntaps = len(taps)
        # Split of the work over workgroups.  Every workgroup ends by writing its accumulators as one fp32 slab at ~10 B per
        # clock per CU (~33 K cycles for the full [ntaps][Mpad][Kpad] output) and the reduction reads all slabs back -- more
        # than the matrix work of a layer with few pixels.  Such layers split the OUTPUT into G column groups (csplit):
        # cus // G pixel partitions, each workgroup staging its partition's tiles for 1/G of the columns.  Cycle model
        # measured on the device: a 16-pixel K-step costs ~660 + 90 * (column tiles per wave) cycles (staging-bound floor ..
        # 21-MFMA K-step); slab write slab/10 B per cycle per workgroup; reduction ~1900 B per cycle for all slabs.
        # (The tile is chosen for the number of workgroups that share the pixels: a workgroup with several tiles needs a
        # tile it can prefetch.)
        slab_bytes = ntaps * Mpad * Kpad * 4
        ctiles = ntaps * Kpad // 32 + 1
        if WGRAD_MEGA and self.cus >= 64 and self.N * self.H * self.W >= MEGA_MIN_PX:
            # planned later, together with the other ops of its launch (_plan_mega): grid, tile, slabs, cost
            self.nwgrad = getattr(self, "nwgrad", 0) + 1
            op = Op("wgrad", dict(layer=layer.name, dz=dz, src0=src0, src1=src1, c0=c0, c1=c1, up0=int(up0), N=N, H=H, W=W,
                                  taps=list(taps), coff=coff, M=Mz, Mpad=Mpad, Ktot=Ktot, Kpad=Kpad, nslabs=0, ltw=0, lth=0, ltn=0,
                                  csplit=0, mblocks=mblocks, slab=None, bslab=None, kreal=int(cin_real) if cblocks is None else 0,
                                  mega=0, cost=0.0, _id=self.nwgrad, _cblocks=cblocks is not None))
            self.bwd.append(op)
            reds = []
            for mb in range(mblocks):
                mo = m_off + mb * Mz
                M_real = min(Mz, layer.M - mo)
                r = Op("wreduce", dict(layer=layer.name, nslabs=0, ntaps=ntaps, M=M_real, Mpad=Mpad, Kpad=Kpad, cin=cin_real,
                                       cin_full=layer.cin, m_off=mo, c_off=c_off, with_bias=with_bias, tapblock=int(cblocks is not None),
                                       mblock=mb, slab=None, bslab=None))
                self.bwd.append(r)
                reds.append(r)
            self._mega_ops = getattr(self, "_mega_ops", []) + [(op, reds)]
            return

        def candidate(G):
            cpw = -(-ctiles // (4 * G))
            # A weight-gradient workgroup owns its CU (one wave per SIMD with the whole register file, ~135 KB of LDS): a launch
            # with one workgroup per CU leaves the data-gradient lane nothing to run on until it drains.  Planning the
            # persistent grids for 3/4 of the CUs keeps both lanes running side by side (measured on BASELINE config 2, two
            # boxes: 256 -> 2.35 ms, 208 -> 2.35, 192 -> 2.27 / 2.30, 160 -> 2.29, 128 -> 2.29 ms per step) and trims the slab
            # traffic by a quarter.
            wcus = self.cus if self.cus < 64 else (WGRAD_CU_FRAC[0] * self.cus // WGRAD_CU_FRAC[1]) & ~7
            cus_eff = max(1, wcus // G)
            if small and small_g > 0:
                cus_eff = max(1, min(self.cus, small_cus) // G)
            if mblocks > 1:
                # the mblocks workgroups of a pixel partition run side by side: cus / mblocks partitions fill the chip in ONE
                # round with 1/mblocks of the slab traffic (output_block.0: 38 MB instead of 151 MB written and read back)
                cus_eff = max(1, wcus // mblocks)
            tile, ntiles = choose_wgrad_tile(N, H, W, taps, max(Kpad, Ktot), Mpad, Ktot, Mz, cus_eff)
            ns = max(1, min(ntiles, cus_eff))
            ksteps = (1 << sum(tile)) // 16
            t = -(-ntiles // ns) * ksteps * (660 + 90 * cpw) + slab_bytes / 10.0 / G + ns * slab_bytes / 1900.0
            return t, tile, ns

        best = None
        groups = [1] if mblocks > 1 else sorted({1, 2, 3, 4, -(-ctiles // 4)})
        # Layers of at most 32768 pixels run inside ONE merged launch per gradient bucket (k_wgrad_multi) and get their
        # parallelism from each other: two column groups on 32 pixel partitions each (a few tiles per workgroup) instead of
        # up to seven groups that every one stage the same tiles -- measured on BASELINE config 2: -85 us per step against the
        # per-layer optimum (G = 1 on 32..64 partitions is equal within noise but needs the 21-accumulator instances).
        small = N * H * W <= WGRAD_SMALL_PX and mblocks == 1 and cblocks is None
        small_g, small_cus = SMALL_WGRAD_G, SMALL_WGRAD_CUS
        default_groups = list(groups)
        if small and small_g > 0:
            groups = [small_g]
        elif mblocks == 1 and cblocks is None and N * H * W <= MID_WGRAD_PX and MID_WGRAD_G > 0:
            # the 32x32 stage: two column groups on half the partitions -- the cycle model (which prices a launch alone on
            # the chip) prefers one group; in situ half the slab traffic wins (-45 us per step, measured)
            groups = [MID_WGRAD_G]
        for attempt in (groups, default_groups):
            for G in attempt:
                if G > 1 and 4 * -(-ctiles // (4 * G)) * (G - 1) >= ctiles:
                    continue                      # the last group would be empty
                try:
                    t, tile_g, ns_g = candidate(G)
                except ValueError:                # no tile the fewer, fatter workgroups could prefetch
                    continue
                if best is None or t < 0.9 * best[0]:        # a split must pay clearly
                    best = (t, G, tile_g, ns_g)
            if best is not None:
                break
        _, G, (ltw, lth, ltn), nslabs = best
        csplit = G if G > 1 else 0
        # every weight-gradient launch owns its slab (1.2 GB in total for BASELINE config 2 -- 0.4 % of the 288 GB of HBM): no
        # ordering between a launch and the reduction of an earlier one is ever needed
        self.nwgrad = getattr(self, "nwgrad", 0) + 1
        slab = self.T("slab%d" % self.nwgrad, "f32", (mblocks * nslabs * ntaps * Mpad * Kpad,))
        bslab = self.T("bslab%d" % self.nwgrad, "f32", (mblocks * nslabs * Mpad,))
        self.bwd.append(Op("wgrad", dict(layer=layer.name, dz=dz, src0=src0, src1=src1, c0=c0, c1=c1, up0=int(up0), N=N, H=H, W=W,
                                         taps=list(taps), coff=coff, M=Mz, Mpad=Mpad, Ktot=Ktot, Kpad=Kpad, nslabs=nslabs,
                                         ltw=ltw, lth=lth, ltn=ltn, csplit=csplit, mblocks=mblocks, slab=slab, bslab=bslab,
                                         kreal=int(cin_real) if cblocks is None else 0)))
        for mb in range(mblocks):
            mo = m_off + mb * Mz
            M_real = min(Mz, layer.M - mo)
            self.bwd.append(Op("wreduce", dict(layer=layer.name, nslabs=nslabs, ntaps=ntaps, M=M_real, Mpad=Mpad, Kpad=Kpad,
                                               cin=cin_real, cin_full=layer.cin, m_off=mo, c_off=c_off, with_bias=with_bias,
                                               tapblock=int(cblocks is not None), mblock=mb, slab=slab, bslab=bslab)))

    # ---- construction --------------------------------------------------------------------------------------
    def _build(self):
        B, N, H, W, C = self.B, self.N, self.H, self.W, self.C
        L = {l.name: l for l in self.layers}
        t3 = self.taps3
        bs = self.blindspot
        f = self.fwd

        # ---------------- forward ----------------
        self.T("in32", "f32", (B, C, H, W))
        x16 = self.act("x16", N, H, W, X16_SLOTS)     # C real channels, zero padded (the first layer reads a 16-channel view)
        f.append(Op("pack_input", dict(src=self.prefix + "in32", dst=View(x16), B=B, C=C, H=H, W=W, R=self.R, cpad=X16_SLOTS)))

        fused_pool = {}
        # LeakyReLU sign bytes (one byte per 8 channels) of the activations whose only use in the backward pass, besides being a
        # weight-gradient operand, is the LeakyReLU' mask of a k_cdma data gradient: activation tensor -> sign tensor.  The producers that
        # can write them: k_cdma on one 96-channel block, k_conv_thin (csrc/conv_dma.hip::conv_dma_signs, csrc/conv_mfma.hip::conv_signs;
        # the engine cross-checks every launch with ssdn_conv_signs)
        signs = self.signs = {}

        def cdma_serves(h, w, ktot):
            """csrc/conv_dma.hip::conv_dma_eligible for the 3x3 layers of this network at h x w"""
            return h % 16 == 0 and w % 16 == 0 and cdma_fills(N * (h // 16) * (w // 16), self.dev_cus, h, w) and ktot % 48 in (0, 16)

        def sign_of(t, h, w, C_, ktot_fwd, thin=False):
            ok = self.train and SIGN_BYTES_CONV and cdma_serves(h, w, C_) and \
                ((thin and 1 <= C <= 3 and w % 64 == 0) or (not thin and C_ == 96 and cdma_serves(h, w, ktot_fwd)))
            if ok:
                signs[t] = self.T("smk_" + t[len(self.prefix):], "u8", (N, h, w, C_ // 8))
            return signs.get(t)

        def enc(name, lname, src, cin_slots, h, w, pool_name=None, sign=False):
            t = self.act(name, N, h, w, 48)
            pv = View(self.act(pool_name, N, h // 2, w // 2, 48)) if pool_name else None
            fused_pool[name] = self._conv(f, L[lname], "fwd", View(src), cin_slots, 0, None, 0, N, h, w, t3, 48, dst=View(t),
                                          pool=pv, pool_shifted=int(bs), sign_out=sign_of(t, h, w, 48, cin_slots, thin=True) if sign else None)
            return t

        routes = self.routes = {}      # full-resolution activation -> route words of its max-pool (ssdn_pool_args.route)

        def pool(name, src, h, w):
            t = self.act(name, N, h // 2, w // 2, 48)
            if not fused_pool.get(src[len(self.prefix):]):        # (else: written by the producing conv's epilogue)
                # route words where the backward op runs on its own (pooled images of > 256 pixels: smaller ones ride in k_conv_chain,
                # which reads the activation planes it holds in LDS anyway)
                if self.train and POOL_ROUTE and (h // 2) * (w // 2) > 256:
                    routes[src] = self.T("route_" + name, "u32", (N, h // 2, w // 2, 48 // 8))
                f.append(Op("pool_fwd", dict(act=View(src), pooled=View(t), N=N, H=h, W=w, C=48, shifted=int(bs), route=routes.get(src))))
            return t

        e0 = enc("e0", "encode_block_1.0", x16, 16, H, W, sign=True)      # (mask of encode_block_1.2's data gradient)
        e1 = enc("e1", "encode_block_1.2", e0, 48, H, W, "p1")
        p1 = pool("p1", e1, H, W)
        e2 = enc("e2", "encode_block_2.0", p1, 48, H // 2, W // 2, "p2")
        p2 = pool("p2", e2, H // 2, W // 2)
        e3 = enc("e3", "encode_block_3.0", p2, 48, H // 4, W // 4, "p3")
        p3 = pool("p3", e3, H // 4, W // 4)
        e4 = enc("e4", "encode_block_4.0", p3, 48, H // 8, W // 8, "p4")
        p4 = pool("p4", e4, H // 8, W // 8)
        e5 = enc("e5", "encode_block_5.0", p4, 48, H // 16, W // 16, "p5")
        p5 = pool("p5", e5, H // 16, W // 16)
        e6 = enc("e6", "encode_block_6.0", p5, 48, H // 32, W // 32)

        def dec(name_a, name_b, la, lb, up_src, c_up, skip, c_skip, h, w, urot=None, urot_smask=None):
            ta = self.act(name_a, N, h, w, 96)
            self._conv(f, L[la], "fwd", View(up_src), c_up, 1, View(skip), c_skip, N, h, w, t3, 96, dst=View(ta),
                       sign_out=sign_of(ta, h, w, 96, c_up + c_skip))       # (mask of the second conv's data gradient)
            if urot is not None:      # the second conv stores un-rotated (fused SSDN_OP_UNROT_FWD): its own output tensor never exists
                self._conv(f, L[lb], "fwd", View(ta), 96, 0, None, 0, N, h, w, t3, 96, dst=None, urot=View(urot), urot_smask=urot_smask)
                return ta, None
            tb = self.act(name_b, N, h, w, 96)
            # (mask of the up-sampled half in the next stage's data gradient -- where that one is a k_cdma launch with the fused UPSUM_BWD)
            self._conv(f, L[lb], "fwd", View(ta), 96, 0, None, 0, N, h, w, t3, 96, dst=View(tb),
                       sign_out=sign_of(tb, h, w, 96, 96) if cdma_serves(2 * h, 2 * w, 96) and 2 * h <= H else None)
            return ta, tb

        d5a, d5b = dec("d5a", "d5b", "decode_block_5.0", "decode_block_5.2", e6, 48, p4, 48, H // 16, W // 16)
        d4a, d4b = dec("d4a", "d4b", "decode_block_4.0", "decode_block_4.2", d5b, 96, p3, 48, H // 8, W // 8)
        d3a, d3b = dec("d3a", "d3b", "decode_block_3.0", "decode_block_3.2", d4b, 96, p2, 48, H // 4, W // 4)
        d2a, d2b = dec("d2a", "d2b", "decode_block_2.0", "decode_block_2.2", d3b, 96, p1, 48, H // 2, W // 2)
        nin = 384 if bs else 96
        if bs:
            u = self.act("u", B, H, W, 384)
            # with training: one LeakyReLU sign byte per 8 channels of d1b, for the fused un-rotation of the backward pass
            fused_unrot = H == W and H & (H - 1) == 0 and (B * H * W) % 256 == 0
            smk = self.T("smk_d1b", "u8", (N, H, W, 96 // 8)) if (self.train and SIGN_BYTES and fused_unrot) else None
            # decode_block_1.2 stores straight into `u` where the library's k_cdma serves it (csrc/conv_dma.hip::conv_dma_eligible:
            # >= one 16x16 tile per CU; the engine cross-checks with ssdn_conv_fuses_urot).  d1b then never exists, so a training
            # plan needs the sign bytes for its backward pass.
            fused_urot = (FUSE_UNROT_FWD and H == W and H % 16 == 0 and cdma_fills(N * (H // 16) * (W // 16), self.dev_cus, H, W) and
                          B * H * W * 384 * 2 < (1 << 31) and (smk is not None or not self.train))
            d1a, d1b = dec("d1a", "d1b", "decode_block_1.0", "decode_block_1.2", d2b, 96, x16, 16, H, W,
                           urot=u if fused_urot else None, urot_smask=smk if fused_urot else None)
            if not fused_urot:
                f.append(Op("unrot_fwd", dict(src=View(d1b), dst=View(u), B=B, P=H, C=96, smask=smk)))
            head_in = u
        else:
            d1a, d1b = dec("d1a", "d1b", "decode_block_1.0", "decode_block_1.2", d2b, 96, x16, 16, H, W)
            head_in = d1b
        na = self.act("na", B, H, W, nin)
        # the 384-channel head layer of a training plan leaves LeakyReLU sign bytes for the data gradient of output_block.2 (k_gdma serves
        # both: csrc/gemm_dma.hip::gemm_dma_signs; the engine cross-checks with ssdn_conv_signs)
        smk_na = self.T("smk_na", "u8", (B, H, W, nin // 8)) if (self.train and SIGN_BYTES_HEAD and nin == 384 and (B * H * W) % 256 == 0) else None
        self._conv(f, L["output_block.0"], "fwd", View(head_in), nin, 0, None, 0, B, H, W, TAPS_1x1, nin, dst=View(na), sign_out=smk_na)
        nb = self.act("nb", B, H, W, 96)
        self._conv(f, L["output_block.2"], "fwd", View(na), nin, 0, None, 0, B, H, W, TAPS_1x1, 96, dst=View(nb))
        out32 = self.T("out32", "f32", (B, self.Cout, H, W))
        self._conv(f, L["output_block.4"], "fwd", View(nb), 96, 0, None, 0, B, H, W, TAPS_1x1, self.Cout, dst32=out32, act=False)

        # ---------------- weight shadows ----------------
        for l in self.layers:
            need_d = l.name != "encode_block_1.0"
            # dgrad GEMM: rows = the forward input-channel slots that need a gradient
            rows = l.Ktot
            if l.name == "decode_block_1.0":
                rows = 96   # the network input needs no gradient
            Mpad_d = ceil_to(rows, 32)
            self.T("wf/" + l.name, "f16", (l.ntaps * l.Mpad_f * l.Ktot,))
            if need_d:
                self.T("wd/" + l.name, "bf16", (l.ntaps * Mpad_d * l.Kd,))
            # chunk-major pre-swizzled copies for the persistent LDS-DMA convolution (3x3 layers whose reduction length is
            # 48 n or 48 n + 16): see ssdn_conv_args.wc
            cm_f = l.ntaps == 9 and l.Ktot % 48 in (0, 16)
            cm_d = need_d and l.ntaps == 9 and l.Kd % 48 in (0, 16)
            if cm_f:
                self.T("wfc/" + l.name, "f16", (l.ntaps * l.Mpad_f * l.Ktot,))
            if cm_d:
                self.T("wdc/" + l.name, "bf16", (l.ntaps * Mpad_d * l.Kd,))
            self.pack.append(Op("wpack", dict(layer=l.name, M=l.M, cin=l.cin, ntaps=l.ntaps, c0=l.c0, c1_real=l.c1_real,
                                              Mpad_f=l.Mpad_f, Ktot=l.Ktot, Mpad_d=Mpad_d, Kd=l.Kd, need_d=need_d,
                                              cm_f=cm_f, cm_d=cm_d)))
        if not self.train:
            return

        # ---------------- backward ----------------
        b = self.bwd
        rt3 = [(-dy, -dx) for dy, dx in t3]   # data gradient reads dZ at (y - dy, x - dx) with the same tap index
        self.T("g32", "f32", (B, self.Cout, H, W))       # d mean(LOSS) / d net_out, written by the loss kernels
        self.T("gmax", "u32", (4,))
        self.T("scale", "f32", (4,))
        gz = self.grad("gz", B, H, W, 16)
        b.append(Op("grad_pack", dict(g=self.prefix + "g32", dst=View(gz), N=B, C=self.Cout, H=H, W=W, cpad=16)))

        def dgrad(layer, src, csrc, n, h, w, taps, M, dst, mask=None, add=None, **kw):
            return self._conv(b, L[layer], "dgrad", View(src), csrc, 0, None, 0, n, h, w, taps, M, dst=dst, bias=False, act=False,
                              mask=mask, add=add, **kw)

        # output_block.4 : 96 -> Cout
        lo4 = L["output_block.4"]
        self._wgrad(lo4, View(gz), 16, View(nb), 96, 0, None, 0, 96, B, H, W, TAPS_1x1)
        g_nb = self.grad("g_nb", B, H, W, 96)
        dgrad("output_block.4", gz, 16, B, H, W, TAPS_1x1, 96, View(g_nb), mask=View(nb))
        # output_block.2 : nin -> 96
        lo2 = L["output_block.2"]
        blocks = list(range(0, nin, 96))
        self._wgrad(lo2, View(g_nb), 96, View(na), nin, 0, None, 0, nin, B, H, W, TAPS_1x1, cblocks=blocks)
        g_na = self.grad("g_na", B, H, W, nin)
        dgrad("output_block.2", g_nb, 96, B, H, W, TAPS_1x1, nin, View(g_na), mask=View(na), mask_sign=smk_na)
        # output_block.0 : nin -> nin
        lo0 = L["output_block.0"]
        self._wgrad(lo0, View(g_na), 96, View(head_in), nin, 0, None, 0, nin, B, H, W, TAPS_1x1, cblocks=blocks, mblocks=nin // 96)
        g_d1b = self.grad("g_d1b", N, H, W, 96)
        if bs and fused_unrot:
            # the data-gradient GEMM scatters its four 96-channel blocks straight into the rotated tensors (fused
            # SSDN_OP_UNROT_BWD, k_gdma; the library's rule: csrc/gemm_dma.hip::gemm_dma_eligible)
            dgrad("output_block.0", g_na, 384, B, H, W, TAPS_1x1, 384, None, unrot=View(g_d1b),
                  unrot_mask=View(d1b) if d1b is not None else None, unrot_smask=smk)
        elif bs:
            g_u = self.grad("g_u", B, H, W, 384)
            dgrad("output_block.0", g_na, 384, B, H, W, TAPS_1x1, 384, View(g_u))
            b.append(Op("unrot_bwd", dict(src=View(g_u), dst=View(g_d1b), mask=View(d1b), B=B, P=H, C=96)))
        else:
            dgrad("output_block.0", g_na, 96, B, H, W, TAPS_1x1, 96, View(g_d1b), mask=View(d1b))

        def dec_bwd(la, lb, ta, tb, g_tb, up_src, c_up, skip, c_skip, c_skip_real, h, w, tag, need_skip_grad=True):
            """backward of conv_b(conv_a(cat(up(up_src), skip))); returns (g_up_src, view of the skip gradient)."""
            self._wgrad(L[lb], View(g_tb), 96, View(ta), 96, 0, None, 0, 96, N, h, w, t3)
            g_ta = self.grad("g_" + tag + "a", N, h, w, 96)
            dgrad(lb, g_tb, 96, N, h, w, rt3, 96, View(g_ta), mask=View(ta), mask_sign=signs.get(ta))
            # one input tensor per weight-gradient launch (the kernel's row loads have one base address)
            self._wgrad(L[la], View(g_ta), 96, View(up_src), c_up, 1, None, 0, c_up, N, h, w, t3, c_off=0, with_bias=True)
            self._wgrad(L[la], View(g_ta), 96, None, 0, 0, View(skip), c_skip, c_skip_real, N, h, w, t3, c_off=c_up, with_bias=False)
            Mx = c_up + (c_skip if need_skip_grad else 0)
            dxs = self.grad("dxs_" + tag, N, h, w, Mx)
            g_up = self.grad("g_up_" + tag, N, h // 2, w // 2, c_up)
            # (the 2x2 sum + LeakyReLU' of the up-sampled half is fused into the data-gradient conv where the library can)
            if not dgrad(la, g_ta, 96, N, h, w, rt3, Mx, View(dxs), upsum=View(g_up), upsum_mask=View(up_src), upsum_c=c_up,
                         upsum_mask_sign=signs.get(up_src)):
                b.append(Op("upsum_bwd", dict(src=View(dxs), mask=View(up_src), dst=View(g_up), N=N, H=h // 2, W=w // 2, C=c_up)))
            return g_up, (View(dxs, c_up) if need_skip_grad else None)

        g_d2b, _ = dec_bwd("decode_block_1.0", "decode_block_1.2", d1a, d1b, g_d1b, d2b, 96, x16, X16_SLOTS, C, H, W, "d1", need_skip_grad=False)
        g_d3b, sk_p1 = dec_bwd("decode_block_2.0", "decode_block_2.2", d2a, d2b, g_d2b, d3b, 96, p1, 48, 48, H // 2, W // 2, "d2")
        g_d4b, sk_p2 = dec_bwd("decode_block_3.0", "decode_block_3.2", d3a, d3b, g_d3b, d4b, 96, p2, 48, 48, H // 4, W // 4, "d3")
        g_d5b, sk_p3 = dec_bwd("decode_block_4.0", "decode_block_4.2", d4a, d4b, g_d4b, d5b, 96, p3, 48, 48, H // 8, W // 8, "d4")
        g_e6, sk_p4 = dec_bwd("decode_block_5.0", "decode_block_5.2", d5a, d5b, g_d5b, e6, 48, p4, 48, 48, H // 16, W // 16, "d5")

        def enc_bwd(lname, g_out, src, h, w, tag, skip_add, act_prev):
            """backward of conv(pool(act_prev)): wgrad, then data gradient w.r.t. `src` (= pooled tensor), plus the
            decoder's skip gradient, routed through the max-pool to the pre-activation of the previous conv."""
            self._wgrad(L[lname], View(g_out), 48, View(src), 48, 0, None, 0, 48, N, h, w, t3)
            g_p = self.grad("g_p_" + tag, N, h, w, 48)
            dgrad(lname, g_out, 48, N, h, w, rt3, 48, View(g_p), add=skip_add)
            g_prev = self.grad("g_e_" + tag, N, 2 * h, 2 * w, 48)
            b.append(Op("pool_bwd", dict(act=View(act_prev), dpool=View(g_p), dz=View(g_prev), N=N, H=2 * h, W=2 * w, C=48, shifted=int(bs),
                                         route=routes.get(act_prev))))
            return g_prev

        g_e5 = enc_bwd("encode_block_6.0", g_e6, p5, H // 32, W // 32, "6", None, e5)
        g_e4 = enc_bwd("encode_block_5.0", g_e5, p4, H // 16, W // 16, "5", sk_p4, e4)
        g_e3 = enc_bwd("encode_block_4.0", g_e4, p3, H // 8, W // 8, "4", sk_p3, e3)
        g_e2 = enc_bwd("encode_block_3.0", g_e3, p2, H // 4, W // 4, "3", sk_p2, e2)
        g_e1 = enc_bwd("encode_block_2.0", g_e2, p1, H // 2, W // 2, "2", sk_p1, e1)
        # encode_block_1.2 (e0 -> e1) and encode_block_1.0 (x16 -> e0)
        self._wgrad(L["encode_block_1.2"], View(g_e1), 48, View(e0), 48, 0, None, 0, 48, N, H, W, t3)
        g_e0 = self.grad("g_e0", N, H, W, 48)
        dgrad("encode_block_1.2", g_e1, 48, N, H, W, rt3, 48, View(g_e0), mask=View(e0), mask_sign=signs.get(e0))
        self._wgrad(L["encode_block_1.0"], View(g_e0), 48, None, 0, 0, View(x16), 16, C, N, H, W, t3)
        if getattr(self, "_mega_ops", None):
            self._plan_mega()

    # ---- chip-wide weight-gradient launches ------------------------------------------------------------------
    def wgrad_group_of(self, layer_name: str, skip_half: bool = False) -> int:
        """index of the chip-wide launch the weight gradients of `layer_name` belong to (WGRAD_MEGA).  skip_half: the op covers the
        skip-connection half of a decoder stage's first layer (its own SSDN_OP_WGRAD / SSDN_OP_WREDUCE pair: "<layer>/skip")."""
        if WGRAD_MEGA == "split":
            return 0 if (layer_name + ("/skip" if skip_half else "/")).startswith(SPLIT_GROUP0) else 1
        if WGRAD_MEGA != "buckets":
            return 0
        off = {l.name: l.w_off for l in self.layers}
        a, b = off["decode_block_1.0"], off["decode_block_5.0"]        # (== ssdn.hip.dp.bucket_layers)
        w = off[layer_name]
        return 0 if w >= a else (1 if w >= b else 2)

    @staticmethod
    def is_skip_half(op) -> bool:
        """the SSDN_OP_WGRAD / SSDN_OP_WREDUCE op of the skip-connection half of a decoder stage's first layer"""
        return (op.type == "wgrad" and op.a.get("src0") is None and op.a["layer"].startswith("decode_block")) or \
            (op.type == "wreduce" and op.a.get("c_off", 0) > 0 and not op.a.get("tapblock"))

    def wgrad_group_info(self, g: int):
        """(workgroups the launch of group g is planned for, lane, name of the layer whose DATA-GRADIENT launch it follows or None =
        where the group's last operand appears).  "split": the head layers' weight gradients (a fifth of the work, operands ready
        at the start of the backward pass) run on HALF the CUs on the side lane next to the latency-bound bottom of the U (the
        16x16 layers and the chained 8x8..2x2 layers: ~270 us of launches that cannot fill the chip); everything else as one
        launch on all CUs behind the last data gradient."""
        if WGRAD_MEGA == "split" and g == 0:
            return max(1, (self.cus * SPLIT_HEAD_CUS[0]) // SPLIT_HEAD_CUS[1]), 1, SPLIT_AFTER
        return self.cus, None, None

    @staticmethod
    def _thin_ok(a) -> bool:
        """csrc/wgrad_mfma.hip::wgrad_thin_ok"""
        if not (1 <= a["kreal"] <= 3) or len(a["taps"]) != 9 or a["mblocks"] > 1:
            return False
        if a["H"] % 16 or a["W"] % 16 or a["M"] % 8 or a["Mpad"] > 96 or a["Kpad"] < a["kreal"]:
            return False
        if a["c0"] > 0 and (a["up0"] or a["c1"] > 0):
            return False
        if a["c0"] == 0 and a["c1"] <= 0:
            return False
        if any(c != 0 for c in a["coff"]):
            return False
        dys = [t[0] for t in a["taps"]]
        dxs = [t[1] for t in a["taps"]]
        return max(dys) - min(dys) == 2 and max(dxs) - min(dxs) == 2

    def _mega_candidates(self, a):
        """(tile, ntiles, cycles per tile, fixed cycles per block) of a weight-gradient op planned as ONE column group."""
        C_ = MEGA_COST
        N, H, W = a["N"], a["H"], a["W"]
        ntaps, Mpad, Kpad = len(a["taps"]), a["Mpad"], a["Kpad"]
        MT = Mpad // 32
        slab_bytes = ntaps * Mpad * Kpad * 4
        if self._thin_ok(a):
            ntiles = N * (H // 16) * (W // 16)
            fixed = C_["fixed"] + 9 * Mpad * 32 * 4 / C_["slab_rate"]
            return (4, 4, 0), ntiles, C_["thin"][0] + C_["thin"][1] * MT, fixed
        ctiles = ntaps * Kpad // 32 + 1
        cpw = -(-ctiles // 4)
        sync = False
        try:
            tile, ntiles = choose_wgrad_tile(N, H, W, a["taps"], max(Kpad, a["Ktot"]), Mpad, a["Ktot"], a["M"], 1)
        except ValueError:
            # no tile the kernel could prefetch while another is on the matrix cores (images of 4x4 pixels and below: many short
            # rows): the workgroup fetches every tile synchronously (csrc/wgrad_body.h, WgAux.sync) -- still several tiles per
            # workgroup, i.e. one slab per workgroup instead of one per tile
            tile, ntiles = choose_wgrad_tile(N, H, W, a["taps"], max(Kpad, a["Ktot"]), Mpad, a["Ktot"], a["M"], 1 << 30)
            sync = True
        ltw, lth, ltn = tile
        ksteps = (1 << sum(tile)) // 16
        HW_ = (1 << ltw) + (max(t[1] for t in a["taps"]) - min(t[1] for t in a["taps"]))
        ix = -(-(HW_ * (a["Ktot"] // 8)) // 64)
        both = ix > 4
        static = (not both) and ltn == 0 and ltw >= 3 and ksteps == 8 and MT >= 2 and cpw >= 2 and _wg_stride(max(Kpad, a["Ktot"]) * 2) == 192
        if a["_cblocks"] and both:
            kind = "head_mb" if a["mblocks"] > 1 else "head"
        else:
            kind = "static" if static else "generic"
            if static and ix <= 2 and cpw == 5:
                kind = "static2"
        # the run-time-staged variants with 21 accumulators per wave are not part of the chip-wide launch (register budget): such an
        # op runs as two column groups (every tile is staged by two blocks; these are the layers with few pixels)
        G = 2 if (kind == "generic" and MT * cpw > 16) else 1
        a["csplit"] = G if G > 1 else 0
        cpw = -(-ctiles // (4 * G))
        base, per = C_[kind]
        fixed = C_["fixed"] + slab_bytes / C_["slab_rate"] / G
        return tile, ntiles, ksteps * (base + per * MT * cpw) + (C_["sync_tile"] if sync else 0.0), fixed

    def _plan_mega(self):
        """Size the grid of every weight-gradient op so that the workgroups of its chip-wide launch (one per CU) finish together,
        allocate the slabs, and give every op its cost per block (the library packs the blocks onto the workgroups by it)."""
        groups: Dict[int, list] = {}
        for op, reds in self._mega_ops:
            groups.setdefault(self.wgrad_group_of(op.a["layer"], self.is_skip_half(op)), []).append((op, reds))
        self.mega_makespan = {}
        for gi, members in sorted(groups.items()):
            W = self.wgrad_group_info(gi)[0]
            cand = []
            for op, reds in members:
                tile, ntiles, c_tile, fixed = self._mega_candidates(op.a)
                mb = max(1, op.a["mblocks"])
                G = max(1, op.a.get("csplit", 0))
                cand.append(dict(op=op, reds=reds, tile=tile, ntiles=ntiles, c_tile=c_tile, fixed=fixed, mb=mb * G, nmb=mb))
            # t = the time a block should take: an op with `work` cycles gets ceil(work / (t - fixed)) blocks.  The launch has one
            # block per (op, partition), dispatched longest first onto the CUs as they free up (a block owns its CU): the plan is the
            # t whose simulated dispatch ends first.
            def blocks_for(t):
                res = []
                for c in cand:
                    ns = -(-int(c["ntiles"] * c["c_tile"]) // max(1, int(t - c["fixed"])))
                    ns = max(1, min(c["ntiles"], ns, max(1, W // c["mb"])))
                    res.append((ns, -(-c["ntiles"] // ns) * c["c_tile"] + c["fixed"]))
                return res

            def makespan(blocks):
                import heapq
                items = sorted((cost for (ns, cost), c in zip(blocks, cand) for _ in range(ns * c["mb"])), reverse=True)
                free = [0.0] * W
                heapq.heapify(free)
                end = 0.0
                for it in items:
                    t0 = heapq.heappop(free)
                    heapq.heappush(free, t0 + it)
                    end = max(end, t0 + it)
                return end, sum(items) / W
            lower = sum(c["ntiles"] * c["c_tile"] * c["mb"] for c in cand) / float(W)
            best = None
            for k in range(120):
                t = lower * (1.0 + 0.01 * k) + max(c["fixed"] for c in cand)
                bl = blocks_for(t)
                ms = makespan(bl)
                if best is None or ms[0] < best[0][0]:
                    best = (ms, bl)
            self.mega_makespan[gi] = best[0]
            for c, (ns, cost) in zip(cand, best[1]):
                c["ns"], c["cost"] = ns, cost
            for c in cand:
                a = c["op"].a
                ns, mb = c["ns"], c["nmb"]
                a["nslabs"], (a["ltw"], a["lth"], a["ltn"]) = ns, c["tile"]
                a["mega"], a["cost"] = W, float(c["cost"])
                ntaps = len(a["taps"])
                a["slab"] = self.T("slab%d" % a["_id"], "f32", (mb * ns * ntaps * a["Mpad"] * a["Kpad"],))
                a["bslab"] = self.T("bslab%d" % a["_id"], "f32", (mb * ns * a["Mpad"],))
                for r in c["reds"]:
                    r.a["nslabs"], r.a["slab"], r.a["bslab"] = ns, a["slab"], a["bslab"]

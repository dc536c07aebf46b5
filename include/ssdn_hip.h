/*
 * ssdn_hip.h -- C-ABI of libssdn_hip.so: the MI355X (gfx950 / CDNA4) hot path of the `ssdn`
 * blind-spot denoising trainer.
 *
 * The reference (COMP6248-Reproducability-Challenge/selfsupervised-denoising) is pure PyTorch and has
 * NO FFI for this path: every "kernel" below replaces an ATen/cuDNN op the
 * reference launches implicitly through torch.nn.  Each entry point cites the reference code it
 * replaces.  The boundary carries plain pointers and sizes only -- no torch types.  The host side
 * (Python, like the reference) keeps the reference's module API (NoiseNetwork / Denoiser) and lowers
 * one forward / training step to a flat list of `ssdn_op` records executed by ssdn_run_ops().
 *
 * Conventions
 *   - all device pointers are caller-owned (the Python host allocates them with torch on the
 *     current device); the library allocates nothing and keeps no state besides the last error.
 *   - activation tensors: NHWC, 16-bit, `cs` = elements per pixel (channel stride), `co` = channel offset
 *     of this view inside the pixel, so concatenations / channel slices are views, never copies.
 *     FORWARD activations are fp16 (11-bit significand: the eval PSNR budget of +-0.05 dB needs it);
 *     every GRADIENT tensor of the backward pass is bf16 (fp32 exponent range: the SSDN loss gradient spans > 7 decades
 *     between pixels when sigma is estimated / Poisson, which no fp16 loss scale can hold) -- accumulation is fp32 always.
 *   - every function returns 0 on success; on failure a negative code and ssdn_last_error() gives text.
 *     No C++ exception crosses the boundary.
 *   - `stream` is a hipStream_t (NULL = the default stream); all work is enqueued asynchronously.
 *   - not thread-safe per stream; one host thread per GPU (one process per GPU).
 */
#ifndef SSDN_HIP_H
#define SSDN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: the functions declared here are its whole dynamic symbol table */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define SSDN_ABI_VERSION 14
#define SSDN_MAX_TAPS 9

/* NHWC fp16 view: element (n,y,x,c) lives at p[((n*H + y)*W + x)*cs + co + c]. */
typedef struct ssdn_view {
    void* p;
    int32_t cs;
    int32_t co;
} ssdn_view;

enum ssdn_op_type {
    SSDN_OP_PACK_INPUT = 1, /* rotate-stack + NCHW f32 -> NHWC f16 */
    SSDN_OP_CONV = 2,       /* implicit-GEMM MFMA convolution (forward or data-gradient role) */
    SSDN_OP_POOL_FWD = 3,
    SSDN_OP_POOL_BWD = 4,
    SSDN_OP_UPSUM_BWD = 5,
    SSDN_OP_UNROT_FWD = 6,
    SSDN_OP_UNROT_BWD = 7,
    SSDN_OP_WGRAD = 8,
    SSDN_OP_WREDUCE = 9,
    SSDN_OP_WPACK = 10,
    SSDN_OP_GRAD_PACK = 11,
    SSDN_OP_HEAD_SSDN = 12,
    SSDN_OP_HEAD_FINAL = 13,
    SSDN_OP_SPATIAL_MEAN = 14,
    SSDN_OP_MSE = 15,
    SSDN_OP_MASK_MSE = 16,
    SSDN_OP_ADAM = 17,
    SSDN_OP_METRICS = 18,  /* per-step loss / PSNR / std-dev sums into a device-resident accumulator (H11) */
    SSDN_OP_ZERO = 19,
    SSDN_OP_EVENT_RECORD = 20, /* hipEventRecord(event) on the op's lane: lets a consumer outside the list (the gradient
                                  all-reduce on its own stream) wait for a PREFIX of the list */
    SSDN_OP_NOISE = 21         /* training patch stream: uint8 clean patches -> noisy / clean / reference fp32 (+ Noise2Void) */
};

/* One record of the op list.  `args` points at the matching ssdn_*_args struct (host memory).
 * lane 0: the op is enqueued on the caller's stream.  Lanes 1..3 are library-owned side streams:
 *   lanes 1 and 3 run an op AFTER everything that precedes it in the list on lane 0 (and on their own lane);
 *   lane 2 runs an op AFTER everything that precedes it in the list on lanes 0, 1 and 3 (and on lane 2).
 * The executor inserts the hipEvent dependencies; a lane never waits for a lane it does not depend on inside a list, but
 * ssdn_run_ops joins all side streams back into the caller's stream before it returns.  The planner guarantees the absence
 * of WAR/WAW hazards between lanes: lanes 1 and 3 carry (alternately) the weight-gradient GEMMs of the backward pass (they
 * read tensors written once per step and write their OWN slab), lane 2 the slab reductions (they read that slab and write
 * the flat gradient, which nothing else touches) -- so the many small, latency-bound backward launches overlap instead of
 * queueing behind each other. */
typedef struct ssdn_op {
    int32_t type;
    int32_t lane;
    const void* args;
} ssdn_op;

/* ---- SSDN_OP_PACK_INPUT ----------------------------------------------------------------------
 * replaces: the 4-rotation batch stacking, noise_network.py:187-189 + utils/data.py:42-67 (flip/transpose/cat),
 * and the NCHW-f32 -> NHWC-f16 layout change.  dst[r*B+b, i, j, c] = rot_r(src[b,c])[i,j], r=0..R-1,
 * rot = (0,90,180,270) with rotate(x,90)[i,j] = x[j, W-1-i]; channels c >= C are written as zero. */
typedef struct ssdn_pack_input_args {
    const float* src; /* [B,C,H,W] f32 */
    ssdn_view dst;    /* [R*B,H,W,cs] f16 */
    int32_t B, C, H, W;
    int32_t R;        /* 4 (blind-spot) or 1 */
    int32_t cpad;     /* channels written (>= C, zero padded) */
} ssdn_pack_input_args;

/* ---- SSDN_OP_CONV ------------------------------------------------------------------------------
 * replaces: ShiftConv2d / nn.Conv2d (+bias) + LeakyReLU(0.1) + nn.Upsample(nearest) + torch.cat
 * (noise_network.py:58,70-156,200-210,241-260) in the forward role, and autograd's conv data-gradient in the
 * backward role.  Computes, for every output pixel (n,y,x) and m < M:
 *     v = bias[m] + sum_{t<ntaps} sum_{k<Ktot} Wp[t][m][k] * IN(n, y+dy[t], x+dx[t], k)
 *     if act:  v = v > 0 ? v : 0.1 v
 *     if add:  v += add(n,y,x,m)
 *     if mask: v *= (mask(n,y,x,m) > 0 ? 1 : 0.1)             (LeakyReLU' of a saved activation)
 * IN is the virtual channel-concatenation [src0 (c0 ch), src1 (c1 ch)], zero outside [0,H)x[0,W);
 * src0 is read at (y>>1, x>>1) when up0 (nearest 2x upsample folded into the load).
 * Wp is fp16 [ntaps][Mpad][Ktot] (packed by SSDN_OP_WPACK), Ktot = c0+c1 (multiple of 16), Mpad multiple of 32.
 * Output: dst (16-bit NHWC view) or, when dst32 != NULL, fp32 NCHW planar [N][M][H][W] (net_out).
 * bf16 = 0: src0/src1/w/dst/add are fp16 (forward role).  bf16 = 1: they are bf16 (data-gradient role); `mask` is
 * always an fp16 saved activation. */
typedef struct ssdn_conv_args {
    ssdn_view src0;
    ssdn_view src1;
    int32_t c0, c1, up0;
    int32_t N, H, W;
    int32_t ntaps;
    int32_t dy[SSDN_MAX_TAPS];
    int32_t dx[SSDN_MAX_TAPS];
    const void* w; /* fp16 [ntaps][Mpad][Ktot] */
    int32_t M, Mpad, Ktot;
    const float* bias; /* [M] or NULL */
    int32_t act;
    ssdn_view mask; /* p == NULL: none */
    ssdn_view add;  /* p == NULL: none */
    ssdn_view dst;
    float* dst32;
    /* tiling chosen by the host (see ssdn/hip/graph.py): tile = 2^ltn images x 2^lth rows x 2^ltw cols = 256 px */
    int32_t ltw, lth, ltn;
    int32_t kc; /* channel chunk staged in LDS at a time (multiple of 16, divides Ktot) */
    int32_t bf16;
    const void* wc; /* optional second copy of the weights for the persistent LDS-DMA kernel (k_cdma), or NULL: chunk-major and
                       pre-swizzled, [tap][chunk][Mpad][kc] with kc = 48 (chunks of 48 input channels, then one optional 16-channel
                       chunk); the 16-byte piece p of row m is stored at piece p ^ ((m >> 3) & 1) -- exactly the LDS image, so a
                       (tap, chunk) slice is one linear, fully coalesced DMA (SSDN_OP_WPACK writes it: wfc / wdc) */
    int32_t kreal; /* real (un-padded) input channels among the Ktot slots (0: = Ktot): the algorithmic flop count of the profiler, and a
                      CONTRACT for the forward role -- the input channels >= kreal are zero (SSDN_OP_PACK_INPUT pads with zeros), so a
                      launch may skip them (k_conv_thin, csrc/conv_thin.hip, serves Ktot = 16 with kreal <= 3) */
    /* Fused Shift2d((1,0)) + nn.MaxPool2d(2) of the conv's activated 16-bit output (SSDN_OP_POOL_FWD semantics; the reference:
     * noise_network.py:64-67): pool.p != NULL makes the epilogue ALSO write pooled[N,H/2,W/2,M] -- the max is taken over the
     * rounded fp16 values the epilogue stores, so the result is bit-identical to SSDN_OP_POOL_FWD applied to dst.  Only the
     * launches ssdn_conv_fuses_pool() accepts can do this (today: forward layers whose 256-pixel tile is made of whole images,
     * i.e. 16x16 pixels and below at BASELINE sizes); any other launch with pool.p set is an error. */
    ssdn_view pool;
    int32_t pool_shifted;
    /* Fused SSDN_OP_UPSUM_BWD (data-gradient role of a decoder's first conv, whose input is cat(upsample(x), skip)):
     * upsum.p != NULL => output channels [0, upsum_c) are NOT stored to dst; their 2x2 sums (over the bf16-rounded values, in
     * fp32, window scan order) times LeakyReLU'(upsum_mask) are stored to upsum[N,H/2,W/2,..] -- bit-identical to
     * SSDN_OP_UPSUM_BWD applied to the tensor this launch would have stored.  Channels >= upsum_c (the skip gradient) go to
     * dst as usual.  Needs bf16 = 1, no mask / add, even H, W; upsum_c a multiple of 8 -- and of 96 for the launches that
     * run k_cdma.  ssdn_conv_fuses_upsum() tells whether a launch can do it; otherwise upsum.p set is an error. */
    ssdn_view upsum;
    ssdn_view upsum_mask;
    int32_t upsum_c;
    /* Fused SSDN_OP_UNROT_BWD (data gradient of the first 1x1 head layer of a blind-spot network, whose input is the
     * un-rotated stack of the four rotated feature maps): unrot.p != NULL => nothing is stored to dst; output channel block r
     * (of M / 4 channels) of pixel (b, i, j) goes to unrot[(r * N + b), u - 1, v, :] times LeakyReLU'(unrot_mask there), with
     * (u, v) the rotated coordinates of SSDN_OP_UNROT_FWD; the pixels with u == 0 (cut off by the shift) write the zero row
     * y = H - 1 instead -- the complete tensor SSDN_OP_UNROT_BWD would have produced, bit-identical.  N of the conv = batch;
     * needs bf16 = 1, a 1x1 layer with M = 384, H == W a power of two, no mask / add (ssdn_conv_fuses_unrot()). */
    ssdn_view unrot;
    ssdn_view unrot_mask;
    /* optional: the sign bytes SSDN_OP_UNROT_FWD wrote for unrot_mask's tensor (ssdn_unrot_args.smask); when set they are read
     * instead of unrot_mask (same result: only the sign of the activation enters LeakyReLU'). */
    const void* unrot_smask;
    /* fused SSDN_OP_UNROT_FWD (forward role; replaces: rotate(x, -90k) + Shift2d((1,0)) + torch.cat of noise_network.py:213-222 applied
     * to this layer's output).  When urot.p is set the launch is the layer over the four rotated copies of a batch (N = 4B images
     * r*B + b, H == W = P, M = Mpad = 96): the output pixel (rB+b, y, x), y <= P-2, is stored at urot[b, i, j, r*M + m] -- the
     * place SSDN_OP_UNROT_FWD would have moved it to -- and nothing goes to dst (which may be null); row y = P-1 is dropped and the
     * rows of urot the one-row shift leaves empty are NOT written (they stay zero in a zero-initialised tensor).  urot_smask,
     * optional: the LeakyReLU sign bytes of the output (ssdn_unrot_args.smask).  ssdn_conv_fuses_urot() tells whether a launch can. */
    ssdn_view urot;
    void* urot_smask;
    /* LeakyReLU sign bytes instead of saved activations in the backward pass (both optional; ssdn_conv_signs() tells whether a launch
     * honours them): sign_out (forward role, 16-bit dst) -- the launch also writes one byte per 8 output channels, bit q of byte
     * [pixel][k] = (output channel 8k+q > 0), [N*H*W][M/8]; mask_sign (data-gradient role) -- such bytes for the tensor `mask` views,
     * read instead of it (same result: only the sign of the activation enters LeakyReLU'; 1/16 of the bytes); upsum_mask_sign
     * (data-gradient role with the fused SSDN_OP_UPSUM_BWD) -- such bytes for the tensor `upsum_mask` views, [N*(H/2)*(W/2)][upsum_c/8]. */
    void* sign_out;
    const void* mask_sign;
    const void* upsum_mask_sign;
} ssdn_conv_args;

/* ---- SSDN_OP_POOL_FWD / SSDN_OP_POOL_BWD ----------------------------------------------------
 * replaces: Shift2d((1,0)) + nn.MaxPool2d(2) (noise_network.py:64-67, models/utility.py:37-53) and its autograd
 * backward fused with the LeakyReLU backward of the producing conv.
 * shifted: window rows {2i-1, 2i}, row -1 is a literal 0 that takes part in the max; else rows {2i, 2i+1}.
 * BWD: dz(n,y,x,c) = (first position in window scan order whose value equals the pooled max is (y,x))
 *                    ? dpool(n,i,j,c) * (act > 0 ? 1 : 0.1) : 0;  if the zero pad row wins the gradient is dropped.
 * route (optional, uint32 [N,H/2,W/2,C/8]): FWD also writes one word per (window, 8 channels) -- nibble q of channel 8k+q: bits 0-1 = scan
 *   position of the first maximum, bit 2 = the zero pad row holds it, bit 3 = the maximum is > 0; BWD launched on its own reads the word
 *   instead of the four activation pieces of the window (same result, 4 bytes instead of 64; a chained launch keeps reading `act`). */
typedef struct ssdn_pool_args {
    ssdn_view act;    /* [N,H,W,C] full-res post-activation */
    ssdn_view pooled; /* [N,H/2,W/2,C]  (FWD: output, BWD: input) */
    ssdn_view dpool;  /* BWD only */
    ssdn_view dz;     /* BWD only: [N,H,W,C] */
    int32_t N, H, W, C;
    int32_t shifted;
    void* route;
} ssdn_pool_args;

/* ---- SSDN_OP_UPSUM_BWD ----------------------------------------------------------------------
 * replaces: autograd backward of nn.Upsample(nearest, 2x) + LeakyReLU backward of the producer.
 * dst(n,i,j,c) = (sum_{a,b<2} src(n,2i+a,2j+b,c)) * (mask(n,i,j,c) > 0 ? 1 : 0.1); dims are those of dst. */
typedef struct ssdn_upsum_args {
    ssdn_view src;  /* [N,2H,2W,..] */
    ssdn_view mask; /* [N,H,W,..] */
    ssdn_view dst;  /* [N,H,W,..] */
    int32_t N, H, W, C;
} ssdn_upsum_args;

/* ---- SSDN_OP_UNROT_FWD / SSDN_OP_UNROT_BWD --------------------------------------------------
 * replaces: final Shift2d((1,0)), chunk(4), un-rotate (0,270,180,90), cat(dim=1) (noise_network.py:213-222).
 * FWD: dst[b,i,j,r*C+c] = S_r(src)[..], S_r[y,x] = y>=1 ? src[r*B+b, y-1, x] : 0, rotated back by (0,270,180,90).
 * BWD: the adjoint, times LeakyReLU'(act) of the tensor that was un-rotated. */
typedef struct ssdn_unrot_args {
    ssdn_view src;  /* FWD: [4B,P,P,C] ; BWD: [B,P,P,4C] (gradient) */
    ssdn_view dst;  /* FWD: [B,P,P,4C] ; BWD: [4B,P,P,C] */
    ssdn_view mask; /* BWD only: [4B,P,P,C] saved activation */
    int32_t B, P, C;
    /* FWD, optional: [4B,P,P,C/8] bytes, bit q of byte k = (src channel 8k+q > 0) -- the LeakyReLU sign of every pixel the
     * un-rotation reads (rows y <= P-2; row P-1 is cut off by the shift and left untouched).  The fused backward
     * (ssdn_conv_args.unrot_smask) reads these 12 bytes per pixel instead of the 192-byte activation. */
    void* smask;
} ssdn_unrot_args;

/* ---- SSDN_OP_WGRAD --------------------------------------------------------------------------
 * replaces: autograd's conv weight/bias gradient.  dz is bf16, IN is fp16 (converted to bf16 while it is staged in
 * LDS; products are exact in the fp32 accumulator).  Every workgroup s < nslabs reduces its share of pixels into
 *   slab[s][t][m][k] = sum_pixels dz(n,y,x,m) * IN(n, y+dy[t], x+dx[t], coff[t]+k)   (fp32, [nslabs][ntaps][Mpad][Kpad])
 *   bslab[s][m]      = sum_pixels dz(n,y,x,m)
 * IN as in SSDN_OP_CONV.  Deterministic: fixed pixel->workgroup assignment, SSDN_OP_WREDUCE sums slabs in order. */
typedef struct ssdn_wgrad_args {
    ssdn_view dz; /* [N,H,W,..] gradient w.r.t. the conv's pre-activation output, M channels */
    ssdn_view src0;
    ssdn_view src1;
    int32_t c0, c1, up0;
    int32_t N, H, W;
    int32_t ntaps;
    int32_t dy[SSDN_MAX_TAPS];
    int32_t dx[SSDN_MAX_TAPS];
    int32_t coff[SSDN_MAX_TAPS]; /* channel offset of tap t inside IN: lets the "taps" of a 1x1 layer be channel blocks */
    int32_t M, Mpad, Ktot, Kpad; /* Ktot = channels staged; every tap covers channels [coff[t], coff[t]+Kpad) */
    float* slab;
    float* bslab;
    int32_t nslabs;
    int32_t ltw, lth, ltn; /* pixel tile per iteration: 2^ltn x 2^lth x 2^ltw */
    int32_t csplit;        /* column groups G (0 or 1: none).  The output is ntaps*Kpad/32 + 1 column tiles of 32 (tile 0 = the
                              bias column); with G > 1 a workgroup owns ceil(tiles / G) consecutive ones and the grid is
                              nslabs x G: for layers with few pixels, where writing a full slab per workgroup (~10 B per
                              clock per CU) and reading it back would dominate */
    int32_t mblocks;       /* >= 1: the launch covers mblocks blocks of M output channels (dz channels [b*M, b*M+M) of the dz
                              view, slabs [b*nslabs, (b+1)*nslabs) of `slab` / `bslab`): grid = mblocks x nslabs workgroups,
                              the mblocks workgroups that stream the same pixels placed on one XCD so that the input tile
                              reaches HBM once (1x1 head layers: 4 blocks of 96) */
    int32_t kreal;         /* real (un-padded) input channels among the Ktot slots, or 0 (unknown).  1..3 real channels under a
                              3x3 window (the network's first layer and the image half of decode_block_1.0) are served by the
                              im2col kernel k_wgrad_thin: 9 * kreal + 1 <= 32 GEMM columns instead of 9 x 32 */
    int32_t mega;          /* > 0: the op is planned as part of ONE merged launch (k_wgrad_mega) together with the SSDN_OP_WGRAD ops next
                              to it in the list that carry the same value (the number of CUs the planner sized the group's grids
                              for): every block of every op's grid becomes one block of the merged launch, sorted by `cost`,
                              longest first, so that the CUs pick them up in that order; each block runs the code of its op's own
                              launch (bit-identical slabs).  0: the op is launched on its own */
    float cost;            /* planner's estimate of the time of ONE block of this op's grid, any unit (only the ratios matter) */
} ssdn_wgrad_args;

/* ---- SSDN_OP_WREDUCE ------------------------------------------------------------------------
 * gw[m][cin][ky][kx] (fp32 OIHW, the checkpoint layout) = inv_scale * sum_s slab[s][t][m][k(cin)],
 * gb[m] = inv_scale * sum_s bslab[s][m];  k(cin) = cin (cin < c0) else c0 + (cin - c0) i.e. padding removed:
 * real input channels are [0,c0) and [c0, c0+c1_real).  inv_scale is read from device memory (loss-scale word).
 * Slabs are summed in a fixed order (groups of 32 in index order, then the groups in order): bit-reproducible.
 * NOTE: the slab buffer is used as scratch (partial sums are written back into it). */
typedef struct ssdn_wreduce_args {
    const float* slab;
    const float* bslab;
    int32_t nslabs, ntaps, M, Mpad, Kpad;
    int32_t cin;      /* real input channels covered by this slab (k < cin are real, the rest is padding) */
    int32_t cin_full; /* input channels of the whole weight tensor (row length of gw) */
    int32_t m_off, c_off; /* this slab is the block gw[m_off.., c_off..] (the 1x1 head layers are computed in blocks) */
    int32_t tapblock;     /* 1: the slab's "taps" are channel blocks of a 1x1 layer: input channel = c_off + t*Kpad + k */
    float* gw;
    float* gb; /* may be NULL (bias gradient is produced by one block column only) */
    const float* inv_scale; /* device scalar, may be NULL (=1) */
} ssdn_wreduce_args;

/* ---- SSDN_OP_WPACK --------------------------------------------------------------------------
 * fp32 OIHW master weights -> the two fp16 shadows the MFMA kernels read:
 *   wf[t][m][k]  (forward role)  = W[m][cin(k)][t]  for m < M, k real; 0 in the padding   [ntaps][Mpad_f][Ktot]
 *   wd[t][c][m]  (dgrad role)    = W[m][cin(c)][t]   stored as bf16                       [ntaps][Mpad_d][Kd]
 * k -> cin: k < c0 ? k : (k - c0 < c1_real ? c0 + k - c0 : padding). */
typedef struct ssdn_wpack_args {
    const float* w; /* [M][cin][ntaps] */
    void* wf;
    void* wd; /* may be NULL (first layer needs no data gradient) */
    int32_t M, cin, ntaps;
    int32_t c0, c1_real;
    int32_t Mpad_f, Ktot; /* forward shadow dims */
    int32_t Mpad_d, Kd;   /* dgrad shadow dims: Mpad_d >= Ktot, Kd >= M (multiple of 16) */
    void* wfc;            /* optional chunk-major pre-swizzled copies of wf / wd (see ssdn_conv_args.wc), same sizes; NULL: none. */
    void* wdc;            /* Only for Ktot (resp. Kd) = 48 n or 48 n + 16 */
} ssdn_wpack_args;

/* ---- SSDN_OP_GRAD_PACK ----------------------------------------------------------------------
 * fp32 NCHW gradient w.r.t. net_out -> bf16 NHWC [N,H,W,cpad] (zero padded channels).  bf16 keeps the fp32 exponent
 * range, so there is NO loss scaling; scale_out[0] = scale_out[1] = 1 is written for SSDN_OP_WREDUCE's inv_scale input.
 * gmax (float bits of max |g|, written by the loss kernels with atomicMax) is kept as an overflow / NaN sentinel: a running maximum --
 * the host clears it (SSDN_OP_ZERO) when it wants a per-step figure. */
typedef struct ssdn_grad_pack_args {
    const float* g; /* [N,C,H,W] */
    ssdn_view dst;
    int32_t N, C, H, W, cpad;
    const uint32_t* gmax; /* float bits of max |g| */
    float* scale_out;     /* [2] */
} ssdn_grad_pack_args;

/* ---- SSDN_OP_HEAD_SSDN / SSDN_OP_HEAD_FINAL -------------------------------------------------
 * replaces: Denoiser._ssdn_pipeline (denoiser.py:218-397) and its autograd backward:
 * per-pixel Gaussian posterior algebra in closed form (3x3 SPD inverse/det by adjugate), fp32.
 *   style: 0 gauss, 1 poisson.   mode: 0 known, 1 const (one learnable scalar), 2 var (sigma-net, one value / sample)
 *   noise_param[b]: sigma (gauss) or lambda (poisson) for mode known.   est_raw: [1] (const) or [B] (var), pre-softplus.
 * Outputs (any may be NULL): mu / pme [B,C,H,W], model_std [B,H,W], noise_std [B] (gauss) or [B,H,W] (poisson),
 *   g_net_out [B,Cout,H,W] = d mean(LOSS) / d net_out, partial[b][chunk][2] = {sum loss, sum dloss/dest_raw} per workgroup,
 *   gmax = atomicMax of |g| float bits.
 * HEAD_FINAL sums the partials in fixed order: loss[b] = mean over pixels; g_est (const: [1], var: [B]);
 *   for var also fills g_sigma_out [B,1,H,W] with g_est[b]/(H*W) (gradient of the spatial mean) and folds it into gmax2. */
typedef struct ssdn_head_args {
    const float* net_out; /* [B,Cout,H,W] */
    const float* noisy;   /* [B,C,H,W] */
    const float* noise_param;
    const float* est_raw;
    int32_t B, C, H, W;
    int32_t style, mode;
    int32_t want_grad;
    float* mu;
    float* pme;
    float* model_std;
    float* noise_std;
    float* g_net_out;
    float* partial;
    int32_t nchunks;
    uint32_t* gmax;
} ssdn_head_args;

typedef struct ssdn_head_final_args {
    const float* partial;
    int32_t B, nchunks, H, W;
    int32_t mode;
    float* loss;        /* [B] */
    float* g_est;       /* [1] or [B] */
    float* g_sigma_out; /* var: [B,1,H,W] */
    uint32_t* gmax2;    /* var: max |g_sigma_out| bits */
} ssdn_head_final_args;

/* mean over H,W of a [B,1,H,W] fp32 map (denoiser.py:264) */
typedef struct ssdn_spatial_mean_args {
    const float* src;
    float* dst;
    int32_t B, HW;
} ssdn_spatial_mean_args;

/* ---- SSDN_OP_MSE / SSDN_OP_MASK_MSE ---------------------------------------------------------
 * replaces: Denoiser._mse_pipeline (denoiser.py:153-154) / loss_mask_mse (utils/n2v_loss.py:6-17, quirk kept:
 * coordinates of batch element 0 are used for every element, squared errors summed over coordinates, mean over C).
 * loss[b]; g = d mean_b(loss) / d out (fp32 NCHW); gmax as above. */
typedef struct ssdn_mse_args {
    const float* out; /* [B,C,H,W] */
    const float* ref;
    const int64_t* coords; /* MASK_MSE: [ncoords][2] (row, col) of batch element 0 */
    int32_t ncoords;
    int32_t B, C, H, W;
    float* loss;
    float* g;
    uint32_t* gmax;
} ssdn_mse_args;

/* ---- SSDN_OP_ADAM ---------------------------------------------------------------------------
 * replaces: torch.optim.Adam.step over all parameters (train.py:100-107,202): one fused pass over the flat fp32
 * master buffer.  p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps), bc = 1 - beta^step (host computes bc1, bc2). */
typedef struct ssdn_adam_args {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
    float lr, b1, b2, eps, bc1, bc2;
    float gscale; /* multiplies g first (1/world_size for data parallel) */
} ssdn_adam_args;

/* ---- SSDN_OP_METRICS ------------------------------------------------------------------------
 * H11: the per-step metric accumulation of the trainer / evaluator (reference train.py:205-218,553-584, utils/data.py:94-105 with
 * utils/utils.py Metric.add) as ONE launch that adds into a device-resident accumulator; the host reads the 16 floats back when it
 * prints (PRINT_INTERVAL), not every step.  Per sample b, over the valid extent ext[b] = (e1, e2) of the two spatial axes (NULL:
 * the whole image -- evaluation batches are padded, metadata IMAGE_SHAPE):
 *     psnr_x[b]   = -10 log10( mean_{c, i < e1, j < e2} (x[b,c,i,j] - clean[b,c,i,j])^2 )     x = out (IMG_DENOISED), mu (IMG_MU)
 *     mstd[b]     = 255 * mean_{i,j} model_std[b,i,j]          nstd[b] = 255 * noise_std[b] (or its mean over the pixels)
 * and acc[2k] += sum_b value_k[b], acc[2k+1] += count_k for k = 0..4 = loss, psnr_out, psnr_mu, noise_std, model_std (count = B; 1
 * for a noise_std that is one value for the whole batch) -- exactly what `Metric.add` of the reference accumulates (sum over the
 * samples of the per-sample value, sample count).  Deterministic: per-sample values go through `per`, the block that arrives last
 * (ticket in acc[15]) adds them in a fixed order (deterministic).  NULL inputs are skipped. */
typedef struct ssdn_metrics_args {
    const float* out;       /* [B,C,H,W] or NULL */
    const float* mu;        /* [B,C,H,W] or NULL */
    const float* clean;     /* [B,C,H,W] */
    const float* loss;      /* [B] or NULL */
    const float* model_std; /* [B,H,W] or NULL */
    const float* noise_std; /* [noise_n] values (noise_n = 1, B or B*H*W) or NULL */
    const int32_t* ext;     /* [B][2] or NULL */
    int32_t B, C, H, W;
    int32_t noise_n;
    float* per;             /* scratch [B][8] */
    float* acc;             /* [16] accumulator (the host zeroes it when it resets its metrics) */
} ssdn_metrics_args;

typedef struct ssdn_zero_args {
    void* p;
    int64_t bytes;
} ssdn_zero_args;

/* ---- SSDN_OP_EVENT_RECORD -------------------------------------------------------------------
 * Records the caller-owned hipEvent_t `event` on the stream of the op's lane, i.e. after every earlier op of that lane (and,
 * through the lane dependencies, after the lane-0 ops those depend on).  Data parallelism (replaces nn.DataParallel's
 * reduce, denoiser.py:102-110): the host puts one after the last weight-gradient reduction of each gradient bucket and lets
 * the RCCL stream wait for it, so a bucket's all-reduce overlaps the rest of the backward pass. */
typedef struct ssdn_event_args {
    void* event;
} ssdn_event_args;

/* ---- SSDN_OP_NOISE --------------------------------------------------------------------------
 * replaces the per-sample preparation of the training stream, reference ssdn/ssdn/datasets/noise_wrapper.py:66-135 with
 * utils/noise.py:54-107 (add_gaussian / add_poisson) and utils/n2v_ups.py:40-88 (Noise2Void pixel selection), for a whole
 * minibatch of CLEAN uint8 patches already on the device:
 *     clean   = u8 / 255
 *     param   = p_lo                                   if p_lo == p_hi
 *               U[p_lo, p_hi) per (sample, CHANNEL)    otherwise (the reference draws a ranged parameter per leading index of an
 *                                                      unbatched CHW sample, i.e. per channel: noise.py:34-39,55-56)
 *     gauss:    noisy = clean + param * N(0,1)                       (param = std dev as a fraction of 1)
 *     poisson:  noisy = (clean * param + Poisson(1)) / param         (RATE-1 noise on lambda x: the reference's quirk, noise.py:101-104)
 *     clip:     noisy = min(max(noisy, 0), 1)
 * `ref32` (optional) is a second, independent realisation with its own parameter draw (the Noise2Noise / Noise2Void reference).  With n2v_box > 0 the
 * first realisation is manipulated like n2v_ups.manipulate: one pixel (c0, c1) per n2v_box x n2v_box box (c0 stratified over W,
 * c1 over H, uniform inside the box), replaced in every channel by the noisy value of a pixel drawn uniformly from
 * [min(c - r, 0), min(c + r, size - 1)) without c itself, per axis (the reference's window: [0, c + r) in the interior, negative
 * indexes wrap like Python's); coords[b][i * (H / box) + j] = (c0, c1) as the reference returns them (image[:, c1, c0] is the
 * replaced pixel).  Random numbers: Philox4x32-10 keyed by `seed`, counter = (element, stream, offset): stateless, every launch
 * must pass a fresh `offset`.  The distributions are the reference's; the random STREAM is not torch's (parity of the noise is
 * statistical by construction). */
typedef struct ssdn_noise_args {
    const void* clean_u8; /* [B,C,H,W] uint8 */
    float* clean32;       /* out [B,C,H,W] or NULL */
    float* noisy32;       /* out [B,C,H,W] */
    float* ref32;         /* out [B,C,H,W] or NULL */
    float* param;         /* out [B*C] or NULL: the parameter of the first realisation */
    float* param_ref;     /* out [B*C] or NULL: the (independently drawn) parameter of the second realisation */
    int64_t* coords;      /* out [B, (W/box)*(H/box), 2] or NULL (required when n2v_box > 0) */
    int32_t B, C, H, W;
    int32_t style;        /* 0 gauss, 1 poisson */
    int32_t clip;
    float p_lo, p_hi;
    int32_t n2v_box;      /* 0: no manipulation */
    int32_t n2v_radius;   /* sub-patch radius (2 for the reference's 5 x 5) */
    uint64_t seed, offset;
} ssdn_noise_args;

/* Execute `n` ops in order on `stream`.  Returns 0 or a negative error (ssdn_last_error()). */
int ssdn_run_ops(const ssdn_op* ops, int n, void* stream);

/* Order two caller streams: everything enqueued on `then` AFTER this call starts after everything enqueued on `first` BEFORE it (one event
 * of the executor's ring: hipEventRecord on `first`, hipStreamWaitEvent on `then`).  How two independent op lists run concurrently and meet
 * again -- the sigma-estimation network's lists beside the main network's (replaces the sequential execution of the two nn.Modules in
 * ssdn/ssdn/denoiser.py:261-265; DenoiserEngine._fork_sigma): order(main, side); run(list, side); order(side, main). */
int ssdn_stream_order(void* first, void* then);

/* Bytes of dynamic LDS a conv op will request (host-side check of a tiling), or < 0 if the tiling is invalid. */
int ssdn_conv_lds_bytes(const ssdn_conv_args* a);
int ssdn_wgrad_lds_bytes(const ssdn_wgrad_args* a);
/* 1 if the op (mega > 0) can be an entry of the chip-wide weight-gradient launch, 0 if it needs a launch of its own */
int ssdn_wgrad_mega_ok(const ssdn_wgrad_args* a);
/* The kernel variant the library runs the op with: out9 = {thin, MT, CPW, NL, BOTH, PS, KS, RWX, RWD}; returns the id of its
 * instance inside the chip-wide launch (>= 0), -1 if there is none, < -1 on invalid arguments.  (planner / test aid) */
int ssdn_wgrad_variant(const ssdn_wgrad_args* a, int32_t* out9);

/* sizeof() of the args struct for an op type (0: ssdn_op itself); lets a binding verify its struct mirrors */
int ssdn_struct_size(int op_type);
int ssdn_abi_version(void);
const char* ssdn_last_error(void);
/* number of compute units of the current device (used to size persistent grids) */
int ssdn_device_cus(void);

/* In-stream profiler used by bench.py's roofline leg: when enabled for a kernel family, every launch of that family is
 * bracketed by a hipEvent pair ON THE LAUNCH STREAM; ssdn_profile_read() synchronises, returns the summed device time,
 * the number of launches and the ALGORITHMIC flops / bytes those launches stood for (DESIGN.md gives the formulas),
 * and resets the counters.  max_launches = 0 disables.  An event pair costs ~10 us of stream time (measured: it turns
 * back-to-back kernels into kernels with gaps), so ssdn_profile_set_stride(kind, s) brackets only every s-th launch of the
 * family: launches, flops, bytes and time then refer to the SAMPLED launches (a family's launch count per step is not a
 * multiple of the stride bench.py uses, so the sample rotates over all layers). */
#define SSDN_PROF_CONV_MT3 0
#define SSDN_PROF_CONV_MT2 1
#define SSDN_PROF_CONV_MT1 2
#define SSDN_PROF_WGRAD 3
#define SSDN_PROF_GEMM 4      /* k_gdma: 1x1 layers as one-pass LDS-DMA GEMMs */
#define SSDN_PROF_CDMA_MT3 5  /* k_cdma<3,*>: the persistent LDS-DMA 3x3 kernel on 96-channel blocks (the dominant kernel) */
#define SSDN_PROF_CDMA_MT21 6 /* k_cdma<2,*>, k_cdma<1,*> */
#define SSDN_PROF_WGRAD_SIDE 7 /* a k_wgrad_mega launch planned for fewer workgroups than the device has CUs (the side-lane launch) */
#define SSDN_PROF_KINDS 8
int ssdn_profile_enable(int kind, int max_launches);
int ssdn_profile_set_stride(int kind, int stride);
int ssdn_profile_read(int kind, double* total_ms, long long* launches, double* flops, double* bytes);

/* Which kernel serves SSDN_OP_CONV: 0 = k_conv always; 1 (default) = the persistent LDS-DMA kernel k_cdma for 3x3 layers of
 * its shape class with at least one 256-pixel tile per CU; 2 = k_cdma for every layer of its shape class (test aid). */
int ssdn_conv_set_mode(int mode);
/* 1 if SSDN_OP_CONV with these arguments writes the fused max-pool output (ssdn_conv_args.pool), 0 if it cannot. */
int ssdn_conv_fuses_pool(const ssdn_conv_args* a);
/* 1 if SSDN_OP_CONV with these arguments applies the fused SSDN_OP_UPSUM_BWD (ssdn_conv_args.upsum), 0 if it cannot. */
int ssdn_conv_fuses_upsum(const ssdn_conv_args* a);
/* 1 if SSDN_OP_CONV with these arguments applies the fused SSDN_OP_UNROT_BWD (ssdn_conv_args.unrot), 0 if it cannot. */
int ssdn_conv_fuses_unrot(const ssdn_conv_args* a);
/* 1 if SSDN_OP_CONV with these arguments stores its output un-rotated (fused SSDN_OP_UNROT_FWD, ssdn_conv_args.urot), 0 if it cannot. */
int ssdn_conv_fuses_urot(const ssdn_conv_args* a);
/* 1 if SSDN_OP_CONV with these arguments writes sign_out / reads mask_sign (whichever are set), 0 if it cannot. */
int ssdn_conv_signs(const ssdn_conv_args* a);

/* ssdn_run_ops executes a run of consecutive ops on one lane as ONE launch (k_conv_chain, csrc/conv_chain.hip: one workgroup per
 * image walks all layers with every tensor of the run resident in LDS) when the run is a "chain":
 *   - SSDN_OP_CONV ops, all forward (bf16 = 0: bias + LeakyReLU, 16-bit output, optional fused max-pool) or all data gradients
 *     (bf16 = 1: one source, optional mask / add / fused upsum), plus -- between data gradients -- SSDN_OP_POOL_BWD ops;
 *   - 3x3 layers with identical taps and N, power-of-two images of at most 64 pixels (pooled size for SSDN_OP_POOL_BWD), kc = 48
 *     with c0, c1 multiples of 48, no dst32 / unrot;
 *   - a tensor read by an op is either written by an earlier op of the run (same p and cs, a channel window of what was written, same
 *     shape) or written by no op of the run; every output is still written to HBM.
 * The result is bit-identical to one launch per op.  ssdn_chain_len returns how many ops of the prefix of ops[0..n) run as one launch
 * (0 = none, else >= 2; at most 12; host code only, usable without a GPU); ssdn_conv_set_chain(0) switches the merging off (test
 * aid; also the folding of SSDN_OP_PACK_INPUT into the first layer's launch and of the narrow fp32-output 1x1 layer into the launch of
 * the 96-channel 1x1 layer in front of it), (1) on (default). */
int ssdn_chain_len(const ssdn_op* ops, int n);
int ssdn_conv_set_chain(int on);

/* ssdn_run_ops executes a run of consecutive SSDN_OP_WGRAD ops on one lane as ONE launch (k_wgrad_multi) when every op of the
 * run is "mergeable": at most 32768 pixels (the layers at the bottom of the U), mblocks <= 1, and a tiling the merged kernel
 * carries an instance for.  The result is bit-identical to one launch per op.  Returns 1 if `a` is mergeable, else 0. */
int ssdn_wgrad_mergeable(const ssdn_wgrad_args* a);

/* ---- step-level entry points: a whole configuration through libssdn_hip.so alone ------------------------------------------------
 * replaces: one iteration of the reference's training loop (ssdn/ssdn/train.py:196-202: run_pipeline -> mean(LOSS).backward() ->
 * optimizer.step()) and Denoiser.forward (ssdn/ssdn/denoiser.py:112-126), for ONE configuration and input shape.
 * A plan blob is written by the Python planner (DenoiserEngine.export_plan(), ssdn/hip/engine.py): the op lists of a step with every
 * pointer replaced by (tensor, offset), the tensor table, and a JSON description (configuration, shapes, the parameter layout of the
 * flat fp32 buffer: per layer name / w_off / b_off / M / cin / k, checkpoint order OIHW).  The library lays the tensors out in ONE
 * caller-owned device arena; the caller fills the named input tensors and reads the named outputs:
 *     "params" "grads" "adam_m" "adam_v"      flat fp32 buffers (a fresh arena must be ZERO: ssdn_plan_arena_bytes() bytes)
 *     "m/in32"        fp32 [B,C,H,W] network input (noisy image)          "noise_param"  fp32 [B] (ssdn, sigma known: std dev / lambda)
 *     "ref"           fp32 [B,C,H,W] reference image (mse / mask_mse)     "coords"       int64 [ncoords,2] (mask_mse)
 *     "loss"          fp32 [B]      "pme" "mu" fp32 [B,C,H,W] (ssdn: posterior mean / mu)     "m/out32" fp32 net_out
 * Phases: SSDN_PLAN_REPACK (fp16 / bf16 MFMA shadows from "params": after the caller has written parameters), SSDN_PLAN_FORWARD
 * (network + loss head), SSDN_PLAN_BACKWARD (gradients into "grads"), SSDN_PLAN_OPTIMISER (fused Adam + re-pack).  Everything is
 * enqueued asynchronously on `stream` (a hipStream_t; the library's side streams are ordered after it and joined into it).
 * Not thread-safe per plan; one plan per process / GPU.  All functions return 0 or a negative status with ssdn_last_error().
 * VALIDATED (tests/test_hip_plan_c.py, a process that imports nothing of this repository): one rank's shard of BASELINE configs 2 (ssdn,
 * sigma known), 3 (+ the sigma-estimation network: the blob concatenates the two networks' lists of a phase into ONE list, where the
 * Python path runs them as separate ssdn_run_ops calls on two streams) and 4 (Noise2Void: masked MSE at the exported coordinates) -- two
 * training steps each, loss and parameters bit-identical to the Python-driven steps.  ssdn_train_step fixes gscale = 1 and Adam's
 * betas 0.9 / 0.99 (the reference's, train.py:100-107) -- use ssdn_plan_set_lr + ssdn_plan_run for anything else. */
#define SSDN_PLAN_REPACK 0
#define SSDN_PLAN_FORWARD 1
#define SSDN_PLAN_BACKWARD 2
#define SSDN_PLAN_OPTIMISER 3
#define SSDN_PLAN_PHASES 4
typedef struct ssdn_plan ssdn_plan;
int ssdn_plan_load(const void* blob, int64_t nbytes, ssdn_plan** out);
void ssdn_plan_destroy(ssdn_plan* plan);
int64_t ssdn_plan_arena_bytes(const ssdn_plan* plan);      /* device bytes the caller provides (256-byte aligned, zero-filled) */
const char* ssdn_plan_meta(const ssdn_plan* plan);         /* the blob's JSON description */
int ssdn_plan_bind(ssdn_plan* plan, void* arena);
int ssdn_plan_tensor(const ssdn_plan* plan, const char* name, void** ptr, int64_t* bytes);
int ssdn_plan_run(ssdn_plan* plan, int phase, void* stream);
int ssdn_plan_set_lr(ssdn_plan* plan, float lr, int step, float gscale);   /* Adam: learning rate, step count (1-based: bias corrections), gradient scale */
int ssdn_net_forward(ssdn_plan* plan, void* stream);                       /* = phase SSDN_PLAN_FORWARD */
int ssdn_train_step(ssdn_plan* plan, float lr, int step, void* stream);    /* forward + loss, backward, Adam(lr, step) + re-pack */

/* Tuning aid (tools/conv_bench.py): device buffer that receives 32 s_memtime stamps per workgroup of the MFMA kernels, or
 * NULL (default) for none. */
void ssdn_debug_set_trace(void* device_buffer);
void* ssdn_debug_get_trace(void);

/* Hardware probes used by the test-suite (tests/test_hip_probe.py): raw lane mapping of
 * v_mfma_f32_32x32x16_f16 and ds_read_b64_tr_b16 on this device.  out: device buffers. */
int ssdn_probe_mfma(const void* a_frag, const void* b_frag, float* d_out, void* stream);
int ssdn_probe_tr16(const void* lds_image, int image_bytes, const int32_t* lane_addr, void* out, void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* SSDN_HIP_H */

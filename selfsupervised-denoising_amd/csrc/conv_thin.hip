// conv_thin.hip -- k_conv_thin: the forward 3x3 convolution of a layer whose input carries 1..3 REAL channels (encode_block_1.0: the
// image, in a 16-channel slot of zero-padded NHWC; noise_network.py:58-60,241-260) as an im2col-shaped GEMM
//     out[m][pixel] = lrelu(bias[m] + sum_n W[m][n] * patch[n][pixel]),   n = (window row, column, channel slot): THREE K-steps of 16
// instead of 9 taps x one 16-channel K-step of which 13 channels are zero padding.  The layer is 0.7 % of the forward flops and is bound
// by its 50 MB output; as a k_cdma launch it cost 28 us (halo tiles of 16 channel slots, 9 weight slices, 18 MFMAs per wave and tile).
//   workgroup = one 16-row x 64-column block of one image (4 waves x 8 column tiles of 32 pixels); the halo of the block's real channels
//   sits in LDS as [row][col][4 x fp16]; a B fragment is two aligned 8-byte reads per window row (K layout below), the A fragments
//   (weights gathered from the packed [tap][Mpad][16] tensor) and the bias (initial value of the accumulators, fp32) live in registers
//   for the whole block; per column tile: 6 LDS reads, 3 x MT MFMAs, LeakyReLU, a wave-private (double-buffered) LDS transpose and
//   16-byte pixel-contiguous stores.
// Contract: the input channels >= kreal are zero (SSDN_OP_PACK_INPUT writes them so; ssdn_conv_args.kreal).
#include "common.h"
#include <cstring>

#define CT_THREADS 256

// K layout (three K-steps of 16): K-step r = window row r; slot s = 4 * d + c: window column d (0..2; 3: zero), channel slot c (>= kreal:
// zero weights) -- the four channel slots of three adjacent pixels are 24 contiguous bytes of the halo, so a B fragment is two aligned
// ds_read_b64 per row (lanes kh = 0: columns 0 and 1; lanes kh = 1: column 2 and a zero slot) and nothing is gathered element-wise.
// PACK: the launch is ALSO the preceding SSDN_OP_PACK_INPUT (rotate-stack, NCHW fp32 -> NHWC fp16, zero-padded channel slots): the halo
// is converted straight from the fp32 images (1.5 MB, L2-resident) and every workgroup writes the packed rows of its own block -- the
// packed tensor is never read by this layer (it costs 64 bytes of HBM traffic per pixel for 6 real bytes) and one launch disappears.
template <int MT, bool PACK>
__global__ __launch_bounds__(CT_THREADS) void k_conv_thin(ssdn_conv_args a, int padT, int padB, int padL, int padR, unsigned long long tapmap, int dy0,
                                                          ssdn_pack_input_args pk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = a.W >> 6, bands = a.H >> 4;
    int bid = blockIdx.x;
    const int x0 = (bid % tiles_x) << 6; bid /= tiles_x;
    const int y0 = (bid % bands) << 4;
    const int n = bid / bands;
    const int HH = 16 + padT + padB, HW = 64 + padL + padR;
    // ---- weights and bias of this lane's rows: registers for the whole block ----
    const h16* wp = (const h16*)a.w;
    half8 af[MT][3];
    f32x16 binit[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = kh * 2 + (j >> 2), c = j & 3;
                const int t = (int)((tapmap >> (4 * (3 * r + (d < 3 ? d : 0)))) & 15u);
                af[mt][r][j] = (d < 3 && c < a.kreal) ? wp[((long long)t * a.Mpad + mt * 32 + l31) * a.Ktot + c] : (h16)0.f;
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mt * 32 + (r >> 2) * 8 + kh * 4 + (r & 3);
            binit[mt][r] = m < a.M ? a.bias[m] : 0.f;
        }
    }
    // ---- the halo of the block: the first four channel slots of every pixel, zero outside the image; then one 8-byte zero slot ----
    // (all loads of a thread's up to six halo entries are issued before the first is used: a guarded load per channel and pixel was a
    //  chain of fifteen exposed L2 round trips)
    const h16* src = (const h16*)a.src0.p + a.src0.co;
    constexpr int NIT = 6;                                     // (16 + 4) x (64 + 4) + 1 entries at most
    float f32v[NIT][3];
    u32x2_t raw[NIT];
    bool ok[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = tid + it * CT_THREADS;
        const int hy = e / HW, hx = e - hy * HW;
        const int y = y0 - padT + hy, x = x0 - padL + hx;
        ok[it] = e < HH * HW && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
        const int yc = ok[it] ? y : 0, xc = ok[it] ? x : 0;
        if constexpr (PACK) {
            const int r = n / pk.B, b = n - r * pk.B, H = a.H, W = a.W;
            int sy, sx;                                        // source coordinates in the un-rotated image (as k_pack_input)
            switch (r) {
                case 0: sy = yc; sx = xc; break;
                case 1: sy = xc; sx = W - 1 - yc; break;
                case 2: sy = H - 1 - yc; sx = W - 1 - xc; break;
                default: sy = H - 1 - xc; sx = yc; break;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) f32v[it][c] = pk.src[(((long long)b * pk.C + (c < pk.C ? c : 0)) * H + sy) * W + sx];
        } else {
            raw[it] = *reinterpret_cast<const u32x2_t*>(src + ((long long)(n * a.H + yc) * a.W + xc) * a.src0.cs);
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = tid + it * CT_THREADS;
        u32x2_t v = {0u, 0u};
        if (ok[it]) {
            if constexpr (PACK) {
                h16 c4[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) c4[c] = c < pk.C ? (h16)f32v[it][c] : (h16)0.f;
                v[0] = (unsigned)__builtin_bit_cast(unsigned short, c4[0]) | ((unsigned)__builtin_bit_cast(unsigned short, c4[1]) << 16);
                v[1] = (unsigned)__builtin_bit_cast(unsigned short, c4[2]);
            } else v = raw[it];
        }
        if (e < HH * HW + 1) *reinterpret_cast<u32x2_t*>(smem + e * 8) = v;
    }
    __syncthreads();
    if constexpr (PACK) {
        // the packed rows of this block's own pixels: cpad / 8 pieces of 16 bytes per pixel (piece 0 = the halo entry, the rest zeros),
        // consecutive lanes = consecutive pieces: every store instruction covers whole, contiguous pixels
        const int ppx = pk.cpad >> 3;
        for (int e = tid; e < 16 * 64 * ppx; e += CT_THREADS) {
            const int p = e / ppx, piece = e - p * ppx;
            const int ty = p >> 6, tx = p & 63;
            u32x4_t o = {0u, 0u, 0u, 0u};
            if (piece == 0) {
                const u32x2_t v = *reinterpret_cast<const u32x2_t*>(smem + (((ty + padT) * HW + tx + padL) << 3));
                o[0] = v[0]; o[1] = v[1];
            }
            *reinterpret_cast<u32x4_t*>((h16*)pk.dst.p + ((long long)(n * a.H + y0 + ty) * a.W + x0 + tx) * pk.dst.cs + pk.dst.co + piece * 8) = o;
        }
    }
    constexpr int OSTR = MT * 64 + 16;
    char* reg = smem + (((HH * HW + 1) * 8 + 15) & ~15) + wave * (32 * OSTR);
    const int npc = a.M >> 3;
    const unsigned npc_magic = npc <= 1 ? 0u : (unsigned)((0x100000000ull + npc - 1) / npc);      // (npc == 1: 2^32 does not fit -- the quotient is e)
    h16* dst = (h16*)a.dst.p + a.dst.co;
    const int zoff = HH * HW * 8;
#pragma unroll 2
    for (int i = 0; i < 8; ++i) {
        const int q = (wave * 8 + i) * 32 + l31;
        const int ty = q >> 6, tx = q & 63;
        // window origin of this pixel: row y + dy0, column x - 1
        const int pb = (((ty + padT + dy0) * HW + tx + padL - 1) << 3) + kh * 16;
        half8 bf[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(smem + pb + r * HW * 8);
            const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(smem + (kh ? zoff : pb + r * HW * 8 + 8));
            const u32x4_t w4 = {lo[0], lo[1], hi[0], hi[1]};
            bf[r] = __builtin_bit_cast(half8, w4);
        }
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[mt] = binit[mt];
#pragma unroll
            for (int r = 0; r < 3; ++r) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt][r], bf[r], acc[mt], 0, 0, 0);
        }
        // registers -> wave-private LDS [pixel][channel] -> 16-byte pieces of consecutive pixels
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[mt][g * 4 + j];
#pragma unroll
                for (int j = 0; j < 4; j += 2) {      // LeakyReLU = max(v, slope v): v_pk_mul_f32 + a raw v_max_f32 (round 5; was compare + multiply + select per value)
                    const f32x2_t t = f32x2_t{v[j], v[j + 1]} * LRELU_SLOPE;
                    asm("v_max_f32 %0, %1, %2" : "=v"(v[j]) : "v"(v[j]), "v"(t[0]));
                    asm("v_max_f32 %0, %1, %2" : "=v"(v[j + 1]) : "v"(v[j + 1]), "v"(t[1]));
                }
                u32x2_t o;
                o[0] = pack_f16x2(v[0], v[1]);
                o[1] = pack_f16x2(v[2], v[3]);
                *reinterpret_cast<u32x2_t*>(reg + (i & 1) * (4 * 32 * OSTR) + l31 * OSTR + (mt * 32 + g * 8 + kh * 4) * 2) = o;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const long long pix0 = (long long)(n * a.H + y0 + (((wave * 8 + i) * 32) >> 6)) * a.W + x0 + (((wave * 8 + i) * 32) & 63);
        for (int e = lane; e < 32 * npc; e += 64) {
            const int p = npc_magic ? (int)__umulhi((unsigned)e, npc_magic) : e, cc = e - p * npc;      // e / npc (exact for e < 2^16: magic = ceil(2^32 / npc))
            const u32x4_t o = *reinterpret_cast<const u32x4_t*>(reg + (i & 1) * (4 * 32 * OSTR) + p * OSTR + cc * 16);
            *reinterpret_cast<u32x4_t*>(dst + (pix0 + p) * a.dst.cs + cc * 8) = o;
            if (a.sign_out) {      // LeakyReLU sign byte of the piece (ssdn_conv_args.sign_out): bit q = (channel q > 0), on the raw fp16 halves
                // min(max(h, 0), 1) per half on the raw 16-bit patterns, the eight 0/1 halves merged by shifts (k_cdma's form: 13 instructions)
                unsigned rq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned t0;
                    asm("v_pk_max_i16 %0, %1, 0" : "=v"(t0) : "v"(o[q]));
                    asm("v_pk_min_i16 %0, %1, %2" : "=v"(rq[q]) : "v"(t0), "s"(0x00010001u));
                }
                const unsigned t01 = (rq[1] << 2) | rq[0], t23 = (rq[3] << 2) | rq[2];
                const unsigned t = (t23 << 4) | t01;              // bits 0,2,4,6: low halves; 16,18,20,22: high halves
                const unsigned sb = t | (t >> 15);
                ((unsigned char*)a.sign_out)[(pix0 + p) * npc + cc] = (unsigned char)sb;
            }
        }
    }
}

static void thin_pads(const ssdn_conv_args* a, int* padT, int* padB, int* padL, int* padR) {
    int mny = 0, mxy = 0, mnx = 0, mxx = 0;
    for (int t = 0; t < a->ntaps; ++t) {
        mny = a->dy[t] < mny ? a->dy[t] : mny; mxy = a->dy[t] > mxy ? a->dy[t] : mxy;
        mnx = a->dx[t] < mnx ? a->dx[t] : mnx; mxx = a->dx[t] > mxx ? a->dx[t] : mxx;
    }
    *padT = -mny; *padB = mxy; *padL = -mnx; *padR = mxx;
}

// the 3x3 window behind the nine taps: tap index of (row r, column d), rows = three consecutive dy starting at *dy0, columns dx = -1, 0, 1
static bool thin_window(const ssdn_conv_args* a, unsigned long long* tapmap, int* dy0) {
    int mny = 1 << 30;
    for (int t = 0; t < 9; ++t) mny = a->dy[t] < mny ? a->dy[t] : mny;
    unsigned long long map = 0;
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
        const int r = a->dy[t] - mny, d = a->dx[t] + 1;
        if (r < 0 || r > 2 || d < 0 || d > 2) return false;
        map |= (unsigned long long)t << (4 * (3 * r + d));
        seen |= 1u << (3 * r + d);
    }
    if (seen != 0x1ffu) return false;
    *tapmap = map; *dy0 = mny;
    return true;
}

bool conv_thin_eligible(const ssdn_conv_args* a) {
    if (a->bf16 || a->ntaps != 9 || a->c1 || a->up0 || a->Ktot != 16 || a->c0 != 16 || a->kreal < 1 || a->kreal > 3) return false;
    if (!a->dst.p || a->dst32 || !a->act || !a->bias || a->pool.p || a->mask.p || a->add.p || a->upsum.p || a->unrot.p) return false;
    if ((a->H & 15) || (a->W & 63) || a->M < 8 || (a->M & 7) || a->Mpad > 64 || a->Mpad < 32) return false;
    if (a->src0.cs < 4 || (a->src0.cs & 3) || (a->src0.co & 3)) return false;      // 8-byte pixel heads
    unsigned long long map; int dy0;
    int pt, pb, pl, pr;
    thin_pads(a, &pt, &pb, &pl, &pr);
    if ((16 + pt + pb) * (64 + pl + pr) + 1 > 6 * CT_THREADS) return false;
    return thin_window(a, &map, &dy0);
}

// the SSDN_OP_PACK_INPUT that writes this layer's input can be folded into the launch
bool conv_thin_fuses_pack(const ssdn_pack_input_args* pk, const ssdn_conv_args* a) {
    if (!conv_thin_eligible(a) || !pk->src || !pk->dst.p) return false;
    if (pk->dst.p != a->src0.p || pk->dst.cs != a->src0.cs || pk->dst.co != a->src0.co) return false;
    if (pk->R * pk->B != a->N || pk->H != a->H || pk->W != a->W || pk->C != a->kreal || pk->C > 3) return false;
    if ((pk->R != 1 && pk->R != 4) || (pk->R == 4 && pk->H != pk->W)) return false;
    return (pk->cpad & 7) == 0 && pk->cpad >= 8 && (pk->dst.cs & 7) == 0 && (pk->dst.co & 7) == 0;
}

int launch_conv_thin(const ssdn_conv_args* a, const ssdn_pack_input_args* pk, hipStream_t s) {
    int pt, pb, pl, pr, dy0 = 0;
    unsigned long long map = 0;
    thin_pads(a, &pt, &pb, &pl, &pr);
    if (!thin_window(a, &map, &dy0)) return ssdn_set_error("conv thin: the taps are not a 3x3 window");
    if (pk && !conv_thin_fuses_pack(pk, a)) return ssdn_set_error("conv thin: this SSDN_OP_PACK_INPUT cannot be folded into the launch");
    const int mt = a->Mpad / 32;
    const size_t lds = (size_t)((((16 + pt + pb) * (64 + pl + pr) + 1) * 8 + 15) & ~15) + 2u * 4u * 32u * (mt * 64 + 16);
    const int grid = a->N * (a->H >> 4) * (a->W >> 6);
    ssdn_pack_input_args none;
    memset(&none, 0, sizeof(none));
    if (pk) {
        if (mt == 2) hipLaunchKernelGGL((k_conv_thin<2, true>), dim3(grid), dim3(CT_THREADS), lds, s, *a, pt, pb, pl, pr, map, dy0, *pk);
        else hipLaunchKernelGGL((k_conv_thin<1, true>), dim3(grid), dim3(CT_THREADS), lds, s, *a, pt, pb, pl, pr, map, dy0, *pk);
    } else {
        if (mt == 2) hipLaunchKernelGGL((k_conv_thin<2, false>), dim3(grid), dim3(CT_THREADS), lds, s, *a, pt, pb, pl, pr, map, dy0, none);
        else hipLaunchKernelGGL((k_conv_thin<1, false>), dim3(grid), dim3(CT_THREADS), lds, s, *a, pt, pb, pl, pr, map, dy0, none);
    }
    SSDN_CHECK_HIP(hipGetLastError());
    return 0;
}

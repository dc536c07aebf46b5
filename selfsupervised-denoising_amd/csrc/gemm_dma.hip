// SSDN_OP_CONV, 1x1 layers (the posterior head's NiN convolutions, noise_network.py:116-130 of the reference, and their data
// gradients): out[p][m] = epi( sum_k in[p][k] * w[m][k] + bias[m] ) over P = N*H*W pixels -- a plain GEMM whose roofline is
// HBM (every input pixel read once, every output written once: output_block.0 at batch 32 moves 2 x 100 MB for 38.6 GFLOP), so
// the design goal is ONE pass over the input:
//
//   * a workgroup owns 256 consecutive pixels and ALL output channels of the layer (384: tile 256 x 384, eight waves of
//     64 px x 192 ch = 12 accumulator tiles of 32x32; 96: eight waves of 32 px x 96 ch), so the activation tensor is fetched
//     exactly once; the weight matrix (<= 288 KiB) streams from L2.
//   * operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), 32 input channels per chunk: rows of 64 bytes, three
//     stages in flight, one barrier per chunk.  A DMA instruction deposits 64 lanes x 16 B linearly, but every lane fetches from
//     its own global address -- the 16-byte piece p of row r is stored at piece p ^ ((r >> 2) & 3), which makes every
//     ds_read_b128 fragment read (32 rows x one K half) conflict-free (16-lane groups hit 16 distinct 16-byte slots).
//   * per chunk a wave issues 2 x (WP + WM) ds_read_b128 and 2 x WP x WM MFMAs (32x32x16): 16 reads / 24 MFMAs for the 384 tile.
//   * epilogue as in k_cdma: accumulators (initialised with the bias) -> LeakyReLU -> 16-bit, widened to 16-byte pieces with
//     v_permlane32_swap, transposed through a wave-private LDS region, stored as pixel-contiguous runs; the data-gradient role
//     multiplies by LeakyReLU'(saved activation) on the way out.
#include "common.h"

namespace {

struct GdAux {
    int nch;      // Ktot / 32
    int ntiles;   // pixel tiles of TP
    int lp;       // fused UNROT_BWD: log2 of the image side
    // EPI bit 2: the launch is ALSO the following narrow 1x1 layer on its own output (the 9-channel net_out layer behind the 96-channel
    // one): out32[n][m][pixel] = [lrelu](bias4[m] + sum_k w4[m][k] * out(pixel, k)), fp32 NCHW, from the rounded 16-bit tile in LDS
    const h16* w4;      // packed [1][32][96]
    const float* b4;    // or NULL
    float* out32;
    int M4, act4, HW;
    int ablate;   // tuning aid (env SSDN_GDMA_ABLATE, `make TUNING=1` builds only): 1 no epilogue, 2 the epilogue's stores are dropped by a zero-size
                  // buffer resource, 8 the epilogue stops behind the registers -> LDS half
};
#ifdef SSDN_TUNING
#define GD_ABL(xx, bit) (((xx).ablate & (bit)) != 0)
#else
#define GD_ABL(xx, bit) false
#endif

// LDS-DMA through the compiler's builtin (round 5): it sets M0 and pads SGPR hazards only where needed; as inline asm every piece carried
// an `s_nop 4` (tools/probes/probe_dmacost.hip: 12-18 ns per piece for an MFMA-issuing wave; a loader wave here issues 4-6 pieces per chunk).
// Exactly one VMEM instruction per call: the vmcnt group accounting below counts them.
typedef __attribute__((address_space(3))) void* gd_lds_ptr;
__device__ __forceinline__ void gd_dma16(unsigned lds_addr, int voff, __amdgpu_buffer_rsrc_t rs, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (gd_lds_ptr)(size_t)__builtin_amdgcn_readfirstlane(lds_addr), 16, voff, __builtin_amdgcn_readfirstlane(soff), 0, 0);
}

template <bool BF>
__device__ __forceinline__ f32x16 gd_mma(half8 av, half8 bv, f32x16 c) {
    if constexpr (BF)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
}

#ifndef GD_WARMUP
#define GD_WARMUP 1
#endif
#ifndef GD_DA_V
#define GD_DA_V 3
#define GD_DB_V 3
#endif
constexpr int GD_DA = GD_DA_V, GD_DB = GD_DB_V;
constexpr int GD_LUT_BYTES = 256;       // 16 x float4 behind the scratch KiB: sign nibble -> LeakyReLU' factors
constexpr int gd_eg(int wm) { return wm == 6 ? 2 : wm; }       // channel tiles per epilogue transpose group

// wait until all but the newest n groups of PER DMA instructions of this wave have landed
template <int PER>
__device__ __forceinline__ void gd_wait_groups(int n) {
    static_assert(3 * PER < 64, "vmcnt immediate");
    if (n <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");
}

}  // namespace

// wave grid NWP (pixels) x NWM (channels); a wave owns WP x 32 pixels and WM x 32 output channels.  EPI bit 0: LeakyReLU' mask;
// bit 2: the next narrow 1x1 layer computed from the output tile (GdAux.w4);
// bit 1: fused SSDN_OP_UNROT_BWD (the 96-channel block r of a pixel goes, times LeakyReLU', to its place in rotation r's tensor);
// bit 3: the LeakyReLU' operand of bit 0 / bit 1 arrives as sign bytes (mask_sign / unrot_smask); bit 4: sign bytes of the output (sign_out).
// PERSISTENT: a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...; the two operand rings run on as one chunk stream
// across tile boundaries, so the loads of tile i+1 are in flight while tile i is converted and stored (the epilogue has its own
// LDS region: wave-private transposes of EG channel tiles at a time).
template <int WP, int WM, int NWP, int NWM, bool BF, int EPI>
__global__ __launch_bounds__(64 * NWP * NWM, 1) void k_gdma(ssdn_conv_args a, GdAux x) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = NWP * NWM, TP = NWP * WP * 32, TM = NWM * WM * 32;
    constexpr int ABYTES = TP * 64, BBYTES = TM * 64;
    constexpr int DA = GD_DA, DB = GD_DB;                    // ring depths: activation chunks (HBM) / weight chunks (L2)
    constexpr int NLA = NW / 2, NLB = NW - NLA;              // loader roles: waves [0, NLA) fetch activations, the rest weights --
                                                             // vmcnt completes in order per WAVE: two streams, two counters
    constexpr int NIA = TP / 16, NIB = TM / 16, PA = (NIA + NLA - 1) / NLA, PB = (NIB + NLB - 1) / NLB;
    constexpr int EG = gd_eg(WM), NEG = WM / EG;             // epilogue: channel tiles per transpose group
    constexpr int OSTR = EG * 64 + 16, NEK = EG * 2, CPP = EG * 4;
    constexpr int BOFF = DA * ABYTES, EOFF = BOFF + DB * BBYTES, BIAS_OFF = EOFF + NW * 32 * OSTR, DUMMY_OFF = BIAS_OFF + TM * 4;
    constexpr int LUT_OFF = DUMMY_OFF + 1024;                // 16 x float4: the LeakyReLU' factors of a sign nibble (GD_LUT_BYTES)
    constexpr bool HAS_MASK = (EPI & 1) != 0, UNROT = (EPI & 2) != 0, OUT4 = (EPI & 4) != 0, SMASK = (EPI & 8) != 0, SOUT = (EPI & 16) != 0;
    static_assert(!OUT4 || (NWM == 1 && WM == 3 && gd_eg(WM) == 3 && !BF), "the fused narrow layer needs the wave's whole 96-channel tile in its LDS region");
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = w / NWM, wm = w - wp * NWM;
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int G = gridDim.x;
    const int ntl = (x.ntiles - (int)blockIdx.x + G - 1) / G;        // tiles of this workgroup
    const int total = ntl * x.nch;                                   // its chunk stream

    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(a.src0.p, 0, (int)0x80000000, SSDN_BUFFER_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (int)0x80000000, SSDN_BUFFER_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc(a.dst.p, 0, GD_ABL(x, 2) ? 0 : (int)0x80000000, SSDN_BUFFER_RSRC_FLAGS);
#if GD_WARMUP
    // L2 warm-up (as k_cdma's): the weight matrix streams from L2 chunk by chunk, in the same order in every workgroup -- and a launch
    // finds it in no L2.  One 128-byte line per thread, the workgroups of an XCD (blockIdx % 8) share the matrix
    const unsigned pf = __builtin_amdgcn_raw_buffer_load_b32(
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.Mpad * a.Ktot * 2, SSDN_BUFFER_RSRC_FLAGS),
        (int)((blockIdx.x >> 3) * (64 * NW) + tid) << 7, 0, 0);
#endif
    const __amdgpu_buffer_rsrc_t rs_mask = __builtin_amdgcn_make_buffer_rsrc(a.mask.p, 0, (EPI & 1) ? (int)0x80000000 : 0, SSDN_BUFFER_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_ur = __builtin_amdgcn_make_buffer_rsrc(a.unrot.p, 0, UNROT && !GD_ABL(x, 2) ? (int)0x80000000 : 0, SSDN_BUFFER_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_um = __builtin_amdgcn_make_buffer_rsrc(a.unrot_mask.p, 0, UNROT ? (int)0x80000000 : 0, SSDN_BUFFER_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_us = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(UNROT ? a.unrot_smask : a.mask_sign), 0, SMASK ? (int)0x80000000 : 0, SSDN_BUFFER_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_so = __builtin_amdgcn_make_buffer_rsrc(a.sign_out, 0, SOUT && !GD_ABL(x, 2) ? (int)0x80000000 : 0, SSDN_BUFFER_RSRC_FLAGS);

    // DMA lane constants: lane -> (row lane >> 2 of a 16-row instruction, LDS piece lane & 3); the piece fetched is the swizzled one
    const int drow = lane >> 2, dpiece = (lane & 3) ^ ((drow >> 2) & 3);
    const int voffA = (drow * a.src0.cs + dpiece * 8) * 2;
    const int voffB = (drow * a.Ktot + dpiece * 8) * 2;
    const bool loadA = w < NLA;
    // loader state: the next chunk to issue is chunk lc of this workgroup's tile lt, into stage lst
    int lt = 0, lc = 0, lst = 0, issued = 0;
    auto issue_next = [&]() __attribute__((always_inline)) {
        if (loadA) {
            const int pix0 = ((int)blockIdx.x + lt * G) * TP;
#pragma unroll
            for (int u = 0; u < PA; ++u) {
                const int i = w + NLA * u;              // wave-uniform
                if (i < NIA) {
                    const int soff = __builtin_amdgcn_readfirstlane((((pix0 + i * 16) * a.src0.cs) + a.src0.co + lc * 32) * 2);
                    gd_dma16(lds0 + lst * ABYTES + i * 1024, voffA, rs_a, soff);
                } else {
                    // keep every loader's DMA count per chunk equal (one vmcnt immediate per role): an out-of-range fetch that
                    // drops zeros into a scratch KiB
                    gd_dma16(lds0 + DUMMY_OFF, (int)0x80000000, rs_w, 0);
                }
            }
            lst = lst == DA - 1 ? 0 : lst + 1;
        } else {
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int j = (w - NLA) + NLB * u;
                if (j < NIB) {
                    const int soff = __builtin_amdgcn_readfirstlane((j * 16 * a.Ktot + lc * 32) * 2);
                    gd_dma16(lds0 + BOFF + lst * BBYTES + j * 1024, voffB, rs_w, soff);
                } else {
                    gd_dma16(lds0 + DUMMY_OFF, (int)0x80000000, rs_w, 0);
                }
            }
            lst = lst == DB - 1 ? 0 : lst + 1;
        }
        ++issued;
        if (++lc == x.nch) { lc = 0; ++lt; }
    };
    const int lead = loadA ? DA - 1 : DB - 1;
    for (int c = 0; c < lead && c < total; ++c) issue_next();
    float* bl = reinterpret_cast<float*>(smem + BIAS_OFF);
    if (tid < TM) bl[tid] = (a.bias && tid < a.M) ? a.bias[tid] : 0.f;
    if constexpr (SMASK) {
        // LeakyReLU' of four channels from their four sign bits: entry n = {bit j of n ? 1 : slope}, one ds_read_b128 per nibble instead of
        // a bit test, a select and a scalar multiply per channel (the epilogue is VALU-bound: profiles/r05_gdma_epilogue.txt)
        if (tid < 64) reinterpret_cast<float*>(smem + LUT_OFF)[tid] = ((tid >> 2) >> (tid & 3)) & 1 ? 1.f : LRELU_SLOPE;
    }
    half8 w4f[6];                         // OUT4: the narrow layer's weights, rows l31, K = 96 in six K-steps: registers for the whole launch
    if constexpr (OUT4) {
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) w4f[ks] = *reinterpret_cast<const half8*>(x.w4 + l31 * 96 + ks * 16 + (lane >> 5) * 8);
    }
    __syncthreads();

    // fragment addresses: row l31 of a 32-row block, K half kh of K-step s -> piece (2s + kh) ^ ((l31 >> 2) & 3)
    const int sw = (l31 >> 2) & 3;
    const int fr0 = l31 * 64 + ((kh ^ sw) << 4), fr1 = l31 * 64 + (((2 + kh) ^ sw) << 4);
    const int pbase = wp * WP * 2048, mbase = BOFF + wm * WM * 2048;
    const float slope = a.act ? LRELU_SLOPE : 1.f;
    char* reg = smem + EOFF + w * (32 * OSTR);

    int ca = 0, cb = 0, done = 0;       // stages of the chunk being consumed; chunks consumed so far
    for (int ti = 0; ti < ntl; ++ti) {
        const int pix0 = ((int)blockIdx.x + ti * G) * TP;
        f32x16 acc[WM][WP];
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = *reinterpret_cast<const float4*>(bl + (wm * WM + mt) * 32 + 8 * g + 4 * kh);
#pragma unroll
                for (int pt = 0; pt < WP; ++pt) {
                    acc[mt][pt][4 * g + 0] = b4.x; acc[mt][pt][4 * g + 1] = b4.y; acc[mt][pt][4 * g + 2] = b4.z; acc[mt][pt][4 * g + 3] = b4.w;
                }
            }
        // ---- chunks of this tile: chunk g sits in stages g % DA / g % DB; the loaders keep D-1 chunks ahead of `done` ----
        for (int c = 0; c < x.nch; ++c) {
            const int left = total - 1 - done;
            if (loadA) gd_wait_groups<PA>(left < DA - 2 ? left : DA - 2);
            else gd_wait_groups<PB>(left < DB - 2 ? left : DB - 2);
            __syncthreads();
            if (issued < total) issue_next();
            const char* pa = smem + ca * ABYTES + pbase;
            const char* pb = smem + cb * BBYTES + mbase;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int fr = s2 ? fr1 : fr0;
                half8 pq[WP], wq[WM];
#pragma unroll
                for (int pt = 0; pt < WP; ++pt) pq[pt] = *reinterpret_cast<const half8*>(pa + pt * 2048 + fr);
#pragma unroll
                for (int mt = 0; mt < WM; ++mt) wq[mt] = *reinterpret_cast<const half8*>(pb + mt * 2048 + fr);
#pragma unroll
                for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                    for (int pt = 0; pt < WP; ++pt) acc[mt][pt] = gd_mma<BF>(wq[mt], pq[pt], acc[mt][pt]);
            }
            ca = ca == DA - 1 ? 0 : ca + 1;
            cb = cb == DB - 1 ? 0 : cb + 1;
            ++done;
        }
        // ---- epilogue of the tile (wave-private; the rings keep filling meanwhile) ----
        if (GD_ABL(x, 1)) continue;
#pragma unroll
        for (int pt = 0; pt < WP; ++pt) {
            const int pix_p = pix0 + (wp * WP + pt) * 32;
#pragma unroll
            for (int eg = 0; eg < NEG; ++eg) {
                const int chb = wm * WM * 32 + eg * EG * 32;       // first channel of the group
#pragma unroll
                for (int me = 0; me < EG; ++me)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        const int mt = eg * EG + me;
                        unsigned pk[2][2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            float v[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] = acc[mt][pt][(2 * gp + h) * 4 + q];
                            if constexpr (!BF) {      // LeakyReLU (slope 1: identity); the data-gradient role (bf16) has none -- gemm_dma_eligible
#pragma unroll
                                for (int q = 0; q < 4; q += 2) {
                                    const f32x2_t t = f32x2_t{v[q], v[q + 1]} * slope;          // v_pk_mul_f32
                                    asm("v_max_f32 %0, %1, %2" : "=v"(v[q]) : "v"(v[q]), "v"(t[0]));           // (fmaxf: + a canonicalising v_max per value)
                                    asm("v_max_f32 %0, %1, %2" : "=v"(v[q + 1]) : "v"(v[q + 1]), "v"(t[1]));
                                }
                            }
                            pk[h][0] = BF ? pack_bf16x2(v[0], v[1]) : pack_f16x2(v[0], v[1]);
                            pk[h][1] = BF ? pack_bf16x2(v[2], v[3]) : pack_f16x2(v[2], v[3]);
                        }
                        u32x4_t o;
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            auto r = __builtin_amdgcn_permlane32_swap(pk[0][d], pk[1][d], false, false);
                            o[d] = r[0]; o[2 + d] = r[1];
                        }
                        const int piece = me * 4 + 2 * gp + kh;
                        *reinterpret_cast<u32x4_t*>(reg + l31 * OSTR + piece * 16) = o;
                    }
                if constexpr (OUT4) {
                    // the narrow layer on the rounded tile just parked in LDS: D[m][pixel] = sum_k w4[m][k] * tile[pixel][k], k ascending on one
                    // accumulator (the order of the separate launch: bit-identical), + bias, fp32 NCHW stores coalesced along the pixels
                    f32x16 a4;
#pragma unroll
                    for (int r = 0; r < 16; ++r) a4[r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 6; ++ks) {
                        const half8 bq = *reinterpret_cast<const half8*>(reg + l31 * OSTR + (ks * 16 + kh * 8) * 2);
                        a4 = gd_mma<false>(w4f[ks], bq, a4);
                    }
                    const int pix = pix_p + l31;
                    const int n4 = pix / x.HW, rem = pix - n4 * x.HW;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = 8 * (r >> 2) + 4 * kh + (r & 3);
                        if (m >= x.M4) continue;
                        float v = a4[r];
                        if (x.b4) v += x.b4[m];
                        if (x.act4) v = lrelu(v);
                        x.out32[((long long)n4 * x.M4 + m) * x.HW + rem] = v;
                    }
                }
                if (GD_ABL(x, 8)) continue;      // tuning: conversion + LDS writes only
                // LDS -> HBM: 32 pixels x CPP 16-byte pieces, pixel-contiguous runs of EG * 64 bytes
                u32x4_t mb[NEK];
                int goff[NEK], loff[NEK];
                bool live[NEK];
                // fused un-rotation: the destination pixel of (b, iy, jx) in rotation r's tensor is AFFINE in (iy, jx) --
                //   r=0: (u, v) = (iy, jx)   r=1: (P-1-jx, iy)   r=2: (P-1-iy, P-1-jx)   r=3: (jx, P-1-iy);   row u-1 (u == 0: the shift cut the
                //   pixel off, zeros go to row P-1 = "row -1 + P") -- so a piece costs two multiply-adds on coefficients picked ONCE per group
                //   from the lane's channel block instead of two four-way selects per piece (round 5: the epilogue is VALU-bound)
                [[maybe_unused]] int ur_cI = 0, ur_cJ = 0, ur_c0 = 0, ur_uI = 0, ur_uJ = 0, ur_u0 = 0, ur_cc = 0;
                if constexpr (UNROT) {
                    static_assert(!UNROT || CPP == 8, "lane -> (pixel lane >> 3, piece lane & 7)");
                    int ln = lane;
                    asm volatile("" : "+v"(ln));          // opaque: per-group values are recomputed here, not hoisted over the K loop (192 accumulators live)
                    const int P = 1 << x.lp, ch = chb + ((ln & 7) << 3);
                    const int r = (ch >= 96 ? 1 : 0) + (ch >= 192 ? 1 : 0) + (ch >= 288 ? 1 : 0);
                    ur_cc = ch - r * 96;
                    ur_cI = r == 0 ? P : (r == 1 ? 1 : (r == 2 ? -P : -1));
                    ur_cJ = r == 0 ? 1 : (r == 1 ? -P : (r == 2 ? -1 : P));
                    ur_c0 = (r == 0 ? -P : (r == 1 ? (P - 2) * P : (r == 2 ? (P - 2) * P + P - 1 : -1))) + ((r * a.N) << (2 * x.lp));
                    ur_uI = r == 0 ? 1 : (r == 2 ? -1 : 0);
                    ur_uJ = r == 3 ? 1 : (r == 1 ? -1 : 0);
                    ur_u0 = (r == 1 || r == 2) ? P - 1 : 0;
                }
#pragma unroll
                for (int k = 0; k < NEK; ++k) {
                    const int p = k * 64 + lane;
                    const int px = p / CPP, c16 = (p - px * CPP) << 4;
                    const int pix = pix_p + px;
                    loff[k] = px * OSTR + c16;
                    live[k] = true;
                    if constexpr (UNROT) {
                        const int lp = x.lp, P = 1 << lp;
                        const int jx = pix & (P - 1), iy = (pix >> lp) & (P - 1), b = pix >> (2 * lp);
                        const int u = __mul24(ur_uI, iy) + __mul24(ur_uJ, jx) + ur_u0;
                        live[k] = u >= 1;
                        const int dpix = __mul24(ur_cI, iy) + __mul24(ur_cJ, jx) + ur_c0 + (b << (2 * lp)) + (live[k] ? 0 : P << lp);
                        goff[k] = (int)__umul24(dpix, a.unrot.cs * 2) + (a.unrot.co + ur_cc) * 2;      // (dpix < 2^24: gemm_dma_eligible bounds 4 N H W cs)
                        if constexpr (SMASK)      // one sign byte per 16-byte piece (written by SSDN_OP_UNROT_FWD), 12 bytes per pixel
                            mb[k][0] = __builtin_amdgcn_raw_buffer_load_b8(rs_us, live[k] ? (int)__umul24(dpix, 12) + (ur_cc >> 3) : (int)0x80000000, 0, 0);
                        else
                            mb[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_um, live[k] ? (int)__umul24(dpix, a.unrot_mask.cs * 2) + (a.unrot_mask.co + ur_cc) * 2 : (int)0x80000000, 0, 0);
                    } else {
                        goff[k] = (pix * a.dst.cs + a.dst.co + chb) * 2 + c16;
                        if constexpr (HAS_MASK && SMASK) mb[k][0] = __builtin_amdgcn_raw_buffer_load_b8(rs_us, pix * (a.M >> 3) + (chb >> 3) + (c16 >> 4), 0, 0);
                        else if constexpr (HAS_MASK) mb[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_mask, (pix * a.mask.cs + a.mask.co + chb) * 2 + c16, 0, 0);
                    }
                }
#pragma unroll
                for (int k = 0; k < NEK; ++k) {
                    u32x4_t o = *reinterpret_cast<const u32x4_t*>(reg + loff[k]);
                    if constexpr (UNROT) {
                        if (!live[k]) o = u32x4_t{0u, 0u, 0u, 0u};
                    }
                    if constexpr ((HAS_MASK || UNROT) && SMASK && BF) {
                        // sign byte: bit 2q = low half of dword q, bit 2q+1 = its high half -> nibble 0 = dwords 0-1, nibble 1 = dwords 2-3
                        const unsigned sb = mb[k][0];
                        const f32x4 m0 = *reinterpret_cast<const f32x4*>(smem + LUT_OFF + ((sb & 15u) << 4));
                        const f32x4 m1 = *reinterpret_cast<const f32x4*>(smem + LUT_OFF + ((sb >> 4) << 4));      // (a zero-extended byte)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x2_t v = f32x2_t{bf_lo(o[q]), bf_hi(o[q])} *
                                              (q < 2 ? f32x2_t{m0[2 * q], m0[2 * q + 1]} : f32x2_t{m1[2 * q - 4], m1[2 * q - 3]});
                            o[q] = pack_bf16x2(v[0], v[1]);
                        }
                    } else if constexpr (HAS_MASK || UNROT) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float v0, v1;
                            if constexpr (BF) { v0 = bf_lo(o[q]); v1 = bf_hi(o[q]); }
                            else { v0 = f16_lo(o[q]); v1 = f16_hi(o[q]); }
                            bool p0, p1;
                            if constexpr (SMASK) { p0 = (mb[k][0] >> (2 * q)) & 1u; p1 = (mb[k][0] >> (2 * q + 1)) & 1u; }
                            else { p0 = (int)(short)(mb[k][q] & 0xffffu) > 0; p1 = ((int)mb[k][q] >> 16) > 0; }
                            v0 *= p0 ? 1.f : LRELU_SLOPE;
                            v1 *= p1 ? 1.f : LRELU_SLOPE;
                            o[q] = BF ? pack_bf16x2(v0, v1) : pack_f16x2(v0, v1);
                        }
                    }
                    if constexpr (UNROT) __builtin_amdgcn_raw_buffer_store_b128(o, rs_ur, goff[k], 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(o, rs_dst, goff[k], 0, 0);
                    if constexpr (SOUT) {
                        // sign byte of the piece: bit 2q = (low half of dword q > 0), bit 2q+1 = (high half > 0), on the raw 16-bit patterns:
                        // min(max(h, 0), 1) per half, then the eight 0/1 halves merged by shifts (k_cdma's form: 13 instructions, not ~21)
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
                        const int p = k * 64 + lane, px = p / CPP, c16 = (p - px * CPP) << 4;
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)sb, rs_so, (pix_p + px) * (a.M >> 3) + (chb >> 3) + (c16 >> 4), 0, 0);
                    }
                }
            }
        }
    }
#if GD_WARMUP
    asm volatile("" :: "v"(pf));
#endif
}

// ---- host ----------------------------------------------------------------------------------------------------------------------
bool gemm_dma_eligible(const ssdn_conv_args* a) {
    if (a->ntaps != 1 || a->dy[0] || a->dx[0] || a->up0 || a->c1 || a->src1.p || a->dst32 || a->add.p) return false;
    if (a->pool.p || a->upsum.p) return false;
    if (a->unrot.p && (!a->bf16 || a->mask.p || a->Mpad != 384 || a->H != a->W || (a->H & (a->H - 1)) ||
                       (long long)4 * a->N * a->H * a->W * a->unrot.cs * 2 >= (1ll << 31))) return false;
    if (a->c0 != a->Ktot || a->Ktot % 32) return false;
    if (a->M != a->Mpad || (a->Mpad != 384 && a->Mpad != 96)) return false;
    const long long px = (long long)a->N * a->H * a->W;
    if (px % 256) return false;
    if (!a->bf16 && a->mask.p) return false;
    if (a->bf16 && a->act) return false;      // the data-gradient role's epilogue has no LeakyReLU (compile-time)
    int csmax = a->dst.cs > a->src0.cs ? a->dst.cs : a->src0.cs;
    csmax = csmax > a->mask.cs ? csmax : a->mask.cs;
    if (px * csmax * 2 >= (1ll << 31)) return false;
    if ((a->src0.co & 7) || (a->src0.cs & 7)) return false;
    return true;
}

// sign_out: the forward role of a 384-channel layer (16-bit output, no fused narrow layer behind it); mask_sign: its data-gradient
// role with a mask (not the fused un-rotation, which has unrot_smask)
bool gemm_dma_signs(const ssdn_conv_args* a) {
    if (!gemm_dma_eligible(a) || a->Mpad != 384 || a->unrot.p) return false;
    if (a->sign_out && (a->bf16 || !a->dst.p)) return false;
    if (a->mask_sign && (!a->bf16 || !a->mask.p)) return false;
    return true;
}

int gemm_dma_lds_bytes(const ssdn_conv_args* a) {
    const int tm = a->Mpad == 384 ? 384 : 96, eg = a->Mpad == 384 ? 2 : 3;
    return GD_DA * 256 * 64 + GD_DB * tm * 64 + 8 * 32 * (eg * 64 + 16) + tm * 4 + 1024 + GD_LUT_BYTES;
}

template <int WP, int WM, int NWP, int NWM, bool BF, int EPI>
static int gd_launch(const ssdn_conv_args* a, hipStream_t s, const ssdn_conv_args* a4 = nullptr) {
    constexpr int TP = NWP * WP * 32, TM = NWM * WM * 32;
    constexpr int LDS = GD_DA * TP * 64 + GD_DB * TM * 64 + NWP * NWM * 32 * (gd_eg(WM) * 64 + 16) + TM * 4 + 1024 + GD_LUT_BYTES;
    static_assert(TP == 256 && NWP * NWM == 8, "gemm_dma_lds_bytes assumes 256-pixel tiles and 8 waves");
    static_assert(LDS <= 160 * 1024, "LDS");
    static bool attr_set_dev[SSDN_MAX_DEVICES_ATTR] = {};
    bool& attr_set = attr_set_dev[ssdn_current_device_slot()];   // (function attributes are per device)
    if (!attr_set) {
        SSDN_CHECK_HIP(hipFuncSetAttribute((const void*)k_gdma<WP, WM, NWP, NWM, BF, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    GdAux x;
    x.nch = a->Ktot / 32;
    x.lp = 0;
    while ((1 << x.lp) < a->H) ++x.lp;
    const double px = (double)a->N * a->H * a->W;
    x.ntiles = (int)(px / TP);
    x.w4 = nullptr; x.b4 = nullptr; x.out32 = nullptr; x.M4 = 0; x.act4 = 0; x.HW = a->H * a->W;
    static const int env_ablate = [] { const char* e = ssdn_tuning_env("SSDN_GDMA_ABLATE"); return e ? atoi(e) : 0; }();
    x.ablate = env_ablate;
    if (a4) { x.w4 = (const h16*)a4->w; x.b4 = a4->bias; x.out32 = a4->dst32; x.M4 = a4->M; x.act4 = a4->act; }
    const int kreal = a->kreal > 0 ? a->kreal : a->Ktot;
    prof_begin(SSDN_PROF_GEMM, s);
    int cus = ssdn_device_cus();
    if (cus <= 0) cus = 256;
    const int grid = x.ntiles < cus ? x.ntiles : cus;           // persistent: one workgroup per CU (LDS-bound occupancy)
    SSDN_LAUNCH((k_gdma<WP, WM, NWP, NWM, BF, EPI>), dim3(grid), dim3(64 * NWP * NWM), LDS, s, *a, x);
    prof_end(SSDN_PROF_GEMM, s, 2.0 * px * a->M * kreal, px * (a->c0 + a->M) * 2.0);
    SSDN_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_dma(const ssdn_conv_args* a, hipStream_t s) {
    const int epi = a->mask.p ? 1 : 0;
    if (a->unrot.p) return a->unrot_smask ? gd_launch<2, 6, 4, 2, true, 10>(a, s) : gd_launch<2, 6, 4, 2, true, 2>(a, s);
    if (a->Mpad == 384) {
        if (!a->bf16) return a->sign_out ? gd_launch<2, 6, 4, 2, false, 16>(a, s) : gd_launch<2, 6, 4, 2, false, 0>(a, s);
        if (epi) return a->mask_sign ? gd_launch<2, 6, 4, 2, true, 9>(a, s) : gd_launch<2, 6, 4, 2, true, 1>(a, s);
        return gd_launch<2, 6, 4, 2, true, 0>(a, s);
    }
    if (!a->bf16) return gd_launch<1, 3, 8, 1, false, 0>(a, s);
    return epi ? gd_launch<1, 3, 8, 1, true, 1>(a, s) : gd_launch<1, 3, 8, 1, true, 0>(a, s);
}

// the narrow 1x1 layer `b` (net_out: <= 32 output channels, fp32 NCHW) directly behind the 96-channel 1x1 layer `a` can ride in a's launch
bool gemm_dma_fuses_next(const ssdn_conv_args* a, const ssdn_conv_args* b) {
    if (!gemm_dma_eligible(a) || a->bf16 || a->mask.p || a->unrot.p || a->Mpad != 96 || !a->dst.p) return false;
    if (b->bf16 || b->ntaps != 1 || b->dy[0] || b->dx[0] || b->up0 || b->c1 || b->c0 != 96 || b->Ktot != 96 || b->Mpad != 32 || b->M > 32) return false;
    if (!b->dst32 || b->mask.p || b->add.p || b->pool.p || b->upsum.p || b->unrot.p || !b->w) return false;
    if (b->src0.p != a->dst.p || b->src0.cs != a->dst.cs || b->src0.co != a->dst.co) return false;
    return b->N == a->N && b->H == a->H && b->W == a->W;
}
int launch_gemm_dma_with_next(const ssdn_conv_args* a, const ssdn_conv_args* b, hipStream_t s) {
    if (!gemm_dma_fuses_next(a, b)) return ssdn_set_error("gemm: the second layer cannot ride in this launch");
    return gd_launch<1, 3, 8, 1, false, 4>(a, s, b);
}

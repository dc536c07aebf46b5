// gradpack_dgrad.hip -- k_gradpack_dgrad: SSDN_OP_GRAD_PACK and the data gradient of the narrow net_out layer (1x1, <= 16 gradient channels
// -> 96) in ONE launch.  The backward pass opens with these two: a 5 us conversion kernel (fp32 NCHW loss gradient -> bf16 NHWC, 16 slots)
// and an 18 us k_conv launch that does ONE MFMA per tile -- both lanes wait for them (the first weight gradients need their outputs).
// Here a wave converts the gradient of 32 pixels straight into its B fragment (lanes kh = 0: channels 0..7, kh = 1: 8..15; the same 16 bytes
// are the packed row it stores), multiplies with the three 32-row blocks of the transposed weights (registers), rounds to bf16, parks the
// tile in its LDS region and finishes as the separate launch does: x LeakyReLU'(saved activation), rounded again, 16-byte pixel-contiguous
// stores.  One MFMA per block from a zero accumulator, the same conversions on the same values: bit-identical to the two launches.
#include "common.h"

#define GP_THREADS 256
#define GP_TILES 2          // 32-pixel tiles per wave

__global__ __launch_bounds__(GP_THREADS) void k_gradpack_dgrad(ssdn_grad_pack_args gp, ssdn_conv_args a) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 32 * (96 * 2 + 16)];
    constexpr int OSTR = 96 * 2 + 16;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = tid >> 6;
    char* reg = smem + wave * (32 * OSTR);
    const int HW = gp.H * gp.W;
    const long long total = (long long)gp.N * HW;
    if (blockIdx.x == 0 && tid == 0 && gp.scale_out) { gp.scale_out[0] = 1.f; gp.scale_out[1] = 1.f; }
    // transposed weights [1][96][16] (bf16): rows mt*32 + l31, k half kh
    half8 af[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) af[mt] = *reinterpret_cast<const half8*>((const h16*)a.w + (mt * 32 + l31) * 16 + kh * 8);
    for (int ti = 0; ti < GP_TILES; ++ti) {
        const long long pix0 = ((long long)(blockIdx.x * 4 + wave) * GP_TILES + ti) * 32;
        if (pix0 >= total) break;
        const long long pix = pix0 + l31;
        const int n = (int)(pix / HW), p = (int)(pix - (long long)n * HW);
        // SSDN_OP_GRAD_PACK of this pixel's channels kh*8 .. kh*8+7
        u16x8 gz = zero_b8();
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (kh * 8 + c < gp.C) gz[c] = f2bf(gp.g[((long long)n * gp.C + kh * 8 + c) * HW + p]);
        st_b8((unsigned short*)gp.dst.p + gp.dst.co + pix * gp.dst.cs + kh * 8, gz);
        // dX = W^T dZ: one MFMA per 32-row block
        f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
            const f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[mt]), __builtin_bit_cast(bf16x8, gz), zero, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2_t o;
                o[0] = pack_bf16x2(acc[g * 4 + 0], acc[g * 4 + 1]);
                o[1] = pack_bf16x2(acc[g * 4 + 2], acc[g * 4 + 3]);
                *reinterpret_cast<u32x2_t*>(reg + l31 * OSTR + (mt * 32 + g * 8 + kh * 4) * 2) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // x LeakyReLU'(saved activation), 16-byte pieces of consecutive pixels
        u32x4_t mk[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int e = lane + 64 * u, px = e / 12, cc = e - px * 12;
            mk[u] = *reinterpret_cast<const u32x4_t*>((const h16*)a.mask.p + a.mask.co + (pix0 + px) * a.mask.cs + cc * 8);
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int e = lane + 64 * u, px = e / 12, cc = e - px * 12;
            u32x4_t o = *reinterpret_cast<const u32x4_t*>(reg + px * OSTR + cc * 16);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float v0 = bf_lo(o[w]) * lrelu_grad(f16_lo(mk[u][w]));
                const float v1 = bf_hi(o[w]) * lrelu_grad(f16_hi(mk[u][w]));
                o[w] = pack_bf16x2(v0, v1);
            }
            *reinterpret_cast<u32x4_t*>((h16*)a.dst.p + a.dst.co + (pix0 + px) * a.dst.cs + cc * 8) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

// the data gradient `a` of the narrow 1x1 layer reads exactly what the SSDN_OP_GRAD_PACK `gp` in front of it writes: one launch
bool gradpack_dgrad_fusable(const ssdn_grad_pack_args* gp, const ssdn_conv_args* a) {
    if (!gp->g || !gp->dst.p || gp->cpad != 16 || gp->C > 16 || gp->C < 1) return false;
    if (!a->bf16 || a->ntaps != 1 || a->dy[0] || a->dx[0] || a->up0 || a->c1 || a->c0 != 16 || a->Ktot != 16 || a->kc != 16) return false;
    if (a->M != 96 || a->Mpad != 96 || !a->dst.p || a->dst32 || a->act || a->bias || !a->mask.p || !a->w) return false;
    if (a->add.p || a->pool.p || a->upsum.p || a->unrot.p) return false;
    if (a->src0.p != gp->dst.p || a->src0.cs != gp->dst.cs || a->src0.co != gp->dst.co) return false;
    if (a->N != gp->N || a->H != gp->H || a->W != gp->W) return false;
    const long long px = (long long)a->N * a->H * a->W;
    if (px % 32 || px >= (1ll << 31) / 256) return false;
    return !(gp->dst.cs & 7) && !(gp->dst.co & 7) && !(a->dst.cs & 7) && !(a->dst.co & 7) && !(a->mask.cs & 7) && !(a->mask.co & 7);
}

int launch_gradpack_dgrad(const ssdn_grad_pack_args* gp, const ssdn_conv_args* a, hipStream_t s) {
    if (!gradpack_dgrad_fusable(gp, a)) return ssdn_set_error("grad_pack: the data gradient behind it cannot ride in the launch");
    const long long tiles = (long long)a->N * a->H * a->W / 32;
    const int grid = (int)((tiles + 4 * GP_TILES - 1) / (4 * GP_TILES));
    SSDN_LAUNCH(k_gradpack_dgrad, dim3(grid), dim3(GP_THREADS), 0, s, *gp, *a);
    SSDN_CHECK_HIP(hipGetLastError());
    return 0;
}

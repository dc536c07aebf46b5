// elementwise.hip -- HBM-bound layout / routing kernels of the blind-spot U-Net (gfx950).
// All activation traffic is NHWC fp16 moved as 16-byte (8-channel) vectors per lane.
#include "common.h"

#define EW_BLOCK 256
static inline int ew_grid(long long n) {
    long long g = (n + EW_BLOCK - 1) / EW_BLOCK;
    return (int)(g < 1 ? 1 : g);
}

// ------------------------------------------------------------------------------------------------
// PACK_INPUT: rotate-stack, NCHW f32 -> NHWC f16   (noise_network.py:187-189, utils/data.py:42-67)
// ------------------------------------------------------------------------------------------------
__global__ void k_pack_input(ssdn_pack_input_args a) {
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;   // (launchers reject > 2^31 elements: 32-bit index math)
    const int H = a.H, W = a.W;
    long long total = (long long)a.R * a.B * H * W;
    if (idx >= total) return;
    int j = idx % W;
    int i = (idx / W) % H;
    int nb = idx / (unsigned)(W * H);
    int r = nb / a.B, b = nb % a.B;
    int sy, sx;  // source coordinates in the un-rotated image
    switch (r) {
        case 0: sy = i; sx = j; break;
        case 1: sy = j; sx = W - 1 - i; break;          // rotate(x,90)[i,j]  = x[j, W-1-i]
        case 2: sy = H - 1 - i; sx = W - 1 - j; break;  // rotate(x,180)[i,j] = x[H-1-i, W-1-j]
        default: sy = H - 1 - j; sx = i; break;         // rotate(x,270)[i,j] = x[H-1-j, i]
    }
    h16* d = (h16*)a.dst.p + (long long)idx * a.dst.cs + a.dst.co;
    for (int c0 = 0; c0 < a.cpad; c0 += 8) {
        half8 v = zero_h8();
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c0 + c < a.C) v[c] = (h16)a.src[(((long long)b * a.C + c0 + c) * H + sy) * W + sx];
        st_h8(d + c0, v);
    }
}
int launch_pack_input(const ssdn_pack_input_args* a, hipStream_t s) {
    if (a->R == 4 && a->H != a->W) return ssdn_set_error("pack_input: blind-spot rotation needs square images");
    long long n = (long long)a->R * a->B * a->H * a->W;
    if (n >= (1ll << 31)) return ssdn_set_error("pack_input: too many elements for 32-bit indexing");
    hipLaunchKernelGGL(k_pack_input, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, *a);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// POOL_FWD / POOL_BWD  (noise_network.py:64-67; models/utility.py:37-53)
// ------------------------------------------------------------------------------------------------
__global__ void k_pool_fwd(ssdn_pool_args a) {
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;   // (launchers reject > 2^31 elements: 32-bit index math)
    const int C8 = a.C >> 3, Ho = a.H >> 1, Wo = a.W >> 1;
    long long total = (long long)a.N * Ho * Wo * C8;
    if (idx >= total) return;
    int c = (idx % C8) * 8;
    const unsigned p = idx / C8;
    int j = p % Wo;
    int i = (p / Wo) % Ho;
    int n = p / (unsigned)(Wo * Ho);
    const h16* src = (const h16*)a.act.p + a.act.co + c;
    int r0 = a.shifted ? 2 * i - 1 : 2 * i;
    const bool padrow = a.shifted && r0 < 0;
    float m[8];
    unsigned win[8];       // route nibble of the channel: bits 0-1 = window position of the first maximum, bit 2 = the zero pad row holds it
#pragma unroll
    for (int q = 0; q < 8; ++q) { m[q] = padrow ? 0.f : -65504.f; win[q] = 4u; }  // literal zero row takes part (and is scanned first)
#pragma unroll
    for (int dr = 0; dr < 2; ++dr) {
        int y = r0 + dr;
        if (y < 0) continue;
#pragma unroll
        for (int dc = 0; dc < 2; ++dc) {
            half8 v = ld_h8(src + (((long long)n * a.H + y) * a.W + 2 * j + dc) * a.act.cs);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float f = (float)v[q];
                // (first maximum in scan order: a later position only takes over when strictly greater; the first real position always
                //  takes over from the -65504 start value of an un-padded window, where no activation is smaller)
                if (f > m[q] || (!padrow && dr == 0 && dc == 0)) win[q] = (unsigned)(2 * dr + dc);
                m[q] = fmaxf(m[q], f);
            }
        }
    }
    half8 o;
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = (h16)m[q];
    st_h8((h16*)a.pooled.p + a.pooled.co + c + (((long long)n * Ho + i) * Wo + j) * a.pooled.cs, o);
    if (a.route) {         // + bit 3 = the winner is > 0 (LeakyReLU' = 1): everything SSDN_OP_POOL_BWD needs of `act`
        unsigned r = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) r |= (win[q] | (m[q] > 0.f ? 8u : 0u)) << (4 * q);
        ((unsigned*)a.route)[idx] = r;
    }
}
int launch_pool_fwd(const ssdn_pool_args* a, hipStream_t s) {
    if ((a->C & 7) || (a->H & 1) || (a->W & 1)) return ssdn_set_error("pool: C%%8, H%%2, W%%2 must be 0");
    long long n = (long long)a->N * (a->H / 2) * (a->W / 2) * (a->C / 8);
    if (n >= (1ll << 31)) return ssdn_set_error("k_pool_fwd: too many elements for 32-bit indexing");
    hipLaunchKernelGGL(k_pool_fwd, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, *a);
    return 0;
}

__global__ void k_pool_bwd(ssdn_pool_args a) {
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;   // (launchers reject > 2^31 elements: 32-bit index math)
    const int C8 = a.C >> 3, Ho = a.H >> 1, Wo = a.W >> 1;
    long long total = (long long)a.N * Ho * Wo * C8;
    if (idx >= total) return;
    int c = (idx % C8) * 8;
    const unsigned p = idx / C8;
    int j = p % Wo;
    int i = (p / Wo) % Ho;
    int n = p / (unsigned)(Wo * Ho);
    const h16* act = (const h16*)a.act.p + a.act.co + c;
    unsigned short* dz = (unsigned short*)a.dz.p + a.dz.co + c;                     // gradients are bf16
    u16x8 g = ld_b8((const unsigned short*)a.dpool.p + a.dpool.co + c + (((long long)n * Ho + i) * Wo + j) * a.dpool.cs);
    int r0 = a.shifted ? 2 * i - 1 : 2 * i;
    if (a.route) {
        // the forward pass left the route word of this (window, 8 channels): winner position, "the pad row won", winner > 0 -- the
        // same decisions the scan below takes from `act` (bit-identical), for 4 bytes instead of 64
        const unsigned r = ((const unsigned*)a.route)[idx];
        float gs[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) gs[q] = bf2f(g[q]) * (((r >> (4 * q)) & 8u) ? 1.f : LRELU_SLOPE);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int y = r0 + (k >> 1);
            if (y < 0) continue;
            u16x8 o;
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = ((r >> (4 * q)) & 7u) == (unsigned)k ? f2bf(gs[q]) : (unsigned short)0;
            st_b8(dz + (((long long)n * a.H + y) * a.W + 2 * j + (k & 1)) * a.dz.cs, o);
        }
        if (a.shifted && i == Ho - 1) {
            u16x8 z = zero_b8();
            st_b8(dz + (((long long)n * a.H + a.H - 1) * a.W + 2 * j) * a.dz.cs, z);
            st_b8(dz + (((long long)n * a.H + a.H - 1) * a.W + 2 * j + 1) * a.dz.cs, z);
        }
        return;
    }
    half8 v[4];
    float m[8];
    bool taken[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        m[q] = (a.shifted && r0 < 0) ? 0.f : -65504.f;
        taken[q] = false;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int y = r0 + (k >> 1);
        if (y < 0) { v[k] = zero_h8(); continue; }
        v[k] = ld_h8(act + (((long long)n * a.H + y) * a.W + 2 * j + (k & 1)) * a.act.cs);
#pragma unroll
        for (int q = 0; q < 8; ++q) m[q] = fmaxf(m[q], (float)v[k][q]);
    }
    // the zero pad row is scanned first: if it holds the max, the gradient is dropped
    if (a.shifted && r0 < 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) taken[q] = (m[q] == 0.f);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int y = r0 + (k >> 1);
        if (y < 0) continue;
        u16x8 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float av = (float)v[k][q];
            bool hit = !taken[q] && av == m[q];
            if (hit) taken[q] = true;
            o[q] = hit ? f2bf(bf2f(g[q]) * lrelu_grad(av)) : (unsigned short)0;
        }
        st_b8(dz + (((long long)n * a.H + y) * a.W + 2 * j + (k & 1)) * a.dz.cs, o);
    }
    // shifted pooling never looks at the last row: its gradient is zero
    if (a.shifted && i == Ho - 1) {
        u16x8 z = zero_b8();
        st_b8(dz + (((long long)n * a.H + a.H - 1) * a.W + 2 * j) * a.dz.cs, z);
        st_b8(dz + (((long long)n * a.H + a.H - 1) * a.W + 2 * j + 1) * a.dz.cs, z);
    }
}
int launch_pool_bwd(const ssdn_pool_args* a, hipStream_t s) {
    if ((a->C & 7) || (a->H & 1) || (a->W & 1)) return ssdn_set_error("pool: C%%8, H%%2, W%%2 must be 0");
    long long n = (long long)a->N * (a->H / 2) * (a->W / 2) * (a->C / 8);
    if (n >= (1ll << 31)) return ssdn_set_error("k_pool_bwd: too many elements for 32-bit indexing");
    SSDN_LAUNCH(k_pool_bwd, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, *a);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// UPSUM_BWD: adjoint of nearest 2x upsample (+ LeakyReLU' of the producer)   (noise_network.py:102,110,120)
// ------------------------------------------------------------------------------------------------
__global__ void k_upsum_bwd(ssdn_upsum_args a) {
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;   // (launchers reject > 2^31 elements: 32-bit index math)
    const int C8 = a.C >> 3;
    long long total = (long long)a.N * a.H * a.W * C8;
    if (idx >= total) return;
    int c = (idx % C8) * 8;
    const unsigned p = idx / C8;
    int j = p % a.W;
    int i = (p / a.W) % a.H;
    int n = p / (unsigned)(a.W * a.H);
    const unsigned short* src = (const unsigned short*)a.src.p + a.src.co + c;   // bf16 gradient
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        u16x8 v = ld_b8(src + (((long long)n * 2 * a.H + 2 * i + (k >> 1)) * 2 * a.W + 2 * j + (k & 1)) * a.src.cs);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += bf2f(v[q]);
    }
    half8 mk = ld_h8((const h16*)a.mask.p + a.mask.co + c + (long long)p * a.mask.cs);
    u16x8 o;
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = f2bf(acc[q] * lrelu_grad((float)mk[q]));
    st_b8((unsigned short*)a.dst.p + a.dst.co + c + (long long)p * a.dst.cs, o);
}
int launch_upsum_bwd(const ssdn_upsum_args* a, hipStream_t s) {
    if (a->C & 7) return ssdn_set_error("upsum: C%%8 must be 0");
    long long n = (long long)a->N * a->H * a->W * (a->C / 8);
    if (n >= (1ll << 31)) return ssdn_set_error("k_upsum_bwd: too many elements for 32-bit indexing");
    SSDN_LAUNCH(k_upsum_bwd, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, *a);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// UNROT_FWD / UNROT_BWD  (noise_network.py:213-222)
//   aligned_r[i,j] = S_r[u,v], S_r[u,v] = u>=1 ? Y[rB+b, u-1, v] : 0
//   r=0: (u,v)=(i,j); r=1 (rot 270): (P-1-j, i); r=2 (rot 180): (P-1-i, P-1-j); r=3 (rot 90): (j, P-1-i)
// ------------------------------------------------------------------------------------------------
__global__ void k_unrot_fwd(ssdn_unrot_args a) {
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;   // (launchers reject > 2^31 elements: 32-bit index math)
    const int P = a.P, C8 = a.C >> 3;
    long long total = (long long)a.B * P * P * 4 * C8;
    if (idx >= total) return;
    int c = (idx % C8) * 8;
    const unsigned t = idx / C8;
    int r = t & 3;
    const unsigned p = t >> 2;
    int j = p % P;
    int i = (p / P) % P;
    int b = p / (unsigned)(P * P);
    int u, v;
    switch (r) {
        case 0: u = i; v = j; break;
        case 1: u = P - 1 - j; v = i; break;
        case 2: u = P - 1 - i; v = P - 1 - j; break;
        default: u = j; v = P - 1 - i; break;
    }
    half8 val = zero_h8();
    if (u >= 1) {
        const long long spix = (((long long)r * a.B + b) * P + (u - 1)) * P + v;
        val = ld_h8((const h16*)a.src.p + a.src.co + c + spix * a.src.cs);
        if (a.smask) {                                   // every source pixel with y <= P-2 is read exactly once: its sign byte
            unsigned m = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) m |= ((float)val[q] > 0.f ? 1u : 0u) << q;
            ((unsigned char*)a.smask)[spix * C8 + (c >> 3)] = (unsigned char)m;
        }
    }
    st_h8((h16*)a.dst.p + a.dst.co + r * a.C + c + (long long)p * a.dst.cs, val);
}
int launch_unrot_fwd(const ssdn_unrot_args* a, hipStream_t s) {
    if (a->C & 7) return ssdn_set_error("unrot: C%%8 must be 0");
    long long n = (long long)a->B * a->P * a->P * 4 * (a->C / 8);
    if (n >= (1ll << 31)) return ssdn_set_error("k_unrot_fwd: too many elements for 32-bit indexing");
    hipLaunchKernelGGL(k_unrot_fwd, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, *a);
    return 0;
}

__global__ void k_unrot_bwd(ssdn_unrot_args a) {
    // dY[rB+b, y, x, c] = (y+1 < P ? dU[b, i, j, r*C + c] : 0) * lrelu'(Y),  (u,v) = (y+1, x) -> (i,j) inverse map
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;   // (launchers reject > 2^31 elements: 32-bit index math)
    const int P = a.P, C8 = a.C >> 3;
    long long total = (long long)4 * a.B * P * P * C8;
    if (idx >= total) return;
    int c = (idx % C8) * 8;
    const unsigned p = idx / C8;
    int x = p % P;
    int y = (p / P) % P;
    int nb = p / (unsigned)(P * P);
    int r = nb / a.B, b = nb % a.B;
    u16x8 o = zero_b8();
    int u = y + 1, v = x;
    if (u < P) {
        int i, j;
        switch (r) {
            case 0: i = u; j = v; break;
            case 1: i = v; j = P - 1 - u; break;          // u = P-1-j, v = i
            case 2: i = P - 1 - u; j = P - 1 - v; break;
            default: i = P - 1 - v; j = u; break;         // u = j, v = P-1-i
        }
        u16x8 g = ld_b8((const unsigned short*)a.src.p + a.src.co + r * a.C + c + (((long long)b * P + i) * P + j) * a.src.cs);
        half8 mk = ld_h8((const h16*)a.mask.p + a.mask.co + c + (long long)p * a.mask.cs);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = f2bf(bf2f(g[q]) * lrelu_grad((float)mk[q]));
    }
    st_b8((unsigned short*)a.dst.p + a.dst.co + c + (long long)p * a.dst.cs, o);
}
int launch_unrot_bwd(const ssdn_unrot_args* a, hipStream_t s) {
    if (a->C & 7) return ssdn_set_error("unrot: C%%8 must be 0");
    long long n = (long long)4 * a->B * a->P * a->P * (a->C / 8);
    if (n >= (1ll << 31)) return ssdn_set_error("k_unrot_bwd: too many elements for 32-bit indexing");
    SSDN_LAUNCH(k_unrot_bwd, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, *a);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// GRAD_PACK: fp32 NCHW loss gradient -> fp16 NHWC with a power-of-two loss scale
// ------------------------------------------------------------------------------------------------
__global__ void k_grad_pack(ssdn_grad_pack_args a) {
    // gradients travel as bf16 (fp32 exponent range): no loss scale is needed, scale_out is written as {1, 1}
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;   // (launchers reject > 2^31 elements: 32-bit index math)
    const unsigned HW = a.H * a.W;
    long long total = (long long)a.N * HW;
    if (idx == 0 && a.scale_out) {
        a.scale_out[0] = 1.f;
        a.scale_out[1] = 1.f;
    }
    if (idx >= total) return;
    int n = idx / HW;
    const unsigned pix = idx % HW;
    unsigned short* d = (unsigned short*)a.dst.p + a.dst.co + (long long)idx * a.dst.cs;
    for (int c0 = 0; c0 < a.cpad; c0 += 8) {
        u16x8 v = zero_b8();
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c0 + c < a.C) v[c] = f2bf(a.g[((long long)n * a.C + c0 + c) * HW + pix]);
        st_b8(d + c0, v);
    }
}
int launch_grad_pack(const ssdn_grad_pack_args* a, hipStream_t s) {
    long long n = (long long)a->N * a->H * a->W;
    if (n >= (1ll << 31)) return ssdn_set_error("k_grad_pack: too many elements for 32-bit indexing");
    SSDN_LAUNCH(k_grad_pack, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, *a);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// WPACK: fp32 OIHW master -> fp16 MFMA shadows (forward [t][m][k], dgrad [t][c][m])
// ------------------------------------------------------------------------------------------------
static __device__ __forceinline__ int k_to_cin(int k, int c0, int c1_real) {
    if (k < c0) return k;
    int r = k - c0;
    return r < c1_real ? c0 + r : -1;
}
// position of element (tap t, row m, column k) in the chunk-major pre-swizzled copy [tap][chunk][rows][kc] (ssdn_conv_args.wc)
static __device__ __forceinline__ long long wpack_cm(int t, int m, int k, int rows, int K) {
    const int nfull = K / 48;
    const int c = k / 48 < nfull ? k / 48 : nfull;
    const int kk = k - c * 48, kcw = c < nfull ? 48 : 16;
    const int piece = (kk >> 3) ^ ((m >> 3) & 1);
    return (long long)t * rows * K + (long long)c * 48 * rows + (long long)m * kcw + piece * 8 + (kk & 7);
}
static __device__ __forceinline__ void wpack_element(const ssdn_wpack_args& a, long long idx) {
    long long nf = (long long)a.ntaps * a.Mpad_f * a.Ktot;
    long long nd = a.wd ? (long long)a.ntaps * a.Mpad_d * a.Kd : 0;
    if (idx < nf) {
        int k = idx % a.Ktot;
        int m = (idx / a.Ktot) % a.Mpad_f;
        int t = idx / ((long long)a.Ktot * a.Mpad_f);
        int ci = k_to_cin(k, a.c0, a.c1_real);
        float v = (m < a.M && ci >= 0) ? a.w[((long long)m * a.cin + ci) * a.ntaps + t] : 0.f;
        ((h16*)a.wf)[idx] = (h16)v;
        if (a.wfc) ((h16*)a.wfc)[wpack_cm(t, m, k, a.Mpad_f, a.Ktot)] = (h16)v;
    } else if (idx < nf + nd) {
        long long e = idx - nf;
        int m = e % a.Kd;                       // reduction index of the dgrad GEMM = forward output channel
        int c = (e / a.Kd) % a.Mpad_d;          // output index of the dgrad GEMM = forward input channel slot
        int t = e / ((long long)a.Kd * a.Mpad_d);
        int ci = c < a.Ktot ? k_to_cin(c, a.c0, a.c1_real) : -1;
        float v = (m < a.M && ci >= 0) ? a.w[((long long)m * a.cin + ci) * a.ntaps + t] : 0.f;
        ((unsigned short*)a.wd)[e] = f2bf(v);   // data-gradient shadow is bf16 (gradients are bf16)
        if (a.wdc) ((unsigned short*)a.wdc)[wpack_cm(t, c, m, a.Mpad_d, a.Kd)] = f2bf(v);
    }
}
static inline long long wpack_count(const ssdn_wpack_args* a) {
    return (long long)a->ntaps * a->Mpad_f * a->Ktot + (a->wd ? (long long)a->ntaps * a->Mpad_d * a->Kd : 0);
}
__global__ void k_wpack(ssdn_wpack_args a) { wpack_element(a, (long long)blockIdx.x * blockDim.x + threadIdx.x); }
int launch_wpack(const ssdn_wpack_args* a, hipStream_t s) {
    hipLaunchKernelGGL(k_wpack, dim3(ew_grid(wpack_count(a))), dim3(EW_BLOCK), 0, s, *a);
    return 0;
}
// Every layer's shadows in ONE launch: the executor merges runs of consecutive SSDN_OP_WPACK ops (after an optimiser step
// all ~20 layers are re-packed; 20 launches of ~4 us each were 2.5 % of a training step).  The per-layer descriptors travel
// in the kernel argument segment; a block serves one layer.
struct WpackTable {
    ssdn_wpack_args e[WPACK_MULTI_MAX];
    int bstart[WPACK_MULTI_MAX + 1];   // first block of each entry
    int n;
};
__global__ void k_wpack_multi(WpackTable t) {
    int i = 0;
    while (i + 1 < t.n && (int)blockIdx.x >= t.bstart[i + 1]) ++i;
    wpack_element(t.e[i], (long long)(blockIdx.x - t.bstart[i]) * blockDim.x + threadIdx.x);
}
int launch_wpack_multi(const ssdn_wpack_args* const* items, int n, hipStream_t s) {
    if (n < 1 || n > WPACK_MULTI_MAX) return ssdn_set_error("wpack: bad batch size %d", n);
    WpackTable t;
    t.n = n;
    int b = 0;
    for (int i = 0; i < n; ++i) {
        t.e[i] = *items[i];
        t.bstart[i] = b;
        b += ew_grid(wpack_count(items[i]));
    }
    t.bstart[n] = b;
    if (b > 0) hipLaunchKernelGGL(k_wpack_multi, dim3(b), dim3(EW_BLOCK), 0, s, t);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// WREDUCE: ordered sum of the per-workgroup weight-gradient slabs -> fp32 OIHW gradient
// ------------------------------------------------------------------------------------------------
#define WR_GROUP 32
// stage 1: float4 per thread, 32 slabs per group, partial sum written back over the group's first slab
__global__ void k_wreduce_partial(float* slab, long long stride, int nslabs) {
    long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 * 4 >= stride) return;
    int s0 = blockIdx.y * WR_GROUP;
    int s1 = s0 + WR_GROUP < nslabs ? s0 + WR_GROUP : nslabs;
    float4* p = reinterpret_cast<float4*>(slab + i4 * 4);
    const long long st4 = stride / 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = s0;
    for (; s + 8 <= s1; s += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long long)(s + u) * st4];
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; s < s1; ++s) { float4 v = p[(long long)s * st4]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    p[(long long)s0 * st4] = acc;
}
// stage 2 (or the only stage when nslabs <= 32): one thread per FOUR consecutive slab elements in SLAB order (k fastest; Kpad is a
// multiple of 4, so they share their row), slabs s = 0, step, 2*step, ... summed in order -- eight 16-byte loads in flight per round
// (one 4-byte load per slab in a dependent loop was a latency chain: 24 us per bucket alone, 70 us in situ); writes the OIHW gradient.
static __device__ __forceinline__ void wreduce_final(const ssdn_wreduce_args& a, int step, long long idx) {
    const long long stride = (long long)a.ntaps * a.Mpad * a.Kpad, stride4 = stride >> 2;
    float inv = a.inv_scale ? *a.inv_scale : 1.f;
    if (idx < stride4) {
        const long long e0 = idx * 4;
        // (32-bit index arithmetic: a slab has < 2^31 elements -- the launcher's grids are ints; three 64-bit divisions per thread were
        //  most of this kernel's instructions)
        const unsigned e32 = (unsigned)e0, row = e32 / (unsigned)a.Kpad;
        int k = (int)(e32 - row * (unsigned)a.Kpad);
        int t = (int)(row / (unsigned)a.Mpad);
        int m = (int)(row - (unsigned)t * (unsigned)a.Mpad);
        const int klim = a.tapblock ? a.Kpad : a.cin;
        if (k < klim && (a.tapblock ? t * a.Kpad + k : k) < a.cin && m < a.M) {
            const float4* p = reinterpret_cast<const float4*>(a.slab + e0);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const long long st4 = stride4 * step;
            const int cnt = (a.nslabs + step - 1) / step;
            int s = 0;
            for (; s + 8 <= cnt; s += 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = p[(long long)(s + u) * st4];
#pragma unroll
                for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
            for (; s < cnt; ++s) { const float4 v = p[(long long)s * st4]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            const float r[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kj = k + j, ci = a.tapblock ? t * a.Kpad + kj : kj;
                if (kj < klim && ci < a.cin) {
                    long long o = a.tapblock ? ((long long)(a.m_off + m) * a.cin_full + a.c_off + ci)
                                             : ((long long)(a.m_off + m) * a.cin_full + a.c_off + kj) * a.ntaps + t;
                    a.gw[o] = r[j] * inv;
                }
            }
        }
    } else if (idx < stride4 + a.M && a.gb) {
        // bias gradient: 16 independent loads in flight per round (a dependent one-at-a-time loop over 256 slabs is
        // pure L2 latency: it alone cost 30-60 us per layer), summed in slab order
        int m = idx - stride4;
        float acc = 0.f;
        int s = 0;
        for (; s + 16 <= a.nslabs; s += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = a.bslab[(long long)(s + u) * a.Mpad + m];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += v[u];
        }
        for (; s < a.nslabs; ++s) acc += a.bslab[(long long)s * a.Mpad + m];
        a.gb[a.m_off + m] = acc * inv;
    }
}
__global__ void k_wreduce(ssdn_wreduce_args a, int step) {
    wreduce_final(a, step, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}
// ---- many reductions in two launches ------------------------------------------------------------------------------------------
// A training step ends up with ~28 slab reductions; as separate launches (stage 1 + stage 2 each) they are latency-bound --
// 51 launches, 0.52 ms per step measured in situ for 0.75 GB of slab traffic (1.4 TB/s).  The executor merges a run of
// consecutive SSDN_OP_WREDUCE ops (the host emits a gradient bucket's reductions together, after the bucket's last
// weight-gradient GEMM) into ONE stage-1 and ONE stage-2 launch that cover all of them at full-chip parallelism.  Same sums
// in the same order as the single-entry kernels: bit-identical results.
struct WrTable {
    ssdn_wreduce_args e[WREDUCE_MULTI_MAX];
    int bstart[WREDUCE_MULTI_MAX + 1];    // first block of each entry (stage 2) / of each entry's stage-1 grid
    int gx[WREDUCE_MULTI_MAX];            // stage 1: blocks per slab group of the entry (0: entry has no stage 1)
    int step[WREDUCE_MULTI_MAX];          // stage 2: slab stride (WR_GROUP after a stage 1, else 1)
    int n;
};
__global__ void k_wreduce_partial_multi(WrTable t) {
    int i = 0;
    while (i + 1 < t.n && (int)blockIdx.x >= t.bstart[i + 1]) ++i;
    const ssdn_wreduce_args& a = t.e[i];
    const int local = blockIdx.x - t.bstart[i];
    if (t.gx[i] == 0) return;
    const int bx = local % t.gx[i], by = local / t.gx[i];
    const long long stride = (long long)a.ntaps * a.Mpad * a.Kpad;
    long long i4 = (long long)bx * blockDim.x + threadIdx.x;
    if (i4 * 4 >= stride) return;
    int s0 = by * WR_GROUP;
    int s1 = s0 + WR_GROUP < a.nslabs ? s0 + WR_GROUP : a.nslabs;
    float4* p = reinterpret_cast<float4*>(const_cast<float*>(a.slab) + i4 * 4);
    const long long st4 = stride / 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = s0;
    for (; s + 8 <= s1; s += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long long)(s + u) * st4];
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; s < s1; ++s) { float4 v = p[(long long)s * st4]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    p[(long long)s0 * st4] = acc;
}
static __device__ __forceinline__ void wreduce_final(const ssdn_wreduce_args& a, int step, long long idx);
__global__ void k_wreduce_multi(WrTable t) {
    int i = 0;
    while (i + 1 < t.n && (int)blockIdx.x >= t.bstart[i + 1]) ++i;
    wreduce_final(t.e[i], t.step[i], (long long)(blockIdx.x - t.bstart[i]) * blockDim.x + threadIdx.x);
}
int launch_wreduce_multi(const ssdn_wreduce_args* const* items, int n, hipStream_t s) {
    if (n < 1 || n > WREDUCE_MULTI_MAX) return ssdn_set_error("wreduce: bad batch size %d", n);
    WrTable t1, t2;
    t1.n = t2.n = n;
    int b1 = 0, b2 = 0;
    for (int i = 0; i < n; ++i) {
        const ssdn_wreduce_args* a = items[i];
        const long long stride = (long long)a->ntaps * a->Mpad * a->Kpad;
        t1.e[i] = t2.e[i] = *a;
        t1.bstart[i] = b1;
        t2.bstart[i] = b2;
        if (a->nslabs > WR_GROUP) {
            t1.gx[i] = ew_grid(stride / 4);
            b1 += t1.gx[i] * ((a->nslabs + WR_GROUP - 1) / WR_GROUP);
            t2.step[i] = WR_GROUP;
        } else {
            t1.gx[i] = 0;
            t2.step[i] = 1;
        }
        t1.step[i] = t2.step[i];
        t2.gx[i] = t1.gx[i];
        if (a->Kpad & 3) return ssdn_set_error("wreduce: Kpad must be a multiple of 4");
        b2 += ew_grid(stride / 4 + a->M);
    }
    t1.bstart[n] = b1;
    t2.bstart[n] = b2;
    if (b1 > 0) hipLaunchKernelGGL(k_wreduce_partial_multi, dim3(b1), dim3(EW_BLOCK), 0, s, t1);
    if (b2 > 0) SSDN_LAUNCH(k_wreduce_multi, dim3(b2), dim3(EW_BLOCK), 0, s, t2);
    return 0;
}

int launch_wreduce(const ssdn_wreduce_args* a, hipStream_t s) {
    long long stride = (long long)a->ntaps * a->Mpad * a->Kpad;
    int step = 1;
    if (a->nslabs > WR_GROUP) {
        int groups = (a->nslabs + WR_GROUP - 1) / WR_GROUP;
        hipLaunchKernelGGL(k_wreduce_partial, dim3(ew_grid(stride / 4), groups), dim3(EW_BLOCK), 0, s, (float*)a->slab, stride, a->nslabs);
        step = WR_GROUP;
    }
    if (a->Kpad & 3) return ssdn_set_error("wreduce: Kpad must be a multiple of 4");
    long long n = stride / 4 + a->M;
    SSDN_LAUNCH(k_wreduce, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, *a, step);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// ADAM  (train.py:100-107,202): one fused pass over the flat master buffer
// ------------------------------------------------------------------------------------------------
// one element of the Adam update (shared by k_adam and k_adam_pack: the two must round identically; explicit fmaf / no re-association)
struct AdamIn { float g, m, v, p; };
static __device__ __forceinline__ AdamIn adam_load(const ssdn_adam_args& a, long long i) { return AdamIn{a.g[i], a.m[i], a.v[i], a.p[i]}; }
static __device__ __forceinline__ float adam_apply(const ssdn_adam_args& a, long long i, const AdamIn& q, float inv_bc2s, float step) {
    const float g = q.g * a.gscale;
    const float m = __fmaf_rn(a.b1, q.m, __fmul_rn(1.f - a.b1, g));
    const float v = __fmaf_rn(a.b2, q.v, __fmul_rn(__fmul_rn(1.f - a.b2, g), g));
    a.m[i] = m;
    a.v[i] = v;
    const float den = __fmaf_rn(sqrtf(v), inv_bc2s, a.eps);
    const float pn = __fsub_rn(q.p, __fdiv_rn(__fmul_rn(step, m), den));
    a.p[i] = pn;
    return pn;
}
static __device__ __forceinline__ float adam_elem(const ssdn_adam_args& a, long long i, float inv_bc2s, float step) {
    return adam_apply(a, i, adam_load(a, i), inv_bc2s, step);
}
__global__ void k_adam(ssdn_adam_args a) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    float inv_bc2s = 1.f / sqrtf(a.bc2);
    float step = a.lr / a.bc1;
    for (; i < a.n; i += stride) adam_elem(a, i, inv_bc2s, step);
}
// ADAM + WPACK in one pass (the executor fuses an SSDN_OP_ADAM that is directly followed by the SSDN_OP_WPACK ops of layers whose
// weights lie in its range): the thread that updates a weight also writes its 16-bit images into the MFMA shadows -- the value it
// just computed, converted exactly as k_wpack converts it (bit-identical shadows; the padding of the shadows is never rewritten, it
// stays the zeros of the first re-pack).  Saves the re-pack launch after every optimiser step (23 us) and one read of the parameters.
// Work split: a block owns ONE tile of one layer -- 8 output channels x 48 input channels x all taps -- or 1024 elements of a stretch
// without shadows (biases, the learnable sigma).  The tile is read in master order (OIHW: runs of 48 x ntaps floats), updated, parked in
// LDS and written out in SHADOW order: [tap][row][k] runs of 96 bytes for the forward shadows, 16-byte pieces (8 consecutive output
// channels) for the transposed data-gradient shadows.  (One thread per parameter scattered 2-byte stores over four tensors: 31 us.)
#define AP_TM 8
#define AP_TK 48
struct AdamPackTable {
    ssdn_adam_args a;
    ssdn_wpack_args e[ADAM_PACK_MAX];
    long long w_off[ADAM_PACK_MAX];     // first element of the layer's weight tensor in the Adam range (ascending)
    int bstart[ADAM_PACK_MAX + 1];      // first block of the layer's tiles
    int tiles_k[ADAM_PACK_MAX];         // 48-channel tiles per row of tiles
    long long gap_start[ADAM_PACK_MAX + 2];   // stretches of the range without shadows: [gap_start[g], gap_end[g])
    long long gap_end[ADAM_PACK_MAX + 2];
    int gap_bstart[ADAM_PACK_MAX + 3];  // first block of each stretch (after all tile blocks)
    int n, ngaps;
};
template <int NT>
static __device__ __forceinline__ void adam_pack_tile(const AdamPackTable& t, int j, int tile, float* lds, float inv_bc2s, float step) {
    const ssdn_adam_args& a = t.a;
    const ssdn_wpack_args& w = t.e[j];
    const int tk = t.tiles_k[j];
    const int mo0 = (tile / tk) * AP_TM, k0 = (tile % tk) * AP_TK;
    const int tid = threadIdx.x;
    // phase 1: master order; the loads of up to 7 elements of a thread are in flight together (one element at a time was a chain of 14
    // exposed HBM round trips per thread)
    constexpr int TOTAL = AP_TM * AP_TK * NT, NIT = (TOTAL + EW_BLOCK - 1) / EW_BLOCK, BATCH = NIT < 7 ? NIT : 7;
#pragma unroll
    for (int it0 = 0; it0 < NIT; it0 += BATCH) {
        AdamIn q[BATCH];
        long long idx[BATCH];
        bool ok[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = tid + (it0 + u) * EW_BLOCK;
            const int mo = e / (AP_TK * NT), r = e - mo * (AP_TK * NT);
            const int kk = r / NT, tp = r - kk * NT;
            ok[u] = it0 + u < NIT && e < TOTAL && mo0 + mo < w.M && k0 + kk < w.cin;
            idx[u] = ok[u] ? t.w_off[j] + ((long long)(mo0 + mo) * w.cin + k0 + kk) * NT + tp : t.w_off[j];
            q[u] = adam_load(a, idx[u]);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = tid + (it0 + u) * EW_BLOCK;
            if (it0 + u >= NIT || e >= TOTAL) continue;
            const int mo = e / (AP_TK * NT), r = e - mo * (AP_TK * NT);
            const int kk = r / NT, tp = r - kk * NT;
            lds[(tp * AP_TM + mo) * (AP_TK + 1) + kk] = ok[u] ? adam_apply(a, idx[u], q[u], inv_bc2s, step) : 0.f;
        }
    }
    __syncthreads();
    // phase 2a: forward shadows, k fastest
    for (int e = tid; e < NT * AP_TM * AP_TK; e += EW_BLOCK) {
        const int tp = e / (AP_TM * AP_TK), r = e - tp * (AP_TM * AP_TK);
        const int mo = r / AP_TK, kk = r - mo * AP_TK;
        const int m = mo0 + mo, k = k0 + kk;
        if (m >= w.M || k >= w.cin) continue;
        const float pn = lds[(tp * AP_TM + mo) * (AP_TK + 1) + kk];
        ((h16*)w.wf)[((long long)tp * w.Mpad_f + m) * w.Ktot + k] = (h16)pn;
        if (w.wfc) ((h16*)w.wfc)[wpack_cm(tp, m, k, w.Mpad_f, w.Ktot)] = (h16)pn;
    }
    // phase 2b: data-gradient shadows (transposed), output channel fastest
    if (w.wd) {
        for (int e = tid; e < NT * AP_TK * AP_TM; e += EW_BLOCK) {
            const int tp = e / (AP_TK * AP_TM), r = e - tp * (AP_TK * AP_TM);
            const int kk = r / AP_TM, mo = r - kk * AP_TM;
            const int m = mo0 + mo, k = k0 + kk;
            if (m >= w.M || k >= w.cin || k >= w.Mpad_d || m >= w.Kd) continue;   // (the shadow may cover fewer input slots: decode_block_1.0's image channels need no gradient)
            const float pn = lds[(tp * AP_TM + mo) * (AP_TK + 1) + kk];
            ((unsigned short*)w.wd)[((long long)tp * w.Mpad_d + k) * w.Kd + m] = f2bf(pn);
            if (w.wdc) ((unsigned short*)w.wdc)[wpack_cm(tp, k, m, w.Mpad_d, w.Kd)] = f2bf(pn);
        }
    }
}
__global__ void k_adam_pack(AdamPackTable t) {
    __shared__ float lds[9 * AP_TM * (AP_TK + 1)];
    const ssdn_adam_args& a = t.a;
    const float inv_bc2s = 1.f / sqrtf(a.bc2);
    const float step = a.lr / a.bc1;
    const int b = blockIdx.x;
    if (b < t.bstart[t.n]) {
        int j = 0;
        while (j + 1 < t.n && b >= t.bstart[j + 1]) ++j;
        if (t.e[j].ntaps == 9) adam_pack_tile<9>(t, j, b - t.bstart[j], lds, inv_bc2s, step);
        else adam_pack_tile<1>(t, j, b - t.bstart[j], lds, inv_bc2s, step);
    } else {
        int g = 0;
        while (g + 1 < t.ngaps && b >= t.gap_bstart[g + 1]) ++g;
        const long long first = t.gap_start[g] + (long long)(b - t.gap_bstart[g]) * (EW_BLOCK * 4) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long i = first + u * EW_BLOCK;
            if (i < t.gap_end[g]) adam_elem(a, i, inv_bc2s, step);
        }
    }
}
// -> 1 if the run (adam, items[0..n)) can be fused: every layer's weights inside the Adam range, ascending, input-channel slots = channels
int adam_pack_fusable(const ssdn_adam_args* a, const ssdn_wpack_args* const* items, int n) {
    if (n < 1 || n > ADAM_PACK_MAX) return 0;
    long long prev = -1;
    for (int i = 0; i < n; ++i) {
        const ssdn_wpack_args* w = items[i];
        const long long off = w->w - a->p;
        if (w->w < a->p || off + (long long)w->M * w->cin * w->ntaps > a->n || off <= prev) return 0;
        if (w->c0 + w->c1_real != w->cin || (w->ntaps != 1 && w->ntaps != 9)) return 0;
        prev = off;
    }
    return 1;
}
int launch_adam_pack(const ssdn_adam_args* a, const ssdn_wpack_args* const* items, int n, hipStream_t s) {
    AdamPackTable t;
    t.a = *a;
    t.n = n;
    int blocks = 0, ng = 0;
    long long pos = 0;
    auto gap = [&](long long start, long long end) {
        if (end <= start) return;
        t.gap_start[ng] = start; t.gap_end[ng] = end;
        ++ng;
    };
    for (int i = 0; i < n; ++i) {
        t.e[i] = *items[i];
        t.w_off[i] = items[i]->w - a->p;
        t.bstart[i] = blocks;
        t.tiles_k[i] = (items[i]->cin + AP_TK - 1) / AP_TK;
        blocks += ((items[i]->M + AP_TM - 1) / AP_TM) * t.tiles_k[i];
        gap(pos, t.w_off[i]);
        pos = t.w_off[i] + (long long)items[i]->M * items[i]->cin * items[i]->ntaps;
    }
    t.bstart[n] = blocks;
    gap(pos, a->n);
    t.ngaps = ng;
    for (int g = 0; g < ng; ++g) {
        t.gap_bstart[g] = blocks;
        blocks += (int)((t.gap_end[g] - t.gap_start[g] + EW_BLOCK * 4 - 1) / (EW_BLOCK * 4));
    }
    t.gap_bstart[ng] = blocks;
    if (blocks > 0) hipLaunchKernelGGL(k_adam_pack, dim3(blocks), dim3(EW_BLOCK), 0, s, t);
    return 0;
}
int launch_adam(const ssdn_adam_args* a, hipStream_t s) {
    int g = ew_grid(a->n);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_adam, dim3(g), dim3(EW_BLOCK), 0, s, *a);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// NOISE: the training patch stream's per-sample work for a whole minibatch (noise_wrapper.py:66-135, utils/noise.py:54-107,
// utils/n2v_ups.py:40-88).  One thread per pixel, all channels; HBM-bound: 1 byte in, 4-12 bytes out per element.
// Random numbers are counter-based (Philox4x32-10): the value of element e of stream s of launch `offset` is a pure function of
// (seed, offset, s, e), so the Noise2Void replacement can RE-DERIVE the noisy value of the neighbour it copies (no second pass).
// ------------------------------------------------------------------------------------------------
struct Ph4 { unsigned v[4]; };
static __device__ __forceinline__ Ph4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Ph4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}
static __device__ __forceinline__ float u01(unsigned x) { return (float)(x >> 8) * (1.f / 16777216.f) + (0.5f / 16777216.f); }   // (0, 1)
enum { NS_INPUT = 0, NS_REF = 1, NS_PARAM = 2, NS_COORD = 3, NS_PARAM_REF = 4 };
// noisy value of element e (channel plane index bc = b*C + c) of realisation `stream`
static __device__ __forceinline__ float noise_apply(const ssdn_noise_args& a, float clean, float param, unsigned e, unsigned stream) {
    const Ph4 r = philox4x32_10(e, stream, (unsigned)a.offset, (unsigned)(a.offset >> 32), (unsigned)a.seed, (unsigned)(a.seed >> 32));
    float v;
    if (a.style == 0) {
        const float z = sqrtf(-2.f * __logf(u01(r.v[0]))) * __cosf(6.28318530718f * u01(r.v[1]));      // Box-Muller
        v = clean + param * z;
    } else {
        // Poisson(1) by inversion: P(k) = e^-1 / k!
        const float u = u01(r.v[0]);
        float pk = 0.36787944117f, cdf = pk;
        int k = 0;
        while (u > cdf && k < 16) { ++k; pk /= (float)k; cdf += pk; }
        v = (clean * param + (float)k) / param;
    }
    if (a.clip) v = fminf(fmaxf(v, 0.f), 1.f);
    return v;
}
static __device__ __forceinline__ float noise_param(const ssdn_noise_args& a, int bc, unsigned stream) {
    if (a.p_lo == a.p_hi) return a.p_lo;
    const Ph4 r = philox4x32_10((unsigned)bc, stream, (unsigned)a.offset, (unsigned)(a.offset >> 32), (unsigned)a.seed, (unsigned)(a.seed >> 32));
    return a.p_lo + (a.p_hi - a.p_lo) * u01(r.v[0]);
}
// uniform integer over [lo, hi) without c (at least one candidate is assumed; n2v_ups.py:40-46); negative results wrap (Python indexing)
static __device__ __forceinline__ int n2v_pick(int c, int r, int size, float u) {
    const int lo = c - r < 0 ? c - r : 0, hi = c + r < size - 1 ? c + r : size - 1;
    int span = hi - lo - (c >= lo && c < hi ? 1 : 0);
    if (span < 1) span = 1;
    int k = (int)(u * (float)span);
    if (k >= span) k = span - 1;
    int v = lo + k;
    if (c >= lo && c < hi && v >= c) ++v;
    if (v < 0) v += size;
    if (v >= size) v = size - 1;
    return v;
}
__global__ void k_noise(ssdn_noise_args a) {
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int H = a.H, W = a.W, HW = H * W;
    if (idx >= (unsigned)(a.B * HW)) return;
    const int x = idx % W, y = (idx / W) % H, b = idx / HW;
    const unsigned char* u8 = (const unsigned char*)a.clean_u8;
    // Noise2Void: is (x, y) the selected pixel of its box, and if so which neighbour does it copy?
    bool blind = false;
    int rx = x, ry = y;
    if (a.n2v_box > 0) {
        const int box = a.n2v_box, n1 = H / box, i = x / box, j = y / box, cell = i * n1 + j;
        const unsigned ce = (unsigned)(b * (W / box) * n1 + cell);
        const Ph4 r = philox4x32_10(ce, NS_COORD, (unsigned)a.offset, (unsigned)(a.offset >> 32), (unsigned)a.seed, (unsigned)(a.seed >> 32));
        int c0 = i * box + (int)(u01(r.v[0]) * (float)box), c1 = j * box + (int)(u01(r.v[1]) * (float)box);
        c0 = c0 > i * box + box - 1 ? i * box + box - 1 : c0;
        c1 = c1 > j * box + box - 1 ? j * box + box - 1 : c1;
        if (c0 == x && c1 == y) {
            blind = true;
            rx = n2v_pick(c0, a.n2v_radius, W, u01(r.v[2]));
            ry = n2v_pick(c1, a.n2v_radius, H, u01(r.v[3]));
            a.coords[((long long)b * (W / box) * n1 + cell) * 2 + 0] = c0;
            a.coords[((long long)b * (W / box) * n1 + cell) * 2 + 1] = c1;
        }
    }
    for (int c = 0; c < a.C; ++c) {
        const int bc = b * a.C + c;
        const unsigned e = (unsigned)(bc * HW + y * W + x);
        const float clean = (float)u8[e] / 255.f;                 // (a true division, like to_tensor's: x * (1/255) differs in the last bit)
        const float param = noise_param(a, bc, NS_PARAM);
        float v;
        if (blind) {
            const unsigned en = (unsigned)(bc * HW + ry * W + rx);
            v = noise_apply(a, (float)u8[en] / 255.f, param, en, NS_INPUT);
        } else {
            v = noise_apply(a, clean, param, e, NS_INPUT);
        }
        a.noisy32[e] = v;
        if (a.clean32) a.clean32[e] = clean;
        if (a.ref32) {
            const float pr = noise_param(a, bc, NS_PARAM_REF);
            a.ref32[e] = noise_apply(a, clean, pr, e, NS_REF);
            if (a.param_ref && x == 0 && y == 0) a.param_ref[bc] = pr;
        }
        if (a.param && x == 0 && y == 0) a.param[bc] = param;
    }
}
int launch_noise(const ssdn_noise_args* a, hipStream_t s) {
    if (!a->clean_u8 || !a->noisy32) return ssdn_set_error("noise: clean_u8 and noisy32 are required");
    if (a->B < 1 || a->C < 1 || a->H < 1 || a->W < 1) return ssdn_set_error("noise: empty shape");
    const long long n = (long long)a->B * a->C * a->H * a->W;
    if (n >= (1ll << 31)) return ssdn_set_error("noise: too many elements for 32-bit indexing");
    if (a->style != 0 && a->style != 1) return ssdn_set_error("noise: style must be 0 (gauss) or 1 (poisson)");
    if (a->style == 1 && !(a->p_lo > 0.f)) return ssdn_set_error("noise: poisson lambda must be > 0");
    if (a->p_hi < a->p_lo) return ssdn_set_error("noise: p_hi < p_lo");
    if (a->n2v_box > 0) {
        if (!a->coords) return ssdn_set_error("noise: Noise2Void manipulation needs a coords output");
        if (a->H % a->n2v_box || a->W % a->n2v_box) return ssdn_set_error("noise: H and W must be multiples of n2v_box");
        if (a->n2v_radius < 1 || a->W < 2 || a->H < 2) return ssdn_set_error("noise: n2v_radius must be >= 1 and the patch larger than 1 pixel");
    }
    hipLaunchKernelGGL(k_noise, dim3(ew_grid((long long)a->B * a->H * a->W)), dim3(EW_BLOCK), 0, s, *a);
    return 0;
}

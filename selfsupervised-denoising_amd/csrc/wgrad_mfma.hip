// wgrad_mfma.hip -- convolution weight/bias gradient on the CDNA4 matrix cores.
//
// dW[t][m][k] = sum_pixels dZ[pixel][m] * IN[pixel + off_t][k]   is a GEMM whose REDUCTION dimension is the pixel axis,
// while both operands live in HBM/LDS as NHWC (channels contiguous, pixels strided).  The MFMA wants 8 consecutive
// reduction elements per lane, i.e. 8 pixels of ONE channel: exactly the transposed access gfx950's
// ds_read_b64_tr_b16 provides -- the tiles are staged in LDS untransposed (coalesced 16-B NHWC loads, same loader as
// the forward kernel) and both MFMA operands are fetched with the LDS transpose-read.  No transposed copy of any
// activation or gradient is ever written to HBM.
//
// Work split: workgroup = 512 threads = 8 waves, persistent over its share of 256-pixel tiles.  For one launch the
// output is [ntaps][Mpad<=96][Kpad<=96] (+ a bias column): ntaps*Kpad/32 (+1) column tiles of 32, dealt round-robin to
// the 8 waves (<= CPW per wave), each wave keeping MT x CPW 32x32 fp32 accumulators in registers for the whole pixel
// range and reading the dZ operand (A) once per K-step for all its column tiles.  At the end every workgroup writes its
// accumulators to its own fp32 slab (plain coalesced stores, no atomics) and SSDN_OP_WREDUCE sums the slabs in a fixed
// order => bit-reproducible gradients.
#include "common.h"

#define WG_THREADS 512

typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 fp16x4_t;
typedef __attribute__((address_space(3))) fp16x4_t lds_fp16x4;

// ds_read_b64_tr_b16: inside every 16-lane group, lane i supplies the address of 4 contiguous halves = row (i>>2),
// column chunk (i&3) of a 4x16 matrix; lane i receives column i (4 rows).  (Verified on the device by
// tests/test_hip_probe.py::test_tr16_mapping through ssdn_probe_tr16.)
static __device__ __forceinline__ half4 tr16(const char* lds_addr) {
    fp16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)lds_addr);
    return __builtin_bit_cast(half4, r);
}
static __device__ __forceinline__ half8 cat8(half4 lo, half4 hi) {
    half8 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = lo[i]; v[4 + i] = hi[i]; }
    return v;
}

struct WgGeom {
    int TW, TH, TN, HH, HW, padT, padL, NP, PSTR, DSTR;
    int tiles_x, tiles_y, groups_n, ntiles;
};
static __host__ __device__ inline WgGeom wg_geom(const ssdn_wgrad_args& a) {
    WgGeom g;
    g.TW = 1 << a.ltw; g.TH = 1 << a.lth; g.TN = 1 << a.ltn;
    int mny = 0, mxy = 0, mnx = 0, mxx = 0;
    for (int t = 0; t < a.ntaps; ++t) {
        mny = a.dy[t] < mny ? a.dy[t] : mny; mxy = a.dy[t] > mxy ? a.dy[t] : mxy;
        mnx = a.dx[t] < mnx ? a.dx[t] : mnx; mxx = a.dx[t] > mxx ? a.dx[t] : mxx;
    }
    g.padT = -mny; g.padL = -mnx;
    g.HH = g.TH - mny + mxy;
    g.HW = g.TW - mnx + mxx;
    g.NP = g.TN * g.HH * g.HW;
    int kmax = a.Ktot;          // channels a transpose-read may touch: staged ones + the (harmless, never used) padding
    for (int t = 0; t < a.ntaps; ++t) kmax = a.coff[t] + a.Kpad > kmax ? a.coff[t] + a.Kpad : kmax;
    g.PSTR = kmax * 2 + 16;
    g.DSTR = a.Mpad * 2 + 16;
    g.tiles_x = (a.W + g.TW - 1) / g.TW;
    g.tiles_y = (a.H + g.TH - 1) / g.TH;
    g.groups_n = (a.N + g.TN - 1) / g.TN;
    g.ntiles = g.tiles_x * g.tiles_y * g.groups_n;
    return g;
}

static __device__ __forceinline__ unsigned fdivw(unsigned x, unsigned magic) { return magic ? __umulhi(x, magic) : x; }
static inline unsigned magic_ofw(unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }

struct WgAux {
    unsigned mg_hw, mg_hh;
    int ccx, ccd;             // 16-B pieces per pixel of the input tile / of the dZ tile
    unsigned mg_ccx, mg_ccd;  // their magic reciprocals
};

template <int MT, int CPW>
__global__ __launch_bounds__(WG_THREADS) void k_wgrad(ssdn_wgrad_args a, WgAux x) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const WgGeom g = wg_geom(a);
    char* xt = smem;                           // input halo tile  [NP][PSTR]
    char* dt = smem + (size_t)g.NP * g.PSTR;   // dZ tile          [256][DSTR]

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, mh = (lane >> 4) & 1;
    const int NTt = a.Kpad >> 5;
    const int CT = a.ntaps * NTt;  // column tiles; tile index CT = the bias column

    // column tiles of this wave
    int ct_tap[CPW], ct_nt[CPW];
    bool ct_on[CPW], ct_bias[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        int ct = wave + 8 * j;
        ct_on[j] = ct <= CT;
        ct_bias[j] = ct == CT;
        int tp = ct_on[j] && !ct_bias[j] ? ct / NTt : 0;
        ct_tap[j] = tp;
        ct_nt[j] = ct_on[j] && !ct_bias[j] ? ct - tp * NTt : 0;
    }

    f32x16 acc[MT][CPW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < CPW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][j][r] = 0.f;

    half8 ones;   // bf16 1.0 = 0x3f80 in every 16-bit slot
    {
        u16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = 0x3f80;
        ones = __builtin_bit_cast(half8, o);
    }

    const int H0 = a.up0 ? (a.H >> 1) : a.H, W0 = a.up0 ? (a.W >> 1) : a.W;
    const h16* s0 = (const h16*)a.src0.p;
    const h16* s1 = (const h16*)a.src1.p;
    const h16* dzp = (const h16*)a.dz.p;
    const int npix_tile = g.TN * g.TH * g.TW;

    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
        int bid = tile;
        const int tx_i = bid % g.tiles_x; bid /= g.tiles_x;
        const int ty_i = bid % g.tiles_y; bid /= g.tiles_y;
        const int n0 = bid * g.TN, y0 = ty_i * g.TH, x0 = tx_i * g.TW;
        __syncthreads();  // previous tile fully consumed
        {   // ---- stage input halo tile: flat index over (halo pixel, 16-B piece), 4 independent loads in flight per thread
            //      (a load -> convert -> store chain per iteration made the staging a sequence of full memory round trips) ----
            const int nflat = g.NP * x.ccx;
            for (int f0 = tid; f0 < nflat; f0 += 4 * WG_THREADS) {
                half8 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int f = f0 + u * WG_THREADS;
                    const int hp = fdivw(f, x.mg_ccx), cc = f - hp * x.ccx;
                    const unsigned r1 = fdivw(hp, x.mg_hw);
                    const int hx = hp - r1 * g.HW;
                    const unsigned tn = fdivw(r1, x.mg_hh);
                    const int hy = r1 - tn * g.HH;
                    const int n = n0 + tn, y = y0 - g.padT + hy, xx = x0 - g.padL + hx;
                    const int k = cc * 8;
                    v[u] = zero_h8();
                    if (f < nflat && n < a.N && (unsigned)y < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) {
                        if (k < a.c0) {
                            const int sh = a.up0;
                            v[u] = ld_h8(s0 + (long long)(((n * H0 + (y >> sh)) * W0 + (xx >> sh)) * a.src0.cs + a.src0.co + k));
                        } else {
                            v[u] = ld_h8(s1 + (long long)(((n * a.H + y) * a.W + xx) * a.src1.cs + a.src1.co + (k - a.c0)));
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int f = f0 + u * WG_THREADS;
                    if (f < nflat) {
                        const int hp = fdivw(f, x.mg_ccx), cc = f - hp * x.ccx;
                        // fp16 activation -> bf16 once, while staging (the gradient operand is bf16; MFMA needs one type)
                        u16x8 vb;
#pragma unroll
                        for (int e = 0; e < 8; ++e) vb[e] = f2bf((float)v[u][e]);
                        *reinterpret_cast<u16x8*>(xt + (size_t)hp * g.PSTR + cc * 16) = vb;
                    }
                }
            }
        }
        {   // ---- stage dZ tile (zero outside the image so overhanging pixels contribute nothing) ----
            const int nflat = npix_tile * x.ccd;
            for (int f0 = tid; f0 < nflat; f0 += 4 * WG_THREADS) {
                half8 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int f = f0 + u * WG_THREADS;
                    const int q = fdivw(f, x.mg_ccd), cc = f - q * x.ccd;
                    const int tx = q & (g.TW - 1), ty = (q >> a.ltw) & (g.TH - 1), tn = q >> (a.ltw + a.lth);
                    const int n = n0 + tn, y = y0 + ty, xx = x0 + tx;
                    v[u] = zero_h8();
                    if (f < nflat && n < a.N && y < a.H && xx < a.W)
                        v[u] = ld_h8(dzp + (long long)(((n * a.H + y) * a.W + xx) * a.dz.cs + a.dz.co + cc * 8));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int f = f0 + u * WG_THREADS;
                    if (f < nflat) {
                        const int q = fdivw(f, x.mg_ccd), cc = f - q * x.ccd;
                        *reinterpret_cast<half8*>(dt + (size_t)q * g.DSTR + cc * 16) = v[u];
                    }
                }
            }
        }
        __syncthreads();
        // ---- K loop over the tile's pixels, 16 per step ----
        for (int q0 = 0; q0 < npix_tile; q0 += 16) {
            // the two pixels this lane addresses for the transpose reads (r = 0,1 -> k elements 0..3 / 4..7)
            int dofs[2], xofs[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                int q = q0 + kh * 8 + r * 4 + (i16 >> 2);
                if (q >= npix_tile) q = 0;  // tiles with < 16 pixels per step cannot occur (npix_tile is a multiple of 16 or handled by host)
                int tx = q & (g.TW - 1), ty = (q >> a.ltw) & (g.TH - 1), tn = q >> (a.ltw + a.lth);
                dofs[r] = q * g.DSTR + (i16 & 3) * 8;
                xofs[r] = ((tn * g.HH + ty + g.padT) * g.HW + tx + g.padL) * g.PSTR + (i16 & 3) * 8;
            }
            half8 af[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int cofs = (mt * 32 + mh * 16) * 2;
                af[mt] = cat8(tr16(dt + dofs[0] + cofs), tr16(dt + dofs[1] + cofs));
            }
#pragma unroll
            for (int j = 0; j < CPW; ++j) {
                if (!ct_on[j]) continue;
                half8 bf;
                if (ct_bias[j]) {
                    bf = ones;
                } else {
                    const int toff = (a.dy[ct_tap[j]] * g.HW + a.dx[ct_tap[j]]) * g.PSTR + (a.coff[ct_tap[j]] + ct_nt[j] * 32 + mh * 16) * 2;
                    bf = cat8(tr16(xt + xofs[0] + toff), tr16(xt + xofs[1] + toff));
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[mt]), __builtin_bit_cast(bf16x8, bf),
                                                                         acc[mt][j], 0, 0, 0);
            }
        }
    }

    // ---- write this workgroup's slab: D row = m (8*(r>>2) + 4*kh + (r&3)), D col = k (l31) ----
    float* slab = a.slab + (long long)blockIdx.x * a.ntaps * a.Mpad * a.Kpad;
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        if (!ct_on[j]) continue;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = mt * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
                if (ct_bias[j]) {
                    if (l31 == 0) a.bslab[(long long)blockIdx.x * a.Mpad + m] = acc[mt][j][r];
                } else {
                    slab[((long long)ct_tap[j] * a.Mpad + m) * a.Kpad + ct_nt[j] * 32 + l31] = acc[mt][j][r];
                }
            }
        }
    }
}

static int wgrad_validate(const ssdn_wgrad_args* a) {
    if (a->ntaps < 1 || a->ntaps > SSDN_MAX_TAPS) return ssdn_set_error("wgrad: ntaps out of range");
    if (a->ltw + a->lth + a->ltn > 8 || a->ltw + a->lth + a->ltn < 4) return ssdn_set_error("wgrad: tile must have 16..256 pixels");
    if (a->Ktot != a->c0 + a->c1 || (a->Ktot & 7)) return ssdn_set_error("wgrad: Ktot must equal c0+c1 (multiple of 8)");
    if ((a->c0 & 7) || (a->c1 & 7)) return ssdn_set_error("wgrad: source channel counts must be multiples of 8");
    if ((a->Kpad & 31) || a->Kpad > 96) return ssdn_set_error("wgrad: Kpad must be 32/64/96");
    for (int t = 0; t < a->ntaps; ++t)
        if (a->coff[t] < 0 || (a->coff[t] & 15) || a->coff[t] >= a->Ktot) return ssdn_set_error("wgrad: bad channel offset of tap %d", t);
    if ((a->Mpad & 31) || a->Mpad > 96 || a->M > a->Mpad || (a->M & 7)) return ssdn_set_error("wgrad: Mpad must be 32/64/96, M %% 8 == 0");
    if (a->nslabs < 1) return ssdn_set_error("wgrad: nslabs < 1");
    return 0;
}
int wgrad_lds_bytes(const ssdn_wgrad_args* a) {
    if (wgrad_validate(a)) return -1;
    WgGeom g = wg_geom(*a);
    return g.NP * g.PSTR + (g.TN * g.TH * g.TW) * g.DSTR + 64;
}

template <int MT, int CPW>
static int wgrad_launch(const ssdn_wgrad_args* a, const WgGeom& g, const WgAux& x, hipStream_t s) {
    size_t lds = (size_t)g.NP * g.PSTR + (size_t)(g.TN * g.TH * g.TW) * g.DSTR + 64;
    if (lds > 160 * 1024) return ssdn_set_error("wgrad: tiling needs %zu B of LDS (> 160 KiB)", lds);
    static bool attr_set = false;
    if (!attr_set) {
        SSDN_CHECK_HIP(hipFuncSetAttribute((const void*)k_wgrad<MT, CPW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    double px = (double)a->N * a->H * a->W;
    prof_begin(SSDN_PROF_WGRAD, s);
    hipLaunchKernelGGL((k_wgrad<MT, CPW>), dim3(a->nslabs), dim3(WG_THREADS), lds, s, *a, x);
    prof_end(SSDN_PROF_WGRAD, s, 2.0 * px * a->M * a->Ktot * a->ntaps, px * 2.0 * (a->M + a->Ktot));
    return 0;
}

int launch_wgrad(const ssdn_wgrad_args* a, hipStream_t s) {
    int rc = wgrad_validate(a);
    if (rc) return rc;
    WgGeom g = wg_geom(*a);
    WgAux x;
    x.mg_hw = magic_ofw(g.HW);
    x.mg_hh = magic_ofw(g.HH);
    x.ccx = a->Ktot / 8;
    x.ccd = a->M / 8;
    x.mg_ccx = magic_ofw(x.ccx);
    x.mg_ccd = magic_ofw(x.ccd);
    const int MT = a->Mpad / 32;
    const int CT = a->ntaps * (a->Kpad / 32) + 1;
    const int CPW = (CT + 7) / 8;
#define WG_CASE(mt, cpw) if (MT == mt && CPW == cpw) rc = wgrad_launch<mt, cpw>(a, g, x, s); else
    WG_CASE(1, 1) WG_CASE(1, 2) WG_CASE(1, 3) WG_CASE(1, 4)
    WG_CASE(2, 1) WG_CASE(2, 2) WG_CASE(2, 3) WG_CASE(2, 4)
    WG_CASE(3, 1) WG_CASE(3, 2) WG_CASE(3, 3) WG_CASE(3, 4)
    rc = ssdn_set_error("wgrad: unsupported shape MT=%d CPW=%d", MT, CPW);
#undef WG_CASE
    if (rc) return rc;
    SSDN_CHECK_HIP(hipGetLastError());
    return 0;
}

// wgrad_body.h -- device bodies and host-side planning helpers of the weight-gradient kernels, shared by wgrad_mfma.hip (one
// launch per op, merged small-layer launches) and wgrad_mega.hip (the chip-wide launch).  See wgrad_mfma.hip for the design.
#pragma once
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>
#include <type_traits>

template <int I, int N, class F>
static __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

#define WG_THREADS 256
#define WG_WAVES 4

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
    int XB, DB;   // bytes of the input / dZ part of one LDS image (each with one extra dummy row that absorbs void row items)
    int tiles_x, tiles_y, groups_n, ntiles;
};
static __host__ __device__ constexpr int wg_stride(int row_bytes) { return ((row_bytes + 63) & ~127) + 64; }
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
    // pixel strides of the LDS images: == 64 (mod 128) bytes, so that the 4 pixels x 64 bytes a 32-lane half of a transpose
    // read touches fall on 4 disjoint groups of 16 banks (a stride of row-bytes + 16 was 2-way bank conflicted)
    g.PSTR = wg_stride(kmax * 2);
    g.DSTR = wg_stride(a.Mpad * 2);
    g.XB = (g.NP + g.HW) * g.PSTR;
    g.DB = (g.TN * g.TH * g.TW + g.TW) * g.DSTR;
    g.tiles_x = (a.W + g.TW - 1) / g.TW;
    g.tiles_y = (a.H + g.TH - 1) / g.TH;
    g.groups_n = (a.N + g.TN - 1) / g.TN;
    g.ntiles = g.tiles_x * g.tiles_y * g.groups_n;
    return g;
}

// x / d for d >= 1 with magic = ceil(2^32 / d) (0 when d == 1), branch-free: umulhi(x, 0) + x when d == 1
static __device__ __forceinline__ unsigned fdivw(unsigned x, unsigned magic) {
    return __umulhi(x, magic) + (x & (unsigned)-(int)(magic == 0));
}
// 8 fp16 -> 8 bf16 (v_cvt_f32_f16 x2 + v_cvt_pk_bf16_f32 per pair, round-to-nearest-even)
static __device__ __forceinline__ half8 cvt_h8_to_bf8(half8 v) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x2_t f = {(float)v[2 * i], (float)v[2 * i + 1]};
        w[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
    }
    u32x4_t r = {w[0], w[1], w[2], w[3]};
    return __builtin_bit_cast(half8, r);
}
static __device__ __forceinline__ half8 mask_h8(half8 v, bool keep) {
    u32x4_t r = __builtin_bit_cast(u32x4_t, v);
    const unsigned m = (unsigned)-(int)keep;
    r[0] &= m; r[1] &= m; r[2] &= m; r[3] &= m;
    return __builtin_bit_cast(half8, r);
}
static inline unsigned magic_ofw(unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }

extern "C" void* ssdn_debug_get_trace();
struct WgAux {
    unsigned long long* trace;
    int ccx, ccd;             // 16-B pieces per pixel of the input tile / of the dZ tile
    unsigned mg_ccx, mg_ccd;  // their magic reciprocals (per-lane: piece of a row -> pixel, chunk)
    unsigned mg_hh;           // scalar: halo row -> (image of the tile, halo y)
    int rswx, rswd;           // rows of the input halo image / of the dZ image each wave stages per tile
    int sync;                 // 1: the tiles cannot be prefetched within their K-steps (images of 4x4 pixels and below: many short rows):
                              // every tile is fetched synchronously into image 0 before its K-steps -- a workgroup may still own several
};

#define WG_ND 3                             // 64-lane loads per dZ row (<= 16 pixels x 12 chunks)
#define WG_ONES_BYTES 4096                  // LDS area of bf16 1.0 behind the two images: B operand of the bias column

// global -> LDS without registers (`buffer_load_dwordx4 ... lds`; see csrc/conv_dma.hip::dma16): M0 = LDS byte address of lane 0's 16
// bytes, lane i lands at +16 i, lanes whose offset is outside the resource's num_records write zeros
static __device__ __forceinline__ void wg_dma16(unsigned lds_addr, int voff, u32x4_t rs, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// MFMA with the accumulator pinned to a register file.  A wave of this kernel owns up to 21 32x32 fp32 accumulators = 336
// registers, more than the 256 AGPRs: the first 16 tiles live in AGPRs, the rest in VGPRs (the compiler will not split
// them itself -- it spills instead).  The compiler does not know these asm statements are MFMAs, so it inserts no hazard
// NOPs: the operands are only ever written by LDS reads (s_waitcnt is tracked per register, asm or not), an accumulator is
// re-used every MT*CPW >= 18 MFMAs, and the epilogue reads the accumulators after a barrier and explicit NOPs.
template <int FILE>   // 0: compiler's choice (builtin), 1: AGPR, 2: VGPR
static __device__ __forceinline__ void mma_bf16(f32x16& c, half8 av, half8 bv) {
    if constexpr (FILE == 0) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c, 0, 0, 0);
    } else if constexpr (FILE == 1) {
        asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv));
    } else {
        asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
    }
}

// One wave per SIMD (256-thread workgroup, 4 waves): each wave may use the full 512-entry unified VGPR/AGPR file, which is
// what holding up to MT*CPW = 21 accumulators PLUS double-buffered operand fragments PLUS the in-flight global prefetch
// takes.  With a single wave per SIMD nothing hides instruction ISSUE (measured: ~10 cycles per instruction), so the loop is
// written for instruction count: a K-step is 21 MFMAs + 20 LDS transpose reads + ~40 instructions of staging.
//
// Staging: a ROW ITEM is one image row of the tile (input halo row or dZ row), loaded by NL (resp. 3) 64-lane
// buffer_load_dwordx4.  Everything about the row (image, y, validity, base address) is wave-uniform and computed on the
// scalar unit into a buffer resource whose num_records is the row length, so the hardware bounds check returns zeros for the
// left/right halo (negative or too large x offset) and for rows above/below the image (num_records = 0); the per-lane part
// (pixel-in-row, channel chunk) -> global byte offset and LDS byte offset is computed ONCE per kernel.  The rows of a tile are
// dealt to the 4 waves; a wave issues one row item per K-step (BOTH: one input and one dZ row) and writes it to the other
// LDS image two K-steps later.
// (the body is a device function of the launch coordinates (bx, by, gdx) = (blockIdx.x, blockIdx.y, gridDim.x) so that
//  k_wgrad_multi can run the workgroups of SEVERAL layers' weight-gradient GEMMs inside one launch)
template <int MT, int CPW, int NL, bool BOTH, int PS, int KS, int RWX, int RWD>
static __device__ __forceinline__ void wgrad_body(const ssdn_wgrad_args& a, const WgAux& x, const unsigned bx, const unsigned by, const unsigned gdx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool SPLIT = MT * CPW > 16;         // accumulators do not fit the AGPR file
    const WgGeom g = wg_geom(a);

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, mh = (lane >> 4) & 1;
    // Workgroup -> (pixel partition = slab index, block of M output channels).  Workgroups go to the 8 XCDs round-robin by
    // id and every XCD has its own L2: the mblocks workgroups of one pixel partition get ids 8 apart, i.e. the same XCD,
    // dispatched back to back -- they stream the same tiles in lock-step and share them through L2.
    int wg_slab = bx, wg_nslabs = gdx, wg_mb = 0;
    if (a.mblocks > 1) {
        const unsigned j = bx >> 3;
        wg_mb = (int)(j % (unsigned)a.mblocks);
        wg_slab = (int)(j / (unsigned)a.mblocks) * 8 + (int)(bx & 7);
        wg_nslabs = a.nslabs;
        if (wg_slab >= a.nslabs) return;      // (grid is rounded up to a multiple of 8 partitions)
    }
    const int NTt = a.Kpad >> 5;
    const int CT = a.ntaps * NTt;  // column tiles; tile index CT = the bias column

    // Column tiles: 0 = the bias column (B operand = the constant 1), 1 .. CT = (tap, 32-channel block) weight columns.  A
    // workgroup owns the 4*CPW consecutive tiles of group blockIdx.y (csplit groups; one group = everything), dealt to its
    // waves as ct = group base + wave + 4 j.  With the bias column FIRST it can only ever be (group 0, wave 0, j = 0), for any
    // number of groups; tiles past CT are absent (computed like tap 0, never written).
    int ct_tap[CPW], ct_nt[CPW];
    bool ct_on[CPW], ct_bias[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        const int ct = by * (WG_WAVES * CPW) + wave + WG_WAVES * j;
        ct_on[j] = ct <= CT;
        ct_bias[j] = ct == 0;
        const int wt = ct_on[j] && !ct_bias[j] ? ct - 1 : 0;
        ct_tap[j] = wt / NTt;
        ct_nt[j] = wt - ct_tap[j] * NTt;
    }
    int stoff[CPW];   // wave-uniform LDS byte offset of the column tile's (tap, 32-channel block) relative to a halo pixel
#pragma unroll
    for (int j = 0; j < CPW; ++j)
        stoff[j] = (a.dy[ct_tap[j]] * g.HW + a.dx[ct_tap[j]]) * g.PSTR + (a.coff[ct_tap[j]] + ct_nt[j] * 32) * 2;
    const bool first_bias = ct_bias[0];

    // the input of one launch comes from ONE tensor (src0, optionally read through the 2x nearest upsampling, or src1)
    const bool use0 = a.c0 > 0;
    const h16* sp = (const h16*)(use0 ? a.src0.p : a.src1.p);
    const int scs = use0 ? a.src0.cs : a.src1.cs, sco = use0 ? a.src0.co : a.src1.co, sh = use0 ? a.up0 : 0;
    const int Hs = a.H >> sh, Ws = a.W >> sh;
    const h16* dzp = (const h16*)a.dz.p + wg_mb * a.M;     // (this block's channels of the dz view)
    const int npix_tile = g.TN * g.TH * g.TW;
    const int RX = g.TN * g.HH, RD = g.TN * g.TH;         // rows of the input halo image / of the dZ image
    const int rowx = g.HW * x.ccx, rowd = g.TW * x.ccd;   // 16-B pieces per row

    // per-lane constants of the row loads: global byte offset relative to (row start + tile x origin) and LDS byte offset
    // relative to the row's first pixel.  Lanes past the end of the row repeat the load and the LDS write of an earlier piece
    // of the same row (same address, same data).
    int relx[NL], lox[NL], reld[WG_ND], lod[WG_ND];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        int j = lane + 64 * i;
        j = j < rowx ? j : j % rowx;
        const int hx = fdivw(j, x.mg_ccx), cc = j - hx * x.ccx;
        relx[i] = ((((hx - g.padL) >> sh) * scs) + sco + cc * 8) * 2;
        lox[i] = hx * g.PSTR + cc * 16;
    }
#pragma unroll
    for (int i = 0; i < WG_ND; ++i) {
        int j = lane + 64 * i;
        j = j < rowd ? j : j % rowd;
        const int tx = fdivw(j, x.mg_ccd), cc = j - tx * x.ccd;
        reld[i] = (tx * a.dz.cs + a.dz.co + cc * 8) * 2;
        lod[i] = tx * g.DSTR + cc * 16;
    }

    // dZ rows by LDS-DMA (static schedules): a row of the LDS image is TW pixels x DSTR bytes = TW * DSTR / 16 pieces, contiguous,
    // fetched by WG_ND 64-lane instructions; piece q = lane + 64 i is (pixel q / ppx, 16-byte piece q % ppx), pieces past the real
    // channels (and past the row) get an offset outside the resource: zeros
    int dvoff[WG_ND];
    {
        const int ppx = g.DSTR >> 4;
#pragma unroll
        for (int i = 0; i < WG_ND; ++i) {
            const int q = lane + 64 * i;
            const int tx = q / ppx, cc = q - tx * ppx;
            dvoff[i] = (tx < g.TW && cc < x.ccd) ? (tx * a.dz.cs + a.dz.co + cc * 8) * 2 : 0x40000000;
        }
    }
    const unsigned lds0 = (unsigned)(size_t)smem;
#ifdef SSDN_TUNING
    int tr_i = 0;
    auto stamp = [&]() {
        if (x.trace && tid == 0 && tr_i < 32) x.trace[(size_t)bx * 32 + tr_i++] = __builtin_amdgcn_s_memtime();
    };
#else
    auto stamp = []() {};
#endif
    stamp();
    // ---- double-buffered pipeline over this workgroup's tiles -------------------------------------------------------
    const int bufsz = g.XB + g.DB;
    {   // the bias column's B operand: an LDS area of bf16 1.0 (0x3f80) that its transpose reads are pointed at
        u16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = 0x3f80;
        *reinterpret_cast<u16x8*>(smem + 2 * bufsz + tid * 16) = o;
    }
    const int ksteps = npix_tile >> 4;
    const int ntl = ((int)g.ntiles - wg_slab + wg_nslabs - 1) / wg_nslabs;

    auto origin = [&](int i, int& n0, int& y0, int& x0) __attribute__((always_inline)) {
        int bid = wg_slab + i * wg_nslabs;
        const int tx_i = bid % g.tiles_x; bid /= g.tiles_x;
        const int ty_i = bid % g.tiles_y; bid /= g.tiles_y;
        n0 = bid * g.TN; y0 = ty_i * g.TH; x0 = tx_i * g.TW;
    };
    // ---- row items ---------------------------------------------------------------------------------------------------
    // scalar description of one row: buffer base / length, x origin byte offset, LDS byte offset of the row (-1: no item)
    struct Row { const h16* base; int num, xs, lds; };
    auto row_x = [&](int rs, bool live, int n0, int y0, int x0) __attribute__((always_inline)) -> Row {
        const int row = wave + WG_WAVES * rs;
        const int tn = fdivw(row, x.mg_hh), hy = row - tn * g.HH;
        const int n = n0 + tn, y = y0 - g.padT + hy;
        const bool item = live && rs < x.rswx && row < RX;
        const bool rowok = item && n < a.N && (unsigned)y < (unsigned)a.H;
        Row r;
        r.base = sp + (long long)((n * Hs + (y >> sh)) * Ws) * scs;
        r.num = rowok ? Ws * scs * 2 : 0;
        r.xs = ((x0 >> sh) * scs) * 2;
        r.lds = item ? row * g.HW * g.PSTR : -1;
        return r;
    };
    auto row_d = [&](int rs, bool live, int n0, int y0, int x0) __attribute__((always_inline)) -> Row {
        const int row = wave + WG_WAVES * rs;
        const int tn = row >> a.lth, ty = row & (g.TH - 1);
        const int n = n0 + tn, y = y0 + ty;
        const bool item = live && rs < x.rswd && row < RD;
        const bool rowok = item && n < a.N && y < a.H;
        Row r;
        r.base = dzp + (long long)((n * a.H + y) * a.W) * a.dz.cs;
        r.num = rowok ? a.W * a.dz.cs * 2 : 0;
        r.xs = x0 * a.dz.cs * 2;
        r.lds = item ? g.XB + row * g.TW * g.DSTR : -1;
        return r;
    };
    auto load16 = [&](const Row& r, int rel) __attribute__((always_inline)) -> half8 {
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)r.base, 0, r.num, SSDN_BUFFER_RSRC_FLAGS);
        return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, rel + r.xs, 0, 0));
    };
    // fp16 activation -> bf16 once, while staging (the gradient operand is bf16; MFMA needs one type)
    auto put_x = [&](char* img, int rowlds, int i, half8 v) __attribute__((always_inline)) {
        *reinterpret_cast<half8*>(img + rowlds + lox[i]) = cvt_h8_to_bf8(v);
    };
    auto put_d = [&](char* img, int rowlds, int i, half8 v) __attribute__((always_inline)) {
        *reinterpret_cast<half8*>(img + rowlds + lod[i]) = v;
    };

    // prefetch register sets by K-step parity (prefetch distance 2 K-steps).  !BOTH: one row item per K-step, input rows
    // first, then dZ rows, sharing the registers; BOTH: an input row and a dZ row per K-step.
    constexpr int NPV = BOTH ? NL + WG_ND : (NL > WG_ND ? NL : WG_ND);
    const int nit = BOTH ? (x.rswx > x.rswd ? x.rswx : x.rswd) : x.rswx + x.rswd;   // row items per wave per tile

    // synchronous fetch of tile ti into image 0, PB row items in flight (the first tile of every workgroup; every tile in sync mode)
    auto sync_load = [&](int ti, auto PBc) __attribute__((always_inline)) {
        constexpr int PB = decltype(PBc)::value;
        int n0, y0, x0;
        origin(ti, n0, y0, x0);
        for (int r0 = 0; r0 < nit; r0 += PB) {
            half8 tv[PB][NPV];
            int tl[PB], td[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int r = r0 + u;
                tl[u] = td[u] = -1;
                if (BOTH || r < x.rswx) {
                    const Row rw = row_x(r, r < nit, n0, y0, x0);
                    tl[u] = rw.lds;
#pragma unroll
                    for (int i = 0; i < NL; ++i) tv[u][i] = load16(rw, relx[i]);
                }
                if (BOTH || r >= x.rswx) {
                    const Row rw = row_d(BOTH ? r : r - x.rswx, r < nit, n0, y0, x0);
                    td[u] = rw.lds;
#pragma unroll
                    for (int i = 0; i < WG_ND; ++i) tv[u][(BOTH ? NL : 0) + i] = load16(rw, reld[i]);
                }
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                if (tl[u] >= 0) {
#pragma unroll
                    for (int i = 0; i < NL; ++i) put_x(smem, tl[u], i, tv[u][i]);
                }
                if (td[u] >= 0) {
#pragma unroll
                    for (int i = 0; i < WG_ND; ++i) put_d(smem, td[u], i, tv[u][(BOTH ? NL : 0) + i]);
                }
            }
        }
    };
    if (ntl > 0) sync_load(0, std::integral_constant<int, (BOTH ? 3 : 6)>{});
    __syncthreads();
    stamp();

    f32x16 acc[MT][CPW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < CPW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][j][r] = 0.f;

    half8 pvA[NPV], pvB[NPV], pvC[KS > 0 ? NPV : 1];
    int rlA = -1, rlB = -1, rdA = -1, rdB = -1, rlC = -1, rdC = -1;   // LDS row offsets of the items in flight (-1: none); rd*: dZ item
#pragma unroll
    for (int u = 0; u < NPV; ++u) pvA[u] = pvB[u] = zero_h8();
    pvC[0] = zero_h8();

    // operand fragments: A (dZ, MT row tiles) double-buffered by K-step parity, B (input, CPW column tiles) refilled in place
    // right after the MFMAs that consumed it -- every LDS transpose read is issued one full K-step before its use.
    half8 afA[MT], afB[MT], bf[CPW];
    // per-lane part of the fragment addresses: the two pixels (of the 16 of a K-step) this lane supplies to the transpose
    // reads (r = 0,1 -> k elements 0..3 / 4..7); the K-step's first pixel adds a wave-uniform offset
    // (a K-step's 16 pixels are whole rows of one image, or whole images: tiles are <= 16 wide and all sizes powers of 2)
    int dlane[2], xlane[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int e = kh * 8 + r * 4 + (i16 >> 2);
        const int tx = e & (g.TW - 1), tyl = (e >> a.ltw) & (g.TH - 1), tnl = e >> (a.ltw + a.lth);
        dlane[r] = e * g.DSTR + (i16 & 3) * 8;
        xlane[r] = ((tnl * g.HH + tyl) * g.HW + tx + g.padL) * g.PSTR + (i16 & 3) * 8;
    }

    int xl_first[2];   // per-lane part of the first column tile's fragment address (the bias column reads the area of ones)
#pragma unroll
    for (int r = 0; r < 2; ++r) xl_first[r] = first_bias ? lane * 8 : xlane[r] + mh * 32;

    // (sync mode only exists for the run-time-staged variants with at most 16 accumulators: the fetch's registers next to 21 would spill)
    constexpr bool CAN_SYNC = KS == 0 && MT * CPW <= 16;
    const bool syncm = CAN_SYNC && x.sync != 0;
    for (int i = 0; i < ntl; ++i) {
        if constexpr (CAN_SYNC) {
            if (syncm && i > 0) {
                sync_load(i, std::integral_constant<int, 2>{});
                __syncthreads();
            }
        }
        const char* xt_c = smem + (syncm ? 0 : (i & 1)) * bufsz;
        const char* dt_c = xt_c + g.XB;
        char* img_n = smem + ((i + 1) & 1) * bufsz;
        const bool more = !syncm && i + 1 < ntl;   // the last tile prefetches nothing
        int n0 = 0, y0 = 0, x0 = 0;
        if (more) origin(i + 1, n0, y0, x0);

        // fragment addresses: ONE vector add per K-step for A and one per column tile for B; everything else is either
        // wave-uniform (scalar unit) or a compile-time constant folded into the ds_read offset field (DSTR follows from MT;
        // PS > 0 is the compile-time input pixel stride -- the lane's second pixel is 4 pixels further in the same row)
        constexpr int DS_C = wg_stride(MT * 64);
        auto abase = [&](int ks) __attribute__((always_inline)) -> const char* { return dt_c + (ks << 4) * DS_C + (dlane[0] + mh * 32); };
        auto read_a1 = [&](const char* ab, int mt) __attribute__((always_inline)) -> half8 {
            return cat8(tr16(ab + mt * 64), tr16(ab + 4 * DS_C + mt * 64));
        };
        auto xbase = [&](int ks) __attribute__((always_inline)) -> const char* {
            const int q0 = ks << 4;                              // first pixel of the K-step
            const int tn = q0 >> (a.ltw + a.lth), ty = (q0 >> a.ltw) & (g.TH - 1);
            return xt_c + ((tn * g.HH + ty + g.padT) * g.HW) * g.PSTR;
        };
        auto read_b = [&](const char* xb, int j) __attribute__((always_inline)) -> half8 {
            // (the bias column reads the area of ones instead: scalar select of the base, per-lane part chosen once)
            const bool ones = j == 0 && first_bias;
            const char* sb = ones ? smem + 2 * bufsz : xb + stoff[j];
            const char* p0 = sb + (j == 0 ? xl_first[0] : xlane[0] + mh * 32);
            if constexpr (PS > 0) return cat8(tr16(p0), tr16(p0 + 4 * PS));
            else return cat8(tr16(p0), tr16(sb + (j == 0 ? xl_first[1] : xlane[1] + mh * 32)));
        };
        auto mma = [&](auto Jc, auto Mc, const half8* afc) __attribute__((always_inline)) {
            constexpr int j = decltype(Jc)::value, mt = decltype(Mc)::value;
            if constexpr (!SPLIT) mma_bf16<0>(acc[mt][j], afc[mt], bf[j]);
            else if constexpr (j * MT + mt < 16) mma_bf16<1>(acc[mt][j], afc[mt], bf[j]);
            else mma_bf16<2>(acc[mt][j], afc[mt], bf[j]);
        };
        // One K-step = MT*CPW "slots", each one MFMA plus a slice of the side work.  A wave issues in order: a second MFMA
        // cannot issue while the matrix pipe is busy (32 cycles), and nothing behind it can either -- so the side work is
        // spread BETWEEN the MFMAs (~5 instructions per slot) instead of after them, and sched_barrier pins that order:
        //   slots 0..MT-1       : A fragments of the next K-step
        //   last slot of column j: refill B fragment j for the next K-step
        //   then, evenly spaced : write the NPV pieces of the row loaded two K-steps ago to the other LDS image, describe the
        //                         next row on the scalar unit, issue its NPV loads.
        // The last K-step "prefetches" the fragments of K-step 0 of the SAME image (wrapped index; they are re-read from the
        // next image after the barrier).
        auto step = [&](int ks, half8* afc, half8* afn, half8* pv, int& rl, int& rd) __attribute__((always_inline)) {
            const int kn = (ks + 1) & (ksteps - 1);
            const char* ab = abase(kn);
            const char* xb = xbase(kn);
            if constexpr (BOTH) {
                if (rl >= 0) {
#pragma unroll
                    for (int u = 0; u < NL; ++u) put_x(img_n, rl, u, pv[u]);
                }
                if (rd >= 0) {
#pragma unroll
                    for (int u = 0; u < WG_ND; ++u) put_d(img_n, rd, u, pv[NL + u]);
                }
                const Row rx = row_x(ks, more, n0, y0, x0), rw = row_d(ks, more, n0, y0, x0);
                rl = rx.lds; rd = rw.lds;
#pragma unroll
                for (int u = 0; u < NL; ++u) pv[u] = load16(rx, relx[u]);
#pragma unroll
                for (int u = 0; u < WG_ND; ++u) pv[NL + u] = load16(rw, reld[u]);
            }
            constexpr int SLOTS = MT * CPW, NSIDE = 2 * NPV + 1;
            const h16* nbase = dzp;   // the row issued in this K-step (!BOTH)
            int nnum = 0, nxs = 0;
            bool nisx = false;
            static_for<0, SLOTS>([&](auto Sc) __attribute__((always_inline)) {
                constexpr int S = decltype(Sc)::value, j = S / MT, mt = S % MT;
                mma(std::integral_constant<int, j>{}, std::integral_constant<int, mt>{}, afc);
                if constexpr (S < MT) afn[S] = read_a1(ab, S);
                if constexpr (mt == MT - 1) {
                    bf[j] = read_b(xb, j);
                }
                if constexpr (!BOTH) {
                    // side item w lives in slot MT + w * (SLOTS - MT) / NSIDE (all in the last slot when there are few slots)
                    static_for<0, NSIDE>([&](auto Wc) __attribute__((always_inline)) {
                        constexpr int w = decltype(Wc)::value;
                        constexpr int sl = SLOTS - MT >= NSIDE ? MT + w * (SLOTS - MT) / NSIDE : SLOTS - 1;
                        if constexpr (sl == S) {
                            if constexpr (w < NPV) {            // write piece w of the row loaded two K-steps ago
                                if (rl >= 0) { if constexpr (w < NL) put_x(img_n, rl, w, pv[w]); }
                                else if (rd >= 0) { if constexpr (w < WG_ND) put_d(img_n, rd, w, pv[w]); }
                            } else if constexpr (w == NPV) {     // describe the next row
                                nisx = ks < x.rswx;
                                // (one formula with selected parameters: no control flow, no values through memory)
                                const int rs = nisx ? ks : ks - x.rswx;
                                const int row = wave + WG_WAVES * rs;
                                const bool item = more && rs < (nisx ? x.rswx : x.rswd) && row < (nisx ? RX : RD);
                                const int tn = nisx ? (int)fdivw(row, x.mg_hh) : row >> a.lth;
                                const int hy = row - tn * (nisx ? g.HH : g.TH);
                                const int n = n0 + tn, y = y0 + hy - (nisx ? g.padT : 0);
                                const int shx = nisx ? sh : 0, csx = nisx ? scs : a.dz.cs;
                                const int Hh = nisx ? Hs : a.H, Ww = nisx ? Ws : a.W;
                                const bool rowok = item && n < a.N && (unsigned)y < (unsigned)a.H;
                                nbase = (nisx ? sp : dzp) + (long long)((n * Hh + (y >> shx)) * Ww) * csx;
                                nnum = rowok ? Ww * csx * 2 : 0;
                                nxs = ((x0 >> shx) * csx) * 2;
                                const int lds = item ? (nisx ? row * g.HW * g.PSTR : g.XB + row * g.TW * g.DSTR) : -1;
                                rl = nisx ? lds : -1;
                                rd = nisx ? -1 : lds;
                            } else {                             // issue load w - NPV - 1 of it
                                constexpr int u = w - NPV - 1;
                                Row nrow;
                                nrow.base = nbase; nrow.num = nnum; nrow.xs = nxs; nrow.lds = 0;
                                int relv = (int)0x80000000, relw = (int)0x80000000;
                                if constexpr (u < NL) relv = relx[u];
                                if constexpr (u < WG_ND) relw = reld[u];
                                pv[u] = load16(nrow, nisx ? relv : relw);
                            }
                        }
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };

        {   // fragments of K-step 0 (the image became visible at the barrier just passed)
            const char* ab = abase(0);
            const char* xb = xbase(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) afA[mt] = read_a1(ab, mt);
#pragma unroll
            for (int j = 0; j < CPW; ++j)
                bf[j] = read_b(xb, j);
        }
        if constexpr (KS > 0) {
            // ---- static schedule (hot shapes: one image per tile, KS K-steps, RWX input + RWD dZ rows per wave) -------------
            // The K loop is fully unrolled and every K-step knows at compile time what it loads (K-step ks loads row item ks:
            // input rows first, then dZ rows) and what it writes to LDS (the item of K-step ks-2): no branches, no selects.  A
            // wave whose row index is past the tile (the row count need not divide by 4) loads zeros and writes them to the
            // image's dummy row.
            // BOTH (wide rows, 1x1 layers): K-step ks loads input row ks AND dZ row ks, prefetch distance 2.
            // !BOTH (round 4): the dZ rows of the next tile arrive by LDS-DMA, issued at the head of K-step 0 (the oldest requests of
            // the tile: vmcnt completes in order, so the input rows' waits cover them) -- no registers, no LDS stores, and the input
            // rows' prefetch distance grows from 3 K-steps to KS - RWX (5 of 8): under full-chip load a row's latency exceeds 3
            // K-steps (per-tile time 3.1 us on a quiet chip, 4.9 us with every CU streaming).
            const int xsx = ((x0 >> sh) * scs) * 2, xsd = x0 * a.dz.cs * 2;
            constexpr bool DZDMA = !BOTH;
            constexpr int DIST = BOTH ? 2 : (RWX <= 3 ? KS - RWX : 3);   // K-steps between a row's loads and its LDS writes
            constexpr int NSET = BOTH ? 2 : 3;          // prefetch register sets
            constexpr int DO = BOTH ? NL : 0;           // first dZ piece inside a prefetch register set
            static_assert((BOTH ? (RWX > RWD ? RWX : RWD) : (DZDMA ? RWX : RWX + RWD)) + DIST <= KS, "row items must be committed within their tile");
            static_assert(WG_ND * 64 >= 16 * 12, "a dZ row is at most 16 pixels x 12 pieces");
            static_for<0, KS>([&](auto Kc) __attribute__((always_inline)) {
                constexpr int ks = decltype(Kc)::value;
                constexpr int kc = ks - DIST;               // the K-step whose loads are written to LDS now
                constexpr bool LX = BOTH ? ks < RWX : ks < RWX;                              // loaded in this K-step
                constexpr bool LD = BOTH ? ks < RWD : (!DZDMA && ks >= RWX && ks < RWX + RWD);
                constexpr bool CX = kc >= 0 && (BOTH ? kc < RWX : kc < RWX);                 // committed in this K-step
                constexpr bool CD = kc >= 0 && (BOTH ? kc < RWD : (!DZDMA && kc >= RWX && kc < RWX + RWD));
                constexpr int rsx = ks, rsd = BOTH ? ks : ks - RWX;                          // row slots of the loads
                half8* afc = (ks & 1) ? afB : afA;
                half8* afn = (ks & 1) ? afA : afB;
                // (a set is loaded at K-step ks and committed at ks + DIST; never both in one K-step unless DIST == NSET)
                constexpr int si = (LX || LD) ? ks % NSET : (kc >= 0 ? kc % NSET : 0);
                half8* pv = si == 0 ? pvA : si == 1 ? pvB : pvC;
                int& rox = si == 0 ? rlA : si == 1 ? rlB : rlC;   // LDS byte offsets (inside an image) of the set's rows
                int& rod = si == 0 ? rdA : si == 1 ? rdB : rdC;
                constexpr int kn = (ks + 1) % KS;
                const char* ab = abase(kn);
                const char* xb = xbase(kn);
                const h16* nbx = dzp;
                const h16* nbd = dzp;
                int nnx = 0, nnd = 0;
                u32x4_t drs = {0u, 0u, 0u, SSDN_BUFFER_RSRC_FLAGS};
                unsigned dlds = 0;
                // side items in issue order: LDS-DMA of the dZ rows (K-step 0), LDS writes of the old rows, then description + loads
                // of the new rows
                constexpr int NDM = (DZDMA && ks == 0) ? RWD * (WG_ND + 1) : 0;
                constexpr int NCX = CX ? NL : 0, NCD = CD ? WG_ND : 0, NLX = LX ? NL + 1 : 0, NLD = LD ? WG_ND + 1 : 0;
                constexpr int NSIDE = NDM + NCX + NCD + NLX + NLD;
                constexpr int SLOTS = MT * CPW;
                static_for<0, SLOTS>([&](auto Sc) __attribute__((always_inline)) {
                    constexpr int S = decltype(Sc)::value, j = S / MT, mt = S % MT;
                    mma(std::integral_constant<int, j>{}, std::integral_constant<int, mt>{}, afc);
                    if constexpr (S < MT) afn[S] = read_a1(ab, S);
                    if constexpr (mt == MT - 1) {
                        bf[j] = read_b(xb, j);
                    }
                    static_for<0, NSIDE>([&](auto Wc) __attribute__((always_inline)) {
                        constexpr int w0 = decltype(Wc)::value;
                        constexpr int sl = SLOTS - MT >= NSIDE ? MT + w0 * (SLOTS - MT) / NSIDE : MT + (w0 * (SLOTS - MT)) / NSIDE;
                        if constexpr (sl == S) {
                            if constexpr (w0 < NDM) {
                                constexpr int r = w0 / (WG_ND + 1), u = w0 % (WG_ND + 1) - 1;
                                if constexpr (u < 0) {
                                    const int row = wave + WG_WAVES * r;
                                    const bool valid = row < g.TH;
                                    const int y = y0 + row;
                                    const bool ok = more && valid && y < a.H;
                                    const unsigned long long bp = (unsigned long long)(dzp + (long long)((n0 * a.H + y) * a.W) * a.dz.cs);
                                    drs[0] = (unsigned)bp;
                                    drs[1] = (unsigned)(bp >> 32) & 0xffffu;
                                    drs[2] = ok ? (unsigned)(a.W * a.dz.cs * 2) : 0u;
                                    dlds = lds0 + (unsigned)(img_n - smem) + (unsigned)(g.XB + (valid ? row : g.TH) * g.TW * g.DSTR);
                                } else {
                                    if (more) wg_dma16(dlds + u * 1024, dvoff[u], drs, xsd);
                                }
                            } else if constexpr (w0 - NDM < NCX) {
                                constexpr int w = w0 - NDM;
                                put_x(img_n, rox, w, pv[w]);
                            } else if constexpr (w0 - NDM < NCX + NCD) {
                                constexpr int w = w0 - NDM;
                                put_d(img_n, rod, w - NCX, pv[DO + w - NCX]);
                            } else if constexpr (w0 - NDM < NCX + NCD + NLX) {
                                constexpr int w = w0 - NDM;
                                constexpr int u = w - NCX - NCD - 1;
                                if constexpr (u < 0) {
                                    const int row = wave + WG_WAVES * rsx;
                                    const bool valid = row < g.HH;
                                    const int y = y0 - g.padT + row;
                                    const bool ok = more && valid && (unsigned)y < (unsigned)a.H;
                                    nbx = sp + (long long)((n0 * Hs + (y >> sh)) * Ws) * scs;
                                    nnx = ok ? Ws * scs * 2 : 0;
                                    rox = (valid ? row : g.HH) * g.HW * g.PSTR;
                                } else {
                                    Row nrow;
                                    nrow.base = nbx; nrow.num = nnx; nrow.lds = 0; nrow.xs = xsx;
                                    pv[u] = load16(nrow, relx[u]);
                                }
                            } else {
                                constexpr int w = w0 - NDM;
                                constexpr int u = w - NCX - NCD - NLX - 1;
                                if constexpr (u < 0) {
                                    const int row = wave + WG_WAVES * rsd;
                                    const bool valid = row < g.TH;
                                    const int y = y0 + row;
                                    const bool ok = more && valid && y < a.H;
                                    nbd = dzp + (long long)((n0 * a.H + y) * a.W) * a.dz.cs;
                                    nnd = ok ? a.W * a.dz.cs * 2 : 0;
                                    rod = g.XB + (valid ? row : g.TH) * g.TW * g.DSTR;
                                } else {
                                    Row nrow;
                                    nrow.base = nbd; nrow.num = nnd; nrow.lds = 0; nrow.xs = xsd;
                                    pv[DO + u] = load16(nrow, reld[u]);
                                }
                            }
                        }
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            if constexpr (DZDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's dZ rows have landed (barrier below)
        } else {
#pragma unroll 1
            for (int ks = 0; ks < ksteps; ks += 2) {   // ksteps is even (tiles have >= 32 pixels)
                step(ks, afA, afB, pvA, rlA, rdA);
                step(ks + 1, afB, afA, pvB, rlB, rdB);
            }
        }
        __syncthreads();   // image i fully consumed by every wave, image i+1 complete
        stamp();
    }

    stamp();
    if constexpr (SPLIT) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    // ---- write this workgroup's slab: D row = m (8*(r>>2) + 4*kh + (r&3)), D col = k (l31) ----
    const long long wg_sidx = (long long)wg_mb * a.nslabs + wg_slab;
    float* slab = a.slab + wg_sidx * a.ntaps * a.Mpad * a.Kpad;
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        if (!ct_on[j]) continue;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = mt * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
                if (ct_bias[j]) {
                    if (l31 == 0) a.bslab[wg_sidx * a.Mpad + m] = acc[mt][j][r];
                } else {
                    slab[((long long)ct_tap[j] * a.Mpad + m) * a.Kpad + ct_nt[j] * 32 + l31] = acc[mt][j][r];
                }
            }
        }
    }
}

// algorithmic flops of one op: 2 * pixels * output channels * input channels per tap * taps.  The "taps" of a 1x1 head layer are
// 96-channel blocks of its input (coff[t]): every tap multiplies M x Kpad, not M x Ktot (the staged width)
static inline double wgrad_flops(const ssdn_wgrad_args* a) {
    bool blocks = a->ntaps > 1;
    for (int t = 0; t < a->ntaps; ++t) blocks = blocks && a->dy[t] == 0 && a->dx[t] == 0;
    const double px = (double)a->N * a->H * a->W;
    const int mb = a->mblocks > 1 ? a->mblocks : 1;
    const int kt = blocks ? (a->Kpad < a->Ktot ? a->Kpad : a->Ktot) : a->Ktot;
    return 2.0 * px * a->M * mb * kt * a->ntaps;
}
static int wgrad_validate(const ssdn_wgrad_args* a) {
    if (a->ntaps < 1 || a->ntaps > SSDN_MAX_TAPS) return ssdn_set_error("wgrad: ntaps out of range");
    if (a->ltw + a->lth + a->ltn > 8 || a->ltw + a->lth + a->ltn < 5) return ssdn_set_error("wgrad: tile must have 32..256 pixels");
    if (a->ltw > 4) return ssdn_set_error("wgrad: tiles are at most 16 pixels wide");
    if (a->Ktot != a->c0 + a->c1 || (a->Ktot & 7)) return ssdn_set_error("wgrad: Ktot must equal c0+c1 (multiple of 8)");
    if ((a->c0 & 7) || (a->c1 & 7)) return ssdn_set_error("wgrad: source channel counts must be multiples of 8");
    if ((a->Kpad & 31) || a->Kpad > 96) return ssdn_set_error("wgrad: Kpad must be 32/64/96");
    for (int t = 0; t < a->ntaps; ++t)
        if (a->coff[t] < 0 || (a->coff[t] & 15) || a->coff[t] >= a->Ktot) return ssdn_set_error("wgrad: bad channel offset of tap %d", t);
    if ((a->Mpad & 31) || a->Mpad > 96 || a->M > a->Mpad || (a->M & 7)) return ssdn_set_error("wgrad: Mpad must be 32/64/96, M %% 8 == 0");
    if (a->nslabs < 1) return ssdn_set_error("wgrad: nslabs < 1");
    if (a->c0 && a->c1) return ssdn_set_error("wgrad: one input tensor per launch (c0 == 0 or c1 == 0)");
    if (a->csplit < 0 || a->csplit > 32) return ssdn_set_error("wgrad: csplit (column groups) must be 0..32");
    if (a->mblocks < 0 || a->mblocks > 16) return ssdn_set_error("wgrad: mblocks must be 0..16 (0 = 1)");
    if (a->mblocks > 1 && a->csplit > 1) return ssdn_set_error("wgrad: mblocks > 1 and csplit > 1 exclude each other");
    if (a->c0 && a->up0 && (a->ltw < 1 || a->lth < 1)) return ssdn_set_error("wgrad: upsampled input needs even tile origins (tile >= 2x2)");
    return 0;
}
// Row items per wave per tile and the kernel variant that fits them into the K-steps 0..ksteps-3 (prefetch distance 2):
//   nl = 4, both = 0: one row item per K-step (input rows, then dZ rows), <= 4 loads per input row
//   nl = 6, both = 1: one input row AND one dZ row per K-step, <= 6 loads per input row
// nl = 99: the tile cannot be prefetched.
struct WgItems { int rswx, rswd, nl, both, sync; };
static WgItems wgrad_items(const ssdn_wgrad_args* a, const WgGeom& g) {
    WgItems t;
    const int npix = g.TN * g.TH * g.TW;
    const int ix = (g.HW * (a->Ktot / 8) + 63) / 64, id = (g.TW * (a->M / 8) + 63) / 64;
    t.rswx = (g.TN * g.HH + WG_WAVES - 1) / WG_WAVES;
    t.rswd = (g.TN * g.TH + WG_WAVES - 1) / WG_WAVES;
    const int steps = (npix >> 4) - 2;             // loads are issued in K-steps 0..ksteps-3
    const bool single = g.ntiles <= a->nslabs;     // one tile per workgroup: nothing to prefetch
    t.nl = 99; t.both = 0; t.sync = 0;
    if (id > WG_ND) return t;
    if (ix <= 4 && (single || t.rswx + t.rswd <= steps)) { t.nl = 4; t.both = 0; }
    else if (ix <= 6 && (single || (t.rswx <= steps && t.rswd <= steps))) { t.nl = 6; t.both = 1; }
    else if (ix <= 6) { t.nl = ix <= 4 ? 4 : 6; t.both = ix <= 4 ? 0 : 1; t.sync = 1; }   // several tiles per workgroup, none prefetched
    return t;
}
// the compile-time schedules of the layers with <= 48 input channels (an input row = 18 pixels x 6 pieces = 108 pieces) take TWO 64-lane
// loads per row instead of four (half of which repeated earlier pieces): half the staging instructions of these classes
static inline bool wgrad_nl2(const ssdn_wgrad_args* a, const WgGeom& g, int MT, int CPW) {
    const int ix = (g.HW * (a->Ktot / 8) + 63) / 64;
    return ix <= 2 && CPW == 5 && (MT == 2 || MT == 3);
}
struct WgPrep { WgGeom g; WgAux x; WgItems wi; int MT, CPW, gx, gy; size_t lds; };
static int wgrad_prepare(const ssdn_wgrad_args* a, WgPrep* p) {
    int rc = wgrad_validate(a);
    if (rc) return rc;
    p->g = wg_geom(*a);
    WgAux& x = p->x;
    memset(&x, 0, sizeof(x));
    x.trace = (unsigned long long*)ssdn_debug_get_trace();
    x.mg_hh = magic_ofw(p->g.HH);
    x.ccx = a->Ktot / 8;
    x.ccd = a->M / 8;
    x.mg_ccx = magic_ofw(x.ccx);
    x.mg_ccd = magic_ofw(x.ccd);
    p->wi = wgrad_items(a, p->g);
    if (p->wi.nl > 6) return ssdn_set_error("wgrad: the tile cannot be prefetched within its K-steps (too many rows / too wide rows)");
    x.rswx = p->wi.rswx; x.rswd = p->wi.rswd;
    x.sync = p->wi.sync;
    p->MT = a->Mpad / 32;
    const int CT = a->ntaps * (a->Kpad / 32) + 1;
    p->gy = a->csplit > 1 ? a->csplit : 1;
    p->CPW = (CT + WG_WAVES * p->gy - 1) / (WG_WAVES * p->gy);   // column tiles per wave (gy = column groups)
    p->gx = a->mblocks > 1 ? ((a->nslabs + 7) / 8) * 8 * a->mblocks : a->nslabs;
    p->lds = 2 * ((size_t)p->g.XB + (size_t)p->g.DB) + WG_ONES_BYTES;
    if (p->lds > 160 * 1024) return ssdn_set_error("wgrad: tiling needs %zu B of LDS (> 160 KiB)", p->lds);
    if (p->wi.sync && p->MT * p->CPW > 16)
        return ssdn_set_error("wgrad: the tile cannot be prefetched within its K-steps and the %d-accumulator variant has no synchronous mode (use column groups)", p->MT * p->CPW);
    return 0;
}

// ---- thin-K weight gradient: 1..3 real input channels under a 3x3 window ---------------------------------------------------------
// encode_block_1.0 and the image half of decode_block_1.0 see a 3-channel input in 16 / 32 channel slots: k_wgrad pays 9 column
// tiles of 32 (30 MFMAs per 16 pixels) and, above all, its row-item staging of the 96-channel gradient for 2.7 GFLOP of real
// work (78 + 55 us in situ).  Here the GEMM is im2col-shaped: D[m][n] += dz^T[m][pixel] * B[pixel][n] with n = 3 * tap + channel
// (27 columns) + the bias column of ones -- ONE column tile.  A workgroup walks 16x16 tiles: the gradient tile and the 18x18
// halo of the first 8 input slots are prefetched into registers during the previous tile's MFMAs and staged in LDS (gradient
// pixel stride == 64 mod 128 bytes for the transpose reads, input converted to bf16); a wave takes the tile rows wave, wave+4,
// ...: per row one A fragment per 32 gradient channels (ds_read_b64_tr_b16, as k_wgrad), one B fragment gathered from the halo
// (8 x ds_read_u16 at this lane's (tap, channel)), MT MFMAs.  The four waves' partial sums meet in LDS in a fixed order; the slab
// has k_wgrad's layout, so SSDN_OP_WREDUCE is unchanged.
#define WGN_THREADS 256
template <int MT>
static __device__ __forceinline__ void wgrad_thin_body(const ssdn_wgrad_args& a, const unsigned bx, const unsigned gdx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int DSTR = wg_stride(MT * 64);            // gradient pixel stride in LDS
    constexpr int DB = 256 * DSTR, XB = 18 * 18 * 16;   // one gradient tile / one halo of 8 channel slots (16 B per pixel)
    constexpr int NPD = MT * 4;                          // 16-byte gradient pieces per pixel
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, mh = (lane >> 4) & 1;
    int mny = 0, mnx = 0;
    for (int t = 0; t < 9; ++t) { mny = a.dy[t] < mny ? a.dy[t] : mny; mnx = a.dx[t] < mnx ? a.dx[t] : mnx; }
    const int padT = -mny, padL = -mnx;
    const bool use0 = a.c0 > 0;
    const h16* xp = (const h16*)(use0 ? a.src0.p : a.src1.p) + (use0 ? a.src0.co : a.src1.co);
    const int xcs = use0 ? a.src0.cs : a.src1.cs;
    const unsigned short* dzp = (const unsigned short*)a.dz.p + a.dz.co;
    const int tiles_x = a.W >> 4, tiles_y = a.H >> 4;
    const int ntiles = a.N * tiles_x * tiles_y;
    const int mpieces = a.M >> 3;                        // real 16-byte pieces per gradient pixel (<= NPD)

    // this lane's B column: n = l31 -> (tap, channel) | bias | nothing
    const int ncol = 9 * a.kreal;
    const int bt = l31 < ncol ? l31 / a.kreal : 0, bc = l31 < ncol ? l31 - bt * a.kreal : 0;
    const int bmode = l31 < ncol ? 0 : (l31 == ncol ? 1 : 2);
    const int boff = ((a.dy[bt] + padT) * 18 + a.dx[bt] + padL + kh * 8) * 16 + bc * 2;     // + row * 18 * 16 per K-step

    // A fragment lane part (k_wgrad: dlane / abase / read_a1)
    const int dlane0 = (kh * 8 + (i16 >> 2)) * DSTR + (i16 & 3) * 8 + mh * 32;

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    half8 pd[NPD], px[2];
    auto origin = [&](int tile, int& n, int& y0, int& x0) __attribute__((always_inline)) {
        const int tx = tile % tiles_x; tile /= tiles_x;
        const int ty = tile % tiles_y;
        n = tile / tiles_y; y0 = ty << 4; x0 = tx << 4;
    };
    auto fetch = [&](int tile) __attribute__((always_inline)) {
        int n, y0, x0;
        origin(tile, n, y0, x0);
#pragma unroll
        for (int u = 0; u < NPD; ++u) {
            const int f = tid + u * WGN_THREADS;
            const int q = f / NPD, cc = f - q * NPD;
            pd[u] = zero_h8();
            if (cc < mpieces)
                pd[u] = __builtin_bit_cast(half8, ld_b8(dzp + ((long long)(n * a.H + y0 + (q >> 4)) * a.W + x0 + (q & 15)) * a.dz.cs + cc * 8));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int hp = tid + u * WGN_THREADS;
            const int hy = hp / 18, hx = hp - hy * 18;
            const int y = y0 - padT + hy, xx = x0 - padL + hx;
            px[u] = zero_h8();
            if (hp < 324 && (unsigned)y < (unsigned)a.H && (unsigned)xx < (unsigned)a.W)
                px[u] = ld_h8(xp + ((long long)(n * a.H + y) * a.W + xx) * xcs);
        }
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
        char* dt = smem + buf * (DB + XB);
        char* xt = dt + DB;
#pragma unroll
        for (int u = 0; u < NPD; ++u) {
            const int f = tid + u * WGN_THREADS;
            const int q = f / NPD, cc = f - q * NPD;
            *reinterpret_cast<half8*>(dt + q * DSTR + cc * 16) = pd[u];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int hp = tid + u * WGN_THREADS;
            if (hp < 324) *reinterpret_cast<half8*>(xt + hp * 16) = cvt_h8_to_bf8(px[u]);
        }
    };

    int it = 0;
    const int t0 = (int)bx;
    if (t0 < ntiles) { fetch(t0); stage(0); }
    __syncthreads();
    for (int tile = t0; tile < ntiles; tile += (int)gdx, ++it) {
        const int nxt = tile + (int)gdx;
        if (nxt < ntiles) fetch(nxt);                       // in flight during this tile's MFMAs
        const char* dt = smem + (it & 1) * (DB + XB);
        const char* xt = dt + DB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = wave + 4 * j;                    // tile row = K-step of 16 pixels
            const char* ab = dt + (row << 4) * DSTR + dlane0;
            half8 af[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = cat8(tr16(ab + mt * 64), tr16(ab + 4 * DSTR + mt * 64));
            const char* bp = xt + row * (18 * 16) + boff;
            u16x8 bv;
#pragma unroll
            for (int q = 0; q < 8; ++q) bv[q] = *reinterpret_cast<const unsigned short*>(bp + q * 16);
            if (bmode != 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) bv[q] = bmode == 1 ? (unsigned short)0x3f80 : (unsigned short)0;
            }
            const half8 bfrag = __builtin_bit_cast(half8, bv);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[mt]), __builtin_bit_cast(bf16x8, bfrag), acc[mt], 0, 0, 0);
        }
        __syncthreads();                                     // every wave is done with buffer it & 1 ... and with (it+1) & 1 long ago
        if (nxt < ntiles) stage((it + 1) & 1);
        __syncthreads();
    }
    // ---- the four waves' partial sums, in wave order ----
    float* red = reinterpret_cast<float*>(smem);            // [wave][mt][r][lane]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * MT + mt) * 16 + r) * 64 + lane] = acc[mt][r];
    __syncthreads();
    float* slab = a.slab + (long long)bx * 9 * a.Mpad * a.Kpad;
    for (int o = tid; o < MT * 16 * 64; o += WGN_THREADS) {
        const int ln = o & 63, r = (o >> 6) & 15, mt = o >> 10;
        float sum = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) sum += red[((w4 * MT + mt) * 16 + r) * 64 + ln];
        const int n = ln & 31, m = mt * 32 + 8 * (r >> 2) + 4 * (ln >> 5) + (r & 3);        // D col = lane & 31, D row as k_wgrad
        if (m >= a.Mpad) continue;
        if (n < ncol) {
            const int t = n / a.kreal, c = n - t * a.kreal;
            slab[((long long)t * a.Mpad + m) * a.Kpad + c] = sum;
        } else if (n == ncol) {
            a.bslab[(long long)bx * a.Mpad + m] = sum;
        }
    }
}
static bool wgrad_thin_ok(const ssdn_wgrad_args* a) {
    static const bool off = ssdn_tuning_env("SSDN_NO_THIN_WGRAD") != nullptr;      // A/B aid, read once
    if (off || a->kreal < 1 || a->kreal > 3 || a->ntaps != 9 || a->csplit > 1 || a->mblocks > 1) return false;
    if ((a->H & 15) || (a->W & 15) || (a->M & 7) || a->Mpad > 96 || a->Kpad < a->kreal) return false;
    if (a->c0 > 0 ? (a->up0 || a->c1 > 0) : a->c1 <= 0) return false;
    const ssdn_view& v = a->c0 > 0 ? a->src0 : a->src1;
    if ((v.cs & 7) || (v.co & 7) || (a->dz.cs & 7) || (a->dz.co & 7)) return false;
    int mny = 0, mxy = 0, mnx = 0, mxx = 0;
    for (int t = 0; t < 9; ++t) {
        if (a->coff[t] != 0) return false;
        mny = a->dy[t] < mny ? a->dy[t] : mny; mxy = a->dy[t] > mxy ? a->dy[t] : mxy;
        mnx = a->dx[t] < mnx ? a->dx[t] : mnx; mxx = a->dx[t] > mxx ? a->dx[t] : mxx;
    }
    return mxy - mny == 2 && mxx - mnx == 2;                 // a 3x3 window: the halo is 18 x 18
}

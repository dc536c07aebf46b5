// conv_chain.hip -- the bottom of the U as ONE launch: a run of consecutive 3x3 forward layers whose images are 8x8 pixels or
// smaller (at BASELINE sizes: encode_block_4..6, decode_block_5, decode_block_4 = 7 convolutions + 2 max-pools,
// noise_network.py:200-210) is executed by one workgroup PER IMAGE with every activation of the chain resident in LDS.
//
// Why: as separate launches these layers are chains of dependent latencies -- a 2x2-pixel layer is 1.7 K cycles of MFMA work behind
// a launch, a weight fetch, a tile fetch and an epilogue round trip (10..14 us each, 95 us for 2.6 % of the forward flops, and nothing
// else runs beside the forward pass).  The images of the batch are independent through the whole chain, so a workgroup can walk
// one image through all layers without ever waiting for another workgroup:
//   * planes: every activation of the image (inputs from HBM, each layer's output, each pooled output) is a halo tile in LDS --
//     [H + padT + padB][W + padL + padR] pixels x (C fp16 + 16 B pad), zeroed once (the halo IS the zero padding; an up-sampled
//     source is the half-resolution plane addressed at (y >> 1, x >> 1), a concatenation is two planes) -- 55 KB for the chain above;
//   * wave w owns output-channel tile w (32 rows) of the layer and all (1 or 2) 32-pixel column tiles of the image; its weights
//     never touch LDS: the A fragment of a K-step is 16 bytes per lane straight from the packed [tap][Mpad][Ktot] tensor, and the
//     27 fragments of the NEXT 48-channel chunk -- of the next LAYER at a layer's end -- are in flight while this chunk is on the
//     matrix cores (weights do not depend on activations, so the weight stream never drains at a layer boundary);
//   * epilogue: bias + LeakyReLU in registers -> the output plane (which is also the transposition buffer for the 16-byte stores to
//     HBM that the backward pass needs, and the source of the fused Shift2d + MaxPool2d); one or two s_barrier per layer.
// Results are BIT-IDENTICAL to the separate launches: every output element accumulates chunk -> tap -> K-step in the same order with
// the same MFMA instruction and the same k-slot assignment as k_conv's flat path, and bias / activation / rounding / pooling are
// applied to the same values in the same order (tests/test_hip_ops.py::test_conv_chain_is_bit_identical).
#include "common.h"
#include <cstring>

#define CH_MAX_LAYERS 8
#define CH_MAX_PLANES 14
#define CH_MAX_LOADS 4
#define CH_THREADS 256
extern "C" void* ssdn_debug_get_trace();
#ifdef SSDN_TUNING
#define CH_ABL(c, bit) (((c).ablate & (bit)) != 0)
#else
#define CH_ABL(c, bit) false
#endif

struct ChPlane { int off, str, roww, org, lw, lh, C; };   // LDS byte offset, pixel stride (B), pixels per halo row, byte offset of pixel (0,0)
struct ChLoad { ssdn_view src; int plane; int pad_; };
struct ChLayer {
    const h16* w;
    const float* bias;
    ssdn_view dst, pool;
    int M, Mpad, Ktot, c0, up0;
    int pool_shifted, has_pool;
    unsigned npc_magic;            // magic reciprocal of M / 8 (16-byte pieces per pixel): exact e / npc for e * npc < 2^32
    ChPlane P0, P1, PD, PP;        // source planes (channels [0, c0) / the rest), output plane, pooled plane
};
struct ChainArgs {
    int N, nloads, nlayers, lds_bytes;
    int ablate;       // tuning aid (SSDN_CHAIN_ABLATE, -DSSDN_TUNING builds): 1 no MFMA loop, 2 no HBM stores, 4 no pool, 8 no weight stream
    unsigned long long* trace;   // tuning aid (ssdn_debug_set_trace, -DSSDN_TUNING builds): 64 s_memtime stamps per workgroup, or NULL
    unsigned tap_dy, tap_dx;     // the nine tap offsets + 4, three bits each (SGPR constants: an s_load in the K loop would drain the LDS queue)
    ChPlane pl[CH_MAX_PLANES];
    ChLoad ld[CH_MAX_LOADS];
    ChLayer ly[CH_MAX_LAYERS];
};

static __device__ __forceinline__ void lds_barrier() {   // this wave's LDS traffic done, then the workgroup barrier; global loads stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// the 27 A fragments (9 taps x 3 K-steps) of one 48-channel chunk of one 32-row tile: lane (row l31, k-half kh) reads 16 bytes
static __device__ __forceinline__ void chain_issue_w(half8 (&wr)[27], const h16* lanep, int tapstride) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) wr[t * 3 + ks] = ld_h8(lanep + (long long)t * tapstride + ks * 16);
}

static __device__ __forceinline__ unsigned ch_div(unsigned x, unsigned magic) { return magic ? __umulhi(x, magic) : x; }

template <int NPT, typename STAMP>
static __device__ __forceinline__ void chain_layer(const ChainArgs& c, const ChLayer& L, char* smem, int mt, int l31, int kh,
                                                   half8 (&wr)[27], const h16* next_lanep, int next_tapstride, STAMP stamp) {
    const ChPlane P0 = L.P0, P1 = L.P1, PD = L.PD;
    const int HWp = 1 << (PD.lw + PD.lh);
    // bias of this lane's 16 output rows: on its way while the K loop runs
    float bb[4][4];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const int m0 = mt * 32 + gq * 8 + kh * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) bb[gq][j] = L.bias[m0 < L.M ? m0 + j : 0];
    }
    int py[NPT], px[NPT];
#pragma unroll
    for (int p = 0; p < NPT; ++p) {
        int q = p * 32 + l31;
        if (q >= HWp) q = 0;                                   // image smaller than the column tile: surplus lanes compute pixel 0, store nothing
        py[p] = q >> PD.lw;
        px[p] = q & ((1 << PD.lw) - 1);
    }
    f32x16 acc[NPT];
#pragma unroll
    for (int p = 0; p < NPT; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    const int nch = L.Ktot / 48;
    const int tapstride = L.Mpad * L.Ktot;
    const h16* lanep = L.w + (long long)(mt * 32 + l31) * L.Ktot + kh * 8;
    for (int ch = 0; ch < nch && !CH_ABL(c, 1); ++ch) {
        const int k0 = ch * 48;
        const bool from0 = k0 < L.c0;
        const int sh = from0 ? L.up0 : 0;
        const int roww = from0 ? P0.roww : P1.roww, str = from0 ? P0.str : P1.str;
        const int cbase = (from0 ? P0.off + P0.org + k0 * 2 : P1.off + P1.org + (k0 - L.c0) * 2) + kh * 16;
        // the stream runs one chunk ahead: chunk ch+1 of this layer, or chunk 0 of the next layer this wave works on (always a valid
        // address: the last chunk of the chain re-loads itself -- unconditional loads keep the compiler's vmcnt accounting exact)
        const h16* np = ch + 1 < nch ? lanep + (ch + 1) * 48 : next_lanep;
        const int nts = ch + 1 < nch ? tapstride : next_tapstride;
        int boff[9][NPT];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = (int)((c.tap_dy >> (3 * t)) & 7u) - 4, dx = (int)((c.tap_dx >> (3 * t)) & 7u) - 4;
#pragma unroll
            for (int p = 0; p < NPT; ++p)
                boff[t][p] = cbase + (((py[p] + dy) >> sh) * roww + ((px[p] + dx) >> sh)) * str;
        }
        // B fragments are read CH_BD K-steps ahead of the MFMA that consumes them (one wave per SIMD: nothing else hides the LDS latency)
        constexpr int BD = 4;
        half8 bq[BD][NPT];
        auto rd = [&](int s, int slot) __attribute__((always_inline)) {
#pragma unroll
            for (int p = 0; p < NPT; ++p) bq[slot][p] = *reinterpret_cast<const half8*>(smem + boff[s / 3][p] + (s % 3) * 32);
        };
        __builtin_amdgcn_sched_barrier(0);                     // (the tap offsets above are computed before the pipeline starts)
#pragma unroll
        for (int s = 0; s < BD; ++s) rd(s, s);
        stamp();
#pragma unroll
        for (int s = 0; s < 27; ++s) {
            const half8 a = wr[s];
            if (!CH_ABL(c, 8)) wr[s] = ld_h8(np + (long long)(s / 3) * nts + (s % 3) * 16);
#pragma unroll
            for (int p = 0; p < NPT; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq[s % BD][p], acc[p], 0, 0, 0);
            if (s + BD < 27) rd(s + BD, s % BD);
            __builtin_amdgcn_sched_barrier(0);                 // pin the step: left alone, the scheduler sinks every read to its MFMA
        }
        stamp();
    }
    // epilogue: bias + LeakyReLU -> fp16 -> the output plane (rows >= M of a padded tile are not stored)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const int m0 = mt * 32 + gq * 8 + kh * 4;
        if (m0 >= L.M) continue;
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
            if (p * 32 + l31 >= HWp) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = lrelu(acc[p][gq * 4 + j] + bb[gq][j]);
            u32x2_t o;
            o[0] = pack_f16x2(v[0], v[1]);
            o[1] = pack_f16x2(v[2], v[3]);
            *reinterpret_cast<u32x2_t*>(smem + PD.off + PD.org + (py[p] * PD.roww + px[p]) * PD.str + m0 * 2) = o;
        }
    }
}

__global__ __launch_bounds__(CH_THREADS) void k_conv_chain(ChainArgs c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    [[maybe_unused]] int tr_i = 0;
    auto stamp = [&]() {
#ifdef SSDN_TUNING
        if (c.trace && tid == 0 && tr_i < 64) c.trace[(size_t)blockIdx.x * 64 + tr_i++] = __builtin_amdgcn_s_memtime();
#endif
    };
    stamp();
    half8 wr[27];
    // weights of the first layer this wave works on: in flight while the planes are zeroed and the inputs arrive
    auto lanep_of = [&](int li) { return c.ly[li].w + (long long)(wave * 32 + l31) * c.ly[li].Ktot + kh * 8; };
    auto active = [&](int li) { return wave * 32 < c.ly[li].Mpad; };
    auto next_active = [&](int li) {                           // first layer after li this wave works on, or -1
        int j = li + 1;
        while (j < c.nlayers && !active(j)) ++j;
        return j < c.nlayers ? j : -1;
    };
    {
        const int f = active(0) ? 0 : next_active(0);
        if (f >= 0) chain_issue_w(wr, lanep_of(f), c.ly[f].Mpad * c.ly[f].Ktot);
    }
    for (int z = tid * 16; z < c.lds_bytes; z += CH_THREADS * 16) *reinterpret_cast<half8*>(smem + z) = zero_h8();
    lds_barrier();
    for (int i = 0; i < c.nloads; ++i) {
        const ChPlane P = c.pl[c.ld[i].plane];
        const int npc = P.C >> 3, total = npc << (P.lw + P.lh);
        const h16* src = (const h16*)c.ld[i].src.p + c.ld[i].src.co;
        for (int e = tid; e < total; e += CH_THREADS) {
            const int q = e / npc, cc = e - q * npc;
            const half8 v = ld_h8(src + ((long long)(n << (P.lw + P.lh)) + q) * c.ld[i].src.cs + cc * 8);
            *reinterpret_cast<half8*>(smem + P.off + P.org + ((q >> P.lw) * P.roww + (q & ((1 << P.lw) - 1))) * P.str + cc * 16) = v;
        }
    }
    lds_barrier();
    stamp();
    for (int li = 0; li < c.nlayers; ++li) {
        const ChLayer& L = c.ly[li];
        const ChPlane PD = L.PD;
        if (active(li)) {
            const int nx = next_active(li);
            const h16* nlp = nx >= 0 ? lanep_of(nx) : lanep_of(li) + (L.Ktot - 48);
            const int nts = nx >= 0 ? c.ly[nx].Mpad * c.ly[nx].Ktot : L.Mpad * L.Ktot;
            if (PD.lw + PD.lh > 5) chain_layer<2>(c, L, smem, wave, l31, kh, wr, nlp, nts, stamp);
            else chain_layer<1>(c, L, smem, wave, l31, kh, wr, nlp, nts, stamp);
        }                                                      // (an idle wave keeps the chunk it holds for its next layer)
        stamp();
        lds_barrier();
        stamp();
        // ---- the output plane -> HBM (16-byte pieces of consecutive pixels), and the fused Shift2d + MaxPool2d ----
        const int lhw = PD.lw + PD.lh;
        const int npc = L.M >> 3;
        {
            const int total = npc << lhw;
            h16* dst = (h16*)L.dst.p + L.dst.co;
            for (int e = tid; e < total; e += CH_THREADS) {
                const int q = ch_div(e, L.npc_magic), cc = e - q * npc;
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(smem + PD.off + PD.org + ((q >> PD.lw) * PD.roww + (q & ((1 << PD.lw) - 1))) * PD.str + cc * 16);
                if (!CH_ABL(c, 2)) *reinterpret_cast<u32x4_t*>(dst + ((long long)(n << lhw) + q) * L.dst.cs + cc * 8) = v;
            }
        }
        stamp();
        if (L.has_pool && !CH_ABL(c, 4)) {
            const ChPlane PP = L.PP;
            const int total = npc << (lhw - 2);
            h16* dst = (h16*)L.pool.p + L.pool.co;
            for (int e = tid; e < total; e += CH_THREADS) {
                const int pq = ch_div(e, L.npc_magic), cc = e - pq * npc;
                const int pj = pq & ((1 << PP.lw) - 1), pi = pq >> PP.lw;
                const int r0 = L.pool_shifted ? 2 * pi - 1 : 2 * pi;
                u32x4_t best;
                bool have = false;
#pragma unroll
                for (int dr = 0; dr < 2; ++dr) {
                    const int r = r0 + dr;
#pragma unroll
                    for (int dc = 0; dc < 2; ++dc) {
                        u32x4_t v = {0u, 0u, 0u, 0u};
                        if (r >= 0) v = *reinterpret_cast<const u32x4_t*>(smem + PD.off + PD.org + (r * PD.roww + 2 * pj + dc) * PD.str + cc * 16);
                        if (!have) { best = v; have = true; }
                        else {
                            const half8 m = __builtin_elementwise_max(__builtin_bit_cast(half8, best), __builtin_bit_cast(half8, v));
                            best = __builtin_bit_cast(u32x4_t, m);
                        }
                    }
                }
                *reinterpret_cast<u32x4_t*>(smem + PP.off + PP.org + (pi * PP.roww + pj) * PP.str + cc * 16) = best;
                if (!CH_ABL(c, 2)) *reinterpret_cast<u32x4_t*>(dst + ((long long)(n << (lhw - 2)) + pq) * L.pool.cs + cc * 8) = best;
            }
            lds_barrier();
        }
        stamp();
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
static bool g_chain_on = true;
extern "C" int ssdn_conv_set_chain(int on) { g_chain_on = on != 0; return 0; }

static int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
static bool same_view(const ssdn_view& a, const ssdn_view& b) { return a.p == b.p && a.cs == b.cs && a.co == b.co; }

struct ChainBuild {
    ChainArgs c;
    int nplanes = 0;
    ssdn_view pv[CH_MAX_PLANES];     // the HBM tensor view a plane mirrors
    int pH[CH_MAX_PLANES], pW[CH_MAX_PLANES];
    bool produced[CH_MAX_PLANES];    // written by a layer of the chain (else: loaded from HBM)
    int padT, padB, padL, padR;
};

static int chain_add_plane(ChainBuild& b, const ssdn_view& v, int H, int W, int C, bool produced) {
    if (b.nplanes >= CH_MAX_PLANES) return -1;
    const int i = b.nplanes++;
    ChPlane& P = b.c.pl[i];
    P.lw = ilog2_exact(W); P.lh = ilog2_exact(H); P.C = C;
    P.str = C * 2 + 16;
    P.roww = W + b.padL + b.padR;
    P.org = (b.padT * P.roww + b.padL) * P.str;
    P.off = b.c.lds_bytes;
    b.c.lds_bytes += (H + b.padT + b.padB) * P.roww * P.str;
    b.c.lds_bytes = (b.c.lds_bytes + 15) & ~15;
    b.pv[i] = v; b.pH[i] = H; b.pW[i] = W; b.produced[i] = produced;
    return i;
}
// plane holding channels [v.co, v.co + C) of tensor v at H x W: an existing one, a new load from HBM, or -2 if the tensor is written
// inside the chain in a shape this reader does not match (the chain must end before the reader)
static int chain_source(ChainBuild& b, const ssdn_view& v, int H, int W, int C) {
    for (int i = 0; i < b.nplanes; ++i) {
        if (b.pv[i].p != v.p) continue;
        if (same_view(b.pv[i], v) && b.pH[i] == H && b.pW[i] == W && b.c.pl[i].C == C) return i;
        if (b.produced[i]) return -2;
    }
    if (b.c.nloads >= CH_MAX_LOADS) return -2;
    const int i = chain_add_plane(b, v, H, W, C, false);
    if (i < 0) return -2;
    b.c.ld[b.c.nloads].src = v;
    b.c.ld[b.c.nloads].plane = i;
    ++b.c.nloads;
    return i;
}

static bool chain_layer_ok(const ssdn_conv_args* a, const ssdn_conv_args* first) {
    if (conv_validate(a)) return false;
    if (a->bf16 || a->ntaps != 9 || !a->dst.p || a->dst32 || !a->act || !a->bias || a->mask.p || a->add.p || a->upsum.p || a->unrot.p) return false;
    if (a->kc != 48 || a->Ktot % 48 || a->c0 % 48 || a->c1 % 48 || a->Ktot <= 0) return false;
    if (ilog2_exact(a->H) < 0 || ilog2_exact(a->W) < 0 || a->H * a->W > 64) return false;
    if ((a->M & 7) || a->Mpad > 32 * (CH_THREADS / 64) || a->N != first->N) return false;
    if (a->up0 && (a->c0 == 0 || ((a->H | a->W) & 1))) return false;
    if (a->pool.p && (((a->H | a->W) & 1) || (a->pool.cs & 7) || (a->pool.co & 7))) return false;
    if ((a->src0.cs & 7) || (a->src0.co & 7) || (a->c1 && ((a->src1.cs & 7) || (a->src1.co & 7)))) return false;
    for (int t = 0; t < 9; ++t)
        if (a->dy[t] != first->dy[t] || a->dx[t] != first->dx[t]) return false;
    return true;
}

// longest prefix of items[0..n) that runs as one k_conv_chain launch (0 or >= 2 layers); fills *out when out != NULL
static int chain_plan(const ssdn_conv_args* const* items, int n, ChainArgs* out) {
    if (!g_chain_on || n < 2) return 0;
    ChainBuild b;
    memset(&b.c, 0, sizeof(b.c));
    const ssdn_conv_args* f = items[0];
    if (!chain_layer_ok(f, f)) return 0;
    int mny = 0, mxy = 0, mnx = 0, mxx = 0;
    for (int t = 0; t < 9; ++t) {
        if (f->dy[t] < -4 || f->dy[t] > 3 || f->dx[t] < -4 || f->dx[t] > 3) return 0;
        b.c.tap_dy |= (unsigned)(f->dy[t] + 4) << (3 * t);
        b.c.tap_dx |= (unsigned)(f->dx[t] + 4) << (3 * t);
        mny = f->dy[t] < mny ? f->dy[t] : mny; mxy = f->dy[t] > mxy ? f->dy[t] : mxy;
        mnx = f->dx[t] < mnx ? f->dx[t] : mnx; mxx = f->dx[t] > mxx ? f->dx[t] : mxx;
    }
    b.padT = -mny; b.padB = mxy; b.padL = -mnx; b.padR = mxx;
    b.c.N = f->N;
    int accepted = 0;
    for (int i = 0; i < n && i < CH_MAX_LAYERS; ++i) {
        const ssdn_conv_args* a = items[i];
        if (!chain_layer_ok(a, f)) break;
        ChainBuild save = b;
        ChLayer& L = b.c.ly[i];
        L.w = (const h16*)a->w; L.bias = a->bias; L.dst = a->dst; L.pool = a->pool;
        L.M = a->M; L.Mpad = a->Mpad; L.Ktot = a->Ktot; L.c0 = a->c0; L.up0 = a->up0; L.pool_shifted = a->pool_shifted;
        int p0 = -1, p1 = -1, pd = -1, pp = -1;
        bool ok = true;
        if (a->c0 > 0) { p0 = chain_source(b, a->src0, a->up0 ? a->H / 2 : a->H, a->up0 ? a->W / 2 : a->W, a->c0); ok = ok && p0 >= 0; }
        if (ok && a->c1 > 0) { p1 = chain_source(b, a->src1, a->H, a->W, a->c1); ok = ok && p1 >= 0; }
        // an output tensor that is already mirrored by a plane (written twice, or written after it was loaded) is not a chain
        for (int j = 0; ok && j < b.nplanes; ++j)
            if (b.pv[j].p == a->dst.p || (a->pool.p && b.pv[j].p == a->pool.p)) ok = false;
        if (ok) { pd = chain_add_plane(b, a->dst, a->H, a->W, a->M, true); ok = pd >= 0; }
        if (ok && a->pool.p) { pp = chain_add_plane(b, a->pool, a->H / 2, a->W / 2, a->M, true); ok = pp >= 0; }
        if (ok) {
            L.P0 = b.c.pl[p0 >= 0 ? p0 : p1];
            L.P1 = b.c.pl[p1 >= 0 ? p1 : p0];
            L.PD = b.c.pl[pd];
            L.has_pool = pp >= 0;
            L.PP = b.c.pl[pp >= 0 ? pp : pd];
            const unsigned npc = (unsigned)a->M / 8;
            L.npc_magic = npc <= 1 ? 0u : (unsigned)((0x100000000ull + npc - 1) / npc);
        }
        if (!ok || b.c.lds_bytes > 160 * 1024) { b = save; break; }
        accepted = i + 1;
    }
    if (accepted < 2) return 0;
    b.c.nlayers = accepted;
    if (out) *out = b.c;
    return accepted;
}

int conv_chain_len(const ssdn_conv_args* const* items, int n) { return chain_plan(items, n, nullptr); }
extern "C" int ssdn_conv_chain_len(const ssdn_conv_args* const* items, int n) {
    if (!items || n < 0) return ssdn_set_error("conv chain: bad arguments");
    for (int i = 0; i < n; ++i)
        if (!items[i]) return ssdn_set_error("conv chain: null item %d", i);
    return conv_chain_len(items, n > CONV_CHAIN_MAX ? CONV_CHAIN_MAX : n);
}

int launch_conv_chain(const ssdn_conv_args* const* items, int n, hipStream_t s) {
    ChainArgs c;
    if (chain_plan(items, n, &c) != n) return ssdn_set_error("conv chain: the run is not a chain of %d layers", n);
    static bool attr_set = false;
    if (!attr_set) {
        SSDN_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    static const int env_ablate = [] { const char* e = ssdn_tuning_env("SSDN_CHAIN_ABLATE"); return e ? atoi(e) : 0; }();
    c.ablate = env_ablate;
    c.trace = (unsigned long long*)ssdn_debug_get_trace();
    hipLaunchKernelGGL(k_conv_chain, dim3(c.N), dim3(CH_THREADS), c.lds_bytes, s, c);
    SSDN_CHECK_HIP(hipGetLastError());
    return 0;
}

// conv_chain.hip -- the bottom of the U as ONE launch per direction: a run of consecutive main-lane ops on images of 8x8 pixels or
// less is executed by one workgroup PER IMAGE with every activation / gradient of the run resident in LDS.  At BASELINE sizes:
//   forward   encode_block_4..6, decode_block_5, decode_block_4: 7 convolutions + 2 max-pools        (noise_network.py:200-210)
//   backward  the data gradients of the same 7 layers with their fused epilogues (LeakyReLU', skip-gradient add, the adjoint of
//             the nearest up-sampling) + the 3 max-pool backward ops between them                     (autograd of the above)
//
// Why: as separate launches these layers are chains of dependent latencies -- a 2x2-pixel layer is 1.7 K cycles of MFMA work behind
// a launch, a weight fetch, a tile fetch and an epilogue round trip (10..30 us each in situ; 82 us forward and ~210 us backward of
// main-lane time for 2.6 % of the flops).  The images of the batch are independent through the whole run, so a workgroup walks one
// image through all layers without ever waiting for another workgroup:
//   * planes: every tensor of the image the run touches (inputs from HBM, each layer's output, pooled / up-summed outputs, the saved
//     activations the backward epilogues need) lives in LDS -- as a halo tile [H + padT + padB][W + padL + padR] pixels x (C 16-bit +
//     16 B pad) when a convolution reads it (zeroed once: the halo IS the zero padding; an up-sampled source is the half-resolution
//     plane addressed at (y >> 1, x >> 1), a concatenation is two planes, a channel window of a tensor is the same plane at a byte
//     offset), as plain [pixel][C + pad] otherwise;
//   * wave w owns the output-channel tiles w, w + 4, .. (32 rows each) of the layer and all (1 or 2) 32-pixel column tiles of the
//     image; its weights never touch LDS: the A fragment of a K-step is 16 bytes per lane straight from the packed [tap][Mpad][Ktot]
//     tensor, and the 27 fragments of the NEXT 48-channel chunk -- of the next tile or LAYER at the end -- are in flight while this
//     chunk is on the matrix cores (weights do not depend on activations: the weight stream never drains at a layer boundary);
//     B fragments are read from the planes four K-steps ahead (pinned with sched_barrier: one wave per SIMD hides nothing by itself);
//   * epilogue: registers -> the output plane (forward: bias + LeakyReLU, fp16; backward: the raw sums, bf16); a second pass over
//     the plane applies what the separate launches apply to their transposed tile (skip-gradient add, LeakyReLU' mask -- in place),
//     stores 16-byte pieces to HBM (every tensor is still written: the weight gradients and the layers outside the run read them),
//     and derives the fused outputs (Shift2d + MaxPool2d; the 2x2 sums of the up-sampled half) into their own planes.
// Round 6 -- what a lone wave per SIMD pays for, found in the disassembly and in the step's kernel trace (forward chain 61.7 -> 47.8 us,
// backward chain 104.5 -> 94.5 us in the training step; results unchanged bit for bit):
//   * every pointer comes out of the argument block, so the compiler took the accesses for FLAT ones (vmcnt AND lgkmcnt, no order it
//     can count on): each wait behind one was vmcnt(0) lgkmcnt(0) -- the K loop waited at every chunk head for the weight fragments
//     it had just requested for the NEXT chunk.  Global accesses now (address space 1): counted waits;
//   * the three chain_tile forms (4 / 2 / 1 column tiles) were arms of one loop body and the 27 weight fragments changed registers
//     on the way into and out of every work item (~150 copies): one copy of the layer body per form now (k_conv_chain);
//   * a launch finds neither its weights nor its argument block in any L2 (a step's traffic lies between two launches), and the weight
//     stream runs one chunk ahead only: the workgroups of an XCD fetch a 128-byte line each of both at kernel entry;
//   * the K loop itself is bound by the L1 path at one or two column tiles: a wave's weight fragment (1 KiB per K-step, rows 96..288 B
//     apart) is used for ONE MFMA per column tile, four waves x 1 KiB per 32 cycles = 128 B/clk against 64 -- measured 67 / 107 / 200
//     cycles per K-step at 1 / 2 / 4 column tiles (MFMA: 32 / 64 / 128).
// Results are BIT-IDENTICAL to the separate launches: every output element accumulates chunk -> tap -> K-step in the same order with
// the same MFMA instruction and k-slot assignment as k_conv's flat path, and the epilogue arithmetic is applied to the same rounded
// values in the same order (tests/test_hip_ops.py::test_conv_chain_is_bit_identical, ::test_backward_chain_is_bit_identical).
#include "common.h"
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#define CH_MAX_LAYERS 12
#define CH_MAX_PLANES 28
#define CH_MAX_LOADS 10
#define CH_THREADS 256
#define CH_TRACE 96
extern "C" void* ssdn_debug_get_trace();
#ifdef SSDN_TUNING
#define CH_ABL(c, bit) (((c).ablate & (bit)) != 0)
#else
#define CH_ABL(c, bit) false
#endif

enum { CH_CONV = 0, CH_POOL_BWD = 1 };
struct ChPlane { int off, str, roww, org, lw, lh, C; };   // LDS byte offset, pixel stride (B), pixels per row, byte offset of pixel (0,0)
struct ChAux {                                            // a saved tensor an epilogue reads: LDS plane (where = 1) or HBM (where = 2)
    ssdn_view v;
    ChPlane P;
    int where, pad_;
};
struct ChLoad { ssdn_view src; ChPlane P; unsigned npc_magic; };   // (npc_magic: of P.C / 8, as ChLayer's)
struct ChNext { const h16* w; int Ktot, npg, ts; };   // the next conv layer with an item for a wave: weights, row length, column groups, tap stride (w == NULL: none)
struct ChLayer {
    int kind;
    int M, Mpad, Ktot, c0, up0;
    ChNext nx[4];                  // per wave (chain_build): what first_from(li + 1) used to find by walking the layer table -- a chain of scalar loads
    const h16* w;
    const float* bias;
    ChPlane P0, P1, PD;            // conv: source planes (channels [0, c0) / the rest), output plane.  POOL_BWD: P0 = dpool, PD = dz
    unsigned npc_magic;            // magic reciprocal of M / 8 (16-byte pieces per pixel): exact e / npc for e * npc < 2^32
    ssdn_view dst;                 // channels [dst_c0, M) of the output go to HBM here (p == NULL: none)
    int dst_c0;
    ChAux AD;                      // backward: + skip gradient (bf16) ...
    ChAux MK;                      // ... x LeakyReLU'(saved activation);  POOL_BWD: the full-resolution activation
    int has_pool, pool_shifted;    // forward: fused Shift2d + MaxPool2d into PP / pool;  POOL_BWD: pool_shifted = shifted
    ChPlane PP;
    ssdn_view pool;
    int upsum_c;                   // backward: channels [0, upsum_c) leave as 2x2 sums x LeakyReLU'(UM) into PU / upsum
    ChPlane PU;
    ChAux UM;
    ssdn_view upsum;
    int has_pd;                    // POOL_BWD: dz is also kept as a plane
    int zero_off[2], zero_bytes[2];   // LDS ranges to clear before the layer writes its output planes (halo planes that reuse the space of dead ones)
    int grp_next;                  // ch_grp of the layer behind this one (k_conv_chain runs one copy of the layer body per group)
};
struct ChainArgs {
    int N, nloads, nlayers, lds_bytes;
    int grp0, pad0_;               // ch_grp of layer 0
    int ablate;                    // tuning aid (SSDN_CHAIN_ABLATE, -DSSDN_TUNING builds): 1 no MFMA loop, 2 no HBM stores, 4 no pool, 8 no weight stream, 16 no L2 warm-up
    unsigned tap_dy, tap_dx;       // the nine tap offsets + 4, three bits each (SGPR constants: an s_load in the K loop would drain the LDS queue)
    int bf;                        // 0: forward (fp16, bias + LeakyReLU), 1: data gradients (bf16)
    ChNext first[4];               // per wave: the first convolution with an item for it (what the kernel's entry used to find by walking ly[])
    // L2 warm-up (k_conv_chain's first instructions): the 128-byte lines of every convolution's weights.  The weight stream runs one
    // chunk ahead of the MFMAs -- 0.4 us at one column tile -- and in a training step its lines are in no L2 (Adam re-packed them a
    // step ago, ~0.5 GB of traffic since): the workgroups of an XCD fetch a line each, once, while the arena is cleared and filled
    const h16* pf_w[CH_MAX_LAYERS];
    int pf_lines[CH_MAX_LAYERS];
    ChLoad ld[CH_MAX_LOADS];
    ChLayer ly[CH_MAX_LAYERS];
};

// Every pointer of this kernel comes out of the argument block in memory, so the compiler takes it for a generic one: FLAT loads and
// stores, which count on BOTH vmcnt and lgkmcnt and return in no order it can rely on -- each wait behind one became vmcnt(0) lgkmcnt(0):
// the K loop waited for the weight fragments it had just requested for the NEXT chunk at every chunk head and again at the LDS waits of
// the first K-steps, and lds_barrier() waited for the stores to HBM.  Global accesses say so (address space 1): counted vmcnt waits.
#define CH_G __attribute__((address_space(1)))
template <typename T> static __device__ __forceinline__ T ch_ldg(const void* p) { return *(const CH_G T*)p; }
template <typename T> static __device__ __forceinline__ void ch_stg(void* p, T v) { *(CH_G T*)p = v; }

static __device__ __forceinline__ void lds_barrier() {   // this wave's LDS traffic done, then the workgroup barrier; global loads stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
static __device__ __forceinline__ unsigned ch_div(unsigned x, unsigned magic) { return magic ? __umulhi(x, magic) : x; }
static __device__ __forceinline__ char* ch_px(char* smem, const ChPlane& P, int y, int x) { return smem + P.off + P.org + (y * P.roww + x) * P.str; }

// the 27 A fragments (9 taps x 3 K-steps) of one 48-channel chunk of one 32-row tile: lane (row l31, k-half kh) reads 16 bytes
static __device__ __forceinline__ void chain_issue_w(half8 (&wr)[27], const h16* lanep, int tapstride) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) wr[t * 3 + ks] = ch_ldg<half8>(lanep + (long long)t * tapstride + ks * 16);
}

template <int N> struct ch_ic { static constexpr int value = N; };
static __host__ __device__ inline int ch_grp(int lhw) { return lhw > 7 ? 4 : lhw > 5 ? 2 : 1; }   // 32-pixel column tiles per work item

template <bool BF>
static __device__ __forceinline__ f32x16 ch_mma(half8 a, half8 b, f32x16 c) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// one 32-row output tile of one layer on this wave: K loop + register epilogue into the output plane
// what the K loops of a layer read of its descriptor: fetched as ONE batch of scalar loads at the top of the layer (k_conv_chain) -- field by
// field, where each was used, the way to a layer's first MFMA led through ~25 scalar-load round trips in sequence (~4 K cycles per layer)
struct ChHot {
    int M, Mpad, Ktot, c0, up0;
    const h16* w;
    const float* bias;
    ChPlane P0, P1, PD;
    unsigned tap_dy, tap_dx;
};
template <int NPT, bool BF, typename STAMP>
static __device__ __forceinline__ void chain_tile(const ChainArgs& c, const ChHot& L, char* smem, int mt, int pg, int l31, int kh,
                                                  half8 (&wr)[27], const h16* next_lanep, int next_tapstride, STAMP stamp) {
    const ChPlane P0 = L.P0, P1 = L.P1, PD = L.PD;
    const int HWp = 1 << (PD.lw + PD.lh);
    // forward: bias of this lane's 16 output rows, on its way while the K loop runs
    float bb[4][4];
    if constexpr (!BF) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int m0 = mt * 32 + gq * 8 + kh * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) bb[gq][j] = ch_ldg<float>(L.bias + (m0 < L.M ? m0 + j : 0));
        }
    }
    int py[NPT], px[NPT];
#pragma unroll
    for (int p = 0; p < NPT; ++p) {
        int q = (pg * NPT + p) * 32 + l31;
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
    __builtin_assume(nch >= 1);
    const int tapstride = L.Mpad * L.Ktot;
    const h16* lanep = L.w + (long long)(mt * 32 + l31) * L.Ktot + kh * 8;
    for (int ch = 0; ch < nch && !CH_ABL(c, 1); ++ch) {
        const int k0 = ch * 48;
        const bool from0 = k0 < L.c0;
        const int sh = from0 ? L.up0 : 0;
        const int roww = from0 ? P0.roww : P1.roww, str = from0 ? P0.str : P1.str;
        const int cbase = (from0 ? P0.off + P0.org + k0 * 2 : P1.off + P1.org + (k0 - L.c0) * 2) + kh * 16;
        // the stream runs one chunk ahead: chunk ch+1 of this tile, or chunk 0 of the next tile / layer this wave works on (always a
        // valid address: the last chunk of the run re-loads itself -- unconditional loads keep the compiler's vmcnt accounting exact)
        const h16* np = ch + 1 < nch ? lanep + (ch + 1) * 48 : next_lanep;
        const int nts = ch + 1 < nch ? tapstride : next_tapstride;
        int boff[9][NPT];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = (int)((L.tap_dy >> (3 * t)) & 7u) - 4, dx = (int)((L.tap_dx >> (3 * t)) & 7u) - 4;
#pragma unroll
            for (int p = 0; p < NPT; ++p)
                boff[t][p] = cbase + (((py[p] + dy) >> sh) * roww + ((px[p] + dx) >> sh)) * str;
        }
        // B fragments are read BD K-steps ahead of the MFMA that consumes them (one wave per SIMD: nothing else hides the LDS latency)
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
#pragma unroll
            for (int p = 0; p < NPT; ++p) acc[p] = ch_mma<BF>(a, bq[s % BD][p], acc[p]);
            // (behind the MFMAs that read the old fragment: its register takes the new one -- issued ahead of them, the 27 new fragments
            //  land in registers of their own and are copied over, behind a vmcnt(0), at the end of every chunk)
            if (!CH_ABL(c, 8)) wr[s] = ch_ldg<half8>(np + (long long)(s / 3) * nts + (s % 3) * 16);
            if (s + BD < 27) rd(s + BD, s % BD);
            __builtin_amdgcn_sched_barrier(0);                 // pin the step: left alone, the scheduler sinks every read to its MFMA
        }
        stamp();
    }
    // epilogue -> the output plane (rows >= M of a padded tile are not stored).  Forward: bias + LeakyReLU, fp16; backward: bf16 sums
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const int m0 = mt * 32 + gq * 8 + kh * 4;
        if (m0 >= L.M) continue;
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
            if ((pg * NPT + p) * 32 + l31 >= HWp) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = acc[p][gq * 4 + j];
                if constexpr (!BF) v[j] = lrelu(v[j] + bb[gq][j]);
            }
            u32x2_t o;
            o[0] = BF ? pack_bf16x2(v[0], v[1]) : pack_f16x2(v[0], v[1]);
            o[1] = BF ? pack_bf16x2(v[2], v[3]) : pack_f16x2(v[2], v[3]);
            *reinterpret_cast<u32x2_t*>(ch_px(smem, PD, py[p], px[p]) + m0 * 2) = o;
        }
    }
}

// two 16-bit channels per instruction on their raw patterns (the max-pool backward pass)
static __device__ __forceinline__ unsigned ch_pk_add_u16(unsigned a, unsigned b) { unsigned r; asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
static __device__ __forceinline__ unsigned ch_pk_max_f16(unsigned a, unsigned b) { unsigned r; asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
static __device__ __forceinline__ unsigned ch_pk_nonzero(unsigned a) {      // 0xffff per half that is not 0
    unsigned t, r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(a), "s"(0x00010001u));
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(t), "s"(0xffffffffu));
    return r;
}
static __device__ __forceinline__ unsigned ch_pk_positive(unsigned a) {     // 0xffff per half whose fp16 value is > 0 (as a signed integer: > 0)
    unsigned t, u, r;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(t) : "v"(a));
    asm("v_pk_min_i16 %0, %1, %2" : "=v"(u) : "v"(t), "s"(0x00010001u));
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(u), "s"(0xffffffffu));
    return r;
}
// 16-byte piece cc of pixel q (row-major at the tensor's own resolution) of a saved tensor
static __device__ __forceinline__ u32x4_t ch_aux(const ChAux& A, const char* smem, int n, int q, int cc) {
    if (A.where == 1) return *reinterpret_cast<const u32x4_t*>(smem + A.P.off + A.P.org + ((q >> A.P.lw) * A.P.roww + (q & ((1 << A.P.lw) - 1))) * A.P.str + cc * 16);
    return ch_ldg<u32x4_t>((const h16*)A.v.p + A.v.co + ((long long)(n << (A.P.lw + A.P.lh)) + q) * A.v.cs + cc * 8);
}

// the same without a branch: BOTH an LDS read (address 0 of the arena when the tensor is not a plane) and a buffer load (out of range -- zeros,
// no memory traffic -- when it is not in HBM) are issued and the result is picked by the wave-uniform `where`: with the branch every fetch sits
// in its own basic block, and the four window fetches of a max-pool entry are four exposed round trips in sequence
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t ch_aux_rsrc(const ChAux& A) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(A.v.p), 0, A.where == 2 ? (int)0x80000000 : 0, SSDN_BUFFER_RSRC_FLAGS);
}
static __device__ __forceinline__ u32x4_t ch_aux_nb(const ChAux& A, __amdgpu_buffer_rsrc_t rs, const char* smem, int n, int q, int cc) {
    const int lofs = A.where == 1 ? A.P.off + A.P.org + ((q >> A.P.lw) * A.P.roww + (q & ((1 << A.P.lw) - 1))) * A.P.str + cc * 16 : 0;
    const u32x4_t l = *reinterpret_cast<const u32x4_t*>(smem + lofs);
    const int gofs = (((n << (A.P.lw + A.P.lh)) + q) * A.v.cs + A.v.co + cc * 8) * 2;      // (< 2^31: chain_build checks the tensors' sizes)
    const u32x4_t g = __builtin_amdgcn_raw_buffer_load_b128(rs, A.where == 2 ? gofs : (int)0x80000000, 0, 0);
    return A.where == 2 ? g : l;
}

template <bool BF>
__global__ __launch_bounds__(CH_THREADS) void k_conv_chain(const ChainArgs* __restrict__ cp, unsigned long long* trace) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ChainArgs& c = *cp;
    const int n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    [[maybe_unused]] int tr_i = 0;
    auto stamp = [&]() {
#ifdef SSDN_TUNING
        if (trace && tid == 0 && tr_i < CH_TRACE) trace[(size_t)blockIdx.x * CH_TRACE + tr_i++] = __builtin_amdgcn_s_memtime();
#endif
    };
    stamp();
    // everything the kernel's entry needs sits at the head of the argument block and is fetched as ONE batch of scalar loads (pinned by the
    // empty asm): a launch finds none of these lines in its L2 (~2 K cycles per round trip), and field by field -- the layer count, the
    // walk to the wave's first convolution, its weights' address, the tables of the warm-up -- they were seven round trips in sequence
    int nlayers = c.nlayers, nloads = c.nloads, lds_bytes = c.lds_bytes, grp0 = c.grp0, nimg = c.N;
    unsigned tap_dy = c.tap_dy, tap_dx = c.tap_dx;
    ChNext F = c.first[wave];
    const h16* pfw[CH_MAX_LAYERS];
    int pfl[CH_MAX_LAYERS];
#pragma unroll
    for (int i = 0; i < CH_MAX_LAYERS; ++i) { pfw[i] = c.pf_w[i]; pfl[i] = c.pf_lines[i]; }
    static_assert(CH_MAX_LAYERS == 12, "the pins below name every table slot");
    asm volatile("" : "+s"(nlayers), "+s"(nloads), "+s"(lds_bytes), "+s"(grp0), "+s"(nimg), "+s"(tap_dy), "+s"(tap_dx), "+s"(F.w), "+s"(F.Ktot),
                 "+s"(F.npg), "+s"(F.ts), "+s"(pfl[0]), "+s"(pfl[1]), "+s"(pfl[2]), "+s"(pfl[3]), "+s"(pfl[4]), "+s"(pfl[5]), "+s"(pfl[6]),
                 "+s"(pfl[7]), "+s"(pfl[8]), "+s"(pfl[9]), "+s"(pfl[10]), "+s"(pfl[11]), "+s"(pfw[0]), "+s"(pfw[1]), "+s"(pfw[2]));
    asm volatile("" : "+s"(pfw[3]), "+s"(pfw[4]), "+s"(pfw[5]), "+s"(pfw[6]), "+s"(pfw[7]), "+s"(pfw[8]), "+s"(pfw[9]), "+s"(pfw[10]), "+s"(pfw[11]));
    half8 wr[27];
    if (F.w) chain_issue_w(wr, F.w + (long long)((wave / F.npg) * 32 + l31) * F.Ktot + kh * 8, F.ts);
    // L2 warm-up: workgroup n runs on XCD n % 8; the (N + 7) / 8 workgroups of an XCD share the lines of each layer, two loads per thread
    // and layer at most (buffer loads: a line past the end of a layer, or a table slot without one, moves nothing) -- no branches, all
    // in flight together
    unsigned pf = 0;
#ifndef CH_NO_WARMUP
    if (!CH_ABL(c, 16)) {
        const int per = (nimg + 7) >> 3, ln0 = (n >> 3) + per * tid;
#pragma unroll
        for (int i = 0; i < CH_MAX_LAYERS; ++i) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(pfw[i]), 0, pfl[i] << 7, SSDN_BUFFER_RSRC_FLAGS);
#pragma unroll
            for (int j = 0; j < 2; ++j) pf ^= __builtin_amdgcn_raw_buffer_load_b32(rs, (ln0 + j * per * CH_THREADS) << 7, 0, 0);
        }
#ifndef CH_NO_ARGWARM
        // ... and the argument block itself: every layer starts with a batch of scalar loads from it, and a step's traffic away from
        // the last launch its lines are in no L2 either (the scalar cache fills through the L2)
        const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<ChainArgs*>(cp), 0, (int)sizeof(ChainArgs), SSDN_BUFFER_RSRC_FLAGS);
        pf ^= __builtin_amdgcn_raw_buffer_load_b32(rs_c, tid << 7, 0, 0);
#endif
    }
#endif
    stamp();
    for (int z = tid * 16; z < lds_bytes; z += CH_THREADS * 16) *reinterpret_cast<half8*>(smem + z) = zero_h8();
    lds_barrier();
    stamp();
    for (int i = 0; i < nloads; ++i) {
        const ChPlane P = c.ld[i].P;
        const int npc = P.C >> 3, total = npc << (P.lw + P.lh);
        const unsigned magic = c.ld[i].npc_magic;
        const h16* src = (const h16*)c.ld[i].src.p + c.ld[i].src.co;
        // four entries of a thread in flight (one at a time, every entry was a round trip to HBM of its own: up to six in sequence per
        // plane).  Measured and not kept: the first entries of EVERY plane requested ahead of the clearing loop through branch-free
        // buffer loads over the whole table (~8 K cycles of address arithmetic on a lone wave per SIMD, more than the round trips
        // saved), and four planes at a time (backward chain -1 us, forward chain -- one plane -- +2.5 us in the step)
        constexpr int LB = 4;
        for (int e0 = tid; e0 < total; e0 += CH_THREADS * LB) {
            half8 v[LB];
            int lo[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int e = e0 + u * CH_THREADS;
                const int ec = e < total ? e : e0;               // (past the end: a valid entry again, not written)
                const int q = ch_div(ec, magic), cc = ec - q * npc;
                v[u] = ch_ldg<half8>(src + ((long long)(n << (P.lw + P.lh)) + q) * c.ld[i].src.cs + cc * 8);
                lo[u] = e < total ? P.off + P.org + ((q >> P.lw) * P.roww + (q & ((1 << P.lw) - 1))) * P.str + cc * 16 : -1;
            }
#pragma unroll
            for (int u = 0; u < LB; ++u)
                if (lo[u] >= 0) *reinterpret_cast<half8*>(smem + lo[u]) = v[u];
        }
    }
    stamp();
    asm volatile("" :: "v"(pf));                               // (the warm-up loads end here, with the arena's)
    lds_barrier();
    stamp();
    // One copy of the layer body per column-tile count (ch_grp: 4 / 2 / 1 tiles of 32 pixels per item), each run over the consecutive layers of
    // its group.  With the three chain_tile forms as arms of one loop body the 27 weight fragments -- live from item to item, across
    // layers -- sat in different registers in every arm: ~150 register copies on the way into and out of EVERY item (half as many cycles
    // as the item's MFMAs at one column tile).  Now they change registers where the image size changes: a handful of times per launch.
    auto layer = [&](const int li, auto nptc) __attribute__((always_inline)) -> int {
        constexpr int NPT = decltype(nptc)::value;
        const ChLayer& L = c.ly[li];
        // ONE batch of scalar loads for everything up to the layer's last MFMA (the empty asm pins the values here: left alone, every field
        // is fetched where it is first used, behind a wait of its own)
        ChHot H;
        H.M = L.M; H.Mpad = L.Mpad; H.Ktot = L.Ktot; H.c0 = L.c0; H.up0 = L.up0; H.w = L.w; H.bias = L.bias;
        H.P0 = L.P0; H.P1 = L.P1; H.PD = L.PD; H.tap_dy = tap_dy; H.tap_dx = tap_dx;
        int kind = L.kind, zb0 = L.zero_bytes[0], zb1 = L.zero_bytes[1], zo0 = L.zero_off[0], zo1 = L.zero_off[1];
        ChNext NX = L.nx[wave];
        int grp_next = L.grp_next;
        asm volatile("" : "+s"(H.M), "+s"(H.Mpad), "+s"(H.Ktot), "+s"(H.c0), "+s"(H.up0), "+s"(H.w), "+s"(H.bias), "+s"(kind), "+s"(zb0), "+s"(zb1),
                     "+s"(zo0), "+s"(zo1), "+s"(NX.w), "+s"(NX.Ktot), "+s"(NX.npg), "+s"(NX.ts), "+s"(grp_next));
        asm volatile("" : "+s"(H.P0.off), "+s"(H.P0.str), "+s"(H.P0.roww), "+s"(H.P0.org), "+s"(H.P1.off), "+s"(H.P1.str), "+s"(H.P1.roww), "+s"(H.P1.org),
                     "+s"(H.PD.off), "+s"(H.PD.str), "+s"(H.PD.roww), "+s"(H.PD.org), "+s"(H.PD.lw), "+s"(H.PD.lh));
        const ChPlane PD = H.PD;
        const int lhw = PD.lw + PD.lh;
        const int npc = H.M >> 3;
        if (zb0) {                                             // output planes of this layer lie where dead planes were: clear (halo = 0)
            for (int z = tid * 16; z < zb0; z += CH_THREADS * 16) *reinterpret_cast<half8*>(smem + zo0 + z) = zero_h8();
            for (int z = tid * 16; z < zb1; z += CH_THREADS * 16) *reinterpret_cast<half8*>(smem + zo1 + z) = zero_h8();
            lds_barrier();
        }
        if (kind == CH_CONV) {
            constexpr int npg = NPT == 4 ? 2 : 1;
            const int nitems = (H.Mpad >> 5) * npg;
            for (int it = wave; it < nitems; it += 4) {
                const int nit = it + 4;
                // the weight stream's next stop: this wave's next item of the layer, else its first item of the next layer that has one
                // (ChLayer.nx), else -- the run's last chunk -- itself once more
                const h16* nlp;
                int nts;
                if (nit < nitems) { nlp = H.w + (long long)((nit / npg) * 32 + l31) * H.Ktot + kh * 8; nts = H.Mpad * H.Ktot; }
                else if (NX.w) { nlp = NX.w + (long long)((wave / NX.npg) * 32 + l31) * NX.Ktot + kh * 8; nts = NX.ts; }
                else { nlp = H.w + (long long)((it / npg) * 32 + l31) * H.Ktot + kh * 8 + (H.Ktot - 48); nts = H.Mpad * H.Ktot; }
                const int mt = it / npg, pg = it - mt * npg;
                chain_tile<NPT, BF>(c, H, smem, mt, pg, l31, kh, wr, nlp, nts, stamp);
            }                                                  // (an idle wave keeps the chunk it holds for its next layer)
            stamp();
            lds_barrier();
            stamp();
            // ---- second pass over the output plane: [+ skip gradient] [x LeakyReLU'] in place, 16-byte pieces of consecutive pixels -> HBM ----
            {
                const int total = npc << lhw;
                h16* dst = (h16*)L.dst.p + L.dst.co;
                const bool fix = BF && (L.AD.where || L.MK.where);
                const bool far = fix && (L.AD.where == 2 || L.MK.where == 2);
                if constexpr (BF) {
                    if (far) {
                        // a saved tensor in HBM (images of more than 64 pixels): the pieces of three entries in flight together, fetched
                        // without a branch (ch_aux_nb) -- one entry at a time, each was a round trip to HBM in sequence, six per thread in
                        // the 16x16 layer (10.3 K -> 8.9 K cycles).  Saved tensors that are LDS planes keep the plain loop below: there
                        // the branch-free fetch's idle buffer load is the longer wait (measured: +40 %)
                        const __amdgpu_buffer_rsrc_t rs_ad = ch_aux_rsrc(L.AD), rs_mk = ch_aux_rsrc(L.MK);
                        const bool has_ad = L.AD.where != 0, has_mk = L.MK.where != 0;
                        constexpr int SB = 3;
                        for (int e0 = tid; e0 < total; e0 += CH_THREADS * SB) {
                            u32x4_t ab[SB], mb[SB];
#pragma unroll
                            for (int u = 0; u < SB; ++u) {
                                const int e = e0 + u * CH_THREADS < total ? e0 + u * CH_THREADS : e0;
                                const int q = ch_div(e, L.npc_magic), cc = e - q * npc;
                                ab[u] = ch_aux_nb(L.AD, rs_ad, smem, n, q, cc);
                                mb[u] = ch_aux_nb(L.MK, rs_mk, smem, n, q, cc);
                            }
#pragma unroll
                            for (int u = 0; u < SB; ++u) {
                                const int e = e0 + u * CH_THREADS;
                                if (e >= total) break;
                                const int q = ch_div(e, L.npc_magic), cc = e - q * npc;
                                char* pp = ch_px(smem, PD, q >> PD.lw, q & ((1 << PD.lw) - 1)) + cc * 16;
                                u32x4_t o = *reinterpret_cast<const u32x4_t*>(pp);
#pragma unroll
                                for (int w = 0; w < 4; ++w) {
                                    float v0 = bf_lo(o[w]) + (has_ad ? bf_lo(ab[u][w]) : 0.f);
                                    float v1 = bf_hi(o[w]) + (has_ad ? bf_hi(ab[u][w]) : 0.f);
                                    if (has_mk) {
                                        v0 *= lrelu_grad(f16_lo(mb[u][w]));
                                        v1 *= lrelu_grad(f16_hi(mb[u][w]));
                                    }
                                    o[w] = pack_bf16x2(v0, v1);
                                }
                                *reinterpret_cast<u32x4_t*>(pp) = o;
                                if (cc * 8 >= L.dst_c0 && L.dst.p && !CH_ABL(c, 2))
                                    ch_stg<u32x4_t>(dst + ((long long)(n << lhw) + q) * L.dst.cs + cc * 8, o);
                            }
                        }
                    }
                }
                for (int e = tid; e < total && !far; e += CH_THREADS) {
                    const int q = ch_div(e, L.npc_magic), cc = e - q * npc;
                    char* pp = ch_px(smem, PD, q >> PD.lw, q & ((1 << PD.lw) - 1)) + cc * 16;
                    u32x4_t o = *reinterpret_cast<const u32x4_t*>(pp);
                    if constexpr (BF) {
                        if (fix) {
                            u32x4_t ab = {0u, 0u, 0u, 0u}, mb = {0u, 0u, 0u, 0u};
                            if (L.AD.where) ab = ch_aux(L.AD, smem, n, q, cc);
                            if (L.MK.where) mb = ch_aux(L.MK, smem, n, q, cc);
#pragma unroll
                            for (int w = 0; w < 4; ++w) {
                                float v0 = bf_lo(o[w]) + (L.AD.where ? bf_lo(ab[w]) : 0.f);
                                float v1 = bf_hi(o[w]) + (L.AD.where ? bf_hi(ab[w]) : 0.f);
                                if (L.MK.where) {
                                    v0 *= lrelu_grad(f16_lo(mb[w]));
                                    v1 *= lrelu_grad(f16_hi(mb[w]));
                                }
                                o[w] = pack_bf16x2(v0, v1);
                            }
                            *reinterpret_cast<u32x4_t*>(pp) = o;
                        }
                    }
                    if (cc * 8 >= L.dst_c0 && L.dst.p && !CH_ABL(c, 2))
                        ch_stg<u32x4_t>(dst + ((long long)(n << lhw) + q) * L.dst.cs + cc * 8, o);
                }
            }
            stamp();
            bool wrote = BF && (L.AD.where || L.MK.where);
            if constexpr (BF) {
                if (L.upsum_c > 0) {
                    // fused SSDN_OP_UPSUM_BWD: 2x2 sums (scan order, fp32) of the bf16 values of channels [0, upsum_c), x LeakyReLU'(UM)
                    const ChPlane PU = L.PU;
                    const int upc = L.upsum_c >> 3, total = upc << (lhw - 2);
                    h16* dst = (h16*)L.upsum.p + L.upsum.co;
                    for (int e = tid; e < total; e += CH_THREADS) {
                        const int pq = e / upc, cc = e - pq * upc;
                        const int pj = pq & ((1 << PU.lw) - 1), pi = pq >> PU.lw;
                        const u32x4_t um = ch_aux(L.UM, smem, n, pq, cc);
                        float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const u32x4_t o = *reinterpret_cast<const u32x4_t*>(ch_px(smem, PD, 2 * pi + (q4 >> 1), 2 * pj + (q4 & 1)) + cc * 16);
#pragma unroll
                            for (int q = 0; q < 4; ++q) { sum[2 * q] += bf_lo(o[q]); sum[2 * q + 1] += bf_hi(o[q]); }
                        }
                        u32x4_t r;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            r[q] = pack_bf16x2(sum[2 * q] * lrelu_grad(f16_lo(um[q])), sum[2 * q + 1] * lrelu_grad(f16_hi(um[q])));
                        *reinterpret_cast<u32x4_t*>(ch_px(smem, PU, pi, pj) + cc * 16) = r;
                        if (!CH_ABL(c, 2)) ch_stg<u32x4_t>(dst + ((long long)(n << (lhw - 2)) + pq) * L.upsum.cs + cc * 8, r);
                    }
                    wrote = true;
                }
            } else {
                if (L.has_pool && !CH_ABL(c, 4)) {
                    // fused Shift2d((1,0)) + MaxPool2d(2): rows {2i-1, 2i} of the rounded values, row -1 is a literal 0 in the max
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
                                if (r >= 0) v = *reinterpret_cast<const u32x4_t*>(ch_px(smem, PD, r, 2 * pj + dc) + cc * 16);
                                if (!have) { best = v; have = true; }
                                else {
                                    const half8 m = __builtin_elementwise_max(__builtin_bit_cast(half8, best), __builtin_bit_cast(half8, v));
                                    best = __builtin_bit_cast(u32x4_t, m);
                                }
                            }
                        }
                        *reinterpret_cast<u32x4_t*>(ch_px(smem, PP, pi, pj) + cc * 16) = best;
                        if (!CH_ABL(c, 2)) ch_stg<u32x4_t>(dst + ((long long)(n << (lhw - 2)) + pq) * L.pool.cs + cc * 8, best);
                    }
                    wrote = true;
                }
            }
            if (wrote) lds_barrier();
            stamp();
        } else if constexpr (BF) {      // (compiled into the backward kernel only)
            // ---- SSDN_OP_POOL_BWD: route dpool (P0, pooled resolution) to the window position that held the max, x LeakyReLU' ----
            const ChPlane PQ = L.P0;
            const int Wo = 1 << PQ.lw, Ho = 1 << PQ.lh, H = 2 * Ho;
            const int total = npc << (PQ.lw + PQ.lh);
            unsigned short* dz = (unsigned short*)L.dst.p + L.dst.co;
            const int lw = PQ.lw + 1, lhwf = PQ.lw + PQ.lh + 2;
            // Round 5: this pass was 47 of the backward chain's 115 us -- not its fetches (an entry's four 16-byte fetches were exposed round
            // trips in sequence, but batching them changed nothing) and not cold code (a second run of the pass took as long): ~2800 scalar-style
            // instructions per entry (fp16 -> fp32 converts, canonicalising maxima, per-channel compares / selects / bool bookkeeping) on a
            // single wave per SIMD.  Now two channels per instruction on the raw 16-bit patterns: ~300 per entry, same results bit for bit
            // (the fp16 maxima and equalities are exact in either width; -0 is folded into +0 first, as == and > treat it).
            const __amdgpu_buffer_rsrc_t rs_mk = ch_aux_rsrc(L.MK);
            constexpr int PB = 3;                                // entries of a thread whose window fetches are in flight together
            for (int e0 = tid; e0 < total; e0 += CH_THREADS * PB) {
              u32x4_t vv[PB][4];
#pragma unroll
              for (int u = 0; u < PB; ++u) {
                const int e = e0 + u * CH_THREADS < total ? e0 + u * CH_THREADS : tid;      // (past the end: a valid entry again, not used)
                const int pq = ch_div(e, L.npc_magic), cc = e - pq * npc;
                const int j = pq & (Wo - 1), i = pq >> PQ.lw;
                const int r0 = L.pool_shifted ? 2 * i - 1 : 2 * i;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int y = r0 + (k >> 1);
                    vv[u][k] = ch_aux_nb(L.MK, rs_mk, smem, n, ((y < 0 ? 0 : y) << lw) + 2 * j + (k & 1), cc);      // (y < 0: fetched, not looked at)
                }
              }
#pragma unroll
              for (int u = 0; u < PB; ++u) {
                const int e = e0 + u * CH_THREADS;
                if (e >= total) break;
                const int pq = ch_div(e, L.npc_magic), cc = e - pq * npc;
                const int j = pq & (Wo - 1), i = pq >> PQ.lw;
                const u32x4_t g = *reinterpret_cast<const u32x4_t*>(ch_px(smem, PQ, i, j) + cc * 16);
                const int r0 = L.pool_shifted ? 2 * i - 1 : 2 * i;
                const bool top = L.pool_shifted && r0 < 0;      // row -1 of a shifted window: the literal 0 of the forward max, scanned first
                const unsigned topm = top ? 0xffffffffu : 0u;
                const u32x4_t (&v)[4] = vv[u];
                u32x4_t o[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    // (no branches on `top`, which differs lane by lane: the pad row enters as two zeros, which change nothing -- the maximum
                    //  starts at 0 then, and a zero never takes the gradient: `avail` is empty when the maximum is 0)
                    unsigned x[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        x[k] = v[k][d] & ch_pk_nonzero(ch_pk_add_u16(v[k][d], v[k][d]));      // -0 -> +0
                        if (k < 2) x[k] &= ~topm;
                    }
                    unsigned m = top ? 0u : 0xfbfffbffu;                      // -65504 | -65504
#pragma unroll
                    for (int k = 0; k < 4; ++k) m = ch_pk_max_f16(m, x[k]);
                    unsigned avail = ch_pk_nonzero(m) | ~topm;               // the pad row holds the maximum (0): the gradient is dropped
                    // the gradient times LeakyReLU'(activation): exact for a positive activation, x slope (rounded) otherwise
                    const unsigned gs = pack_bf16x2(bf_lo(g[d]) * LRELU_SLOPE, bf_hi(g[d]) * LRELU_SLOPE);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned ne = ch_pk_nonzero(x[k] ^ m);         // 0xffff where the position does NOT hold the maximum
                        const unsigned hit = ~ne & avail;                    // ... the first that does takes the gradient
                        avail &= ne;
                        const unsigned pos = ch_pk_positive(x[k]);
                        o[k][d] = ((g[d] & pos) | (gs & ~pos)) & hit;
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int y = r0 + (k >> 1);
                    if (y < 0) continue;
                    const int x = 2 * j + (k & 1);
                    if (L.has_pd) *reinterpret_cast<u32x4_t*>(ch_px(smem, PD, y, x) + cc * 16) = o[k];
                    if (!CH_ABL(c, 2)) ch_stg<u32x4_t>(dz + ((long long)(n << lhwf) + (y << lw) + x) * L.dst.cs + cc * 8, o[k]);
                }
                if (L.pool_shifted && i == Ho - 1) {            // shifted pooling never looks at the last row: its gradient is zero
                    const u16x8 z = zero_b8();
#pragma unroll
                    for (int dc = 0; dc < 2; ++dc) {
                        const int x = 2 * j + dc;
                        if (L.has_pd) *reinterpret_cast<u16x8*>(ch_px(smem, PD, H - 1, x) + cc * 16) = z;
                        if (!CH_ABL(c, 2)) ch_stg<u16x8>(dz + ((long long)(n << lhwf) + ((H - 1) << lw) + x) * L.dst.cs + cc * 8, z);
                    }
                }
              }
            }
            if (L.has_pd) lds_barrier();
            stamp();
        }
        return grp_next;
    };
    int li = 0, g = grp0;
    while (li < nlayers) {
        if (g == 4) do g = layer(li++, ch_ic<4>{}); while (li < nlayers && g == 4);
        else if (g == 2) do g = layer(li++, ch_ic<2>{}); while (li < nlayers && g == 2);
        else do g = layer(li++, ch_ic<1>{}); while (li < nlayers && g == 1);
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
static bool g_chain_on = true;
static int g_chain_max_px = 256;          // largest image (pixels) a chained launch takes (ssdn_conv_set_chain(2): 64 -- A/B aid)
extern "C" int ssdn_conv_set_chain(int on) { g_chain_on = on != 0; g_chain_max_px = on == 2 ? 64 : 256; return 0; }
bool chain_merging_on() { return g_chain_on; }

static int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

namespace {
struct PlaneRec {
    ssdn_view v;            // the HBM tensor window the plane mirrors: channels [v.co, v.co + C)
    int H, W, C;
    bool produced;          // written by a layer of the run (else: loaded from HBM at the start)
    bool halo;              // read by a convolution
    int hidden_c;           // channels [0, hidden_c) never reach HBM in this form (raw sums of an up-summed half): no reader may see them
    int off;
    int first, last;        // layers that create (-1: loaded at the start) / last read the plane
    int bytes;
};
struct PlaneRef { int idx = -1; int coff = 0; };   // plane + channel offset of the window inside it
struct AuxRef { int where = 0; PlaneRef pl; ssdn_view v{nullptr, 0, 0}; int H = 1, W = 1; };
struct LayerRec {
    int kind;
    const ssdn_conv_args* a;
    const ssdn_pool_args* pa;
    PlaneRef p0, p1, pd, pp, pu;
    AuxRef mk, um, ad;
};
struct ChainBuild {
    std::vector<PlaneRec> planes;
    std::vector<LayerRec> layers;
    int nloads = 0;
    int cur = 0;            // layer being built (plane lifetimes)
    int padT = 0, padB = 0, padL = 0, padR = 0;
};
}  // namespace

static int cb_add_plane(ChainBuild& b, const ssdn_view& v, int H, int W, int C, bool produced, int hidden_c = 0) {
    if ((int)b.planes.size() >= CH_MAX_PLANES) return -1;
    b.planes.push_back(PlaneRec{v, H, W, C, produced, false, hidden_c, 0, produced ? b.cur : -1, b.cur, 0});
    return (int)b.planes.size() - 1;
}
// the plane (+ channel offset) that holds channels [v.co, v.co + C) of tensor v at H x W: an existing one, a new load from HBM, or
// idx = -2 if the tensor is written inside the run in a shape / window this reader does not match (the run must end before it)
static PlaneRef cb_source(ChainBuild& b, const ssdn_view& v, int H, int W, int C, bool may_load = true) {
    PlaneRef r;
    for (int i = 0; i < (int)b.planes.size(); ++i) {
        const PlaneRec& P = b.planes[i];
        if (P.v.p != v.p) continue;
        const int rel = v.co - P.v.co;
        if (P.v.cs == v.cs && P.H == H && P.W == W && rel >= P.hidden_c && rel + C <= P.C && rel >= 0) {
            r.idx = i; r.coff = rel;
            b.planes[i].last = b.cur;
            return r;
        }
        if (P.produced) { r.idx = -2; return r; }
    }
    r.idx = -2;
    if (!may_load || b.nloads >= CH_MAX_LOADS) return r;
    const int i = cb_add_plane(b, v, H, W, C, false);
    if (i < 0) return r;
    ++b.nloads;
    r.idx = i;
    return r;
}
// a saved tensor an epilogue reads: an LDS plane when the image has <= 64 pixels (loaded at the start, or produced by the run), else HBM
static bool cb_aux(ChainBuild& b, const ssdn_view& v, int H, int W, int C, AuxRef* out) {
    out->v = v; out->H = H; out->W = W;
    for (const PlaneRec& P : b.planes)
        if (P.v.p == v.p && P.produced) {
            const PlaneRef r = cb_source(b, v, H, W, C, false);
            if (r.idx < 0) return false;
            out->where = 1; out->pl = r;
            return true;
        }
    if (H * W <= 64) {
        const PlaneRef r = cb_source(b, v, H, W, C);
        if (r.idx >= 0) { out->where = 1; out->pl = r; return true; }
    }
    out->where = 2;
    return true;
}
static bool cb_written(const ChainBuild& b, const void* p) {
    for (const PlaneRec& P : b.planes)
        if (P.v.p == p) return true;
    return false;
}

static bool chain_conv_ok(const ssdn_conv_args* a, const ssdn_conv_args* first) {
    if (conv_validate(a)) return false;
    if (a->bf16 != first->bf16 || a->ntaps != 9 || a->dst32 || a->unrot.p) return false;
    if (a->kc != 48 || a->Ktot % 48 || a->c0 % 48 || a->c1 % 48 || a->Ktot <= 0) return false;
    if (ilog2_exact(a->H) < 0 || ilog2_exact(a->W) < 0 || a->H * a->W > g_chain_max_px || a->H * a->W == 128) return false;
    if (a->H * a->W > 64 && (a->Mpad > 64 || a->Ktot > 48 || a->upsum.p)) return false;   // 256 pixels: only the thin (48-channel) layers pay
    if ((a->M & 7) || a->Mpad > 32 * 8 || a->N != first->N) return false;
    if (a->up0 && (a->c0 == 0 || ((a->H | a->W) & 1))) return false;
    if ((a->src0.cs & 7) || (a->src0.co & 7) || (a->c1 && ((a->src1.cs & 7) || (a->src1.co & 7)))) return false;
    for (int t = 0; t < 9; ++t)
        if (a->dy[t] != first->dy[t] || a->dx[t] != first->dx[t]) return false;
    if (!a->bf16) {
        if (!a->dst.p || !a->act || !a->bias || a->mask.p || a->add.p || a->upsum.p) return false;
        if (a->pool.p && (((a->H | a->W) & 1) || (a->pool.cs & 7) || (a->pool.co & 7))) return false;
    } else {
        if (a->act || a->bias || a->pool.p || a->c1 || a->up0) return false;
        if (a->upsum.p) {
            if (a->mask.p || a->add.p || ((a->H | a->W) & 1) || (a->upsum_c & 7) || a->upsum_c <= 0 || a->upsum_c > a->M) return false;
            if ((a->upsum.cs & 7) || (a->upsum.co & 7) || (a->upsum_mask.cs & 7) || (a->upsum_mask.co & 7) || !a->upsum_mask.p) return false;
            if (a->upsum_c < a->M && !a->dst.p) return false;
        } else if (!a->dst.p) return false;
    }
    return true;
}
static bool chain_pool_ok(const ssdn_pool_args* a, int N) {
    if ((a->C & 7) || (a->H & 1) || (a->W & 1) || a->N != N) return false;
    if (ilog2_exact(a->H) < 0 || ilog2_exact(a->W) < 0 || (a->H / 2) * (a->W / 2) > g_chain_max_px) return false;
    for (const ssdn_view* v : {&a->act, &a->dpool, &a->dz})
        if (!v->p || (v->cs & 7) || (v->co & 7)) return false;
    return true;
}

// try to run exactly items[0..n) as one launch; on success fills *out (LDS offsets assigned)
// (tuning builds: SSDN_CHAIN_DEBUG=1 says which rule ended a candidate run)
#define CH_FAIL(code) do { if (ssdn_tuning_env("SSDN_CHAIN_DEBUG")) fprintf(stderr, "chain_build(%d ops): rule %d at op %d\n", n, code, dbg_i); return false; } while (0)
static bool chain_build(const ssdn_op* items, int n, ChainArgs* out) {
    int dbg_i = -1;
    if (n < 2 || n > CH_MAX_LAYERS || items[0].type != SSDN_OP_CONV) CH_FAIL(1);
    const ssdn_conv_args* f = (const ssdn_conv_args*)items[0].args;
    ChainBuild b;
    int mny = 0, mxy = 0, mnx = 0, mxx = 0;
    unsigned tdy = 0, tdx = 0;
    for (int t = 0; t < 9; ++t) {
        if (f->dy[t] < -4 || f->dy[t] > 3 || f->dx[t] < -4 || f->dx[t] > 3) CH_FAIL(7);
        tdy |= (unsigned)(f->dy[t] + 4) << (3 * t);
        tdx |= (unsigned)(f->dx[t] + 4) << (3 * t);
        mny = f->dy[t] < mny ? f->dy[t] : mny; mxy = f->dy[t] > mxy ? f->dy[t] : mxy;
        mnx = f->dx[t] < mnx ? f->dx[t] : mnx; mxx = f->dx[t] > mxx ? f->dx[t] : mxx;
    }
    b.padT = -mny; b.padB = mxy; b.padL = -mnx; b.padR = mxx;
    for (int i = 0; i < n; ++i) {
        LayerRec R{};
        b.cur = i;
        dbg_i = i;
        if (items[i].type == SSDN_OP_CONV) {
            const ssdn_conv_args* a = (const ssdn_conv_args*)items[i].args;
            if (!chain_conv_ok(a, f)) CH_FAIL(19);
            R.kind = CH_CONV; R.a = a;
            if (a->c0 > 0) {
                R.p0 = cb_source(b, a->src0, a->up0 ? a->H / 2 : a->H, a->up0 ? a->W / 2 : a->W, a->c0);
                if (R.p0.idx < 0) CH_FAIL(23);
                b.planes[R.p0.idx].halo = true;
            }
            if (a->c1 > 0) {
                R.p1 = cb_source(b, a->src1, a->H, a->W, a->c1);
                if (R.p1.idx < 0) CH_FAIL(28);
                b.planes[R.p1.idx].halo = true;
            }
            if (a->bf16) {
                if (a->add.p && !cb_aux(b, a->add, a->H, a->W, a->M, &R.ad)) CH_FAIL(32);
                if (a->mask.p && !cb_aux(b, a->mask, a->H, a->W, a->M, &R.mk)) CH_FAIL(33);
                if (a->upsum.p && !cb_aux(b, a->upsum_mask, a->H / 2, a->W / 2, a->upsum_c, &R.um)) CH_FAIL(34);
            }
            // an output tensor that is already mirrored by a plane (written twice, or written after it was loaded) is not a chain
            if ((a->dst.p && cb_written(b, a->dst.p)) || (a->pool.p && cb_written(b, a->pool.p)) || (a->upsum.p && cb_written(b, a->upsum.p))) CH_FAIL(37);
            ssdn_view dv = a->dst;
            if (!dv.p) { dv.p = (void*)a; dv.cs = 0; dv.co = 0; }           // (never read back: a private key)
            R.pd.idx = cb_add_plane(b, dv, a->H, a->W, a->M, true, a->upsum.p ? a->upsum_c : 0);
            if (R.pd.idx < 0) CH_FAIL(41);
            if (a->pool.p) { R.pp.idx = cb_add_plane(b, a->pool, a->H / 2, a->W / 2, a->M, true); if (R.pp.idx < 0) CH_FAIL(42); }
            if (a->upsum.p) { R.pu.idx = cb_add_plane(b, a->upsum, a->H / 2, a->W / 2, a->upsum_c, true); if (R.pu.idx < 0) CH_FAIL(43); }
        } else if (items[i].type == SSDN_OP_POOL_BWD) {
            const ssdn_pool_args* a = (const ssdn_pool_args*)items[i].args;
            if (!f->bf16 || !chain_pool_ok(a, f->N)) CH_FAIL(46);
            R.kind = CH_POOL_BWD; R.pa = a;
            R.p0 = cb_source(b, a->dpool, a->H / 2, a->W / 2, a->C);
            if (R.p0.idx < 0) CH_FAIL(49);
            if (!cb_aux(b, a->act, a->H, a->W, a->C, &R.mk)) CH_FAIL(50);
            if (cb_written(b, a->dz.p)) CH_FAIL(51);
            if (a->H * a->W <= 256) { R.pd.idx = cb_add_plane(b, a->dz, a->H, a->W, a->C, true); if (R.pd.idx < 0) CH_FAIL(52); }
        } else CH_FAIL(53);
        b.layers.push_back(R);
    }
    // ---- LDS layout: a plane may take the space of planes that were last read two or more layers before it is created (every wave
    //      has passed a barrier since); first fit, lowest offset ----
    int lds = 0;
    std::vector<std::vector<std::pair<int, int>>> zero((size_t)n);
    for (int i = 0; i < (int)b.planes.size(); ++i) {
        PlaneRec& P = b.planes[i];
        const int str = P.C * 2 + 16;
        const int rows = P.halo ? P.H + b.padT + b.padB : P.H, roww = P.halo ? P.W + b.padL + b.padR : P.W;
        P.bytes = (rows * roww * str + 15) & ~15;
        int off = 0;
        bool moved = true, reused = false;
        while (moved) {
            moved = false;
            for (int j = 0; j < i; ++j) {
                const PlaneRec& Q = b.planes[j];
                if (off >= Q.off + Q.bytes || off + P.bytes <= Q.off) continue;
                if (P.first >= 0 && Q.last + 2 <= P.first) continue;          // dead long enough: may be overwritten
                off = Q.off + Q.bytes;
                moved = true;
            }
        }
        for (int j = 0; j < i; ++j)
            if (!(off >= b.planes[j].off + b.planes[j].bytes || off + P.bytes <= b.planes[j].off)) reused = true;
        P.off = off;
        lds = off + P.bytes > lds ? off + P.bytes : lds;
        if (reused && P.halo) zero[P.first].push_back({off, P.bytes});   // stale bytes under a halo: the producing layer clears the plane first
    }
    if (lds > 160 * 1024) CH_FAIL(86);
    for (int i = 0; i < n; ++i)
        if (zero[i].size() > 2) CH_FAIL(87);
    auto desc = [&](PlaneRef r) {
        ChPlane D{};
        if (r.idx < 0) return D;
        const PlaneRec& P = b.planes[r.idx];
        D.str = P.C * 2 + 16;
        D.roww = P.halo ? P.W + b.padL + b.padR : P.W;
        D.org = P.halo ? (b.padT * D.roww + b.padL) * D.str : 0;
        D.off = P.off + r.coff * 2;
        D.lw = ilog2_exact(P.W); D.lh = ilog2_exact(P.H); D.C = P.C;
        return D;
    };
    bool aux_ok = true;          // a saved tensor in HBM may be fetched through a buffer resource: 32-bit byte offsets
    auto aux = [&](const AuxRef& r) {
        ChAux A{};
        A.v = r.v; A.where = r.where;
        if (r.where == 1) A.P = desc(r.pl);
        else { A.P.lw = ilog2_exact(r.W); A.P.lh = ilog2_exact(r.H); }
        if (r.where == 2 && ((long long)f->N * r.H * r.W * r.v.cs + r.v.co) * 2 >= (1ll << 31)) aux_ok = false;
        return A;
    };
    memset(out, 0, sizeof(*out));
    out->N = f->N; out->nlayers = n; out->lds_bytes = lds; out->tap_dy = tdy; out->tap_dx = tdx; out->bf = f->bf16;
    for (int i = 0; i < (int)b.planes.size(); ++i) {
        if (b.planes[i].produced) continue;
        ChLoad& Ld = out->ld[out->nloads++];
        Ld.src = b.planes[i].v;
        Ld.P = desc(PlaneRef{i, 0});
        const unsigned lnpc = (unsigned)Ld.P.C / 8;
        Ld.npc_magic = lnpc <= 1 ? 0u : (unsigned)((0x100000000ull + lnpc - 1) / lnpc);
    }
    for (int i = 0; i < n; ++i) {
        const LayerRec& R = b.layers[i];
        ChLayer& L = out->ly[i];
        L.kind = R.kind;
        if (R.kind == CH_CONV) {
            const ssdn_conv_args* a = R.a;
            L.M = a->M; L.Mpad = a->Mpad; L.Ktot = a->Ktot; L.c0 = a->c0; L.up0 = a->up0;
            L.w = (const h16*)a->w; L.bias = a->bias;
            L.P0 = desc(R.p0.idx >= 0 ? R.p0 : R.p1);
            L.P1 = desc(R.p1.idx >= 0 ? R.p1 : R.p0);
            L.PD = desc(R.pd);
            L.dst = a->dst; L.dst_c0 = a->upsum.p ? a->upsum_c : 0;
            L.AD = aux(R.ad);
            L.MK = aux(R.mk);
            L.has_pool = R.pp.idx >= 0; L.pool_shifted = a->pool_shifted; L.PP = desc(R.pp); L.pool = a->pool;
            L.upsum_c = a->upsum.p ? a->upsum_c : 0; L.PU = desc(R.pu); L.UM = aux(R.um); L.upsum = a->upsum;
        } else {
            const ssdn_pool_args* a = R.pa;
            L.M = a->C;
            L.P0 = desc(R.p0);
            L.has_pd = R.pd.idx >= 0;
            L.PD = desc(R.pd);
            if (!L.has_pd) { L.PD.lw = ilog2_exact(a->W); L.PD.lh = ilog2_exact(a->H); }
            L.MK = aux(R.mk);
            L.dst = a->dz; L.pool_shifted = a->shifted;
        }
        for (size_t r = 0; r < zero[i].size(); ++r) { L.zero_off[r] = zero[i][r].first; L.zero_bytes[r] = zero[i][r].second; }
        const unsigned npc = (unsigned)L.M / 8;
        L.npc_magic = npc <= 1 ? 0u : (unsigned)((0x100000000ull + npc - 1) / npc);
    }
    // the layer body a layer runs in (k_conv_chain): by its image size; a POOL_BWD layer stays in the body of the convolution behind it
    // (or, behind the last convolution, of the one before it)
    {
        int grp[CH_MAX_LAYERS], g = 0;
        for (int i = n - 1; i >= 0; --i) {
            if (out->ly[i].kind == CH_CONV) g = ch_grp(out->ly[i].PD.lw + out->ly[i].PD.lh);
            grp[i] = g;
        }
        for (int i = 0; i < n; ++i)
            if (!grp[i]) grp[i] = i ? grp[i - 1] : 1;
        out->grp0 = grp[0];
        for (int i = 0; i < n; ++i)
            if (out->ly[i].kind == CH_CONV) {
                out->pf_w[i] = out->ly[i].w;
                out->pf_lines[i] = (int)(((long long)9 * out->ly[i].Mpad * out->ly[i].Ktot * 2) >> 7);
            }
        for (int i = 0; i < n; ++i) out->ly[i].grp_next = grp[i + 1 < n ? i + 1 : i];
    }
    for (int w = 0; w < 4; ++w) {
        out->first[w] = ChNext{nullptr, 0, 1, 0};
        for (int j = 0; j < n; ++j) {
            const ChLayer& J = out->ly[j];
            const int npg = J.PD.lw + J.PD.lh > 7 ? 2 : 1;
            if (J.kind == CH_CONV && w < (J.Mpad >> 5) * npg) { out->first[w] = ChNext{J.w, J.Ktot, npg, J.Mpad * J.Ktot}; break; }
        }
    }
    // the weight stream's next stop behind each layer, per wave: the first later convolution that has an item for the wave
    for (int i = 0; i < n; ++i)
        for (int w = 0; w < 4; ++w) {
            ChNext& X = out->ly[i].nx[w];
            X = ChNext{nullptr, 0, 1, 0};
            for (int j = i + 1; j < n; ++j) {
                const ChLayer& J = out->ly[j];
                const int npg = J.PD.lw + J.PD.lh > 7 ? 2 : 1;
                if (J.kind == CH_CONV && w < (J.Mpad >> 5) * npg) { X = ChNext{J.w, J.Ktot, npg, J.Mpad * J.Ktot}; break; }
            }
        }
    return aux_ok;
}

#undef CH_FAIL
// plans of the op-list runs seen so far (the op lists of an engine are static: a handful per process), keyed by the runs' argument bytes
namespace {
struct ChainHit { int len = 0; ChainArgs* dev = nullptr; size_t lds = 0; int N = 0; int bf = 0; };
struct ChainCache { std::vector<char> key; int device; ChainHit hit; };
std::vector<ChainCache> g_chain_cache;
std::mutex g_chain_mutex;
}  // namespace

static size_t chain_args_size(int type) { return type == SSDN_OP_CONV ? sizeof(ssdn_conv_args) : sizeof(ssdn_pool_args); }

// the candidate run at ops[0..n): consecutive SSDN_OP_CONV / SSDN_OP_POOL_BWD ops of one lane.  *hit = the cached / new plan of its
// longest prefix that runs as one launch (len 0: none).  Returns 0, or a negative error.
static int chain_lookup(const ssdn_op* ops, int n, bool any_lane, ChainHit* hit) {
    *hit = ChainHit{};
    if (!g_chain_on || n < 2 || ops[0].type != SSDN_OP_CONV || !ops[0].args) return 0;
    const ssdn_conv_args* f = (const ssdn_conv_args*)ops[0].args;
    if (f->H * f->W > 256 || f->ntaps != 9 || f->kc != 48) return 0;           // (cheap reject before any bookkeeping)
    int m = 0;
    while (m < n && m < CH_MAX_LAYERS && (ops[m].type == SSDN_OP_CONV || ops[m].type == SSDN_OP_POOL_BWD) && ops[m].args &&
           (any_lane || ops[m].lane == ops[0].lane)) ++m;
    if (m < 2) return 0;
    std::vector<char> key;
    for (int i = 0; i < m; ++i) {
        key.push_back((char)ops[i].type);
        const char* p = (const char*)ops[i].args;
        key.insert(key.end(), p, p + chain_args_size(ops[i].type));
    }
    int dev = 0;
    SSDN_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_chain_mutex);
    for (const ChainCache& cc : g_chain_cache)
        if (cc.device == dev && cc.key == key) { *hit = cc.hit; return 0; }
    ChainCache cc{key, dev, ChainHit{}};
    ChainArgs host;
    for (int len = m; len >= 2; --len)
        if (chain_build(ops, len, &host)) { cc.hit.len = len; break; }
    if (cc.hit.len && ssdn_tuning_env("SSDN_CHAIN_DEBUG"))
        fprintf(stderr, "conv chain: %d of %d candidate ops, %s, N %d, LDS %d B, %d loads\n", cc.hit.len, m, host.bf ? "bwd" : "fwd", host.N, host.lds_bytes, host.nloads);
    if (cc.hit.len) {
        static const int env_ablate = [] { const char* e = ssdn_tuning_env("SSDN_CHAIN_ABLATE"); return e ? atoi(e) : 0; }();
        host.ablate = env_ablate;
        SSDN_CHECK_HIP(hipMalloc((void**)&cc.hit.dev, sizeof(ChainArgs)));
        SSDN_CHECK_HIP(hipMemcpy(cc.hit.dev, &host, sizeof(ChainArgs), hipMemcpyHostToDevice));
        cc.hit.lds = (size_t)host.lds_bytes; cc.hit.N = host.N; cc.hit.bf = host.bf;
    }
    if (g_chain_cache.size() >= 64) {                 // bounded: drop the oldest plan (no launch that uses it can still be queued after
        SSDN_CHECK_HIP(hipDeviceSynchronize());       // the synchronisation)
        if (g_chain_cache.front().hit.dev) (void)hipFree(g_chain_cache.front().hit.dev);
        g_chain_cache.erase(g_chain_cache.begin());
    }
    g_chain_cache.push_back(cc);
    *hit = cc.hit;
    return 0;
}

int chain_len(const ssdn_op* ops, int n, bool any_lane) {
    ChainHit h;
    if (chain_lookup(ops, n, any_lane, &h)) return -1;
    return h.len;
}
// (the query plans without touching a device -- no cache, no table upload: usable without a GPU)
extern "C" int ssdn_chain_len(const ssdn_op* ops, int n) {
    if (!ops || n < 0) return ssdn_set_error("conv chain: bad arguments");
    for (int i = 0; i < n; ++i)
        if (!ops[i].args) return ssdn_set_error("conv chain: null args in op %d", i);
    if (!g_chain_on || n < 2 || ops[0].type != SSDN_OP_CONV) return 0;
    int m = 0;
    while (m < n && m < CH_MAX_LAYERS && (ops[m].type == SSDN_OP_CONV || ops[m].type == SSDN_OP_POOL_BWD) && ops[m].lane == ops[0].lane) ++m;
    ChainArgs host;
    for (int len = m; len >= 2; --len)
        if (chain_build(ops, len, &host)) return len;
    return 0;
}

int launch_chain(const ssdn_op* ops, int n, bool any_lane, hipStream_t s) {
    ChainHit h;
    if (chain_lookup(ops, n, any_lane, &h)) return -1;
    if (h.len != n) return ssdn_set_error("conv chain: the run is not a chain of %d ops", n);
    static bool attr_set_dev[SSDN_MAX_DEVICES_ATTR] = {};
    bool& attr_set = attr_set_dev[ssdn_current_device_slot()];   // (function attributes are per device)
    if (!attr_set) {
        SSDN_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv_chain<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SSDN_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv_chain<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    unsigned long long* trace = nullptr;
#ifdef SSDN_TUNING
    trace = (unsigned long long*)ssdn_debug_get_trace();
#endif
    if (h.bf) SSDN_LAUNCH(k_conv_chain<true>, dim3(h.N), dim3(CH_THREADS), h.lds, s, (const ChainArgs*)h.dev, trace);
    else SSDN_LAUNCH(k_conv_chain<false>, dim3(h.N), dim3(CH_THREADS), h.lds, s, (const ChainArgs*)h.dev, trace);
    SSDN_CHECK_HIP(hipGetLastError());
    return 0;
}

// conv_dma.hip -- k_cdma: persistent, LDS-DMA fed implicit-GEMM 3x3 convolution for the layers that carry the flops.
//
// Same contract as SSDN_OP_CONV (include/ssdn_hip.h; replaces ShiftConv2d / nn.Conv2d + LeakyReLU + Upsample + cat of
// /root/reference/ssdn/ssdn/models/noise_network.py:58-156,241-260 in the forward role and autograd's conv backward-data in the
// data-gradient role); launch_conv() routes a layer here when it fits the shape class below and keeps k_conv for the rest.
//
// What is different from k_conv (conv_mfma.hip):
//   * PERSISTENT workgroups: 2 per CU (80 KiB of LDS each), each walks vertical strips of 16x16-pixel tiles top to bottom.
//   * NOTHING is staged through registers: the input halo tile of a 48-channel chunk (18x18 pixels) and every (tap, chunk) weight
//     slice arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`: no VGPRs, no ds_write, asynchronous, zero fill by range check).
//     The tile of the NEXT chunk / next tile and the weights of the NEXT step are in flight while the current step is on the
//     matrix cores; ONE raw s_barrier per step (a step = one tap x one chunk = 18 MFMAs per wave).
//   * Loader ROLES by wave: waves 0-1 fetch the weight slices, waves 2-3 the halo tiles.  vmcnt is one in-order counter per
//     wave: a wave that issued both kinds would have to wait for a just-issued tile fetch (HBM latency) whenever it needs the
//     weights of the next step (measured: +20 % launch time).  With roles, the weight waves drain their short queue every step
//     and the tile waves only once per chunk, eight steps after their first fetch.
//   * LDS images are unpadded (DMA writes 64 consecutive 16-byte pieces per wave instruction) and XOR-swizzled on the SOURCE
//     side: piece c of halo pixel (hy, hx) sits at piece c ^ (hy & 1); piece c of weight row m at c ^ ((m >> 3) & 1).  Checked by
//     brute force over the ds_read_b128 lane groups: every fragment read is bank-conflict free.
//   * Tile fetches are row items: one halo row = two 54-lane DMA instructions whose per-lane offsets never change.
//   * Fragment addresses are one VGPR base + compile-time immediates (the 9 steps of a chunk are unrolled): per step a wave
//     issues 18 MFMAs, 15 ds_read_b128, a few DMA instructions and their scalar bookkeeping: ~85 instructions beside the MFMAs
//     (k_conv: ~300).  Over a whole launch the SQ counters say 8.4 non-MFMA instructions per MFMA in round 4 (profiles/r04_pmc_cdma_fwd.txt:
//     4.8 VALU + 2.7 SALU + 0.9 LDS), of which the epilogue and the per-tile head are more than half; round 5's direct epilogue
//     halves the epilogue's share (profiles/r05_pmc_cdma_fwd.txt).
//   * Epilogue (round 5, every variant but the fused UPSUM_BWD): a wave owns 4 tile rows; it converts its accumulators, widens the
//     8-byte MFMA fragments to 16-byte pieces with v_permlane32_swap and stores each piece straight from registers (one per-lane
//     offset + an immediate per piece + a scalar base per pass; LeakyReLU sign bytes from the lane's own eight channels); the fused
//     UPSUM_BWD still transposes through a wave-private LDS region, because its 2x2 sums cross pixels.  mask / skip-gradient
//     operands of the data-gradient role are fetched as one batch per pass.  The other workgroup of the CU keeps the matrix cores
//     busy meanwhile -- when it is not in its own epilogue: the two workgroups of a CU start together and stay in phase
//     (profiles/r05_k_cpipe_experiment.txt).
//   * bias enters as the initial value of the accumulators.
//
// Shape class: 9 taps forming a 3x3 window (blind-spot, plain, or either one mirrored = data gradient), H and W multiples of
// 16, 16-bit NHWC output, input channels = n chunks of 48 (+ one optional 16-channel tail chunk), each chunk from one source.
#include "common.h"
#include <cstdlib>
#include <type_traits>

// Tuning aids (ablation bits, phase stamps, weight replication) are compiled in only with -DSSDN_TUNING (`make TUNING=1`; tools/cdma_probe.sh
// and the trace mode of tools/conv_bench.py need such a build): as run-time flags they cost ~40 scalar instructions and 10 branches PER STEP of a kernel that is
// instruction-issue bound (two waves per SIMD, ~300 instructions per 18 MFMAs).
#if defined(SSDN_CDMA_TUNING) || defined(SSDN_TUNING)
#define CD_ABL(xx, bit) (((xx).ablate & (bit)) != 0)
#define CD_TUNING 1
#else
#define CD_ABL(xx, bit) false
#define CD_TUNING 0
#endif

namespace {

constexpr int CD_TBYTES = 31744;   // 18 x 18 pixels x 96 B = 31104, + 640 B that hold the bias (see below)

struct CdAux {
    int m_base, m_cnt;          // output channels [m_base, m_base + m_cnt) of this launch (m_cnt % 8 == 0, <= MT*32)
    int padT, padL, rev;        // halo origin (y0 - padT, x0 - padL); weight tap of halo offset (i,j): rev ? 8-(3i+j) : 3i+j
    int tiles_x, tiles_y;       // 16x16 tiles per image
    int segs, tps;              // a strip (image, tile column) is cut into `segs` work items of `tps` consecutive tiles
    int nitems, xcd_map;
    int nfull, tail16;          // 48-channel chunks, then an optional 16-channel chunk
    int ablate;                 // tuning aid (env SSDN_CDMA_ABLATE, read once): 1 no MFMA, 2 no weight DMA, 4 no tile DMA, 8 no epilogue,
                                // 16 no DMA waits, 32 no step barriers (16, 32: wrong results, timing only), 64 epilogue stores dropped by a
                                // zero-size buffer resource (the arithmetic and the issue slots stay)
    int wrep;                   // experiment (env SSDN_CDMA_WREP): the weight tensor exists in `wrep` consecutive copies
    unsigned long long* trace;  // tuning aid (ssdn_debug_set_trace): s_memtime stamps, 32 per workgroup
};

// LDS-DMA through the compiler's builtin (round 5): it sets M0 and pads the SGPR hazards itself, only where needed.  As inline asm every
// piece carried an `s_nop 4` (nothing inside an asm statement is padded by the compiler, and an SGPR operand may be fresh from a
// v_readlane), which costs an MFMA-issuing wave 12-18 ns per piece (tools/probes/probe_dmacost.hip, profiles/r05_k_cpipe_experiment.txt).
// lds_addr: wave-uniform LDS address of lane 0's 16 bytes; lane i lands at +16 i; EXEC-masked lanes write nothing, out-of-range lanes zeros
typedef __attribute__((address_space(3))) void* cd_lds_ptr;
__device__ __forceinline__ void dma16(unsigned lds_addr, int voff, __amdgpu_buffer_rsrc_t rs, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (cd_lds_ptr)(size_t)__builtin_amdgcn_readfirstlane(lds_addr), 16, voff, __builtin_amdgcn_readfirstlane(soff), 0, 0);
}

template <bool BF>
__device__ __forceinline__ f32x16 cd_mma(half8 av, half8 bv, f32x16 c) {
    if constexpr (BF)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
}

// ---- one step on the matrix cores: tap (I, J) of a KS*16-channel chunk ---------------------------------------------------------
// The LDS -> register -> MFMA pipeline is written by hand: left to itself the compiler (which aims at minimum register pressure
// here) re-uses ONE fragment register set and waits for every ds_read right after issuing it -- nine exposed LDS round trips per
// step.  The fragment reads are inline asm (the compiler neither reorders volatile asm statements nor knows that their results
// arrive late), the wait is an asm statement that takes every fragment register as a read-write operand (so no consumer can be
// scheduled above it), and sched_barrier pins the reads of K-step k+1 in front of the MFMAs of K-step k.
template <int OFF>
__device__ __forceinline__ void lds_rd128(half8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int MT>
__device__ __forceinline__ void lds_wait0(half8 (&aq)[MT], half8 (&bq)[2]) {
    if constexpr (MT == 3)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(aq[0]), "+v"(aq[1]), "+v"(aq[2]), "+v"(bq[0]), "+v"(bq[1]));
    else if constexpr (MT == 2)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(aq[0]), "+v"(aq[1]), "+v"(bq[0]), "+v"(bq[1]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(aq[0]), "+v"(bq[0]), "+v"(bq[1]));
}
// fragment reads of K-step KS_ of tap (I, J): ap = LDS address of the weight slice + lane base, bp = of the halo tile + lane base
template <int MT, int KS, int I, int J, int KS_>
__device__ __forceinline__ void cd_reads(half8 (&aq)[MT], half8 (&bq)[2], unsigned ap, unsigned bp) {
    constexpr int PSTR = KS * 32, PITCH = 18 * PSTR, BOFF = I * PITCH + J * PSTR;
    lds_rd128<BOFF + KS_ * 32>(bq[0], bp);
    lds_rd128<0 * 32 * PSTR + KS_ * 32>(aq[0], ap);
    lds_rd128<BOFF + 2 * PITCH + KS_ * 32>(bq[1], bp);
    if constexpr (MT > 1) lds_rd128<1 * 32 * PSTR + KS_ * 32>(aq[1], ap);
    if constexpr (MT > 2) lds_rd128<2 * 32 * PSTR + KS_ * 32>(aq[2], ap);
}
template <int MT, bool BF>
__device__ __forceinline__ void cd_mmas(f32x16 (&acc)[MT][2], const half8 (&aq)[MT], const half8 (&bq)[2]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        acc[mt][0] = cd_mma<BF>(aq[mt], bq[0], acc[mt][0]);
        acc[mt][1] = cd_mma<BF>(aq[mt], bq[1], acc[mt][1]);
    }
}

struct CdTile { int n, y0, x0; };

}  // namespace

// EPI: bit 0 = multiply by LeakyReLU'(mask), bit 1 = add the skip gradient (data-gradient role only), bit 2 = fused
// SSDN_OP_UPSUM_BWD: a pass of the epilogue is 2 rows x 16 pixels = 8 low-resolution pixels, whose 2x2 sums times
// LeakyReLU'(upsum_mask) are stored instead of the 32 pixels; bit 3 (forward role) = fused SSDN_OP_UNROT_FWD: every pixel goes to
// its un-rotated place in ssdn_conv_args.urot (+ optional LeakyReLU sign bytes) instead of dst; bit 4 (forward role) = the launch also
// writes the LeakyReLU sign bytes of its output (ssdn_conv_args.sign_out); bit 5 (data-gradient role, with bit 0 or bit 2) = the
// LeakyReLU' operand arrives as sign bytes (mask_sign / upsum_mask_sign: one byte per 16-byte piece instead of the piece)
template <int MT, bool BF, int EPI>
__global__ __launch_bounds__(256, 2) void k_cdma(ssdn_conv_args a, CdAux x) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WROWS = MT * 32;
    constexpr int WBYTES = WROWS * 96;        // one (tap, 48-channel chunk) weight slice
    constexpr int NWQ = MT * 3;               // 1 KiB DMA instructions per 48-channel weight slice; the 16-channel slice has MT
    constexpr int OSTR = MT * 64 + 16;        // epilogue: LDS bytes per pixel (16 B x odd: conflict-free ds_write_b128)
    constexpr int NEK = MT * 2;               // epilogue: 64-lane 16-byte row instructions per 32-pixel pass
    constexpr bool HAS_MASK = (EPI & 1) != 0, HAS_ADD = (EPI & 2) != 0, HAS_UPS = (EPI & 4) != 0, UROT = (EPI & 8) != 0;
    constexpr bool SOUT = (EPI & 16) != 0, SMASK = (EPI & 32) != 0;
    constexpr int NUK = (8 * MT * 4 + 63) / 64;   // upsum: 64-lane instructions per pass (8 pixels x cpp pieces)
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool wload = w < 2;                 // waves 0-1 fetch weights, waves 2-3 fetch tiles
    const int lw = w & 1;                     // index inside the role
    const unsigned lds0 = (unsigned)(size_t)smem;
    char* const tbuf0 = smem;
    const unsigned tlds0 = lds0, wlds0 = lds0 + 2 * CD_TBYTES;

    // ---- work items of this workgroup ---------------------------------------------------------------------------------
    // xcd_map: workgroups are dealt to the 8 XCDs round-robin by id and each XCD has its own L2 -> XCD x walks the contiguous
    // item range [x*I/8, (x+1)*I/8), so neighbouring strips (which share halo columns) meet in one L2.
    const int G = gridDim.x;
    int it_first, it_stride, it_end;
    if (x.xcd_map) {
        const int i8 = x.nitems >> 3, xcd = blockIdx.x & 7;
        it_first = xcd * i8 + (blockIdx.x >> 3); it_stride = G >> 3; it_end = (xcd + 1) * i8;
    } else {
        it_first = blockIdx.x; it_stride = G; it_end = x.nitems;
    }
    if (it_first >= it_end) return;

    // ---- per-lane constants ---------------------------------------------------------------------------------------------
    const int xl = l31 & 15, tyl = 4 * w + (l31 >> 4);
    // B fragments (input pixels): lane = pixel (tyl, xl) of the tile, k half kh; tap row parity picks the swizzled piece
    const int par = tyl & 1;
    const int bE48 = tyl * 1728 + xl * 96 + ((kh ^ par) << 4), bO48 = tyl * 1728 + xl * 96 + ((kh ^ par ^ 1) << 4);
    const int bE16 = tyl * 576 + xl * 32 + ((kh ^ par) << 4), bO16 = tyl * 576 + xl * 32 + ((kh ^ par ^ 1) << 4);
    // A fragments (weights): lane = row l31 (+32 mt), k half kh
    const int asw = (kh ^ ((l31 >> 3) & 1)) << 4;
    const int aB48 = l31 * 96 + asw, aB16 = l31 * 32 + asw;
    constexpr int NWU = (NWQ + 1) / 2;       // weight DMA instructions per wave per slice (waves 0-1, q = lw + 2u)
    // epilogue: row instruction k covers 16-byte pieces [64k, 64k+64) of this wave's 32-pixel pass; piece p = (pixel p / cpp,
    // piece p % cpp); pixel px = (row px >> 4 of the pass, column px & 15)
    const int cpp = x.m_cnt >> 3;
    int e_pc[NEK];     // pixel | piece << 8, or -1
#pragma unroll
    for (int k = 0; k < NEK; ++k) {
        const int p = k * 64 + lane;
        const int px = p / cpp, c = p - px * cpp;
        e_pc[k] = p < 32 * cpp ? (px | (c << 8)) : -1;
    }

    // ---- buffer resources ---------------------------------------------------------------------------------------------------
    // (num_records = 2 GiB for every resource: all tensors are smaller -- checked by the launcher -- and the one out-of-range
    //  offset used, 0x80000000, still reads as zero; constants cost no live SGPRs)
    const __amdgpu_buffer_rsrc_t rs_wc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wc), 0, (int)0x80000000, SSDN_BUFFER_RSRC_FLAGS);
    const int H0 = a.up0 ? (a.H >> 1) : a.H, W0 = a.up0 ? (a.W >> 1) : a.W;
    const __amdgpu_buffer_rsrc_t rs_s0 = __builtin_amdgcn_make_buffer_rsrc(a.src0.p, 0, (int)0x80000000, SSDN_BUFFER_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_s1 = __builtin_amdgcn_make_buffer_rsrc(a.src1.p ? a.src1.p : a.src0.p, 0, (int)0x80000000, SSDN_BUFFER_RSRC_FLAGS);
    const int nch = x.nfull + x.tail16;

    int tr_i = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if (CD_TUNING && x.trace && tid == 0 && tr_i < 32) x.trace[(size_t)blockIdx.x * 32 + tr_i++] = __builtin_amdgcn_s_memtime();
    };
    // waves 0-1: DMA of the weight slice (chunk c, halo tap tseq = 3i+j) into weight buffer `wpar`
    // byte offset of the (tap, chunk) slice in the chunk-major copy = wchunk(c) + wt0 + tseq * wts  (tseq = 3i+j of the halo tap; the
    // mirrored window of the data-gradient role walks the taps backwards)
    const int wtap = a.Mpad * a.Ktot * 2;
    const int wt0 = x.rev ? 8 * wtap : 0, wts = x.rev ? -wtap : wtap;
    auto wchunk = [&](int c) __attribute__((always_inline)) {
        return __builtin_amdgcn_readfirstlane((c * 48 * a.Mpad + x.m_base * (c < x.nfull ? 48 : 16)) * 2 + wt0);
    };
    auto issue_w = [&](int wcb, bool full, int tseq, int wpar) __attribute__((always_inline)) {
        if (CD_ABL(x, 2)) return;
        const unsigned dst = wlds0 + wpar * WBYTES;
        // chunk-major pre-swizzled copy [tap][chunk][Mpad][kc] (ssdn_conv_args.wc, mandatory for this kernel): the slice IS the
        // LDS image -> linear 1 KiB pieces
        const int sbase = wcb + tseq * wts;
        const int nq = full ? NWQ : MT;
#pragma unroll
        for (int u = 0; u < NWU; ++u)
            if (lw + 2 * u < nq) dma16(dst + (lw + 2 * u) * 1024, lane * 16, rs_wc, sbase + (lw + 2 * u) * 1024);
    };
    // waves 2-3: DMA of halo-tile ROWS.  A row of a 48-channel chunk is 18 pixels x 6 pieces = 108 pieces = two instructions of
    // 54 active lanes (the other 10 are EXEC-masked: masked lanes write nothing); a row of the 16-channel chunk is 36 pieces =
    // one instruction.  Everything about a ROW is wave-uniform (image row, validity, base address -> SGPRs / soffset); per lane
    // only the (pixel, piece) -> byte offset inside the row remains, the same for every row of a (tile, chunk) up to the
    // swizzle's row parity: prepared once per chunk (row_lane), so a row item costs ~10 scalar and 1 vector
    // instruction.  (First version: ~60 instructions per item, which made the tile waves the slow ones at every barrier.)
    // row item r of a chunk: 48-ch: halo row r >> 1, half r & 1 (36 items); 16-ch: halo row r (18 items).  Wave lw takes r = lw + 2u.
    // LDS piece p of a row holds (pixel p / PP, channel piece (p % PP) ^ (hy & 1)): the swizzle is applied on the SOURCE side.
    // byte offset of this lane's 16 bytes inside a source row of (tile t, chunk c), for an even (pr = 0) / odd halo row, or OOB.
    // (A wave always fetches the same half of the 48-channel rows: half = lw.)
    auto row_lane = [&](const CdTile& t, int c, int pr) __attribute__((always_inline)) {
        const int k0 = c * 48;
        const bool from0 = k0 < a.c0;
        const bool up = from0 && a.up0;
        const int cs = from0 ? a.src0.cs : a.src1.cs;
        const bool full = c < x.nfull;
        const int hp48 = lane / 6;
        const int hx = full ? hp48 + 9 * lw : lane >> 1;
        const int cc = full ? lane - hp48 * 6 : lane & 1;
        const int xx = t.x0 - x.padL + hx;
        const bool ok = (unsigned)xx < (unsigned)a.W;
        const int xs = up ? xx >> 1 : xx;
        return ok ? (xs * cs + (cc ^ pr) * 8) * 2 : (int)0x80000000;
    };
    // everything about the rows of one (tile, chunk) that does not depend on the row: prepared ONCE per chunk, on the scalar unit
    // (plain scalars, not a struct: in some instantiations a struct of them was placed in scratch)
    auto row_ctx = [&](const CdTile& t, int c, int tpar, bool& o_rs, int& o_rbase, int& o_rstride, int& o_ybs, int& o_ush, unsigned& o_dst) __attribute__((always_inline)) {
        const int k0 = c * 48;
        const bool from0 = k0 < a.c0;
        const bool up = from0 && a.up0;
        const int cs = from0 ? a.src0.cs : a.src1.cs;
        const int cbase = from0 ? a.src0.co + k0 : a.src1.co + k0 - a.c0;
        const int Hs = up ? H0 : a.H, Ws = up ? W0 : a.W;
        // (readfirstlane: the values ARE wave-uniform; saying so once per chunk keeps the nine steps' row arithmetic on the scalar unit)
        o_rs = from0;
        o_rstride = __builtin_amdgcn_readfirstlane(Ws * cs * 2);
        o_rbase = __builtin_amdgcn_readfirstlane((t.n * Hs * Ws * cs + cbase) * 2);
        o_ybs = __builtin_amdgcn_readfirstlane(t.y0 - x.padT);
        o_ush = up ? 1 : 0;
        o_dst = __builtin_amdgcn_readfirstlane(tlds0 + tpar * CD_TBYTES);
    };
    auto issue_rows = [&](bool rc_from0, int rc_rbase, int rc_rstride, int rc_ybs, int rc_ush, unsigned rc_dst, bool full, int rlE, int rlO, int u0, int nu) __attribute__((always_inline)) {
        if (CD_ABL(x, 4)) return;
        const bool act = full ? lane < 54 : lane < 36;
#pragma unroll
        for (int uu = 0; uu < 18; ++uu) {
            if (uu >= nu) break;               // (nu is a constant at every call site: the loop unrolls to nu items)
            const int r = lw + 2 * (u0 + uu);
            const int hy = full ? r >> 1 : r, half = full ? r & 1 : 0;
            if (hy >= 18) continue;
            const int y = rc_ybs + hy;
            const bool rowok = (unsigned)y < (unsigned)a.H;
            const int soff = rc_rbase + (y >> rc_ush) * rc_rstride;            // (an out-of-image row fetches nothing: voff is out of range)
            const int voff = rowok ? ((hy & 1) ? rlO : rlE) : (int)0x80000000;
            const unsigned ldsrow = rc_dst + hy * (full ? 1728 : 576) + half * 864;
            if (act) {
                if (rc_from0) dma16(ldsrow, voff, rs_s0, soff); else dma16(ldsrow, voff, rs_s1, soff);
            }
        }
    };

    // ---- the walk -------------------------------------------------------------------------------------------------------------
    auto tile_of = [&](int item, int k) {
        const int strip = item / x.segs, seg = item - strip * x.segs;
        CdTile t;
        t.n = strip / x.tiles_x;
        t.x0 = (strip - t.n * x.tiles_x) * 16;
        t.y0 = (seg * x.tps + k) * 16;
        return t;
    };
    int item = it_first, kt = 0;
    CdTile cur = tile_of(item, 0);
    int tpar = 0, wpar = 0;      // buffers holding the CURRENT chunk's tile / the CURRENT step's weights
    // bias of this launch's channels lives in the 640-byte pad behind tile buffer 0 (the lanes of the last tile DMA
    // instruction that would land there are EXEC-masked)
    float* const bl = reinterpret_cast<float*>(smem + 18 * 18 * 96);
    if (tid < WROWS) bl[tid] = (a.bias && tid < x.m_cnt) ? a.bias[x.m_base + tid] : 0.f;
    // ... and, behind the bias (96 floats at most), 16 x float4: the LeakyReLU' factors of four channels from their four sign bits (entry n =
    // {bit j of n ? 1 : slope}): one ds_read_b128 per nibble instead of a bit test, a select and a scalar multiply per channel
    constexpr int CD_LUT = 18 * 18 * 96 + 384;
    static_assert(CD_LUT + 256 <= CD_TBYTES && WROWS * 4 <= 384, "pad behind tile buffer 0");
    if constexpr (SMASK && HAS_MASK) {
        if (tid < 64) reinterpret_cast<float*>(smem + CD_LUT)[tid] = ((tid >> 2) >> (tid & 3)) & 1 ? 1.f : LRELU_SLOPE;
    }
    if (wload) issue_w(wchunk(0), 0 < x.nfull, 0, 0);
    else {
        bool q_rs; int q_rbase, q_rstride, q_ybs, q_ush; unsigned q_dst;
        row_ctx(cur, 0, 0, q_rs, q_rbase, q_rstride, q_ybs, q_ush, q_dst);
        issue_rows(q_rs, q_rbase, q_rstride, q_ybs, q_ush, q_dst, 0 < x.nfull, row_lane(cur, 0, 0), row_lane(cur, 0, 1), 0, 18);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc(UROT ? a.urot.p : a.dst.p, 0, CD_ABL(x, 64) ? 0 : (int)0x80000000, SSDN_BUFFER_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_sgn = __builtin_amdgcn_make_buffer_rsrc(SOUT ? a.sign_out : a.urot_smask, 0, (!CD_ABL(x, 64) && (SOUT || (UROT && a.urot_smask))) ? (int)0x80000000 : 0, SSDN_BUFFER_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_ms = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HAS_UPS ? a.upsum_mask_sign : a.mask_sign), 0, SMASK ? (int)0x80000000 : 0, SSDN_BUFFER_RSRC_FLAGS);
    const float slope = a.act ? LRELU_SLOPE : 1.f;
    const __amdgpu_buffer_rsrc_t rs_mask = __builtin_amdgcn_make_buffer_rsrc(a.mask.p, 0, HAS_MASK ? (int)0x80000000 : 0, SSDN_BUFFER_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc(a.add.p, 0, HAS_ADD ? (int)0x80000000 : 0, SSDN_BUFFER_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_up = __builtin_amdgcn_make_buffer_rsrc(a.upsum.p, 0, HAS_UPS ? (int)0x80000000 : 0, SSDN_BUFFER_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_um = __builtin_amdgcn_make_buffer_rsrc(a.upsum_mask.p, 0, HAS_UPS ? (int)0x80000000 : 0, SSDN_BUFFER_RSRC_FLAGS);
    int u_pc[NUK];     // upsum: low-res pixel of the pass | piece << 8, or -1
#pragma unroll
    for (int k = 0; k < NUK; ++k) {
        const int p = k * 64 + lane;
        const int j = p / cpp, c = p - j * cpp;
        u_pc[k] = p < 8 * cpp ? (j | (c << 8)) : -1;
    }

    stamp();
    for (;;) {
        // next tile of this workgroup (same item one tile down, or the top of its next item)
        int nitem = item, nkt = kt + 1;
        if (nkt >= x.tps) { nkt = 0; nitem = item + it_stride; }
        const bool has_next = nitem < it_end;
        const CdTile nxt = tile_of(has_next ? nitem : item, nkt);

        // accumulators start at the bias: row = 32 mt + 8 (r >> 2) + 4 kh + (r & 3)
        f32x16 acc[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(bl + mt * 32 + g * 8 + kh * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[mt][0][g * 4 + j] = bb[j]; acc[mt][1][g * 4 + j] = bb[j]; }
            }

        for (int c = 0; c < nch; ++c) {
            const bool last_chunk = c + 1 == nch;
            const bool pf = !last_chunk || has_next;           // is there a chunk to prefetch while this one computes
            const int pc = last_chunk ? 0 : c + 1;
            CdTile pt;                                           // (by value: a reference picked at run time would put both tiles in scratch)
            pt.n = last_chunk ? nxt.n : cur.n; pt.y0 = last_chunk ? nxt.y0 : cur.y0; pt.x0 = last_chunk ? nxt.x0 : cur.x0;
            const bool pf_full = pc < x.nfull;
            int rlE = 0, rlO = 0;
            if (!wload && pf) { rlE = row_lane(pt, pc, 0); rlO = row_lane(pt, pc, 1); }
            bool rc_rs; int rc_rbase, rc_rstride, rc_ybs, rc_ush; unsigned rc_dst;      // (unconditional: scalar values defined on one
            row_ctx(pt, pc, tpar ^ 1, rc_rs, rc_rbase, rc_rstride, rc_ybs, rc_ush, rc_dst);  //  path only end up in VGPRs)
            const int wcb_c = wchunk(c), wcb_p = wchunk(pc);
            const bool cfull = c < x.nfull;
            // step head: the loaders start the fetches that must have landed one step (weights) / one chunk (tile) from now
            auto step_head = [&](int t) __attribute__((always_inline)) {
                if (wload) {
                    if (t < 8) issue_w(wcb_c, cfull, t + 1, wpar ^ 1);
                    else if (pf) issue_w(wcb_p, pf_full, 0, wpar ^ 1);
                } else if (pf && t < 8) {
                    if (pf_full) issue_rows(rc_rs, rc_rbase, rc_rstride, rc_ybs, rc_ush, rc_dst, true, rlE, rlO, t < 2 ? 3 * t : 2 * t + 2, t < 2 ? 3 : 2);   // 18 row items per wave over steps 0..7
                    else issue_rows(rc_rs, rc_rbase, rc_rstride, rc_ybs, rc_ush, rc_dst, false, rlE, rlO, t < 1 ? 0 : t + 1, t < 1 ? 2 : 1);            // 9 row items per wave
                }
            };
            // step tail: own DMA landed, then ONE barrier: every wave's DMA landed and every wave is done with this step's buffers
            auto step_tail = [&](int t) __attribute__((always_inline)) {
                if ((wload || t == 8) && !CD_ABL(x, 16)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!CD_ABL(x, 32)) __builtin_amdgcn_s_barrier();
                wpar ^= 1;
            };
            if (c < x.nfull) {
                const unsigned bE = tlds0 + tpar * CD_TBYTES + bE48, bO = tlds0 + tpar * CD_TBYTES + bO48;
#define CD_STEP48(T, I, J)                                                  \
    {                                                                       \
        const unsigned ap = wlds0 + wpar * WBYTES + aB48, bp = (I & 1) ? bO : bE; \
        half8 aq0[MT], bq0[2], aq1[MT], bq1[2];                             \
        cd_reads<MT, 3, I, J, 0>(aq0, bq0, ap, bp);                         \
        step_head(T);                 /* the loaders' DMA issue covers the latency of the first fragment reads */ \
        lds_wait0<MT>(aq0, bq0);                                            \
        cd_reads<MT, 3, I, J, 1>(aq1, bq1, ap, bp);                         \
        __builtin_amdgcn_sched_barrier(0);                                  \
        if (!CD_ABL(x, 1)) cd_mmas<MT, BF>(acc, aq0, bq0);                \
        __builtin_amdgcn_sched_barrier(0);                                  \
        lds_wait0<MT>(aq1, bq1);                                            \
        cd_reads<MT, 3, I, J, 2>(aq0, bq0, ap, bp);                         \
        __builtin_amdgcn_sched_barrier(0);                                  \
        if (!CD_ABL(x, 1)) cd_mmas<MT, BF>(acc, aq1, bq1);                \
        __builtin_amdgcn_sched_barrier(0);                                  \
        lds_wait0<MT>(aq0, bq0);                                            \
        if (!CD_ABL(x, 1)) cd_mmas<MT, BF>(acc, aq0, bq0);                \
        __builtin_amdgcn_sched_barrier(0);                                  \
        step_tail(T);                                                       \
    }
                CD_STEP48(0, 0, 0) CD_STEP48(1, 0, 1) CD_STEP48(2, 0, 2)
                CD_STEP48(3, 1, 0) CD_STEP48(4, 1, 1) CD_STEP48(5, 1, 2)
                CD_STEP48(6, 2, 0) CD_STEP48(7, 2, 1) CD_STEP48(8, 2, 2)
#undef CD_STEP48
            } else {
                const unsigned bE = tlds0 + tpar * CD_TBYTES + bE16, bO = tlds0 + tpar * CD_TBYTES + bO16;
#define CD_STEP16(T, I, J)                                                  \
    {                                                                       \
        const unsigned ap = wlds0 + wpar * WBYTES + aB16, bp = (I & 1) ? bO : bE; \
        half8 aq0[MT], bq0[2];                                              \
        cd_reads<MT, 1, I, J, 0>(aq0, bq0, ap, bp);                         \
        step_head(T);                                                       \
        lds_wait0<MT>(aq0, bq0);                                            \
        cd_mmas<MT, BF>(acc, aq0, bq0);                                     \
        __builtin_amdgcn_sched_barrier(0);                                  \
        step_tail(T);                                                       \
    }
                CD_STEP16(0, 0, 0) CD_STEP16(1, 0, 1) CD_STEP16(2, 0, 2)
                CD_STEP16(3, 1, 0) CD_STEP16(4, 1, 1) CD_STEP16(5, 1, 2)
                CD_STEP16(6, 2, 0) CD_STEP16(7, 2, 1) CD_STEP16(8, 2, 2)
#undef CD_STEP16
            }
            tpar ^= 1;
            stamp();
        }

        // ---- epilogue, direct form (round 5; every variant but the fused UPSUM_BWD, whose 2x2 sums need the transposed tile): after the
        // permlane swap lane (pixel l31, kh) holds the whole 16-byte piece mt*4 + 2gp + kh of its pixel -- it stores it straight from
        // registers (32-byte runs at a 192-byte pitch; the L2 merges the four pieces of a 64-byte sector, they arrive within a few hundred
        // cycles) and, for the LeakyReLU sign bytes, works on its own eight channels.  ONE per-lane offset per tensor + an immediate per
        // (mt, gp) + the pass's scalar base as soffset: no LDS round trip (24 LDS ops, their waits), no per-piece address arithmetic
        // (~170 VALU per wave and tile), sign byte in 13 instead of ~28 instructions: ~890 -> ~450 instructions per wave and tile in the
        // forward role, ~990 -> ~530 in the data-gradient role (no LeakyReLU there: compile-time), measured in tools/ab_libs.sh.
        if (!CD_ABL(x, 8) && !HAS_UPS && (cpp & 1) == 0) {
            const int pix_t = (cur.n * a.H + cur.y0 + 4 * w) * a.W + cur.x0;     // first pixel of this wave's 4 rows
            const int lrow = l31 >> 4, lcol = l31 & 15;
            const int lpix = lrow * a.W + lcol;
            int ur_base = 0, ur_iu = 0, ur_iv = 0, ur_ju = 0, ur_jv = 0;
            if constexpr (UROT) {
                const int Bq = a.N >> 2, P1 = a.H - 1;
                const int r = (cur.n >= Bq ? 1 : 0) + (cur.n >= 2 * Bq ? 1 : 0) + (cur.n >= 3 * Bq ? 1 : 0);
                ur_iu = r == 0 ? 1 : (r == 2 ? -1 : 0); ur_iv = r == 1 ? 1 : (r == 3 ? -1 : 0);
                ur_ju = r == 3 ? 1 : (r == 1 ? -1 : 0); ur_jv = r == 0 ? 1 : (r == 2 ? -1 : 0);
                const int i0 = r >= 2 ? P1 : 0, j0 = (r == 1 || r == 2) ? P1 : 0;
                ur_base = (((cur.n - r * Bq) * a.H + i0) * a.W + j0) * a.urot.cs + a.urot.co + r * a.M;
            }
            auto pass = [&](auto NtC, auto ActC) __attribute__((always_inline)) {
                constexpr int nt = decltype(NtC)::value;
                constexpr bool ACTC = decltype(ActC)::value != 0;
                const int pix_p = pix_t + 2 * nt * a.W;
                // per-lane offsets of piece kh of this lane's pixel (pieces mt*4 + 2gp + kh: + an immediate), scalar bases as soffset
                int d_lane, d_so = 0;
                bool live = true;
                if constexpr (UROT) {
                    const int u = cur.y0 + 4 * w + 2 * nt + lrow + 1, v = cur.x0 + lcol;
                    live = u < a.H;                                   // row P-1 falls off the shifted image
                    const int dpx = (ur_iu * u + ur_iv * v) * a.W + ur_ju * u + ur_jv * v;
                    d_lane = live ? (ur_base + dpx * a.urot.cs + x.m_base) * 2 + kh * 16 : (int)0x80000000;
                } else {
                    d_lane = lpix * a.dst.cs * 2 + kh * 16;
                    d_so = __builtin_amdgcn_readfirstlane((pix_p * a.dst.cs + a.dst.co + x.m_base) * 2);
                }
                const int s_lane = live ? lpix * (a.M >> 3) + kh : (int)0x80000000;       // sign bytes: M / 8 bytes per pixel
                const int s_so = __builtin_amdgcn_readfirstlane(pix_p * (a.M >> 3) + (x.m_base >> 3));
                // operands of the data-gradient role: one batch of loads per pass, in flight while the accumulators are converted
                u32x4_t ab[MT * 2], mb[MT * 2];
#pragma unroll
                for (int i = 0; i < MT * 2; ++i) {
                    if (i * 2 >= cpp) break;
                    if constexpr (HAS_ADD)
                        ab[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_add, lpix * a.add.cs * 2 + kh * 16 + i * 32,
                                                                      __builtin_amdgcn_readfirstlane((pix_p * a.add.cs + a.add.co + x.m_base) * 2), 0);
                    if constexpr (HAS_MASK && SMASK)
                        mb[i][0] = __builtin_amdgcn_raw_buffer_load_b8(rs_ms, s_lane + i * 2, s_so, 0);
                    else if constexpr (HAS_MASK)
                        mb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_mask, lpix * a.mask.cs * 2 + kh * 16 + i * 32,
                                                                      __builtin_amdgcn_readfirstlane((pix_p * a.mask.cs + a.mask.co + x.m_base) * 2), 0);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        constexpr int DUMMY = 0; (void)DUMMY;
                        const int i = mt * 2 + gp;
                        if (i * 2 >= cpp) break;                  // (cpp is even here: the pieces 2i, 2i+1 exist together)
                        unsigned pk[2][2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            float v[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][(2 * gp + h) * 4 + j];
                            if constexpr (ACTC) {      // LeakyReLU = max(v, slope v): v_pk_mul_f32 + a raw v_max_f32 (fmaxf: + a canonicalising v_max per value)
#pragma unroll
                                for (int j = 0; j < 4; j += 2) {
                                    const f32x2_t t = f32x2_t{v[j], v[j + 1]} * LRELU_SLOPE;
                                    asm("v_max_f32 %0, %1, %2" : "=v"(v[j]) : "v"(v[j]), "v"(t[0]));
                                    asm("v_max_f32 %0, %1, %2" : "=v"(v[j + 1]) : "v"(v[j + 1]), "v"(t[1]));
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
                        if constexpr (HAS_MASK && SMASK && BF) {
                            // sign byte: bit 2q = low half of dword q, bit 2q+1 = its high half -> nibble 0 = dwords 0-1, nibble 1 = dwords 2-3
                            const unsigned sb = mb[i][0];
                            const f32x4 m0 = *reinterpret_cast<const f32x4*>(smem + CD_LUT + ((sb & 15u) << 4));
                            const f32x4 m1 = *reinterpret_cast<const f32x4*>(smem + CD_LUT + ((sb >> 4) << 4));      // (a zero-extended byte)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                f32x2_t v = f32x2_t{bf_lo(o[q]), bf_hi(o[q])};
                                if constexpr (HAS_ADD) v = v + f32x2_t{bf_lo(ab[i][q]), bf_hi(ab[i][q])};
                                v = v * (q < 2 ? f32x2_t{m0[2 * q], m0[2 * q + 1]} : f32x2_t{m1[2 * q - 4], m1[2 * q - 3]});
                                o[q] = pack_bf16x2(v[0], v[1]);
                            }
                        } else if constexpr (HAS_MASK || HAS_ADD) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                float v0, v1;
                                if constexpr (BF) {
                                    v0 = bf_lo(o[q]); v1 = bf_hi(o[q]);
                                    if constexpr (HAS_ADD) { v0 += bf_lo(ab[i][q]); v1 += bf_hi(ab[i][q]); }
                                } else {
                                    v0 = f16_lo(o[q]); v1 = f16_hi(o[q]);
                                    if constexpr (HAS_ADD) { v0 += f16_lo(ab[i][q]); v1 += f16_hi(ab[i][q]); }
                                }
                                if constexpr (HAS_MASK) {
                                    int mlo, mhi;
                                    if constexpr (SMASK) { mlo = (mb[i][0] >> (2 * q)) & 1u; mhi = (mb[i][0] >> (2 * q + 1)) & 1u; }
                                    else { mlo = (int)(short)(mb[i][q] & 0xffffu); mhi = (int)mb[i][q] >> 16; }
                                    v0 *= mlo > 0 ? 1.f : LRELU_SLOPE;
                                    v1 *= mhi > 0 ? 1.f : LRELU_SLOPE;
                                }
                                o[q] = BF ? pack_bf16x2(v0, v1) : pack_f16x2(v0, v1);
                            }
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(o, rs_dst, d_lane + i * 32, d_so, 0);
                        if constexpr (UROT || SOUT) {
                            if (SOUT || a.urot_smask) {
                                // sign byte of the piece: bit 2q = (low half of dword q > 0), bit 2q+1 = (high half > 0), on the raw 16-bit
                                // patterns: min(max(h, 0), 1) per half, then the eight 0/1 halves are merged by shifts
                                // (asm: written with __builtin_elementwise_max / _min on short2 the compiler folded the four dwords into one)
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
                                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)sb, rs_sgn, s_lane + i * 2, s_so, 0);
                            }
                        }
                    }
            };
            if (a.act) { pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); pass(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}); }
            else { pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); pass(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}); }
        } else
        // ---- epilogue through LDS (fused UPSUM_BWD; blocks with an odd number of 8-channel pieces): the tile buffer of the last chunk
        // (tpar ^ 1 now) is dead; the other one holds / receives the next tile ----
        if (!CD_ABL(x, 8)) {
            char* reg = tbuf0 + (tpar ^ 1) * CD_TBYTES + w * (32 * OSTR);
            const int pix_t = (cur.n * a.H + cur.y0 + 4 * w) * a.W + cur.x0;     // first pixel of this wave's 4 rows
            // fused UNROT_FWD: image n = r*B + b; the pixel (y, x) is S_r[u, v] with (u, v) = (y + 1, x) and lands at (b, i, j),
            //   r=0: (i,j) = (u,v); r=1: (v, P-1-u); r=2: (P-1-u, P-1-v); r=3: (P-1-v, u)    -- as affine forms with uniform coefficients
            int ur_base = 0, ur_iu = 0, ur_iv = 0, ur_ju = 0, ur_jv = 0;
            if constexpr (UROT) {
                const int Bq = a.N >> 2, P1 = a.H - 1;
                const int r = (cur.n >= Bq ? 1 : 0) + (cur.n >= 2 * Bq ? 1 : 0) + (cur.n >= 3 * Bq ? 1 : 0);
                ur_iu = r == 0 ? 1 : (r == 2 ? -1 : 0); ur_iv = r == 1 ? 1 : (r == 3 ? -1 : 0);
                ur_ju = r == 3 ? 1 : (r == 1 ? -1 : 0); ur_jv = r == 0 ? 1 : (r == 2 ? -1 : 0);
                const int i0 = r >= 2 ? P1 : 0, j0 = (r == 1 || r == 2) ? P1 : 0;
                // element offset of destination pixel (b, i0, j0), channel block r
                ur_base = (((cur.n - r * Bq) * a.H + i0) * a.W + j0) * a.urot.cs + a.urot.co + r * a.M;
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int pix_p = pix_t + 2 * nt * a.W;
                // operands of the data-gradient role: one batch of loads per pass, in flight while the accumulators are converted
                u32x4_t ab[NEK], mb[NEK];
                int goff[NEK], soff[NEK];
#pragma unroll
                for (int k = 0; k < NEK; ++k) {
                    int epc = e_pc[k];
                    asm volatile("" : "+v"(epc));     // opaque: derived offsets are recomputed here, not hoisted out of the tile loop and spilled
                    const bool on = epc >= 0;
                    const int px = epc & 255, c16 = (epc >> 8) << 4;
                    const int pix = pix_p + (px >> 4) * a.W + (px & 15);
                    if constexpr (UROT) {
                        const int u = cur.y0 + 4 * w + 2 * nt + (px >> 4) + 1, v = cur.x0 + (px & 15);
                        const bool live = on && u < a.H;            // row P-1 falls off the shifted image
                        const int dpx = (ur_iu * u + ur_iv * v) * a.W + ur_ju * u + ur_jv * v;
                        goff[k] = live ? (ur_base + dpx * a.urot.cs + x.m_base) * 2 + c16 : (int)0x80000000;
                        soff[k] = live ? pix * (a.M >> 3) + (x.m_base >> 3) + (c16 >> 4) : (int)0x80000000;
                    } else
                    goff[k] = on ? (pix * a.dst.cs + a.dst.co + x.m_base) * 2 + c16 : (int)0x80000000;
                    if constexpr (HAS_ADD)
                        ab[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_add, on ? (pix * a.add.cs + a.add.co + x.m_base) * 2 + c16 : (int)0x80000000, 0, 0);
                    if constexpr (HAS_MASK && SMASK)      // one sign byte per 16-byte piece (written by the producer: sign_out), M / 8 bytes per pixel
                        mb[k][0] = __builtin_amdgcn_raw_buffer_load_b8(rs_ms, on ? pix * (a.M >> 3) + (x.m_base >> 3) + (c16 >> 4) : (int)0x80000000, 0, 0);
                    else if constexpr (HAS_MASK)
                        mb[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_mask, on ? (pix * a.mask.cs + a.mask.co + x.m_base) * 2 + c16 : (int)0x80000000, 0, 0);
                    if constexpr (SOUT) soff[k] = on ? pix * (a.M >> 3) + (x.m_base >> 3) + (c16 >> 4) : (int)0x80000000;
                }
                // registers -> LDS: lane (pixel l31, kh) holds channels 32 mt + 8 g + 4 kh + (0..3) in acc[mt][nt][4g..4g+3];
                // v_permlane32_swap pairs group g of the kh = 1 lanes with group g+1 of the kh = 0 lanes: afterwards the low
                // lanes hold all 8 channels of group g and the high lanes all 8 of group g+1 of their pixel
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        unsigned pk[2][2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            float v[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][(2 * gp + h) * 4 + j];
                            if constexpr (!BF) {      // LeakyReLU (slope 1: identity); the data-gradient role has none (conv_dma_eligible)
#pragma unroll
                                for (int j = 0; j < 4; j += 2) {
                                    const f32x2_t t = f32x2_t{v[j], v[j + 1]} * slope;
                                    asm("v_max_f32 %0, %1, %2" : "=v"(v[j]) : "v"(v[j]), "v"(t[0]));
                                    asm("v_max_f32 %0, %1, %2" : "=v"(v[j + 1]) : "v"(v[j + 1]), "v"(t[1]));
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
                        const int piece = mt * 4 + 2 * gp + kh;
                        *reinterpret_cast<u32x4_t*>(reg + l31 * OSTR + piece * 16) = o;
                    }
                if constexpr (HAS_UPS) {
                    // fused UPSUM_BWD: low-res pixel j of the pass = pixels (row 0|1, column 2j|2j+1), summed in scan order
                    const int pixl = (cur.n * (a.H >> 1) + ((cur.y0 + 4 * w + 2 * nt) >> 1)) * (a.W >> 1) + (cur.x0 >> 1);
                    u32x4_t um[NUK];
#pragma unroll
                    for (int k = 0; k < NUK; ++k) {
                        int upc = u_pc[k];
                        asm volatile("" : "+v"(upc));
                        const bool on = upc >= 0;
                        const int j = upc & 255, c16 = (upc >> 8) << 4;
                        if constexpr (SMASK)
                            um[k][0] = __builtin_amdgcn_raw_buffer_load_b8(rs_ms, on ? (pixl + j) * (a.upsum_c >> 3) + (x.m_base >> 3) + (c16 >> 4) : (int)0x80000000, 0, 0);
                        else
                        um[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_um, on ? ((pixl + j) * a.upsum_mask.cs + a.upsum_mask.co + x.m_base) * 2 + c16 : (int)0x80000000, 0, 0);
                    }
#pragma unroll
                    for (int k = 0; k < NUK; ++k) {
                        int upc = u_pc[k];
                        asm volatile("" : "+v"(upc));
                        const bool on = upc >= 0;
                        const int j = upc & 255, c16 = (upc >> 8) << 4;
                        float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const u32x4_t o = *reinterpret_cast<const u32x4_t*>(reg + (on ? ((q4 >> 1) * 16 + 2 * j + (q4 & 1)) * OSTR + c16 : 0));
#pragma unroll
                            for (int q = 0; q < 4; ++q) { sum[2 * q] += bf_lo(o[q]); sum[2 * q + 1] += bf_hi(o[q]); }
                        }
                        u32x4_t r;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            int mlo, mhi;
                            if constexpr (SMASK) { mlo = (um[k][0] >> (2 * q)) & 1u; mhi = (um[k][0] >> (2 * q + 1)) & 1u; }
                            else { mlo = (int)(short)(um[k][q] & 0xffffu); mhi = (int)um[k][q] >> 16; }
                            r[q] = pack_bf16x2(sum[2 * q] * (mlo > 0 ? 1.f : LRELU_SLOPE), sum[2 * q + 1] * (mhi > 0 ? 1.f : LRELU_SLOPE));
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(r, rs_up, on ? ((pixl + j) * a.upsum.cs + a.upsum.co + x.m_base) * 2 + c16 : (int)0x80000000, 0, 0);
                    }
                } else {
                // LDS -> HBM: whole 16-byte pieces, pixel-contiguous
#pragma unroll
                for (int k = 0; k < NEK; ++k) {
                    int epc = e_pc[k];
                    asm volatile("" : "+v"(epc));
                    const bool on = epc >= 0;
                    const int px = epc & 255, c16 = (epc >> 8) << 4;
                    u32x4_t o = *reinterpret_cast<const u32x4_t*>(reg + (on ? px * OSTR + c16 : 0));
                    if constexpr (HAS_MASK || HAS_ADD) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float v0, v1;
                            if constexpr (BF) {
                                v0 = bf_lo(o[q]); v1 = bf_hi(o[q]);
                                if constexpr (HAS_ADD) { v0 += bf_lo(ab[k][q]); v1 += bf_hi(ab[k][q]); }
                            } else {
                                v0 = f16_lo(o[q]); v1 = f16_hi(o[q]);
                                if constexpr (HAS_ADD) { v0 += f16_lo(ab[k][q]); v1 += f16_hi(ab[k][q]); }
                            }
                            if constexpr (HAS_MASK) {
                                // LeakyReLU'(saved fp16 activation): slope where it is <= 0 (sign test on the raw halves: the
                                // activation is > 0 iff its bits, read as a signed 16-bit integer, are > 0; NaN never occurs)
                                int mlo, mhi;
                                if constexpr (SMASK) { mlo = (mb[k][0] >> (2 * q)) & 1u; mhi = (mb[k][0] >> (2 * q + 1)) & 1u; }
                                else { mlo = (int)(short)(mb[k][q] & 0xffffu); mhi = (int)mb[k][q] >> 16; }
                                v0 *= mlo > 0 ? 1.f : LRELU_SLOPE;
                                v1 *= mhi > 0 ? 1.f : LRELU_SLOPE;
                            }
                            o[q] = BF ? pack_bf16x2(v0, v1) : pack_f16x2(v0, v1);
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(o, rs_dst, goff[k], 0, 0);
                    if constexpr (UROT || SOUT) {
                        if (SOUT || a.urot_smask) {      // sign byte of the piece: bit q = (channel q > 0), on the raw fp16 halves
                            unsigned sb = 0;
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                sb |= ((int)(short)(o[q] & 0xffffu) > 0 ? 1u : 0u) << (2 * q) | (((int)o[q] >> 16) > 0 ? 1u : 0u) << (2 * q + 1);
                            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)sb, rs_sgn, soff[k], 0, 0);
                        }
                    }
                }
                }
            }
        }
        stamp();
        if (!has_next) break;
        // the dead buffer becomes the target of the next tile fetches (issued from step 0 on): every wave must be done with it
        __builtin_amdgcn_s_barrier();
        item = nitem; kt = nkt; cur = nxt;
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
// halo window of a 3x3 tap set; returns false if the taps are not a full 3x3 window in forward or mirrored order
static bool cd_window(const ssdn_conv_args* a, int* padT, int* padL, int* rev) {
    if (a->ntaps != 9) return false;
    int mny = 0, mnx = 0;
    for (int t = 0; t < 9; ++t) { mny = a->dy[t] < mny ? a->dy[t] : mny; mnx = a->dx[t] < mnx ? a->dx[t] : mnx; }
    bool fwd = true, mir = true;
    for (int t = 0; t < 9; ++t) {
        const int i = a->dy[t] - mny, j = a->dx[t] - mnx;
        if (i != t / 3 || j != t % 3) fwd = false;
        if (i != 2 - t / 3 || j != 2 - t % 3) mir = false;
    }
    if (!fwd && !mir) return false;
    *padT = -mny; *padL = -mnx; *rev = fwd ? 0 : 1;
    return true;
}

bool conv_dma_eligible(const ssdn_conv_args* a, bool any_size) {
    int pt, pl, rv;
    if (!cd_window(a, &pt, &pl, &rv)) return false;
    if (a->dst32 || (a->H & 15) || (a->W & 15) || !a->wc) return false;       // (wc: the chunk-major weight copy this kernel streams)
    const int tail = a->Ktot % 48;
    if (tail != 0 && tail != 16) return false;
    if (a->c0 % 48 && a->c0 != a->Ktot) return false;           // a chunk never straddles the two sources
    if ((a->M & 7) || (a->Mpad & 31)) return false;
    if (!a->bf16 && (a->mask.p || a->add.p)) return false;      // mask / skip gradient: data-gradient role only
    if (a->bf16 && a->act) return false;                        // ... which has no LeakyReLU (compile-time in the epilogue)
    if (a->urot.p && (a->bf16 || a->M != 96 || a->Mpad != 96 || a->H != a->W || (a->N & 3) || a->mask.p || a->add.p || a->upsum.p || a->pool.p ||
                      (a->urot.co & 7) || (a->urot.cs & 7) || (long long)(a->N / 4) * a->H * a->W * a->urot.cs * 2 >= (1ll << 31))) return false;
    if (a->upsum.p && (!a->bf16 || a->mask.p || a->add.p || a->upsum_c % 96 || a->upsum_c > a->M)) return false;
    // persistent grid: worth it from about one 256-pixel tile per CU upwards (smaller layers: k_conv's 32-channel blocks)
    int cus = ssdn_device_cus();
    if (cus <= 0) cus = 256;
    const long long tiles = (long long)a->N * (a->H >> 4) * (a->W >> 4);
    if (tiles < cus && !any_size) return false;
    int csmax = a->dst.cs > a->src1.cs ? a->dst.cs : a->src1.cs;
    csmax = csmax > a->src0.cs ? csmax : a->src0.cs;
    csmax = csmax > a->mask.cs ? csmax : a->mask.cs;
    csmax = csmax > a->add.cs ? csmax : a->add.cs;
    if ((long long)a->N * a->H * a->W * csmax * 2 >= (1ll << 31)) return false;
    if (9ll * a->Mpad * a->Ktot * 2 >= (1ll << 31)) return false;
    return true;
}

// LeakyReLU sign bytes (ssdn_conv_args.sign_out / mask_sign / upsum_mask_sign): forward role of one 96-channel block writes them; the
// data-gradient role reads them for a mask that views a whole tensor of M channels (no skip gradient in the same launch) or for the
// up-sampled half's mask of the fused UPSUM_BWD (a whole tensor of upsum_c channels)
bool conv_dma_signs(const ssdn_conv_args* a) {
    if (!conv_dma_eligible(a, false)) return false;
    if (a->sign_out && (a->bf16 || !a->dst.p || a->urot.p || a->M != 96 || a->Mpad != 96)) return false;
    if (a->mask_sign && (!a->bf16 || !a->mask.p || a->add.p || a->upsum.p || a->mask.cs != a->M || a->mask.co != 0 || (a->Mpad != 96 && a->Mpad != 64))) return false;
    if (a->upsum_mask_sign && (!a->bf16 || !a->upsum.p || a->upsum_mask.cs != a->upsum_c || a->upsum_mask.co != 0)) return false;
    return true;
}

int conv_dma_lds_bytes(int mt) { return 2 * CD_TBYTES + 2 * mt * 32 * 96; }

template <int MT, bool BF, int EPI>
static int cd_launch(const ssdn_conv_args* a, CdAux x, hipStream_t s) {
    static bool attr_set_dev[SSDN_MAX_DEVICES_ATTR] = {};
    bool& attr_set = attr_set_dev[ssdn_current_device_slot()];   // (function attributes are per device)
    if (!attr_set) {
        SSDN_CHECK_HIP(hipFuncSetAttribute((const void*)k_cdma<MT, BF, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    int cus = ssdn_device_cus();
    if (cus <= 0) return -1;
    // cut the strips into segments until there are two work items per CU (or one tile per item)
    const int nstrips = a->N * x.tiles_x;
    x.segs = 1;
    while (nstrips * x.segs < 2 * cus && x.segs < x.tiles_y && x.tiles_y % (x.segs * 2) == 0) x.segs *= 2;
    x.tps = x.tiles_y / x.segs;
    x.nitems = nstrips * x.segs;
    const int grid = x.nitems < 2 * cus ? x.nitems : 2 * cus;
    x.xcd_map = (x.nitems % 8 == 0 && grid % 8 == 0) ? 1 : 0;
    const double px = (double)a->N * a->H * a->W;
    const int kreal = a->kreal > 0 ? a->kreal : a->Ktot;
    const double flops = 2.0 * px * x.m_cnt * kreal * 9;
    const double bytes = px * (a->c0 * 2.0 / (a->up0 ? 4.0 : 1.0) + a->c1 * 2.0) + px * x.m_cnt * 2.0;
    prof_begin(MT == 3 ? SSDN_PROF_CDMA_MT3 : SSDN_PROF_CDMA_MT21, s);
    SSDN_LAUNCH((k_cdma<MT, BF, EPI>), dim3(grid), dim3(256), conv_dma_lds_bytes(MT), s, *a, x);
    prof_end(MT == 3 ? SSDN_PROF_CDMA_MT3 : SSDN_PROF_CDMA_MT21, s, flops, bytes);
    return 0;
}

template <int MT>
static int cd_launch_role(const ssdn_conv_args* a, const CdAux& x, hipStream_t s) {
    if (a->urot.p) {                               // fused UNROT_FWD: one 96-channel block (conv_dma_eligible)
        if constexpr (MT == 3) return cd_launch<3, false, 8>(a, x, s);
        return ssdn_set_error("conv_dma: fused UNROT_FWD needs M = 96");
    }
    if (!a->bf16) {
        if (a->sign_out) {                         // + LeakyReLU sign bytes of the output (conv_dma_signs: one 96-channel block)
            if constexpr (MT == 3) return cd_launch<3, false, 16>(a, x, s);
            return ssdn_set_error("conv_dma: sign_out needs M = 96");
        }
        return cd_launch<MT, false, 0>(a, x, s);
    }
    if (a->upsum.p && x.m_base < a->upsum_c) {      // block of up-sampled-input channels: fused UPSUM_BWD (MT = 3 only)
        if constexpr (MT == 3) {
            if (x.m_base + 96 <= a->upsum_c && x.m_cnt == 96 && !a->mask.p && !a->add.p)
                return a->upsum_mask_sign ? cd_launch<3, true, 4 | 32>(a, x, s) : cd_launch<3, true, 4>(a, x, s);
        }
        return ssdn_set_error("conv_dma: fused upsum needs whole 96-channel blocks without mask / add");
    }
    const int epi = (a->mask.p ? 1 : 0) | (a->add.p ? 2 : 0);
    if (a->mask_sign) {                            // LeakyReLU' from sign bytes (conv_dma_signs: mask without add, Mpad 96 or 64)
        if constexpr (MT >= 2) { if (epi == 1) return cd_launch<MT, true, 1 | 32>(a, x, s); }
        return ssdn_set_error("conv_dma: mask_sign needs a mask, no skip gradient, Mpad = 96 or 64");
    }
    switch (epi) {
        case 0: return cd_launch<MT, true, 0>(a, x, s);
        case 1: return cd_launch<MT, true, 1>(a, x, s);
        case 2: return cd_launch<MT, true, 2>(a, x, s);
        default: return cd_launch<MT, true, 3>(a, x, s);
    }
}

int launch_conv_dma(const ssdn_conv_args* a, hipStream_t s) {
    CdAux x;
    if (!cd_window(a, &x.padT, &x.padL, &x.rev)) return ssdn_set_error("conv_dma: not a 3x3 window");
    x.tiles_x = a->W >> 4; x.tiles_y = a->H >> 4;
    x.segs = 1; x.tps = x.tiles_y; x.nitems = 0; x.xcd_map = 0;
    x.nfull = a->Ktot / 48; x.tail16 = (a->Ktot % 48) ? 1 : 0;
    static const int env_ablate = [] { const char* e = ssdn_tuning_env("SSDN_CDMA_ABLATE"); return e ? atoi(e) : 0; }();
    x.ablate = env_ablate;
    static const int env_wrep = [] { const char* e = ssdn_tuning_env("SSDN_CDMA_WREP"); return e ? atoi(e) : 0; }();
    x.wrep = env_wrep;
    x.trace = (unsigned long long*)ssdn_debug_get_trace();
    int rc = 0;
    for (int mb = 0; mb < a->Mpad && !rc; mb += 96) {
        int rows = a->Mpad - mb;
        rows = rows > 96 ? 96 : rows;
        const int mt = rows / 32;
        x.m_base = mb;
        x.m_cnt = a->M - mb < rows ? a->M - mb : rows;
        if (x.m_cnt <= 0) break;
        if (mt == 3) rc = cd_launch_role<3>(a, x, s);
        else if (mt == 2) rc = cd_launch_role<2>(a, x, s);
        else rc = cd_launch_role<1>(a, x, s);
    }
    if (rc) return rc;
    SSDN_CHECK_HIP(hipGetLastError());
    return 0;
}

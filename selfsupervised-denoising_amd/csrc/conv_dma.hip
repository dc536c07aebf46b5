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
//     weights of the next step (measured: +20 % launch time).  Since round 6 each role runs its OWN copy of the whole tile walk
//     (`run<WL>`): no wave-uniform role branch inside a step.
//   * LDS images are unpadded (DMA writes 64 consecutive 16-byte pieces per wave instruction) and XOR-swizzled on the SOURCE
//     side: piece c of halo pixel (hy, hx) sits at piece c ^ (hy & 1); piece c of weight row m at c ^ ((m >> 3) & 1).  Checked by
//     brute force over the ds_read_b128 lane groups: every fragment read is bank-conflict free.
//   * A step = one tap of one chunk = 18 MFMAs per wave (MT = 3).  Its loader pieces sit in the GAPS behind the MFMAs of its first
//     K-step (an MFMA occupies the matrix core for 8 issue slots; what the wave issues meanwhile is free): a weight piece is one DMA
//     instruction (the wave's pieces of a slice share one M0 / soffset and differ by the immediate offset), a halo-row item is
//     s_add (running soffset), s_add (M0), the DMA -- all 64 lanes active (overlapping row windows), rows issued in the chunk's first
//     three steps.  Fragment addresses are one VGPR base + compile-time immediates; ~50 instructions beside the 18 MFMAs per step
//     (round 5: ~120): 3.7 non-MFMA instructions per MFMA over a launch (profiles/r06_pmc_cdma_fwd.txt; round 5: 7.2, round 4: 8.4).
//   * ONE s_barrier per step, in FRONT of the step's last K-step (every fragment of the step has been read by then); the next step's
//     first fragments are read behind it, under those MFMAs (KIND 1 / 2; CD_MOVE_BARRIER).
//   * Chunk kinds (48 / 16 channels) are a compile-time property of the instantiation (KIND): one unrolled body per role.  As a
//     run-time choice between bodies the register allocator kept the accumulators in different registers per body and copied 96
//     registers per chunk (see the comment at k_cdma).
//   * Epilogue (round 5, every variant but the fused UPSUM_BWD): a wave owns 4 tile rows; it converts its accumulators, widens the
//     8-byte MFMA fragments to 16-byte pieces with v_permlane32_swap and stores each piece straight from registers (one per-lane
//     offset + an immediate per piece + a scalar base per pass; LeakyReLU sign bytes from the lane's own eight channels); the fused
//     UPSUM_BWD still transposes through a wave-private LDS region, because its 2x2 sums cross pixels.  The sign bytes of the
//     data-gradient role's LeakyReLU' are requested for both passes up front; the next tile's tap-1 weights are requested BEFORE the
//     epilogue (the tile's first barrier waits with vmcnt(N)).  Every 16-byte store is followed by s_nop 7 (its data registers are
//     read later than the compiler's two wait states when the memory pipeline is backed up: profiles/r06_store_data_hazard.txt).
//   * bias enters as the initial value of the accumulators.
//
// Shape class: 9 taps forming a 3x3 window (blind-spot, plain, or either one mirrored = data gradient), H and W multiples of
// 16, 16-bit NHWC output, input channels = n chunks of 48 (+ one optional 16-channel tail chunk), each chunk from one source.
#include "common.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

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

// A/B aid (EXTRA=-DCD_MOVE_BARRIER=0|1): the step barrier in front of the step's last K-step, the next step's first fragment reads behind it
#ifndef CD_DEPHASE
#define CD_DEPHASE 0
#endif
#ifndef CD_MOVE_BARRIER
#define CD_MOVE_BARRIER 1
#endif
#ifndef CD_WARMUP
#define CD_WARMUP 1
#endif

namespace {

constexpr int CD_TBYTES = 31744;   // 18 x 18 pixels x 96 B = 31104, + 640 B that hold the bias (see below)

struct CdAux {
    int m_base, m_cnt;          // output channels [m_base, m_base + m_cnt) of this launch (m_cnt % 8 == 0, <= MT*32)
    int padT, padL, rev;        // halo origin (y0 - padT, x0 - padL); weight tap of halo offset (i,j): rev ? 8-(3i+j) : 3i+j
    int tiles_x, tiles_y;       // 16x16 tiles per image
    int segs, tps;              // a strip (image, tile column) is cut into `segs` work items of `tps` consecutive tiles (segs: a power of two)
    int segs_sh;                // log2(segs)
    unsigned tx_magic;          // ceil(2^32 / tiles_x): strip / tiles_x as one s_mul_hi_u32 (tiles_x > 1; strips and tiles_x < 2^16)
    int nitems, xcd_map;
    int nfull, tail16;          // 48-channel chunks, then an optional 16-channel chunk
    int ablate;                 // tuning aid (env SSDN_CDMA_ABLATE, read once): 1 no MFMA, 2 no weight DMA, 4 no tile DMA, 8 no epilogue,
                                // 16 no DMA waits, 32 no step barriers (16, 32: wrong results, timing only), 64 epilogue stores dropped by a
                                // zero-size buffer resource (the arithmetic and the issue slots stay)
    int wrep;                   // experiment (env SSDN_CDMA_WREP): the weight tensor exists in `wrep` consecutive copies
    int dephase;                // the second workgroup of a CU starts `dephase` x ~0.5 us late (0: together; see k_cdma)
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
__device__ __forceinline__ void cd_mma(f32x16& c, half8 av, half8 bv) {
    if constexpr (BF) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
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
        cd_mma<BF>(acc[mt][0], aq[mt], bq[0]);
        cd_mma<BF>(acc[mt][1], aq[mt], bq[1]);
    }
}

struct CdTile { int n, y0, x0; };

// how waves 0-1 share the 1 KiB pieces of one (tap, chunk) weight slice: each wave's pieces sit behind ONE M0 / soffset pair and differ by the
// instruction's immediate offset (0 .. 3072: the immediate advances the global AND the LDS address, and the slice is the LDS image)
template <int MT, bool FULL>
struct CdWSplit {
    static constexpr int nq = FULL ? MT * 3 : MT;                       // pieces of the slice (48-channel / 16-channel chunk)
    static constexpr int NB = (nq + 1) / 2 > 4 ? 4 : (nq + 1) / 2;      // pieces per wave
    static constexpr bool extra = nq > 2 * NB;                          // MT = 3, 48 channels: piece 8 -- wave 0 fetches it
    static constexpr int b1 = extra ? NB : nq - NB;                     // first piece of wave 1 (an overlap re-fetches the same bytes)
};
template <int N> using cd_ic = std::integral_constant<int, N>;
template <class F, int... Is>
__device__ __forceinline__ void cd_static_for_impl(F& f, std::integer_sequence<int, Is...>) { (f(cd_ic<Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void cd_static_for(F&& f) { cd_static_for_impl(f, std::make_integer_sequence<int, N>{}); }
template <int IMM, int AUX = 0>
__device__ __forceinline__ void dma16i(unsigned lds_addr, int voff, __amdgpu_buffer_rsrc_t rs, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (cd_lds_ptr)(size_t)lds_addr, 16, voff, soff, IMM, AUX);
}

}  // namespace

// EPI: bit 0 = multiply by LeakyReLU'(mask), bit 1 = add the skip gradient (data-gradient role only), bit 2 = fused
// SSDN_OP_UPSUM_BWD: a pass of the epilogue is 2 rows x 16 pixels = 8 low-resolution pixels, whose 2x2 sums times
// LeakyReLU'(upsum_mask) are stored instead of the 32 pixels; bit 3 (forward role) = fused SSDN_OP_UNROT_FWD: every pixel goes to
// its un-rotated place in ssdn_conv_args.urot (+ optional LeakyReLU sign bytes) instead of dst; bit 4 (forward role) = the launch also
// writes the LeakyReLU sign bytes of its output (ssdn_conv_args.sign_out); bit 5 (data-gradient role, with bit 0 or bit 2) = the
// LeakyReLU' operand arrives as sign bytes (mask_sign / upsum_mask_sign: one byte per 16-byte piece instead of the piece)
// AF ("all full"): every chunk of the launch has 48 channels (Ktot % 48 == 0) -- the K loop is then ONE role-specialised body.  With the
// chunk kind as a run-time choice between bodies (AF = false: a 16-channel tail chunk exists) the register allocator keeps the accumulators
// in different registers per body and copies 96 registers at every chunk boundary (252 registers, SGPR spills); with one body the same
// kernel takes ~155 registers and no copy (round 6).
// KIND 2: n >= 1 chunks of 48 channels, then ONE 16-channel chunk (decode_block_1.0: 96 up-sampled + 3 image channels in a 16-channel slot) --
// the bodies follow each other in program order (full .. full, full with a tail prefetch, tail), again without a run-time choice.
template <int MT, bool BF, int EPI, int KIND>
__global__ __launch_bounds__(256, 2) void k_cdma(ssdn_conv_args a, CdAux x) {
    constexpr bool AF = KIND == 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WROWS = MT * 32;
    constexpr int WBYTES = WROWS * 96;        // one (tap, 48-channel chunk) weight slice
    constexpr int OSTR = MT * 64 + 16;        // epilogue: LDS bytes per pixel (16 B x odd: conflict-free ds_write_b128)
    constexpr int NEK = MT * 2;               // epilogue: 64-lane 16-byte row instructions per 32-pixel pass
    constexpr bool HAS_MASK = (EPI & 1) != 0, HAS_ADD = (EPI & 2) != 0, HAS_UPS = (EPI & 4) != 0, UROT = (EPI & 8) != 0;
    constexpr bool SOUT = (EPI & 16) != 0, SMASK = (EPI & 32) != 0;
    constexpr int NUK = (8 * MT * 4 + 63) / 64;   // upsum: 64-lane instructions per pass (8 pixels x cpp pieces)
    constexpr int G = 2 * MT;                 // MFMAs of a K-step = gaps the loader pieces of a step are dealt to
    constexpr bool MOVE = CD_MOVE_BARRIER != 0;
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
    const int G_ = gridDim.x;
    int it_first, it_stride, it_end;
    if (x.xcd_map) {
        const int i8 = x.nitems >> 3, xcd = blockIdx.x & 7;
        it_first = xcd * i8 + (blockIdx.x >> 3); it_stride = G_ >> 3; it_end = (xcd + 1) * i8;
    } else {
        it_first = blockIdx.x; it_stride = G_; it_end = x.nitems;
    }
    if (it_first >= it_end) return;
#if CD_WARMUP
    // L2 warm-up: a launch finds its weights in no L2, and all workgroups of an XCD walk them in the same order at the same time -- every
    // slice of the first tile is a miss for all of them, a step ahead of its use.  The workgroups of the XCD (blockIdx % 8) fetch one
    // 128-byte line per thread, all lines of the chunk-major copy, before anything else
    const int pf_line = (blockIdx.x >> 3) * 256 + tid;
    const unsigned pf = __builtin_amdgcn_raw_buffer_load_b32(
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wc), 0, 9 * a.Mpad * a.Ktot * 2, SSDN_BUFFER_RSRC_FLAGS), pf_line << 7, 0, 0);
#endif

    // ---- per-lane constants ---------------------------------------------------------------------------------------------
    // (few on purpose: with two fragment sets and 96 accumulator registers the allocator has ~40 registers for everything else; what the
    //  16-channel chunk and the epilogue need is re-derived from the lane id where it is used)
    // B fragments (input pixels): lane = pixel (tyl, xl) of the tile, k half kh; tap row parity picks the swizzled piece (the odd rows' base
    // is the even rows' ^ 16).  A fragments (weights): lane = row l31 (+32 mt), k half kh
    const int bE48 = (4 * w + (l31 >> 4)) * 1728 + (l31 & 15) * 96 + ((kh ^ ((l31 >> 4) & 1)) << 4);
    const int aB48 = l31 * 96 + ((kh ^ ((l31 >> 3) & 1)) << 4);
    const int cpp = x.m_cnt >> 3;

    // ---- buffer resources ---------------------------------------------------------------------------------------------------
    // (num_records = 2 GiB for every resource: all tensors are smaller -- checked by the launcher -- and the one out-of-range
    //  offset used, 0x80000000, still reads as zero; constants cost no live SGPRs)
    const __amdgpu_buffer_rsrc_t rs_wc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wc), 0, (int)0x80000000, SSDN_BUFFER_RSRC_FLAGS);
    const int H0 = a.up0 ? (a.H >> 1) : a.H, W0 = a.up0 ? (a.W >> 1) : a.W;
    const int nch = x.nfull + x.tail16;

    int tr_i = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if (CD_TUNING && x.trace && tid == 0 && tr_i < 32) x.trace[(size_t)blockIdx.x * 32 + tr_i++] = __builtin_amdgcn_s_memtime();
    };

    // ---- waves 0-1: the weight stream --------------------------------------------------------------------------------------------
    // byte offset of the (tap, chunk) slice in the chunk-major copy = wchunk(c) + tseq * wts  (tseq = 3i+j of the halo tap; the
    // mirrored window of the data-gradient role walks the taps backwards).  The slice IS the LDS image (ssdn_conv_args.wc): linear
    // 1 KiB pieces, every lane at +16 lane.
    const int wtap = a.Mpad * a.Ktot * 2;
    const int wt0 = x.rev ? 8 * wtap : 0, wts = x.rev ? -wtap : wtap;
    const int wvoff = lane * 16;
    const bool lw0 = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;       // (its own scalar compare: not the lane mask `wload` and `lw` share)
    auto wchunk = [&](int c) __attribute__((always_inline)) {
        return __builtin_amdgcn_readfirstlane((c * 48 * a.Mpad + x.m_base * ((AF || c < x.nfull) ? 48 : 16)) * 2 + wt0);
    };
    // piece j of this wave's share of a slice: d / so = LDS address / byte offset of the wave's first piece
    auto wpiece = [&](auto FULLc, auto Jc, unsigned d, int so) __attribute__((always_inline)) {
        using S = CdWSplit<MT, decltype(FULLc)::value != 0>;
        constexpr int j = decltype(Jc)::value;
        if (CD_ABL(x, 2)) return;
        if constexpr (j < S::NB) dma16i<j * 1024>(d, wvoff, rs_wc, so);
        else if constexpr (S::extra) { if (lw0) dma16i<0>(d + 2 * S::NB * 1024, wvoff, rs_wc, so + 2 * S::NB * 1024); }
    };
    auto wbase = [&](bool full) __attribute__((always_inline)) {      // byte offset of this wave's first piece inside a slice
        return lw ? (full ? CdWSplit<MT, true>::b1 : CdWSplit<MT, false>::b1) * 1024 : 0;
    };
    auto wslice = [&](bool full, unsigned dbuf, int sl) __attribute__((always_inline)) {     // a whole slice (start-up, chunk change)
        const int wb = wbase(full);
        if (AF || full) cd_static_for<CdWSplit<MT, true>::NB + 1>([&](auto jc) { wpiece(cd_ic<1>{}, jc, dbuf + wb, sl + wb); });
        else cd_static_for<CdWSplit<MT, false>::NB + 1>([&](auto jc) { wpiece(cd_ic<0>{}, jc, dbuf + wb, sl + wb); });
    };

    // ---- waves 2-3: the halo-tile stream -----------------------------------------------------------------------------------------
    // A halo row of a 48-channel chunk is 18 pixels x 6 pieces = 108 pieces.  Wave 0 of the role fetches row pieces 0..63, wave 1 pieces
    // 44..107 (ALL 64 lanes active: no EXEC juggling; the 20 pieces both fetch are the same bytes); a row of the 16-channel chunk is 36
    // pieces = one instruction of 36 lanes, rows dealt alternately.  Everything about a ROW is wave-uniform and lives on the scalar unit
    // as RUNNING values (soffset of the next row, its increment after an even / odd halo row -- an up-sampled source advances every
    // other row); per lane only the (pixel, piece) -> byte offset inside a source row remains, one register per swizzle parity,
    // prepared once per chunk.  A row item is: s_add (soffset), s_add (M0), the DMA -- and, for the halo rows that can fall off the
    // image (0, 1, 16, 17), a compare + select of the out-of-range offset (zero fill by the range check).
    // LDS piece p of a row holds (pixel p / PP, channel piece (p % PP) ^ (hy & 1)): the swizzle is applied on the SOURCE side.
    __amdgpu_buffer_rsrc_t rs_t = rs_wc;
    int t_so = 0, t_incA = 0, t_incB = 0, t_ybs = 0, rlE = 0, rlO = 0;
    unsigned t_tb = 0;
    // prepare the rows of (tile (tn, ty0, tx0), chunk c) for tile buffer `tbi`; live = false: a resource of size 0 (every lane out of range)
    auto tile_setup = [&](int tn, int ty0, int tx0, int c, int tbi, bool live) __attribute__((always_inline)) {
        const int k0 = c * 48;
        const bool from0 = k0 < a.c0;
        const bool up = from0 && a.up0;
        const int cs = from0 ? a.src0.cs : a.src1.cs;
        const int cbase = from0 ? a.src0.co + k0 : a.src1.co + k0 - a.c0;
        const int Hs = up ? H0 : a.H, Ws = up ? W0 : a.W;
        const bool full = AF || c < x.nfull;
        // (readfirstlane: the values ARE wave-uniform; saying so keeps the row arithmetic on the scalar unit)
        const int rstride = __builtin_amdgcn_readfirstlane(Ws * cs * 2);
        const int rbase = __builtin_amdgcn_readfirstlane(((CD_ABL(x, 128) ? 0 : tn) * Hs * Ws * cs + cbase) * 2);
        t_ybs = __builtin_amdgcn_readfirstlane(ty0 - x.padT);
        const int y_first = t_ybs + (full ? 0 : lw);
        t_so = rbase + (up ? y_first >> 1 : y_first) * rstride;
        t_incA = up ? ((y_first & 1) ? rstride : 0) : rstride;            // after this wave's 1st, 3rd, .. row
        t_incB = up ? ((y_first & 1) ? 0 : rstride) : rstride;            // after its 2nd, 4th, .. row
        if (!full) { t_incA = up ? rstride : 2 * rstride; t_incB = t_incA; }    // (rows lw, lw + 2, ..)
        t_tb = __builtin_amdgcn_readfirstlane(tlds0 + tbi * CD_TBYTES + (full ? lw * (44 * 16) : lw * 576));
        rs_t = __builtin_amdgcn_make_buffer_rsrc(from0 ? a.src0.p : (a.src1.p ? a.src1.p : a.src0.p), 0, live ? (int)0x80000000 : 0, SSDN_BUFFER_RSRC_FLAGS);
        // byte offset of this lane's 16 bytes inside a source row, for an even / odd halo row, or out of range
        const int rp = full ? lane + 44 * lw : lane;
        const int hp = rp / 6;
        const int hx = full ? hp : rp >> 1;
        const int cc = full ? rp - hp * 6 : rp & 1;
        const int xx = tx0 - x.padL + hx;
        const bool ok = (unsigned)xx < (unsigned)a.W;
        const int xs = up ? xx >> 1 : xx;
        rlE = ok ? (xs * cs + cc * 8) * 2 : (int)0x80000000;
        rlO = ok ? (xs * cs + (cc ^ 1) * 8) * 2 : (int)0x80000000;
    };
    // row item: halo row HY (48-channel chunk; compile-time) / this wave's U-th row (16-channel chunk: halo row lw + 2 U)
    auto trow = [&](auto FULLc, auto HYc) __attribute__((always_inline)) {
        constexpr bool full = decltype(FULLc)::value != 0;
        constexpr int hy = decltype(HYc)::value;
        if (CD_ABL(x, 4)) return;
        const int so = t_so;
        t_so += (hy & 1) ? t_incB : t_incA;
        if constexpr (full) {
            int voff = (hy & 1) ? rlO : rlE;
            if constexpr (hy < 2 || hy >= 16) voff = (unsigned)(t_ybs + hy) < (unsigned)a.H ? voff : (int)0x80000000;
            if (CD_ABL(x, 512)) dma16i<0, 2>(t_tb + hy * 1728, voff, rs_t, so);
            else dma16i<0>(t_tb + hy * 1728, voff, rs_t, so);
        } else {
            int voff = lw ? rlO : rlE;
            voff = (unsigned)(t_ybs + lw + 2 * hy) < (unsigned)a.H ? voff : (int)0x80000000;
            if (lane < 36) dma16i<0>(t_tb + hy * 1152, voff, rs_t, so);
        }
    };

    // ---- the walk -------------------------------------------------------------------------------------------------------------
    auto tile_of = [&](int item, int k) {
        const int strip = item >> x.segs_sh, seg = item & (x.segs - 1);
        CdTile t;
        t.n = x.tiles_x == 1 ? strip : (int)__umulhi((unsigned)strip, x.tx_magic);
        t.x0 = (strip - t.n * x.tiles_x) * 16;
        t.y0 = (seg * x.tps + k) * 16;
        return t;
    };
    int item = it_first, kt = 0;
    CdTile cur = tile_of(item, 0);
    int tpar = 0, wpar = 0;      // buffers holding the CURRENT chunk's tile / the CURRENT step's weights
    // bias of this launch's channels lives in the 640-byte pad behind tile buffer 0 (nothing is fetched there)
    float* const bl = reinterpret_cast<float*>(smem + 18 * 18 * 96);
    if (tid < WROWS) bl[tid] = (a.bias && tid < x.m_cnt) ? a.bias[x.m_base + tid] : 0.f;
    // ... and, behind the bias (96 floats at most), 16 x float4: the LeakyReLU' factors of four channels from their four sign bits (entry n =
    // {bit j of n ? 1 : slope}): one ds_read_b128 per nibble instead of a bit test, a select and a scalar multiply per channel
    constexpr int CD_LUT = 18 * 18 * 96 + 384;
    static_assert(CD_LUT + 256 <= CD_TBYTES && WROWS * 4 <= 384, "pad behind tile buffer 0");
    if constexpr (SMASK && HAS_MASK) {
        if (tid < 64) reinterpret_cast<float*>(smem + CD_LUT)[tid] = ((tid >> 2) >> (tid & 3)) & 1 ? 1.f : LRELU_SLOPE;
    }

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

    // The whole walk exists once per loader ROLE (waves 0-1 / waves 2-3): no wave-uniform role branch inside it, and the register allocator
    // sees two independent paths (one shared K loop with role branches around the MFMAs made it keep the accumulators in different
    // registers per arm and spill what lives across them).
    auto run = [&](auto WLc) __attribute__((always_inline)) {
    constexpr bool WL = decltype(WLc)::value != 0;
    if constexpr (WL) {
        wslice(KIND != 0 || 0 < x.nfull, wlds0, wchunk(0));
        wslice(KIND != 0 || 0 < x.nfull, wlds0 + WBYTES, wchunk(0) + wts);       // tap 1: see "first step of a tile" below
    } else {
        tile_setup(cur.n, cur.y0, cur.x0, 0, 0, true);
        if (KIND != 0 || 0 < x.nfull) cd_static_for<18>([&](auto hc) { trow(cd_ic<1>{}, hc); });
        else cd_static_for<9>([&](auto hc) { trow(cd_ic<0>{}, hc); });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // The two workgroups of a CU start together and, with equal work, stay in phase: both sit in their epilogue (stores, no MFMAs) and in
    // the waits at a chunk's end at the same time.  Starting the one in the CU's second workgroup slot (HW_ID.TG_ID) a few microseconds late
    // puts one's latency phases under the other's MFMAs; the price is that delay at the end of the launch.
    if (CD_TUNING && x.trace && tid == 0) x.trace[(size_t)blockIdx.x * 32 + 31] = 0x100000000ull | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    if (x.dephase > 0 && ((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 16) & 1))
        for (int d = 0; d < x.dephase; ++d) __builtin_amdgcn_s_sleep(16);
    stamp();
    for (;;) {
        // next tile of this workgroup (same item one tile down, or the top of its next item)
        int nitem = item, nkt = kt + 1;
        if (nkt >= x.tps) { nkt = 0; nitem = item + it_stride; }
        const bool has_next = nitem < it_end;
        const CdTile nxt = tile_of(has_next ? nitem : item, nkt);

        // accumulators start at the bias: row = 32 mt + 8 (r >> 2) + 4 kh + (r & 3)
        f32x16 acc[MT][2];
        int bo = kh * 16;
        asm volatile("" : "+v"(bo));      // (opaque: otherwise the 96 initial values are loop invariants -- a second register set kept live for the whole launch)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(smem + 18 * 18 * 96 + (mt * 32 + g * 8) * 4 + bo);
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[mt][0][g * 4 + j] = bb[j]; acc[mt][1][g * 4 + j] = bb[j]; }
            }

        // one chunk; CFc / PFc: kind of this chunk / of the prefetched one -- 1 = 48 channels, 0 = 16 channels, 2 = decided at run time
        auto chunk = [&](int c, auto CFc, auto PFc) __attribute__((always_inline)) {
            constexpr int CF = decltype(CFc)::value, PFK = decltype(PFc)::value;
            const bool last_chunk = c + 1 == nch;
            const bool pf = !last_chunk || has_next;           // is there a chunk to prefetch while this one computes
            const int pc = last_chunk ? 0 : c + 1;
            const bool cfull = CF == 2 ? c < x.nfull : CF == 1, pf_full = PFK == 2 ? pc < x.nfull : PFK == 1;
            // fragment addresses of this chunk: tile buffer tpar; weight buffer wpar in the even steps, the other one in the odd steps
            const unsigned tb = tlds0 + tpar * CD_TBYTES;
            int bE_ = bE48, aB_ = aB48;
            if (!cfull) {
                int lo = lane;
                asm volatile("" : "+v"(lo));          // (opaque: derived here, not hoisted out of the tile loop into live registers)
                const int l31o = lo & 31, kho = lo >> 5;
                bE_ = (4 * w + (l31o >> 4)) * 576 + (l31o & 15) * 32 + ((kho ^ ((l31o >> 4) & 1)) << 4);
                aB_ = l31o * 32 + ((kho ^ ((l31o >> 3) & 1)) << 4);
            }
            const unsigned bE = tb + bE_, bO = tb + (bE_ ^ 16);
            const unsigned apE = wlds0 + wpar * WBYTES + aB_, apO = wlds0 + (wpar ^ 1) * WBYTES + aB_;
            // loader state of this chunk.  weights: the slice of tap t+1 goes to the buffer step t does NOT read; wso = its byte offset
            // (this wave's first piece), advanced by one tap per step.  tiles: the rows of the next chunk / the next tile's first chunk.
            // First step of a TILE (waves 0-1): its weight slice (tap 1) was requested BEFORE the previous tile's epilogue (below; the very
            // first one by the start-up code), so the step issues nothing and its barrier waits with vmcnt(N), N = the memory instructions
            // the epilogue issued behind that request (vmcnt counts in order: the slice has landed, the epilogue's stores may still be on
            // their way).  As a fetch issued in step 0 it sat behind those stores in the counter, and the step waited for their round trip.
            const bool wfirst = c == 0;
            int wso = 0;
            unsigned wdA = 0, wdB = 0;
            if constexpr (WL) {
                const int wb = wbase(cfull);
                wso = wchunk(c) + wb;
                wdA = __builtin_amdgcn_readfirstlane(wlds0 + wpar * WBYTES + wb);
                wdB = __builtin_amdgcn_readfirstlane(wlds0 + (wpar ^ 1) * WBYTES + wb);
            } else {
                tile_setup(last_chunk ? nxt.n : cur.n, last_chunk ? nxt.y0 : cur.y0, last_chunk ? nxt.x0 : cur.x0, pc, tpar ^ 1, pf);
            }
            // ONE role-specialised body per (role, kind of this chunk, kind of the prefetched chunk): no wave-uniform branch inside a step.
            // A step = one tap: KS K-steps of 2 MT MFMAs; the loader pieces of the step sit in the gaps BEHIND the MFMAs of its first
            // K-step (an MFMA occupies the matrix core for 8 issue slots; what the wave issues meanwhile is free), K-step k+1's fragment
            // reads are issued before K-step k's MFMAs; one s_barrier per step.
            auto body = [&](auto FULLc, auto PFULLc) __attribute__((always_inline)) {
                constexpr bool FULL = decltype(FULLc)::value != 0, PFULL = decltype(PFULLc)::value != 0;
                constexpr int KS = FULL ? 3 : 1;
                constexpr bool MV = MOVE && FULL && KIND != 0;      // (the multi-body form has no registers to spare for fragments that live across steps)
                half8 fa[2][MT], fb[2][2];
                auto gap = [&](auto Tc, auto Ic) __attribute__((always_inline)) {
                    constexpr int T = decltype(Tc)::value, i = decltype(Ic)::value;
                    if constexpr (WL) {
                        using S = CdWSplit<MT, FULL>;
                        constexpr int NP = S::NB + (S::extra ? 1 : 0);
                        if constexpr (T < 8) {
                            if constexpr (i == 0) wso += wts;
                            cd_static_for<NP>([&](auto jc) {
                                constexpr int j = decltype(jc)::value;
                                if constexpr ((j < G - 1 ? j : G - 1) == i) {
                                    if constexpr (T == 0) { if (!wfirst) wpiece(FULLc, jc, wdB, wso); }
                                    else wpiece(FULLc, jc, (T & 1) ? wdA : wdB, wso);
                                }
                            });
                        } else if constexpr (i == 0) {
                            if (pf) wslice(pf_full, wlds0 + (wpar ^ 1) * WBYTES, wchunk(pc));
                        }
                    } else {
                        // the rows of the prefetched chunk, EARLY: one item per gap from step 0 on (MT = 3: 18 items in steps 0-2), so that
                        // the last of them has six steps to land before the chunk's last barrier waits for it (with 3 / 2 items per step
                        // up to step 7 that wait was ~1.9 us per chunk: rows from HBM need more than one step)
                        constexpr int N = PFULL ? 18 : 9, NS = G >= 3 ? G : 3;
                        constexpr int k0 = T * NS, n = k0 >= N ? 0 : (N - k0 < NS ? N - k0 : NS);
                        cd_static_for<n>([&](auto kc) {
                            constexpr int k = decltype(kc)::value;
                            if constexpr ((k < G - 1 ? k : G - 1) == i) trow(PFULLc, cd_ic<k0 + k>{});
                        });
                    }
                };
                auto sync = [&](auto Tc) __attribute__((always_inline)) {
                    constexpr int T = decltype(Tc)::value;
                    if constexpr (WL && T == 0) {
                        // (N: what the DIRECT epilogue of a whole MT*32-channel block issues per wave and tile; any other epilogue: 0)
                        constexpr int NEPI = HAS_UPS ? 0 : 4 * MT * (1 + (HAS_ADD ? 1 : 0) + (HAS_MASK ? 1 : 0) + (SOUT ? 1 : 0));
                        if (wfirst && NEPI > 0 && cpp == 4 * MT && !CD_ABL(x, 8 | 1024)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NEPI) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    } else if (WL || T == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                };
                auto step = [&](auto Tc) __attribute__((always_inline)) {
                    constexpr int T = decltype(Tc)::value, I = T / 3, J = T % 3;
                    constexpr int P = MV ? (T & 1) : 0;          // fragment set that holds K-step 0
                    const unsigned ap = (T & 1) ? apO : apE, bp = (I & 1) ? bO : bE;
                    if constexpr (!MV || T == 0) cd_reads<MT, KS, I, J, 0>(fa[P], fb[P], ap, bp);
                    lds_wait0<MT>(fa[P], fb[P]);
                    if constexpr (KS > 1) cd_reads<MT, KS, I, J, 1>(fa[P ^ 1], fb[P ^ 1], ap, bp);
                    __builtin_amdgcn_sched_barrier(0);
                    cd_static_for<G>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        if (!CD_ABL(x, 1)) cd_mma<BF>(acc[i >> 1][i & 1], fa[P][i >> 1], fb[P][i & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        gap(Tc, ic);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    if constexpr (KS > 1) {
                        lds_wait0<MT>(fa[P ^ 1], fb[P ^ 1]);
                        cd_reads<MT, KS, I, J, 2>(fa[P], fb[P], ap, bp);
                        __builtin_amdgcn_sched_barrier(0);
                        if (!CD_ABL(x, 1)) cd_mmas<MT, BF>(acc, fa[P ^ 1], fb[P ^ 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        lds_wait0<MT>(fa[P], fb[P]);
                        if constexpr (MV && T < 8) {
                            // the barrier in FRONT of the last K-step's MFMAs: every fragment of this step has been read; the next step's
                            // first fragments travel while these MFMAs run
                            sync(Tc);
                            constexpr int T1 = T + 1, I1 = T1 / 3, J1 = T1 % 3;
                            cd_reads<MT, KS, I1, J1, 0>(fa[P ^ 1], fb[P ^ 1], (T1 & 1) ? apO : apE, (I1 & 1) ? bO : bE);
                            __builtin_amdgcn_sched_barrier(0);
                            if (!CD_ABL(x, 1)) cd_mmas<MT, BF>(acc, fa[P], fb[P]);
                            __builtin_amdgcn_sched_barrier(0);
                        } else {
                            if (!CD_ABL(x, 1)) cd_mmas<MT, BF>(acc, fa[P], fb[P]);
                            __builtin_amdgcn_sched_barrier(0);
                            sync(Tc);
                        }
                    } else sync(Tc);
                };
                step(cd_ic<0>{});
                stamp();          // (tuning builds: the chunk's first step on its own -- it is the one that waits for the epilogue's stores)
                cd_static_for<8>([&](auto tc) { step(cd_ic<decltype(tc)::value + 1>{}); });
            };
            if constexpr (CF != 2 && PFK != 2) body(cd_ic<CF>{}, cd_ic<PFK>{});
            else if constexpr (WL) {
                if (cfull) body(cd_ic<1>{}, cd_ic<1>{}); else body(cd_ic<0>{}, cd_ic<1>{});
            } else if (cfull) {
                if (pf_full) body(cd_ic<1>{}, cd_ic<1>{}); else body(cd_ic<1>{}, cd_ic<0>{});
            } else {
                if (pf_full) body(cd_ic<0>{}, cd_ic<1>{}); else body(cd_ic<0>{}, cd_ic<0>{});
            }
            tpar ^= 1; wpar ^= 1;
            stamp();
        };
        if constexpr (KIND == 1) {
            for (int c = 0; c < nch; ++c) chunk(c, cd_ic<1>{}, cd_ic<1>{});
        } else if constexpr (KIND == 2) {
            for (int c = 0; c + 1 < x.nfull; ++c) chunk(c, cd_ic<1>{}, cd_ic<1>{});
            chunk(x.nfull - 1, cd_ic<1>{}, cd_ic<0>{});
            chunk(x.nfull, cd_ic<0>{}, cd_ic<1>{});
        } else {
            for (int c = 0; c < nch; ++c) chunk(c, cd_ic<2>{}, cd_ic<2>{});
        }
        if constexpr (WL) {
            if (has_next) wslice(KIND != 0 || 0 < x.nfull, wlds0 + (wpar ^ 1) * WBYTES, wchunk(0) + wts);      // next tile, tap 1 (see `wfirst`)
        }
        // ---- epilogue, direct form (round 5; every variant but the fused UPSUM_BWD, whose 2x2 sums need the transposed tile): after the
        // permlane swap lane (pixel l31, kh) holds the whole 16-byte piece mt*4 + 2gp + kh of its pixel -- it stores it straight from
        // registers (32-byte runs at a 192-byte pitch; the L2 merges the four pieces of a 64-byte sector, they arrive within a few hundred
        // cycles) and, for the LeakyReLU sign bytes, works on its own eight channels.  ONE per-lane offset per tensor + an immediate per
        // (mt, gp) + the pass's scalar base as soffset: no LDS round trip (24 LDS ops, their waits), no per-piece address arithmetic
        // (~170 VALU per wave and tile), sign byte in 13 instead of ~28 instructions: ~890 -> ~450 instructions per wave and tile in the
        // forward role, ~990 -> ~530 in the data-gradient role (no LeakyReLU there: compile-time), measured in tools/ab_libs.sh.
        if (!CD_ABL(x, 8) && !HAS_UPS && (cpp & 1) == 0 && !CD_ABL(x, 1024)) {
            const int pix_t = (cur.n * a.H + cur.y0 + 4 * w) * a.W + cur.x0;     // first pixel of this wave's 4 rows
            // pieces come in pairs (2i, 2i+1); their count as a SCALAR integer: compared against the loop index it is s_cmp + s_cbranch_scc (as
            // `i * 2 >= cpp` the allocator kept the comparison as a spilled lane mask and re-made it through v_cndmask / v_cmp per piece)
            const int npair = __builtin_amdgcn_readfirstlane(cpp >> 1);
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));      // (opaque: the epilogue's per-lane offsets are derived per tile, not kept live through the K loop)
            const int l31 = lane_o & 31, kh = lane_o >> 5;
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
            // sign-byte form of the LeakyReLU' operand: the bytes of BOTH passes are requested up front (12 one-byte loads: one exposed round
            // trip per tile instead of one per pass)
            constexpr bool HOIST = HAS_MASK && SMASK && !HAS_ADD && !UROT;
            unsigned mbh[2][MT * 2];
            if constexpr (HOIST) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int pix_p = pix_t + 2 * nt * a.W;
                    const int s_so = __builtin_amdgcn_readfirstlane(pix_p * (a.M >> 3) + (x.m_base >> 3));
#pragma unroll
                    for (int i = 0; i < MT * 2; ++i) {
                        if (i >= npair) break;
                        mbh[nt][i] = __builtin_amdgcn_raw_buffer_load_b8(rs_ms, lpix * (a.M >> 3) + kh + i * 2, s_so, 0);
                    }
                }
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
                    if (i >= npair) break;
                    if constexpr (HAS_ADD)
                        ab[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_add, lpix * a.add.cs * 2 + kh * 16 + i * 32,
                                                                      __builtin_amdgcn_readfirstlane((pix_p * a.add.cs + a.add.co + x.m_base) * 2), 0);
                    if constexpr (HOIST)
                        mb[i][0] = mbh[nt][i];
                    else if constexpr (HAS_MASK && SMASK)
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
                        if (i >= npair) break;                    // (cpp is even here: the pieces 2i, 2i+1 exist together)
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
                        if (CD_ABL(x, 256)) __builtin_amdgcn_raw_buffer_store_b128(o, rs_dst, d_lane + i * 32, d_so, 2);
                        else __builtin_amdgcn_raw_buffer_store_b128(o, rs_dst, d_lane + i * 32, d_so, 0);
                        // A 16-byte store reads its data registers some cycles after it issued -- more than the two wait states the
                        // compiler pads when the memory pipeline is backed up (here: behind the LDS-DMA streams and 11 other stores): a
                        // VALU write to `o` five instructions later reached HBM in ~1e-5 of the pieces (round 6, tools/r6_ab3.sh).
                        asm volatile("s_nop 7" ::: "memory");
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
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int l31 = lane_o & 31, kh = lane_o >> 5;
            // row instruction k covers 16-byte pieces [64k, 64k+64) of this wave's 32-pixel pass; piece p = (pixel p / cpp, piece p % cpp);
            // pixel px = (row px >> 4 of the pass, column px & 15)
            int e_pc[NEK];     // pixel | piece << 8, or -1
            if constexpr (!HAS_UPS) {
#pragma unroll
                for (int k = 0; k < NEK; ++k) {
                    const int p = k * 64 + lane_o;
                    const int px = p / cpp, c = p - px * cpp;
                    e_pc[k] = p < 32 * cpp ? (px | (c << 8)) : -1;
                }
            } else {
#pragma unroll
                for (int k = 0; k < NEK; ++k) e_pc[k] = -1;
            }
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
        // LDS form of the epilogue: the dead buffer becomes the target of the next tile fetches (issued from step 0 on): every wave must be done
        // with it.  (The direct form touches no tile buffer: the barrier of the tile's last step is enough.)
        if (CD_ABL(x, 1024) || HAS_UPS || (cpp & 1)) __builtin_amdgcn_s_barrier();
        item = nitem; kt = nkt; cur = nxt;
    }
    };
    if (wload) run(cd_ic<1>{}); else run(cd_ic<0>{});
#if CD_WARMUP
    asm volatile("" :: "v"(pf));
#endif
}

#ifndef CD_KERNEL_ONLY      // (tuning aid: a translation unit that includes this file to compile single instantiations)
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
    // worth it from one 256-pixel tile per CU upwards, and from one per TWO CUs where an image is more than one tile (round 6: the plain
    // network's 32x32 stage at batch 32; not the blind-spot network's 16x16 stage, which runs beside the half-chip weight-gradient launch:
    // ssdn/hip/graph.py::cdma_fills mirrors the rule and says what was measured)
    int cus = ssdn_device_cus();
    if (cus <= 0) cus = 256;
    const long long tiles = (long long)a->N * (a->H >> 4) * (a->W >> 4);
    if (!(tiles >= cus || (2 * tiles >= cus && a->H * a->W > 256)) && !any_size) return false;
    if ((long long)a->N * (a->W >> 4) >= 65536) return false;    // (strips: the tile walk's reciprocal multiply, CdAux.tx_magic)
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

template <int MT, bool BF, int EPI, int KIND>
static int cd_launch_kind(const ssdn_conv_args* a, CdAux x, hipStream_t s) {
    static bool attr_set_dev[SSDN_MAX_DEVICES_ATTR] = {};
    bool& attr_set = attr_set_dev[ssdn_current_device_slot()];   // (function attributes are per device)
    if (!attr_set) {
        SSDN_CHECK_HIP(hipFuncSetAttribute((const void*)k_cdma<MT, BF, EPI, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    int cus = ssdn_device_cus();
    if (cus <= 0) return -1;
    // cut the strips into segments until there are two work items per CU (or one tile per item)
    const int nstrips = a->N * x.tiles_x;
    x.segs = 1;
    while (nstrips * x.segs < 2 * cus && x.segs < x.tiles_y && x.tiles_y % (x.segs * 2) == 0) x.segs *= 2;
    x.tps = x.tiles_y / x.segs;
    x.segs_sh = 0;
    while ((1 << x.segs_sh) < x.segs) ++x.segs_sh;
    x.tx_magic = x.tiles_x > 1 ? (unsigned)((0x100000000ull + x.tiles_x - 1) / x.tiles_x) : 0u;
    x.nitems = nstrips * x.segs;
    const int grid = x.nitems < 2 * cus ? x.nitems : 2 * cus;
    x.xcd_map = (x.nitems % 8 == 0 && grid % 8 == 0) ? 1 : 0;
    const double px = (double)a->N * a->H * a->W;
    const int kreal = a->kreal > 0 ? a->kreal : a->Ktot;
    const double flops = 2.0 * px * x.m_cnt * kreal * 9;
    const double bytes = px * (a->c0 * 2.0 / (a->up0 ? 4.0 : 1.0) + a->c1 * 2.0) + px * x.m_cnt * 2.0;
    prof_begin(MT == 3 ? SSDN_PROF_CDMA_MT3 : SSDN_PROF_CDMA_MT21, s);
    SSDN_LAUNCH((k_cdma<MT, BF, EPI, KIND>), dim3(grid), dim3(256), conv_dma_lds_bytes(MT), s, *a, x);
    prof_end(MT == 3 ? SSDN_PROF_CDMA_MT3 : SSDN_PROF_CDMA_MT21, s, flops, bytes);
    return 0;
}
template <int MT, bool BF, int EPI>
static int cd_launch(const ssdn_conv_args* a, const CdAux& x, hipStream_t s) {
    if (x.tail16 == 0 && x.nfull > 0) return cd_launch_kind<MT, BF, EPI, 1>(a, x, s);
    if constexpr (!BF) {      // (a 16-channel tail behind full chunks: the forward role of decode_block_1.0; anything else takes the general form)
        if (x.tail16 == 1 && x.nfull > 0) return cd_launch_kind<MT, BF, EPI, 2>(a, x, s);
    }
    return cd_launch_kind<MT, BF, EPI, 0>(a, x, s);
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
    x.segs = 1; x.tps = x.tiles_y; x.nitems = 0; x.xcd_map = 0; x.segs_sh = 0; x.tx_magic = 0;
    x.nfull = a->Ktot / 48; x.tail16 = (a->Ktot % 48) ? 1 : 0;
    static const int env_ablate = [] { const char* e = ssdn_tuning_env("SSDN_CDMA_ABLATE"); return e ? atoi(e) : 0; }();
    x.ablate = env_ablate;
    static const int env_wrep = [] { const char* e = ssdn_tuning_env("SSDN_CDMA_WREP"); return e ? atoi(e) : 0; }();
    x.wrep = env_wrep;
    static const int env_dephase = [] { const char* e = ssdn_tuning_env("SSDN_CDMA_DEPHASE"); return e ? atoi(e) : CD_DEPHASE; }();
    x.dephase = env_dephase;
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

#endif

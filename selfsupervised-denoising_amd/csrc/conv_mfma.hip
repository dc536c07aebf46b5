// conv_mfma.hip -- im2col-free implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x16_{f16,bf16}).
//
// One kernel serves every 3x3 / 1x1 convolution of the U-Net in both roles:
//   forward       out = lrelu(bias + sum_t W_t * in(y+dy_t, x+dx_t))          (ShiftConv2d / Conv2d, noise_network.py:58-156,241-260)
//   data gradient dX  = (sum_t W_t^T * dZ(y-dy_t, x-dx_t) [+ skip grad]) * lrelu'(X)   (autograd's conv backward-data)
// The blind-spot shift, nearest-upsample and channel-concat of the reference are folded into the tap offsets and the
// two-source tile loader; nothing is padded, cropped, upsampled or concatenated in HBM.
//
// GEMM view per workgroup: D[m][pixel] += A[m][k] * B[k][pixel];  A = packed weights (rows = output channels),
// B = the input halo tile staged ONCE per channel chunk in LDS as NHWC (every tap is a different LDS offset of the same tile).
//   workgroup = 256 threads = 4 waves; tile = up to 256 output pixels (2^ltn images x 2^lth rows x 2^ltw cols)
//   wave w owns pixels [64w, 64w+64) = two 32-wide MFMA column tiles, and all MT (<=3) 32-row output-channel tiles
//   LDS: halo tile [TN][TH+padT+padB][TW+padL+padR] pixels x (KC fp16 + 16 B pad)  +  TWO weight slices [32*MT][KC] (+pad).
//   Pixel / weight-row stride = 16 B x odd  =>  ds_read_b128 of 16 different pixels hits 16 different 16-B bank slots.
// Software pipeline over (channel chunk, tap) steps: the weight slices of steps s+1 and s+2 are in flight global->registers
// while step s runs on the matrix cores; a slice is committed to the idle LDS buffer one step before its use; one barrier
// per step.  Inside a step the KS = KC/16 K-steps are fully unrolled with ping-pong fragment registers and immediate LDS
// offsets (KC is a template parameter): PMC showed the first version issuing 12 VALU per MFMA -- register moves and address
// arithmetic -- which is what this structure removes.
// Epilogue: bias (+LeakyReLU) in registers, tile transposed through LDS, 16-byte pixel-contiguous stores.
#include "common.h"
// tuning aids (ablation bits, s_memtime stamps, start-up desynchronisation) exist only in -DSSDN_TUNING builds: as run-time flags
// they are instructions and branches in kernels that are bound by exactly those
#ifdef SSDN_TUNING
#define CV_ABL(xx, bit) (((xx).ablate & (bit)) != 0)
#define CV_TUNING 1
#else
#define CV_ABL(xx, bit) false
#define CV_TUNING 0
#endif
#include <cstdlib>


struct ConvGeom {
    int TW, TH, TN, HH, HW, padT, padB, padL, padR;
    int PSTR, WSTR, NP;  // bytes, bytes, halo pixels
    int tiles_x, tiles_y, groups_n;
};

static __host__ __device__ inline ConvGeom conv_geom(int ltw, int lth, int ltn, int ntaps, const int* dy, const int* dx,
                                                     int N, int H, int W, int kc) {
    ConvGeom g;
    g.TW = 1 << ltw; g.TH = 1 << lth; g.TN = 1 << ltn;
    int mny = 0, mxy = 0, mnx = 0, mxx = 0;
    for (int t = 0; t < ntaps; ++t) {
        mny = dy[t] < mny ? dy[t] : mny; mxy = dy[t] > mxy ? dy[t] : mxy;
        mnx = dx[t] < mnx ? dx[t] : mnx; mxx = dx[t] > mxx ? dx[t] : mxx;
    }
    g.padT = -mny; g.padB = mxy; g.padL = -mnx; g.padR = mxx;
    g.HH = g.TH + g.padT + g.padB;
    g.HW = g.TW + g.padL + g.padR;
    g.NP = g.TN * g.HH * g.HW;
    g.PSTR = kc * 2 + 16;
    g.WSTR = kc * 2 + 16;
    g.tiles_x = (W + g.TW - 1) / g.TW;
    g.tiles_y = (H + g.TH - 1) / g.TH;
    g.groups_n = (N + g.TN - 1) / g.TN;
    return g;
}

// 1-tap (1x1) layers with several channel chunks use the asynchronous tile pipeline (see k_conv): two unpadded LDS tile
// buffers, chunk c+1 fetched with buffer_load ... lds while chunk c is on the matrix cores.  Needs every chunk to come from
// one plain (not upsampled) source tensor.
static __host__ __device__ inline bool conv_async(const ssdn_conv_args& a, int kc) {
    if (a.ntaps != 1 || a.Ktot / kc < 2 || a.dy[0] != 0 || a.dx[0] != 0) return false;
    if (a.c0 > 0 && a.up0) return false;
    return a.c1 == 0 || a.c0 == 0 || a.c0 % kc == 0;
}

// exact x / d for x*d < 2^32 via a 32-bit magic reciprocal; magic == 0 encodes d == 1 (its reciprocal does not fit)
static __device__ __forceinline__ unsigned fdiv(unsigned x, unsigned magic) { return magic ? __umulhi(x, magic) : x; }
static inline unsigned magic_of(unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }
static __device__ __forceinline__ unsigned magic_dev(unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }

// ALLW (latency-bound layers that run as 32-channel blocks, 3x3): the whole [9 taps][32 rows][kc] weight block of a channel
// chunk is staged together with the tile, ONE barrier, then all 9 taps x kc/16 K-steps run back to back -- instead of 9..27
// pipeline steps of 6..12 MFMAs each separated by a barrier and a weight commit (a small layer's launch IS that chain).
static __host__ __device__ inline bool conv_allw(const ssdn_conv_args& a, const ConvGeom& g, int mt, int kc) {
    if (mt != 1 || a.ntaps != 9 || a.dst32) return false;
    const size_t str = (size_t)kc * 2 + 16;
    return (size_t)g.NP * str + 9u * 32u * str <= 160u * 1024u;
}

struct ConvAux {  // host-computed helpers passed by value
    unsigned mg_hw, mg_hh;  // magic reciprocals of HW and HH
    unsigned mg_ntaps;      // magic reciprocal of ntaps
    int m_base;             // first output channel of this launch (multiple of 32)
    int nblk;               // blocks of MT*32 output channels in this launch (consecutive workgroups share a pixel tile)
    int ablate;             // tuning aid (env SSDN_CONV_ABLATE): 1 no MFMA, 2 no tile staging, 4 no weight stream, 8 no stores
    unsigned long long* trace;   // tuning aid (ssdn_debug_set_trace): 32 s_memtime stamps per workgroup, or NULL
    int desync;             // first-round workgroups start (hash(block) & 7) * desync * 8128 cycles late (0 = off)
    int allw;               // all taps' weights of a channel chunk resident in LDS: no per-step weight stream / barriers
    int flat;               // allw + the tile is 256 pixels of WHOLE images: flat staging / epilogue, register prefetch of chunk c+1
};

static unsigned long long* g_conv_trace = nullptr;
extern "C" void ssdn_debug_set_trace(void* p) { g_conv_trace = (unsigned long long*)p; }
extern "C" void* ssdn_debug_get_trace() { return g_conv_trace; }

template <bool BF>
static __device__ __forceinline__ f32x16 mma(half8 av, half8 bv, f32x16 c) {
    if constexpr (BF)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
}

// BF = false: fp16 operands / fp16 output (forward);  BF = true: bf16 operands / bf16 output (data gradient).
// Tiles are moved through LDS as raw 16-bit words, so only the MFMA opcode and the epilogue conversions differ.
// KS = channel chunk / 16 (K-steps per pipeline step).  CONV_THREADS = 256 (4 waves, tile <= 256 pixels) or 512 (8 waves,
// tile <= 512 pixels): every workgroup streams the whole weight tensor through the CU's vector-memory path once per tile, and
// on the 96-channel layers that stream (166 KB per tile) outweighs the activations -- the wide variant halves it per pixel.
template <int MT, bool BF, int KS, int CONV_THREADS>
__global__ __launch_bounds__(CONV_THREADS, 2) void k_conv(ssdn_conv_args a, ConvAux x) {
    constexpr int KC = KS * 16;            // channels per chunk
    constexpr int CC8 = KS * 2;            // 16-byte pieces per pixel / weight row
    constexpr int STR = KC * 2 + 16;       // LDS stride of a pixel and of a weight row (bytes)
    constexpr int WROWS = MT * 32;
    constexpr int NW = (WROWS * CC8 + CONV_THREADS - 1) / CONV_THREADS;   // 16-B registers per thread per weight slice
    constexpr int WBUF = WROWS * STR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvGeom g = conv_geom(a.ltw, a.lth, a.ltn, a.ntaps, a.dy, a.dx, a.N, a.H, a.W, KC);
    // blockIdx.x % nblk: block of MT*32 output channels (the 384-channel 1x1 layers are 4 such blocks; as ONE launch the
    // four workgroups that share a pixel tile are dispatched back to back and share its input through L2)
    // Workgroups are dealt to the 8 XCDs round-robin by id and each XCD has its own L2: the nblk workgroups of one pixel
    // tile get ids 8 apart (id = 8 * (nblk * (tile / 8) + block) + tile % 8), i.e. the SAME XCD, back to back.
    const unsigned wg_j = blockIdx.x >> 3, wg_xcd = blockIdx.x & 7;
    const unsigned wg_tile = x.nblk > 1 ? (wg_j / (unsigned)x.nblk) * 8 + wg_xcd : blockIdx.x;
    x.m_base += (x.nblk > 1 ? (int)(wg_j % (unsigned)x.nblk) : 0) * (MT * 32);
    if (wg_tile >= (unsigned)(g.tiles_x * g.tiles_y * g.groups_n)) return;   // (grid is rounded up to a multiple of 8 tiles)
    // ASYNC (1x1 layers, >= 2 channel chunks): the tile of chunk c+1 is fetched by the LDS-DMA path (buffer_load ... lds:
    // no registers, no ds_write, asynchronous) into the second of two UNPADDED tile buffers while chunk c is on the matrix
    // cores -- a 1x1 layer re-stages its tile for every 18 MFMAs per wave, and with synchronous staging that latency was
    // most of its time.  (LDS-DMA writes 64 consecutive 16-byte pieces per wave instruction, hence the unpadded layout.)
    const bool ASYNC = conv_async(a, KC) && !CV_ABL(x, 128);
    const int tstr = ASYNC ? KC * 2 : STR;                 // LDS stride of a tile pixel
    const int tbytes = g.NP * tstr;
    char* tile = smem;
    char* wl0 = smem + (size_t)(ASYNC ? 2 : 1) * tbytes;
    char* wl1 = wl0 + WBUF;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = wg_tile;
    const int tx_i = bid % g.tiles_x; bid /= g.tiles_x;
    const int ty_i = bid % g.tiles_y; bid /= g.tiles_y;
    const int n0 = bid * g.TN, y0 = ty_i * g.TH, x0 = tx_i * g.TW;
    const int npix = g.TN * g.TH * g.TW;

    // this lane's two output pixels (MFMA columns)
    int bbase[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        int q = wave * 64 + nt * 32 + l31;
        int tx = q & (g.TW - 1), ty = (q >> a.ltw) & (g.TH - 1), tn = q >> (a.ltw + a.lth);
        if (tn >= g.TN) tn = ty = tx = 0;   // tile smaller than the workgroup: surplus lanes compute on pixel 0, store nothing
        bbase[nt] = ((tn * g.HH + ty + g.padT) * g.HW + tx + g.padL) * tstr + kh * 16;
    }
    const int abase = l31 * STR + kh * 16;

    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    int tr_i = 0;
    auto stamp = [&]() {
        if (CV_TUNING && x.trace && tid == 0 && tr_i < 32) x.trace[(size_t)blockIdx.x * 32 + tr_i++] = __builtin_amdgcn_s_memtime();
    };
    const int H0 = a.up0 ? (a.H >> 1) : a.H, W0 = a.up0 ? (a.W >> 1) : a.W;
    const int nchunks = a.Ktot / KC;
    const int nsteps = nchunks * a.ntaps;
    const h16* s0 = (const h16*)a.src0.p;
    const h16* s1 = (const h16*)a.src1.p;
    const h16* wp = (const h16*)a.w;

    // ---- weight-slice prefetch: element e = tid + 256*i  ->  row e / CC8, 16-B piece e % CC8 (compile-time divisions) ----
    int w_goff[NW], w_loff[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int e = tid + i * CONV_THREADS;
        const int m = e / CC8, cc = e % CC8;
        // threads without an element in slot i still LOAD (element 0, discarded): unconditional, branch-free loads are what
        // lets the compiler count outstanding loads (s_waitcnt vmcnt(N)) instead of draining the whole queue (vmcnt(0))
        w_goff[i] = e < WROWS * CC8 ? ((x.m_base + m) * a.Ktot + cc * 8) : x.m_base * a.Ktot;
        w_loff[i] = e < WROWS * CC8 ? m * STR + cc * 16 : -1;
    }
    half8 wrA[NW], wrB[NW], wrC[NW];    // three register sets: the weight stream runs THREE steps ahead of the MFMA work
    // every workgroup walks the taps in a different rotation: all workgroups of a launch stream the SAME 166 KB of weights,
    // and in lock-step they would all hit the same few L2 channels with the same 9 KB slice at the same moment
    const int rot = (CV_ABL(x, 64) || x.allw) ? 0 : (int)(wg_tile % (unsigned)a.ntaps);
    auto tap_of = [&](int tseq) { int t = tseq + rot; return t >= a.ntaps ? t - a.ntaps : t; };
    auto w_issue = [&](half8 (&wr)[NW], int step) {
        step = step < nsteps ? step : nsteps - 1;      // past the end: re-load the last slice (never committed)
        const int ch = fdiv(step, x.mg_ntaps), t = tap_of(step - ch * a.ntaps);
        const h16* base = wp + (long long)t * a.Mpad * a.Ktot + ch * KC;
#pragma unroll
        for (int i = 0; i < NW; ++i) wr[i] = ld_h8(base + w_goff[i]);
    };
    auto w_commit = [&](half8 (&wr)[NW], char* buf) {
#pragma unroll
        for (int i = 0; i < NW; ++i)
            if (w_loff[i] >= 0) *reinterpret_cast<half8*>(buf + w_loff[i]) = wr[i];
    };

    // ---- halo tile staging: flat index f = tid + 256*j over (halo pixel, 16-B piece); 8 loads in flight per thread ----
    const int nflat = g.NP * CC8;
    auto stage_tile = [&](int ch) {
        if (CV_ABL(x, 2)) return;
        for (int f0 = tid; f0 < nflat; f0 += 8 * CONV_THREADS) {
            half8 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int f = f0 + u * CONV_THREADS;
                const int hp = f / CC8, cc = f % CC8;
                const unsigned r1 = fdiv(hp, x.mg_hw);
                const int hx = hp - r1 * g.HW;
                const unsigned tn = fdiv(r1, x.mg_hh);
                const int hy = r1 - tn * g.HH;
                const int n = n0 + tn, y = y0 - g.padT + hy, xx = x0 - g.padL + hx;
                const int k = ch * KC + cc * 8;
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
            for (int u = 0; u < 8; ++u) {
                const int f = f0 + u * CONV_THREADS;
                if (f < nflat) *reinterpret_cast<half8*>(tile + (f / CC8) * tstr + (f % CC8) * 16) = v[u];
            }
        }
    };

    // ---- ASYNC: LDS-DMA fetch of channel chunk `ch` into tile buffer `buf` (no halo: ntaps == 1) ------------------------
    // flat piece f = tid + 256 u  ->  (pixel f / CC8, 16-byte piece f % CC8); lane i of a wave instruction lands at
    // M0 + 16 i, i.e. at piece (wave*64 + 256 u + i) of the unpadded tile.  Out-of-image pixels get an offset beyond
    // num_records: the hardware writes zeros.
    auto async_issue = [&](int ch, int buf) __attribute__((always_inline)) {
        const int k0 = ch * KC;
        const bool from0 = k0 < a.c0;
        const unsigned long long bp = (unsigned long long)((const h16*)(from0 ? a.src0.p : a.src1.p) + (from0 ? a.src0.co + k0 : a.src1.co + k0 - a.c0));
        const int scs = from0 ? a.src0.cs : a.src1.cs;
        const u32x4_t rs = {(unsigned)bp, (unsigned)(bp >> 32) & 0xffffu, (unsigned)(((long long)a.N * a.H * a.W - 1) * scs + KC) * 2u, SSDN_BUFFER_RSRC_FLAGS};
        for (int f0 = 0; f0 < nflat; f0 += CONV_THREADS) {
            const int f = f0 + tid;
            const int q = f / CC8, cc = f % CC8;
            const int tx = q & (g.TW - 1), ty = (q >> a.ltw) & (g.TH - 1), tn = q >> (a.ltw + a.lth);
            const int n = n0 + tn, y = y0 + ty, xx = x0 + tx;
            const bool ok = f < nflat && n < a.N && y < a.H && xx < a.W;
            const int voff = ok ? (((n * a.H + y) * a.W + xx) * scs + cc * 8) * 2 : (int)0x80000000;
            const unsigned ldsbase = (unsigned)(size_t)(tile + buf * tbytes) + (unsigned)(f0 + wave * 64) * 16u;
            asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(ldsbase), "v"(voff), "s"(rs) : "memory");
        }
    };
    auto async_wait = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    // ---- one pipeline step on the matrix cores: KS K-steps, fully unrolled, ping-pong fragments, immediate offsets ----
    auto compute = [&](const char* wl, int step) {
        if (CV_ABL(x, 1)) return;
        const int ch = fdiv(step, x.mg_ntaps), t = tap_of(step - ch * a.ntaps);
        const int toff = (a.dy[t] * g.HW + a.dx[t]) * tstr;
        const char* tcur = tile + (ASYNC ? (step & 1) * tbytes : 0);
        const char* b0p = tcur + bbase[0] + toff;
        const char* b1p = tcur + bbase[1] + toff;
        const char* ap = wl + abase;
        half8 bq[2][2], aq[2][MT];
        bq[0][0] = *reinterpret_cast<const half8*>(b0p);
        bq[0][1] = *reinterpret_cast<const half8*>(b1p);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) aq[0][mt] = *reinterpret_cast<const half8*>(ap + mt * 32 * STR);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < KS) {
                bq[nxt][0] = *reinterpret_cast<const half8*>(b0p + (ks + 1) * 32);
                bq[nxt][1] = *reinterpret_cast<const half8*>(b1p + (ks + 1) * 32);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) aq[nxt][mt] = *reinterpret_cast<const half8*>(ap + mt * 32 * STR + (ks + 1) * 32);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt][0] = mma<BF>(aq[cur][mt], bq[cur][0], acc[mt][0]);
                acc[mt][1] = mma<BF>(aq[cur][mt], bq[cur][1], acc[mt][1]);
            }
        }
    };
    // after the MFMA work of `step`: make step+1 runnable (re-stage the tile if it starts a new channel chunk, move its
    // prefetched weights registers -> idle LDS buffer), then ONE barrier
    auto advance = [&](half8 (&wr_next)[NW], char* buf_next, int step) {
        if (step + 1 < nsteps) {
            const int ch = fdiv(step, x.mg_ntaps), nch = fdiv(step + 1, x.mg_ntaps);
            if (nch != ch && !ASYNC) {
                __syncthreads();
                stage_tile(nch);
            }
            w_commit(wr_next, buf_next);
        }
        if (ASYNC) async_wait();          // the tile of step+1 has landed (every wave waits for its own loads, then the barrier)
        if (!CV_ABL(x, 32)) __syncthreads();
        if (ASYNC && step + 2 < nsteps) async_issue(step + 2, step & 1);   // the buffer of `step` is free now
        stamp();
    };

    bool flat_done = false;
    if constexpr (MT == 1 && CONV_THREADS == 256 && KS <= 4) {
    if (x.flat) {
        // ---- FLAT ALLW (layers of whole-image tiles: 16x16 pixels and below) ------------------------------------------------
        // The tile's 256 pixels are 256 CONSECUTIVE pixels of the NHWC tensors (whole images), so staging and the epilogue
        // are flat loops with shift arithmetic; the halo ring is zeroed once (it never changes); chunk c+1 (tile + all nine
        // weight slices, 13..26 16-byte loads per thread) is prefetched into registers while chunk c is on the matrix cores.
        // These launches are chains of dependent latencies (a handful of workgroups): before, 2-3 exposed load batches per
        // chunk, a bias round trip and a row-by-row epilogue made a 2x2-pixel layer cost 43 K cycles for 1.7 K cycles of MFMA.
        stamp();
        constexpr int NWQ = (9 * WROWS * CC8 + CONV_THREADS - 1) / CONV_THREADS;
        constexpr int wtotal = 9 * WROWS * CC8;
        half8 pw[NWQ], pt[CC8];
        const int lhw = a.ltw + a.lth;
        const long long pixb = (long long)n0 << lhw;               // first pixel of the tile (TW == W, TH == H)
        auto issue_chunk = [&](int ch) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NWQ; ++i) {
                const int e = tid + i * CONV_THREADS;
                const int ee = e < wtotal ? e : 0;
                const int t = ee / (WROWS * CC8), r = ee - t * (WROWS * CC8), m = r / CC8, cc = r - m * CC8;
                pw[i] = ld_h8(wp + ((long long)t * a.Mpad + x.m_base + m) * a.Ktot + ch * KC + cc * 8);
            }
#pragma unroll
            for (int u = 0; u < CC8; ++u) {
                const int f = tid + u * CONV_THREADS;
                const int q = f / CC8, cc = f - q * CC8;
                const int tn = q >> lhw;
                const int k = ch * KC + cc * 8;
                pt[u] = zero_h8();
                if (n0 + tn < a.N) {
                    if (k < a.c0) {
                        if (a.up0) {
                            const int ty = (q >> a.ltw) & (g.TH - 1), tx = q & (g.TW - 1);
                            pt[u] = ld_h8(s0 + ((((long long)(n0 + tn) * H0 + (ty >> 1)) * W0 + (tx >> 1)) * a.src0.cs + a.src0.co + k));
                        } else {
                            pt[u] = ld_h8(s0 + ((pixb + q) * a.src0.cs + a.src0.co + k));
                        }
                    } else {
                        pt[u] = ld_h8(s1 + ((pixb + q) * a.src1.cs + a.src1.co + (k - a.c0)));
                    }
                }
            }
        };
        issue_chunk(0);
        float bias_r = 0.f;
        if (tid < WROWS && a.bias && x.m_base + tid < a.M) bias_r = a.bias[x.m_base + tid];
        for (int z = tid * 16; z < tbytes; z += CONV_THREADS * 16) *reinterpret_cast<half8*>(tile + z) = zero_h8();
        __syncthreads();
        for (int ch = 0; ch < nchunks; ++ch) {
#pragma unroll
            for (int i = 0; i < NWQ; ++i) {
                const int e = tid + i * CONV_THREADS;
                if (e < wtotal) {
                    const int t = e / (WROWS * CC8), r = e - t * (WROWS * CC8), m = r / CC8, cc = r - m * CC8;
                    *reinterpret_cast<half8*>(wl0 + t * WBUF + m * STR + cc * 16) = pw[i];
                }
            }
#pragma unroll
            for (int u = 0; u < CC8; ++u) {
                const int f = tid + u * CONV_THREADS;
                const int q = f / CC8, cc = f - q * CC8;
                const int tx = q & (g.TW - 1), ty = (q >> a.ltw) & (g.TH - 1), tn = q >> lhw;
                *reinterpret_cast<half8*>(tile + ((tn * g.HH + ty + g.padT) * g.HW + tx + g.padL) * tstr + cc * 16) = pt[u];
            }
            __syncthreads();
            stamp();
            if (ch + 1 < nchunks) issue_chunk(ch + 1);
            if (!CV_ABL(x, 1)) {
                // the 9 x KS K-steps of the chunk as ONE software pipeline: fragments are read two K-steps ahead, across tap
                // boundaries (per-tap pipelines exposed an LDS round trip at every tap: ~250 cycles per 2-MFMA K-step)
                constexpr int S = 9 * KS;
                half8 fa[3], f0[3], f1[3];
                auto rd = [&](int sq, int slot) __attribute__((always_inline)) {
                    const int t = sq / KS, ks = sq - t * KS;
                    const int toff = (a.dy[t] * g.HW + a.dx[t]) * tstr + ks * 32;
                    f0[slot] = *reinterpret_cast<const half8*>(tile + bbase[0] + toff);
                    f1[slot] = *reinterpret_cast<const half8*>(tile + bbase[1] + toff);
                    fa[slot] = *reinterpret_cast<const half8*>(wl0 + t * WBUF + abase + ks * 32);
                };
                rd(0, 0);
                rd(1, 1);
                for (int sq = 0; sq < S; sq += 3) {
                    rd(sq + 2, 2);
                    acc[0][0] = mma<BF>(fa[0], f0[0], acc[0][0]);
                    acc[0][1] = mma<BF>(fa[0], f1[0], acc[0][1]);
                    if (sq + 3 < S) rd(sq + 3, 0);
                    acc[0][0] = mma<BF>(fa[1], f0[1], acc[0][0]);
                    acc[0][1] = mma<BF>(fa[1], f1[1], acc[0][1]);
                    if (sq + 4 < S) rd(sq + 4, 1);
                    acc[0][0] = mma<BF>(fa[2], f0[2], acc[0][0]);
                    acc[0][1] = mma<BF>(fa[2], f1[2], acc[0][1]);
                }
            }
            stamp();
            __syncthreads();
        }
        if (CV_ABL(x, 16)) return;
        // ---- flat epilogue: registers -> LDS [pixel][OSTR] -> 16-byte pieces of 256 consecutive pixels ----
        constexpr int OSTRF = MT * 64 + 16;
        char* otf = smem;
        float* blf = reinterpret_cast<float*>(smem + (size_t)CONV_THREADS * OSTRF);
        if (tid < WROWS) blf[tid] = bias_r;
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int q = wave * 64 + nt * 32 + l31;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int ml = gq * 8 + kh * 4;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(blf + ml);
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = acc[0][nt][gq * 4 + j] + bb[j];
                    if (a.act) v[j] = lrelu(v[j]);
                }
                u32x2_t o;
                o[0] = BF ? pack_bf16x2(v[0], v[1]) : pack_f16x2(v[0], v[1]);
                o[1] = BF ? pack_bf16x2(v[2], v[3]) : pack_f16x2(v[2], v[3]);
                *reinterpret_cast<u32x2_t*>(otf + q * OSTRF + ml * 2) = o;
            }
        }
        __syncthreads();
        stamp();
        int m_cntf = a.M - x.m_base;
        m_cntf = m_cntf > WROWS ? WROWS : m_cntf;
        const int lcpp = m_cntf >= 32 ? 2 : (m_cntf >= 16 ? 1 : 0);          // pieces per pixel: 4, 2 or 1 (m_cnt = 32, 16, 8)
        const bool has_maskf = a.mask.p != nullptr, has_addf = a.add.p != nullptr;
        const int npieces = CONV_THREADS << lcpp;
        half8 mk[4], ad[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + u * CONV_THREADS;
            const int q = e >> lcpp, c = e & ((1 << lcpp) - 1);
            const bool on = e < npieces && n0 + (q >> lhw) < a.N;
            mk[u] = zero_h8(); ad[u] = zero_h8();
            if (on && has_maskf) mk[u] = ld_h8((const h16*)a.mask.p + ((pixb + q) * a.mask.cs + a.mask.co + x.m_base + c * 8));
            if (on && has_addf) ad[u] = ld_h8((const h16*)a.add.p + ((pixb + q) * a.add.cs + a.add.co + x.m_base + c * 8));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + u * CONV_THREADS;
            const int q = e >> lcpp, c = e & ((1 << lcpp) - 1);
            const bool on = e < npieces && n0 + (q >> lhw) < a.N && !(a.upsum.p && x.m_base + c * 8 < a.upsum_c);
            if (!on) continue;
            u32x4_t o = *reinterpret_cast<const u32x4_t*>(otf + q * OSTRF + c * 16);
            if (has_addf || has_maskf) {
                const u32x4_t ab = __builtin_bit_cast(u32x4_t, ad[u]), mb = __builtin_bit_cast(u32x4_t, mk[u]);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    float v0, v1;
                    if constexpr (BF) {
                        v0 = bf_lo(o[w]) + (has_addf ? bf_lo(ab[w]) : 0.f);
                        v1 = bf_hi(o[w]) + (has_addf ? bf_hi(ab[w]) : 0.f);
                    } else {
                        v0 = f16_lo(o[w]) + (has_addf ? f16_lo(ab[w]) : 0.f);
                        v1 = f16_hi(o[w]) + (has_addf ? f16_hi(ab[w]) : 0.f);
                    }
                    if (has_maskf) {
                        v0 *= lrelu_grad(f16_lo(mb[w]));
                        v1 *= lrelu_grad(f16_hi(mb[w]));
                    }
                    o[w] = BF ? pack_bf16x2(v0, v1) : pack_f16x2(v0, v1);
                }
            }
            *reinterpret_cast<u32x4_t*>((h16*)a.dst.p + ((pixb + q) * a.dst.cs + a.dst.co + x.m_base + c * 8)) = o;
        }
        if constexpr (BF) {
        if (a.upsum.p) {
            // fused SSDN_OP_UPSUM_BWD for the channels below upsum_c: 2x2 sums (scan order, fp32) of the bf16 values in the
            // transposed tile, times LeakyReLU'(upsum_mask), to the half-resolution tensor
            const int e = tid;
            const int pq = e >> lcpp, c = e & ((1 << lcpp) - 1);
            const int lw2 = a.ltw - 1, lh2 = a.lth - 1;
            const int pj = pq & ((1 << lw2) - 1), pi = (pq >> lw2) & ((1 << lh2) - 1), tn = pq >> (lw2 + lh2);
            if (e < (64 << lcpp) && tn < g.TN && x.m_base + c * 8 < a.upsum_c) {
                const long long pp = (((((long long)(n0 + tn)) << lh2) + pi) << lw2) + pj;
                const u32x4_t um = *reinterpret_cast<const u32x4_t*>((const h16*)a.upsum_mask.p + (pp * a.upsum_mask.cs + a.upsum_mask.co + x.m_base + c * 8));
                float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const u32x4_t o = *reinterpret_cast<const u32x4_t*>(otf + (((tn << a.lth) + 2 * pi + (q4 >> 1)) * g.TW + 2 * pj + (q4 & 1)) * OSTRF + c * 16);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { sum[2 * q] += bf_lo(o[q]); sum[2 * q + 1] += bf_hi(o[q]); }
                }
                u32x4_t r;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    r[q] = pack_bf16x2(sum[2 * q] * lrelu_grad(f16_lo(um[q])), sum[2 * q + 1] * lrelu_grad(f16_hi(um[q])));
                *reinterpret_cast<u32x4_t*>((h16*)a.upsum.p + (pp * a.upsum.cs + a.upsum.co + x.m_base + c * 8)) = r;
            }
        }
        }
        if (a.pool.p) {
            // fused Shift2d((1,0)) + MaxPool2d(2) (SSDN_OP_POOL_FWD): 64 pooled pixels x (1 << lcpp) pieces, straight from the
            // transposed tile (the rounded values just stored); shifted: rows {2i-1, 2i}, row -1 is a literal 0 in the max
            const int e = tid;
            const int pq = e >> lcpp, c = e & ((1 << lcpp) - 1);
            const int lw2 = a.ltw - 1, lh2 = a.lth - 1;
            const int pj = pq & ((1 << lw2) - 1), pi = (pq >> lw2) & ((1 << lh2) - 1), tn = pq >> (lw2 + lh2);
            if (e < (64 << lcpp) && tn < g.TN) {
                const int r0 = a.pool_shifted ? 2 * pi - 1 : 2 * pi;
                u32x4_t best;
                bool have = false;
#pragma unroll
                for (int dr = 0; dr < 2; ++dr) {
                    const int r = r0 + dr;
#pragma unroll
                    for (int dc = 0; dc < 2; ++dc) {
                        u32x4_t v = {0u, 0u, 0u, 0u};
                        if (r >= 0) v = *reinterpret_cast<const u32x4_t*>(otf + (((tn << a.lth) + r) * g.TW + 2 * pj + dc) * OSTRF + c * 16);
                        if (!have) { best = v; have = true; }
                        else {
                            const half8 m = __builtin_elementwise_max(__builtin_bit_cast(half8, best), __builtin_bit_cast(half8, v));
                            best = __builtin_bit_cast(u32x4_t, m);
                        }
                    }
                }
                const long long pp = (((((long long)(n0 + tn)) << lh2) + pi) << lw2) + pj;
                *reinterpret_cast<u32x4_t*>((h16*)a.pool.p + (pp * a.pool.cs + a.pool.co + x.m_base + c * 8)) = best;
            }
        }
        stamp();
        flat_done = true;
    }
    }
    if (flat_done) return;
    if (x.allw) {
        stamp();
        // ---- ALLW: per channel chunk, stage tile + all nine weight slices, one barrier, 9 x KS K-steps barrier-free ----
        const int wtotal = a.ntaps * WROWS * CC8;                    // 16-byte pieces of one chunk's weight block
        for (int ch = 0; ch < nchunks; ++ch) {
            if (ch > 0) __syncthreads();
            for (int e0 = tid; e0 < wtotal; e0 += 8 * CONV_THREADS) {
                half8 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u * CONV_THREADS;
                    const int ee = e < wtotal ? e : 0;
                    const int t = ee / (WROWS * CC8), r = ee - t * (WROWS * CC8), m = r / CC8, cc = r - m * CC8;
                    v[u] = ld_h8(wp + ((long long)t * a.Mpad + x.m_base + m) * a.Ktot + ch * KC + cc * 8);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u * CONV_THREADS;
                    if (e < wtotal) {
                        const int t = e / (WROWS * CC8), r = e - t * (WROWS * CC8), m = r / CC8, cc = r - m * CC8;
                        *reinterpret_cast<half8*>(wl0 + t * WBUF + m * STR + cc * 16) = v[u];
                    }
                }
            }
            stage_tile(ch);
            __syncthreads();
            stamp();
            for (int t = 0; t < a.ntaps; ++t) compute(wl0 + t * WBUF, ch * a.ntaps + t);
            stamp();
        }
        __syncthreads();
        stamp();
    } else {
    // buffers alternate wl0 / wl1 by step parity; register sets rotate A, B, C by step mod 3.
    // invariant at the top of step s: LDS buffer s&1 holds W(s); W(s+1), W(s+2) are in flight in their register sets.
    if (CV_TUNING && x.desync && blockIdx.x < 512) {
        const int k = (int)((blockIdx.x * 2654435761u) >> 29) * x.desync;
        for (int i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(127);
    }
    stamp();
    w_issue(wrA, 0);
    w_issue(wrB, 1);
    w_issue(wrC, 2);
    if (ASYNC) async_issue(1, 1);
    stage_tile(0);
    w_commit(wrA, wl0);
    __syncthreads();
    stamp();
    for (int step = 0; step < nsteps; step += 6) {
        w_issue(wrA, step + 3);
        compute(wl0, step);
        advance(wrB, wl1, step);
        if (step + 1 >= nsteps) break;
        w_issue(wrB, step + 4);
        compute(wl1, step + 1);
        advance(wrC, wl0, step + 1);
        if (step + 2 >= nsteps) break;
        w_issue(wrC, step + 5);
        compute(wl0, step + 2);
        advance(wrA, wl1, step + 2);
        if (step + 3 >= nsteps) break;
        w_issue(wrA, step + 6);
        compute(wl1, step + 3);
        advance(wrB, wl0, step + 3);
        if (step + 4 >= nsteps) break;
        w_issue(wrB, step + 7);
        compute(wl0, step + 4);
        advance(wrC, wl1, step + 4);
        if (step + 5 >= nsteps) break;
        w_issue(wrC, step + 8);
        compute(wl1, step + 5);
        advance(wrA, wl0, step + 5);
    }
    }

    if (CV_ABL(x, 16)) return;
    // ---- epilogue: D row = 8*(r>>2) + 4*(lane>>5) + (r&3)  (output channel), D col = lane&31 (pixel) -------------
    if (a.dst32) {
        // fp32 NCHW planar output (net_out of the last 1x1 layer: M <= 9 channels): direct stores, coalesced along x
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int q = wave * 64 + nt * 32 + l31;
            const int tx = q & (g.TW - 1), ty = (q >> a.ltw) & (g.TH - 1), tn = q >> (a.ltw + a.lth);
            const int n = n0 + tn, y = y0 + ty, xx = x0 + tx;
            if (q >= npix || n >= a.N || y >= a.H || xx >= a.W || CV_ABL(x, 8)) continue;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = x.m_base + mt * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
                    if (m >= a.M) continue;
                    float v = acc[mt][nt][r];
                    if (a.bias) v += a.bias[m];
                    if (a.act) v = lrelu(v);
                    a.dst32[(((long long)n * a.M + m) * a.H + y) * a.W + xx] = v;
                }
        }
        return;
    }
    // 16-bit NHWC output: the tile is transposed through LDS (the input tile is dead after the last barrier) so that HBM
    // sees whole 16-byte-per-lane, pixel-contiguous stores instead of 8-byte fragments of every cache line
    // (measured on the 96-channel full-resolution layers: 60-75 us of a 190 us launch were the fragmented stores).
    constexpr int OSTR = MT * 64 + 16;
    char* ot = smem;
    // ---- LDS -> HBM goes one image ROW of the tile at a time ------------------------------------------------------------
    // A row of the tile is TW pixels x cpp 16-byte pieces; its position (image, y), validity and base addresses are
    // wave-uniform and live on the scalar unit as buffer resources whose num_records is the part of the row inside the image,
    // so the hardware drops the stores (and zero-fills the mask / skip-gradient loads) of pixels past the right edge; per lane
    // only (pixel, piece) -> two byte offsets remain.  (The per-piece index arithmetic of the flat version was ~40 VALU
    // instructions per 16 bytes and, with the mask and skip-gradient handled in scalar bf16 emulation, made the epilogue of
    // the data-gradient role 38 K of its 84 K cycles per tile.)
    int m_cnt = a.M - x.m_base;
    m_cnt = m_cnt > WROWS ? WROWS : m_cnt;
    const int cpp = m_cnt >> 3;                       // 16-byte pieces per pixel
    const unsigned mg = magic_dev(cpp);
    const int rowp = g.TW * cpp;                      // pieces per row
    const int ipr = (rowp + 63) >> 6;                 // 64-lane instructions per row
    const int nrows = g.TN * g.TH;
    constexpr int NWAVES = CONV_THREADS / 64;
    const int items = ((nrows + NWAVES - 1) / NWAVES) * ipr;   // (row, instruction) items of this wave
    const bool has_mask = a.mask.p != nullptr, has_add = a.add.p != nullptr;
    // bias of this launch's channels, staged once in LDS behind the output tile (a per-lane global gather of 4*12 floats
    // showed up as ~30 us on the 96-channel layers)
    float* bl = reinterpret_cast<float*>(smem + (size_t)npix * OSTR);
    if (tid < WROWS) bl[tid] = (a.bias && x.m_base + tid < a.M) ? a.bias[x.m_base + tid] : 0.f;
    __syncthreads();
    // 16-bit NHWC output: the tile is transposed through LDS (the input tile is dead after the last barrier) so that HBM sees
    // whole 16-byte-per-lane, pixel-contiguous stores instead of 8-byte fragments of every cache line
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int q = wave * 64 + nt * 32 + l31;
        if (q >= npix) continue;      // tile smaller than 256 pixels: surplus lanes have nothing to write
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int ml = mt * 32 + gq * 8 + kh * 4;
                float v[4];
                const f32x4 bb = *reinterpret_cast<const f32x4*>(bl + ml);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = acc[mt][nt][gq * 4 + j] + bb[j];
                    if (a.act) v[j] = lrelu(v[j]);
                }
                // packed round-to-nearest-even conversions (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32)
                u32x2_t o;
                o[0] = BF ? pack_bf16x2(v[0], v[1]) : pack_f16x2(v[0], v[1]);
                o[1] = BF ? pack_bf16x2(v[2], v[3]) : pack_f16x2(v[2], v[3]);
                *reinterpret_cast<u32x2_t*>(ot + q * OSTR + ml * 2) = o;
            }
        }
    }
    __syncthreads();
    stamp();
    if (CV_ABL(x, 8)) return;
    // items in flight: the mask / skip-gradient loads of a batch are all issued before any is consumed.  (Measured: batches of
    // 6 or 12, or issuing the first batch before the accumulators are converted, are slower than batches of 4.)
    constexpr int EB = 4;
    for (int it0 = 0; it0 < items; it0 += EB) {
        half8 mk[EB], ad[EB];
        __amdgpu_buffer_rsrc_t rd[EB];
        int lo[EB], go[EB];
#pragma unroll
        for (int u = 0; u < EB; ++u) {
            const int it = it0 + u;
            const int rs = it / ipr, ii = it - rs * ipr;            // scalar
            const int row = wave + NWAVES * rs;
            const int tn = row >> a.lth, ty = row & (g.TH - 1);
            const int n = n0 + tn, y = y0 + ty;
            const bool ok = it < items && row < nrows && n < a.N && y < a.H && x0 < a.W;
            const long long pix0 = ((long long)n * a.H + y) * a.W + x0;
            const int wpx = a.W - x0 < g.TW ? a.W - x0 : g.TW;      // pixels of the row inside the image
            const int j = lane + 64 * ii;
            const int px = mg ? __umulhi((unsigned)j, mg) : j, c = j - px * cpp;
            const bool on = j < rowp;
            lo[u] = ((row << a.ltw) + px) * OSTR + c * 16;
            go[u] = on ? (px * a.dst.cs + c * 8) * 2 : (int)0x80000000;
            rd[u] = __builtin_amdgcn_make_buffer_rsrc((void*)((h16*)a.dst.p + pix0 * a.dst.cs + a.dst.co + x.m_base), 0,
                                                      ok ? ((wpx - 1) * a.dst.cs + m_cnt) * 2 : 0, SSDN_BUFFER_RSRC_FLAGS);
            mk[u] = zero_h8(); ad[u] = zero_h8();
            if (has_mask) {
                __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)((const h16*)a.mask.p + pix0 * a.mask.cs + a.mask.co + x.m_base), 0, ok ? ((wpx - 1) * a.mask.cs + m_cnt) * 2 : 0,
                    SSDN_BUFFER_RSRC_FLAGS);
                mk[u] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rm, on ? (px * a.mask.cs + c * 8) * 2 : (int)0x80000000, 0, 0));
            }
            if (has_add) {
                __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)((const h16*)a.add.p + pix0 * a.add.cs + a.add.co + x.m_base), 0, ok ? ((wpx - 1) * a.add.cs + m_cnt) * 2 : 0,
                    SSDN_BUFFER_RSRC_FLAGS);
                ad[u] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(ra, on ? (px * a.add.cs + c * 8) * 2 : (int)0x80000000, 0, 0));
            }
        }
#pragma unroll
        for (int u = 0; u < EB; ++u) {
            u32x4_t o = *reinterpret_cast<const u32x4_t*>(ot + lo[u]);      // 8 outputs as 4 packed pairs (fp16 or bf16)
            if (has_add || has_mask) {
                const u32x4_t ab = __builtin_bit_cast(u32x4_t, ad[u]), mb = __builtin_bit_cast(u32x4_t, mk[u]);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    float v0, v1;
                    if constexpr (BF) {
                        v0 = bf_lo(o[w]) + (has_add ? bf_lo(ab[w]) : 0.f);
                        v1 = bf_hi(o[w]) + (has_add ? bf_hi(ab[w]) : 0.f);
                    } else {
                        v0 = f16_lo(o[w]) + (has_add ? f16_lo(ab[w]) : 0.f);
                        v1 = f16_hi(o[w]) + (has_add ? f16_hi(ab[w]) : 0.f);
                    }
                    if (has_mask) {   // LeakyReLU'(pre-activation sign): the mask is the fp16 activation, slope where it is <= 0
                        v0 *= lrelu_grad(f16_lo(mb[w]));
                        v1 *= lrelu_grad(f16_hi(mb[w]));
                    }
                    o[w] = BF ? pack_bf16x2(v0, v1) : pack_f16x2(v0, v1);
                }
            }
            __builtin_amdgcn_raw_buffer_store_b128(o, rd[u], go[u], 0, 0);
        }
    }
    stamp();
}

// Layers with at most one pixel tile per CU are latency-bound (one workgroup's pass over its tile IS the launch): they run as
// blocks of 32 output channels (MT = 1) on three times as many workgroups -- and may then take all 96 input channels of a
// chunk at once (kc = 96: half the pipeline steps, no second tile staging).
// (measured per layer of BASELINE config 2: -20..-25 % up to one tile per CU, +35 % at two tiles per CU)
static bool conv_uses_mt1(const ssdn_conv_args* a, const ConvGeom& g) {
    static int mt1_tiles = -1;
    if (mt1_tiles < 0) {
        const char* e = ssdn_tuning_env("SSDN_CONV_MT1_TILES");       // tuning override
        mt1_tiles = e ? atoi(e) : ssdn_device_cus();
        if (mt1_tiles <= 0) mt1_tiles = 256;                  // no device (planning on a CPU-only host)
    }
    return !a->dst32 && g.tiles_x * g.tiles_y * g.groups_n <= mt1_tiles && a->Mpad >= 64;
}

// which kernel serves a layer: 0 = always k_conv, 1 = k_cdma where its shape class fits AND the layer has >= 1 tile per CU
// (default), 2 = k_cdma wherever the shape class fits (lets the test-suite drive it at fixture sizes).  Initial value from
// the environment (SSDN_CONV_DMA, read once), changed with ssdn_conv_set_mode().
static int g_conv_mode = [] { const char* e = ssdn_tuning_env("SSDN_CONV_DMA"); return e ? atoi(e) : 1; }();
extern "C" int ssdn_conv_set_mode(int mode) {
    if (mode < 0 || mode > 2) return ssdn_set_error("conv mode must be 0, 1 or 2");
    g_conv_mode = mode;
    return 0;
}
// (a launch that must write the fused max-pool output takes k_conv's flat path)
static bool conv_use_dma(const ssdn_conv_args* a) { return g_conv_mode > 0 && !a->pool.p && conv_dma_eligible(a, g_conv_mode == 2); }
static bool conv_use_gemm(const ssdn_conv_args* a) { return g_conv_mode > 0 && gemm_dma_eligible(a); }
static bool conv_use_thin(const ssdn_conv_args* a) { return g_conv_mode > 0 && conv_thin_eligible(a); }
bool conv_gradpack_fusable(const ssdn_grad_pack_args* gp, const ssdn_conv_args* a) {
    return !conv_validate(a) && !conv_use_gemm(a) && !conv_use_thin(a) && !conv_use_dma(a) && gradpack_dgrad_fusable(gp, a);
}
bool conv_pair_fusable(const ssdn_conv_args* a, const ssdn_conv_args* b) {
    return !conv_validate(a) && !conv_validate(b) && conv_use_gemm(a) && gemm_dma_fuses_next(a, b);
}
bool conv_pack_fusable(const ssdn_pack_input_args* pk, const ssdn_conv_args* a) {
    return !conv_validate(a) && !conv_use_gemm(a) && conv_use_thin(a) && conv_thin_fuses_pack(pk, a);
}

int conv_validate(const ssdn_conv_args* a) {
    if (a->ntaps < 1 || a->ntaps > SSDN_MAX_TAPS) return ssdn_set_error("conv: ntaps out of range");
    if (a->ltw + a->lth + a->ltn > 9 || a->ltw < 0 || a->lth < 0 || a->ltn < 0) return ssdn_set_error("conv: tile must have <= 512 pixels");
    if (a->Ktot != a->c0 + a->c1 || (a->Ktot & 15)) return ssdn_set_error("conv: Ktot must equal c0+c1 and be a multiple of 16");
    if ((a->c0 & 7) || (a->c1 & 7)) return ssdn_set_error("conv: source channel counts must be multiples of 8");
    if (a->kc < 16 || (a->kc & 15) || a->Ktot % a->kc || (a->kc > 64 && a->kc != 96)) return ssdn_set_error("conv: kc must be 16, 32, 48, 64 or 96 and divide Ktot");
    if ((a->Mpad & 31) || a->M > a->Mpad) return ssdn_set_error("conv: Mpad must be a multiple of 32 and >= M");
    if (!a->dst32 && ((a->M & 7) || (a->dst.co & 7) || (a->dst.cs & 7))) return ssdn_set_error("conv: 16-bit output needs M, dst.co, dst.cs %% 8 == 0");
    if (a->add.p && ((a->add.co & 7) || (a->add.cs & 7))) return ssdn_set_error("conv: add view must be 16-byte aligned");
    if (a->mask.p && ((a->mask.co & 7) || (a->mask.cs & 7))) return ssdn_set_error("conv: mask view must be 16-byte aligned");
    if (a->up0 && ((a->H | a->W) & 1)) return ssdn_set_error("conv: upsampled source needs even H, W");
    if (a->c1 > 0 && !a->src1.p) return ssdn_set_error("conv: src1 missing");
    if (a->bf16 && a->dst32) return ssdn_set_error("conv: fp32 output is only implemented for the fp16 (forward) role");
    int csmax = a->src0.cs > a->src1.cs ? a->src0.cs : a->src1.cs;
    if ((long long)a->N * a->H * a->W * csmax >= (1ll << 31)) return ssdn_set_error("conv: tensor too large for 32-bit element offsets");
    return 0;
}

static size_t conv_lds(const ssdn_conv_args* a, const ConvGeom& g, int mt) {
    size_t main_b = (conv_async(*a, a->kc) ? 2 * (size_t)g.NP * a->kc * 2 : (size_t)g.NP * g.PSTR) + 2 * (size_t)mt * 32 * g.WSTR;
    if (conv_allw(*a, g, mt, a->kc) && !conv_async(*a, a->kc)) main_b = (size_t)g.NP * g.PSTR + 9 * (size_t)32 * g.WSTR;
    size_t epi_b = a->dst32 ? 0 : (size_t)(g.TN * g.TH * g.TW) * (mt * 64 + 16) + mt * 32 * 4;
    return main_b > epi_b ? main_b : epi_b;
}

int conv_lds_bytes(const ssdn_conv_args* a) {
    if (conv_validate(a)) return -1;
    if (conv_use_gemm(a)) return gemm_dma_lds_bytes(a);
    if (conv_use_thin(a)) return 32 * 1024;                     // (halo of four channel slots + four wave-private transpose tiles: < 32 KB)
    if (conv_use_dma(a)) return conv_dma_lds_bytes(a->Mpad >= 96 ? 3 : a->Mpad / 32);
    ConvGeom g = conv_geom(a->ltw, a->lth, a->ltn, a->ntaps, a->dy, a->dx, a->N, a->H, a->W, a->kc);
    int mt = a->Mpad / 32;
    if (mt > 3) mt = 3;
    if (conv_uses_mt1(a, g)) mt = 1;
    else if (a->kc == 96) { ssdn_set_error("conv: kc = 96 is only built for layers that run as 32-channel blocks (<= 1 tile per CU)"); return -1; }
    return (int)conv_lds(a, g, mt);
}

// FLAT: ALLW + the 256-pixel tile is made of whole images (see k_conv)
static bool conv_flat_ok(const ssdn_conv_args* a, const ConvGeom& g, int ks, int threads) {
    static const bool no_flat = ssdn_tuning_env("SSDN_CONV_NO_FLAT") != nullptr;      // A/B aid, read once
    const int m_last = a->M - (a->Mpad - 32);                                // real channels of the last 32-channel block
    return !no_flat && threads == 256 && ks <= 4 && g.TW == a->W && g.TH == a->H && g.TN * g.TH * g.TW == 256 && a->N % g.TN == 0 &&
           (m_last == 32 || m_last == 16 || m_last == 8) && (!a->up0 || !((a->H | a->W) & 1));
}

template <int MT, bool BF, int KS, int CONV_THREADS>
static int conv_launch_mt(const ssdn_conv_args* a, const ConvGeom& g, ConvAux x, int nblk_y, hipStream_t s) {
    size_t lds = conv_lds(a, g, MT);
    if (lds > 160 * 1024) return ssdn_set_error("conv: tiling needs %zu B of LDS (> 160 KiB)", lds);
    static bool attr_set_dev[SSDN_MAX_DEVICES_ATTR] = {};
    bool& attr_set = attr_set_dev[ssdn_current_device_slot()];   // (function attributes are per device)
    if (!attr_set) {
        SSDN_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv<MT, BF, KS, CONV_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    int grid = g.tiles_x * g.tiles_y * g.groups_n;
    // algorithmic work of THIS launch: real output channels x input channel slots x taps x real pixels
    int m_real = a->M - x.m_base;
    m_real = m_real < 0 ? 0 : (m_real > nblk_y * MT * 32 ? nblk_y * MT * 32 : m_real);
    double px = (double)a->N * a->H * a->W;
    // algorithmic work: REAL input channels (kreal; padded slots do not count), every input read once, every output written once
    const int kreal = a->kreal > 0 ? a->kreal : a->Ktot;
    double flops = 2.0 * px * m_real * kreal * a->ntaps;
    double bytes = px * (a->c0 * 2.0 / (a->up0 ? 4.0 : 1.0) + a->c1 * 2.0) + px * m_real * (a->dst32 ? 4.0 : 2.0);
    prof_begin(3 - MT, s);
    x.nblk = nblk_y;
    static const bool no_allw = ssdn_tuning_env("SSDN_CONV_NO_ALLW") != nullptr;      // A/B aid, read once
    x.allw = (!no_allw && conv_allw(*a, g, MT, a->kc) && !conv_async(*a, a->kc)) ? 1 : 0;
    x.flat = (x.allw && conv_flat_ok(a, g, KS, CONV_THREADS)) ? 1 : 0;
    if (a->pool.p && !x.flat) return ssdn_set_error("conv: fused max-pool requested for a launch that does not take the flat path");
    if (a->upsum.p && !x.flat) return ssdn_set_error("conv: fused upsum requested for a launch that does not take the flat path");
    const int grid_all = nblk_y > 1 ? ((grid + 7) / 8) * 8 * nblk_y : grid;
    SSDN_LAUNCH((k_conv<MT, BF, KS, CONV_THREADS>), dim3(grid_all), dim3(CONV_THREADS), lds, s, *a, x);
    prof_end(3 - MT, s, flops, bytes);
    return 0;
}

template <int MT, bool BF>
static int conv_launch_ks(const ssdn_conv_args* a, const ConvGeom& g, ConvAux x, int nblk_y, hipStream_t s) {
    const bool wide = a->ltw + a->lth + a->ltn > 8;     // 257..512-pixel tiles run with 8 waves
    switch (a->kc) {
        case 16: return wide ? conv_launch_mt<MT, BF, 1, 512>(a, g, x, nblk_y, s) : conv_launch_mt<MT, BF, 1, 256>(a, g, x, nblk_y, s);
        case 32: return wide ? conv_launch_mt<MT, BF, 2, 512>(a, g, x, nblk_y, s) : conv_launch_mt<MT, BF, 2, 256>(a, g, x, nblk_y, s);
        case 48: return wide ? conv_launch_mt<MT, BF, 3, 512>(a, g, x, nblk_y, s) : conv_launch_mt<MT, BF, 3, 256>(a, g, x, nblk_y, s);
        case 64: return wide ? conv_launch_mt<MT, BF, 4, 512>(a, g, x, nblk_y, s) : conv_launch_mt<MT, BF, 4, 256>(a, g, x, nblk_y, s);
    }
    if constexpr (MT == 1) {
        if (a->kc == 96 && !wide) return conv_launch_mt<1, BF, 6, 256>(a, g, x, nblk_y, s);
    }
    return ssdn_set_error("conv: unsupported kc %d", a->kc);
}


bool conv_fuses_pool(const ssdn_conv_args* a) {
    static const bool no_allw = ssdn_tuning_env("SSDN_CONV_NO_ALLW") != nullptr;
    if (conv_validate(a) || a->bf16 || a->dst32 || !a->act || (a->H & 1) || (a->W & 1)) return false;
    if (conv_use_gemm(a)) return false;
    ConvGeom g = conv_geom(a->ltw, a->lth, a->ltn, a->ntaps, a->dy, a->dx, a->N, a->H, a->W, a->kc);
    const bool wide = a->ltw + a->lth + a->ltn > 8;
    return conv_uses_mt1(a, g) && !no_allw && conv_allw(*a, g, 1, a->kc) && !conv_async(*a, a->kc) &&
           conv_flat_ok(a, g, a->kc / 16, wide ? 512 : 256);
}

static bool conv_flat_path(const ssdn_conv_args* a) {
    static const bool no_allw = ssdn_tuning_env("SSDN_CONV_NO_ALLW") != nullptr;
    ConvGeom g = conv_geom(a->ltw, a->lth, a->ltn, a->ntaps, a->dy, a->dx, a->N, a->H, a->W, a->kc);
    const bool wide = a->ltw + a->lth + a->ltn > 8;
    return conv_uses_mt1(a, g) && !no_allw && conv_allw(*a, g, 1, a->kc) && !conv_async(*a, a->kc) &&
           conv_flat_ok(a, g, a->kc / 16, wide ? 512 : 256);
}
bool conv_signs(const ssdn_conv_args* a) {
    if (!(a->sign_out || a->mask_sign || a->upsum_mask_sign) || conv_validate(a)) return false;
    if (conv_use_gemm(a)) return !a->upsum_mask_sign && gemm_dma_signs(a);
    if (conv_use_thin(a)) return a->sign_out && !a->mask_sign && !a->upsum_mask_sign;       // k_conv_thin writes them (forward only)
    return conv_use_dma(a) && conv_dma_signs(a);
}
bool conv_fuses_urot(const ssdn_conv_args* a) {
    return a->urot.p && !conv_validate(a) && !conv_use_gemm(a) && !conv_use_thin(a) && conv_use_dma(a);   // (conv_dma_eligible checks the shape)
}
bool conv_fuses_unrot(const ssdn_conv_args* a) {
    // (the query is about the launch WITH the fused output requested)
    ssdn_conv_args q = *a;
    if (!q.unrot.p) { q.unrot = q.dst; q.unrot_mask = q.dst; }
    return !conv_validate(&q) && conv_use_gemm(&q);
}
bool conv_fuses_upsum(const ssdn_conv_args* a) {
    if (conv_validate(a) || !a->bf16 || a->dst32 || a->mask.p || a->add.p || (a->H & 1) || (a->W & 1)) return false;
    if ((a->upsum_c & 7) || a->upsum_c > a->M || a->upsum_c <= 0) return false;
    if (conv_use_gemm(a)) return false;
    if (conv_use_dma(a)) return a->upsum_c % 96 == 0;
    return conv_flat_path(a);
}

int launch_conv(const ssdn_conv_args* a, hipStream_t s) {
    int rc = conv_validate(a);
    if (rc) return rc;
    if ((a->sign_out || a->mask_sign || a->upsum_mask_sign) && !conv_signs(a)) return ssdn_set_error("conv: sign bytes requested for a launch that cannot write / read them (ssdn_conv_signs)");
    if (a->urot.p && !conv_fuses_urot(a)) return ssdn_set_error("conv: fused UNROT_FWD requested for a launch that cannot fuse it (ssdn_conv_fuses_urot)");
    if (a->unrot.p && !conv_fuses_unrot(a)) return ssdn_set_error("conv: fused UNROT_BWD requested for a launch that cannot fuse it (ssdn_conv_fuses_unrot)");
    if (a->upsum.p && !conv_fuses_upsum(a)) return ssdn_set_error("conv: fused upsum requested for a launch that cannot fuse it (ssdn_conv_fuses_upsum)");
    if (a->pool.p && !conv_fuses_pool(a)) return ssdn_set_error("conv: fused max-pool requested for a launch that cannot fuse it (ssdn_conv_fuses_pool)");
    if (conv_use_gemm(a)) return launch_gemm_dma(a, s);
    if (conv_use_thin(a)) return launch_conv_thin(a, nullptr, s);
    if (conv_use_dma(a)) return launch_conv_dma(a, s);
    ConvGeom g = conv_geom(a->ltw, a->lth, a->ltn, a->ntaps, a->dy, a->dx, a->N, a->H, a->W, a->kc);
    ConvAux x;
    x.mg_hw = magic_of(g.HW);
    x.mg_hh = magic_of(g.HH);
    x.mg_ntaps = magic_of(a->ntaps);
    x.m_base = 0;
    {
        // tuning aids, read ONCE per process (not on the launch path)
        static const int env_ablate = [] { const char* e = ssdn_tuning_env("SSDN_CONV_ABLATE"); return e ? atoi(e) : 0; }();
        static const int env_desync = [] { const char* e = ssdn_tuning_env("SSDN_CONV_DESYNC"); return e ? atoi(e) : 0; }();
        x.ablate = env_ablate;
        x.trace = g_conv_trace;
        x.desync = env_desync;
    }
    // output channels in blocks of 96 (MT=3); the tail uses MT = 1 or 2.  Layers with few pixel tiles are latency-bound (one
    // workgroup's pass over its tile IS the launch): they run as blocks of 32 channels (MT=1) on three times as many
    // workgroups, each with a third of the MFMA and epilogue work.
    int full = a->Mpad / 96, rem = (a->Mpad % 96) / 32;
    const bool bf = a->bf16 != 0;
    if (conv_uses_mt1(a, g)) {
        rc = bf ? conv_launch_ks<1, true>(a, g, x, a->Mpad / 32, s) : conv_launch_ks<1, false>(a, g, x, a->Mpad / 32, s);
        if (rc) return rc;
        SSDN_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (full) {
        rc = bf ? conv_launch_ks<3, true>(a, g, x, full, s) : conv_launch_ks<3, false>(a, g, x, full, s);
        if (rc) return rc;
    }
    x.m_base = full * 96;
    if (rem == 2) rc = bf ? conv_launch_ks<2, true>(a, g, x, 1, s) : conv_launch_ks<2, false>(a, g, x, 1, s);
    else if (rem == 1) rc = bf ? conv_launch_ks<1, true>(a, g, x, 1, s) : conv_launch_ks<1, false>(a, g, x, 1, s);
    if (rc) return rc;
    SSDN_CHECK_HIP(hipGetLastError());
    return 0;
}

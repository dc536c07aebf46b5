// wgrad_mfma.hip -- convolution weight/bias gradient on the CDNA4 matrix cores.
//
// dW[t][m][k] = sum_pixels dZ[pixel][m] * IN[pixel + off_t][k]   is a GEMM whose REDUCTION dimension is the pixel axis,
// while both operands live in HBM/LDS as NHWC (channels contiguous, pixels strided).  The MFMA wants 8 consecutive
// reduction elements per lane, i.e. 8 pixels of ONE channel: exactly the transposed access gfx950's
// ds_read_b64_tr_b16 provides -- the tiles are staged in LDS untransposed (coalesced 16-B NHWC loads, same loader as
// the forward kernel) and both MFMA operands are fetched with the LDS transpose-read.  No transposed copy of any
// activation or gradient is ever written to HBM.
//
// Work split: workgroup = 256 threads = 4 waves (one per SIMD), persistent over its share of (<=256-pixel) tiles that are
// double-buffered in LDS.  For one launch the output is [ntaps][Mpad<=96][Kpad<=96] (+ a bias column): ntaps*Kpad/32 (+1)
// column tiles of 32, dealt round-robin to the 4 waves (<= CPW <= 7 per wave), each wave keeping MT x CPW 32x32 fp32
// accumulators in registers for the whole pixel range and reading the dZ operand (A) once per K-step for all its column
// tiles.  At the end every workgroup writes its accumulators to its own fp32 slab (plain coalesced stores, no atomics) and
// SSDN_OP_WREDUCE sums the slabs in a fixed order => bit-reproducible gradients.
#include "wgrad_body.h"

template <int MT, int CPW, int NL, bool BOTH, int PS, int KS, int RWX, int RWD>
__global__ __launch_bounds__(WG_THREADS) void k_wgrad(ssdn_wgrad_args a, WgAux x) {
    wgrad_body<MT, CPW, NL, BOTH, PS, KS, RWX, RWD>(a, x, blockIdx.x, blockIdx.y, gridDim.x);
}

// ---- several layers' weight-gradient GEMMs in ONE launch ------------------------------------------------------------------
// The layers at the bottom of the U (16x16 pixels and below) have a few hundred tiles each: one launch per layer is a chain of
// K-steps a handful of tiles long plus the slab write, on a chip it cannot fill -- 13 such launches are ~230 us of the
// weight-gradient lane.  k_wgrad_multi runs the workgroups of a RUN of consecutive SSDN_OP_WGRAD ops (ssdn_run_ops merges
// them when every op is in the instance set below) side by side: entry e owns blocks [first, first + gx*gy) of the grid and
// executes exactly the code of its own launch (same template instance, same (bx, by, gdx)), so results are bit-identical.
struct WgMultiEntry {
    ssdn_wgrad_args a;
    WgAux x;
    int first, gx, gy, inst;
};
#define WG_MULTI_INSTANCES(X) \
    X(0, 2, 1, 4, false) X(1, 2, 2, 4, false) X(2, 2, 3, 4, false) X(3, 2, 5, 4, false) \
    X(4, 3, 1, 4, false) X(5, 3, 2, 4, false) X(6, 3, 3, 4, false) X(7, 3, 4, 4, false) X(8, 3, 5, 4, false) \
    X(9, 3, 1, 6, true) X(10, 2, 1, 6, true) X(11, 3, 2, 6, true) X(12, 3, 4, 6, true) X(13, 2, 3, 6, true) \
    X(16, 2, 5, 6, true) X(17, 3, 5, 6, true) X(18, 3, 3, 6, true) X(19, 2, 2, 6, true)
__global__ __launch_bounds__(WG_THREADS) void k_wgrad_multi(const WgMultiEntry* __restrict__ tab, int n) {
    int e = 0;
    while (e + 1 < n && (int)blockIdx.x >= tab[e + 1].first) ++e;
    const WgMultiEntry& E = tab[e];
    const unsigned local = blockIdx.x - (unsigned)E.first;
    const unsigned gx = (unsigned)E.gx;
    const unsigned by = local / gx, bx = local - by * gx;
    switch (E.inst) {
#define WG_X(id, mt, cpw, nl, both) case id: wgrad_body<mt, cpw, nl, both, 0, 0, 0, 0>(E.a, E.x, bx, by, gx); break;
        WG_MULTI_INSTANCES(WG_X)
#undef WG_X
        default: break;
    }
}

int wgrad_lds_bytes(const ssdn_wgrad_args* a) {
    if (wgrad_validate(a)) return -1;
    WgGeom g = wg_geom(*a);
    if (wgrad_items(a, g).nl > 6) { ssdn_set_error("wgrad: the tile cannot be prefetched within its K-steps (too many rows / too wide rows)"); return -1; }
    return 2 * (g.XB + g.DB) + WG_ONES_BYTES;
}

template <int MT, int CPW, int NL, bool BOTH, int PS, int KS, int RWX, int RWD>
static int wgrad_launch(const ssdn_wgrad_args* a, const WgGeom& g, const WgAux& x, hipStream_t s) {
    size_t lds = 2 * ((size_t)g.XB + (size_t)g.DB) + WG_ONES_BYTES;
    if (lds > 160 * 1024) return ssdn_set_error("wgrad: tiling needs %zu B of LDS (> 160 KiB)", lds);
    static bool attr_set_dev[SSDN_MAX_DEVICES_ATTR] = {};
    bool& attr_set = attr_set_dev[ssdn_current_device_slot()];   // (function attributes are per device)
    if (!attr_set) {
        SSDN_CHECK_HIP(hipFuncSetAttribute((const void*)k_wgrad<MT, CPW, NL, BOTH, PS, KS, RWX, RWD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    double px = (double)a->N * a->H * a->W;
    prof_begin(SSDN_PROF_WGRAD, s);
    const int gy = a->csplit > 1 ? a->csplit : 1;
    const int gx = a->mblocks > 1 ? ((a->nslabs + 7) / 8) * 8 * a->mblocks : a->nslabs;
    hipLaunchKernelGGL((k_wgrad<MT, CPW, NL, BOTH, PS, KS, RWX, RWD>), dim3(gx, gy), dim3(WG_THREADS), lds, s, *a, x);
    prof_end(SSDN_PROF_WGRAD, s, wgrad_flops(a), px * 2.0 * (a->M * (a->mblocks > 1 ? a->mblocks : 1) + a->Ktot));
    return 0;
}

// instance of k_wgrad_multi that runs this op, or -1 (the op needs its own launch)
static int wgrad_multi_inst(const WgPrep& p) {
#define WG_X(id_, mt_, cpw_, nl_, both_) if (p.MT == mt_ && p.CPW == cpw_ && (p.wi.both != 0) == both_) return id_;
    WG_MULTI_INSTANCES(WG_X)
#undef WG_X
    return -1;
}
bool wgrad_mergeable(const ssdn_wgrad_args* a) {
    if (wgrad_thin_ok(a)) return false;                             // (has its own kernel)
    // layers of at most 128 images x 16 x 16 pixels: their own launch cannot fill the chip
    static const long long small_px = [] { const char* e = ssdn_tuning_env("SSDN_WGRAD_SMALL_PX"); return e ? atoll(e) : 32768ll; }();   // experiment knob, read once
    if ((long long)a->N * a->H * a->W > small_px || a->mblocks > 1) return false;
    WgPrep p;
    if (wgrad_validate(a)) return false;
    p.g = wg_geom(*a);
    p.wi = wgrad_items(a, p.g);
    if (p.wi.nl > 6) return false;
    p.MT = a->Mpad / 32;
    const int CT = a->ntaps * (a->Kpad / 32) + 1;
    p.gy = a->csplit > 1 ? a->csplit : 1;
    p.CPW = (CT + WG_WAVES * p.gy - 1) / (WG_WAVES * p.gy);
    return wgrad_multi_inst(p) >= 0;
}

namespace {
struct WgMultiCache { std::vector<WgMultiEntry> host; WgMultiEntry* dev; int device; };
std::vector<WgMultiCache> g_wg_multi_cache;      // device copies of the tables of the op-list runs seen so far (a handful per
std::mutex g_wg_multi_mutex;                      // process: the op lists of an engine are static)
}

int launch_wgrad_multi(const ssdn_wgrad_args* const* items, int n, hipStream_t s) {
    if (n < 1 || n > WGRAD_MULTI_MAX) return ssdn_set_error("wgrad: bad batch size %d", n);
    std::vector<WgMultiEntry> tab((size_t)n);
    memset(tab.data(), 0, sizeof(WgMultiEntry) * (size_t)n);
    size_t lds = 0;
    int blocks = 0;
    double flops = 0, bytes = 0;
    // heaviest workgroups first: blocks are dispatched in index order, so the short ones fill in behind the long ones
    std::vector<std::pair<double, int>> order;
    std::vector<WgPrep> preps((size_t)n);
    for (int i = 0; i < n; ++i) {
        int rc = wgrad_prepare(items[i], &preps[i]);
        if (rc) return rc;
        const WgPrep& p = preps[i];
        const double tiles_per_wg = (double)p.g.ntiles / (items[i]->nslabs > 0 ? items[i]->nslabs : 1);
        const double cost = tiles_per_wg * (p.g.TN * p.g.TH * p.g.TW / 16) * (660.0 + 90.0 * p.CPW) +
                            (double)items[i]->ntaps * items[i]->Mpad * items[i]->Kpad * 4.0 / 10.0 / p.gy;
        order.push_back({-cost, i});
    }
    std::sort(order.begin(), order.end());
    for (int k = 0; k < n; ++k) {
        const int i = order[k].second;
        const WgPrep& p = preps[i];
        WgMultiEntry& e = tab[k];
        memcpy(&e.a, items[i], sizeof(ssdn_wgrad_args));
        e.x = p.x;
        e.inst = wgrad_multi_inst(p);
        if (e.inst < 0) return ssdn_set_error("wgrad: op %d of a merged run has no k_wgrad_multi instance (MT=%d CPW=%d both=%d)", i, p.MT, p.CPW, p.wi.both);
        e.first = blocks; e.gx = p.gx; e.gy = p.gy;
        blocks += p.gx * p.gy;
        lds = p.lds > lds ? p.lds : lds;
        const double px = (double)items[i]->N * items[i]->H * items[i]->W;
        flops += wgrad_flops(items[i]);
        bytes += px * 2.0 * (items[i]->M + items[i]->Ktot);
    }
    int dev = 0;
    SSDN_CHECK_HIP(hipGetDevice(&dev));
    WgMultiEntry* dtab = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_wg_multi_mutex);
        for (const WgMultiCache& c : g_wg_multi_cache)
            if (c.device == dev && c.host.size() == tab.size() && !memcmp(c.host.data(), tab.data(), sizeof(WgMultiEntry) * tab.size())) { dtab = c.dev; break; }
        if (!dtab) {
            if (g_wg_multi_cache.size() >= 64) {          // bounded: drop the oldest table (no launch that uses it can still be
                SSDN_CHECK_HIP(hipDeviceSynchronize());   // queued after the synchronisation)
                SSDN_CHECK_HIP(hipFree(g_wg_multi_cache.front().dev));
                g_wg_multi_cache.erase(g_wg_multi_cache.begin());
            }
            SSDN_CHECK_HIP(hipMalloc((void**)&dtab, sizeof(WgMultiEntry) * tab.size()));
            SSDN_CHECK_HIP(hipMemcpy(dtab, tab.data(), sizeof(WgMultiEntry) * tab.size(), hipMemcpyHostToDevice));
            g_wg_multi_cache.push_back({tab, dtab, dev});
        }
    }
    static bool attr_set_dev[SSDN_MAX_DEVICES_ATTR] = {};
    bool& attr_set = attr_set_dev[ssdn_current_device_slot()];   // (function attributes are per device)
    if (!attr_set) {
        SSDN_CHECK_HIP(hipFuncSetAttribute((const void*)k_wgrad_multi, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    prof_begin(SSDN_PROF_WGRAD, s);
    hipLaunchKernelGGL(k_wgrad_multi, dim3(blocks), dim3(WG_THREADS), lds, s, (const WgMultiEntry*)dtab, n);
    prof_end(SSDN_PROF_WGRAD, s, flops, bytes);
    SSDN_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int MT>
__global__ __launch_bounds__(WGN_THREADS) void k_wgrad_thin(ssdn_wgrad_args a) {
    wgrad_thin_body<MT>(a, blockIdx.x, gridDim.x);
}
template <int MT>
static int wgrad_thin_launch(const ssdn_wgrad_args* a, hipStream_t s) {
    const size_t lds = 2 * ((size_t)256 * wg_stride(MT * 64) + 18 * 18 * 16);
    static bool attr_set_dev[SSDN_MAX_DEVICES_ATTR] = {};
    bool& attr_set = attr_set_dev[ssdn_current_device_slot()];   // (function attributes are per device)
    if (!attr_set) {
        SSDN_CHECK_HIP(hipFuncSetAttribute((const void*)k_wgrad_thin<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const double px = (double)a->N * a->H * a->W;
    prof_begin(SSDN_PROF_WGRAD, s);
    hipLaunchKernelGGL(k_wgrad_thin<MT>, dim3(a->nslabs), dim3(WGN_THREADS), lds, s, *a);
    prof_end(SSDN_PROF_WGRAD, s, 2.0 * px * a->M * a->kreal * 9, px * 2.0 * (a->M + a->kreal));
    SSDN_CHECK_HIP(hipGetLastError());
    return 0;
}
static int launch_wgrad_thin(const ssdn_wgrad_args* a, hipStream_t s) {
    switch (a->Mpad / 32) {
        case 1: return wgrad_thin_launch<1>(a, s);
        case 2: return wgrad_thin_launch<2>(a, s);
        default: return wgrad_thin_launch<3>(a, s);
    }
}

int launch_wgrad(const ssdn_wgrad_args* a, hipStream_t s) {
    WgPrep prep;
    int rc = wgrad_prepare(a, &prep);
    if (rc) return rc;
    if (wgrad_thin_ok(a)) return launch_wgrad_thin(a, s);
    const WgGeom& g = prep.g;
    const WgAux& x = prep.x;
    const WgItems& wi = prep.wi;
    const int MT = a->Mpad / 32;
    const int CT = a->ntaps * (a->Kpad / 32) + 1;
    const int gy = a->csplit > 1 ? a->csplit : 1;
    const int CPW = (CT + WG_WAVES * gy - 1) / (WG_WAVES * gy);   // column tiles per wave (gy = column groups)
    // the hot shapes get the input pixel stride and the whole staging schedule as compile-time constants:
    //   3x3, 48..96 input channels: stride 192 B, 16x8 tiles  -> 8 K-steps, 3 input + 2 dZ rows per wave
    //   3x3, 16..32 input channels: stride  64 B, 16x16 tiles -> 16 K-steps, 5 + 4 rows per wave
    const int ksteps = (g.TN * g.TH * g.TW) >> 4;
    const bool st = !wi.both && !wi.sync && a->ltn == 0 && a->ltw >= 3 && g.ntiles > a->nslabs;
    const bool st8 = st && g.PSTR == 192 && ksteps == 8 && wi.rswx == 3 && wi.rswd == 2;
    const bool st16 = st && g.PSTR == 64 && ksteps == 16 && wi.rswx == 5 && wi.rswd == 4;
    // 1x1 layers over four 96-channel input blocks: stride 832 B, 8x8 tiles -> 4 K-steps, an input row AND a dZ row per K-step
    const bool st4b = wi.both && !wi.sync && a->ltn == 0 && a->ltw >= 3 && g.ntiles > a->nslabs && g.PSTR == 832 && ksteps == 4 && wi.rswx == 2 && wi.rswd == 2;
    // (a static variant is only instantiated for the (mt, cpw) it is used with: elsewhere its template arguments collapse to
    //  the generic kernel's)
    if (st8 && wgrad_nl2(a, g, MT, CPW)) {
        rc = MT == 2 ? wgrad_launch<2, 5, 2, false, 192, 8, 3, 2>(a, g, x, s) : wgrad_launch<3, 5, 2, false, 192, 8, 3, 2>(a, g, x, s);
        if (rc) return rc;
        SSDN_CHECK_HIP(hipGetLastError());
        return 0;
    }
#define WG_S8(mt, cpw) ((mt) >= 2 && (cpw) >= 2)
#define WG_S16(mt, cpw) ((mt) >= 2 && (cpw) <= 3)
#define WG_S4B(mt, cpw) ((mt) == 3 && (cpw) == 4)
#define WG_CASE(mt, cpw) if (MT == mt && CPW == cpw) { \
        if (st4b && WG_S4B(mt, cpw)) rc = wgrad_launch<mt, cpw, 6, true, WG_S4B(mt, cpw) ? 832 : 0, WG_S4B(mt, cpw) ? 4 : 0, WG_S4B(mt, cpw) ? 2 : 0, WG_S4B(mt, cpw) ? 2 : 0>(a, g, x, s); \
        else if (wi.both) rc = wgrad_launch<mt, cpw, 6, true, 0, 0, 0, 0>(a, g, x, s); \
        else if (st8 && WG_S8(mt, cpw)) rc = wgrad_launch<mt, cpw, 4, false, WG_S8(mt, cpw) ? 192 : 0, WG_S8(mt, cpw) ? 8 : 0, WG_S8(mt, cpw) ? 3 : 0, WG_S8(mt, cpw) ? 2 : 0>(a, g, x, s); \
        else if (st16 && WG_S16(mt, cpw)) rc = wgrad_launch<mt, cpw, 4, false, WG_S16(mt, cpw) ? 64 : 0, WG_S16(mt, cpw) ? 16 : 0, WG_S16(mt, cpw) ? 5 : 0, WG_S16(mt, cpw) ? 4 : 0>(a, g, x, s); \
        else rc = wgrad_launch<mt, cpw, 4, false, 0, 0, 0, 0>(a, g, x, s); \
    } else
    WG_CASE(1, 1) WG_CASE(1, 2) WG_CASE(1, 3) WG_CASE(1, 4) WG_CASE(1, 5) WG_CASE(1, 6) WG_CASE(1, 7)
    WG_CASE(2, 1) WG_CASE(2, 2) WG_CASE(2, 3) WG_CASE(2, 4) WG_CASE(2, 5) WG_CASE(2, 6) WG_CASE(2, 7)
    WG_CASE(3, 1) WG_CASE(3, 2) WG_CASE(3, 3) WG_CASE(3, 4) WG_CASE(3, 5) WG_CASE(3, 6) WG_CASE(3, 7)
    rc = ssdn_set_error("wgrad: unsupported shape MT=%d CPW=%d", MT, CPW);
#undef WG_CASE
#undef WG_S8
#undef WG_S16
#undef WG_S4B
    if (rc) return rc;
    SSDN_CHECK_HIP(hipGetLastError());
    return 0;
}

// wgrad_mega.hip -- the weight gradients of a whole gradient bucket (or of the whole network) as ONE chip-wide launch.
#include "wgrad_body.h"

namespace { std::mutex g_wg_multi_mutex; }


// ---- the weight gradients of a whole gradient bucket -- or of the whole network -- in ONE launch (round 4) ----------------------------
// 25 launches per step, each a persistent grid that fills the chip for 20-110 us, pay 25 prologues (first tile fetched
// synchronously), 25 slab writes of up to 332 KB per workgroup and 25 ragged tails; the layers at the bottom of the U cannot fill the
// chip at all.  A weight-gradient workgroup owns its CU anyway (512 registers per lane on every SIMD), so the natural unit is the CU:
// k_wgrad_mega is ONE grid with a block per ITEM; an item is one block (bx, by) of one op's own launch grid and runs exactly the code
// that launch would run (same template instance, same (bx, by, gdx): bit-identical slabs).  The planner (ssdn/hip/graph.py) sizes every op's grid from its cost model so that the items of one launch add up to equal
// times -- a layer with 27 % of the work gets 27 % of the CUs for the whole launch instead of all CUs for 27 % of the time, which
// divides the number of slabs (written, then read back by SSDN_OP_WREDUCE) by the number of layers -- and gives each op its estimated
// cost per block (ssdn_wgrad_args.cost); the items are sorted by it, longest first (stable: deterministic tables), and the hardware
// hands them to the CUs in that order.
struct WgMegaEntry {
    ssdn_wgrad_args a;
    WgAux x;
    int inst, gx, gy, pad;
};
struct WgMegaItem { int entry, bx, by, pad; };
// id, MT, CPW, NL, BOTH, PS, KS, RWX, RWD -- the instances launch_wgrad would pick for the ops of the BASELINE configurations when
// every op is planned as ONE column group; thin layers: ids 100 + MT
#define WG_MEGA_INSTANCES(X) \
    X(0, 3, 7, 4, false, 192, 8, 3, 2) X(1, 3, 5, 4, false, 192, 8, 3, 2) X(2, 2, 5, 4, false, 192, 8, 3, 2) \
    X(3, 3, 4, 6, true, 832, 4, 2, 2) X(4, 3, 4, 4, false, 0, 0, 0, 0) X(5, 3, 5, 4, false, 0, 0, 0, 0) \
    X(6, 2, 5, 4, false, 0, 0, 0, 0) X(7, 1, 1, 4, false, 0, 0, 0, 0) X(8, 3, 1, 4, false, 0, 0, 0, 0) \
    X(10, 3, 5, 6, true, 0, 0, 0, 0) X(11, 2, 5, 6, true, 0, 0, 0, 0) \
    X(12, 3, 4, 6, true, 0, 0, 0, 0) X(13, 1, 1, 6, true, 0, 0, 0, 0) X(14, 3, 1, 6, true, 0, 0, 0, 0) \
    X(15, 3, 5, 2, false, 192, 8, 3, 2) X(16, 2, 5, 2, false, 192, 8, 3, 2)
// One block per ITEM, items in descending order of cost: the hardware hands the next block to the next CU that frees up (a
// weight-gradient workgroup owns its CU), i.e. it performs the longest-first packing itself, with the true run times.
// The tables are read through the CONSTANT address space, like kernel arguments: the entry's fields then are invariant scalar loads
// the register allocator may re-load instead of spilling.  (Tried first: a loop over a per-workgroup item list calling one function
// per instance -- the callee-saved registers of a 512-register function need ~1.9 KB of scratch per lane, and a dispatch with that
// much scratch is throttled by the runtime to a fraction of the CUs: 2.2 ms for 0.45 ms of work.)
typedef const __attribute__((address_space(4))) WgMegaEntry* WgMegaEntryC;
typedef const __attribute__((address_space(4))) WgMegaItem* WgMegaItemC;
// The register allocation of this kernel sits on an edge (the 21-accumulator bodies use 480+ of the 512 registers): one more kernel
// argument once flipped it from 0 to 764 spilled VGPRs.  `make` therefore checks the code object (check_scratch.py): a build whose
// weight-gradient kernels need scratch FAILS.  Block timeline for tools/wgrad_calib.py (ssdn_debug_set_trace): the buffer's address
// travels in the entry table (WgAux.trace) and is loaded at the end of the block, where both stamps are written; s_memrealtime is
// the chip-wide 100 MHz clock (s_memtime counts per XCD).
__global__ __launch_bounds__(WG_THREADS) void k_wgrad_mega(const WgMegaEntry* __restrict__ ent, const WgMegaItem* __restrict__ items) {
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();     // (both stamps are STORED at the end: a store in front of the
                                                                              //  body would cost its table loads their invariance)
    const WgMegaItemC it = (WgMegaItemC)(unsigned long long)(items + blockIdx.x);
    const WgMegaEntryC E4 = (WgMegaEntryC)(unsigned long long)(ent + it->entry);
    const unsigned bx = (unsigned)it->bx, by = (unsigned)it->by;
    const WgMegaEntry& E = *(const WgMegaEntry*)E4;
    const unsigned gx = (unsigned)E.gx;
    switch (E.inst) {
#define WG_X(id, mt, cpw, nl, both, ps, ks, rwx, rwd) case id: wgrad_body<mt, cpw, nl, both, ps, ks, rwx, rwd>(E.a, E.x, bx, by, gx); break;
        WG_MEGA_INSTANCES(WG_X)
#undef WG_X
        case 101: wgrad_thin_body<1>(E.a, bx, gx); break;
        case 102: wgrad_thin_body<2>(E.a, bx, gx); break;
        case 103: wgrad_thin_body<3>(E.a, bx, gx); break;
        default: break;
    }
    {
        unsigned long long* tr = E.x.trace;
        if (tr && threadIdx.x == 0) {      // {start, end, entry | bx << 16} per block
            tr[3 * blockIdx.x] = t_start;
            tr[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
            tr[3 * blockIdx.x + 2] = (unsigned long long)it->entry | ((unsigned long long)bx << 16);
        }
    }
}

// the template instance launch_wgrad runs an op with (the WG_CASE dispatch below, as data)
struct WgVariant { int thin, mt, cpw, nl, both, ps, ks, rwx, rwd; };
static WgVariant wgrad_variant(const ssdn_wgrad_args* a, const WgPrep& p) {
    WgVariant v = {};
    v.mt = p.MT; v.cpw = p.CPW;
    if (wgrad_thin_ok(a)) { v.thin = 1; return v; }
    const WgGeom& g = p.g;
    const WgItems& wi = p.wi;
    const int ksteps = (g.TN * g.TH * g.TW) >> 4;
    const bool st = !wi.both && !wi.sync && a->ltn == 0 && a->ltw >= 3 && g.ntiles > a->nslabs;
    const bool st8 = st && g.PSTR == 192 && ksteps == 8 && wi.rswx == 3 && wi.rswd == 2;
    const bool st16 = st && g.PSTR == 64 && ksteps == 16 && wi.rswx == 5 && wi.rswd == 4;
    const bool st4b = wi.both && !wi.sync && a->ltn == 0 && a->ltw >= 3 && g.ntiles > a->nslabs && g.PSTR == 832 && ksteps == 4 && wi.rswx == 2 && wi.rswd == 2;
    v.nl = wi.both ? 6 : 4; v.both = wi.both ? 1 : 0;
    if (st4b && v.mt == 3 && v.cpw == 4) { v.ps = 832; v.ks = 4; v.rwx = 2; v.rwd = 2; }
    else if (wi.both) {}
    else if (st8 && v.mt >= 2 && v.cpw >= 2) { v.ps = 192; v.ks = 8; v.rwx = 3; v.rwd = 2; if (wgrad_nl2(a, g, v.mt, v.cpw)) v.nl = 2; }
    else if (st16 && v.mt >= 2 && v.cpw <= 3) { v.ps = 64; v.ks = 16; v.rwx = 5; v.rwd = 4; }
    return v;
}
static int wgrad_mega_inst(const WgVariant& v) {
    if (v.thin) return v.mt >= 1 && v.mt <= 3 ? 100 + v.mt : -1;
#define WG_X(id, mt_, cpw_, nl_, both_, ps_, ks_, rwx_, rwd_) \
    if (v.mt == mt_ && v.cpw == cpw_ && v.nl == nl_ && (v.both != 0) == both_ && v.ps == ps_ && v.ks == ks_ && v.rwx == rwx_ && v.rwd == rwd_) return id;
    WG_MEGA_INSTANCES(WG_X)
#undef WG_X
    return -1;
}
static int wgrad_mega_prepare(const ssdn_wgrad_args* a, WgPrep* p) {
    int rc = wgrad_prepare(a, p);
    if (rc) return rc;
    if (wgrad_thin_ok(a)) {
        p->gx = a->nslabs; p->gy = 1;
        p->lds = 2 * ((size_t)256 * wg_stride(p->MT * 64) + 18 * 18 * 16);
        const size_t red = (size_t)4 * p->MT * 16 * 64 * 4;
        p->lds = p->lds > red ? p->lds : red;
    }
    return 0;
}
// 1: the op can be an entry of a k_wgrad_mega launch
int wgrad_mega_ok(const ssdn_wgrad_args* a) {
    if (!a || a->mega <= 0) return 0;
    WgPrep p;
    if (wgrad_mega_prepare(a, &p)) return 0;
    return wgrad_mega_inst(wgrad_variant(a, p)) >= 0 ? 1 : 0;
}
extern "C" int ssdn_wgrad_variant(const ssdn_wgrad_args* a, int32_t* out9) {
    WgPrep p;
    int rc = wgrad_mega_prepare(a, &p);
    if (rc) return rc < -1 ? rc : -2;
    const WgVariant v = wgrad_variant(a, p);
    const int t[9] = {v.thin, v.mt, v.cpw, v.nl, v.both, v.ps, v.ks, v.rwx, v.rwd};
    for (int i = 0; i < 9; ++i) out9[i] = t[i];
    return wgrad_mega_inst(v);
}

namespace {
struct WgMegaCache {
    std::vector<char> key;      // entries + items + starts, byte for byte
    char* dev;
    int device;
};
std::vector<WgMegaCache> g_wg_mega_cache;
}

int launch_wgrad_mega(const ssdn_wgrad_args* const* ops, int n, hipStream_t s) {
    if (n < 1 || n > WGRAD_MEGA_MAX) return ssdn_set_error("wgrad: bad mega run length %d", n);
    const int W = ops[0]->mega;
    if (W < 1 || W > 4096) return ssdn_set_error("wgrad: bad mega grid %d", W);
    std::vector<WgMegaEntry> ent((size_t)n);
    memset(ent.data(), 0, sizeof(WgMegaEntry) * (size_t)n);
    struct It { double cost; int e, bx, by; };
    std::vector<It> its;
    size_t lds = 0;
    double flops = 0, bytes = 0;
    for (int i = 0; i < n; ++i) {
        if (ops[i]->mega != W) return ssdn_set_error("wgrad: the ops of one merged launch disagree on its grid (%d vs %d)", ops[i]->mega, W);
        WgPrep p;
        int rc = wgrad_mega_prepare(ops[i], &p);
        if (rc) return rc;
        WgMegaEntry& e = ent[i];
        memcpy(&e.a, ops[i], sizeof(ssdn_wgrad_args));
        e.x = p.x;
        e.inst = wgrad_mega_inst(wgrad_variant(ops[i], p));
        if (e.inst < 0) return ssdn_set_error("wgrad: op %d of a merged run has no k_wgrad_mega instance (MT=%d CPW=%d both=%d)", i, p.MT, p.CPW, p.wi.both);
        e.gx = p.gx; e.gy = p.gy;
        lds = p.lds > lds ? p.lds : lds;
        const double px = (double)ops[i]->N * ops[i]->H * ops[i]->W;
        const int mb = ops[i]->mblocks > 1 ? ops[i]->mblocks : 1;
        const bool thin = e.inst >= 100;
        flops += thin ? 2.0 * px * ops[i]->M * ops[i]->kreal * ops[i]->ntaps : wgrad_flops(ops[i]);
        bytes += px * 2.0 * (ops[i]->M * mb + (thin ? ops[i]->kreal : ops[i]->Ktot));
        const double c = ops[i]->cost > 0.f ? (double)ops[i]->cost : 1.0;
        for (int by = 0; by < p.gy; ++by)
            for (int bx = 0; bx < p.gx; ++bx) {
                if (mb > 1 && (((bx >> 3) / mb) * 8 + (bx & 7)) >= ops[i]->nslabs) continue;       // (grid padding of the mblocks launch: no work)
                its.push_back({c, i, bx, by});
            }
    }
    // longest item first: blocks are dispatched in index order as CUs free up (stable: ties keep list order -- deterministic tables)
    std::stable_sort(its.begin(), its.end(), [](const It& a, const It& b) { return a.cost > b.cost; });
    std::vector<WgMegaItem> items;
    for (const It& t : its) items.push_back({t.e, t.bx, t.by, 0});
    std::vector<int> starts(1, (int)items.size());
    // device copy of the three tables, cached per distinct run
    const size_t b0 = sizeof(WgMegaEntry) * ent.size(), b1 = sizeof(WgMegaItem) * items.size(), b2 = sizeof(int) * starts.size();
    const size_t off_items = (b0 + 255) & ~(size_t)255, off_starts = (off_items + b1 + 255) & ~(size_t)255;
    std::vector<char> key(off_starts + b2, 0);
    memcpy(key.data(), ent.data(), b0);
    memcpy(key.data() + off_items, items.data(), b1);
    memcpy(key.data() + off_starts, starts.data(), b2);
    int dev = 0;
    SSDN_CHECK_HIP(hipGetDevice(&dev));
    char* dtab = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_wg_multi_mutex);
        for (const WgMegaCache& c : g_wg_mega_cache)
            if (c.device == dev && c.key.size() == key.size() && !memcmp(c.key.data(), key.data(), key.size())) { dtab = c.dev; break; }
        if (!dtab) {
            if (g_wg_mega_cache.size() >= 64) {
                SSDN_CHECK_HIP(hipDeviceSynchronize());
                SSDN_CHECK_HIP(hipFree(g_wg_mega_cache.front().dev));
                g_wg_mega_cache.erase(g_wg_mega_cache.begin());
            }
            SSDN_CHECK_HIP(hipMalloc((void**)&dtab, key.size()));
            SSDN_CHECK_HIP(hipMemcpy(dtab, key.data(), key.size(), hipMemcpyHostToDevice));
            g_wg_mega_cache.push_back({key, dtab, dev});
        }
    }
    static bool attr_set[16] = {};
    if (dev >= 0 && dev < 16 && !attr_set[dev]) {
        SSDN_CHECK_HIP(hipFuncSetAttribute((const void*)k_wgrad_mega, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[dev] = true;
    }
    if (lds > 160 * 1024) return ssdn_set_error("wgrad: merged launch needs %zu B of LDS", lds);
    const int prof_kind = W < ssdn_device_cus() ? SSDN_PROF_WGRAD_SIDE : SSDN_PROF_WGRAD;
    prof_begin(prof_kind, s);
    SSDN_LAUNCH(k_wgrad_mega, dim3((unsigned)items.size()), dim3(WG_THREADS), lds, s, (const WgMegaEntry*)dtab, (const WgMegaItem*)(dtab + off_items));
    prof_end(prof_kind, s, flops, bytes);
    SSDN_CHECK_HIP(hipGetLastError());
    return 0;
}


// api.hip -- the extern "C" surface of libssdn_hip.so (see include/ssdn_hip.h): op-list executor, error text,
// device query and two hardware probes used by the GPU test-suite.
#include "common.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

static thread_local char g_err[512] = "";
thread_local hipEvent_t g_ssdn_stop_event = nullptr;
thread_local bool g_ssdn_stop_used = false;
thread_local hipEvent_t g_ssdn_prof_start = nullptr, g_ssdn_prof_stop = nullptr;
thread_local bool g_ssdn_prof_used = false;

int ssdn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}

// ---- in-stream kernel profiler: HIP events around every launch of a chosen kernel family (bench.py's roofline leg) ----
#include <vector>
struct ProfSlot {
    bool on = false;
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    double flops = 0.0, bytes = 0.0;
    long long seen = 0;      // launches of the family since enable
    int stride = 1;          // every stride-th launch is bracketed (an event pair costs ~10 us of stream time)
    bool armed = false;      // the launch in flight between prof_begin and prof_end is a sampled one
    bool attached = false;   // ... and its two events ride on the kernel's dispatch (SSDN_LAUNCH) instead of being recorded around it
};
// kinds whose bracket holds exactly ONE SSDN_LAUNCH (csrc/conv_dma.hip::cd_launch, csrc/gemm_dma.hip::gd_launch)
static bool prof_single_launch(int id) { return id == SSDN_PROF_CDMA_MT3 || id == SSDN_PROF_CDMA_MT21 || id == SSDN_PROF_GEMM; }
static ProfSlot g_prof[SSDN_PROF_KINDS];

void prof_begin(int id, hipStream_t s) {
    ProfSlot& p = g_prof[id];
    p.armed = false;
    if (!p.on) return;
    const bool pick = p.seen++ % p.stride == 0;
    if (!pick || p.used + 2 > p.ev.size()) return;
    p.armed = true;
    // (a launch that already carries another lane's stop event keeps the recorded bracket)
    p.attached = prof_single_launch(id) && !g_ssdn_stop_event;
    if (p.attached) {
        g_ssdn_prof_start = p.ev[p.used];
        g_ssdn_prof_stop = p.ev[p.used + 1];
        g_ssdn_prof_used = false;
    } else (void)hipEventRecord(p.ev[p.used], s);
}
void prof_end(int id, hipStream_t s, double flops, double bytes) {
    ProfSlot& p = g_prof[id];
    if (!p.on || !p.armed) return;
    p.armed = false;
    if (p.attached) {
        const bool used = g_ssdn_prof_used;
        g_ssdn_prof_start = g_ssdn_prof_stop = nullptr;
        g_ssdn_prof_used = false;
        if (!used) return;                                     // (no SSDN_LAUNCH in the bracket: no sample)
    } else (void)hipEventRecord(p.ev[p.used + 1], s);
    p.used += 2;
    p.flops += flops;
    p.bytes += bytes;
}

__global__ void k_zero(uint4* p, long long n16) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n16; i += (long long)gridDim.x * blockDim.x) p[i] = make_uint4(0, 0, 0, 0);
}

extern "C" {

int ssdn_abi_version(void) { return SSDN_ABI_VERSION; }
const char* ssdn_last_error(void) { return g_err; }

int ssdn_device_cus(void) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ssdn_set_error("hipGetDevice failed");
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return ssdn_set_error("hipDeviceGetAttribute failed");
    return cus;
}

/* sizeof() of the args struct of an op type -- lets the Python binding verify its ctypes mirrors (tests/test_abi.py) */
int ssdn_struct_size(int op_type) {
    switch (op_type) {
        case 0: return (int)sizeof(ssdn_op);
        case SSDN_OP_PACK_INPUT: return (int)sizeof(ssdn_pack_input_args);
        case SSDN_OP_CONV: return (int)sizeof(ssdn_conv_args);
        case SSDN_OP_POOL_FWD: case SSDN_OP_POOL_BWD: return (int)sizeof(ssdn_pool_args);
        case SSDN_OP_UPSUM_BWD: return (int)sizeof(ssdn_upsum_args);
        case SSDN_OP_UNROT_FWD: case SSDN_OP_UNROT_BWD: return (int)sizeof(ssdn_unrot_args);
        case SSDN_OP_WGRAD: return (int)sizeof(ssdn_wgrad_args);
        case SSDN_OP_WREDUCE: return (int)sizeof(ssdn_wreduce_args);
        case SSDN_OP_WPACK: return (int)sizeof(ssdn_wpack_args);
        case SSDN_OP_GRAD_PACK: return (int)sizeof(ssdn_grad_pack_args);
        case SSDN_OP_HEAD_SSDN: return (int)sizeof(ssdn_head_args);
        case SSDN_OP_HEAD_FINAL: return (int)sizeof(ssdn_head_final_args);
        case SSDN_OP_SPATIAL_MEAN: return (int)sizeof(ssdn_spatial_mean_args);
        case SSDN_OP_MSE: case SSDN_OP_MASK_MSE: return (int)sizeof(ssdn_mse_args);
        case SSDN_OP_ADAM: return (int)sizeof(ssdn_adam_args);
        case SSDN_OP_METRICS: return (int)sizeof(ssdn_metrics_args);
        case SSDN_OP_ZERO: return (int)sizeof(ssdn_zero_args);
        case SSDN_OP_EVENT_RECORD: return (int)sizeof(ssdn_event_args);
        case SSDN_OP_NOISE: return (int)sizeof(ssdn_noise_args);
        default: return -1;
    }
}

int ssdn_profile_enable(int kind, int max_launches) {
    if (kind < 0 || kind >= SSDN_PROF_KINDS) return ssdn_set_error("profile: bad kernel kind %d", kind);
    ProfSlot& p = g_prof[kind];
    for (hipEvent_t e : p.ev) (void)hipEventDestroy(e);
    p.ev.clear();
    p.used = 0;
    p.flops = p.bytes = 0.0;
    p.on = max_launches > 0;
    p.seen = 0;
    p.armed = false;
    for (int i = 0; i < 2 * max_launches; ++i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return ssdn_set_error("profile: hipEventCreate failed");
        p.ev.push_back(e);
    }
    return 0;
}

int ssdn_profile_set_stride(int kind, int stride) {
    if (kind < 0 || kind >= SSDN_PROF_KINDS) return ssdn_set_error("profile: bad kernel kind %d", kind);
    if (stride < 1) return ssdn_set_error("profile: stride must be >= 1");
    g_prof[kind].stride = stride;
    return 0;
}

int ssdn_profile_read(int kind, double* total_ms, long long* launches, double* flops, double* bytes) {
    if (kind < 0 || kind >= SSDN_PROF_KINDS) return ssdn_set_error("profile: bad kernel kind %d", kind);
    ProfSlot& p = g_prof[kind];
    double ms = 0.0;
    for (size_t i = 0; i + 1 < p.used; i += 2) {
        if (hipEventSynchronize(p.ev[i + 1]) != hipSuccess) return ssdn_set_error("profile: hipEventSynchronize failed");
        float t = 0.f;
        if (hipEventElapsedTime(&t, p.ev[i], p.ev[i + 1]) != hipSuccess) return ssdn_set_error("profile: hipEventElapsedTime failed");
        ms += t;
    }
    *total_ms = ms;
    *launches = (long long)(p.used / 2);
    *flops = p.flops;
    *bytes = p.bytes;
    p.used = 0;
    p.flops = p.bytes = 0.0;
    return 0;
}

int ssdn_conv_lds_bytes(const ssdn_conv_args* a) { return conv_lds_bytes(a); }
int ssdn_wgrad_lds_bytes(const ssdn_wgrad_args* a) { return wgrad_lds_bytes(a); }
int ssdn_wgrad_mergeable(const ssdn_wgrad_args* a) { return a && wgrad_mergeable(a) ? 1 : 0; }
int ssdn_wgrad_mega_ok(const ssdn_wgrad_args* a) { return wgrad_mega_ok(a); }
int ssdn_conv_fuses_pool(const ssdn_conv_args* a) { return a && conv_fuses_pool(a) ? 1 : 0; }
int ssdn_conv_fuses_upsum(const ssdn_conv_args* a) { return a && conv_fuses_upsum(a) ? 1 : 0; }
int ssdn_conv_fuses_unrot(const ssdn_conv_args* a) { return a && conv_fuses_unrot(a) ? 1 : 0; }
int ssdn_conv_fuses_urot(const ssdn_conv_args* a) { return a && conv_fuses_urot(a) ? 1 : 0; }
int ssdn_conv_signs(const ssdn_conv_args* a) { return a && conv_signs(a) ? 1 : 0; }

#define SSDN_NEVENTS 256
#define SSDN_NLANES 4
#define SSDN_MAX_DEVICES 16
// side streams and dependency events belong to ONE device: one set per device ordinal, created on first use while that
// device is current (a process normally drives a single GPU, but nothing here assumes it)
struct LaneSet {
    hipStream_t side[SSDN_NLANES] = {nullptr, nullptr, nullptr, nullptr};   // [0] unused (= caller's stream)
    hipEvent_t ev[SSDN_NEVENTS];
    int ev_next = 0;
    bool ready = false;
};
static LaneSet g_lanesets[SSDN_MAX_DEVICES];
static LaneSet* lanes_get() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SSDN_MAX_DEVICES) { ssdn_set_error("lanes: bad current device"); return nullptr; }
    LaneSet& L = g_lanesets[dev];
    if (L.ready) return &L;
    for (int l = 1; l < SSDN_NLANES; ++l)
        if (hipStreamCreateWithFlags(&L.side[l], hipStreamNonBlocking) != hipSuccess) { ssdn_set_error("lanes: hipStreamCreate failed"); return nullptr; }
    for (int i = 0; i < SSDN_NEVENTS; ++i)
        if (hipEventCreateWithFlags(&L.ev[i], hipEventDisableTiming) != hipSuccess) { ssdn_set_error("lanes: hipEventCreate failed"); return nullptr; }
    L.ready = true;
    return &L;
}
// lanes a lane is ordered after (bit l = lane l): see ssdn_op in the header
static const unsigned g_lane_deps[SSDN_NLANES] = {0u, 1u << 0, (1u << 0) | (1u << 1) | (1u << 3), 1u << 0};

int ssdn_stream_order(void* first, void* then) {
    LaneSet* LS = lanes_get();
    if (!LS) return -1;
    hipEvent_t e = LS->ev[LS->ev_next++ % SSDN_NEVENTS];
    SSDN_CHECK_HIP(hipEventRecord(e, (hipStream_t)first));
    SSDN_CHECK_HIP(hipStreamWaitEvent((hipStream_t)then, e, 0));
    return 0;
}

int ssdn_run_ops(const ssdn_op* ops, int n, void* stream) {
    hipStream_t lane_s[SSDN_NLANES] = {(hipStream_t)stream, nullptr, nullptr, nullptr};
    g_ssdn_stop_event = nullptr;                                   // (an earlier call may have left through an error path)
    // dirty[s][d]: lane s has enqueued work that lane d (which depends on s) has not been ordered after yet
    bool dirty[SSDN_NLANES][SSDN_NLANES] = {}, used[SSDN_NLANES] = {true, false, false, false};
    for (int d = 1; d < SSDN_NLANES; ++d) dirty[0][d] = true;   // whatever the caller enqueued before this list
    static const bool one_lane = ssdn_tuning_env("SSDN_ONE_LANE") != nullptr;   // tuning / debugging aid
    LaneSet* LS = nullptr;
    // cover[s]: an event that stands for everything lane s has enqueued so far (covered[s]); a kernel whose completion the NEXT op of the
    // list (another lane) waits for carries one as its stop event (SSDN_LAUNCH): the dependent lane waits without a hipEventRecord
    hipEvent_t cover[SSDN_NLANES] = {nullptr, nullptr, nullptr, nullptr};
    bool covered[SSDN_NLANES] = {false, false, false, false};
    // ring position at which cover[s] was drawn: a cover stays live until the lane's next op or the final join, while every later
    // draw advances the ring -- once SSDN_NEVENTS - 1 further events have been drawn the slot has been handed out again and the
    // cover no longer stands for lane s: it is dropped, and whoever needs it records a fresh event
    long long cover_at[SSDN_NLANES] = {0, 0, 0, 0};
    auto cover_live = [&](int l) { return covered[l] && (!LS || (long long)LS->ev_next - cover_at[l] < SSDN_NEVENTS - 1); };
    bool has_side = false;
    static const bool no_stop = ssdn_tuning_env("SSDN_NO_STOP_EVENTS") != nullptr;      // A/B aid, read once
    for (int i = 0; i < n && !one_lane; ++i) has_side = has_side || ops[i].lane > 0;
    if (has_side && !(LS = lanes_get())) return -1;
    for (int i = 0; i < n; ++i) {
        const void* p = ops[i].args;
        int rc = 0;
        if (!p) return ssdn_set_error("op %d: null args", i);
        int lane = one_lane ? 0 : ops[i].lane;
        if (lane < 0 || lane >= SSDN_NLANES) return ssdn_set_error("op %d: bad lane %d", i, lane);
        if (lane > 0) {
            if (!LS && !(LS = lanes_get())) return -1;
            for (int l = 1; l < SSDN_NLANES; ++l) lane_s[l] = LS->side[l];
            for (int src = 0; src < SSDN_NLANES; ++src) {
                if (ops[i].type == SSDN_OP_EVENT_RECORD) break;      // a mark stands for its OWN lane's work so far, nothing else
                if (!((g_lane_deps[lane] >> src) & 1) || !dirty[src][lane]) continue;
                if (!cover_live(src)) {
                    cover_at[src] = LS->ev_next;
                    cover[src] = LS->ev[LS->ev_next++ % SSDN_NEVENTS];
                    SSDN_CHECK_HIP(hipEventRecord(cover[src], lane_s[src]));
                    covered[src] = true;
                }
                SSDN_CHECK_HIP(hipStreamWaitEvent(lane_s[lane], cover[src], 0));
                dirty[src][lane] = false;
            }
        }
        // (a mark adds no work to its lane: whatever covers the lane's work so far still does)
        const bool is_mark = ops[i].type == SSDN_OP_EVENT_RECORD;
        if (!is_mark) {
            for (int d = 0; d < SSDN_NLANES; ++d) dirty[lane][d] = true;
            covered[lane] = false;
        }
        used[lane] = true;
        hipStream_t s = lane_s[lane];
        // arm(j): the op (or merged run) being launched ends at list index j - 1; if the op at j runs on a lane that is ordered after
        // this one, the launch carries a stop event (attaching one to EVERY kernel costs each ~5 us of completion handling)
        bool armed = false;
        long long armed_at = 0;
        auto arm = [&](int j) {
            if (!has_side || no_stop || j >= n) return;
            const int lj = one_lane ? 0 : ops[j].lane;
            if (lj == lane || lj < 0 || lj >= SSDN_NLANES || !((g_lane_deps[lj] >> lane) & 1)) return;
            armed_at = LS->ev_next;
            g_ssdn_stop_event = LS->ev[LS->ev_next++ % SSDN_NEVENTS];
            g_ssdn_stop_used = false;
            armed = true;
        };
        // the last op of a side lane: the join at the end of the list waits for its stop event
        auto arm_last = [&](int j) {
            if (!has_side || no_stop || lane == 0 || armed) return;
            for (; j < n; ++j)
                if ((one_lane ? 0 : ops[j].lane) == lane && ops[j].type != SSDN_OP_EVENT_RECORD) return;
            armed_at = LS->ev_next;
            g_ssdn_stop_event = LS->ev[LS->ev_next++ % SSDN_NEVENTS];
            g_ssdn_stop_used = false;
            armed = true;
        };
        switch (ops[i].type) {
            case SSDN_OP_PACK_INPUT: {   // ... directly followed by the thin first layer that reads it: one launch (conv_thin.hip)
                const bool next_conv = i + 1 < n && ops[i + 1].type == SSDN_OP_CONV && ops[i + 1].args && (one_lane ? 0 : ops[i + 1].lane) == lane;
                if (next_conv && chain_merging_on() && conv_pack_fusable((const ssdn_pack_input_args*)p, (const ssdn_conv_args*)ops[i + 1].args)) {
                    rc = launch_conv_thin((const ssdn_conv_args*)ops[i + 1].args, (const ssdn_pack_input_args*)p, s);
                    ++i;
                } else rc = launch_pack_input((const ssdn_pack_input_args*)p, s);
                break;
            }
            case SSDN_OP_CONV: {    // a run of consecutive small-image ops on the same lane is one launch (conv_chain.hip)
                const int m = chain_len(ops + i, n - i, one_lane);
                if (m < 0) return -1;
                // ... and the narrow net_out layer directly behind the 96-channel 1x1 layer rides in that layer's launch (gemm_dma.hip)
                const bool pair = m <= 1 && i + 1 < n && ops[i + 1].type == SSDN_OP_CONV && ops[i + 1].args && (one_lane ? 0 : ops[i + 1].lane) == lane &&
                                  chain_merging_on() && conv_pair_fusable((const ssdn_conv_args*)p, (const ssdn_conv_args*)ops[i + 1].args);
                arm(i + (m > 1 ? m : (pair ? 2 : 1)));
                if (m > 1) { rc = launch_chain(ops + i, m, one_lane, s); i += m - 1; }
                else if (pair) { rc = launch_gemm_dma_with_next((const ssdn_conv_args*)p, (const ssdn_conv_args*)ops[i + 1].args, s); ++i; }
                else rc = launch_conv((const ssdn_conv_args*)p, s);
                break;
            }
            case SSDN_OP_POOL_FWD: rc = launch_pool_fwd((const ssdn_pool_args*)p, s); break;
            case SSDN_OP_POOL_BWD: arm(i + 1); rc = launch_pool_bwd((const ssdn_pool_args*)p, s); break;
            case SSDN_OP_UPSUM_BWD: arm(i + 1); rc = launch_upsum_bwd((const ssdn_upsum_args*)p, s); break;
            case SSDN_OP_UNROT_FWD: rc = launch_unrot_fwd((const ssdn_unrot_args*)p, s); break;
            case SSDN_OP_UNROT_BWD: arm(i + 1); rc = launch_unrot_bwd((const ssdn_unrot_args*)p, s); break;
            case SSDN_OP_WGRAD: {   // a run of consecutive small-layer weight-gradient GEMMs on the same lane is one launch
                static const bool no_merge = ssdn_tuning_env("SSDN_NO_WGRAD_MERGE") != nullptr;      // A/B aid, read once
                // ... and a run of ops planned for one chip-wide launch (ssdn_wgrad_args.mega) is ONE launch, one workgroup per CU
                {
                    const ssdn_wgrad_args* mg[WGRAD_MEGA_MAX];
                    int mm = 0;
                    while (!no_merge && mm < WGRAD_MEGA_MAX && i + mm < n && ops[i + mm].type == SSDN_OP_WGRAD && ops[i + mm].args &&
                           (one_lane ? 0 : ops[i + mm].lane) == lane && wgrad_mega_ok((const ssdn_wgrad_args*)ops[i + mm].args) &&
                           ((const ssdn_wgrad_args*)ops[i + mm].args)->mega == ((const ssdn_wgrad_args*)p)->mega)
                        mg[mm] = (const ssdn_wgrad_args*)ops[i + mm].args, ++mm;
                    if (mm > 1) { arm(i + mm); rc = launch_wgrad_mega(mg, mm, s); i += mm - 1; break; }
                }
                const ssdn_wgrad_args* items[WGRAD_MULTI_MAX];
                int m = 0;
                while (!no_merge && m < WGRAD_MULTI_MAX && i + m < n && ops[i + m].type == SSDN_OP_WGRAD && ops[i + m].args &&
                       (one_lane ? 0 : ops[i + m].lane) == lane && wgrad_mergeable((const ssdn_wgrad_args*)ops[i + m].args))
                    items[m] = (const ssdn_wgrad_args*)ops[i + m].args, ++m;
                if (m > 1) { rc = launch_wgrad_multi(items, m, s); i += m - 1; }
                else rc = launch_wgrad((const ssdn_wgrad_args*)p, s);
                break;
            }
            case SSDN_OP_WREDUCE: {   // a run of consecutive reductions on the same lane is two launches in total
                const ssdn_wreduce_args* items[WREDUCE_MULTI_MAX];
                int m = 0;
                while (m < WREDUCE_MULTI_MAX && i + m < n && ops[i + m].type == SSDN_OP_WREDUCE && ops[i + m].args &&
                       (one_lane ? 0 : ops[i + m].lane) == lane)
                    items[m] = (const ssdn_wreduce_args*)ops[i + m].args, ++m;
                arm_last(i + m);
                rc = m > 1 ? launch_wreduce_multi(items, m, s) : launch_wreduce((const ssdn_wreduce_args*)p, s);
                i += m - 1;
                break;
            }
            case SSDN_OP_WPACK: {   // a run of consecutive re-packs on the same lane is one launch
                const ssdn_wpack_args* items[WPACK_MULTI_MAX];
                int m = 0;
                while (m < WPACK_MULTI_MAX && i + m < n && ops[i + m].type == SSDN_OP_WPACK && ops[i + m].args &&
                       (one_lane ? 0 : ops[i + m].lane) == lane)
                    items[m] = (const ssdn_wpack_args*)ops[i + m].args, ++m;
                rc = m > 1 ? launch_wpack_multi(items, m, s) : launch_wpack((const ssdn_wpack_args*)p, s);
                i += m - 1;
                break;
            }
            case SSDN_OP_GRAD_PACK: {    // ... directly followed by the narrow layer's data gradient that reads it: one launch (gradpack_dgrad.hip)
                const bool next_conv = i + 1 < n && ops[i + 1].type == SSDN_OP_CONV && ops[i + 1].args && (one_lane ? 0 : ops[i + 1].lane) == lane;
                if (next_conv && chain_merging_on() && conv_gradpack_fusable((const ssdn_grad_pack_args*)p, (const ssdn_conv_args*)ops[i + 1].args)) {
                    arm(i + 2);
                    rc = launch_gradpack_dgrad((const ssdn_grad_pack_args*)p, (const ssdn_conv_args*)ops[i + 1].args, s);
                    ++i;
                } else { arm(i + 1); rc = launch_grad_pack((const ssdn_grad_pack_args*)p, s); }
                break;
            }
            case SSDN_OP_HEAD_SSDN: rc = launch_head((const ssdn_head_args*)p, s); break;
            case SSDN_OP_HEAD_FINAL: rc = launch_head_final((const ssdn_head_final_args*)p, s); break;
            case SSDN_OP_SPATIAL_MEAN: rc = launch_spatial_mean((const ssdn_spatial_mean_args*)p, s); break;
            case SSDN_OP_MSE: rc = launch_mse((const ssdn_mse_args*)p, 0, s); break;
            case SSDN_OP_MASK_MSE: rc = launch_mse((const ssdn_mse_args*)p, 1, s); break;
            case SSDN_OP_ADAM: {    // ... directly followed by the re-packs of its layers: one launch (k_adam_pack)
                const ssdn_wpack_args* items[ADAM_PACK_MAX];
                int m = 0;
                while (m < ADAM_PACK_MAX && i + 1 + m < n && ops[i + 1 + m].type == SSDN_OP_WPACK && ops[i + 1 + m].args &&
                       (one_lane ? 0 : ops[i + 1 + m].lane) == lane)
                    items[m] = (const ssdn_wpack_args*)ops[i + 1 + m].args, ++m;
                const bool whole_run = !(i + 1 + m < n && ops[i + 1 + m].type == SSDN_OP_WPACK);      // (never split a run of re-packs)
                if (m > 0 && whole_run && adam_pack_fusable((const ssdn_adam_args*)p, items, m)) {
                    rc = launch_adam_pack((const ssdn_adam_args*)p, items, m, s);
                    i += m;
                } else rc = launch_adam((const ssdn_adam_args*)p, s);
                break;
            }
            case SSDN_OP_METRICS: rc = launch_metrics((const ssdn_metrics_args*)p, s); break;
            case SSDN_OP_NOISE: rc = launch_noise((const ssdn_noise_args*)p, s); break;
            case SSDN_OP_ZERO: {
                const ssdn_zero_args* z = (const ssdn_zero_args*)p;
                if (z->bytes & 15) return ssdn_set_error("op %d: zero size must be a multiple of 16", i);
                long long n16 = z->bytes / 16;
                int g = (int)((n16 + 255) / 256);
                if (g > 2048) g = 2048;
                if (g > 0) hipLaunchKernelGGL(k_zero, dim3(g), dim3(256), 0, s, (uint4*)z->p, n16);
                break;
            }
            case SSDN_OP_EVENT_RECORD: {
                const ssdn_event_args* e = (const ssdn_event_args*)p;
                if (!e->event) return ssdn_set_error("op %d: null event", i);
                SSDN_CHECK_HIP(hipEventRecord((hipEvent_t)e->event, s));
                break;
            }
            default: return ssdn_set_error("op %d: unknown type %d", i, ops[i].type);
        }
        if (armed) {
            if (g_ssdn_stop_used && !rc) { cover[lane] = g_ssdn_stop_event; covered[lane] = true; cover_at[lane] = armed_at; }   // (the op's last launch carries it)
            g_ssdn_stop_event = nullptr;
        }
        if (rc) {
            char tmp[400];
            snprintf(tmp, sizeof(tmp), "%s", g_err);
            return ssdn_set_error("op %d (type %d): %s", i, ops[i].type, tmp);
        }
    }
    for (int l = 1; l < SSDN_NLANES; ++l) {   // join every side lane back into the caller's stream
        if (!used[l]) continue;
        if (!cover_live(l)) {
            cover[l] = LS->ev[LS->ev_next++ % SSDN_NEVENTS];
            SSDN_CHECK_HIP(hipEventRecord(cover[l], lane_s[l]));
        }
        SSDN_CHECK_HIP(hipStreamWaitEvent(lane_s[0], cover[l], 0));
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ssdn_set_error("launch error: %s", hipGetErrorString(e));
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// hardware probes (test infrastructure on the device side; they pin the lane maps the MFMA kernels rely on)
// ---------------------------------------------------------------------------------------------------------
__global__ void k_probe_mfma(const half8* a, const half8* b, float* d) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) d[threadIdx.x * 16 + r] = acc[r];
}

typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 p_fp16x4_t;
typedef __attribute__((address_space(3))) p_fp16x4_t p_lds_fp16x4;
__global__ void k_probe_tr16(const uint4* image, int n16, const int* lane_addr, half4* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < n16; i += blockDim.x) reinterpret_cast<uint4*>(smem)[i] = image[i];
    __syncthreads();
    p_fp16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((p_lds_fp16x4*)(smem + lane_addr[threadIdx.x]));
    out[threadIdx.x] = __builtin_bit_cast(half4, r);
}

extern "C" {
int ssdn_probe_mfma(const void* a_frag, const void* b_frag, float* d_out, void* stream) {
    hipLaunchKernelGGL(k_probe_mfma, dim3(1), dim3(64), 0, (hipStream_t)stream, (const half8*)a_frag, (const half8*)b_frag, d_out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : ssdn_set_error("probe_mfma: %s", hipGetErrorString(e));
}
int ssdn_probe_tr16(const void* lds_image, int image_bytes, const int32_t* lane_addr, void* out, void* stream) {
    if (image_bytes & 15 || image_bytes > 64 * 1024) return ssdn_set_error("probe_tr16: bad image size");
    hipLaunchKernelGGL(k_probe_tr16, dim3(1), dim3(64), image_bytes, (hipStream_t)stream, (const uint4*)lds_image,
                       image_bytes / 16, lane_addr, (half4*)out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : ssdn_set_error("probe_tr16: %s", hipGetErrorString(e));
}
}

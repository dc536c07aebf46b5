// plan.hip -- step-level entry points (ssdn_net_forward / ssdn_train_step): a PLAN is what ssdn/hip/graph.py +
// engine.py make of one configuration and input shape -- the op lists of a whole step with every pointer replaced by (tensor, offset) --
// written to a blob by DenoiserEngine.export_plan().  This file loads such a blob, lays its tensors out in ONE caller-owned device arena,
// patches the pointers and runs the lists through ssdn_run_ops: a binder needs libssdn_hip.so and the blob, not the Python package
// (tests/test_hip_plan_c.py drives it from a script that imports neither `ssdn` nor anything of this repository).
// replaces: one iteration of the reference's training loop, train.py:196-202 (run_pipeline + mean(LOSS).backward() + optimizer.step()).
#include "common.h"
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

namespace {
struct PlanReloc { uint32_t off, tensor; uint64_t delta; };
struct PlanOp { int32_t type, lane; std::vector<unsigned char> args; std::vector<PlanReloc> rel; };
struct PlanTensor { std::string name; uint64_t bytes; int32_t alias; uint64_t off; };
}
struct ssdn_plan {
    std::vector<PlanTensor> t;
    std::vector<PlanOp> ops[SSDN_PLAN_PHASES];
    std::vector<ssdn_op> recs[SSDN_PLAN_PHASES];
    std::string meta;
    uint64_t arena_bytes = 0;
    char* arena = nullptr;
};

namespace {
struct Rd {
    const unsigned char* p; int64_t n, i = 0; bool bad = false;
    template <class T> T get() { T v{}; if (i + (int64_t)sizeof(T) > n) { bad = true; return v; } memcpy(&v, p + i, sizeof(T)); i += sizeof(T); return v; }
    const unsigned char* bytes(int64_t k) { if (k < 0 || i + k > n) { bad = true; return nullptr; } const unsigned char* q = p + i; i += k; return q; }
};
}

extern "C" {

int ssdn_plan_load(const void* blob, int64_t nbytes, ssdn_plan** out) {
    if (!blob || !out || nbytes < 32) return ssdn_set_error("plan: no blob");
    Rd r{(const unsigned char*)blob, nbytes};
    const unsigned char* magic = r.bytes(8);
    if (r.bad || memcmp(magic, "SSDNPLAN", 8)) return ssdn_set_error("plan: bad magic");
    const uint32_t ver = r.get<uint32_t>(), abi = r.get<uint32_t>(), nt = r.get<uint32_t>(), nph = r.get<uint32_t>();
    if (ver != 1 || nph != SSDN_PLAN_PHASES) return ssdn_set_error("plan: blob version %u with %u phases (this library: 1, %d)", ver, nph, SSDN_PLAN_PHASES);
    if (abi != SSDN_ABI_VERSION) return ssdn_set_error("plan: blob was written for ABI %u, this library is ABI %d (argument structs may differ)", abi, SSDN_ABI_VERSION);
    ssdn_plan* P = new ssdn_plan;
    uint64_t off = 0;
    for (uint32_t k = 0; k < nt && !r.bad; ++k) {
        const unsigned char* nm = r.bytes(56);
        PlanTensor t;
        if (nm) t.name.assign((const char*)nm, strnlen((const char*)nm, 56));
        t.bytes = r.get<uint64_t>();
        t.alias = r.get<int32_t>();
        (void)r.get<int32_t>();
        if (t.alias >= 0) {      // a view of an earlier tensor: never larger than what it views (a relocation is bounded by the alias's own size)
            if ((uint32_t)t.alias >= k || t.bytes > P->t[t.alias].bytes) r.bad = true; else t.off = P->t[t.alias].off;
        }
        else { t.off = off; off += (t.bytes + 255) & ~255ull; }
        P->t.push_back(t);
    }
    P->arena_bytes = off;
    for (int ph = 0; ph < SSDN_PLAN_PHASES && !r.bad; ++ph) {
        const uint32_t nops = r.get<uint32_t>();
        for (uint32_t k = 0; k < nops && !r.bad; ++k) {
            PlanOp op;
            op.type = r.get<int32_t>(); op.lane = r.get<int32_t>();
            const uint32_t ab = r.get<uint32_t>(), nr = r.get<uint32_t>();
            const int want = ssdn_struct_size(op.type);
            if (want < 0 || (uint32_t)want != ab) { delete P; return ssdn_set_error("plan: op type %d with %u argument bytes (this library: %d)", op.type, ab, want); }
            if (op.type == SSDN_OP_EVENT_RECORD) { delete P; return ssdn_set_error("plan: event records do not travel in a blob"); }
            const unsigned char* a = r.bytes((ab + 7) & ~7u);
            if (a) op.args.assign(a, a + ab);
            for (uint32_t j = 0; j < nr && !r.bad; ++j) {
                PlanReloc q;
                q.off = r.get<uint32_t>(); q.tensor = r.get<uint32_t>(); q.delta = r.get<uint64_t>();
                if ((uint64_t)q.off + 8 > ab || q.tensor >= nt || q.delta > P->t[q.tensor].bytes) r.bad = true;      // (64-bit: q.off + 8 must not wrap)
                op.rel.push_back(q);
            }
            P->ops[ph].push_back(std::move(op));
        }
    }
    const uint32_t nm = r.get<uint32_t>();
    const unsigned char* m = r.bytes(nm);
    if (m) P->meta.assign((const char*)m, nm);
    if (r.bad) { delete P; return ssdn_set_error("plan: truncated or inconsistent blob"); }
    *out = P;
    return 0;
}

void ssdn_plan_destroy(ssdn_plan* P) { delete P; }
int64_t ssdn_plan_arena_bytes(const ssdn_plan* P) { return P ? (int64_t)P->arena_bytes : -1; }
const char* ssdn_plan_meta(const ssdn_plan* P) { return P ? P->meta.c_str() : ""; }

int ssdn_plan_bind(ssdn_plan* P, void* arena) {
    if (!P || !arena) return ssdn_set_error("plan: bind needs a plan and a device arena");
    if ((uintptr_t)arena & 255) return ssdn_set_error("plan: the arena must be 256-byte aligned");
    P->arena = (char*)arena;
    for (int ph = 0; ph < SSDN_PLAN_PHASES; ++ph) {
        P->recs[ph].clear();
        for (PlanOp& op : P->ops[ph]) {
            for (const PlanReloc& q : op.rel) {
                const uint64_t v = (uint64_t)(uintptr_t)(P->arena + P->t[q.tensor].off + q.delta);
                memcpy(op.args.data() + q.off, &v, 8);
            }
            ssdn_op rec;
            rec.type = op.type; rec.lane = op.lane; rec.args = op.args.data();
            P->recs[ph].push_back(rec);
        }
    }
    return 0;
}

int ssdn_plan_tensor(const ssdn_plan* P, const char* name, void** ptr, int64_t* bytes) {
    if (!P || !name) return ssdn_set_error("plan: tensor lookup needs a plan and a name");
    for (const PlanTensor& t : P->t)
        if (t.name == name) {
            if (ptr) *ptr = P->arena ? (void*)(P->arena + t.off) : nullptr;
            if (bytes) *bytes = (int64_t)t.bytes;
            return 0;
        }
    return ssdn_set_error("plan: no tensor named %s", name);
}

int ssdn_plan_run(ssdn_plan* P, int phase, void* stream) {
    if (!P || !P->arena) return ssdn_set_error("plan: not bound to an arena");
    if (phase < 0 || phase >= SSDN_PLAN_PHASES) return ssdn_set_error("plan: bad phase %d", phase);
    if (P->recs[phase].empty()) return 0;
    return ssdn_run_ops(P->recs[phase].data(), (int)P->recs[phase].size(), stream);
}

int ssdn_plan_set_lr(ssdn_plan* P, float lr, int step, float gscale) {
    if (!P) return ssdn_set_error("plan: null");
    if (step < 1) return ssdn_set_error("plan: Adam steps count from 1");
    int n = 0;
    for (PlanOp& op : P->ops[SSDN_PLAN_OPTIMISER])
        if (op.type == SSDN_OP_ADAM) {
            ssdn_adam_args* a = (ssdn_adam_args*)op.args.data();
            a->lr = lr; a->gscale = gscale;
            a->bc1 = (float)(1.0 - std::pow(0.9, (double)step)); a->bc2 = (float)(1.0 - std::pow(0.99, (double)step));    // (in double, as the Python host does)
            ++n;
        }
    return n ? 0 : ssdn_set_error("plan: no optimiser in this plan (an inference plan)");
}

int ssdn_net_forward(ssdn_plan* P, void* stream) { return ssdn_plan_run(P, SSDN_PLAN_FORWARD, stream); }

int ssdn_train_step(ssdn_plan* P, float lr, int step, void* stream) {
    int rc = ssdn_plan_set_lr(P, lr, step, 1.f);
    if (!rc) rc = ssdn_plan_run(P, SSDN_PLAN_FORWARD, stream);
    if (!rc) rc = ssdn_plan_run(P, SSDN_PLAN_BACKWARD, stream);
    if (!rc) rc = ssdn_plan_run(P, SSDN_PLAN_OPTIMISER, stream);
    return rc;
}

}  // extern "C"

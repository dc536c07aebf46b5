// probe_kloop.hip -- hardware probe (tuning aid, not part of the library): what does the K loop of k_cdma<3,*> (csrc/conv_dma.hip) cost
// per tap step, as a function of the LDS -> register -> MFMA schedule?  Everything stays in LDS (no DMA, no epilogue): a tap step is
// 18 x v_mfma_f32_32x32x16_f16 (3 weight-row blocks x 2 pixel blocks x 3 K-steps of a 48-channel chunk) fed by 15 ds_read_b128 with
// exactly the fragment addresses of the kernel (conflict-free swizzle).
//   variant 0: the kernel's round-2..4 schedule  -- per K-step: wait-all, 5 reads of the NEXT K-step, 6 MFMAs; pipeline drained at every step
//   variant 1: fine-grained schedule -- one ds_read of the next K-step behind each MFMA, partial lgkmcnt waits, pipeline continuous across
//              steps; the step barrier sits in front of the last two MFMAs of the step and the next step's weight reads behind it
//   variant 2: variant 1 + NF filler VALU instructions behind every MFMA (a deferred epilogue's issue load)
// build: hipcc --offload-arch=gfx950 -O3 probe_kloop.hip -o probe_kloop ; run: ./probe_kloop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OFF>
__device__ __forceinline__ void rd(half8& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF)); }
#define SB() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ f32x16 mma(half8 a, half8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

constexpr int PSTR = 96, PITCH = 18 * 96, TBYTES = 31744, WBYTES = 9216;

struct Frag { half8 a[3], b[2]; };

// ---- variant 0 --------------------------------------------------------------------------------------------------------------------
template <int I, int J, int KS>
__device__ __forceinline__ void reads_all(Frag& f, unsigned ap, unsigned bp) {
    constexpr int BOFF = I * PITCH + J * PSTR;
    rd<BOFF + KS * 32>(f.b[0], bp);
    rd<0 * 32 * PSTR + KS * 32>(f.a[0], ap);
    rd<BOFF + 2 * PITCH + KS * 32>(f.b[1], bp);
    rd<1 * 32 * PSTR + KS * 32>(f.a[1], ap);
    rd<2 * 32 * PSTR + KS * 32>(f.a[2], ap);
}
__device__ __forceinline__ void wait_all(Frag& f) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.b[0]), "+v"(f.b[1])); }
__device__ __forceinline__ void mmas(f32x16 (&acc)[3][2], const Frag& f) {
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) { acc[mt][0] = mma(f.a[mt], f.b[0], acc[mt][0]); acc[mt][1] = mma(f.a[mt], f.b[1], acc[mt][1]); }
}
template <int I, int J>
__device__ __forceinline__ void step_v0(f32x16 (&acc)[3][2], unsigned ap, unsigned bE, unsigned bO, bool barrier) {
    const unsigned bp = (I & 1) ? bO : bE;
    Frag f0, f1;
    reads_all<I, J, 0>(f0, ap, bp);
    wait_all(f0);
    reads_all<I, J, 1>(f1, ap, bp);
    SB(); mmas(acc, f0); SB();
    wait_all(f1);
    reads_all<I, J, 2>(f0, ap, bp);
    SB(); mmas(acc, f1); SB();
    wait_all(f0);
    mmas(acc, f0); SB();
    if (barrier) __builtin_amdgcn_s_barrier();
}

// ---- variant 1 / 2 ----------------------------------------------------------------------------------------------------------------
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;
struct Dma { u32x4_t rs; unsigned lds; int voff; int soff; };
// FORM 0: s_mov m0 + s_nop 4 + load (k_cdma);  1: s_mov m0 + s_nop 0 + load;  2: one m0 for all pieces of a gap, pieces by inst offset
template <int FORM, int OFF>
__device__ __forceinline__ void dma1(const Dma& d) {
    if (FORM == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(d.lds + OFF), "v"(d.voff), "s"(d.rs), "s"(d.soff + OFF) : "memory");
    else if (FORM == 1) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(d.lds + OFF), "v"(d.voff), "s"(d.rs), "s"(d.soff + OFF) : "memory");
    else asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" ::"v"(d.voff), "s"(d.rs), "s"(d.soff), "i"(OFF) : "memory");
}
template <int NF>
__device__ __forceinline__ void filler(float (&fz)[4]) {
#pragma unroll
    for (int i = 0; i < NF; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(fz[i & 3]));
}
#define W1(n, r0) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(r0))
#define W2(n, r0, r1) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(r0), "+v"(r1))
// one K-step on fragments `c` while the fragments of the next K-step (same tap: KS+1) are read into `n`; arrival order of c: B0,A0,B1,A1,A2
// (FIRST: B0,B1,A0,A1,A2 -- the order in which the previous step's tail issued them)
template <int I, int J, int KSN, bool FIRST, int NF>
__device__ __forceinline__ void kstep_mid(f32x16 (&acc)[3][2], Frag& c, Frag& n, unsigned ap, unsigned bp, float (&fz)[4]) {
    constexpr int BOFF = I * PITCH + J * PSTR;
    if (FIRST) W2(2, c.b[0], c.a[0]); else W2(3, c.b[0], c.a[0]);
    acc[0][0] = mma(c.a[0], c.b[0], acc[0][0]); SB();
    rd<BOFF + KSN * 32>(n.b[0], bp); filler<NF>(fz); SB();
    if (!FIRST) W1(3, c.b[1]);
    acc[0][1] = mma(c.a[0], c.b[1], acc[0][1]); SB();
    rd<0 * 32 * PSTR + KSN * 32>(n.a[0], ap); filler<NF>(fz); SB();
    W1(3, c.a[1]);
    acc[1][0] = mma(c.a[1], c.b[0], acc[1][0]); SB();
    rd<BOFF + 2 * PITCH + KSN * 32>(n.b[1], bp); filler<NF>(fz); SB();
    acc[1][1] = mma(c.a[1], c.b[1], acc[1][1]); SB();
    rd<1 * 32 * PSTR + KSN * 32>(n.a[1], ap); filler<NF>(fz); SB();
    W1(4, c.a[2]);
    acc[2][0] = mma(c.a[2], c.b[0], acc[2][0]); SB();
    rd<2 * 32 * PSTR + KSN * 32>(n.a[2], ap); filler<NF>(fz); SB();
    acc[2][1] = mma(c.a[2], c.b[1], acc[2][1]); SB();
    filler<NF>(fz); SB();
}
// last K-step of tap (I, J): the next step's (tap (IN, JN), weight slice at apn, K-step 0) fragments are read into `n`; the pixel fragments
// in front of the barrier (the tile does not change), the weight fragments behind it
template <int I, int J, int IN, int JN, int NF>
__device__ __forceinline__ void kstep_last(f32x16 (&acc)[3][2], Frag& c, Frag& n, unsigned apn, unsigned bpn, bool barrier, float (&fz)[4]) {
    constexpr int BOFFN = IN * PITCH + JN * PSTR;
    W2(3, c.b[0], c.a[0]);
    acc[0][0] = mma(c.a[0], c.b[0], acc[0][0]); SB();
    rd<BOFFN>(n.b[0], bpn); filler<NF>(fz); SB();
    W1(3, c.b[1]);
    acc[0][1] = mma(c.a[0], c.b[1], acc[0][1]); SB();
    rd<BOFFN + 2 * PITCH>(n.b[1], bpn); filler<NF>(fz); SB();
    W1(3, c.a[1]);
    acc[1][0] = mma(c.a[1], c.b[0], acc[1][0]); SB();
    filler<NF>(fz); SB();
    acc[1][1] = mma(c.a[1], c.b[1], acc[1][1]); SB();
    filler<NF>(fz); SB();
    W1(2, c.a[2]);                 // every read of THIS step's weight slice has completed
    if (barrier) __builtin_amdgcn_s_barrier();
    SB();
    acc[2][0] = mma(c.a[2], c.b[0], acc[2][0]); SB();
    rd<0>(n.a[0], apn); filler<NF>(fz); SB();
    acc[2][1] = mma(c.a[2], c.b[1], acc[2][1]); SB();
    rd<1 * 32 * PSTR>(n.a[1], apn); rd<2 * 32 * PSTR>(n.a[2], apn); filler<NF>(fz); SB();
}
// a tap step: K-steps 0, 1 on (c, n), (n, c), K-step 2 on c with the next step's first fragments into n.  Fragment sets swap per step.
template <int I, int J, int IN, int JN, int NF>
__device__ __forceinline__ void step_v1(f32x16 (&acc)[3][2], Frag& c, Frag& n, unsigned ap, unsigned apn, unsigned bE, unsigned bO, bool barrier, float (&fz)[4]) {
    const unsigned bp = (I & 1) ? bO : bE, bpn = (IN & 1) ? bO : bE;
    kstep_mid<I, J, 1, true, NF>(acc, c, n, ap, bp, fz);
    kstep_mid<I, J, 2, false, NF>(acc, n, c, ap, bp, fz);
    kstep_last<I, J, IN, JN, NF>(acc, c, n, apn, bpn, barrier, fz);
}

template <int VAR, int NF>
__global__ __launch_bounds__(256, 2) void k(float* out, unsigned long long* ticks, int nchunks, int barrier) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // LDS: two tile buffers + two weight buffers, small pseudo-random halves
    for (int i = tid; i < (2 * TBYTES + 2 * WBYTES) / 2; i += 256) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        reinterpret_cast<_Float16*>(smem)[i] = (_Float16)(((int)(h >> 20) & 255) - 128) * (_Float16)0.001f;
    }
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)smem, wlds0 = lds0 + 2 * TBYTES;
    const int xl = l31 & 15, tyl = 4 * w + (l31 >> 4), par = tyl & 1;
    const unsigned bE = lds0 + tyl * 1728 + xl * 96 + ((kh ^ par) << 4), bO = lds0 + tyl * 1728 + xl * 96 + ((kh ^ par ^ 1) << 4);
    const unsigned aB = l31 * 96 + ((kh ^ ((l31 >> 3) & 1)) << 4);
    f32x16 acc[3][2];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int j = 0; j < 16; ++j) { acc[mt][0][j] = 0.f; acc[mt][1][j] = 0.f; }
    float fz[4] = {0.5f, 0.25f, 0.125f, 0.75f};
    const unsigned a0 = wlds0 + aB, a1 = wlds0 + WBYTES + aB;       // weight buffers of even / odd steps
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (VAR == 0) {
        for (int c = 0; c < nchunks; ++c) {
            step_v0<0, 0>(acc, a0, bE, bO, barrier); step_v0<0, 1>(acc, a1, bE, bO, barrier); step_v0<0, 2>(acc, a0, bE, bO, barrier);
            step_v0<1, 0>(acc, a1, bE, bO, barrier); step_v0<1, 1>(acc, a0, bE, bO, barrier); step_v0<1, 2>(acc, a1, bE, bO, barrier);
            step_v0<2, 0>(acc, a0, bE, bO, barrier); step_v0<2, 1>(acc, a1, bE, bO, barrier); step_v0<2, 2>(acc, a0, bE, bO, barrier);
        }
    } else {
        Frag f0, f1;
        // prologue: first fragments (order B0, B1, A0, A1, A2)
        rd<0>(f0.b[0], bE); rd<2 * PITCH>(f0.b[1], bE); rd<0>(f0.a[0], a0); rd<32 * PSTR>(f0.a[1], a0); rd<64 * PSTR>(f0.a[2], a0);
        for (int c = 0; c < nchunks; ++c) {
            step_v1<0, 0, 0, 1, NF>(acc, f0, f1, a0, a1, bE, bO, barrier, fz); step_v1<0, 1, 0, 2, NF>(acc, f1, f0, a1, a0, bE, bO, barrier, fz);
            step_v1<0, 2, 1, 0, NF>(acc, f0, f1, a0, a1, bE, bO, barrier, fz); step_v1<1, 0, 1, 1, NF>(acc, f1, f0, a1, a0, bE, bO, barrier, fz);
            step_v1<1, 1, 1, 2, NF>(acc, f0, f1, a0, a1, bE, bO, barrier, fz); step_v1<1, 2, 2, 0, NF>(acc, f1, f0, a1, a0, bE, bO, barrier, fz);
            step_v1<2, 0, 2, 1, NF>(acc, f0, f1, a0, a1, bE, bO, barrier, fz); step_v1<2, 1, 2, 2, NF>(acc, f1, f0, a1, a0, bE, bO, barrier, fz);
            step_v1<2, 2, 0, 0, NF>(acc, f0, f1, a0, a1, bE, bO, barrier, fz);
            // nine steps: the prefetched fragments sit in f1; the next chunk starts on f0 again
            f0 = f1;
            SB();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f0.a[0]), "+v"(f0.a[1]), "+v"(f0.a[2]), "+v"(f0.b[0]), "+v"(f0.b[1]));
        acc[0][0][0] += (float)f0.a[0][0] + (float)f0.a[1][0] + (float)f0.a[2][0] + (float)f0.b[0][0] + (float)f0.b[1][0];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = fz[0] + fz[1] + fz[2] + fz[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[mt][0][j] + acc[mt][1][j];
    out[(size_t)blockIdx.x * 256 + tid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int VAR, int NF>
static void run(const char* name, int wg_per_cu, int barrier, float* out, unsigned long long* ticks) {
    const int cus = 256, nchunks = 400;
    const int lds = wg_per_cu == 1 ? 120 * 1024 : 80 * 1024;       // (1 per CU: too much LDS for a second workgroup)
    hipFuncSetAttribute((const void*)k<VAR, NF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<VAR, NF>), dim3(cus * wg_per_cu), dim3(256), lds, 0, out, ticks, nchunks, barrier);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> t(cus * wg_per_cu);
    hipMemcpy(t.data(), ticks, t.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : t) avg += (double)v;
    avg /= t.size();
    const double steps = nchunks * 9.0;
    const double flops = 2.0 * 96 * 256 * 48 * steps * cus * wg_per_cu;
    printf("%-44s wg/cu %d barrier %d : %8.1f us  %7.1f TF/s   %7.0f ticks/step/wave (s_memtime)  %6.1f ns/step\n", name, wg_per_cu, barrier, ms * 1e3,
           flops / (ms * 1e-3) / 1e12, avg / steps, ms * 1e6 / steps);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 512 * 256 * 4); hipMalloc(&ticks, 512 * 8);
    for (int wg = 1; wg <= 2; ++wg)
        for (int b = 0; b <= 1; ++b) {
            run<0, 0>("v0 drained per step (k_cdma round 4)", wg, b, out, ticks);
            run<1, 0>("v1 interleaved, continuous", wg, b, out, ticks);
            run<2, 1>("v2 = v1 + 1 VALU filler per MFMA", wg, b, out, ticks);
            run<2, 2>("v2 = v1 + 2 VALU fillers per MFMA", wg, b, out, ticks);
            run<2, 4>("v2 = v1 + 4 VALU fillers per MFMA", wg, b, out, ticks);
        }
    return 0;
}

// probe_dmacost.hip -- hardware probe (tuning aid): what does ONE LDS-DMA instruction cost a wave that is otherwise issuing MFMAs back to back
// (one wave per SIMD, as k_cpipe's compute waves)?  A stream of independent v_mfma_f32_32x32x16_f16 with a DMA behind every EVERY-th one.
//   form 0: s_mov m0 + s_nop 4 + buffer_load_dwordx4 ... lds   (k_cdma / k_cpipe round 5 start)
//   form 1: s_mov m0 + s_nop 0 + ...
//   form 2: m0 untouched, LDS destination and source advanced by the instruction's offset field
//   form 3: buffer_load_dwordx4 into VGPRs (no LDS)
//   form 4: no DMA (baseline)
//   form 6: __builtin_amdgcn_raw_buffer_load_lds (the compiler sets m0 and pads hazards itself)
//   form 5: as 0 but the wave-uniform operands come through v_readfirstlane (what the compiler emitted in k_cpipe)
// build: hipcc --offload-arch=gfx950 -O3 probe_dmacost.hip -o probe_dmacost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int FORM, int EVERY>
__global__ __launch_bounds__(256, 1) void k(const char* src, float* out, unsigned long long* ticks, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long bp = (unsigned long long)src;
    const u32x4_t rs = {(unsigned)bp, (unsigned)(bp >> 32) & 0xffffu, 0x80000000u, 0x00020000u};
    const unsigned lds = (unsigned)(size_t)smem + w * 16384;
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)0x80000000, 0x00020000);
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * ((lane * 7 + i * 3) % 17 - 8)); b[i] = (_Float16)(0.002f * ((lane * 5 + i) % 13 - 6)); }
    f32x16 acc[6];
    for (int j = 0; j < 6; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    const int voff = lane * 16;
    int vuni = (w * 4096 + 1024) ;     // a wave-uniform value the compiler keeps in a VGPR (form 5)
    asm volatile("v_mov_b32 %0, %1" : "=v"(vuni) : "s"(w * 4096 + 1024));
    unsigned soff = w * 36864;
    u32x4_t sink = {0, 0, 0, 0};
    if (FORM == 2) asm volatile("s_mov_b32 m0, %0" ::"s"(lds));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 18; ++m) {
            acc[m % 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % 6], 0, 0, 0); SB();
            if (EVERY > 0 && m % EVERY == 0 && FORM != 4) {
                const unsigned so = soff + (m & 7) * 1024;
                if (FORM == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds + (m & 7) * 1024), "v"(voff), "s"(rs), "s"(so) : "memory");
                else if (FORM == 1) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds + (m & 7) * 1024), "v"(voff), "s"(rs), "s"(so) : "memory");
                else if (FORM == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:1024 lds" ::"v"(voff), "s"(rs), "s"(so) : "memory");
                else if (FORM == 3) { u32x4_t v; asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rs), "s"(so) : "memory"); sink[0] ^= 0; (void)v; }
                else if (FORM == 6) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (__attribute__((address_space(3))) void*)(smem + w * 16384 + (m & 7) * 1024), 16, voff, so, 0, 0);
                }
                else if (FORM == 5) {
                    const unsigned l2 = __builtin_amdgcn_readfirstlane(lds + vuni + (m & 7) * 1024 - (w * 4096 + 1024));
                    const unsigned s2 = __builtin_amdgcn_readfirstlane(so + vuni - (w * 4096 + 1024));
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(l2), "v"(voff), "s"(rs), "s"(s2) : "memory");
                }
                SB();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        soff += 9216; if (soff >= 160000) soff -= 147456;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = (float)sink[0];
    for (int j = 0; j < 6; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[(size_t)blockIdx.x * 256 + tid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int FORM, int EVERY>
static void run(const char* name, const char* src, float* out, unsigned long long* ticks) {
    const int cus = 256, iters = 4000;
    (void)hipFuncSetAttribute((const void*)k<FORM, EVERY>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<FORM, EVERY>), dim3(cus), dim3(256), 120 * 1024, 0, src, out, ticks, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double nm = 18.0 * iters, nd = EVERY > 0 && FORM != 4 ? (double)((18 + EVERY - 1) / EVERY) * iters : 0;
    printf("%-56s : %7.2f ns per MFMA   (%5.2f DMA per 18 MFMAs)  extra per DMA vs baseline: see table\n", name, ms * 1e6 / nm, nd / iters);
}

int main() {
    char* src; float* out; unsigned long long* ticks;
    (void)hipMalloc(&src, 1 << 20); (void)hipMemset(src, 1, 1 << 20);
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&ticks, 256 * 8);
    run<4, 1>("no DMA", src, out, ticks);
    run<0, 6>("form 0 (m0 + s_nop 4), every 6th MFMA", src, out, ticks);
    run<0, 3>("form 0, every 3rd", src, out, ticks);
    run<0, 2>("form 0, every 2nd", src, out, ticks);
    run<0, 1>("form 0, every MFMA", src, out, ticks);
    run<1, 3>("form 1 (m0 + s_nop 0), every 3rd", src, out, ticks);
    run<1, 1>("form 1, every MFMA", src, out, ticks);
    run<2, 3>("form 2 (m0 fixed, inst offset), every 3rd", src, out, ticks);
    run<2, 1>("form 2, every MFMA", src, out, ticks);
    run<3, 3>("form 3 (to VGPRs), every 3rd", src, out, ticks);
    run<3, 1>("form 3, every MFMA", src, out, ticks);
    run<6, 3>("form 6 (compiler builtin raw_buffer_load_lds), every 3rd", src, out, ticks);
    run<6, 1>("form 6, every MFMA", src, out, ticks);
    run<5, 3>("form 5 (form 0 + readfirstlane operands), every 3rd", src, out, ticks);
    run<5, 1>("form 5, every MFMA", src, out, ticks);
    return 0;
}

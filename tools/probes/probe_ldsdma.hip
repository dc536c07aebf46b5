// probe_ldsdma.hip -- hardware probe (tuning aid, not part of the library): semantics of `buffer_load_dwordx4 ... lds` on gfx950
//   1. does M0 address LDS beyond 64 KiB (160 KiB LDS per CU)?
//   2. do EXEC-masked lanes write LDS?  3. do out-of-range lanes write zeros?
// build: hipcc --offload-arch=gfx950 -O2 probe_ldsdma.hip -o probe_ldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;

__global__ void k(const unsigned* src, unsigned* out, int lds_off, int mask_mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    // fill the whole LDS window [lds_off - 1024, lds_off + 2048) with a sentinel
    for (int i = tid; i < 3072 / 4; i += 64) reinterpret_cast<unsigned*>(smem + lds_off - 1024)[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned long long bp = (unsigned long long)src;
    const u32x4_t rs = {(unsigned)bp, (unsigned)(bp >> 32) & 0xffffu, 1024u, 0x00020000u};
    int voff = tid * 16;
    if (mask_mode == 2 && (tid & 1)) voff = (int)0x80000000;   // out of range lanes
    const unsigned ldsbase = (unsigned)(size_t)(smem + lds_off);
    if (mask_mode != 1 || (tid & 1) == 0) {   // mask_mode 1: odd lanes EXEC-masked
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(ldsbase), "v"(voff), "s"(rs) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 3072 / 4; i += 64) out[i] = reinterpret_cast<unsigned*>(smem + lds_off - 1024)[i];
}

int main() {
    std::vector<unsigned> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 0x1000 + i;
    unsigned *d_src, *d_out;
    hipMalloc(&d_src, 1024); hipMalloc(&d_out, 3072);
    hipMemcpy(d_src, h.data(), 1024, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int offs[] = {4096, 60000 / 16 * 16, 70000 / 16 * 16, 100000 / 16 * 16, 150000 / 16 * 16};
    for (int mode = 0; mode < 3; ++mode)
        for (int off : offs) {
            hipMemset(d_out, 0, 3072);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024, 0, d_src, d_out, off, mode);
            hipError_t e = hipDeviceSynchronize();
            std::vector<unsigned> o(768);
            hipMemcpy(o.data(), d_out, 3072, hipMemcpyDeviceToHost);
            int ok = 0, zero = 0, sent = 0, other = 0, stray = 0;
            for (int i = 0; i < 256; ++i) {
                unsigned v = o[256 + i];
                if (v == 0x1000u + i) ++ok; else if (v == 0) ++zero; else if (v == 0xdeadbeefu) ++sent; else ++other;
            }
            for (int i = 0; i < 256; ++i) { if (o[i] != 0xdeadbeefu) ++stray; if (o[512 + i] != 0xdeadbeefu) ++stray; }
            printf("mode %d lds_off %6d: %s  dwords ok %3d zero %3d untouched %3d other %3d  stray-outside %d\n", mode, off,
                   hipGetErrorString(e), ok, zero, sent, other, stray);
        }
    return 0;
}

// probe_dma_rate.hip -- hardware probe (tuning aid, not part of the library): how fast can a CU pull data into LDS / registers?
//   * LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per wave instruction) vs plain `buffer_load_dwordx4` into VGPRs,
//   * by the number of issuing waves per workgroup (1, 2, 4, 8) and workgroups per CU (1, 2),
//   * from an L2-resident source (the 166 KB weight tensor of a 96->96 3x3 layer, read by every CU) and from a streaming source
//     (each CU its own 2 MB), contiguous 1 KiB pieces or "tile rows" (96-byte segments at a 192-byte pitch, like k_cdma's halo rows).
// Answers the round-3 question "is k_cdma6's single weight-loader wave issue-bound?" (DESIGN.md section 3.1).
// build: hipcc --offload-arch=gfx950 -O2 probe_dma_rate.hip -o probe_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;

// MODE 0: LDS-DMA, 1: VGPR loads (consumed by a dummy xor).  PAT 0: contiguous 1 KiB pieces, 1: 96-byte segments at 192-byte pitch
template <int MODE, int PAT>
__global__ __launch_bounds__(512) void k(const char* src, unsigned* sink, int pieces_per_wave, unsigned span_bytes, unsigned cu_stride, int depth) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    const unsigned long long bp = (unsigned long long)(src + (size_t)blockIdx.x * cu_stride);
    const u32x4_t rs = {(unsigned)bp, (unsigned)(bp >> 32) & 0xffffu, 0x80000000u, 0x00020000u};
    const unsigned lds = (unsigned)(size_t)smem + w * 8192;
    int voff;
    if (PAT == 0) voff = lane * 16;
    else { const int px = lane / 6, c = lane - px * 6; voff = px * 192 + c * 16; }     // 10 pixels x 96 B (+4 lanes of an 11th)
    unsigned acc = 0;
    unsigned off = (unsigned)(w * 1024) % span_bytes;
    for (int i = 0; i < pieces_per_wave; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned soff = off;
            off += nw * 1024;
            if (off >= span_bytes) off -= span_bytes;
            if (MODE == 0) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds + j * 1024), "v"(voff), "s"(rs), "s"(soff) : "memory");
            } else {
                u32x4_t v;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rs), "s"(soff) : "memory");
                asm volatile("s_waitcnt vmcnt(7)" ::: "memory");      // (keeps <= 8 in flight; the register is consumed below)
                acc ^= v[0];
            }
        }
        if (MODE == 0) {
            if (depth == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 0) acc = reinterpret_cast<unsigned*>(smem)[tid];
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int PAT>
static void run(const char* name, const char* src, unsigned* sink, int nw, int wg_per_cu, unsigned span, unsigned cu_stride, int depth) {
    const int cus = 256, pieces = 4096;
    hipFuncSetAttribute((const void*)k<MODE, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, PAT>), dim3(cus * wg_per_cu), dim3(64 * nw), 80 * 1024 - (wg_per_cu == 1 ? 0 : 1024), 0, src, sink, pieces, span, cu_stride, depth);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)cus * wg_per_cu * nw * pieces * (PAT == 0 ? 1024 : 1024);
    printf("%-34s waves/WG %d WG/CU %d depth %d : %8.1f us  %7.2f TB/s chip  %6.1f GB/s per CU  %5.1f B/clk/CU@2.4GHz  (%.0f ns per 1 KiB piece per wave)\n", name, nw, wg_per_cu, depth,
           ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / cus, bytes / (ms * 1e-3) / cus / 2.4e9, ms * 1e6 / pieces);
}

int main() {
    char* src;
    unsigned* sink;
    const size_t total = (size_t)512 * 2 * 1024 * 1024;
    hipMalloc(&src, total + 65536);
    hipMemset(src, 1, total);
    hipMalloc(&sink, 64);
    const unsigned W = 165888;          // weights of a 96->96 3x3 layer, fp16
    for (int nw : {1, 2, 4, 8})
        for (int wg : {1, 2}) run<0, 0>("LDS-DMA contiguous, L2-resident", src, sink, nw, wg, W / 1024 * 1024, 0, 8);
    run<0, 0>("LDS-DMA contiguous, L2-resident", src, sink, 1, 2, W / 1024 * 1024, 0, 0);
    run<0, 0>("LDS-DMA contiguous, L2-resident", src, sink, 2, 2, W / 1024 * 1024, 0, 0);
    for (int nw : {1, 2, 4, 8}) run<1, 0>("VGPR loads contiguous, L2-resident", src, sink, nw, 2, W / 1024 * 1024, 0, 8);
    for (int nw : {1, 2, 4}) run<0, 1>("LDS-DMA tile rows, L2-resident", src, sink, nw, 2, W / 1024 * 1024, 0, 8);
    for (int nw : {1, 2, 4, 8}) run<0, 0>("LDS-DMA contiguous, streaming", src, sink, nw, 2, 2 * 1024 * 1024, 2 * 1024 * 1024, 8);
    for (int nw : {1, 4}) run<0, 1>("LDS-DMA tile rows, streaming", src, sink, nw, 2, 2 * 1024 * 1024, 2 * 1024 * 1024, 8);
    return 0;
}

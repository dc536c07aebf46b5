// common.h -- shared device helpers for the gfx950 kernels of libssdn_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "../../include/ssdn_hip.h"
#include <cstdlib>

// A/B and ablation knobs read the process environment ONLY in a `make TUNING=1` build (tools/*.sh build one); in the default
// build the lookup is a constant nullptr, so the library's behaviour never depends on the environment.
#if defined(SSDN_TUNING)
static inline const char* ssdn_tuning_env(const char* name) { return std::getenv(name); }
#else
static inline const char* ssdn_tuning_env(const char*) { return nullptr; }
#endif

typedef _Float16 h16;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LRELU_SLOPE 0.1f

// error plumbing (api.hip)
int ssdn_set_error(const char* fmt, ...);
#define SSDN_CHECK_HIP(expr)                                                                 \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) return ssdn_set_error("%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

static __device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : LRELU_SLOPE * v; }
static __device__ __forceinline__ float lrelu_grad(float act) { return act > 0.f ? 1.f : LRELU_SLOPE; }

static __device__ __forceinline__ half8 ld_h8(const h16* p) { return *reinterpret_cast<const half8*>(p); }
static __device__ __forceinline__ void st_h8(h16* p, half8 v) { *reinterpret_cast<half8*>(p) = v; }
static __device__ __forceinline__ half4 ld_h4(const h16* p) { return *reinterpret_cast<const half4*>(p); }
static __device__ __forceinline__ void st_h4(h16* p, half4 v) { *reinterpret_cast<half4*>(p) = v; }

// ---- bf16 (gradient tensors): stored as raw 16-bit words; round-to-nearest-even from fp32, NaN kept quiet ----
static __device__ __forceinline__ unsigned short f2bf(float f) {
    return __builtin_bit_cast(unsigned short, (__bf16)f);   // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
}
static __device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
// packed 16-bit pairs in one dword: round-to-nearest-even hardware conversions (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32)
typedef __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned u32x2_t;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;
typedef __attribute__((__vector_size__(2 * sizeof(float)))) float f32x2_t;
typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2_t;
typedef __attribute__((__vector_size__(2 * sizeof(_Float16)))) _Float16 f16x2_t;
static __device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    f32x2_t f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
static __device__ __forceinline__ unsigned pack_f16x2(float lo, float hi) {
    f32x2_t f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, f16x2_t));
}
static __device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
static __device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
static __device__ __forceinline__ float f16_lo(unsigned w) { return (float)__builtin_bit_cast(f16x2_t, w)[0]; }
static __device__ __forceinline__ float f16_hi(unsigned w) { return (float)__builtin_bit_cast(f16x2_t, w)[1]; }
#define SSDN_BUFFER_RSRC_FLAGS 0x00020000   // raw buffer, 32-bit data format (gfx90a / gfx94x / gfx950)
static __device__ __forceinline__ u16x8 ld_b8(const unsigned short* p) { return *reinterpret_cast<const u16x8*>(p); }
static __device__ __forceinline__ void st_b8(unsigned short* p, u16x8 v) { *reinterpret_cast<u16x8*>(p) = v; }
static __device__ __forceinline__ u16x4 ld_b4(const unsigned short* p) { return *reinterpret_cast<const u16x4*>(p); }
static __device__ __forceinline__ void st_b4(unsigned short* p, u16x4 v) { *reinterpret_cast<u16x4*>(p) = v; }
static __device__ __forceinline__ u16x8 zero_b8() {
    u16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = 0;
    return z;
}

static __device__ __forceinline__ half8 zero_h8() {
    half8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (h16)0.f;
    return z;
}

// Launch macro of the kernels whose completion another lane may wait for: when the executor has set a stop event for the op in flight
// (api.hip), the event rides on the kernel's own completion signal (hipExtLaunchKernelGGL) -- a separate hipEventRecord costs the
// producing stream a 5-14 us bubble behind every such kernel (a barrier packet the next dispatch queues behind).
extern thread_local hipEvent_t g_ssdn_stop_event;
extern thread_local bool g_ssdn_stop_used;
// The in-stream profiler's sample of a launch (prof_begin .. prof_end, api.hip) rides on the kernel's own dispatch the same way: start and
// stop event of hipExtLaunchKernelGGL are the dispatch's begin / end timestamps -- the kernel's duration as the rocprofv3 kernel trace
// reports it.  (Two hipEventRecord calls around the launch measured ~10 us more per sample: the records' own stream time.)
extern thread_local hipEvent_t g_ssdn_prof_start, g_ssdn_prof_stop;
extern thread_local bool g_ssdn_prof_used;
#define SSDN_LAUNCH(kernel, grid, block, lds, stream, ...)                                                                       \
    do {                                                                                                                         \
        if (g_ssdn_stop_event) {                                                                                                 \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, g_ssdn_stop_event, 0, __VA_ARGS__);                 \
            g_ssdn_stop_used = true;                                                                                             \
        } else if (g_ssdn_prof_start) {                                                                                          \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, g_ssdn_prof_start, g_ssdn_prof_stop, 0, __VA_ARGS__);        \
            g_ssdn_prof_start = nullptr;                                                                                         \
            g_ssdn_prof_used = true;                                                                                             \
        } else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                                \
    } while (0)

// slot of the current device in the per-device tables of the launchers (hipFuncSetAttribute is per device)
#define SSDN_MAX_DEVICES_ATTR 16
static inline int ssdn_current_device_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev % SSDN_MAX_DEVICES_ATTR;
}

// in-stream profiler (api.hip)
void prof_begin(int kind, hipStream_t s);
void prof_end(int kind, hipStream_t s, double flops, double bytes);

// launchers implemented in the individual .hip files; all return 0 / negative error
int launch_conv(const ssdn_conv_args* a, hipStream_t s);
int launch_wgrad(const ssdn_wgrad_args* a, hipStream_t s);
bool wgrad_mergeable(const ssdn_wgrad_args* a);                 // small layer with a k_wgrad_multi instance
int launch_wgrad_multi(const ssdn_wgrad_args* const* items, int n, hipStream_t s);
#define WGRAD_MEGA_MAX 64
int wgrad_mega_ok(const ssdn_wgrad_args* a);                    // mega > 0 and the op has a k_wgrad_mega instance
int launch_wgrad_mega(const ssdn_wgrad_args* const* ops, int n, hipStream_t s);   // a whole bucket's weight gradients, one workgroup per CU
int launch_pack_input(const ssdn_pack_input_args* a, hipStream_t s);
int launch_pool_fwd(const ssdn_pool_args* a, hipStream_t s);
int launch_pool_bwd(const ssdn_pool_args* a, hipStream_t s);
int launch_upsum_bwd(const ssdn_upsum_args* a, hipStream_t s);
int launch_unrot_fwd(const ssdn_unrot_args* a, hipStream_t s);
int launch_unrot_bwd(const ssdn_unrot_args* a, hipStream_t s);
int launch_wreduce(const ssdn_wreduce_args* a, hipStream_t s);
int launch_wpack(const ssdn_wpack_args* a, hipStream_t s);
#define WPACK_MULTI_MAX 24
int launch_wpack_multi(const ssdn_wpack_args* const* items, int n, hipStream_t s);
#define WREDUCE_MULTI_MAX 32
#define WGRAD_MULTI_MAX 32
int launch_wreduce_multi(const ssdn_wreduce_args* const* items, int n, hipStream_t s);
int launch_grad_pack(const ssdn_grad_pack_args* a, hipStream_t s);
int launch_head(const ssdn_head_args* a, hipStream_t s);
int launch_head_final(const ssdn_head_final_args* a, hipStream_t s);
int launch_spatial_mean(const ssdn_spatial_mean_args* a, hipStream_t s);
int launch_mse(const ssdn_mse_args* a, int masked, hipStream_t s);
int launch_adam(const ssdn_adam_args* a, hipStream_t s);
#define ADAM_PACK_MAX 24
int adam_pack_fusable(const ssdn_adam_args* a, const ssdn_wpack_args* const* items, int n);
int launch_adam_pack(const ssdn_adam_args* a, const ssdn_wpack_args* const* items, int n, hipStream_t s);
int launch_metrics(const ssdn_metrics_args* a, hipStream_t s);
int launch_noise(const ssdn_noise_args* a, hipStream_t s);
int conv_lds_bytes(const ssdn_conv_args* a);
int conv_validate(const ssdn_conv_args* a);                     // conv_mfma.hip: argument checks shared by every conv launcher
// conv_chain.hip: a run of consecutive main-lane ops on images of <= 64 pixels (3x3 forward layers; data gradients + SSDN_OP_POOL_BWD)
// as ONE launch, one workgroup per image, tensors resident in LDS
int chain_len(const ssdn_op* ops, int n, bool any_lane);         // ops of the prefix of ops[0..n) that run as one launch (0 or >= 2; < 0: error)
int launch_chain(const ssdn_op* ops, int n, bool any_lane, hipStream_t s);
bool chain_merging_on();                                         // ssdn_conv_set_chain: run merging (chains, folded input pack) enabled
// conv_dma.hip: persistent LDS-DMA convolution for the 3x3 layers that carry the flops
bool conv_dma_eligible(const ssdn_conv_args* a, bool any_size);
int conv_dma_lds_bytes(int mt);
bool conv_signs(const ssdn_conv_args* a);                      // conv_mfma.hip: the launch honours sign_out / mask_sign (k_gdma, M = 384)
bool gemm_dma_signs(const ssdn_conv_args* a);
bool conv_dma_signs(const ssdn_conv_args* a);                  // conv_dma.hip: k_cdma writes / reads LeakyReLU sign bytes for this launch
bool conv_fuses_urot(const ssdn_conv_args* a);                 // conv_mfma.hip: k_cdma stores un-rotated (fused UNROT_FWD)
bool conv_fuses_unrot(const ssdn_conv_args* a);                // conv_mfma.hip: k_gdma applies the fused UNROT_BWD
bool conv_fuses_upsum(const ssdn_conv_args* a);                // conv_mfma.hip: k_cdma or the flat path applies the fused UPSUM_BWD
bool conv_fuses_pool(const ssdn_conv_args* a);                  // conv_mfma.hip: the launch takes the flat path (fused max-pool)
int launch_conv_dma(const ssdn_conv_args* a, hipStream_t s);
// conv_thin.hip: forward 3x3 layer with 1..3 real input channels (encode_block_1.0) as an im2col-shaped GEMM
bool conv_thin_eligible(const ssdn_conv_args* a);
bool conv_thin_fuses_pack(const ssdn_pack_input_args* pk, const ssdn_conv_args* a);   // (the conv must still pass launch_conv's own routing)
int launch_conv_thin(const ssdn_conv_args* a, const ssdn_pack_input_args* pk, hipStream_t s);
bool conv_pack_fusable(const ssdn_pack_input_args* pk, const ssdn_conv_args* a);      // conv_mfma.hip: launch_conv would route `a` to k_conv_thin
// gradpack_dgrad.hip: SSDN_OP_GRAD_PACK + the data gradient of the narrow net_out layer behind it as one launch
bool gradpack_dgrad_fusable(const ssdn_grad_pack_args* gp, const ssdn_conv_args* a);
int launch_gradpack_dgrad(const ssdn_grad_pack_args* gp, const ssdn_conv_args* a, hipStream_t s);
bool conv_gradpack_fusable(const ssdn_grad_pack_args* gp, const ssdn_conv_args* a);   // conv_mfma.hip: ... and launch_conv would run `a` as plain k_conv
// gemm_dma.hip: 1x1 layers with 96 / 384 output channels as a one-pass LDS-DMA GEMM
bool gemm_dma_eligible(const ssdn_conv_args* a);
int gemm_dma_lds_bytes(const ssdn_conv_args* a);
int launch_gemm_dma(const ssdn_conv_args* a, hipStream_t s);
bool gemm_dma_fuses_next(const ssdn_conv_args* a, const ssdn_conv_args* b);   // the narrow 1x1 layer b behind the 96-channel 1x1 layer a: one launch
int launch_gemm_dma_with_next(const ssdn_conv_args* a, const ssdn_conv_args* b, hipStream_t s);
bool conv_pair_fusable(const ssdn_conv_args* a, const ssdn_conv_args* b);     // conv_mfma.hip: launch_conv would route `a` to k_gdma and b can ride along
extern "C" int ssdn_device_cus(void);
int wgrad_lds_bytes(const ssdn_wgrad_args* a);

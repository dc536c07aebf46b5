// head.hip -- per-pixel Gaussian posterior head + SSDN negative-log-likelihood, MSE and masked-MSE losses,
// forward AND hand-derived backward, fp32 (gfx950).  Replaces Denoiser._ssdn_pipeline / _mse_pipeline /
// _mask_mse_pipeline (/root/reference/ssdn/ssdn/denoiser.py:140-397, utils/n2v_loss.py:6-17) and their autograd graphs.
// The math (closed-form 3x3 SPD algebra and its derivative) is documented in DESIGN.md section "posterior head".
#include "common.h"

#define HB 256

static __device__ __forceinline__ float block_sum(float v, float* sh) {
    // 256 threads = 4 waves of 64
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
static __device__ __forceinline__ void atomic_max_abs(uint32_t* gmax, float v) {
    // |v| as uint is monotone in |v| for finite floats; one atomic per wave
    float a = fabsf(v);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a = fmaxf(a, __shfl_down(a, o, 64));
    if ((threadIdx.x & 63) == 0 && a > 0.f) atomicMax(gmax, __float_as_uint(a));
}
// ... one atomic per BLOCK (256 threads): the atomics of a launch all hit one address and serialise (~12 ns each); with one per wave a
// launch of one pixel per thread spent more time in them than in its arithmetic
static __device__ __forceinline__ void atomic_max_abs_block(uint32_t* gmax, float v, float* sh4) {
    float a = fabsf(v);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a = fmaxf(a, __shfl_down(a, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(sh4[0], sh4[1]), fmaxf(sh4[2], sh4[3]));
        if (m > 0.f) atomicMax(gmax, __float_as_uint(m));
    }
}
static __device__ __forceinline__ float softplus_m4(float raw) {
    // torch.nn.Softplus(beta=1, threshold=20) applied to (raw - 4), + 1e-3   (denoiser.py:274-275)
    float x = raw - 4.f;
    return (x > 20.f ? x : log1pf(expf(x))) + 1e-3f;
}
static __device__ __forceinline__ float sigmoid_m4(float raw) {
    float x = raw - 4.f;
    return x > 20.f ? 1.f : 1.f / (1.f + expf(-x));
}

__global__ void k_head(ssdn_head_args a) {
    __shared__ float sh[4];
    const int b = blockIdx.y;
    const long long HW = (long long)a.H * a.W;
    const int C = a.C;
    const int Cout = C + C * (C + 1) / 2;
    const float inv_total = 1.f / ((float)a.B * (float)HW);  // mean over pixels, then mean over the batch
    float est = 0.f, dest_draw = 0.f;
    if (a.mode != 0) {
        float raw = a.est_raw[a.mode == 2 ? b : 0];
        est = softplus_m4(raw);
        dest_draw = sigmoid_m4(raw);
    }
    const float npar = a.noise_param ? a.noise_param[b] : 0.f;
    float loss_acc = 0.f, gest_acc = 0.f, gabs = 0.f;
    const long long per = (HW + a.nchunks - 1) / a.nchunks;
    const long long p0 = (long long)blockIdx.x * per;
    const long long p1 = p0 + per < HW ? p0 + per : HW;
    const float* no = a.net_out + (long long)b * Cout * HW;
    const float* ny = a.noisy + (long long)b * C * HW;
    for (long long p = p0 + threadIdx.x; p < p1; p += HB) {
        if (C == 1) {
            float mu = no[p], av = no[HW + p], y = ny[p];
            float sig, dsig_dmu = 0.f, dsig_dest = 0.f;
            if (a.style == 0) {
                sig = a.mode == 0 ? fmaxf(npar, 1e-3f) : est;
                dsig_dest = 1.f;
            } else {
                float m = fmaxf(mu, 1e-3f);
                float f = a.mode == 0 ? 1.f / npar : est;
                sig = sqrtf(m * f);
                dsig_dmu = mu > 1e-3f ? 0.5f * f / sig : 0.f;
                dsig_dest = 0.5f * m / sig;
            }
            float sx = av * av, sn = sig * sig, sy = sx + sn;
            float d = y - mu;
            float l = d * d / sy + logf(sy);
            if (a.mode != 0) l -= 0.1f * sig;
            loss_acc += l;
            if (a.mu) a.mu[(long long)b * HW + p] = mu;
            if (a.pme) a.pme[(long long)b * HW + p] = (y * sx + mu * sn) / sy;
            if (a.model_std) a.model_std[(long long)b * HW + p] = fabsf(av);
            if (a.noise_std && a.style == 1) a.noise_std[(long long)b * HW + p] = sig;
            if (a.want_grad) {
                float dsy = -d * d / (sy * sy) + 1.f / sy;
                float dsig = 2.f * sig * dsy - (a.mode != 0 ? 0.1f : 0.f);
                float gmu = (-2.f * d / sy + dsig * dsig_dmu) * inv_total;
                float ga = 2.f * av * dsy * inv_total;
                a.g_net_out[(long long)b * 2 * HW + p] = gmu;
                a.g_net_out[(long long)b * 2 * HW + HW + p] = ga;
                gabs = fmaxf(gabs, fmaxf(fabsf(gmu), fabsf(ga)));
                gest_acc += dsig * dsig_dest;
            }
        } else {
            float mu[3], A[6], y[3], sig[3], dsig_dmu[3], dsig_dest[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { mu[c] = no[c * HW + p]; y[c] = ny[c * HW + p]; }
#pragma unroll
            for (int c = 0; c < 6; ++c) A[c] = no[(3 + c) * HW + p];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (a.style == 0) {
                    sig[c] = a.mode == 0 ? fmaxf(npar, 1e-3f) : est;
                    dsig_dmu[c] = 0.f;
                    dsig_dest[c] = 1.f;
                } else {
                    float m = fmaxf(mu[c], 1e-3f);
                    float f = a.mode == 0 ? 1.f / npar : est;
                    sig[c] = sqrtf(m * f);
                    dsig_dmu[c] = mu[c] > 1e-3f ? 0.5f * f / sig[c] : 0.f;
                    dsig_dest[c] = 0.5f * m / sig[c];
                }
            }
            // Sigma_x = U U^T, U = [[a0,a1,a2],[0,a3,a4],[0,0,a5]]   (denoiser.py:246-255)
            float x00 = A[0] * A[0] + A[1] * A[1] + A[2] * A[2];
            float x01 = A[1] * A[3] + A[2] * A[4];
            float x02 = A[2] * A[5];
            float x11 = A[3] * A[3] + A[4] * A[4];
            float x12 = A[4] * A[5];
            float x22 = A[5] * A[5];
            float n0 = sig[0] * sig[0], n1 = sig[1] * sig[1], n2 = sig[2] * sig[2];
            float s00 = x00 + n0, s01 = x01, s02 = x02, s11 = x11 + n1, s12 = x12, s22 = x22 + n2;
            // adjugate / determinant of the SPD 3x3 Sigma_y
            float c00 = s11 * s22 - s12 * s12, c01 = s02 * s12 - s01 * s22, c02 = s01 * s12 - s02 * s11;
            float c11 = s00 * s22 - s02 * s02, c12 = s01 * s02 - s00 * s12, c22 = s00 * s11 - s01 * s01;
            float det = s00 * c00 + s01 * c01 + s02 * c02;
            float rdet = 1.f / det;
            float i00 = c00 * rdet, i01 = c01 * rdet, i02 = c02 * rdet, i11 = c11 * rdet, i12 = c12 * rdet, i22 = c22 * rdet;
            float d0 = y[0] - mu[0], d1 = y[1] - mu[1], d2 = y[2] - mu[2];
            float q0 = i00 * d0 + i01 * d1 + i02 * d2;
            float q1 = i01 * d0 + i11 * d1 + i12 * d2;
            float q2 = i02 * d0 + i12 * d1 + i22 * d2;
            float quad = d0 * q0 + d1 * q1 + d2 * q2;
            float detc = fmaxf(det, 0.f);
            float l = 0.5f * logf(detc) + 0.5f * quad;
            if (a.mode != 0) l -= 0.1f * (sig[0] + sig[1] + sig[2]) * (1.f / 3.f);
            loss_acc += l;
            long long o3 = (long long)b * 3 * HW + p;
            if (a.mu) { a.mu[o3] = mu[0]; a.mu[o3 + HW] = mu[1]; a.mu[o3 + 2 * HW] = mu[2]; }
            if (a.pme) {
                // posterior mean (denoiser.py:366-372) in its algebraically equal, well-conditioned form
                //   (Sx'^-1 + Sn'^-1)^-1 (Sx'^-1 mu + Sn'^-1 y) = mu + Sx' (Sx' + Sn')^-1 (y - mu),  Sx' = Sx + eps I, Sn' = Sn + eps I
                const float e = 1e-6f;
                float t00 = s00 + 2 * e, t11 = s11 + 2 * e, t22 = s22 + 2 * e;
                float k00 = t11 * t22 - s12 * s12, k01 = s02 * s12 - s01 * t22, k02 = s01 * s12 - s02 * t11;
                float k11 = t00 * t22 - s02 * s02, k12 = s01 * s02 - t00 * s12, k22 = t00 * t11 - s01 * s01;
                float rd = 1.f / (t00 * k00 + s01 * k01 + s02 * k02);
                float r0 = (k00 * d0 + k01 * d1 + k02 * d2) * rd;
                float r1 = (k01 * d0 + k11 * d1 + k12 * d2) * rd;
                float r2 = (k02 * d0 + k12 * d1 + k22 * d2) * rd;
                a.pme[o3] = mu[0] + (x00 + e) * r0 + x01 * r1 + x02 * r2;
                a.pme[o3 + HW] = mu[1] + x01 * r0 + (x11 + e) * r1 + x12 * r2;
                a.pme[o3 + 2 * HW] = mu[2] + x02 * r0 + x12 * r1 + (x22 + e) * r2;
            }
            if (a.model_std) a.model_std[(long long)b * HW + p] = cbrtf(fabsf(A[0] * A[3] * A[5]));  // det(U U^T)^(1/6)
            if (a.noise_std && a.style == 1) a.noise_std[(long long)b * HW + p] = cbrtf(sig[0] * sig[1] * sig[2]);
            if (a.want_grad) {
                // G = dl/dSigma_y = 1/2 Sy^-1 [det>0] - 1/2 q q^T
                float hd = det > 0.f ? 0.5f : 0.f;
                float g00 = hd * i00 - 0.5f * q0 * q0, g01 = hd * i01 - 0.5f * q0 * q1, g02 = hd * i02 - 0.5f * q0 * q2;
                float g11 = hd * i11 - 0.5f * q1 * q1, g12 = hd * i12 - 0.5f * q1 * q2, g22 = hd * i22 - 0.5f * q2 * q2;
                float reg = a.mode != 0 ? 0.1f / 3.f : 0.f;
                float ds0 = 2.f * sig[0] * g00 - reg, ds1 = 2.f * sig[1] * g11 - reg, ds2 = 2.f * sig[2] * g22 - reg;
                float g[9];
                g[0] = -q0 + ds0 * dsig_dmu[0];
                g[1] = -q1 + ds1 * dsig_dmu[1];
                g[2] = -q2 + ds2 * dsig_dmu[2];
                // dl/dU = 2 G U on the upper triangle
                g[3] = 2.f * (g00 * A[0]);
                g[4] = 2.f * (g00 * A[1] + g01 * A[3]);
                g[5] = 2.f * (g00 * A[2] + g01 * A[4] + g02 * A[5]);
                g[6] = 2.f * (g01 * A[1] + g11 * A[3]);
                g[7] = 2.f * (g01 * A[2] + g11 * A[4] + g12 * A[5]);
                g[8] = 2.f * (g02 * A[2] + g12 * A[4] + g22 * A[5]);
#pragma unroll
                for (int c = 0; c < 9; ++c) {
                    float v = g[c] * inv_total;
                    a.g_net_out[((long long)b * 9 + c) * HW + p] = v;
                    gabs = fmaxf(gabs, fabsf(v));
                }
                gest_acc += ds0 * dsig_dest[0] + ds1 * dsig_dest[1] + ds2 * dsig_dest[2];
            }
        }
    }
    float ls = block_sum(loss_acc, sh);
    float gs = block_sum(gest_acc, sh);
    if (threadIdx.x == 0) {
        float* pp = a.partial + ((long long)b * a.nchunks + blockIdx.x) * 2;
        pp[0] = ls;
        pp[1] = gs * dest_draw * inv_total;
    }
    if (a.want_grad && a.gmax) atomic_max_abs_block(a.gmax, gabs, sh);
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.noise_std && a.style == 0)
        a.noise_std[b] = a.mode == 0 ? fmaxf(npar, 1e-3f) : est;
}
int launch_head(const ssdn_head_args* a, hipStream_t s) {
    if (a->C != 1 && a->C != 3) return ssdn_set_error("head: C must be 1 or 3 (denoiser.py:199)");
    if (a->nchunks < 1) return ssdn_set_error("head: nchunks < 1");
    hipLaunchKernelGGL(k_head, dim3(a->nchunks, a->B), dim3(HB), 0, s, *a);
    return 0;
}

__global__ void k_head_final(ssdn_head_final_args a) {
    // one thread per sample sums that sample's partials in index order (deterministic)
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    long long HW = (long long)a.H * a.W;
    if (b < a.B) {
        float l = 0.f, g = 0.f;
        for (int c = 0; c < a.nchunks; ++c) {
            l += a.partial[((long long)b * a.nchunks + c) * 2];
            g += a.partial[((long long)b * a.nchunks + c) * 2 + 1];
        }
        a.loss[b] = l / (float)HW;
        if (a.mode == 2 && a.g_est) a.g_est[b] = g;
    }
    if (a.mode == 1 && a.g_est && b == 0) {
        float g = 0.f;
        for (int bb = 0; bb < a.B; ++bb)
            for (int c = 0; c < a.nchunks; ++c) g += a.partial[((long long)bb * a.nchunks + c) * 2 + 1];
        a.g_est[0] = g;
    }
}
__global__ void k_fill_sigma_grad(ssdn_head_final_args a) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long HW = (long long)a.H * a.W;
    if (idx >= a.B * HW) return;
    int b = idx / HW;
    float v = a.g_est[b] / (float)HW;  // gradient of the spatial mean (denoiser.py:264)
    a.g_sigma_out[idx] = v;
    if ((idx % HW) == 0 && v != 0.f && a.gmax2) atomicMax(a.gmax2, __float_as_uint(fabsf(v)));
}
int launch_head_final(const ssdn_head_final_args* a, hipStream_t s) {
    hipLaunchKernelGGL(k_head_final, dim3((a->B + 63) / 64), dim3(64), 0, s, *a);
    if (a->mode == 2 && a->g_sigma_out) {
        long long n = (long long)a->B * a->H * a->W;
        hipLaunchKernelGGL(k_fill_sigma_grad, dim3((int)((n + 255) / 256)), dim3(256), 0, s, *a);
    }
    return 0;
}

__global__ void k_spatial_mean(ssdn_spatial_mean_args a) {
    __shared__ float sh[4];
    int b = blockIdx.x;
    float acc = 0.f;
    for (int i = threadIdx.x; i < a.HW; i += HB) acc += a.src[(long long)b * a.HW + i];
    float t = block_sum(acc, sh);
    if (threadIdx.x == 0) a.dst[b] = t / (float)a.HW;
}
int launch_spatial_mean(const ssdn_spatial_mean_args* a, hipStream_t s) {
    hipLaunchKernelGGL(k_spatial_mean, dim3(a->B), dim3(HB), 0, s, *a);
    return 0;
}

// MSE (denoiser.py:153-154): loss[b] = mean_{chw} (out-ref)^2 ; g = 2 (out-ref) / (CHW * B)
__global__ void k_mse(ssdn_mse_args a) {
    __shared__ float sh[4];
    int b = blockIdx.x;
    long long n = (long long)a.C * a.H * a.W;
    float acc = 0.f, gabs = 0.f;
    float k = 2.f / ((float)n * (float)a.B);
    for (long long i = threadIdx.x; i < n; i += HB) {
        float d = a.out[b * n + i] - a.ref[b * n + i];
        acc += d * d;
        if (a.g) {
            float g = k * d;
            a.g[b * n + i] = g;
            gabs = fmaxf(gabs, fabsf(g));
        }
    }
    float t = block_sum(acc, sh);
    if (threadIdx.x == 0) a.loss[b] = t / (float)n;
    if (a.g && a.gmax) atomic_max_abs(a.gmax, gabs);
}
// masked MSE (n2v_loss.py:6-17 + denoiser.py:175-176): coordinates of batch element 0 for every element, summed over
// coordinates, mean over channels.  Duplicate coordinates count twice (as in the reference's Python loop).
__global__ void k_mask_mse_zero(ssdn_mse_args a) {
    long long n = (long long)a.B * a.C * a.H * a.W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) a.g[i] = 0.f;
}
__global__ void k_mask_mse(ssdn_mse_args a) {
    __shared__ float sh[4];
    int b = blockIdx.x;
    long long HW = (long long)a.H * a.W;
    float acc = 0.f;
    float k = 2.f / ((float)a.C * (float)a.B);
    // sequential over coordinates per (b,c) so that duplicate coordinates accumulate deterministically
    for (int c = threadIdx.x; c < a.C; c += HB) {
        for (int q = 0; q < a.ncoords; ++q) {
            long long r = a.coords[2 * q], cc = a.coords[2 * q + 1];
            if (r < 0 || r >= a.H || cc < 0 || cc >= a.W) continue;   // (the host validates; never touch memory outside the image)
            long long off = ((long long)b * a.C + c) * HW + r * a.W + cc;
            float d = a.out[off] - a.ref[off];
            acc += d * d;
            if (a.g) a.g[off] += k * d;
        }
    }
    float t = block_sum(acc, sh);
    if (threadIdx.x == 0) a.loss[b] = t / (float)a.C;
    if (a.g && a.gmax) {
        float gabs = 0.f;
        for (int c = threadIdx.x; c < a.C; c += HB)
            for (int q = 0; q < a.ncoords; ++q) {
                long long r = a.coords[2 * q], cc = a.coords[2 * q + 1];
                if (r < 0 || r >= a.H || cc < 0 || cc >= a.W) continue;
                gabs = fmaxf(gabs, fabsf(a.g[((long long)b * a.C + c) * HW + r * a.W + cc]));
            }
        atomic_max_abs(a.gmax, gabs);
    }
}
int launch_mse(const ssdn_mse_args* a, int masked, hipStream_t s) {
    if (!masked) {
        hipLaunchKernelGGL(k_mse, dim3(a->B), dim3(HB), 0, s, *a);
    } else {
        if (a->g) hipLaunchKernelGGL(k_mask_mse_zero, dim3(256), dim3(256), 0, s, *a);
        hipLaunchKernelGGL(k_mask_mse, dim3(a->B), dim3(HB), 0, s, *a);
    }
    return 0;
}

// H11 (SSDN_OP_METRICS): one block per sample; the block that arrives last adds the per-sample values in a fixed order
#define MB 1024          // threads of a k_metrics block: 16 waves (a sample is C*H*W = 12288 elements at BASELINE sizes: 12 per thread)
// the four per-sample sums of a block together: wave shuffle trees, one LDS slot per (wave, sum), every thread adds the 16 waves' values in
// wave order (two barriers for all four instead of two each)
static __device__ __forceinline__ void metrics_block_sum4(float (&v)[4], float (*sh)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) sh[w][k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < MB / 64; ++j) t += sh[j][k];
        v[k] = t;
    }
}
__global__ __launch_bounds__(MB) void k_metrics(ssdn_metrics_args a) {
    __shared__ float sh[MB / 64][4];
    __shared__ int last;
    const int b = blockIdx.x;
    const int HW = a.H * a.W;
    const int e1 = a.ext ? a.ext[2 * b] : a.H, e2 = a.ext ? a.ext[2 * b + 1] : a.W;
    const bool crop = e1 < a.H || e2 < a.W;      // (block-uniform: the coordinates of an element are only needed for a cropped extent)
    float so = 0.f, sm = 0.f, ss = 0.f, sn = 0.f;
    const long long base = (long long)b * a.C * HW;
    // (the trainer's kernel trace showed this launch at 50 us per step: 256 threads per sample, one element -- three loads and two
    //  integer divisions -- at a time.  Now 1024 threads with six elements' loads in flight each)
    constexpr int UNR = 6;
    const int n = a.C * HW;
    for (int i0 = threadIdx.x; i0 < n; i0 += MB * UNR) {
        float c[UNR], o[UNR], m[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int i = i0 + u * MB;
            bool ok = i < n;
            if (crop) {
                const int p = i % HW, y = p / a.W, x = p - y * a.W;
                ok = ok && y < e1 && x < e2;
            }
            c[u] = ok ? a.clean[base + i] : 0.f;
            o[u] = (ok && a.out) ? a.out[base + i] : c[u];
            m[u] = (ok && a.mu) ? a.mu[base + i] : c[u];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const float d0 = o[u] - c[u], d1 = m[u] - c[u];
            so += d0 * d0;
            sm += d1 * d1;
        }
    }
    const bool npix = a.noise_std && a.noise_n == a.B * HW;
    for (int i0 = threadIdx.x; i0 < HW; i0 += MB * 4) {
        float v[4], w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * MB;
            v[u] = (a.model_std && i < HW) ? a.model_std[(long long)b * HW + i] : 0.f;
            w[u] = (npix && i < HW) ? a.noise_std[(long long)b * HW + i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { ss += v[u]; sn += w[u]; }
    }
    float sums[4] = {so, sm, ss, sn};
    metrics_block_sum4(sums, sh);
    so = sums[0]; sm = sums[1]; ss = sums[2]; sn = sums[3];
    if (threadIdx.x == 0) {
        const float cnt = (float)a.C * (float)e1 * (float)e2;
        float* q = a.per + 8 * b;
        q[0] = a.loss ? a.loss[b] : 0.f;
        q[1] = a.out ? -10.f * log10f(so / cnt) : 0.f;
        q[2] = a.mu ? -10.f * log10f(sm / cnt) : 0.f;
        q[3] = !a.noise_std ? 0.f : 255.f * (npix ? sn / (float)HW : a.noise_std[a.noise_n == a.B ? b : 0]);
        q[4] = a.model_std ? 255.f * ss / (float)HW : 0.f;
        __threadfence();
        const unsigned t = atomicAdd(reinterpret_cast<unsigned*>(a.acc + 15), 1u);
        last = t == (unsigned)a.B - 1;
    }
    __syncthreads();
    if (last && threadIdx.x < 64) {
        // the first wave of the last block: lane l fetches the five values of samples l, l + 64, ... (all loads independent -- one thread
        // per metric walking the samples was a chain of B dependent L2 round trips: 32 us at batch 32), then a fixed shuffle tree per metric
        __threadfence();
        const int l = threadIdx.x;
        float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j = l; j < a.B; j += 64) {
            float q[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) q[k] = __hip_atomic_load(a.per + 8 * j + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < 5; ++k) v[k] += q[k];
        }
        const float first3 = __hip_atomic_load(a.per + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (sample 0's value: the metric of a batch with ONE noise level)
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off);
        if (l == 0) {
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const bool on = k == 0 ? a.loss != nullptr : k == 1 ? a.out != nullptr : k == 2 ? a.mu != nullptr : k == 3 ? a.noise_std != nullptr : a.model_std != nullptr;
                if (!on) continue;
                const bool once = k == 3 && a.noise_n == 1;         // one noise level for the whole batch: a single sample of the metric
                a.acc[2 * k] += once ? first3 : v[k];
                a.acc[2 * k + 1] += once ? 1.f : (float)a.B;
            }
            *reinterpret_cast<unsigned*>(a.acc + 15) = 0u;
        }
    }
}
int launch_metrics(const ssdn_metrics_args* a, hipStream_t s) {
    if (!a->clean || !a->per || !a->acc) return ssdn_set_error("metrics: clean, per and acc must be given");
    if (a->B < 1 || a->C < 1 || a->H < 1 || a->W < 1) return ssdn_set_error("metrics: bad shape");
    if (a->noise_std && a->noise_n != 1 && a->noise_n != a->B && a->noise_n != a->B * a->H * a->W) return ssdn_set_error("metrics: noise_n must be 1, B or B*H*W");
    hipLaunchKernelGGL(k_metrics, dim3(a->B), dim3(MB), 0, s, *a);
    return 0;
}

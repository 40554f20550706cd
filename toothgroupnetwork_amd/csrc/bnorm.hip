// bnorm.hip -- training-mode BatchNorm1d over the ROWS of a (rows, C) matrix, optionally fused with the ReLU behind it: the
// normalisations of the Point-Transformer training path (blocks.py:37,40 transpose to (n, c, nsample) for nn.BatchNorm1d; the
// mirrors normalise the flattened (n * nsample, c) rows, point_transformer._mlp_rows), 124 of them per step of the tgnet_fps
// network, most on small deep-stage tensors where torch's five-odd launches per normalisation and direction (statistics,
// transform, counter, ReLU / zero-fill, reduce, element-wise) cost more than their bytes.
//
//   forward : (1) per-column sum and sum of squares in DOUBLE (one pass; LDS double atomics per block, one global double atomic per
//                 column and block); the last block to finish turns them into mean / 1/sqrt(var + eps), updates the running
//                 statistics exactly like nn.BatchNorm1d (momentum, unbiased variance) and leaves the workspace zeroed;
//             (2) y = [relu]((x - mean) * invstd * gamma + beta), 16 B per lane when C % 4 == 0.
//   backward: (3) per-column sum(g) and sum(g * xhat) with g = dy * (y > 0) under ReLU -- same reduction, the last block writes
//                 dbeta / dgamma; (4) dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)).
// A thread keeps ONE column: the grid is sized so that the total thread count is a multiple of C and element e of thread t sits in
// column t % C on every trip.
#include "tgn_common.h"

namespace tgn {

constexpr int kBnThreads = 256;
constexpr int kBnMaxC = 1024;

__host__ __device__ inline size_t bn_ws_bytes(int C) { return (size_t)2 * C * sizeof(double) + 16; }

template <bool BWD>
__global__ __launch_bounds__(kBnThreads) void bn_rows_reduce_kernel(
    long long total, long long rows, int C, const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ dy,
    const float *__restrict__ mean_in, const float *__restrict__ invstd_in, int relu, double *__restrict__ acc,
    unsigned *__restrict__ counter,
    // forward finalisation
    float eps, float momentum, float *__restrict__ running_mean, float *__restrict__ running_var,
    long long *__restrict__ num_batches_tracked, float *__restrict__ save_mean, float *__restrict__ save_invstd,
    // backward finalisation
    float *__restrict__ dgamma, float *__restrict__ dbeta) {
    __shared__ double lacc[2][kBnMaxC];
    __shared__ unsigned ticket_s;
    const int tid = threadIdx.x;
    for (int c = tid; c < C; c += kBnThreads) lacc[0][c] = lacc[1][c] = 0.0;
    __syncthreads();
    const long long T = (long long)gridDim.x * kBnThreads;        // a multiple of C
    const long long t0 = (long long)blockIdx.x * kBnThreads + tid;
    const int col = (int)(t0 % C);
    double s1 = 0.0, s2 = 0.0;
    float mu = 0.0f, is = 0.0f;
    if (BWD) {
        mu = mean_in[col];
        is = invstd_in[col];
    }
    long long e = t0;
    for (; e + 3 * T < total; e += 4 * T) {                        // four loads in flight
        float v[4], g[4], w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] = x[e + u * T];
            if (BWD) {
                g[u] = dy[e + u * T];
                w[u] = relu ? y[e + u * T] : 1.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (BWD) {
                const float gg = w[u] > 0.0f ? g[u] : 0.0f;
                s1 += (double)gg;
                s2 += (double)(gg * ((v[u] - mu) * is));
            } else {
                s1 += (double)v[u];
                s2 += (double)v[u] * (double)v[u];
            }
        }
    }
    for (; e < total; e += T) {
        const float v = x[e];
        if (BWD) {
            const float gg = (!relu || y[e] > 0.0f) ? dy[e] : 0.0f;
            s1 += (double)gg;
            s2 += (double)(gg * ((v - mu) * is));
        } else {
            s1 += (double)v;
            s2 += (double)v * (double)v;
        }
    }
    atomicAdd(&lacc[0][col], s1);
    atomicAdd(&lacc[1][col], s2);
    __syncthreads();
    for (int c = tid; c < C; c += kBnThreads) {
        atomicAdd(&acc[c], lacc[0][c]);
        atomicAdd(&acc[C + c], lacc[1][c]);
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) ticket_s = atomicAdd(counter, 1u);
    __syncthreads();
    if (ticket_s != gridDim.x - 1) return;
    // last block: every partial sum is in
    __threadfence();
    const double n = (double)rows;
    for (int c = tid; c < C; c += kBnThreads) {
        const double a1 = __hip_atomic_load(&acc[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double a2 = __hip_atomic_load(&acc[C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc[c] = 0.0;
        acc[C + c] = 0.0;
        if (BWD) {
            dbeta[c] = (float)a1;
            dgamma[c] = (float)a2;
        } else {
            const double mean = a1 / n;
            double var = a2 / n - mean * mean;                     // biased, what the batch is normalised with
            if (var < 0.0) var = 0.0;
            save_mean[c] = (float)mean;
            save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
            if (running_mean) {
                const double unbiased = var * (n / (n - 1.0));     // nn.BatchNorm1d keeps the unbiased estimate
                running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
                running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
            }
        }
    }
    if (tid == 0) {
        *counter = 0u;
        if (!BWD && num_batches_tracked) *num_batches_tracked += 1;
    }
}

// element-wise halves.  VEC: C % 4 == 0, 16 B per lane.
template <bool BWD, bool VEC>
__global__ __launch_bounds__(kBnThreads) void bn_rows_apply_kernel(long long total, long long rows, int C,
                                                                   const float *__restrict__ x, const float *__restrict__ y_in,
                                                                   const float *__restrict__ dy, const float *__restrict__ gamma,
                                                                   const float *__restrict__ beta, const float *__restrict__ mean,
                                                                   const float *__restrict__ invstd, const float *__restrict__ dgamma,
                                                                   const float *__restrict__ dbeta, int relu, float *__restrict__ out) {
    // forward: out = x * p0 + p1;  backward: out = p0 * (g - p1 - (x - p3) * p2)
    __shared__ float p0[kBnMaxC], p1[kBnMaxC], p2[kBnMaxC], p3[kBnMaxC];
    const float inv_n = 1.0f / (float)rows;
    for (int c = threadIdx.x; c < C; c += kBnThreads) {
        const float is = invstd[c], sc = is * gamma[c];
        if (BWD) {
            p0[c] = sc;
            p1[c] = dbeta[c] * inv_n;
            p2[c] = is * (dgamma[c] * inv_n);                      // xhat * mean(g xhat) = (x - mean) * p2
            p3[c] = mean[c];
        } else {
            p0[c] = sc;
            p1[c] = beta[c] - mean[c] * sc;
        }
    }
    __syncthreads();
    constexpr int W = VEC ? 4 : 1;
    const long long units = total / W;
    const long long T = (long long)gridDim.x * kBnThreads;
    for (long long u = (long long)blockIdx.x * kBnThreads + threadIdx.x; u < units; u += T) {
        const long long e = u * W;
        int c = (int)(e % C);
        float xv[W], gv[W], yv[W], ov[W];
        if (VEC) {
            *(float4 *)xv = *(const float4 *)(x + e);
            if (BWD) {
                *(float4 *)gv = *(const float4 *)(dy + e);
                if (relu) *(float4 *)yv = *(const float4 *)(y_in + e);
            }
        } else {
            xv[0] = x[e];
            if (BWD) {
                gv[0] = dy[e];
                if (relu) yv[0] = y_in[e];
            }
        }
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const int cc = c + i;                                  // (VEC: C % 4 == 0, so the four stay inside one row)
            if (BWD) {
                const float g = (!relu || yv[i] > 0.0f) ? gv[i] : 0.0f;
                ov[i] = p0[cc] * (g - p1[cc] - (xv[i] - p3[cc]) * p2[cc]);
            } else {
                const float v = xv[i] * p0[cc] + p1[cc];
                ov[i] = relu ? fmaxf(v, 0.0f) : v;
            }
        }
        if (VEC)
            *(float4 *)(out + e) = *(float4 *)ov;
        else
            out[e] = ov[0];
    }
}

static int bn_gcd(int a, int b) { return b ? bn_gcd(b, a % b) : a; }

// blocks of the reduction: enough to fill the chip on big inputs, a multiple of C / gcd(256, C) so that a thread keeps its column
static unsigned bn_reduce_blocks(long long total, int C) {
    const int m = C / bn_gcd(kBnThreads, C);
    long long b = (total + (long long)kBnThreads * 32 - 1) / ((long long)kBnThreads * 32);
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    b = (b + m - 1) / m * m;
    return (unsigned)b;
}

static bool bn_vec_ok(int C, const void *a, const void *b, const void *c, const void *d) {
    auto al = [](const void *p) { return p == nullptr || ((unsigned long long)p & 15ull) == 0ull; };
    return C % 4 == 0 && al(a) && al(b) && al(c) && al(d);
}

}  // namespace tgn

using namespace tgn;

TGN_API size_t tgn_bn_rows_workspace_bytes(int C) { return C > 0 && C <= kBnMaxC ? bn_ws_bytes(C) : 0; }

TGN_API int tgn_bn_rows_forward(long long rows, int C, const float *x, const float *gamma, const float *beta, float eps,
                                float momentum, float *running_mean, float *running_var, long long *num_batches_tracked,
                                int relu, float *y, float *save_mean, float *save_invstd, void *workspace, tgn_stream_t stream) {
    if (rows < 2 || C <= 0 || C > kBnMaxC) {
        set_error("tgn_bn_rows_forward: rows = %lld, C = %d (needs rows >= 2 and 1 <= C <= %d)", rows, C, kBnMaxC);
        return TGN_ERR_UNSUPPORTED;
    }
    if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || !workspace || (!running_mean != !running_var)) {
        set_error("tgn_bn_rows_forward: null pointer");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    hipStream_t st = (hipStream_t)stream;
    const long long total = rows * C;
    double *acc = (double *)workspace;
    unsigned *counter = (unsigned *)((char *)workspace + (size_t)2 * C * sizeof(double));
    hipLaunchKernelGGL((bn_rows_reduce_kernel<false>), dim3(bn_reduce_blocks(total, C)), dim3(kBnThreads), 0, st, total, rows, C, x,
                       (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, 0, acc, counter,
                       eps, momentum, running_mean, running_var, num_batches_tracked, save_mean, save_invstd, (float *)nullptr,
                       (float *)nullptr);
    if (int rc = check_launch("bn_rows_reduce_kernel")) return rc;
    const bool vec = bn_vec_ok(C, x, y, nullptr, nullptr);
    long long blocks = (total / (vec ? 4 : 1) + kBnThreads * 4 - 1) / (kBnThreads * 4);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    if (vec)
        hipLaunchKernelGGL((bn_rows_apply_kernel<false, true>), dim3((unsigned)blocks), dim3(kBnThreads), 0, st, total, rows, C, x,
                           (const float *)nullptr, (const float *)nullptr, gamma, beta, save_mean, save_invstd, (const float *)nullptr,
                           (const float *)nullptr, relu, y);
    else
        hipLaunchKernelGGL((bn_rows_apply_kernel<false, false>), dim3((unsigned)blocks), dim3(kBnThreads), 0, st, total, rows, C, x,
                           (const float *)nullptr, (const float *)nullptr, gamma, beta, save_mean, save_invstd, (const float *)nullptr,
                           (const float *)nullptr, relu, y);
    return check_launch("bn_rows_apply_kernel");
}

TGN_API int tgn_bn_rows_backward(long long rows, int C, const float *x, const float *y, const float *dy, const float *gamma,
                                 const float *save_mean, const float *save_invstd, int relu, float *dx, float *dgamma, float *dbeta,
                                 void *workspace, tgn_stream_t stream) {
    if (rows < 2 || C <= 0 || C > kBnMaxC) {
        set_error("tgn_bn_rows_backward: rows = %lld, C = %d (needs rows >= 2 and 1 <= C <= %d)", rows, C, kBnMaxC);
        return TGN_ERR_UNSUPPORTED;
    }
    if (!x || !dy || !gamma || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || !workspace || (relu && !y)) {
        set_error("tgn_bn_rows_backward: null pointer");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    hipStream_t st = (hipStream_t)stream;
    const long long total = rows * C;
    double *acc = (double *)workspace;
    unsigned *counter = (unsigned *)((char *)workspace + (size_t)2 * C * sizeof(double));
    hipLaunchKernelGGL((bn_rows_reduce_kernel<true>), dim3(bn_reduce_blocks(total, C)), dim3(kBnThreads), 0, st, total, rows, C, x, y,
                       dy, save_mean, save_invstd, relu, acc, counter, 0.0f, 0.0f, (float *)nullptr, (float *)nullptr,
                       (long long *)nullptr, (float *)nullptr, (float *)nullptr, dgamma, dbeta);
    if (int rc = check_launch("bn_rows_reduce_kernel")) return rc;
    const bool vec = bn_vec_ok(C, x, dy, relu ? y : nullptr, dx);
    long long blocks = (total / (vec ? 4 : 1) + kBnThreads * 4 - 1) / (kBnThreads * 4);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    if (vec)
        hipLaunchKernelGGL((bn_rows_apply_kernel<true, true>), dim3((unsigned)blocks), dim3(kBnThreads), 0, st, total, rows, C, x, y,
                           dy, gamma, (const float *)nullptr, save_mean, save_invstd, dgamma, dbeta, relu, dx);
    else
        hipLaunchKernelGGL((bn_rows_apply_kernel<true, false>), dim3((unsigned)blocks), dim3(kBnThreads), 0, st, total, rows, C, x, y,
                           dy, gamma, (const float *)nullptr, save_mean, save_invstd, dgamma, dbeta, relu, dx);
    return check_launch("bn_rows_apply_kernel");
}

// sa.hip -- fused set abstraction: the first shared-MLP layer of PointNetSetAbstraction[Msg]
// (pointnet2_utils.py:229-236 / 281-294) without the grouped (B,S,K,3+D) tensor, eval mode (BatchNorm folded).
//
//   y[b,s,k,:] = relu(bn(W * [x[idx]-c, f[idx]] + bias))            (reference, per (query, neighbour) row)
//
// Two forms, chosen by the launcher:
//   * direct (3+D <= 16: level 1 of a network): a wave owns one query; the K gathered rows are the A operand of
//     v_mfma_f32_32x32x2_f32, the folded weights sit in registers, the max over the K neighbours is taken on the
//     accumulators (tgn_sa_direct_max);
//   * commuted (wide rows): a 1x1 convolution commutes with the gather,
//         scale*(W*[x[idx]-c, f[idx]] + bias) + shift  =  A[idx] - Wxs*c + b2,     A = [f, x] * Wt   per POINT,
//     so the contraction runs over the N points once (tgn_sa_point_transform: an fp32-MFMA GEMM, S*K/N = 8x fewer
//     flops than per grouped row) and the per-query part is a gather-max of A rows (tgn_sa_gather_max) or the
//     gather-add-relu of gather.hip's tgn_sa_first_layer when more layers follow.
// fp32 MFMA (v_mfma_f32_32x32x2_f32) is an exact fp32 fma chain (MI355X_MICROARCH.md): results differ from the
// reference's BLAS only by summation order (tests: 1e-5 relative to the row magnitude).
#include "tgn_common.h"

namespace tgn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sa_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

// ------------------------------------------------------------------------------------------------------------
// Per-point transform: A[m, :] = [points[m, 0..D), xyz[m, 0..3)] * Wt      (M = B*N rows, Kc = D+3, C1 columns)
// Wt: (D+3, C1) row-major, rows ordered [features..., x, y, z], BatchNorm scale already folded into its columns.
// 128 x 128 output tile per 256-thread block, 2 x 2 MFMA tiles (32x32) per wave, K in steps of 16 through LDS
// (k-major tiles: a fragment read is 64 consecutive floats), next K-tile prefetched into registers during the MFMAs.
// MFMA operand layout (v_mfma_f32_32x32x2_f32): lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31];
// accumulator register r of lane l is C[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31].
// ------------------------------------------------------------------------------------------------------------
constexpr int kTM = 128, kTN = 128, kTK = 16, kPad = 4;

__global__ __launch_bounds__(256) void sa_point_transform_kernel(long long M, int D, int C1,
                                                                  const float *__restrict__ xyz,
                                                                  const float *__restrict__ points,
                                                                  const float *__restrict__ Wt,
                                                                  float *__restrict__ A) {
    __shared__ float As[kTK][kTM + kPad];
    __shared__ float Bs[kTK][kTN + kPad];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int Kc = D + 3;
    const long long row0 = (long long)blockIdx.y * kTM;
    const int col0 = blockIdx.x * kTN;
    // global -> register staging: A tile: thread (r = tid & 127, h = tid >> 7) holds 8 consecutive channels of row r;
    // B tile: thread (k = tid >> 4, n = (tid & 15) * 8) holds 8 consecutive columns of weight row k
    const int ar = tid & 127, ah = tid >> 7;
    const int bk = tid >> 4, bn = (tid & 15) * 8;
    const long long grow = row0 + ar;
    const bool row_ok = grow < M;
    const bool feat4 = (D & 3) == 0 && points != nullptr;
    const bool col4 = (C1 & 3) == 0;
    float ra[8], rb[8];
    auto fetch = [&](int k0) {
        const int c = k0 + ah * 8;   // first channel of this thread's 8
        if (row_ok && feat4 && c + 8 <= D) {
            const f32x4 u = *(const f32x4 *)(points + (size_t)grow * D + c);
            const f32x4 w = *(const f32x4 *)(points + (size_t)grow * D + c + 4);
            ra[0] = u[0]; ra[1] = u[1]; ra[2] = u[2]; ra[3] = u[3];
            ra[4] = w[0]; ra[5] = w[1]; ra[6] = w[2]; ra[7] = w[3];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ci = c + i;
                float val = 0.0f;
                if (row_ok) {
                    if (ci < D) val = points[(size_t)grow * D + ci];
                    else if (ci < Kc) val = xyz[(size_t)grow * 3 + (ci - D)];
                }
                ra[i] = val;
            }
        }
        const int kr = k0 + bk;
        const int n = col0 + bn;
        if (kr < Kc && col4 && n + 8 <= C1) {
            const f32x4 u = *(const f32x4 *)(Wt + (size_t)kr * C1 + n);
            const f32x4 w = *(const f32x4 *)(Wt + (size_t)kr * C1 + n + 4);
            rb[0] = u[0]; rb[1] = u[1]; rb[2] = u[2]; rb[3] = u[3];
            rb[4] = w[0]; rb[5] = w[1]; rb[6] = w[2]; rb[7] = w[3];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) rb[i] = (kr < Kc && n + i < C1) ? Wt[(size_t)kr * C1 + n + i] : 0.0f;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j2][r] = 0.0f;
    fetch(0);
    const int lo = lane & 31, hi = lane >> 5;
    for (int k0 = 0; k0 < Kc; k0 += kTK) {
        __syncthreads();   // the previous tile has been consumed
#pragma unroll
        for (int i = 0; i < 8; ++i) As[ah * 8 + i][ar] = ra[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) Bs[bk][bn + i] = rb[i];
        __syncthreads();
        if (k0 + kTK < Kc) fetch(k0 + kTK);   // in flight during the MFMAs below
#pragma unroll
        for (int s = 0; s < kTK / 2; ++s) {
            const float a0 = As[2 * s + hi][wm * 64 + lo], a1 = As[2 * s + hi][wm * 64 + 32 + lo];
            const float b0 = Bs[2 * s + hi][wn * 64 + lo], b1 = Bs[2 * s + hi][wn * 64 + 32 + lo];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
            const int col = col0 + wn * 64 + j2 * 32 + lo;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (row < M && col < C1) A[(size_t)row * C1 + col] = acc[i][j2][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------------------
// Gather-max: out[q, c] = relu( max_k A[b, idx[q,k], c] - (Wxs[0,c]*cx + Wxs[1,c]*cy + Wxs[2,c]*cz) + b2[c] )
// One wave per (query, 128-channel block): a wave-load fetches TWO neighbour rows (lanes 0-31: row k, lanes 32-63:
// row k+1; 16 B per lane = 128 channels each), eight loads in flight; the two halves are merged at the end.
// Work is ordered (scan, channel block, query): an XCD gathers from ONE 128-channel slab of ONE scan's A at a time
// (N * 512 B: 2 MB at N = 4096, against 8 MB for all 512 channels of a level-2 scan), which is what stays in its L2.
// ------------------------------------------------------------------------------------------------------------
template <typename IdxT>
__global__ __launch_bounds__(256) void sa_gather_max_kernel(int B, int N, int S, int K, int C1,
                                                             const float *__restrict__ A,
                                                             const float *__restrict__ new_xyz,
                                                             const float *__restrict__ Wxs,   // (3, C1), BN scale folded
                                                             const float *__restrict__ b2,    // (C1): shift + scale*bias
                                                             const IdxT *__restrict__ idx, int relu,
                                                             float *__restrict__ out, int *__restrict__ err) {
    const unsigned lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lo = lane & 31u, hi = lane >> 5;
    const unsigned cblocks = ((unsigned)C1 + 127u) >> 7;
    const long long per_scan = (long long)cblocks * S;
    const long long items = per_scan * B;
    const unsigned nb = gridDim.x;
    const unsigned lb = (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3);   // XCD-contiguous item ranges (speed only)
    for (long long it = (long long)lb * 4 + wv; it < items; it += (long long)nb * 4) {
        const int b = __builtin_amdgcn_readfirstlane((int)(it / per_scan));
        const unsigned r = (unsigned)(it - (long long)b * per_scan);
        const unsigned cb = __builtin_amdgcn_readfirstlane((int)(r / (unsigned)S));
        const long long q = (long long)b * S + (r - cb * (unsigned)S);
        const unsigned c = cb * 128u + lo * 4u;
        bool bad = false;
        unsigned roff = 0;   // lane k: byte offset of row idx[q,k] in the scan's A block
        for (unsigned k = lane; k < (unsigned)K; k += 64u) {   // K <= 64
            long long v = (long long)idx[q * K + k];
            if (v < 0) v += N;
            if (v < 0 || v >= N) {
                bad = true;
                v = 0;
            }
            roff = (unsigned)v * (unsigned)C1 * 4u;
        }
        if (err && cb == 0 && __any(bad) && lane == 0) atomicOr(err, 1);
        const __amdgpu_buffer_rsrc_t rs = sa_rsrc(A + (size_t)b * N * C1, (unsigned)N * (unsigned)C1 * 4u);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        const bool live = c < (unsigned)C1;
        for (int k0 = 0; k0 < K; k0 += 16) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int k = k0 + 2 * u + (int)hi;        // this half-wave's row
                k = k < K ? k : K - 1;               // (repeat the last row: harmless under max)
                const unsigned ro = (unsigned)__shfl((int)roff, k);
                v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, live ? ro + c * 4u : 0u, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                m[0] = fmaxf(m[0], v[u][0]);
                m[1] = fmaxf(m[1], v[u][1]);
                m[2] = fmaxf(m[2], v[u][2]);
                m[3] = fmaxf(m[3], v[u][3]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = fmaxf(m[i], __shfl_xor(m[i], 32));
        if (live && hi == 0) {
            const float cx = new_xyz[q * 3 + 0], cy = new_xyz[q * 3 + 1], cz = new_xyz[q * 3 + 2];
            const f32x4 w0 = *(const f32x4 *)(Wxs + c), w1 = *(const f32x4 *)(Wxs + C1 + c), w2 = *(const f32x4 *)(Wxs + 2 * C1 + c);
            const f32x4 bb = *(const f32x4 *)(b2 + c);
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float t = (m[i] - ((w0[i] * cx + w1[i] * cy) + w2[i] * cz)) + bb[i];
                o[i] = relu ? fmaxf(t, 0.0f) : t;
            }
            *(f32x4 *)(out + (size_t)q * C1 + c) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Direct form for narrow inputs (Kc = 3 + D <= 16, C1 a multiple of 32 up to 256, K <= 64): a wave owns a query.
//   out[q, c] = relu( max_k ( [x_k - c_q, f_k] . Wd[:, c] ) + b2[c] ),   Wd: (Kc rounded up to even, C1), rows ordered
//   [x, y, z, f0..f_{D-1}] (+ a zero row), BN scale folded.
// The 32 neighbours of an M-tile are the rows of the MFMA A operand: lane l supplies channel 2s + (l >> 5) of
// neighbour l & 31 in k-step s; the weights of the wave's column tiles live in registers for the whole kernel.
// ------------------------------------------------------------------------------------------------------------
template <typename IdxT, int KS, int NT>   // KS k-steps of 2 channels, NT column tiles of 32
__global__ __launch_bounds__(256) void sa_direct_max_kernel(long long queries, int N, int S, int K, int D, int C1,
                                                             const float *__restrict__ xyz,
                                                             const float *__restrict__ new_xyz,
                                                             const float *__restrict__ points,
                                                             const float *__restrict__ Wd,
                                                             const float *__restrict__ b2,
                                                             const IdxT *__restrict__ idx, int relu,
                                                             float *__restrict__ out, int *__restrict__ err) {
    const unsigned lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lo = lane & 31u, hi = lane >> 5;
    const int Kc = 3 + D;
    float w[KS][NT];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) w[s][t] = Wd[(size_t)(2 * s + hi) * C1 + t * 32 + lo];
    float bias[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bias[t] = b2[t * 32 + lo];
    const unsigned nb = gridDim.x;
    const unsigned lb = (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3);
    for (long long q = (long long)lb * 4 + wv; q < queries; q += (long long)nb * 4) {
        const int b = __builtin_amdgcn_readfirstlane((int)(q / S));
        const float cq[3] = {new_xyz[q * 3 + 0], new_xyz[q * 3 + 1], new_xyz[q * 3 + 2]};
        const float *__restrict__ sx = xyz + (size_t)b * N * 3;
        const float *__restrict__ sp = points + (size_t)b * N * D;
        float best[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) best[t] = -INFINITY;
        bool bad = false;
        for (int m0 = 0; m0 < K; m0 += 32) {
            // neighbour of this lane's row; rows past K repeat neighbour 0 (harmless under max)
            const int kk = m0 + (int)lo < K ? m0 + (int)lo : 0;
            long long v = (long long)idx[q * K + kk];
            if (v < 0) v += N;
            if (v < 0 || v >= N) {
                bad = true;
                v = 0;
            }
            f32x16 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int ch = 2 * s + (int)hi;   // channel of [x - c, f] this lane supplies
                float a = 0.0f;
                if (ch < 3) a = sx[(size_t)v * 3 + ch] - cq[ch];
                else if (ch < Kc) a = sp[(size_t)v * D + (ch - 3)];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w[s][t], acc[t], 0, 0, 0);
            }
            // accumulator register r of lane l = neighbour row (r&3) + 8*(r>>2) + 4*hi, column lo: max over the rows
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float mx = acc[t][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[t][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                best[t] = fmaxf(best[t], mx);
            }
        }
        if (err && __any(bad) && lane == 0) atomicOr(err, 1);
        if (hi == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float o = best[t] + bias[t];
                out[(size_t)q * C1 + t * 32 + lo] = relu ? fmaxf(o, 0.0f) : o;
            }
        }
    }
}

}  // namespace tgn

using namespace tgn;

TGN_API int tgn_sa_point_transform(long long M, int D, int C1, const float *xyz, const float *points, const float *Wt,
                                   float *A, tgn_stream_t stream) {
    if (M <= 0 || C1 <= 0) return TGN_OK;
    if (!xyz || !Wt || !A || (D > 0 && !points) || D < 0) {
        set_error("tgn_sa_point_transform: null pointer / negative width");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    const long long by = (M + kTM - 1) / kTM;
    if (by > 0x7FFFFFFFLL) {
        set_error("tgn_sa_point_transform: too many rows");
        return TGN_ERR_UNSUPPORTED;
    }
    const dim3 grid((unsigned)((C1 + kTN - 1) / kTN), (unsigned)by);
    hipLaunchKernelGGL(sa_point_transform_kernel, grid, dim3(256), 0, (hipStream_t)stream, M, D, C1, xyz, points, Wt, A);
    return check_launch("sa_point_transform_kernel");
}

TGN_API int tgn_sa_gather_max(int B, int N, int S, int K, int C1, const float *A, const float *new_xyz, const float *Wxs,
                              const float *b2, const void *idx, int idx_is_int64, int relu, float *out,
                              tgn_stream_t stream) {
    const long long queries = (long long)B * S;
    if (queries <= 0 || K <= 0 || C1 <= 0) return TGN_OK;
    if (!A || !new_xyz || !Wxs || !b2 || !idx || !out) {
        set_error("tgn_sa_gather_max: null pointer");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    if (K > 64 || (C1 & 3) || (long long)N * C1 >= (1LL << 30) || (((uintptr_t)A | (uintptr_t)out | (uintptr_t)Wxs | (uintptr_t)b2) & 15)) {
        set_error("tgn_sa_gather_max: needs nsample <= 64, C1 %% 4 == 0, N*C1 < 2^30, 16-byte aligned tensors");
        return TGN_ERR_UNSUPPORTED;
    }
    int *err = index_error_word((hipStream_t)stream);
    const long long items = queries * ((C1 + 127) / 128);
    if ((long long)S * ((C1 + 127) / 128) >= (1LL << 31)) {
        set_error("tgn_sa_gather_max: too many queries per scan");
        return TGN_ERR_UNSUPPORTED;
    }
    long long blocks = ((items + 3) / 4 + 7) / 8 * 8;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (idx_is_int64)
        hipLaunchKernelGGL((sa_gather_max_kernel<long long>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, B, N, S,
                           K, C1, A, new_xyz, Wxs, b2, (const long long *)idx, relu, out, err);
    else
        hipLaunchKernelGGL((sa_gather_max_kernel<int>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, B, N, S, K, C1,
                           A, new_xyz, Wxs, b2, (const int *)idx, relu, out, err);
    return check_launch("sa_gather_max_kernel");
}

// 1 if tgn_sa_direct_max takes this shape
TGN_API int tgn_sa_direct_supported(int K, int D, int C1) {
    return (D >= 0 && 3 + D <= 16 && C1 >= 32 && C1 <= 256 && C1 % 32 == 0 && K >= 1 && K <= 64) ? 1 : 0;
}

TGN_API int tgn_sa_direct_max(int B, int N, int S, int K, int D, int C1, const float *xyz, const float *new_xyz,
                              const float *points, const float *Wd, const float *b2, const void *idx, int idx_is_int64,
                              int relu, float *out, tgn_stream_t stream) {
    const long long queries = (long long)B * S;
    if (queries <= 0) return TGN_OK;
    if (!xyz || !new_xyz || !Wd || !b2 || !idx || !out || (D > 0 && !points)) {
        set_error("tgn_sa_direct_max: null pointer");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    if (!tgn_sa_direct_supported(K, D, C1)) {
        set_error("tgn_sa_direct_max: needs 3+D <= 16, C1 a multiple of 32 up to 256, nsample <= 64");
        return TGN_ERR_UNSUPPORTED;
    }
    int *err = index_error_word((hipStream_t)stream);
    long long blocks = ((queries + 3) / 4 + 7) / 8 * 8;
    if (blocks > 256 * 8) blocks = 256 * 8;
    const int ks = (3 + D + 1) / 2, nt = C1 / 32;
    const float *pts = points ? points : xyz;
    hipStream_t st = (hipStream_t)stream;
#define TGN_SA_DIRECT(IT, KS, NT)                                                                                         \
    hipLaunchKernelGGL((sa_direct_max_kernel<IT, KS, NT>), dim3((unsigned)blocks), dim3(256), 0, st, queries, N, S, K, D, C1, \
                       xyz, new_xyz, pts, Wd, b2, (const IT *)idx, relu, out, err)
#define TGN_SA_DIRECT_NT(IT, KS)                                                            \
    switch (nt) {                                                                           \
        case 1: TGN_SA_DIRECT(IT, KS, 1); break;                                            \
        case 2: TGN_SA_DIRECT(IT, KS, 2); break;                                            \
        case 3: TGN_SA_DIRECT(IT, KS, 3); break;                                            \
        case 4: TGN_SA_DIRECT(IT, KS, 4); break;                                            \
        case 5: TGN_SA_DIRECT(IT, KS, 5); break;                                            \
        case 6: TGN_SA_DIRECT(IT, KS, 6); break;                                            \
        case 7: TGN_SA_DIRECT(IT, KS, 7); break;                                            \
        default: TGN_SA_DIRECT(IT, KS, 8); break;                                           \
    }
#define TGN_SA_DIRECT_KS(IT)                                                                \
    if (ks <= 2) { TGN_SA_DIRECT_NT(IT, 2) }                                                \
    else if (ks <= 5) { TGN_SA_DIRECT_NT(IT, 5) }                                           \
    else { TGN_SA_DIRECT_NT(IT, 8) }
    if (idx_is_int64) {
        TGN_SA_DIRECT_KS(long long)
    } else {
        TGN_SA_DIRECT_KS(int)
    }
#undef TGN_SA_DIRECT_KS
#undef TGN_SA_DIRECT_NT
#undef TGN_SA_DIRECT
    return check_launch("sa_direct_max_kernel");
}

// ------------------------------------------------------------------------------------------------------------
// First layer of a MULTI-layer shared MLP: out[b,s,k,c] = act( A[b, idx[b,s,k], c] - Wxs[:,c].centre + b2[c] )
// (B,S,K,C1): what the remaining layers consume.  One wave per query, 16 B per lane along the channels.
// ------------------------------------------------------------------------------------------------------------
namespace tgn {
template <typename IdxT>
__global__ __launch_bounds__(256) void sa_gather_act_kernel(long long queries, int N, int S, int K, int C1,
                                                             const float *__restrict__ A,
                                                             const float *__restrict__ new_xyz,
                                                             const float *__restrict__ Wxs, const float *__restrict__ b2,
                                                             const IdxT *__restrict__ idx, int relu,
                                                             float *__restrict__ out, int *__restrict__ err) {
    const unsigned lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nb = gridDim.x;
    const unsigned lb = (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3);
    for (long long q = (long long)lb * 4 + wv; q < queries; q += (long long)nb * 4) {
        const int b = __builtin_amdgcn_readfirstlane((int)(q / S));
        const float cx = new_xyz[q * 3 + 0], cy = new_xyz[q * 3 + 1], cz = new_xyz[q * 3 + 2];
        bool bad = false;
        unsigned roff = 0;
        for (unsigned k = lane; k < (unsigned)K; k += 64u) {   // K <= 64
            long long v = (long long)idx[q * K + k];
            if (v < 0) v += N;
            if (v < 0 || v >= N) {
                bad = true;
                v = 0;
            }
            roff = (unsigned)v * (unsigned)C1 * 4u;
        }
        if (err && __any(bad) && lane == 0) atomicOr(err, 1);
        const __amdgpu_buffer_rsrc_t rs = sa_rsrc(A + (size_t)b * N * C1, (unsigned)N * (unsigned)C1 * 4u);
        for (unsigned c = lane * 4u; c < (unsigned)C1; c += 256u) {
            const f32x4 w0 = *(const f32x4 *)(Wxs + c), w1 = *(const f32x4 *)(Wxs + C1 + c), w2 = *(const f32x4 *)(Wxs + 2 * C1 + c);
            const f32x4 bb = *(const f32x4 *)(b2 + c);
            f32x4 cst;
#pragma unroll
            for (int i = 0; i < 4; ++i) cst[i] = bb[i] - ((w0[i] * cx + w1[i] * cy) + w2[i] * cz);
            for (int k0 = 0; k0 < K; k0 += 4) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + u < K ? k0 + u : K - 1;
                    v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                         rs, c * 4u, (unsigned)__builtin_amdgcn_readlane((int)roff, k), 0));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (k0 + u >= K) break;
                    f32x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float t = v[u][i] + cst[i];
                        o[i] = relu ? fmaxf(t, 0.0f) : t;
                    }
                    *(f32x4 *)(out + ((size_t)q * K + (k0 + u)) * C1 + c) = o;
                }
            }
        }
    }
}
}  // namespace tgn

TGN_API int tgn_sa_gather_act(int B, int N, int S, int K, int C1, const float *A, const float *new_xyz, const float *Wxs,
                              const float *b2, const void *idx, int idx_is_int64, int relu, float *out,
                              tgn_stream_t stream) {
    const long long queries = (long long)B * S;
    if (queries <= 0 || K <= 0 || C1 <= 0) return TGN_OK;
    if (!A || !new_xyz || !Wxs || !b2 || !idx || !out) {
        set_error("tgn_sa_gather_act: null pointer");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    if (K > 64 || (C1 & 3) || (long long)N * C1 >= (1LL << 30) || (((uintptr_t)A | (uintptr_t)out | (uintptr_t)Wxs | (uintptr_t)b2) & 15)) {
        set_error("tgn_sa_gather_act: needs nsample <= 64, C1 %% 4 == 0, N*C1 < 2^30, 16-byte aligned tensors");
        return TGN_ERR_UNSUPPORTED;
    }
    int *err = index_error_word((hipStream_t)stream);
    long long blocks = ((queries + 3) / 4 + 7) / 8 * 8;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (idx_is_int64)
        hipLaunchKernelGGL((sa_gather_act_kernel<long long>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, queries,
                           N, S, K, C1, A, new_xyz, Wxs, b2, (const long long *)idx, relu, out, err);
    else
        hipLaunchKernelGGL((sa_gather_act_kernel<int>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, queries, N, S,
                           K, C1, A, new_xyz, Wxs, b2, (const int *)idx, relu, out, err);
    return check_launch("sa_gather_act_kernel");
}

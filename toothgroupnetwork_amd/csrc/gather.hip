// gather.hip -- the gather / scatter family (HBM-bound kernels, gfx950).
//
//   grouping     fwd/bwd  grouping_cuda_kernel.cu:5-25
//   interpolation fwd/bwd interpolation_cuda_kernel.cu:5-33
//   subtraction  fwd/bwd  subtraction_cuda_kernel.cu:5-30
//   aggregation  fwd/bwd  aggregation_cuda_kernel.cu:5-39
//   group_points          sample_and_group (pointnet2_utils.py:162-169) / SetAbstractionMsg (:281-285)
//   gather_points         index_points (pointnet2_utils.py:44-61)
//   three_interpolate     PointNetFeaturePropagation (pointnet2_utils.py:337-340)
//
// The reference uses one thread per output ELEMENT with two integer divisions each; here a block is a
// (rows x channel-lanes) tile: the gather index is read once per row, lanes run along the contiguous
// channel axis (coalesced row reads and writes), and there is no per-element division.
#include "tgn_common.h"

#include <stdlib.h>

namespace tgn {

struct RowShape {
    int cx_log2;  // lanes along the channel axis = 1 << cx_log2 (<= 64)
    int rows_per_block;
    unsigned blocks;
};

static RowShape row_shape(long long rows, int c) {
    int l = 0;
    while ((1 << l) < c && l < 6) ++l;
    RowShape s;
    s.cx_log2 = l;
    s.rows_per_block = 256 >> l;
    long long blocks = (rows + s.rows_per_block - 1) / s.rows_per_block;
    const long long cap = 256LL * 32;  // grid-stride beyond 32 blocks per CU
    s.blocks = (unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
    return s;
}

#define TGN_ROW_LOOP(rows)                                                                          \
    const int cx = 1 << cx_log2;                                                                    \
    const int tx = threadIdx.x & (cx - 1);                                                          \
    const int ty = threadIdx.x >> cx_log2;                                                          \
    const int ry = blockDim.x >> cx_log2;                                                           \
    for (long long r = (long long)blockIdx.x * ry + ty; r < (rows); r += (long long)gridDim.x * ry)

// ---- grouping ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grouping_fwd_kernel(long long rows, int c, int cx_log2,
                                                            const float *__restrict__ input,
                                                            const int *__restrict__ idx, float *__restrict__ output) {
    TGN_ROW_LOOP(rows) {
        const float *src = input + (size_t)idx[r] * c;
        float *dst = output + (size_t)r * c;
        for (int ci = tx; ci < c; ci += cx) dst[ci] = src[ci];
    }
}

__global__ __launch_bounds__(256) void grouping_bwd_kernel(long long rows, int c, int cx_log2,
                                                            const float *__restrict__ grad_output,
                                                            const int *__restrict__ idx, float *__restrict__ grad_input) {
    TGN_ROW_LOOP(rows) {
        float *dst = grad_input + (size_t)idx[r] * c;
        const float *src = grad_output + (size_t)r * c;
        for (int ci = tx; ci < c; ci += cx) atomicAdd(dst + ci, src[ci]);
    }
}

// ---- interpolation (weighted gather-sum over k neighbours) --------------------------------------
__global__ __launch_bounds__(256) void interpolation_fwd_kernel(long long rows, int c, int k, int cx_log2,
                                                                 const float *__restrict__ input,
                                                                 const int *__restrict__ idx,
                                                                 const float *__restrict__ weight,
                                                                 float *__restrict__ output) {
    TGN_ROW_LOOP(rows) {
        float *dst = output + (size_t)r * c;
        for (int ci = tx; ci < c; ci += cx) {
            float acc = dst[ci];  // the reference accumulates into the (pre-zeroed) output
            for (int i = 0; i < k; ++i) acc += input[(size_t)idx[r * k + i] * c + ci] * weight[r * k + i];
            dst[ci] = acc;
        }
    }
}

__global__ __launch_bounds__(256) void interpolation_bwd_kernel(long long rows, int c, int k, int cx_log2,
                                                                 const float *__restrict__ grad_output,
                                                                 const int *__restrict__ idx,
                                                                 const float *__restrict__ weight,
                                                                 float *__restrict__ grad_input) {
    TGN_ROW_LOOP(rows) {
        const float *src = grad_output + (size_t)r * c;
        for (int i = 0; i < k; ++i) {
            float *dst = grad_input + (size_t)idx[r * k + i] * c;
            const float w = weight[r * k + i];
            for (int ci = tx; ci < c; ci += cx) atomicAdd(dst + ci, src[ci] * w);
        }
    }
}

// ---- subtraction --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void subtraction_fwd_kernel(long long rows, int nsample, int c, int cx_log2,
                                                               const float *__restrict__ input1,
                                                               const float *__restrict__ input2,
                                                               const int *__restrict__ idx, float *__restrict__ output) {
    TGN_ROW_LOOP(rows) {  // r = n_idx * nsample + j
        const float *a = input1 + (size_t)(r / nsample) * c;
        const float *b = input2 + (size_t)idx[r] * c;
        float *dst = output + (size_t)r * c;
        for (int ci = tx; ci < c; ci += cx) dst[ci] = a[ci] - b[ci];
    }
}

__global__ __launch_bounds__(256) void subtraction_bwd_kernel(long long rows, int nsample, int c, int cx_log2,
                                                               const int *__restrict__ idx,
                                                               const float *__restrict__ grad_output,
                                                               float *__restrict__ grad_input1,
                                                               float *__restrict__ grad_input2) {
    TGN_ROW_LOOP(rows) {
        float *g1 = grad_input1 + (size_t)(r / nsample) * c;
        float *g2 = grad_input2 + (size_t)idx[r] * c;
        const float *src = grad_output + (size_t)r * c;
        for (int ci = tx; ci < c; ci += cx) {
            const float g = src[ci];
            atomicAdd(g1 + ci, g);
            atomicAdd(g2 + ci, -g);
        }
    }
}

// ---- aggregation --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void aggregation_fwd_kernel(long long rows, int nsample, int c, int w_c,
                                                               int cx_log2, const float *__restrict__ input,
                                                               const float *__restrict__ position,
                                                               const float *__restrict__ weight,
                                                               const int *__restrict__ idx, float *__restrict__ output) {
    TGN_ROW_LOOP(rows) {  // r = n_idx
        float *dst = output + (size_t)r * c;
        for (int ci = tx; ci < c; ci += cx) {
            const int wci = ci % w_c;
            float acc = dst[ci];
            for (int j = 0; j < nsample; ++j) {
                const size_t ii = (size_t)r * nsample + j;
                acc += (input[(size_t)idx[ii] * c + ci] + position[ii * c + ci]) * weight[ii * w_c + wci];
            }
            dst[ci] = acc;
        }
    }
}

__global__ __launch_bounds__(256) void aggregation_bwd_kernel(long long rows, int nsample, int c, int w_c,
                                                               int cx_log2, const float *__restrict__ input,
                                                               const float *__restrict__ position,
                                                               const float *__restrict__ weight,
                                                               const int *__restrict__ idx,
                                                               const float *__restrict__ grad_output,
                                                               float *__restrict__ grad_input,
                                                               float *__restrict__ grad_position,
                                                               float *__restrict__ grad_weight) {
    TGN_ROW_LOOP(rows) {
        for (int ci = tx; ci < c; ci += cx) {
            const int wci = ci % w_c;
            const float go = grad_output[(size_t)r * c + ci];
            for (int j = 0; j < nsample; ++j) {
                const size_t ii = (size_t)r * nsample + j;
                const float w = weight[ii * w_c + wci];
                const size_t in_i = (size_t)idx[ii] * c + ci;
                atomicAdd(grad_input + in_i, go * w);
                grad_position[ii * c + ci] = go * w;
                atomicAdd(grad_weight + ii * w_c + wci, go * (input[in_i] + position[ii * c + ci]));
            }
        }
    }
}

// ---- the same eight kernels with 16-byte lanes --------------------------------------------------------
// Taken when c % 4 == 0 and every row base is 16-byte aligned (the shapes of the Point-Transformer stages: c = 32 ... 512,
// w_c = c / 8).  A lane owns FOUR consecutive channels (global_load/store_dwordx4), a group of cx = 2^cx_log2 lanes owns a row.
//   * forward gathers (grouping, subtraction) walk the OUTPUT rows, four rows per lane in flight;
//   * the reductions walk the POINTS: a lane group keeps the sum over a point's nsample neighbours (aggregation forward, in the
//     reference's j order: bit-identical) or over its rows (subtraction backward: grad_input1 is ONE read-modify-write per point
//     instead of nsample atomics; aggregation backward: grad_weight is summed over the c / w_c channels that share a weight
//     across the lanes and added once) -- every such element has exactly one owner, so no atomics; what the caller pre-filled is
//     still accumulated into, like the reference's atomicAdd onto a zeroed buffer;
//   * the scatters (grad of a gathered operand) stay hardware fp32 atomics: their targets are data dependent.
typedef float f4 __attribute__((ext_vector_type(4)));

struct Vec4Shape {
    int cx_log2;
    int groups_per_block;
    unsigned blocks;
};

static Vec4Shape vec4_shape(long long units, int c, int units_per_group_pass) {
    const int c4 = c >> 2;
    int l = 0;
    while ((1 << l) < c4 && l < 6) ++l;
    Vec4Shape s;
    s.cx_log2 = l;
    s.groups_per_block = 256 >> l;
    const long long per_block = (long long)s.groups_per_block * units_per_group_pass;
    long long blocks = (units + per_block - 1) / per_block;
    const long long cap = 256LL * 64;
    s.blocks = (unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
    return s;
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define TGN_V4_LANES                        \
    const int cx = 1 << cx_log2;            \
    const int tx = threadIdx.x & (cx - 1);  \
    const int ty = threadIdx.x >> cx_log2;  \
    const int ry = blockDim.x >> cx_log2;

__device__ __forceinline__ void atomic_add4(float *dst, f4 v) {
    atomicAdd(dst + 0, v.x);
    atomicAdd(dst + 1, v.y);
    atomicAdd(dst + 2, v.z);
    atomicAdd(dst + 3, v.w);
}

constexpr int kV4Rows = 4;  // output rows a lane keeps in flight

template <bool SUB>
__global__ __launch_bounds__(256) void gather_rows_v4_kernel(long long rows, int nsample, int c4, int cx_log2,
                                                              const f4 *__restrict__ input1,
                                                              const f4 *__restrict__ input2,
                                                              const int *__restrict__ idx, f4 *__restrict__ output) {
    // SUB = false: output[r] = input2[idx[r]]                    (grouping forward; input1 unused)
    // SUB = true : output[r] = input1[r / nsample] - input2[idx[r]]   (subtraction forward)
    TGN_V4_LANES
    const long long step = (long long)gridDim.x * kV4Rows * ry;
    for (long long r0 = (long long)blockIdx.x * kV4Rows * ry + ty; r0 < rows; r0 += step) {
        int id[kV4Rows];
        long long own[kV4Rows];
#pragma unroll
        for (int u = 0; u < kV4Rows; ++u) {
            const long long r = r0 + (long long)u * ry;
            id[u] = r < rows ? idx[r] : 0;
            own[u] = SUB ? (r < rows ? r / nsample : 0) : 0;
        }
        for (int ci = tx; ci < c4; ci += cx) {
            f4 v[kV4Rows], a[kV4Rows];
#pragma unroll
            for (int u = 0; u < kV4Rows; ++u) {
                v[u] = input2[(size_t)id[u] * c4 + ci];
                if (SUB) a[u] = input1[(size_t)own[u] * c4 + ci];
            }
#pragma unroll
            for (int u = 0; u < kV4Rows; ++u) {
                const long long r = r0 + (long long)u * ry;
                if (r < rows) output[(size_t)r * c4 + ci] = SUB ? a[u] - v[u] : v[u];
            }
        }
    }
}

__global__ __launch_bounds__(256) void grouping_bwd_v4_kernel(long long rows, int c4, int cx_log2,
                                                               const f4 *__restrict__ grad_output,
                                                               const int *__restrict__ idx, float *__restrict__ grad_input) {
    TGN_V4_LANES
    for (long long r = (long long)blockIdx.x * ry + ty; r < rows; r += (long long)gridDim.x * ry) {
        float *dst = grad_input + (size_t)idx[r] * c4 * 4;
        for (int ci = tx; ci < c4; ci += cx) atomic_add4(dst + ci * 4, grad_output[(size_t)r * c4 + ci]);
    }
}

__global__ __launch_bounds__(256) void interpolation_fwd_v4_kernel(long long rows, int c4, int k, int cx_log2,
                                                                    const f4 *__restrict__ input,
                                                                    const int *__restrict__ idx,
                                                                    const float *__restrict__ weight,
                                                                    f4 *__restrict__ output) {
    TGN_V4_LANES
    for (long long r = (long long)blockIdx.x * ry + ty; r < rows; r += (long long)gridDim.x * ry) {
        for (int ci = tx; ci < c4; ci += cx) {
            f4 acc = output[(size_t)r * c4 + ci];  // the reference accumulates into the (pre-zeroed) output
            for (int i = 0; i < k; ++i) acc = acc + input[(size_t)idx[r * k + i] * c4 + ci] * weight[r * k + i];
            output[(size_t)r * c4 + ci] = acc;
        }
    }
}

__global__ __launch_bounds__(256) void interpolation_bwd_v4_kernel(long long rows, int c4, int k, int cx_log2,
                                                                    const f4 *__restrict__ grad_output,
                                                                    const int *__restrict__ idx,
                                                                    const float *__restrict__ weight,
                                                                    float *__restrict__ grad_input) {
    TGN_V4_LANES
    for (long long r = (long long)blockIdx.x * ry + ty; r < rows; r += (long long)gridDim.x * ry) {
        for (int ci = tx; ci < c4; ci += cx) {
            const f4 g = grad_output[(size_t)r * c4 + ci];
            for (int i = 0; i < k; ++i)
                atomic_add4(grad_input + ((size_t)idx[r * k + i] * c4 + ci) * 4, g * weight[r * k + i]);
        }
    }
}

__global__ __launch_bounds__(256) void subtraction_bwd_v4_kernel(long long n, int nsample, int c4, int cx_log2,
                                                                  const int *__restrict__ idx,
                                                                  const f4 *__restrict__ grad_output,
                                                                  f4 *__restrict__ grad_input1,
                                                                  float *__restrict__ grad_input2) {
    TGN_V4_LANES
    for (long long p = (long long)blockIdx.x * ry + ty; p < n; p += (long long)gridDim.x * ry) {
        const int *__restrict__ ip = idx + p * nsample;
        for (int ci = tx; ci < c4; ci += cx) {
            const f4 *__restrict__ src = grad_output + (size_t)p * nsample * c4 + ci;
            f4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
            for (int j = 0; j < nsample; ++j) {
                const f4 g = src[(size_t)j * c4];
                s = s + g;
                atomic_add4(grad_input2 + ((size_t)ip[j] * c4 + ci) * 4, -g);
            }
            grad_input1[(size_t)p * c4 + ci] = grad_input1[(size_t)p * c4 + ci] + s;  // the point's only owner: no atomic
        }
    }
}

__global__ __launch_bounds__(256) void aggregation_fwd_v4_kernel(long long n, int nsample, int c4, int w4, int cx_log2,
                                                                  const f4 *__restrict__ input,
                                                                  const f4 *__restrict__ position,
                                                                  const f4 *__restrict__ weight,
                                                                  const int *__restrict__ idx, f4 *__restrict__ output) {
    // w4 = w_c / 4: channel quad ci reads weight quad ci % w4 (channels 4ci..4ci+3 -> weights (4ci % w_c)..+3)
    TGN_V4_LANES
    for (long long p = (long long)blockIdx.x * ry + ty; p < n; p += (long long)gridDim.x * ry) {
        const int *__restrict__ ip = idx + p * nsample;
        for (int ci = tx; ci < c4; ci += cx) {
            const f4 *__restrict__ pp = position + (size_t)p * nsample * c4 + ci;
            const f4 *__restrict__ wp = weight + (size_t)p * nsample * w4 + (ci % w4);
            f4 acc = output[(size_t)p * c4 + ci];
#pragma unroll 4
            for (int j = 0; j < nsample; ++j)   // the reference's order over j, (input + position) * weight unfused
                acc = acc + (input[(size_t)ip[j] * c4 + ci] + pp[(size_t)j * c4]) * wp[(size_t)j * w4];
            output[(size_t)p * c4 + ci] = acc;
        }
    }
}

__global__ __launch_bounds__(256) void aggregation_bwd_v4_kernel(long long n, int nsample, int w4, int cx_log2,
                                                                  const f4 *__restrict__ input,
                                                                  const f4 *__restrict__ position,
                                                                  const f4 *__restrict__ weight,
                                                                  const int *__restrict__ idx,
                                                                  const f4 *__restrict__ grad_output,
                                                                  float *__restrict__ grad_input,
                                                                  f4 *__restrict__ grad_position,
                                                                  f4 *__restrict__ grad_weight) {
    // one lane per channel quad (c4 == cx, a power of two <= 64; w4 a power of two dividing it): the lanes tx, tx + w4, tx + 2 w4 ...
    // share weight quad tx % w4 -- their products are summed by xor-shuffles over the lane bits >= log2(w4)
    TGN_V4_LANES
    const int c4 = cx;
    const int wq = tx & (w4 - 1);
    for (long long p = (long long)blockIdx.x * ry + ty; p < n; p += (long long)gridDim.x * ry) {
        const int *__restrict__ ip = idx + p * nsample;
        const f4 go = grad_output[(size_t)p * c4 + tx];
        for (int j = 0; j < nsample; ++j) {
            const size_t ii = (size_t)p * nsample + j;
            const f4 w = weight[ii * w4 + wq];
            const size_t in_i = (size_t)ip[j] * c4 + tx;
            const f4 gw = go * w;
            f4 t = go * (input[in_i] + position[ii * c4 + tx]);
            atomic_add4(grad_input + in_i * 4, gw);
            grad_position[ii * c4 + tx] = gw;
            for (int m = w4; m < cx; m <<= 1) {
                t.x += __shfl_xor(t.x, m, kWave);
                t.y += __shfl_xor(t.y, m, kWave);
                t.z += __shfl_xor(t.z, m, kWave);
                t.w += __shfl_xor(t.w, m, kWave);
            }
            if (tx < w4) grad_weight[ii * w4 + tx] = grad_weight[ii * w4 + tx] + t;  // one owner per weight quad
        }
    }
}

// ---- the two reducing backward kernels with one lane per CHANNEL ---------------------------------------------------------
// An fp32 atomic instruction whose lanes cover whole 128-byte lines (32 consecutive floats of one target row) is one request
// per line; the 16-byte-lane form above spreads a row over four instructions that each touch every fourth dword.  So the
// scatters keep dword lanes, and only the owner-side sums (grad_input1, grad_weight) change.
__global__ __launch_bounds__(256) void subtraction_bwd_own_kernel(long long n, int nsample, int c, int cx_log2,
                                                                   const int *__restrict__ idx,
                                                                   const float *__restrict__ grad_output,
                                                                   float *__restrict__ grad_input1,
                                                                   float *__restrict__ grad_input2) {
    TGN_V4_LANES
    for (long long p = (long long)blockIdx.x * ry + ty; p < n; p += (long long)gridDim.x * ry) {
        const int *__restrict__ ip = idx + p * nsample;
        for (int ci = tx; ci < c; ci += cx) {
            const float *__restrict__ src = grad_output + (size_t)p * nsample * c + ci;
            float s = 0.0f;
#pragma unroll 4
            for (int j = 0; j < nsample; ++j) {
                const float g = src[(size_t)j * c];
                s += g;
                atomicAdd(grad_input2 + (size_t)ip[j] * c + ci, -g);
            }
            grad_input1[(size_t)p * c + ci] += s;
        }
    }
}

__global__ __launch_bounds__(256) void aggregation_bwd_own_kernel(long long n, int nsample, int w_c, int cx_log2,
                                                                   const float *__restrict__ input,
                                                                   const float *__restrict__ position,
                                                                   const float *__restrict__ weight,
                                                                   const int *__restrict__ idx,
                                                                   const float *__restrict__ grad_output,
                                                                   float *__restrict__ grad_input,
                                                                   float *__restrict__ grad_position,
                                                                   float *__restrict__ grad_weight) {
    // c == cx lanes (a power of two <= 64), w_c a power of two dividing it: lanes tx, tx + w_c, ... share weight tx % w_c
    TGN_V4_LANES
    const int c = cx;
    const int wci = tx & (w_c - 1);
    for (long long p = (long long)blockIdx.x * ry + ty; p < n; p += (long long)gridDim.x * ry) {
        const int *__restrict__ ip = idx + p * nsample;
        const float go = grad_output[(size_t)p * c + tx];
        for (int j = 0; j < nsample; ++j) {
            const size_t ii = (size_t)p * nsample + j;
            const float w = weight[ii * w_c + wci];
            const size_t in_i = (size_t)ip[j] * c + tx;
            const float gw = go * w;
            float t = go * (input[in_i] + position[ii * c + tx]);
            atomicAdd(grad_input + in_i, gw);
            grad_position[ii * c + tx] = gw;
            for (int m = w_c; m < cx; m <<= 1) t += __shfl_xor(t, m, kWave);
            if (tx < w_c) grad_weight[ii * w_c + tx] += t;
        }
    }
}

// ---- pointnet2_utils composites (group_points lives in group.hip) ---------------------------------
constexpr int kGroupMaxK = 128;

// Set-abstraction first layer, fused.  A 1x1 convolution commutes with the gather:
//   W * [points[idx], xyz[idx] - c] + b  ==  (W_p*points + W_x*xyz)[idx] - W_x*c + b
// so the first shared-MLP layer is evaluated once per POINT (N rows, a plain GEMM done by the host side)
// instead of once per (query, neighbour) pair (S*K rows), and the grouped (B,S,K,3+D) tensor of
// pointnet2_utils.py:162-169 / 281-285 is never materialised.  With eval-mode BatchNorm folded in,
//   out[b,s,k,:] = relu(A[b, idx[b,s,k], :] + Cst[b,s,:])       (this kernel; same lane-contiguous walk as group_points)
// and for a single-layer MLP the max over k (pointnet2_utils.py:236/294) is taken in registers:
//   outmax[b,s,:] = relu(max_k A[b, idx[b,s,k], :] + Cst[b,s,:]).
template <typename IdxT, bool MAXK>
__global__ __launch_bounds__(256) void sa_first_layer_kernel(long long queries, int N, int S, int K, int C,
                                                              unsigned magicC, const float *__restrict__ A,
                                                              const float *__restrict__ Cst,
                                                              const IdxT *__restrict__ idx, int relu,
                                                              float *__restrict__ out, int *__restrict__ err) {
    __shared__ unsigned sfb[4][kGroupMaxK];
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const unsigned nb = gridDim.x;  // multiple of 8: XCD-aware order as in group_points_kernel
    const unsigned lb = (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3);
    for (long long q = (long long)lb * 4 + wv; q < queries; q += (long long)nb * 4) {
        const int b = (int)(q / S);
        const size_t pbase = (size_t)b * N;
        const IdxT *__restrict__ qidx = idx + q * K;
        bool bad = false;
        for (int k = lane; k < K; k += kWave) {
            long long v64 = (long long)qidx[k];
            if (v64 < 0) v64 += N;
            if (v64 < 0 || v64 >= N) {
                bad = true;
                v64 = 0;
            }
            sfb[wv][k] = ((unsigned)pbase + (unsigned)v64) * (unsigned)C;
        }
        if (err && __any(bad) && lane == 0) atomicOr(err, 1);
        const float *__restrict__ cst = Cst + (size_t)q * C;
        if constexpr (MAXK) {
            float *__restrict__ dst = out + (size_t)q * C;
            for (int c = lane; c < C; c += kWave) {
                float m = -INFINITY;
                for (int k = 0; k < K; ++k) m = fmaxf(m, A[sfb[wv][k] + (unsigned)c]);
                const float v = m + cst[c];
                dst[c] = relu ? fmaxf(v, 0.0f) : v;
            }
        } else {
            const int total = K * C;
            float *__restrict__ dst = out + (size_t)q * total;
#pragma unroll 2
            for (int e = lane; e < total; e += kWave) {
                const unsigned k = __umulhi((unsigned)e, magicC);
                const unsigned c = (unsigned)e - k * (unsigned)C;
                const float v = A[sfb[wv][k] + c] + cst[c];
                dst[e] = relu ? fmaxf(v, 0.0f) : v;
            }
        }
    }
}

template <typename IdxT>
__global__ __launch_bounds__(256) void gather_points_kernel(long long rows, int N, int M, int C, int cx_log2,
                                                             const float *__restrict__ points,
                                                             const IdxT *__restrict__ idx, float *__restrict__ out,
                                                             int *__restrict__ err) {
    TGN_ROW_LOOP(rows) {  // r = b*M + j
        const int b = (int)(r / M);
        long long k = (long long)idx[r];
        if (k < 0) k += N;  // torch's advanced indexing wraps negative indices (pointnet2_utils.py:56-60)
        float *dst = out + (size_t)r * C;
        if (k < 0 || k >= N) {  // the reference raises here: the row is zero-filled and the error word latched
            if (err && tx == 0) atomicOr(err, 1);
            for (int ci = tx; ci < C; ci += cx) dst[ci] = 0.0f;
            continue;
        }
        const float *src = points + ((size_t)b * N + k) * C;
        for (int ci = tx; ci < C; ci += cx) dst[ci] = src[ci];
    }
}

template <typename IdxT>
__global__ __launch_bounds__(256) void scatter_add_points_kernel(long long rows, int N, int M, int C, int cx_log2,
                                                                  const float *__restrict__ grad_out,
                                                                  const IdxT *__restrict__ idx,
                                                                  float *__restrict__ grad_points) {
    TGN_ROW_LOOP(rows) {
        const int b = (int)(r / M);
        long long k = (long long)idx[r];
        if (k < 0) k += N;
        if (k < 0 || k >= N) continue;
        float *dst = grad_points + ((size_t)b * N + k) * C;
        const float *src = grad_out + (size_t)r * C;
        for (int ci = tx; ci < C; ci += cx) atomicAdd(dst + ci, src[ci]);
    }
}

template <typename IdxT>
__global__ __launch_bounds__(256) void three_interpolate_kernel(long long rows, int N, int S, int C, int cx_log2,
                                                                 const float *__restrict__ points2,
                                                                 const float *__restrict__ dist,
                                                                 const IdxT *__restrict__ idx, float *__restrict__ out,
                                                                 float *__restrict__ weight) {
    TGN_ROW_LOOP(rows) {  // r = b*N + n
        const int b = (int)(r / N);
        // pointnet2_utils.py:337-339: 1/(d + 1e-8), normalised by the row sum
        const float r0 = 1.0f / (dist[r * 3 + 0] + 1e-8f);
        const float r1 = 1.0f / (dist[r * 3 + 1] + 1e-8f);
        const float r2 = 1.0f / (dist[r * 3 + 2] + 1e-8f);
        const float norm = (r0 + r1) + r2;
        const float w0 = r0 / norm, w1 = r1 / norm, w2 = r2 / norm;
        if (weight && tx == 0) {
            weight[r * 3 + 0] = w0;
            weight[r * 3 + 1] = w1;
            weight[r * 3 + 2] = w2;
        }
        const float *f0 = points2 + ((size_t)b * S + (long long)idx[r * 3 + 0]) * C;
        const float *f1 = points2 + ((size_t)b * S + (long long)idx[r * 3 + 1]) * C;
        const float *f2 = points2 + ((size_t)b * S + (long long)idx[r * 3 + 2]) * C;
        float *dst = out + (size_t)r * C;
        for (int ci = tx; ci < C; ci += cx)
            dst[ci] = ((f0[ci] * w0) + (f1[ci] * w1)) + (f2[ci] * w2);
    }
}

// three_interpolate with 16-byte lanes and the epilogue of the eval-mode feature propagation fused in
// (pointnet2_utils.py:337-347 with the first convolution commuted onto the coarse points): out = [relu](interp (+ add)).
// `add` may be `out` itself (every element is read and written by the same lane).
template <typename IdxT>
__global__ __launch_bounds__(256) void three_interpolate_v4_kernel(long long rows, int N, int S, int c4, int cx_log2,
                                                                    const f4 *__restrict__ points2,
                                                                    const float *__restrict__ dist,
                                                                    const IdxT *__restrict__ idx, const f4 *add, int relu,
                                                                    f4 *out, float *__restrict__ weight) {
    TGN_V4_LANES
    for (long long r = (long long)blockIdx.x * ry + ty; r < rows; r += (long long)gridDim.x * ry) {
        const int b = (int)(r / N);
        const float r0 = 1.0f / (dist[r * 3 + 0] + 1e-8f);
        const float r1 = 1.0f / (dist[r * 3 + 1] + 1e-8f);
        const float r2 = 1.0f / (dist[r * 3 + 2] + 1e-8f);
        const float norm = (r0 + r1) + r2;
        const float w0 = r0 / norm, w1 = r1 / norm, w2 = r2 / norm;
        if (weight && tx == 0) {
            weight[r * 3 + 0] = w0;
            weight[r * 3 + 1] = w1;
            weight[r * 3 + 2] = w2;
        }
        const f4 *f0 = points2 + ((size_t)b * S + (long long)idx[r * 3 + 0]) * c4;
        const f4 *f1 = points2 + ((size_t)b * S + (long long)idx[r * 3 + 1]) * c4;
        const f4 *f2 = points2 + ((size_t)b * S + (long long)idx[r * 3 + 2]) * c4;
        for (int ci = tx; ci < c4; ci += cx) {
            f4 v = ((f0[ci] * w0) + (f1[ci] * w1)) + (f2[ci] * w2);
            if (add) v = v + add[(size_t)r * c4 + ci];
            if (relu) {
                v.x = fmaxf(v.x, 0.0f);
                v.y = fmaxf(v.y, 0.0f);
                v.z = fmaxf(v.z, 0.0f);
                v.w = fmaxf(v.w, 0.0f);
            }
            out[(size_t)r * c4 + ci] = v;
        }
    }
}

template <typename IdxT>
__global__ __launch_bounds__(256) void three_interpolate_epilogue_kernel(long long rows, int N, int S, int C, int cx_log2,
                                                                          const float *__restrict__ points2,
                                                                          const float *__restrict__ dist,
                                                                          const IdxT *__restrict__ idx, const float *add,
                                                                          int relu, float *out, float *__restrict__ weight) {
    TGN_ROW_LOOP(rows) {   // dword lanes: channel counts that are no multiple of 4 / unaligned rows
        const int b = (int)(r / N);
        const float r0 = 1.0f / (dist[r * 3 + 0] + 1e-8f);
        const float r1 = 1.0f / (dist[r * 3 + 1] + 1e-8f);
        const float r2 = 1.0f / (dist[r * 3 + 2] + 1e-8f);
        const float norm = (r0 + r1) + r2;
        const float w0 = r0 / norm, w1 = r1 / norm, w2 = r2 / norm;
        if (weight && tx == 0) {
            weight[r * 3 + 0] = w0;
            weight[r * 3 + 1] = w1;
            weight[r * 3 + 2] = w2;
        }
        const float *f0 = points2 + ((size_t)b * S + (long long)idx[r * 3 + 0]) * C;
        const float *f1 = points2 + ((size_t)b * S + (long long)idx[r * 3 + 1]) * C;
        const float *f2 = points2 + ((size_t)b * S + (long long)idx[r * 3 + 2]) * C;
        for (int ci = tx; ci < C; ci += cx) {
            float v = ((f0[ci] * w0) + (f1[ci] * w1)) + (f2[ci] * w2);
            if (add) v = v + add[(size_t)r * C + ci];
            out[(size_t)r * C + ci] = relu ? fmaxf(v, 0.0f) : v;
        }
    }
}

}  // namespace tgn

using namespace tgn;

TGN_API int tgn_grouping_forward(int m, int nsample, int c, const float *input, const int *idx, float *output,
                                 tgn_stream_t stream) {
    const long long rows = (long long)m * nsample;
    if (rows <= 0 || c <= 0) return TGN_OK;
    if ((tuning(kTuneGatherV4) & 1) && c % 4 == 0 && aligned16(input) && aligned16(output)) {
        const Vec4Shape v = vec4_shape(rows, c, kV4Rows);
        hipLaunchKernelGGL(gather_rows_v4_kernel<false>, dim3(v.blocks), dim3(256), 0, (hipStream_t)stream, rows, nsample, c / 4,
                           v.cx_log2, (const f4 *)nullptr, (const f4 *)input, idx, (f4 *)output);
        return check_launch("grouping_fwd_v4_kernel");
    }
    const RowShape s = row_shape(rows, c);
    hipLaunchKernelGGL(grouping_fwd_kernel, dim3(s.blocks), dim3(256), 0, (hipStream_t)stream, rows, c, s.cx_log2, input,
                       idx, output);
    return check_launch("grouping_fwd_kernel");
}

TGN_API int tgn_grouping_backward(int m, int nsample, int c, const float *grad_output, const int *idx,
                                  float *grad_input, tgn_stream_t stream) {
    const long long rows = (long long)m * nsample;
    if (rows <= 0 || c <= 0) return TGN_OK;
    if ((tuning(kTuneGatherV4) & 2) && c % 4 == 0 && aligned16(grad_output)) {
        const Vec4Shape v = vec4_shape(rows, c, 1);
        hipLaunchKernelGGL(grouping_bwd_v4_kernel, dim3(v.blocks), dim3(256), 0, (hipStream_t)stream, rows, c / 4, v.cx_log2,
                           (const f4 *)grad_output, idx, grad_input);
        return check_launch("grouping_bwd_v4_kernel");
    }
    const RowShape s = row_shape(rows, c);
    hipLaunchKernelGGL(grouping_bwd_kernel, dim3(s.blocks), dim3(256), 0, (hipStream_t)stream, rows, c, s.cx_log2,
                       grad_output, idx, grad_input);
    return check_launch("grouping_bwd_kernel");
}

TGN_API int tgn_interpolation_forward(int n, int c, int k, const float *input, const int *idx, const float *weight,
                                      float *output, tgn_stream_t stream) {
    if (n <= 0 || c <= 0) return TGN_OK;
    if ((tuning(kTuneGatherV4) & 1) && c % 4 == 0 && aligned16(input) && aligned16(output)) {
        const Vec4Shape v = vec4_shape(n, c, 1);
        hipLaunchKernelGGL(interpolation_fwd_v4_kernel, dim3(v.blocks), dim3(256), 0, (hipStream_t)stream, (long long)n, c / 4, k,
                           v.cx_log2, (const f4 *)input, idx, weight, (f4 *)output);
        return check_launch("interpolation_fwd_v4_kernel");
    }
    const RowShape s = row_shape(n, c);
    hipLaunchKernelGGL(interpolation_fwd_kernel, dim3(s.blocks), dim3(256), 0, (hipStream_t)stream, (long long)n, c, k,
                       s.cx_log2, input, idx, weight, output);
    return check_launch("interpolation_fwd_kernel");
}

TGN_API int tgn_interpolation_backward(int n, int c, int k, const float *grad_output, const int *idx,
                                       const float *weight, float *grad_input, tgn_stream_t stream) {
    if (n <= 0 || c <= 0) return TGN_OK;
    if ((tuning(kTuneGatherV4) & 2) && c % 4 == 0 && aligned16(grad_output)) {
        const Vec4Shape v = vec4_shape(n, c, 1);
        hipLaunchKernelGGL(interpolation_bwd_v4_kernel, dim3(v.blocks), dim3(256), 0, (hipStream_t)stream, (long long)n, c / 4, k,
                           v.cx_log2, (const f4 *)grad_output, idx, weight, grad_input);
        return check_launch("interpolation_bwd_v4_kernel");
    }
    const RowShape s = row_shape(n, c);
    hipLaunchKernelGGL(interpolation_bwd_kernel, dim3(s.blocks), dim3(256), 0, (hipStream_t)stream, (long long)n, c, k,
                       s.cx_log2, grad_output, idx, weight, grad_input);
    return check_launch("interpolation_bwd_kernel");
}

TGN_API int tgn_subtraction_forward(int n, int nsample, int c, const float *input1, const float *input2,
                                    const int *idx, float *output, tgn_stream_t stream) {
    const long long rows = (long long)n * nsample;
    if (rows <= 0 || c <= 0) return TGN_OK;
    if ((tuning(kTuneGatherV4) & 1) && c % 4 == 0 && aligned16(input1) && aligned16(input2) && aligned16(output)) {
        const Vec4Shape v = vec4_shape(rows, c, kV4Rows);
        hipLaunchKernelGGL(gather_rows_v4_kernel<true>, dim3(v.blocks), dim3(256), 0, (hipStream_t)stream, rows, nsample, c / 4,
                           v.cx_log2, (const f4 *)input1, (const f4 *)input2, idx, (f4 *)output);
        return check_launch("subtraction_fwd_v4_kernel");
    }
    const RowShape s = row_shape(rows, c);
    hipLaunchKernelGGL(subtraction_fwd_kernel, dim3(s.blocks), dim3(256), 0, (hipStream_t)stream, rows, nsample, c,
                       s.cx_log2, input1, input2, idx, output);
    return check_launch("subtraction_fwd_kernel");
}

TGN_API int tgn_subtraction_backward(int n, int nsample, int c, const int *idx, const float *grad_output,
                                     float *grad_input1, float *grad_input2, tgn_stream_t stream) {
    const long long rows = (long long)n * nsample;
    if (rows <= 0 || c <= 0) return TGN_OK;
    if (tuning(kTuneGatherV4) & 4) {
        const RowShape s = row_shape(n, c);
        hipLaunchKernelGGL(subtraction_bwd_own_kernel, dim3(s.blocks), dim3(256), 0, (hipStream_t)stream, (long long)n, nsample, c,
                           s.cx_log2, idx, grad_output, grad_input1, grad_input2);
        return check_launch("subtraction_bwd_own_kernel");
    }
    if ((tuning(kTuneGatherV4) & 2) && c % 4 == 0 && aligned16(grad_output) && aligned16(grad_input1)) {
        const Vec4Shape v = vec4_shape(n, c, 1);
        hipLaunchKernelGGL(subtraction_bwd_v4_kernel, dim3(v.blocks), dim3(256), 0, (hipStream_t)stream, (long long)n, nsample,
                           c / 4, v.cx_log2, idx, (const f4 *)grad_output, (f4 *)grad_input1, grad_input2);
        return check_launch("subtraction_bwd_v4_kernel");
    }
    const RowShape s = row_shape(rows, c);
    hipLaunchKernelGGL(subtraction_bwd_kernel, dim3(s.blocks), dim3(256), 0, (hipStream_t)stream, rows, nsample, c,
                       s.cx_log2, idx, grad_output, grad_input1, grad_input2);
    return check_launch("subtraction_bwd_kernel");
}

TGN_API int tgn_aggregation_forward(int n, int nsample, int c, int w_c, const float *input, const float *position,
                                    const float *weight, const int *idx, float *output, tgn_stream_t stream) {
    if (n <= 0 || c <= 0) return TGN_OK;
    if (w_c <= 0) {
        set_error("tgn_aggregation_forward: w_c must be positive");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    if ((tuning(kTuneGatherV4) & 1) && c % 4 == 0 && w_c % 4 == 0 && aligned16(input) && aligned16(position) && aligned16(weight) && aligned16(output)) {
        const Vec4Shape v = vec4_shape(n, c, 1);
        hipLaunchKernelGGL(aggregation_fwd_v4_kernel, dim3(v.blocks), dim3(256), 0, (hipStream_t)stream, (long long)n, nsample,
                           c / 4, w_c / 4, v.cx_log2, (const f4 *)input, (const f4 *)position, (const f4 *)weight, idx,
                           (f4 *)output);
        return check_launch("aggregation_fwd_v4_kernel");
    }
    const RowShape s = row_shape(n, c);
    hipLaunchKernelGGL(aggregation_fwd_kernel, dim3(s.blocks), dim3(256), 0, (hipStream_t)stream, (long long)n, nsample,
                       c, w_c, s.cx_log2, input, position, weight, idx, output);
    return check_launch("aggregation_fwd_kernel");
}

TGN_API int tgn_aggregation_backward(int n, int nsample, int c, int w_c, const float *input, const float *position,
                                     const float *weight, const int *idx, const float *grad_output,
                                     float *grad_input, float *grad_position, float *grad_weight,
                                     tgn_stream_t stream) {
    if (n <= 0 || c <= 0) return TGN_OK;
    if (w_c <= 0) {
        set_error("tgn_aggregation_backward: w_c must be positive");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    if ((tuning(kTuneGatherV4) & 4) && c <= 64 && (c & (c - 1)) == 0 && (w_c & (w_c - 1)) == 0 && w_c <= c) {
        const RowShape s = row_shape(n, c);
        hipLaunchKernelGGL(aggregation_bwd_own_kernel, dim3(s.blocks), dim3(256), 0, (hipStream_t)stream, (long long)n, nsample, w_c,
                           s.cx_log2, input, position, weight, idx, grad_output, grad_input, grad_position, grad_weight);
        return check_launch("aggregation_bwd_own_kernel");
    }
    if (tuning(kTuneGatherV4) & 2) {
        const int c4 = c / 4, w4 = w_c / 4;
        const bool pow2 = c % 4 == 0 && w_c % 4 == 0 && c4 <= 64 && (c4 & (c4 - 1)) == 0 && (w4 & (w4 - 1)) == 0 && w4 <= c4;
        if (pow2 && aligned16(input) && aligned16(position) && aligned16(weight) && aligned16(grad_output) &&
            aligned16(grad_position) && aligned16(grad_weight)) {
            const Vec4Shape v = vec4_shape(n, c, 1);
            hipLaunchKernelGGL(aggregation_bwd_v4_kernel, dim3(v.blocks), dim3(256), 0, (hipStream_t)stream, (long long)n,
                               nsample, w4, v.cx_log2, (const f4 *)input, (const f4 *)position, (const f4 *)weight, idx,
                               (const f4 *)grad_output, grad_input, (f4 *)grad_position, (f4 *)grad_weight);
            return check_launch("aggregation_bwd_v4_kernel");
        }
    }
    const RowShape s = row_shape(n, c);
    hipLaunchKernelGGL(aggregation_bwd_kernel, dim3(s.blocks), dim3(256), 0, (hipStream_t)stream, (long long)n, nsample,
                       c, w_c, s.cx_log2, input, position, weight, idx, grad_output, grad_input, grad_position,
                       grad_weight);
    return check_launch("aggregation_bwd_kernel");
}

static int sa_first_layer_launch(int B, int N, int S, int K, int C, const float *A, const float *Cst, const void *idx,
                                 int idx_is_int64, int relu, float *out, bool maxk, hipStream_t st) {
    const long long queries = (long long)B * S;
    if (queries <= 0 || K <= 0 || C <= 0) return TGN_OK;
    if (!A || !Cst || !idx || !out) {
        set_error("tgn_sa_first_layer: null pointer");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    if (K > kGroupMaxK || (long long)K * C >= (1LL << 31) / C || (long long)B * N * C >= (1LL << 32)) {
        set_error("tgn_sa_first_layer: nsample %d / channels %d out of the supported range", K, C);
        return TGN_ERR_UNSUPPORTED;
    }
    int *err = index_error_word(st);
    if (C == 1) {  // e / C as umulhi(e, ceil(2^32 / C)) needs the magic to fit 32 bits
        set_error("tgn_sa_first_layer: a single output channel is not supported by the fused kernel");
        return TGN_ERR_UNSUPPORTED;
    }
    const unsigned magicC = (unsigned)((0x100000000ULL + C - 1) / C);
    const long long blocks = ((queries + 3) / 4 + 7) / 8 * 8;
#define TGN_SA_LAUNCH(IT, MK)                                                                                     \
    hipLaunchKernelGGL((sa_first_layer_kernel<IT, MK>), dim3((unsigned)blocks), dim3(256), 0, st, queries, N, S, K, C, \
                       magicC, A, Cst, (const IT *)idx, relu, out, err)
    if (idx_is_int64) {
        if (maxk) TGN_SA_LAUNCH(long long, true); else TGN_SA_LAUNCH(long long, false);
    } else {
        if (maxk) TGN_SA_LAUNCH(int, true); else TGN_SA_LAUNCH(int, false);
    }
#undef TGN_SA_LAUNCH
    return check_launch("sa_first_layer_kernel");
}

TGN_API int tgn_sa_first_layer(int B, int N, int S, int K, int C, const float *A, const float *Cst, const void *idx,
                               int idx_is_int64, int relu, float *out, tgn_stream_t stream) {
    return sa_first_layer_launch(B, N, S, K, C, A, Cst, idx, idx_is_int64, relu, out, false, (hipStream_t)stream);
}

TGN_API int tgn_sa_first_layer_max(int B, int N, int S, int K, int C, const float *A, const float *Cst, const void *idx,
                                   int idx_is_int64, int relu, float *out, tgn_stream_t stream) {
    return sa_first_layer_launch(B, N, S, K, C, A, Cst, idx, idx_is_int64, relu, out, true, (hipStream_t)stream);
}

TGN_API int tgn_gather_points(int B, int N, int M, int C, const float *points, const void *idx, int idx_is_int64,
                              float *out, tgn_stream_t stream) {
    const long long rows = (long long)B * M;
    if (rows <= 0 || C <= 0) return TGN_OK;
    int *err = index_error_word((hipStream_t)stream);
    const RowShape s = row_shape(rows, C);
    if (idx_is_int64)
        hipLaunchKernelGGL((gather_points_kernel<long long>), dim3(s.blocks), dim3(256), 0, (hipStream_t)stream, rows, N,
                           M, C, s.cx_log2, points, (const long long *)idx, out, err);
    else
        hipLaunchKernelGGL((gather_points_kernel<int>), dim3(s.blocks), dim3(256), 0, (hipStream_t)stream, rows, N, M, C,
                           s.cx_log2, points, (const int *)idx, out, err);
    return check_launch("gather_points_kernel");
}

TGN_API int tgn_scatter_add_points(int B, int N, int M, int C, const float *grad_out, const void *idx,
                                   int idx_is_int64, float *grad_points, tgn_stream_t stream) {
    const long long rows = (long long)B * M;
    if (rows <= 0 || C <= 0) return TGN_OK;
    const RowShape s = row_shape(rows, C);
    if (idx_is_int64)
        hipLaunchKernelGGL((scatter_add_points_kernel<long long>), dim3(s.blocks), dim3(256), 0, (hipStream_t)stream,
                           rows, N, M, C, s.cx_log2, grad_out, (const long long *)idx, grad_points);
    else
        hipLaunchKernelGGL((scatter_add_points_kernel<int>), dim3(s.blocks), dim3(256), 0, (hipStream_t)stream, rows, N,
                           M, C, s.cx_log2, grad_out, (const int *)idx, grad_points);
    return check_launch("scatter_add_points_kernel");
}

TGN_API int tgn_three_interpolate(int B, int N, int S, int C, const float *points2, const float *dist,
                                  const void *idx, int idx_is_int64, float *out, float *weight, tgn_stream_t stream) {
    const long long rows = (long long)B * N;
    if (rows <= 0 || C <= 0) return TGN_OK;
    const RowShape s = row_shape(rows, C);
    if (idx_is_int64)
        hipLaunchKernelGGL((three_interpolate_kernel<long long>), dim3(s.blocks), dim3(256), 0, (hipStream_t)stream, rows,
                           N, S, C, s.cx_log2, points2, dist, (const long long *)idx, out, weight);
    else
        hipLaunchKernelGGL((three_interpolate_kernel<int>), dim3(s.blocks), dim3(256), 0, (hipStream_t)stream, rows, N, S,
                           C, s.cx_log2, points2, dist, (const int *)idx, out, weight);
    return check_launch("three_interpolate_kernel");
}

TGN_API int tgn_three_interpolate_ex(int B, int N, int S, int C, const float *points2, const float *dist, const void *idx,
                                     int idx_is_int64, const float *add, int relu, float *out, float *weight,
                                     tgn_stream_t stream) {
    const long long rows = (long long)B * N;
    if (rows <= 0 || C <= 0) return TGN_OK;
    if (!points2 || !dist || !idx || !out) {
        set_error("tgn_three_interpolate_ex: null pointer");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    if (C % 4 == 0 && aligned16(points2) && aligned16(out) && (!add || aligned16(add))) {
        const Vec4Shape v = vec4_shape(rows, C, 1);
        if (idx_is_int64)
            hipLaunchKernelGGL((three_interpolate_v4_kernel<long long>), dim3(v.blocks), dim3(256), 0, (hipStream_t)stream, rows, N, S,
                               C / 4, v.cx_log2, (const f4 *)points2, dist, (const long long *)idx, (const f4 *)add, relu, (f4 *)out,
                               weight);
        else
            hipLaunchKernelGGL((three_interpolate_v4_kernel<int>), dim3(v.blocks), dim3(256), 0, (hipStream_t)stream, rows, N, S, C / 4,
                               v.cx_log2, (const f4 *)points2, dist, (const int *)idx, (const f4 *)add, relu, (f4 *)out, weight);
        return check_launch("three_interpolate_v4_kernel");
    }
    const RowShape s = row_shape(rows, C);
    if (idx_is_int64)
        hipLaunchKernelGGL((three_interpolate_epilogue_kernel<long long>), dim3(s.blocks), dim3(256), 0, (hipStream_t)stream, rows, N,
                           S, C, s.cx_log2, points2, dist, (const long long *)idx, add, relu, out, weight);
    else
        hipLaunchKernelGGL((three_interpolate_epilogue_kernel<int>), dim3(s.blocks), dim3(256), 0, (hipStream_t)stream, rows, N, S, C,
                           s.cx_log2, points2, dist, (const int *)idx, add, relu, out, weight);
    return check_launch("three_interpolate_epilogue_kernel");
}

// ---- reference ABI (default stream, void) ---------------------------------------------------------
TGN_API void grouping_forward_cuda_launcher(int m, int nsample, int c, const float *input, const int *idx,
                                            float *output) {
    (void)tgn_grouping_forward(m, nsample, c, input, idx, output, (tgn_stream_t)default_stream());
}
TGN_API void grouping_backward_cuda_launcher(int m, int nsample, int c, const float *grad_output, const int *idx,
                                             float *grad_input) {
    (void)tgn_grouping_backward(m, nsample, c, grad_output, idx, grad_input, (tgn_stream_t)default_stream());
}
TGN_API void interpolation_forward_cuda_launcher(int n, int c, int k, const float *input, const int *idx,
                                                 const float *weight, float *output) {
    (void)tgn_interpolation_forward(n, c, k, input, idx, weight, output, (tgn_stream_t)default_stream());
}
TGN_API void interpolation_backward_cuda_launcher(int n, int c, int k, const float *grad_output, const int *idx,
                                                  const float *weight, float *grad_input) {
    (void)tgn_interpolation_backward(n, c, k, grad_output, idx, weight, grad_input, (tgn_stream_t)default_stream());
}
TGN_API void subtraction_forward_cuda_launcher(int n, int nsample, int c, const float *input1, const float *input2,
                                               const int *idx, float *output) {
    (void)tgn_subtraction_forward(n, nsample, c, input1, input2, idx, output, (tgn_stream_t)default_stream());
}
TGN_API void subtraction_backward_cuda_launcher(int n, int nsample, int c, const int *idx, const float *grad_output,
                                                float *grad_input1, float *grad_input2) {
    (void)tgn_subtraction_backward(n, nsample, c, idx, grad_output, grad_input1, grad_input2,
                                   (tgn_stream_t)default_stream());
}
TGN_API void aggregation_forward_cuda_launcher(int n, int nsample, int c, int w_c, const float *input,
                                               const float *position, const float *weight, const int *idx,
                                               float *output) {
    (void)tgn_aggregation_forward(n, nsample, c, w_c, input, position, weight, idx, output,
                                  (tgn_stream_t)default_stream());
}
TGN_API void aggregation_backward_cuda_launcher(int n, int nsample, int c, int w_c, const float *input,
                                                const float *position, const float *weight, const int *idx,
                                                const float *grad_output, float *grad_input, float *grad_position,
                                                float *grad_weight) {
    (void)tgn_aggregation_backward(n, nsample, c, w_c, input, position, weight, idx, grad_output, grad_input,
                                   grad_position, grad_weight, (tgn_stream_t)default_stream());
}

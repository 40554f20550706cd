// linear.hip -- weight gradient of a TALL, NARROW linear layer (the training path of the Point-Transformer mirrors):
//
//   dW[o][i] = sum_r gy[r][o] * x[r][i],   db[o] = sum_r gy[r][o]        rows: 4 096 ... 864 000, widths 3 ... 256
//
// a contraction over the ROWS with a tiny (cout x cin) result.  rocBLAS runs it as one or two tiles walking the whole K (370 us for
// a 32 x 32 result over 24 000 rows); cut into slices and batched it picks 256 x 256 tiles for 32 x 32 outputs (94 us).  Here a
// wave owns a slice of rows and contracts it on v_mfma_f32_32x32x2_f32 straight from global memory -- the operand layout of that
// instruction (lane l: A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31]) is exactly a coalesced read of two consecutive rows of
// gy and of x, so there is no LDS, no barrier and no transpose -- and writes its (cout x cin) partial; the caller sums the partials.
// The kernel is bound by the one pass over gy and x.
#include "tgn_common.h"

namespace tgn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void linear_wgrad_partial_kernel(long long rows, int cin, int cout, int rows_per_slice,
                                                                    int tiles_i, int tiles, long long waves,
                                                                    const float *__restrict__ x, const float *__restrict__ gy,
                                                                    float *__restrict__ part, float *__restrict__ bpart) {
    // wave -> (row slice, 32 x 32 output tile): consecutive waves take the tiles of ONE slice (they re-read the same rows of gy
    // and x out of the caches); a 256 x 256 gradient over 375 rows is 64 tiles x 6 slices, a 32 x 32 one over 864 000 rows
    // 1 tile x 2048 slices
    const int lane = threadIdx.x & 63, lo = lane & 31, hi = lane >> 5;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= waves) return;
    const long long slice = wave / tiles;
    const int tile = (int)(wave - slice * tiles);
    const int o0 = (tile / tiles_i) * 32, i0 = (tile % tiles_i) * 32;
    const long long r0 = slice * rows_per_slice;
    const long long r1 = r0 + rows_per_slice < rows ? r0 + rows_per_slice : rows;
    float *__restrict__ pw = part + (size_t)slice * cout * cin;
    const int o = o0 + lo, i = i0 + lo;
    const bool ok_o = o < cout, ok_i = i < cin;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float bsum = 0.0f;
    const int steps = (int)((r1 - r0 + 1) / 2);   // wave-uniform trip count: an MFMA is a wave-wide instruction
    for (int s0 = 0; s0 < steps; s0 += 8) {       // eight row pairs requested before the first is consumed
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long r = r0 + 2 * (s0 + u) + hi;
            const bool ok = r < r1;                // (rows past the slice, the odd tail row: zeros)
            a[u] = (ok && ok_o) ? gy[(size_t)r * cout + o] : 0.0f;
            b[u] = (ok && ok_i) ? x[(size_t)r * cin + i] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
            bsum += a[u];
        }
    }
    if (i0 == 0 && bpart) {
        bsum += __shfl_xor(bsum, 32);
        if (hi == 0 && ok_o) bpart[(size_t)slice * cout + o] = bsum;
    }
    // accumulator register q of lane l: row (q & 3) + 8 (q >> 2) + 4 hi of the tile (= output channel), column lo (= input channel)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int oo = o0 + (q & 3) + 8 * (q >> 2) + 4 * hi;
        if (oo < cout && ok_i) pw[(size_t)oo * cin + i] = acc[q];
    }
}

// dW[e] = sum_s part[s][e] (e over cout * cin), db[o] = sum_s bpart[s][o]: a block takes 64 consecutive outputs, its 16 groups of 64
// threads split the slices (coalesced 256-B reads), LDS adds the groups.
__global__ __launch_bounds__(1024) void linear_wgrad_reduce_kernel(long long slices, int nw, int nb, const float *__restrict__ part,
                                                                    const float *__restrict__ bpart, float *__restrict__ dW,
                                                                    float *__restrict__ db) {
    __shared__ float red[16][64];
    const int j = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int blocks_w = (nw + 63) / 64;
    const bool is_b = (int)blockIdx.x >= blocks_w;
    const int n = is_b ? nb : nw;
    const int e = (is_b ? (int)blockIdx.x - blocks_w : (int)blockIdx.x) * 64 + j;
    const float *__restrict__ src = is_b ? bpart : part;
    float s = 0.0f;
    if (e < n)
        for (long long t = g; t < slices; t += 16) s += src[(size_t)t * n + e];
    red[g][j] = s;
    __syncthreads();
    if (g == 0 && e < n) {
        float tot = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += red[k][j];
        (is_b ? db : dW)[e] = tot;
    }
}

struct WgradPlan {
    int tiles_i, tiles, per;
    long long slices;
};
static WgradPlan wgrad_plan(long long rows, int cin, int cout) {
    WgradPlan p;
    p.tiles_i = (cin + 31) / 32;
    p.tiles = p.tiles_i * ((cout + 31) / 32);
    long long want = (2048 + p.tiles - 1) / p.tiles;       // about 2048 waves in all
    const long long most = (rows + 63) / 64;               // slices of at least 64 rows
    if (want > most) want = most;
    if (want < 1) want = 1;
    long long per = (rows + want - 1) / want;
    per = (per + 1) / 2 * 2;                               // even: both halves of a wave walk the same number of row pairs
    p.per = (int)per;
    p.slices = (rows + per - 1) / per;
    return p;
}

}  // namespace tgn

using namespace tgn;

// Scratch of tgn_linear_wgrad: the per-slice partial results.
TGN_API size_t tgn_linear_wgrad_workspace_bytes(long long rows, int cin, int cout) {
    if (rows <= 0 || cin <= 0 || cout <= 0) return 0;
    return (size_t)wgrad_plan(rows, cin, cout).slices * cout * (cin + 1) * sizeof(float);
}

TGN_API int tgn_linear_wgrad(long long rows, int cin, int cout, const float *x, const float *gy, float *dW, float *db,
                             void *workspace, size_t workspace_bytes, tgn_stream_t stream) {
    if (rows <= 0 || cin <= 0 || cout <= 0) return TGN_OK;
    if (!x || !gy || !dW || !workspace) {
        set_error("tgn_linear_wgrad: null pointer");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    if (workspace_bytes < tgn_linear_wgrad_workspace_bytes(rows, cin, cout)) {
        set_error("tgn_linear_wgrad: workspace of %zu bytes, %zu needed", workspace_bytes, tgn_linear_wgrad_workspace_bytes(rows, cin, cout));
        return TGN_ERR_INVALID_ARGUMENT;
    }
    const WgradPlan p = wgrad_plan(rows, cin, cout);
    float *part = (float *)workspace, *bpart = part + (size_t)p.slices * cout * cin;
    const long long waves = p.slices * p.tiles;
    hipLaunchKernelGGL(linear_wgrad_partial_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rows, cin, cout,
                       p.per, p.tiles_i, p.tiles, waves, x, gy, part, db ? bpart : (float *)nullptr);
    if (int rc = check_launch("linear_wgrad_partial_kernel")) return rc;
    const int nw = cout * cin;
    const unsigned blocks = (unsigned)((nw + 63) / 64 + (db ? (cout + 63) / 64 : 0));
    hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, p.slices, nw, cout, part, bpart, dW, db);
    return check_launch("linear_wgrad_reduce_kernel");
}

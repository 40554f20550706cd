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

__global__ __launch_bounds__(256) void linear_wgrad_partial_kernel(long long rows, int cin, int cout, int rows_per_wave,
                                                                    const float *__restrict__ x, const float *__restrict__ gy,
                                                                    float *__restrict__ part, float *__restrict__ bpart) {
    const int lane = threadIdx.x & 63, lo = lane & 31, hi = lane >> 5;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long r0 = wave * rows_per_wave;
    if (r0 >= rows) return;
    const long long r1 = r0 + rows_per_wave < rows ? r0 + rows_per_wave : rows;
    float *__restrict__ pw = part + (size_t)wave * cout * cin;
    for (int o0 = 0; o0 < cout; o0 += 32) {
        const int o = o0 + lo;
        const bool ok_o = o < cout;
        for (int i0 = 0; i0 < cin; i0 += 32) {
            const int i = i0 + lo;
            const bool ok_i = i < cin;
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
                if (hi == 0 && ok_o) bpart[(size_t)wave * cout + o] = bsum;
            }
            // accumulator register q of lane l: row (q & 3) + 8 (q >> 2) + 4 hi of the tile (= output channel), column lo (= input channel)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int oo = o0 + (q & 3) + 8 * (q >> 2) + 4 * hi;
                if (oo < cout && ok_i) pw[(size_t)oo * cin + i] = acc[q];
            }
        }
    }
}

}  // namespace tgn

using namespace tgn;

// Number of row slices (= partial results) tgn_linear_wgrad_partials writes for `rows` rows.
TGN_API long long tgn_linear_wgrad_slices(long long rows) {
    if (rows <= 0) return 0;
    long long per = (rows + 4095) / 4096;          // about 4096 waves on big inputs, slices of >= 64 rows
    if (per < 64) per = 64;
    per = (per + 1) / 2 * 2;                       // even: both halves of a wave walk the same number of row pairs
    return (rows + per - 1) / per;
}

TGN_API int tgn_linear_wgrad_partials(long long rows, int cin, int cout, const float *x, const float *gy, float *part,
                                      float *bpart, tgn_stream_t stream) {
    if (rows <= 0 || cin <= 0 || cout <= 0) return TGN_OK;
    if (!x || !gy || !part) {
        set_error("tgn_linear_wgrad_partials: null pointer");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    const long long slices = tgn_linear_wgrad_slices(rows);
    long long per = (rows + 4095) / 4096;
    if (per < 64) per = 64;
    per = (per + 1) / 2 * 2;
    const long long blocks = (slices + 3) / 4;
    hipLaunchKernelGGL(linear_wgrad_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rows, cin, cout,
                       (int)per, x, gy, part, bpart);
    return check_launch("linear_wgrad_partial_kernel");
}

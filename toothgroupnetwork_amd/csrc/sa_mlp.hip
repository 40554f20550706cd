// sa_mlp.hip -- a WHOLE two-layer set-abstraction level in one kernel (second half):
//
//   out[b,s,:] = max_k relu(bn2(W2 * relu(bn1(W1 * [x[idx]-c, f[idx]] + bias1)) + bias2))      (pointnet2_utils.py:286-294)
//
// for the two-layer shared MLPs every set-abstraction level of the reference networks has (pointnet_pp.py:13-15,
// tsg_centroid_module.py:10-12, tsg_seg_module.py:11-28), eval mode, BatchNorms folded.  Neither the grouped
// (B,S,K,3+D) tensor nor the (B,S,K,C1) output of the first layer nor the (B,S,K,C2) output of the second exists
// anywhere: a workgroup owns 128 (query, neighbour) rows -- 4 queries of 32 neighbours or 2 of 64 -- and a 128-column
// slab of the second layer;
//   * the first layer's rows are produced on the fly, 16 channels at a time, straight into the A fragments of the GEMM:
//       commuted form (wide inputs): h1[r, c] = relu(A1[b, idx_r, c] + cst[q_r, c]), A1 = [f, x] * W1t per POINT
//           (tgn_sa_point_transform, an fp32-MFMA GEMM over the N points, S*K/N times fewer flops than per row) and
//           cst = bias - Wxs . centre per query, kept in LDS;
//       direct form (3 + D <= 16: the first level of a network): h1[r, c] = relu(b1[c] + sum_j g_r[j] * Wd[j, c]) from the
//           gathered row g_r = [x - c, f] held in registers (the weights of 8 channels are wave-uniform: scalar loads);
//   * the second layer runs on v_mfma_f32_32x32x2_f32 (exact fp32 fma chains: the 1e-5 contract holds): 128 x 128 x 16
//     tiles, 2 x 2 MFMA tiles per wave, fragments in LDS in the layout the MFMA operands want ([k parity][row][k-step]: a
//     lane fetches its eight k-steps of a tile with two ds_read_b128, producers store two ds_write_b128), two tile
//     buffers, ONE barrier per 16-wide K tile, the next tile's gathers in flight underneath;
//   * the max over a query's neighbours is taken on the accumulators (bias and ReLU commute with it) and (B,S,C2) is all
//     that is written.
// Rows past K (K < 32 or 32 < K < 64) repeat neighbour 0, which a max does not see.
#include "tgn_common.h"
#include <type_traits>

namespace tgn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMlpMT = 128, kMlpNT = 128, kMlpKT = 16;
constexpr int kFS = 12;                        // fragment row stride in floats: 8 k-steps + 4 pad (b128 conflict-free)
constexpr int kFrag = 2 * 128 * kFS;           // one operand tile [k parity][row][k-step]

template <typename IdxT, bool DIRECT>
__global__ __launch_bounds__(256, 2) void sa_mlp2_max_kernel(long long Q, int N, int S, int K, int D, int C1p, int C2, int ostride,
                                                              const float *__restrict__ A1,       // commuted: (B,N,C1p)
                                                              const float *__restrict__ xyz,      // direct: (B,N,3)
                                                              const float *__restrict__ points,   // direct: (B,N,D)
                                                              const float *__restrict__ new_xyz,  // (B,S,3)
                                                              const float *__restrict__ W1,       // commuted: Wxs (3,C1p); direct: Wd (16,C1p)
                                                              const float *__restrict__ b1,       // (C1p)
                                                              const IdxT *__restrict__ idx,       // (B,S,K)
                                                              const float *__restrict__ W2f,      // (C1p/8, C2, 8)
                                                              const float *__restrict__ b2,       // (C2)
                                                              float *__restrict__ out,            // (B,S,C2), row stride `ostride` floats
                                                              int *__restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *FA = smem, *FB = smem + 2 * kFrag, *cst = smem + 4 * kFrag;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int lo = lane & 31, hi = lane >> 5;
    const int kshift = K > 32 ? 6 : 5, Kp = 1 << kshift, QPT = kMlpMT >> kshift;
    const unsigned ntiles = ((unsigned)C2 + kMlpNT - 1) / kMlpNT;
    const long long mtiles = (Q + QPT - 1) / QPT;
    // XCD-contiguous item ranges, column tile fastest: the blocks that gather the same rows run next to each other on one L2
    const unsigned nb = gridDim.x;
    const long long item = (long long)(blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3);
    const long long mt = item / ntiles;
    if (mt >= mtiles) return;
    const int col0 = (int)(item - mt * ntiles) * kMlpNT;
    const long long q0 = mt * QPT;

    // ---- producer roles.  A: thread (row ar, k half ah) makes 8 consecutive channels of one (query, neighbour) row per K tile;
    //      B: thread (column bc, k half bh) fetches 8 consecutive k of one column of W2 (32 contiguous bytes of W2f)
    const int ar = tid & 127;
    const int ah = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int ql = ar >> kshift;
    long long q = q0 + ql;
    if (q >= Q) q = Q - 1;
    const int b = (int)(q / S);
    int kk = ar & (Kp - 1);
    if (kk >= K) kk = 0;
    long long v;
    if (idx) {
        v = (long long)idx[q * K + kk];
        if (v < 0) v += N;
        bool bad = false;
        if (v < 0 || v >= N) {
            bad = true;
            v = 0;
        }
        if (err && col0 == 0 && bad) atomicOr(err, 1);
    } else {
        // group_all (pointnet2_utils.py:178-195): "query" s of a cloud owns the points s*Kp .. s*Kp + Kp-1, no centre; a short
        // last chunk repeats its first point, which a max does not see
        const long long first = (q - (long long)b * S) * Kp;
        v = first + kk < N ? first + kk : first;
    }
    const float *__restrict__ arow = A1 + ((size_t)b * N + (size_t)v) * C1p + ah * 8;
    float g[16];
    if (DIRECT) {
        const float *__restrict__ px = xyz + ((size_t)b * N + (size_t)v) * 3;
        const float *__restrict__ pf = points + ((size_t)b * N + (size_t)v) * D;
        g[0] = px[0] - (new_xyz ? new_xyz[q * 3 + 0] : 0.0f);
        g[1] = px[1] - (new_xyz ? new_xyz[q * 3 + 1] : 0.0f);
        g[2] = px[2] - (new_xyz ? new_xyz[q * 3 + 2] : 0.0f);
#pragma unroll
        for (int j = 3; j < 16; ++j) g[j] = j - 3 < D ? pf[j - 3] : 0.0f;
    } else {
        for (int l = 0; l < QPT; ++l) {   // cst[l][c] = b1[c] - Wxs[:,c] . centre of query l
            long long qq = q0 + l;
            if (qq >= Q) qq = Q - 1;
            if (!new_xyz) {
                for (int c = tid; c < C1p; c += 256) cst[l * C1p + c] = b1[c];
                continue;
            }
            const float cx = new_xyz[qq * 3 + 0], cy = new_xyz[qq * 3 + 1], cz = new_xyz[qq * 3 + 2];
            for (int c = tid; c < C1p; c += 256)
                cst[l * C1p + c] = b1[c] - ((W1[c] * cx + W1[C1p + c] * cy) + W1[2 * C1p + c] * cz);
        }
    }
    const int bc = tid & 127, bh = tid >> 7;
    const bool bok = col0 + bc < C2;
    const float *__restrict__ brow = W2f + ((size_t)bh * C2 + (size_t)(bok ? col0 + bc : 0)) * 8;
    const size_t bstep = (size_t)2 * C2 * 8;   // one K tile = two k-blocks of 8

    f32x4 ra0, ra1, rb0 = {0.f, 0.f, 0.f, 0.f}, rb1 = {0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int t) {
        if (!DIRECT) {
            ra0 = *(const f32x4 *)(arow + t * kMlpKT);
            ra1 = *(const f32x4 *)(arow + t * kMlpKT + 4);
        }
        if (bok) {
            rb0 = *(const f32x4 *)(brow + (size_t)t * bstep);
            rb1 = *(const f32x4 *)(brow + (size_t)t * bstep + 4);
        }
    };
    auto stage = [&](int t, int buf) {
        float h[8];
        if (DIRECT) {
            const float *__restrict__ wd = W1 + t * kMlpKT + ah * 8;   // wave-uniform: scalar loads
            const float *__restrict__ bb = b1 + t * kMlpKT + ah * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) h[i] = bb[i];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j < 3 + D) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) h[i] = __builtin_fmaf(g[j], wd[(size_t)j * C1p + i], h[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) h[i] = fmaxf(h[i], 0.0f);
        } else {
            const float *cs = cst + ql * C1p + t * kMlpKT + ah * 8;
            const f32x4 c0 = *(const f32x4 *)cs, c1 = *(const f32x4 *)(cs + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                h[i] = fmaxf(ra0[i] + c0[i], 0.0f);
                h[4 + i] = fmaxf(ra1[i] + c1[i], 0.0f);
            }
        }
        // channel 8*ah + i of the tile is k-step 4*ah + (i >> 1), parity i & 1
        float *fa = FA + buf * kFrag + ar * kFS + ah * 4;
        *(f32x4 *)fa = f32x4{h[0], h[2], h[4], h[6]};
        *(f32x4 *)(fa + 128 * kFS) = f32x4{h[1], h[3], h[5], h[7]};
        float *fb = FB + buf * kFrag + bc * kFS + bh * 4;
        *(f32x4 *)fb = f32x4{rb0[0], rb0[2], rb1[0], rb1[2]};
        *(f32x4 *)(fb + 128 * kFS) = f32x4{rb0[1], rb0[3], rb1[1], rb1[3]};
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    auto compute = [&](int buf) {
        const float *fa = FA + buf * kFrag + (hi * 128 + wm * 64 + lo) * kFS;
        const float *fb = FB + buf * kFrag + (hi * 128 + wn * 64 + lo) * kFS;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const f32x4 a0 = *(const f32x4 *)(fa + half * 4), a1 = *(const f32x4 *)(fa + 32 * kFS + half * 4);
            const f32x4 w0 = *(const f32x4 *)(fb + half * 4), w1 = *(const f32x4 *)(fb + 32 * kFS + half * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], w0[s], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], w1[s], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], w0[s], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], w1[s], acc[1][1], 0, 0, 0);
            }
        }
    };

    const int T = C1p / kMlpKT;
    fetch(0);
    __syncthreads();   // cst is complete
    stage(0, 0);
    if (T > 1) fetch(1);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        compute(t & 1);
        if (t + 1 < T) stage(t + 1, (t + 1) & 1);   // the other buffer: last read before the barrier that ended tile t-1
        if (t + 2 < T) fetch(t + 2);
        __syncthreads();
    }

    // ---- max over the rows of a query.  Accumulator register r of lane l is row (r & 3) + 8 (r >> 2) + 4 hi, column lo of its tile.
    float cm[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float m = acc[i][j][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[i][j][r]);
            cm[i][j] = fmaxf(m, __shfl_xor(m, 32));
        }
    if (hi == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = col0 + wn * 64 + j * 32 + lo;
            if (col >= C2) continue;
            const float bias = b2[col];
            if (kshift == 6) {   // 64 rows = one query
                const long long qq = q0 + wm;
                if (qq < Q) out[(size_t)qq * ostride + col] = fmaxf(fmaxf(cm[0][j], cm[1][j]) + bias, 0.0f);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const long long qq = q0 + wm * 2 + i;
                    if (qq < Q) out[(size_t)qq * ostride + col] = fmaxf(cm[i][j] + bias, 0.0f);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same level with the second layer on the BF16 matrix cores at fp32 accuracy ("bf16x3"): v_mfma_f32_32x32x16_bf16 runs at 16x
// the rate of v_mfma_f32_32x32x2_f32, so an fp32 product computed as SIX bf16 products is still 16 / 6 = 2.7x faster.  Both operands
// are written as a sum of three bf16 numbers, x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (every
// subtraction exact in fp32; 24 bits of mantissa kept), and
//     a * b  ~=  a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1)          dropped: a2 b3 + a3 b2 + a3 b3 <= 3 * 2^-26 |a b|
// Every bf16 x bf16 product is exact in fp32 and the matrix core accumulates in fp32: the result carries fp32-class error (measured
// against the float64 oracle in tests/test_gpu_sa_fused.py, elementwise 1e-5 like the fp32-MFMA form).  The weights are split ONCE
// (tgn_sa_mlp2_split_weights) into the image the kernel's LDS tiles have, so a K tile of B is 12 contiguous KiB that arrive by LDS-DMA
// (buffer_load_dwordx4 ... lds: no VGPR round trip, no ds_write); the activations are split in registers behind the ReLU
// (v_cvt_pk_bf16_f32 + shift / mask + v_pk_add_f32: 36 vector instructions per 8 channels).
//
// LDS tile of one operand and K tile (16 k): [component 3][row 128][16 k as bf16 = 32 B]; a lane's MFMA fragment (row l & 31 of a
// 32-row block, k half l >> 5: 8 consecutive k) is ONE ds_read_b128.  The two 16-B halves of a row are swapped where
// ((row >> 2) ^ (row >> 3)) & 1: with that, ds_read_b128 (lane groups of 16, 64 banks) and ds_write_b128 (8 contiguous lanes, 32
// banks) are both free of bank conflicts (checked by enumeration, tools/lds_swizzle_check.py).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kSplitPlane = 128 * 32;         // bytes: one bf16 component of one operand tile
constexpr int kSplitTile = 3 * kSplitPlane;   // 12 KiB

__device__ __forceinline__ int split_chunk(int row, int khalf) {
    return row * 32 + ((khalf ^ (((row >> 2) ^ (row >> 3)) & 1)) << 4);
}

// x[0..7] -> three bf16x8 (components 1, 2, 3)
__device__ __forceinline__ void split3(const float *x, bf16x8 &c1, bf16x8 &c2, bf16x8 &c3) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2 v = {x[i], x[i + 1]};
        const bf16x2 p1 = __builtin_convertvector(v, bf16x2);
        const f32x2 r1 = v - __builtin_convertvector(p1, f32x2);
        const bf16x2 p2 = __builtin_convertvector(r1, bf16x2);
        const f32x2 r2 = r1 - __builtin_convertvector(p2, f32x2);
        const bf16x2 p3 = __builtin_convertvector(r2, bf16x2);
        c1[i] = p1[0], c1[i + 1] = p1[1];
        c2[i] = p2[0], c2[i + 1] = p2[1];
        c3[i] = p3[0], c3[i + 1] = p3[1];
    }
}

// W2f (C1p/8, C2, 8) fp32 -> the kernel's B image: [column tile of 128][K tile of 16][component][column][swizzled 32 B]
__global__ __launch_bounds__(256) void sa_mlp2_split_weights_kernel(int C1p, int C2, const float *__restrict__ W2f,
                                                                    unsigned char *__restrict__ W2s) {
    const int T = C1p / kMlpKT;
    const long long chunks = (long long)((C2 + kMlpNT - 1) / kMlpNT) * T * 256;   // (column, k half) pairs
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= chunks) return;
    const int col = (int)(i & 127), khalf = (int)((i >> 7) & 1);
    const long long tile = i >> 8;   // nt * T + t
    const int t = (int)(tile % T), nt = (int)(tile / T);
    const int c = nt * kMlpNT + col;
    float w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = c < C2 ? W2f[((size_t)(2 * t + khalf) * C2 + c) * 8 + k] : 0.0f;
    bf16x8 c1, c2, c3;
    split3(w, c1, c2, c3);
    unsigned char *dst = W2s + (size_t)tile * kSplitTile + split_chunk(col, khalf);
    *(bf16x8 *)dst = c1;
    *(bf16x8 *)(dst + kSplitPlane) = c2;
    *(bf16x8 *)(dst + 2 * kSplitPlane) = c3;
}

// Tile shape: WM x 2 waves, each 64 rows x (32 TN) columns -- (WM, TN) = (2, 2): 128 x 128 per workgroup of 256 threads, two workgroups
// per CU; (4, 4): 256 x 256 per workgroup of 512 threads, one per CU.  At 2.7x the fp32-MFMA rate the 128 x 128 tile is bound by
// the memory system, not the matrix cores: every 128 rows re-read the whole weight image (4.7 MB at 784 x 1024: more than an XCD's
// L2) and every 128 columns re-gather the rows -- 26 flop per byte, 7.8 TB/s of L2 / Infinity-Cache traffic at 200 TFLOP/s
// (profiles/r04_sa_split_counters.txt).  The 256 x 256 tile halves both streams.
#ifndef TGN_SA_SPLIT_BLOCKS
#define TGN_SA_SPLIT_BLOCKS 2   // workgroups per CU the 128 x 128 form is compiled for (A/B builds: 3 = 168 VGPRs, spills; tools/ab_build.sh)
#endif
template <typename IdxT, bool DIRECT, int WM, int TN>
__global__ __launch_bounds__(WM * 128, WM == 2 ? TGN_SA_SPLIT_BLOCKS : 1) void sa_mlp2_max_split_kernel(
    long long Q, int N, int S, int K, int D, int C1p, int C2, int ostride, const float *__restrict__ A1, const float *__restrict__ xyz,
    const float *__restrict__ points, const float *__restrict__ new_xyz, const float *__restrict__ W1, const float *__restrict__ b1,
    const IdxT *__restrict__ idx, const unsigned char *__restrict__ W2s,   // split image of W2 (128-column tiles)
    const float *__restrict__ b2, float *__restrict__ out, int *__restrict__ err) {
    constexpr int MT = WM * 64, NTL = TN * 64, THREADS = WM * 128;
    constexpr int kAPlane = MT * 32, kATile = 3 * kAPlane;          // bytes
    constexpr int kBTile = (NTL / 128) * kSplitTile;                // one or two 128-column tiles of the image, back to back
    constexpr int kPieces = kBTile / 1024 / (2 * WM);               // LDS-DMA pieces per wave and K tile
    static_assert(kPieces * 2 * WM * 1024 == kBTile, "whole pieces");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char *FA = smem_b, *FB = smem_b + 2 * kATile;
    float *cst = (float *)(smem_b + 2 * kATile + 2 * kBTile);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int lo = lane & 31, hi = lane >> 5;
    const int kshift = K > 32 ? 6 : 5, Kp = 1 << kshift, QPT = MT >> kshift;
    const unsigned ntiles = ((unsigned)C2 + NTL - 1) / NTL;
    const unsigned ntiles128 = ((unsigned)C2 + kMlpNT - 1) / kMlpNT;
    const long long mtiles = (Q + QPT - 1) / QPT;
    // XCD-contiguous item ranges.  Within an XCD the items run SCAN by scan, within a scan COLUMN TILE by column tile, the scan's row
    // tiles fastest: the ~32-64 workgroups an XCD holds at a time then share ONE column tile of the weight image (1.2 MB at 784 x 256)
    // and gather from ONE scan's first-layer rows (1.6 MB at 512 x 784) -- both stay in its 4 MB L2.  (Column tile fastest, the
    // fp32 kernel's order, has all column tiles -- the whole 4.7 MB image -- cycling through the L2 at once.)
    const unsigned nb = gridDim.x;
    const long long item = (long long)(blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3);
    const long long R = (S + QPT - 1) / QPT;                       // row tiles per scan (tiles of a ragged S straddle scans: harmless)
    const long long per_group = R * ntiles;
    const long long grp = item / per_group, rem = item - grp * per_group;
    const long long base_mt = grp * R;
    if (base_mt >= mtiles) return;
    const long long Rg = mtiles - base_mt < R ? mtiles - base_mt : R;   // row tiles of this group (the last one may be short)
    if (rem >= Rg * ntiles) return;
    const int ntile = (int)(rem / Rg);
    const long long mt = base_mt + (rem - (long long)ntile * Rg);
    const int col0 = ntile * NTL;
    const long long q0 = mt * QPT;
    const int T = C1p / kMlpKT;

    // ---- producer of A: thread (row ar, k half ah) makes 8 consecutive channels of one (query, neighbour) row per K tile
    const int ar = tid & (MT - 1);
    const int ah = __builtin_amdgcn_readfirstlane(tid / MT);
    const int ql = ar >> kshift;
    long long q = q0 + ql;
    if (q >= Q) q = Q - 1;
    const int b = (int)(q / S);
    int kk = ar & (Kp - 1);
    if (kk >= K) kk = 0;
    long long v;
    if (idx) {
        v = (long long)idx[q * K + kk];
        if (v < 0) v += N;
        bool bad = false;
        if (v < 0 || v >= N) {
            bad = true;
            v = 0;
        }
        if (err && col0 == 0 && bad) atomicOr(err, 1);
    } else {
        const long long first = (q - (long long)b * S) * Kp;
        v = first + kk < N ? first + kk : first;
    }
    const float *__restrict__ arow = A1 + ((size_t)b * N + (size_t)v) * C1p + ah * 8;
    float g[16];
    if (DIRECT) {
        const float *__restrict__ px = xyz + ((size_t)b * N + (size_t)v) * 3;
        const float *__restrict__ pf = points + ((size_t)b * N + (size_t)v) * D;
        g[0] = px[0] - (new_xyz ? new_xyz[q * 3 + 0] : 0.0f);
        g[1] = px[1] - (new_xyz ? new_xyz[q * 3 + 1] : 0.0f);
        g[2] = px[2] - (new_xyz ? new_xyz[q * 3 + 2] : 0.0f);
#pragma unroll
        for (int j = 3; j < 16; ++j) g[j] = j - 3 < D ? pf[j - 3] : 0.0f;
    } else {
        for (int l = 0; l < QPT; ++l) {
            long long qq = q0 + l;
            if (qq >= Q) qq = Q - 1;
            if (!new_xyz) {
                for (int c = tid; c < C1p; c += THREADS) cst[l * C1p + c] = b1[c];
                continue;
            }
            const float cx = new_xyz[qq * 3 + 0], cy = new_xyz[qq * 3 + 1], cz = new_xyz[qq * 3 + 2];
            for (int c = tid; c < C1p; c += THREADS)
                cst[l * C1p + c] = b1[c] - ((W1[c] * cx + W1[C1p + c] * cy) + W1[2 * C1p + c] * cz);
        }
    }
    // ---- B: K tile t of 128-column tile j is 12 contiguous KiB of the image at (j T + t); a 256-wide workgroup takes tiles 2 ntile
    // and 2 ntile + 1 (the second may lie past the last one: the descriptor's bound then returns zeros).  Wave wv moves kPieces pieces.
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(W2s), 0, (int)((size_t)ntiles128 * T * kSplitTile), 0x00020000);
    auto dma = [&](int t, int buf) {
#pragma unroll
        for (int p = 0; p < kPieces; ++p) {
            const int piece = wv * kPieces + p;          // 0 .. kBTile / 1024 - 1; 12 pieces per 128-column tile
            const int half = piece / 12, within = piece - half * 12;
            const int j128 = ntile * (NTL / 128) + half;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *)(FB + buf * kBTile + piece * 1024), 16,
                                                     lane * 16, (j128 * T + t) * kSplitTile + within * 1024, 0, 0);
        }
    };
    f32x4 ra0, ra1;
    auto fetch = [&](int t) {
        if (!DIRECT) {
            ra0 = *(const f32x4 *)(arow + t * kMlpKT);
            ra1 = *(const f32x4 *)(arow + t * kMlpKT + 4);
        }
    };
    auto stage = [&](int t, int buf) {
        float h[8];
        if (DIRECT) {
            const float *__restrict__ wd = W1 + t * kMlpKT + ah * 8;   // wave-uniform: scalar loads
            const float *__restrict__ bb = b1 + t * kMlpKT + ah * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) h[i] = bb[i];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j < 3 + D) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) h[i] = __builtin_fmaf(g[j], wd[(size_t)j * C1p + i], h[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) h[i] = fmaxf(h[i], 0.0f);
        } else {
            const float *cs = cst + ql * C1p + t * kMlpKT + ah * 8;
            const f32x4 c0 = *(const f32x4 *)cs, c1 = *(const f32x4 *)(cs + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                h[i] = fmaxf(ra0[i] + c0[i], 0.0f);
                h[4 + i] = fmaxf(ra1[i] + c1[i], 0.0f);
            }
        }
        bf16x8 p1, p2, p3;
        split3(h, p1, p2, p3);
        unsigned char *fa = FA + buf * kATile + split_chunk(ar, ah);
        *(bf16x8 *)fa = p1;
        *(bf16x8 *)(fa + kAPlane) = p2;
        *(bf16x8 *)(fa + 2 * kAPlane) = p3;
    };
    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    bf16x8 fa_[2][3], fw_[TN][3];   // fragments: [tile of the wave][bf16 component]
    int ca[2], cb[TN];              // their byte offsets inside a component plane
#pragma unroll
    for (int i = 0; i < 2; ++i) ca[i] = split_chunk(wm * 64 + i * 32 + lo, hi);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = wn * (NTL / 2) + j * 32 + lo;   // column within the workgroup's tile; 128-column halves are separate images
        cb[j] = (col >> 7) * kSplitTile + split_chunk(col & 127, hi);
    }
    auto ra = [&](int buf, int c) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fa_[i][c] = *(const bf16x8 *)(FA + buf * kATile + c * kAPlane + ca[i]);
    };
    auto rw = [&](int buf, int c) {
#pragma unroll
        for (int j = 0; j < TN; ++j) fw_[j][c] = *(const bf16x8 *)(FB + buf * kBTile + c * kSplitPlane + cb[j]);
    };
    // one of the six products of a K tile: component CA of the activations x component CB of the weights, 2 TN MFMAs on 2 TN different
    // accumulators.  (The order of the six is irrelevant to the rounding: the accumulator already holds the sum over the earlier K tiles.)
    auto product = [&](auto CA, auto CB) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[i][CA()], fw_[j][CB()], acc[i][j], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    const int last = T - 1;

    fetch(0);
    dma(0, 0);
    __syncthreads();   // cst is complete
    stage(0, 0);
    fetch(last < 1 ? last : 1);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the LDS-DMA pieces have landed (hipcc orders no ds_read behind them)
    __syncthreads();
    if constexpr (DIRECT) {
        // (255 registers: this read order and this product order are the ones the form does not spill in)
        auto tile = [&](int buf) {
            ra(buf, 0);
            ra(buf, 1);
            ra(buf, 2);
            rw(buf, 0);
            rw(buf, 1);
            rw(buf, 2);
        };
        auto mma = [&]() {
            product(I2{}, I0{});
            product(I0{}, I2{});
            product(I1{}, I1{});
            product(I1{}, I0{});
            product(I0{}, I1{});
            product(I0{}, I0{});
        };
        for (int t = 0; t < last; ++t) {
            dma(t + 1, (t + 1) & 1);          // FB / FA[(t+1)&1] were last read in trip t-1: every wave is past the barrier that ended it
            tile(t & 1);                      // IN FRONT of the producer's LDS stores: hipcc keeps an LDS read behind an earlier LDS write
                                              // it cannot tell apart, which would chain  split -> ds_write -> ds_read -> MFMA
            stage(t + 1, (t + 1) & 1);
            mma();
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
        }
        tile(last & 1);
        mma();
    } else {
        // One trip per K tile, one basic block each (the scheduling hints need it).  The waves of a CU leave the barrier together and all
        // read their fragments at once (20 b128 reads x 8 waves = 1280 cycles of LDS), so the matrix cores used to idle behind the first
        // reads of every trip.  Two things shorten that: the fragments are read in the order the products need them -- a2, w2, a1, w3,
        // then a3, w1 -- and the LAST product of a tile (a3 x w1) is held back and issued after the barrier, under the reads of the next
        // tile: it needs no LDS data, and its registers are the last ones the new tile overwrites.  (+2.5..4 % on the reference net's
        // levels, profiles/r04_sa_held_group.txt.)
        auto trip = [&](int t, auto FIRST, auto LAST) {
            const int buf = t & 1;
            f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
            if constexpr (!LAST()) {
                dma(t + 1, buf ^ 1);          // FB / FA[buf ^ 1] were last read in trip t-1: every wave is past the barrier that ended it
                const float *cs = cst + ql * C1p + (t + 1) * kMlpKT + ah * 8;   // the producer's per-query constants first: its
                c0 = *(const f32x4 *)cs;                                         // arithmetic can start under the first MFMAs
                c1 = *(const f32x4 *)(cs + 4);
            }
            // the fragment reads stand IN FRONT of the producer's LDS stores: hipcc keeps an LDS read behind an earlier LDS write it
            // cannot tell apart, which would chain  split -> ds_write -> ds_read -> MFMA  and serialise the trip
            ra(buf, 1);
            rw(buf, 1);
            ra(buf, 0);
            rw(buf, 2);
            if constexpr (!FIRST()) product(I2{}, I0{});   // of tile t-1
            ra(buf, 2);
            rw(buf, 0);
            if constexpr (!LAST()) {                   // tile t+1 of the activations, from the registers fetch(t+1) filled one trip ago
                float h[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    h[i] = fmaxf(ra0[i] + c0[i], 0.0f);
                    h[4 + i] = fmaxf(ra1[i] + c1[i], 0.0f);
                }
                bf16x8 p1, p2, p3;
                split3(h, p1, p2, p3);
                unsigned char *sa = FA + (buf ^ 1) * kATile + split_chunk(ar, ah);
                *(bf16x8 *)sa = p1;
                *(bf16x8 *)(sa + kAPlane) = p2;
                *(bf16x8 *)(sa + 2 * kAPlane) = p3;
                fetch(t + 2 < last ? t + 2 : last);   // in flight under the MFMAs (the last trip re-reads tile T-1: unused)
            }
            product(I1{}, I1{});
            product(I0{}, I1{});
            product(I0{}, I2{});
            product(I1{}, I0{});
            product(I0{}, I0{});
            if constexpr (!LAST()) {
                // the waves of a SIMD run in phase (SQ counters in profiles/), so the producer's work has to hide under this wave's OWN
                // matrix instructions: a 32x32x16 bf16 MFMA holds the pipe for 32 cycles, room for ~4 other instructions.  The hints are
                // best effort (the greedy solver drops what it cannot place); this set measured best of six (same file).
                if constexpr (!FIRST()) __builtin_amdgcn_sched_group_barrier(0x008, 2 * TN, 0);   // the held product first
#pragma unroll
                for (int m = 0; m < 10 * TN; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, TN == 2 ? 3 : 2, 0);   // VALU
                }
                __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);                     // DS writes of the staged tile
                __builtin_amdgcn_s_waitcnt(0x0F70);
                __syncthreads();
            }
        };
        if (last == 0) {
            trip(0, std::true_type{}, std::true_type{});
        } else {
            trip(0, std::true_type{}, std::false_type{});
            for (int t = 1; t < last; ++t) trip(t, std::false_type{}, std::false_type{});
            trip(last, std::false_type{}, std::true_type{});
        }
        product(I2{}, I0{});
    }

    // ---- max over the rows of a query.  Accumulator register r of lane l is row (r & 3) + 8 (r >> 2) + 4 hi, column lo of its tile.
    float cm[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float m = acc[i][j][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[i][j][r]);
            cm[i][j] = fmaxf(m, __shfl_xor(m, 32));
        }
    if (hi == 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + wn * (NTL / 2) + j * 32 + lo;
            if (col >= C2) continue;
            const float bias = b2[col];
            if (kshift == 6) {   // the wave's 64 rows = one query
                const long long qq = q0 + wm;
                if (qq < Q) out[(size_t)qq * ostride + col] = fmaxf(fmaxf(cm[0][j], cm[1][j]) + bias, 0.0f);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const long long qq = q0 + wm * 2 + i;
                    if (qq < Q) out[(size_t)qq * ostride + col] = fmaxf(cm[i][j] + bias, 0.0f);
                }
            }
        }
    }
}

// Per-point first layer of the commuted form, A[m, :] = [points[m, :], xyz[m, :]] * Wt (tgn_sa_point_transform), on the same bf16x3
// scheme: rows [features..., x, y, z] split in registers, the weights as a split image (tgn_sa_mlp2_split_weights of Wt arranged as
// (Kp/8, C1, 8)), 128 x 128 tile, the full tile written (no ReLU, no max: the chained kernel adds the per-query constant).
__global__ __launch_bounds__(256, 2) void sa_point_transform_split_kernel(long long M, int D, int Kp, int C1,
                                                                           const float *__restrict__ xyz, const float *__restrict__ points,
                                                                           const unsigned char *__restrict__ Wts, float *__restrict__ A) {
    __shared__ __attribute__((aligned(16))) unsigned char smem_t[4 * kSplitTile];
    unsigned char *FA = smem_t, *FB = smem_t + 2 * kSplitTile;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int lo = lane & 31, hi = lane >> 5;
    const int Kc = D + 3, T = Kp / kMlpKT;
    const unsigned ntiles = ((unsigned)C1 + kMlpNT - 1) / kMlpNT;
    const long long row0 = (long long)blockIdx.y * kMlpMT;
    const int ntile = (int)blockIdx.x, col0 = ntile * kMlpNT;
    const int ar = tid & 127;
    const int ah = __builtin_amdgcn_readfirstlane(tid >> 7);
    const long long grow = row0 + ar;
    const bool row_ok = grow < M;
    const bool feat4 = (D & 3) == 0 && points != nullptr;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(Wts), 0,
                                                                          (int)((size_t)ntiles * T * kSplitTile), 0x00020000);
    auto dma = [&](int t, int buf) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int piece = wv * 3 + p;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *)(FB + buf * kSplitTile + piece * 1024), 16,
                                                     lane * 16, (ntile * T + t) * kSplitTile + piece * 1024, 0, 0);
        }
    };
    float ra[8];
    auto fetch = [&](int t) {
        const int c = t * kMlpKT + ah * 8;   // first channel of this thread's 8
        if (row_ok && feat4 && c + 8 <= D) {
            const f32x4 u = *(const f32x4 *)(points + (size_t)grow * D + c);
            const f32x4 w = *(const f32x4 *)(points + (size_t)grow * D + c + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = u[i];
                ra[4 + i] = w[i];
            }
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
    };
    auto stage = [&](int buf) {
        bf16x8 p1, p2, p3;
        split3(ra, p1, p2, p3);
        unsigned char *fa = FA + buf * kSplitTile + split_chunk(ar, ah);
        *(bf16x8 *)fa = p1;
        *(bf16x8 *)(fa + kSplitPlane) = p2;
        *(bf16x8 *)(fa + 2 * kSplitPlane) = p3;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    bf16x8 fa_[2][3], fw_[2][3];
    auto load_frags = [&](int buf) {
        const unsigned char *fa = FA + buf * kSplitTile, *fb = FB + buf * kSplitTile;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ca = split_chunk(wm * 64 + i * 32 + lo, hi), cb = split_chunk(wn * 64 + i * 32 + lo, hi);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                fa_[i][c] = *(const bf16x8 *)(fa + c * kSplitPlane + ca);
                fw_[i][c] = *(const bf16x8 *)(fb + c * kSplitPlane + cb);
            }
        }
    };
    auto mma = [&]() {
#define TGN_SPLIT_STEP(CA, CB)                                                                                \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[i][CA], fw_[j][CB], acc[i][j], 0, 0, 0)
        TGN_SPLIT_STEP(0, 0);
        TGN_SPLIT_STEP(0, 1);
        TGN_SPLIT_STEP(1, 0);
        TGN_SPLIT_STEP(1, 1);
        TGN_SPLIT_STEP(0, 2);
        TGN_SPLIT_STEP(2, 0);
#undef TGN_SPLIT_STEP
    };
    fetch(0);
    dma(0, 0);
    stage(0);
    if (T > 1) fetch(1);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the LDS-DMA pieces have landed
    __syncthreads();
    for (int t = 0; t + 1 < T; ++t) {
        dma(t + 1, (t + 1) & 1);
        load_frags(t & 1);
        stage((t + 1) & 1);
        fetch(t + 2 < T ? t + 2 : T - 1);
        mma();
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
    }
    load_frags((T - 1) & 1);
    mma();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = col0 + wn * 64 + j * 32 + lo;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (row < M && col < C1) A[(size_t)row * C1 + col] = acc[i][j][r];
            }
        }
}

// out[b, c] = max_s part[b, s, c]: the chunks of a group_all level (post-ReLU values, so the order of the two maxima is free)
__global__ __launch_bounds__(256) void sa_chunks_max_kernel(int B, int S, int C, int ostride, const float *__restrict__ part,
                                                            float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * C) return;
    const int b = (int)(i / C), c = (int)(i - (long long)b * C);
    const float *p = part + (size_t)b * S * C + c;
    float m = p[0];
    for (int s = 1; s < S; ++s) m = fmaxf(m, p[(size_t)s * C]);
    out[(size_t)b * ostride + c] = m;
}

}  // namespace tgn

using namespace tgn;

// 1 if tgn_sa_mlp2_max takes the direct form for this first layer (the caller then passes xyz / points / Wd, no A1)
TGN_API int tgn_sa_mlp2_direct_supported(int K, int D) { return (D >= 0 && 3 + D <= 16 && K >= 1 && K <= 64) ? 1 : 0; }

// shared launcher: idx == nullptr selects the group_all form (chunks of 32 / 64 consecutive points, new_xyz may be null)
static int sa_mlp2_launch(const char *who, int B, int N, int S, int K, int D, int C1p, int C2, const float *A1, const float *xyz,
                          const float *points, const float *new_xyz, const float *W1, const float *b1, const void *idx,
                          int idx_is_int64, const float *W2f, const float *b2, float *out, int out_stride, tgn_stream_t stream,
                          bool split = false) {
    const long long Q = (long long)B * S;
    const bool direct = A1 == nullptr;
    if (!b1 || !W2f || !b2 || !out || (direct && (!xyz || !W1 || (D > 0 && !points))) || (!direct && new_xyz && !W1)) {
        set_error("%s: null pointer", who);
        return TGN_ERR_INVALID_ARGUMENT;
    }
    if (K < 1 || K > 64 || C1p < 16 || (C1p & 15) || N < 1 || out_stride < C2 || (direct && !tgn_sa_mlp2_direct_supported(K, D)) ||
        (((uintptr_t)A1 | (uintptr_t)W2f | (uintptr_t)W1 | (uintptr_t)b1) & 15)) {
        set_error("%s: needs 1 <= nsample <= 64, a first-layer width padded to a multiple of 16, 16-byte aligned "
                  "operands, and 3+D <= 16 for the direct form", who);
        return TGN_ERR_UNSUPPORTED;
    }
    const int qpt = K > 32 ? 2 : 4;
    static_assert(4 * kFrag * sizeof(float) == 4 * kSplitTile, "both forms keep four 12-KiB operand tiles");
    const size_t lds = (size_t)(4 * kFrag + (direct ? 0 : qpt * C1p)) * sizeof(float);
    if (lds > 80 * 1024) {   // two workgroups per CU
        set_error("%s: first-layer width %d needs %zu bytes of LDS per workgroup (limit 80 KiB)", who, C1p, lds);
        return TGN_ERR_UNSUPPORTED;
    }
    const long long mtiles = (Q + qpt - 1) / qpt, ntiles = (C2 + kMlpNT - 1) / kMlpNT;
    const long long blocks = (mtiles * ntiles + 7) / 8 * 8;
    if (blocks > 0x7FFFFFFFLL) {
        set_error("%s: too many tiles", who);
        return TGN_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    int *err = idx ? index_error_word(st) : nullptr;
    if (direct && !points) points = xyz;   // D == 0: never read
#define TGN_MLP2(IT, DIR)                                                                                                 \
    if (lds > 48 * 1024)                                                                                                  \
        (void)hipFuncSetAttribute((const void *)sa_mlp2_max_kernel<IT, DIR>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  80 * 1024);                                                                             \
    hipLaunchKernelGGL((sa_mlp2_max_kernel<IT, DIR>), dim3((unsigned)blocks), dim3(256), lds, st, Q, N, S, K, D, C1p, C2, out_stride, \
                       A1, xyz, points, new_xyz, W1, b1, (const IT *)idx, W2f, b2, out, err)
#define TGN_MLP2S(IT, DIR, WM_, TN_, LDS_, BLOCKS_)                                                                                \
    if ((LDS_) > 48 * 1024)                                                                                                        \
        (void)hipFuncSetAttribute((const void *)sa_mlp2_max_split_kernel<IT, DIR, WM_, TN_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)(LDS_));                                                                                    \
    hipLaunchKernelGGL((sa_mlp2_max_split_kernel<IT, DIR, WM_, TN_>), dim3((unsigned)(BLOCKS_)), dim3(WM_ * 128), (LDS_), st, Q, N, S, K, D, \
                       C1p, C2, out_stride, A1, xyz, points, new_xyz, W1, b1, (const IT *)idx, (const unsigned char *)W2f, b2, out, err)
    if (split) {
        if (ntiles * (long long)(C1p / kMlpKT) * kSplitTile > 0x7FFFFFFFLL || ((uintptr_t)W2f & 15)) {
            set_error("%s: split weight image too large or misaligned", who);
            return TGN_ERR_UNSUPPORTED;
        }
        // 256 x 256 tiles (512 threads, one workgroup per CU) where the level is wide and tall enough to fill the chip with them:
        // half the weight and gather traffic per flop of the 128 x 128 form, which the memory system bounds ("sa_tile": 0 picks,
        // 128 / 256 force)
        const int qpt_big = K > 32 ? 4 : 8;
        const long long mt_big = (Q + qpt_big - 1) / qpt_big, nt_big = (C2 + 255) / 256;
        const size_t lds_big = (size_t)4 * 2 * kSplitTile + (size_t)qpt_big * C1p * sizeof(float);
        const int forced = tuning(kTuneSaTile);
        const bool big = !direct && lds_big <= 150 * 1024 && forced != 128 &&
                         (forced == 256 || (C2 % 256 == 0 && mt_big * nt_big >= 256));
        if (big) {
            const long long blocks_big = (mt_big * nt_big + 7) / 8 * 8;
            if (idx_is_int64) {
                TGN_MLP2S(long long, false, 4, 4, lds_big, blocks_big);
            } else {
                TGN_MLP2S(int, false, 4, 4, lds_big, blocks_big);
            }
        } else if (idx_is_int64) {
            if (direct) {
                TGN_MLP2S(long long, true, 2, 2, lds, blocks);
            } else {
                TGN_MLP2S(long long, false, 2, 2, lds, blocks);
            }
        } else {
            if (direct) {
                TGN_MLP2S(int, true, 2, 2, lds, blocks);
            } else {
                TGN_MLP2S(int, false, 2, 2, lds, blocks);
            }
        }
        return check_launch("sa_mlp2_max_split_kernel");
    }
    if (idx_is_int64) {
        if (direct) {
            TGN_MLP2(long long, true);
        } else {
            TGN_MLP2(long long, false);
        }
    } else {
        if (direct) {
            TGN_MLP2(int, true);
        } else {
            TGN_MLP2(int, false);
        }
    }
#undef TGN_MLP2
#undef TGN_MLP2S
    return check_launch("sa_mlp2_max_kernel");
}

// Bytes of the split (bf16 x 3) image of a second-layer weight matrix, and the one-off conversion (device to device, on `stream`).
TGN_API size_t tgn_sa_mlp2_split_bytes(int C1p, int C2) {
    if (C1p < 16 || (C1p & 15) || C2 < 1) return 0;
    return (size_t)((C2 + kMlpNT - 1) / kMlpNT) * (size_t)(C1p / kMlpKT) * kSplitTile;
}

TGN_API int tgn_sa_mlp2_split_weights(int C1p, int C2, const float *W2f, void *W2s, tgn_stream_t stream) {
    if (!W2f || !W2s || C1p < 16 || (C1p & 15) || C2 < 1 || ((uintptr_t)W2s & 15)) {
        set_error("tgn_sa_mlp2_split_weights: needs W2f (C1p/8, C2, 8) with C1p a multiple of 16 and a 16-byte aligned image");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    const long long chunks = (long long)((C2 + kMlpNT - 1) / kMlpNT) * (C1p / kMlpKT) * 256;
    hipLaunchKernelGGL(sa_mlp2_split_weights_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, C1p, C2,
                       W2f, (unsigned char *)W2s);
    return check_launch("sa_mlp2_split_weights_kernel");
}

// tgn_sa_mlp2_max with the second layer on the bf16 matrix cores at fp32 accuracy (six bf16 products per fp32 product, see the
// kernel): W2s = tgn_sa_mlp2_split_weights(W2f).  Same arguments and results otherwise (fp32-class rounding, not bit-identical).
TGN_API int tgn_sa_mlp2_max_bf16x3(int B, int N, int S, int K, int D, int C1p, int C2, const float *A1, const float *xyz,
                                   const float *points, const float *new_xyz, const float *W1, const float *b1, const void *idx,
                                   int idx_is_int64, const void *W2s, const float *b2, float *out, int out_stride, tgn_stream_t stream) {
    if ((long long)B * S <= 0 || C2 <= 0) return TGN_OK;
    if (out_stride <= 0) out_stride = C2;
    if (!new_xyz || !W1 || !idx || !W2s) {
        set_error("tgn_sa_mlp2_max_bf16x3: null pointer");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    return sa_mlp2_launch("tgn_sa_mlp2_max_bf16x3", B, N, S, K, D, C1p, C2, A1, xyz, points, new_xyz, W1, b1, idx, idx_is_int64,
                          (const float *)W2s, b2, out, out_stride, stream, true);
}

TGN_API int tgn_sa_mlp2_max(int B, int N, int S, int K, int D, int C1p, int C2, const float *A1, const float *xyz,
                            const float *points, const float *new_xyz, const float *W1, const float *b1, const void *idx,
                            int idx_is_int64, const float *W2f, const float *b2, float *out, int out_stride, tgn_stream_t stream) {
    if ((long long)B * S <= 0 || C2 <= 0) return TGN_OK;
    if (out_stride <= 0) out_stride = C2;
    if (!new_xyz || !W1 || !idx) {
        set_error("tgn_sa_mlp2_max: null pointer");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    return sa_mlp2_launch("tgn_sa_mlp2_max", B, N, S, K, D, C1p, C2, A1, xyz, points, new_xyz, W1, b1, idx, idx_is_int64, W2f, b2, out,
                          out_stride, stream);
}

// group_all set abstraction (pointnet2_utils.py:178-195 + 214-239 with group_all=True; tsg_seg_module.py:28 is the one
// instantiation): out[b,:] = max over ALL N points of relu(bn2(W2 * relu(bn1(W1 * [x, f] + bias1)) + bias2)), no centre, no
// index tensor.  The cloud is cut into chunks of 64 (N <= 32: one chunk of 32) consecutive points, every chunk runs through the
// two-layer kernel above as if it were a ball, and a second small launch takes the maximum over a cloud's chunks.
//   A1 (B,N,C1p): per-point first layer before bias (tgn_sa_point_transform), or NULL for the direct form (3+D <= 16: xyz,
//   points, Wd (16,C1p)); part: workspace of tgn_sa_all_chunks(N) * B * C2 floats (unused when a cloud is one chunk).
TGN_API int tgn_sa_all_chunks(int N) { return N <= 64 ? 1 : (N + 63) / 64; }

TGN_API int tgn_sa_all_mlp2_max(int B, int N, int D, int C1p, int C2, const float *A1, const float *xyz, const float *points,
                                const float *Wd, const float *b1, const float *W2f, const float *b2, float *part, float *out,
                                int out_stride, tgn_stream_t stream) {
    if (B <= 0 || C2 <= 0) return TGN_OK;
    if (N < 1) {
        set_error("tgn_sa_all_mlp2_max: empty clouds");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    if (out_stride <= 0) out_stride = C2;
    const int S = tgn_sa_all_chunks(N), K = N < 64 ? N : 64;
    if (S > 1 && !part) {
        set_error("tgn_sa_all_mlp2_max: null workspace");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    int rc = sa_mlp2_launch("tgn_sa_all_mlp2_max", B, N, S, K, D, C1p, C2, A1, xyz, points, nullptr, Wd, b1, nullptr, 0, W2f, b2,
                            S > 1 ? part : out, S > 1 ? C2 : out_stride, stream);
    if (rc != TGN_OK || S == 1) return rc;
    const long long n = (long long)B * C2;
    hipLaunchKernelGGL(sa_chunks_max_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, S, C2, out_stride,
                       part, out);
    return check_launch("sa_chunks_max_kernel");
}

// tgn_sa_point_transform on the bf16 matrix cores at fp32 accuracy (the scheme of tgn_sa_mlp2_max_bf16x3): Wts = the split image of Wt
// -- tgn_sa_mlp2_split_weights(Kp, C1, Wtf, Wts) with Wtf (Kp/8, C1, 8), Wtf[kb][c][i] = Wt[8 kb + i][c], zero rows past D + 3;
// Kp = D + 3 rounded up to a multiple of 16.  A (M, C1) as tgn_sa_point_transform writes it (fp32-class rounding, not bit-identical).
TGN_API int tgn_sa_point_transform_bf16x3(long long M, int D, int Kp, int C1, const float *xyz, const float *points, const void *Wts,
                                          float *A, tgn_stream_t stream) {
    if (M <= 0 || C1 <= 0) return TGN_OK;
    if (!xyz || !Wts || !A || (D > 0 && !points) || D < 0 || Kp < D + 3 || (Kp & 15) || ((uintptr_t)Wts & 15)) {
        set_error("tgn_sa_point_transform_bf16x3: null pointer, or Kp is not D + 3 rounded up to a multiple of 16");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    const long long by = (M + kMlpMT - 1) / kMlpMT;
    const long long nt = (C1 + kMlpNT - 1) / kMlpNT;
    if (by > 65535 || nt * (Kp / kMlpKT) * kSplitTile > 0x7FFFFFFFLL) {   // (grid.y; the fp32 form has the same bound)
        set_error("tgn_sa_point_transform_bf16x3: too many rows (> 65535 x 128) or too large a weight image");
        return TGN_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(sa_point_transform_split_kernel, dim3((unsigned)nt, (unsigned)by), dim3(256), 0, (hipStream_t)stream, M, D, Kp, C1,
                       xyz, points, (const unsigned char *)Wts, A);
    return check_launch("sa_point_transform_split_kernel");
}

// group.hip -- group_points: the gather + centre + concat of sample_and_group (pointnet2_utils.py:162-169,
// [xyz[idx]-new_xyz, points[idx]]) and of PointNetSetAbstractionMsg (pointnet2_utils.py:281-285,
// [points[idx], xyz[idx]-new_xyz]) as ONE kernel that writes the grouped (B,S,K,3+D) tensor exactly once.
//
// The kernel is a store stream (K*(3+D) floats per query, 4.4 GB per 256 scans at the Shape-A levels 2 and 3)
// fed by a gather that must hit L2: a scan's feature block (N*D*4 B = 2 MB) is re-read S*K/N = 8 times.
// What decides its speed (profiles/r01_store_bench.txt, r01_pmc_group_l2.txt, r02_*):
//   * bytes in flight per wave: one 4-B gather per iteration is latency-bound (1.6-3 TB/s); v2 keeps FOUR
//     256-B gathers in flight per 1-KiB step;
//   * the stores: a query's K*C floats are one contiguous, 16-B aligned region (128-B aligned for K = 32/64), so
//     v2 transposes each 1-KiB step through LDS and stores 16 B per lane -- whole lines, and wide enough that
//     write-through (`sc1`) stores cost no more than plain ones.  The store policy is a template parameter
//     because it decides what the XCD's L2 keeps: plain / nt stores allocate the output lines in L2 and evict
//     the feature block under 17 MB of output per scan; sc1 stores leave L2 to the gather source;
//   * how many scans an XCD has in flight: each XCD walks ONE contiguous range of queries with a bounded
//     number of resident blocks, so all its waves gather from the same scan's feature block.
#include "tgn_common.h"

#include <stdlib.h>

namespace tgn {

constexpr int kGroupMaxK = 128;

// Index of neighbour k: negative values wrap like torch's advanced indexing (pointnet2_utils.py:56-60 on the
// reference side); anything still outside [0,N) -- the reference raises there, e.g. an empty ball yields N --
// reads point 0 and latches the device error word (tgn_take_index_error).
template <typename IdxT>
__device__ __forceinline__ unsigned checked_index(IdxT raw, int N, bool &bad) {
    long long v = (long long)raw;
    if (v < 0) v += N;
    if (v < 0 || v >= N) {
        bad = true;
        v = 0;
    }
    return (unsigned)v;
}

// ---- v1: 4 B per lane, one gather in flight (kept for shapes v2 does not take: K*(3+D) not a multiple of 4) ---
template <typename IdxT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(24))) void group_points_kernel(
    long long queries, int N, int S, int K, int D, unsigned magicC, const float *__restrict__ xyz,
    const float *__restrict__ new_xyz, const float *__restrict__ points, const IdxT *__restrict__ idx, int xyz_first,
    float *__restrict__ out, int *__restrict__ err) {
    __shared__ unsigned sfb[4][kGroupMaxK];
    __shared__ float srel[4][kGroupMaxK * 3];
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int C = 3 + D;
    const unsigned xo = xyz_first ? 0 : D;
    const unsigned fo = xyz_first ? 3 : 0;
    const int total = K * C;
    const unsigned nb = gridDim.x;  // multiple of 8; hardware block i runs on XCD i % 8 (observed; speed only)
    const unsigned lb = (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3);
    for (long long q = (long long)lb * 4 + wv; q < queries; q += (long long)nb * 4) {
        const int b = (int)(q / S);
        const float cq0 = new_xyz[q * 3 + 0], cq1 = new_xyz[q * 3 + 1], cq2 = new_xyz[q * 3 + 2];
        const size_t pbase = (size_t)b * N;
        bool bad = false;
        const IdxT *__restrict__ qidx = idx + q * K;
        for (int k = lane; k < K; k += kWave) {
            const unsigned pv = (unsigned)pbase + checked_index(qidx[k], N, bad);
            sfb[wv][k] = pv * (unsigned)D;
            srel[wv][k * 3 + 0] = xyz[pv * 3u + 0u] - cq0;
            srel[wv][k * 3 + 1] = xyz[pv * 3u + 1u] - cq1;
            srel[wv][k * 3 + 2] = xyz[pv * 3u + 2u] - cq2;
        }
        if (err && __any(bad) && lane == 0) atomicOr(err, 1);
        float *__restrict__ dst = out + (size_t)q * total;
#pragma unroll 1
        for (int e = lane; e < total; e += kWave) {
            const unsigned k = __umulhi((unsigned)e, magicC);
            const unsigned c = (unsigned)e - k * (unsigned)C;
            const unsigned cx = c - xo;
            const bool isx = cx < 3u;
            const float rel = srel[wv][k * 3 + (isx ? cx : 0u)];
            const float ld = points[sfb[wv][k] + (isx ? 0u : c - fo)];
            dst[e] = isx ? rel : ld;
        }
    }
}

// ---- v2 -------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Raw buffer descriptor (gfx9 family): 48-bit base, stride 0, num_records in bytes, dword3 = 0x00020000 (untyped
// 32-bit data format).  Accesses beyond num_records are dropped / return 0 in hardware.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

// POLICY = the cache-policy immediate of the output stores: bit 0 sc0, bit 1 nt, bit 4 sc1 (gfx940+).
// WIDE = rows of at least 64 floats and K <= 64 (Shape-A levels 2 and 3): a 64-float sub-step then lies in one row or
// straddles exactly one row boundary, and (row, channel) of its first float are tracked in SGPRs -- no per-lane
// division; !WIDE (level 1, C = 9) computes (k, c) per lane.
// Few VGPRs on purpose: beside an FPS level-1 workgroup (2 waves x 232 of a SIMD's 512 registers) 48 are left per
// SIMD lane (tests/test_build_resources.py guards the numbers).
template <typename IdxT, int POLICY, bool WIDE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(24))) void group_points_v2_kernel(
    long long queries, long long q_per_xcd, int N, int S, int K, int D, unsigned magicC,
    const float *__restrict__ xyz, const float *__restrict__ new_xyz, const float *__restrict__ points,
    const IdxT *__restrict__ idx, int xyz_first, float *__restrict__ out, int *__restrict__ err) {
    __shared__ unsigned sfb[4][kGroupMaxK];                            // per neighbour: element offset of its feature row
    __shared__ float srel[4][kGroupMaxK * 3];                          // per neighbour: centred coordinates
    __shared__ __attribute__((aligned(16))) float stage[4][256];       // one 1-KiB step of the output, per wave
    const unsigned lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const unsigned C = 3u + (unsigned)D;
    const unsigned xo = xyz_first ? 0u : (unsigned)D;   // first channel of the relative coordinates
    const unsigned fo = xyz_first ? 3u : 0u;            // first channel of the features
    const unsigned total = (unsigned)K * C;             // multiple of 4 (launcher)
    const unsigned units = total >> 2;                  // 16-B units of the query's region
    const unsigned lane4 = lane * 4u;
    // XCD x (hardware block i runs on XCD i % 8: observed dispatch, speed only) owns the contiguous query range
    // [x*q_per_xcd, (x+1)*q_per_xcd) -- whole scans when there are at least 8 -- and walks it with its gridDim/8
    // blocks, so the waves of an XCD gather from as few feature blocks as possible at any time.
    const unsigned x = blockIdx.x & 7u, j = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    long long q1 = (long long)(x + 1) * q_per_xcd;
    if (q1 > queries) q1 = queries;
    for (long long q = (long long)x * q_per_xcd + (long long)j * 4 + wv; q < q1; q += (long long)nbx * 4) {
        const int b = (int)(q / S);
        const float cq0 = new_xyz[q * 3 + 0], cq1 = new_xyz[q * 3 + 1], cq2 = new_xyz[q * 3 + 2];
        const __amdgpu_buffer_rsrc_t rs_xyz = make_rsrc(xyz + (size_t)b * N * 3, (unsigned)N * 12u);
        const __amdgpu_buffer_rsrc_t rs_idx = make_rsrc(idx + q * K, (unsigned)K * (unsigned)sizeof(IdxT));
        bool bad = false;
        unsigned r0 = 0;  // lane k: feature-row offset of neighbour k (WIDE: K <= 64)
        for (unsigned k = lane; k < (unsigned)K; k += kWave) {
            IdxT raw;
            if constexpr (sizeof(IdxT) == 8) {
                const auto t = __builtin_amdgcn_raw_buffer_load_b64(rs_idx, k * 8u, 0, 0);
                raw = (IdxT)(((unsigned long long)t[1] << 32) | t[0]);
            } else {
                raw = (IdxT)__builtin_amdgcn_raw_buffer_load_b32(rs_idx, k * 4u, 0, 0);
            }
            const unsigned v = checked_index(raw, N, bad);
            const unsigned ro = v * (unsigned)D;
            sfb[wv][k] = ro;
            r0 = ro;
            const auto p = __builtin_amdgcn_raw_buffer_load_b96(rs_xyz, v * 12u, 0, 0);
            srel[wv][k * 3 + 0] = __builtin_bit_cast(float, p[0]) - cq0;
            srel[wv][k * 3 + 1] = __builtin_bit_cast(float, p[1]) - cq1;
            srel[wv][k * 3 + 2] = __builtin_bit_cast(float, p[2]) - cq2;
        }
        if (err && __any(bad) && lane == 0) atomicOr(err, 1);
        // (the same wave wrote the tables: LDS operations of one wave complete in order, no barrier needed)
        const __amdgpu_buffer_rsrc_t rs_pts = make_rsrc(points + (size_t)b * N * D, (unsigned)N * (unsigned)D * 4u);
        const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(out + (size_t)q * total, total * 4u);
        // One step = 256 consecutive output floats = four 64-float sub-steps, each ONE 256-B gather; all four gathers
        // are in flight before any is consumed, then the step goes through LDS and leaves as 16 B per lane.
        if constexpr (WIDE) {
            unsigned k0 = 0, c0 = 0;  // SGPRs: row and first channel of the next sub-step
#pragma unroll 1
            for (unsigned u0 = 0; u0 < units; u0 += kWave) {
                float v[4];
                unsigned mk[4], mc[4];  // SGPRs: (k0, c0) of the sub-steps that touch a boundary; mc = ~0: plain
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const unsigned kk = k0 & 63u;   // (past the end of the query: garbage rows, lanes never stored)
                    const unsigned rowA = (unsigned)__builtin_amdgcn_readlane((int)r0, (int)kk);
                    const unsigned sA = (rowA + c0 - fo) * 4u;
                    unsigned voff = lane4, soff = sA;
                    mk[s] = kk;
                    mc[s] = ~0u;
                    if (!(c0 >= fo && c0 + 64u <= fo + (unsigned)D)) {
                        // lanes >= C - c0 belong to row k0 + 1, channel c0 + lane - C.  The hardware range-checks
                        // voff only; a coordinate lane's offset may come out "negative": it reads 0 and is patched.
                        const unsigned rowB = (unsigned)__builtin_amdgcn_readlane((int)r0, (int)((kk + 1u) & 63u));
                        const unsigned sB = (rowB + c0 - C - fo) * 4u;
                        voff = lane4 + (lane >= C - c0 ? sB : sA);
                        soff = 0;
                        mc[s] = c0;
                    }
                    v[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_pts, voff, soff, 0));
                    c0 += 64u;
                    if (c0 >= C) {
                        c0 -= C;
                        ++k0;
                    }
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) stage[wv][s * 64 + lane] = v[s];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (mc[s] == ~0u) continue;  // wave-uniform
                    const bool wrap = lane >= C - mc[s];
                    const unsigned cx = mc[s] + lane - (wrap ? C : 0u) - xo;
                    const unsigned k = mk[s] + (wrap ? 1u : 0u);
                    if (cx < 3u && k < (unsigned)K) stage[wv][s * 64 + lane] = srel[wv][k * 3u + cx];
                }
                const f32x4 w = *(const f32x4 *)&stage[wv][lane * 4];
                const unsigned u = u0 + lane;
                if (u < units)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, w),
                                                           rs_out, u * 16u, 0, POLICY);
            }
        } else {
#pragma unroll 1
            for (unsigned u0 = 0; u0 < units; u0 += kWave) {
                float v[4], rel[4];
                bool isx[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    v[s] = rel[s] = 0.0f;
                    isx[s] = false;
                    if (u0 * 4u + (unsigned)s * 64u >= total) continue;  // wave-uniform: nothing left in this step
                    const unsigned e = u0 * 4u + (unsigned)s * 64u + lane;
                    unsigned k = __umulhi(e, magicC);
                    const unsigned c = e - k * C;
                    k = k < (unsigned)K ? k : (unsigned)K - 1u;
                    const unsigned cx = c - xo;
                    isx[s] = cx < 3u;
                    rel[s] = srel[wv][k * 3u + (isx[s] ? cx : 0u)];
                    const unsigned off = sfb[wv][k] + (isx[s] ? 0u : c - fo);
                    v[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_pts, off * 4u, 0, 0));
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) stage[wv][s * 64 + lane] = isx[s] ? rel[s] : v[s];
                const f32x4 w = *(const f32x4 *)&stage[wv][lane * 4];
                const unsigned u = u0 + lane;
                if (u < units)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, w),
                                                           rs_out, u * 16u, 0, POLICY);
            }
        }
    }
}

static int env_int(const char *name, int dflt) {
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

}  // namespace tgn

using namespace tgn;

// impl: 0 = choose, 1 = v1 (4-B stores), 2 = v2 (16-B stores).  store_policy: cache-policy bits of the v2 output
// stores (0 plain, 2 nt, 16 sc1, 17 sc0|sc1, 18 sc1|nt), -1 = default.  max_blocks: upper bound on the v2 grid
// (0 = no bound): a caller that overlaps the grouping with a register-hungry kernel keeps CUs free this way.
TGN_API int tgn_group_points_ex(int B, int N, int S, int K, int D, const float *xyz, const float *new_xyz,
                                const float *points, const void *idx, int idx_is_int64, int xyz_first, float *out,
                                int impl, int store_policy, int max_blocks, tgn_stream_t stream) {
    const long long queries = (long long)B * S;
    if (queries <= 0 || K <= 0) return TGN_OK;
    if (!xyz || !new_xyz || !idx || !out) {
        set_error("tgn_group_points: null pointer");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    if (!points) D = 0;
    if (K > kGroupMaxK || D < 0 || (long long)K * (3 + D) >= (1LL << 31) / (3 + D) ||
        (long long)B * N * (D > 3 ? D : 3) >= (1LL << 32)) {
        set_error("tgn_group_points: nsample %d / channels %d out of the supported range", K, 3 + D);
        return TGN_ERR_UNSUPPORTED;
    }
    static const int env_impl = env_int("TGN_GROUP_IMPL", 0);
    static const int env_policy = env_int("TGN_GROUP_POLICY", 16);
    static const int env_blocks = env_int("TGN_GROUP_MAX_BLOCKS", 0);
    if (impl <= 0) impl = env_impl;
    if (store_policy < 0) store_policy = env_policy;
    if (max_blocks <= 0) max_blocks = env_blocks;
    int *err = index_error_word();
    const int C = 3 + D;
    const unsigned magicC = (unsigned)((0x100000000ULL + C - 1) / C);
    const float *pts = points ? points : xyz;
    hipStream_t st = (hipStream_t)stream;
    const bool v2_ok = ((long long)K * C) % 4 == 0 && ((uintptr_t)out & 15) == 0 && (long long)N * (D > 0 ? D : 1) < (1LL << 30);
    if (impl == 2 && !v2_ok) impl = 1;
    if (impl == 0) impl = v2_ok ? 2 : 1;
    if (impl == 1) {
        long long blocks = ((queries + 3) / 4 + 7) / 8 * 8;  // one query per wave, grid a multiple of the 8 XCDs
        if (blocks > (1LL << 30)) blocks = 1LL << 30;
        if (idx_is_int64)
            hipLaunchKernelGGL((group_points_kernel<long long>), dim3((unsigned)blocks), dim3(256), 0, st, queries, N, S,
                               K, D, magicC, xyz, new_xyz, pts, (const long long *)idx, xyz_first, out, err);
        else
            hipLaunchKernelGGL((group_points_kernel<int>), dim3((unsigned)blocks), dim3(256), 0, st, queries, N, S, K, D,
                               magicC, xyz, new_xyz, pts, (const int *)idx, xyz_first, out, err);
        return check_launch("group_points_kernel");
    }
    // v2: per-XCD query ranges (whole scans when B >= 8), at most 8 blocks per CU resident
    long long q_per_xcd = B >= 8 ? (long long)((B + 7) / 8) * S : ((queries + 7) / 8 + 3) / 4 * 4;
    long long nbx = (q_per_xcd + 3) / 4;
    long long cap = 256;  // 32 CUs x 8 blocks per XCD
    if (max_blocks > 0 && max_blocks / 8 < cap) cap = max_blocks / 8 > 0 ? max_blocks / 8 : 1;
    if (nbx > cap) nbx = cap;
    const dim3 grid((unsigned)(nbx * 8));
#define TGN_GROUP_V2(IT, POL)                                                                                            \
    do {                                                                                                                 \
        if (C >= 64 && K <= 64)                                                                                          \
            hipLaunchKernelGGL((group_points_v2_kernel<IT, POL, true>), grid, dim3(256), 0, st, queries, q_per_xcd, N,   \
                               S, K, D, magicC, xyz, new_xyz, pts, (const IT *)idx, xyz_first, out, err);                \
        else                                                                                                             \
            hipLaunchKernelGGL((group_points_v2_kernel<IT, POL, false>), grid, dim3(256), 0, st, queries, q_per_xcd, N,  \
                               S, K, D, magicC, xyz, new_xyz, pts, (const IT *)idx, xyz_first, out, err);                \
    } while (0)
#define TGN_GROUP_V2_POL(IT)                                   \
    switch (store_policy) {                                    \
        case 0: TGN_GROUP_V2(IT, 0); break;                    \
        case 2: TGN_GROUP_V2(IT, 2); break;                    \
        case 17: TGN_GROUP_V2(IT, 17); break;                  \
        case 18: TGN_GROUP_V2(IT, 18); break;                  \
        default: TGN_GROUP_V2(IT, 16); break;                  \
    }
    if (idx_is_int64) {
        TGN_GROUP_V2_POL(long long)
    } else {
        TGN_GROUP_V2_POL(int)
    }
#undef TGN_GROUP_V2_POL
#undef TGN_GROUP_V2
    return check_launch("group_points_v2_kernel");
}

TGN_API int tgn_group_points(int B, int N, int S, int K, int D, const float *xyz, const float *new_xyz,
                             const float *points, const void *idx, int idx_is_int64, int xyz_first, float *out,
                             tgn_stream_t stream) {
    return tgn_group_points_ex(B, N, S, K, D, xyz, new_xyz, points, idx, idx_is_int64, xyz_first, out, 0, -1, 0, stream);
}

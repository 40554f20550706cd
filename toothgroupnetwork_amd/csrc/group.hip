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
// The same for a base the caller KNOWS to be wave-uniform but hipcc may not (anything downstream of an integer
// division): both halves go through v_readfirstlane, so the descriptor sits in SGPRs and the access is not wrapped in
// a waterfall loop (cdna_hip_programming.md T20).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_uniform(const void *base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
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
            // (three dword loads: hipcc 7.2 narrows a raw_buffer_load_b96 whose lanes are used separately to ONE dword)
            const float px = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_xyz, v * 12u, 0, 0));
            const float py = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_xyz, v * 12u + 4u, 0, 0));
            const float pz = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_xyz, v * 12u + 8u, 0, 0));
            srel[wv][k * 3 + 0] = px - cq0;
            srel[wv][k * 3 + 1] = py - cq1;
            srel[wv][k * 3 + 2] = pz - cq2;
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

// ---- explicit vmcnt waits.  hipcc does not order a ds_read behind a pending LDS-DMA (MI355X_MICROARCH.md): the waits are
// written by hand.  On gfx9 the vector-memory operations of a wave retire in issue order on ONE counter, loads and stores
// alike, so "X has landed" == "at most (operations issued after X) outstanding".
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70);  // expcnt / lgkmcnt fields: no wait
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wait_vmcnt_dyn(unsigned n) {  // wave-uniform n <= 15
    switch (n) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<1>(); break;
        case 2: wait_vmcnt<2>(); break;
        case 3: wait_vmcnt<3>(); break;
        case 4: wait_vmcnt<4>(); break;
        case 5: wait_vmcnt<5>(); break;
        case 6: wait_vmcnt<6>(); break;
        case 7: wait_vmcnt<7>(); break;
        case 8: wait_vmcnt<8>(); break;
        case 9: wait_vmcnt<9>(); break;
        case 10: wait_vmcnt<10>(); break;
        case 11: wait_vmcnt<11>(); break;
        case 12: wait_vmcnt<12>(); break;
        case 13: wait_vmcnt<13>(); break;
        case 14: wait_vmcnt<14>(); break;
        default: wait_vmcnt<15>(); break;
    }
}

#define TGN_WAIT16(NAME, BASE)                                                                                     \
    __device__ __forceinline__ void NAME(unsigned n) {                                                             \
        switch (n - BASE) {                                                                                        \
            case 0: wait_vmcnt<BASE + 0>(); break;   case 1: wait_vmcnt<BASE + 1>(); break;                        \
            case 2: wait_vmcnt<BASE + 2>(); break;   case 3: wait_vmcnt<BASE + 3>(); break;                        \
            case 4: wait_vmcnt<BASE + 4>(); break;   case 5: wait_vmcnt<BASE + 5>(); break;                        \
            case 6: wait_vmcnt<BASE + 6>(); break;   case 7: wait_vmcnt<BASE + 7>(); break;                        \
            case 8: wait_vmcnt<BASE + 8>(); break;   case 9: wait_vmcnt<BASE + 9>(); break;                        \
            case 10: wait_vmcnt<BASE + 10>(); break; case 11: wait_vmcnt<BASE + 11>(); break;                      \
            case 12: wait_vmcnt<BASE + 12>(); break; case 13: wait_vmcnt<BASE + 13>(); break;                      \
            case 14: wait_vmcnt<BASE + 14>(); break; default: wait_vmcnt<BASE + 15>(); break;                      \
        }                                                                                                          \
    }
TGN_WAIT16(wait_vmcnt_dyn2, 16)
TGN_WAIT16(wait_vmcnt_dyn3, 32)
TGN_WAIT16(wait_vmcnt_dyn4, 48)
#undef TGN_WAIT16
__device__ __forceinline__ void wait_vmcnt_any(unsigned n) {   // wave-uniform n; n > 63 cannot be outstanding
    if (n >= 63u) return;
    if (n >= 48u) wait_vmcnt_dyn4(n);
    else if (n >= 32u) wait_vmcnt_dyn3(n);
    else if (n >= 16u) wait_vmcnt_dyn2(n);
    else wait_vmcnt_dyn(n);
}

// ---- v5 (wide rows): row pieces straight into an LDS image of the output, ONE wave per workgroup --------------------
// Measured in round 2 (profiles/r02_group_*.txt, r02_gather_bench.txt): with their stores removed the round-1 style kernels
// still needs 1.3-1.9 ms for 4.3 GB, while a bare LDS-DMA gather loop reaches 6-7 TB/s from HBM with only 4-8 waves
// per CU -- the grouping kernels were bound by their own instruction streams (~170 mostly scalar, mostly dependent
// instructions per KiB: the scalar unit of a CU saturates at full occupancy, a lone wave crawls at low occupancy), not
// by memory.  This version spends ~35 instructions per KiB:
//   * the output of R consecutive neighbours (R*C floats, R a multiple of 4) is assembled in LDS exactly as it will
//     lie in memory;
//   * a neighbour's feature row arrives as ceil(D/64) LDS-DMA pieces of 64 floats, each ONE instruction whose operands
//     are an SGPR offset and M0 bumped by 256 B -- no per-lane address arithmetic, no row-boundary cases;
//   * the 3 centred coordinates of each of the R rows are dropped in by one masked ds_write;
//   * an image leaves as 16 B per lane and in WHOLE 128-B LINES: it is streamed from the line boundary at or before
//     its first float to the last line boundary inside it, and the <= 31 floats beyond are carried over into the head
//     of the next image (write-through stores of partial lines cost a fabric write each: the level-3 images start
//     48 B into a line);
//   * two images per wave, the next one loading while this one is stored; the next QUERY's index row and coordinates
//     are fetched underneath the current query, so a wave never sits through their two dependent round trips.
// The vector-memory operations of a wave retire in issue order on one counter (gfx9), loads and stores alike, so
// "X has landed" is "at most (operations issued after X) outstanding" -- acknowledgements of stores are never waited
// for.
// W = dwords per lane of a feature-row piece: 4 (`buffer_load_dwordx4 ... lds`, gfx950: 1 KiB per instruction; it takes
// LDS destinations and sources that are only 4-byte aligned -- tools/dma_test/lds_dma_align.hip) when D % 4 == 0, else 1.
// A level-2 row (128 floats) is ONE instruction instead of three, a level-3 row two instead of nine: the kernel was bound
// by the issue cost of its LDS-DMA instructions (~60-180 cycles each), not by bytes in flight.
template <typename IdxT, int POLICY, int W = 1>
__global__ __launch_bounds__(64) void group_points_rows_kernel(
    long long queries, long long q_per_xcd, int N, int S, int K, int D, int R, const float *__restrict__ xyz,
    const float *__restrict__ new_xyz, const float *__restrict__ points, const IdxT *__restrict__ idx, int xyz_first,
    float *__restrict__ out, int *__restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) float lds_img[];   // [2][32 + R*C] images, [64*3] centred coordinates, [5][64] set-up planes
    const unsigned lane = threadIdx.x;
    const unsigned C = 3u + (unsigned)D;
    const unsigned xo = xyz_first ? 0u : (unsigned)D;
    const unsigned fo = xyz_first ? 3u : 0u;
    const unsigned IMG = 32u + (unsigned)R * C;          // floats per image buffer (multiple of 4)
    float *const srel = lds_img + 2u * IMG;
    constexpr unsigned FP = 64u * (unsigned)W;           // floats per piece
    const unsigned pieces = ((unsigned)D + FP - 1u) / FP;   // pieces per feature row
    const unsigned last_len = ((unsigned)D - FP * (pieces - 1u)) / (unsigned)W;   // active lanes of a row's last piece
    const unsigned nb = ((unsigned)K + (unsigned)R - 1u) / (unsigned)R;
    const unsigned total = (unsigned)K * C;
    const unsigned max_stores = (IMG + 255u) / 256u;     // 1-KiB store instructions per image, at most
    const bool overlap = (unsigned)R * pieces + max_stores + 4u <= 62u;   // everything in flight fits the 6-bit counter
    const unsigned lane4 = lane * 4u * (unsigned)W;
    const unsigned pr = lane / 3u, pi = lane - pr * 3u;  // coordinate patch: lane -> (row of the image, axis)
    const unsigned x = blockIdx.x & 7u, j = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    long long q1 = (long long)(x + 1) * q_per_xcd;
    if (q1 > queries) q1 = queries;
    long long q = (long long)x * q_per_xcd + j;
    if (q >= q1) return;

    // the scan a query belongs to, as an SGPR (a 64-bit division comes out of the vector ALU: hipcc would wrap every
    // buffer access that depends on it in a waterfall loop)
    auto scan_of = [&](long long qq) { return __builtin_amdgcn_readfirstlane((int)((unsigned)qq / (unsigned)S)); };
    // Set-up data of a query -- its index row, then the coordinates of the K neighbours -- is fetched by LDS-DMA as
    // well (planes of 64 dwords): nothing arrives in a VGPR, so hipcc has no reason to drain the counter (it answers a
    // pending VGPR load with vmcnt(0), which would also wait for every store acknowledgement), and the two dependent
    // round trips run underneath the previous query: the index row next to its first image, the coordinates next to
    // the stores of its last one.
    constexpr unsigned NIDX = sizeof(IdxT) / 4;              // dwords per index
    float *const plane = srel + 64 * 3;                      // [NIDX] index planes, then x, y, z planes, 64 dwords each
    float *const plane_xyz = plane + 64 * NIDX;
    auto fetch_index = [&](long long qq) {
        const __amdgpu_buffer_rsrc_t rs_idx = make_rsrc_uniform(idx + qq * K, (unsigned)K * (unsigned)sizeof(IdxT));
#pragma unroll
        for (unsigned w = 0; w < NIDX; ++w)   // lanes >= K: out of range, LDS gets 0
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_idx, (__attribute__((address_space(3))) void *)(plane + 64 * w), 4,
                                                     lane * (unsigned)sizeof(IdxT) + 4u * w, 0, 0, 0);
    };
    auto read_index = [&](bool &bad_) -> unsigned {          // after the planes have landed
        long long raw;
        if constexpr (NIDX == 2)
            raw = (long long)(((unsigned long long)__float_as_uint(plane[64 + lane]) << 32) | __float_as_uint(plane[lane]));
        else
            raw = (long long)(int)__float_as_uint(plane[lane]);
        return lane < (unsigned)K ? checked_index(raw, N, bad_) : 0u;
    };
    auto fetch_xyz = [&](long long qq, unsigned vv) {
        const __amdgpu_buffer_rsrc_t rs_xyz = make_rsrc_uniform(xyz + (size_t)scan_of(qq) * N * 3, (unsigned)N * 12u);
#pragma unroll
        for (unsigned a_ = 0; a_ < 3; ++a_)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_xyz, (__attribute__((address_space(3))) void *)(plane_xyz + 64 * a_), 4,
                                                     vv * 12u + 4u * a_, 0, 0, 0);
    };
    bool bad = false;
    fetch_index(q);
    wait_vmcnt<0>();
    unsigned v = read_index(bad);
    fetch_xyz(q, v);
    unsigned tail_stores = 0;   // stores issued after a query's coordinate fetch: what may still be outstanding when it is needed
    for (;;) {
        const int b = scan_of(q);
        const float cq0 = new_xyz[q * 3 + 0], cq1 = new_xyz[q * 3 + 1], cq2 = new_xyz[q * 3 + 2];
        // the coordinates of THIS query have landed once at most the stores issued after their fetch are outstanding
        wait_vmcnt_dyn(tail_stores);
        const unsigned r0 = v * (unsigned)D * 4u;   // lane k: byte offset of neighbour k's feature row in the scan's block
        if (lane < (unsigned)K) {
            srel[lane * 3 + 0] = plane_xyz[lane] - cq0;
            srel[lane * 3 + 1] = plane_xyz[64 + lane] - cq1;
            srel[lane * 3 + 2] = plane_xyz[128 + lane] - cq2;
        }
        if (err && __any(bad) && lane == 0) atomicOr(err, 1);
        bad = false;
        const __amdgpu_buffer_rsrc_t rs_pts = make_rsrc_uniform(points + (size_t)b * N * D, (unsigned)N * (unsigned)D * 4u);
        const __amdgpu_buffer_rsrc_t rs_out = make_rsrc_uniform(out + (size_t)q * total, total * 4u);
        const long long qn = q + nbx;
        const bool has_next = qn < q1;
        auto issue = [&](unsigned ib) {   // image ib: neighbours ib*R .. ib*R + rows - 1, at float offset (ib*R*C) % 32 of its buffer
            const unsigned k0 = ib * (unsigned)R;
            float *img = lds_img + (ib & 1u) * IMG + (k0 * C & 31u);
            const unsigned rows = (unsigned)K - k0 < (unsigned)R ? (unsigned)K - k0 : (unsigned)R;
            auto piece = [&](float *dst, unsigned soff, unsigned p) {   // (the size operand must be a literal)
                if constexpr (W == 4)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_pts, (__attribute__((address_space(3))) void *)(dst + p * FP), 16,
                                                             lane4, soff + p * FP * 4u, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_pts, (__attribute__((address_space(3))) void *)(dst + p * FP), 4,
                                                             lane4, soff + p * FP * 4u, 0, 0);
            };
            if (pieces == 1u) {
                // one (partial) instruction per row -- level 2: the lane mask is set ONCE around the whole loop and the loop is
                // readlane + M0 + DMA (the general form below re-derives the mask and tests an empty inner loop per row: ~22
                // scalar instructions per 512 B, which is what a lone wave beside an FPS workgroup is bound by)
                if (lane < last_len) {
                    // rows is a multiple of 4 (R and K are): four rows per trip -- a taken branch costs a lone wave ~25 cycles
                    for (unsigned r = 0; r < rows; r += 4u) {
#pragma unroll
                        for (unsigned i = 0; i < 4u; ++i)
                            piece(img + (r + i) * C + fo, (unsigned)__builtin_amdgcn_readlane((int)r0, (int)(k0 + r + i)), 0u);
                    }
                }
                return rows;
            }
            if (pieces == 2u && last_len == 64u) {   // level 3 (512-float rows): two whole instructions per row, two rows per trip
                for (unsigned r = 0; r < rows; r += 2u) {
#pragma unroll
                    for (unsigned i = 0; i < 2u; ++i) {
                        const unsigned soff = (unsigned)__builtin_amdgcn_readlane((int)r0, (int)(k0 + r + i));
                        float *dst = img + (r + i) * C + fo;
                        piece(dst, soff, 0u);
                        piece(dst, soff, 1u);
                    }
                }
                return rows;
            }
            for (unsigned r = 0; r < rows; ++r) {
                const unsigned soff = (unsigned)__builtin_amdgcn_readlane((int)r0, (int)(k0 + r));
                float *dst = img + r * C + fo;
                for (unsigned p = 0; p + 1u < pieces; ++p) piece(dst, soff, p);
                if (lane < last_len) piece(dst, soff, pieces - 1u);
            }
            return rows;
        };
        unsigned rows_next = issue(0);
        if (has_next) fetch_index(qn);     // behind image 0, in front of image 1: lands underneath this query
        unsigned prev_stores = 0;          // store instructions of the previous image of this query
        for (unsigned ib = 0; ib < nb; ++ib) {
            const unsigned rows = rows_next;
            const bool last = ib + 1u == nb;
            const unsigned idx_ops = (ib == 0 && has_next) ? NIDX : 0u;   // issued after image 0's pieces
            if (overlap && !last) {
                rows_next = issue(ib + 1u);
                // image ib has landed <=> at most [what was issued after its pieces] is outstanding
                wait_vmcnt_any(rows_next * pieces + idx_ops + prev_stores);
            } else if (last && has_next && ib == 0) {
                wait_vmcnt<0>();           // single-image queries: the index row of the next query must be in as well
            } else {
                wait_vmcnt_any(overlap ? prev_stores : 0u);   // nothing was issued after image ib but those stores
            }
            if (last && has_next) {
                // everything issued before this query's last image has landed, the next query's index row included:
                // fetch its coordinates underneath the last image's stores
                v = read_index(bad);
                fetch_xyz(qn, v);
            }
            const unsigned k0 = ib * (unsigned)R;
            const unsigned e0 = k0 * C;                       // first float of the image in the query's region
            const unsigned a0 = e0 & ~31u;                    // ... rounded down to its 128-B line: the buffer's float 0
            float *buf = lds_img + (ib & 1u) * IMG;
            float *img = buf + (e0 - a0);
            if (pr < rows) img[pr * C + xo + pi] = srel[(k0 + pr) * 3u + pi];
            const unsigned e1 = e0 + rows * C;                // one past the image's last float
            const unsigned f1 = last ? e1 : (e1 & ~31u);      // stream whole lines; the rest rides with the next image
            if (!last && lane < e1 - f1)                      // carry: the floats beyond the last whole line -> head of the other buffer
                lds_img[((ib + 1u) & 1u) * IMG + lane] = buf[(f1 - a0) + lane];
            const unsigned units = (f1 - a0) >> 2;
            unsigned nst = 0;
            for (unsigned u = lane, u0 = 0; u0 < units; u += 64u, u0 += 64u, ++nst) {
                if (u < units) {
                    const f32x4 w = *(const f32x4 *)&buf[u * 4u];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, w),
                                                           rs_out, ((a0 >> 2) + u) * 16u, 0, POLICY);
                }
            }
            prev_stores = nst;
            if (!overlap && !last) rows_next = issue(ib + 1u);
        }
        if (!has_next) break;
        tail_stores = prev_stores;
        q = qn;
    }
}


// ---- pairs kernel (narrow rows: level 1, C = 3 + D <= 15) -------------------------------------------------------------
// The grouped tensor of a batch is ONE flat array of (query, neighbour) pairs, C floats each, in the order of the index
// tensor.  A lane owns a pair: it loads its index, gathers the point's coordinates and its D features (6 floats = three
// 8-byte loads), centres the coordinates and drops its finished row into an LDS image of the wave's 64 rows exactly as
// they lie in memory (row stride C floats, odd for the usual C = 9: conflict-free); the image -- 64*C*4 bytes, whole
// 128-B lines, line-aligned -- then leaves as 16 B per lane.  ~45 instructions per 2.3 KiB instead of the per-element
// walk of v2<WIDE = false> (5 four-byte gathers and a multiply-high per lane and KiB).  Two chunks per loop trip, all
// loads of both issued before either is consumed, and the index loads of the NEXT trip issued right behind them (one
// exposed round trip per trip instead of two); chunk = 64 consecutive pairs of one scan (S*K % 64 == 0, K a power of
// two, so a query never straddles lanes of different chunks in a way that matters: q = pair >> log2(K)).
// XCD x walks the contiguous chunk range [x*cpx, (x+1)*cpx): whole scans when B is a multiple of 8.
template <typename IdxT, int D, int POLICY>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(48))) void group_points_pairs_kernel(
    unsigned chunks, unsigned cpx, unsigned cps, int N, int log2K, const float *__restrict__ xyz,
    const float *__restrict__ new_xyz, const float *__restrict__ points, const IdxT *__restrict__ idx, int xyz_first,
    float *__restrict__ out, int *__restrict__ err) {
    constexpr unsigned C = 3u + (unsigned)D;
    constexpr unsigned IMG = 64u * C;                    // floats per image
    constexpr unsigned UNITS = IMG / 4u;                 // 16-B units per image
    constexpr unsigned NST = (UNITS + 63u) / 64u;        // store instructions per image
    __shared__ __attribute__((aligned(16))) float img[4][2][IMG];
    const unsigned lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const unsigned xo = xyz_first ? 0u : (unsigned)D;
    const unsigned fo = xyz_first ? 3u : 0u;
    const unsigned x = blockIdx.x & 7u, nwx = (gridDim.x >> 3) * 4u;
    unsigned c = x * cpx + (blockIdx.x >> 3) * 4u + (unsigned)wv;
    unsigned c_end = (x + 1u) * cpx;
    if (c_end > chunks) c_end = chunks;
    if (c >= c_end) return;
    struct Chunk {
        unsigned v;
        float p[3], q[3], f[D > 0 ? D : 1];
    };
    bool bad = false;
    auto load_index = [&](unsigned cc) -> long long {
        const __amdgpu_buffer_rsrc_t rs_idx = make_rsrc_uniform(idx + (size_t)cc * 64u, 64u * (unsigned)sizeof(IdxT));
        if constexpr (sizeof(IdxT) == 8) {
            const unsigned lo = __builtin_amdgcn_raw_buffer_load_b32(rs_idx, lane * 8u, 0, 0);
            const unsigned hi = __builtin_amdgcn_raw_buffer_load_b32(rs_idx, lane * 8u + 4u, 0, 0);
            return (long long)(((unsigned long long)hi << 32) | lo);
        } else {
            return (long long)(int)__builtin_amdgcn_raw_buffer_load_b32(rs_idx, lane * 4u, 0, 0);
        }
    };
    auto gather = [&](unsigned cc, unsigned b, Chunk &k) {
        const __amdgpu_buffer_rsrc_t rs_xyz = make_rsrc_uniform(xyz + (size_t)b * N * 3, (unsigned)N * 12u);
        // (three dword loads: hipcc 7.2 narrows a raw_buffer_load_b96 whose lanes are used separately to ONE dword)
#pragma unroll
        for (unsigned a_ = 0; a_ < 3; ++a_)
            k.p[a_] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_xyz, k.v * 12u + 4u * a_, 0, 0));
        const unsigned q0 = (unsigned)(((unsigned long long)cc * 64u) >> log2K);   // first query of the chunk
        const unsigned nq = (64u >> log2K) ? (64u >> log2K) : 1u;
        const __amdgpu_buffer_rsrc_t rs_q = make_rsrc_uniform(new_xyz + (size_t)q0 * 3, nq * 12u);
        const unsigned ql = (lane >> log2K) * 12u;
#pragma unroll
        for (unsigned a_ = 0; a_ < 3; ++a_)
            k.q[a_] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_q, ql + 4u * a_, 0, 0));
        if constexpr (D > 0) {
            const __amdgpu_buffer_rsrc_t rs_pts = make_rsrc_uniform(points + (size_t)b * N * D, (unsigned)N * (unsigned)D * 4u);
            // (dword loads, merged by the compiler where it is safe: hipcc 7.2 narrows raw_buffer_load_b64 / _b96 whose
            //  lanes are used separately to ONE dword -- seen in the ISA)
#pragma unroll
            for (unsigned i = 0; i < (unsigned)D; ++i)
                k.f[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_pts, k.v * (unsigned)(D * 4) + 4u * i, 0, 0));
        }
    };
    auto emit = [&](unsigned cc, const Chunk &k, float *im, bool on) {
        float *row = im + lane * C;
#pragma unroll
        for (unsigned a_ = 0; a_ < 3; ++a_) row[xo + a_] = k.p[a_] - k.q[a_];
#pragma unroll
        for (unsigned i = 0; i < (unsigned)D; ++i) row[fo + i] = k.f[i];
        const __amdgpu_buffer_rsrc_t rs_out = make_rsrc_uniform(out + (size_t)cc * IMG, on ? IMG * 4u : 0u);   // off: every store out of range
#pragma unroll
        for (unsigned t = 0; t < NST; ++t) {
            const unsigned u = t * 64u + lane;
            if (UNITS % 64u == 0u || u < UNITS) {
                const f32x4 w = *(const f32x4 *)&im[u * 4u];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, w), rs_out,
                                                       u * 16u, 0, POLICY);
            }
        }
    };
    // scan of a chunk, tracked incrementally in SGPRs (b = c / cps, r = c % cps): no division in the loop
    unsigned b1 = (unsigned)__builtin_amdgcn_readfirstlane((int)(c / cps));
    unsigned r1 = (unsigned)__builtin_amdgcn_readfirstlane((int)(c - b1 * cps));
    // The indices of a trip are fetched one trip ahead, behind the gathers of the trip before: a lone wave per SIMD (the
    // kernel's place beside the sampling) pays every dependent round trip in full, and index -> gather -> store was two.
    // (not with 64-bit indices and 9-float rows: the four extra registers would push the kernel past 48 VGPRs)
    constexpr bool AHEAD = !(sizeof(IdxT) == 8 && D == 6);
    auto second = [&](unsigned cc) { return cc + nwx < c_end ? cc + nwx : cc; };   // no second chunk: the first again
    long long raw1 = 0, raw2 = 0;
    if constexpr (AHEAD) {
        raw1 = load_index(c);
        raw2 = load_index(second(c));
    }
#pragma unroll 1
    for (; c < c_end; c += 2u * nwx) {
        const bool has2 = c + nwx < c_end;                       // wave-uniform
        const unsigned c2 = has2 ? c + nwx : c;                   // (its stores disabled when it is the first again)
        unsigned b2 = b1, r2 = r1;
        if (has2) {
            r2 += nwx;
            while (r2 >= cps) {
                r2 -= cps;
                ++b2;
            }
        }
        Chunk k1, k2;
        if constexpr (!AHEAD) {
            raw1 = load_index(c);
            raw2 = load_index(c2);
        }
        k1.v = checked_index(raw1, N, bad);
        k2.v = checked_index(raw2, N, bad);
        gather(c, b1, k1);
        gather(c2, b2, k2);
        if constexpr (AHEAD) {
            const unsigned cn = c + 2u * nwx < c_end ? c + 2u * nwx : c;   // next trip (or this one again: in range, unused)
            raw1 = load_index(cn);
            raw2 = load_index(second(cn));
        }
        emit(c, k1, img[wv][0], true);
        emit(c2, k2, img[wv][1], has2);
        b1 = b2;
        r1 = r2 + nwx;
        while (r1 >= cps) {
            r1 -= cps;
            ++b1;
        }
    }
    if (err && __any(bad) && lane == 0) atomicOr(err, 1);
}


}  // namespace tgn

using namespace tgn;

// impl: 0 = choose, 1 = the per-element kernel (any shape), 2 = 16-B stores through an LDS staging line (24 VGPRs: two waves
// per SIMD beside a register-hungry kernel), 7 = row pieces into an LDS image (rows of >= 64 floats, one wave per workgroup;
// max_blocks counts 4 of its waves as one block), 10 = pairs kernel (rows of 3 / 6 / 9 floats).  A kernel that does not take
// the shape falls back to the next one down.  store_policy: cache-policy bits of the 16-B output stores (0 plain, 2 nt, 16 sc1,
// 17 sc0|sc1, 18 sc1|nt), -1 = per-kernel default.  max_blocks: upper bound on the grid (0 = none): a caller that overlaps the
// grouping with a register-hungry kernel keeps it to what is resident beside that kernel.
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
    if (impl != 0 && impl != 1 && impl != 2 && impl != 7 && impl != 10) {
        set_error("tgn_group_points_ex: impl %d (0 choose, 1 per element, 2 staged 16-B stores, 7 row pieces, 10 pairs)", impl);
        return TGN_ERR_INVALID_ARGUMENT;
    }
    if (max_blocks < 0) max_blocks = 0;
    int *err = index_error_word((hipStream_t)stream);
    const int C = 3 + D;
    const unsigned magicC = (unsigned)((0x100000000ULL + C - 1) / C);
    const float *pts = points ? points : xyz;
    hipStream_t st = (hipStream_t)stream;
    const bool v2_ok = ((long long)K * C) % 4 == 0 && ((uintptr_t)out & 15) == 0 && (long long)N * (D > 0 ? D : 1) < (1LL << 30);
    const bool rows_ok = v2_ok && C >= 64 && K <= 64 && (long long)N * D * 4 >= 256 && K % 4 == 0 && queries < (1LL << 31);
    // narrow rows (level 1): one lane per (query, neighbour) pair, rows assembled in an LDS image
    {
        const long long pairs_per_scan = (long long)S * K;
        const bool pairs_ok = (D == 0 || D == 3 || D == 6) && (K & (K - 1)) == 0 && pairs_per_scan % 64 == 0 &&
                              ((uintptr_t)out & 15) == 0 && (long long)B * pairs_per_scan / 64 < (1LL << 31) &&
                              (long long)N * (D > 3 ? D : 3) * 4 < (1LL << 31);
        if (impl == 10 && !pairs_ok) impl = 0;
        if (impl == 0 && pairs_ok && C < 64) impl = 10;
        if (impl == 10) {
            // default policy of the image kernels (pairs, row pieces): nt.  Measured at every grid bound and beside the FPS
            // workgroups (profiles/r02_run8_extra_bench.txt): pairs 0.65 vs 0.67-0.86 ms with sc1, rows 1.29 / 1.00 vs 1.30-1.35 /
            // 1.06-1.09 ms; their gather source stays in L2 either way (1.0x traffic, r02_pmc_traffic.json)
            if (store_policy < 0) store_policy = 2;
            const unsigned cps = (unsigned)(pairs_per_scan / 64), chunks = (unsigned)B * cps;
            const unsigned cpx = B % 8 == 0 ? (unsigned)(B / 8) * cps : (chunks + 7u) / 8u;
            int log2K = 0;
            while ((1 << log2K) < K) ++log2K;
            long long blocks = ((long long)cpx + 7) / 8;          // a wave walks at least two chunks
            if (blocks > 256) blocks = 256;                        // per XCD: 32 CUs x 8 blocks
            if (max_blocks > 0 && max_blocks / 8 < blocks) blocks = max_blocks / 8 > 0 ? max_blocks / 8 : 1;
            const dim3 grid((unsigned)blocks * 8u);
#define TGN_GROUP_PAIRS(IT, DD)                                                                                        \
    do {                                                                                                               \
        if (store_policy == 0)                                                                                         \
            hipLaunchKernelGGL((group_points_pairs_kernel<IT, DD, 0>), grid, dim3(256), 0, st, chunks, cpx, cps, N, log2K, xyz, \
                               new_xyz, pts, (const IT *)idx, xyz_first, out, err);                                    \
        else if (store_policy == 2)                                                                                    \
            hipLaunchKernelGGL((group_points_pairs_kernel<IT, DD, 2>), grid, dim3(256), 0, st, chunks, cpx, cps, N, log2K, xyz, \
                               new_xyz, pts, (const IT *)idx, xyz_first, out, err);                                    \
        else                                                                                                           \
            hipLaunchKernelGGL((group_points_pairs_kernel<IT, DD, 16>), grid, dim3(256), 0, st, chunks, cpx, cps, N, log2K, xyz, \
                               new_xyz, pts, (const IT *)idx, xyz_first, out, err);                                    \
    } while (0)
            if (idx_is_int64) {
                if (D == 0) TGN_GROUP_PAIRS(long long, 0); else if (D == 3) TGN_GROUP_PAIRS(long long, 3); else TGN_GROUP_PAIRS(long long, 6);
            } else {
                if (D == 0) TGN_GROUP_PAIRS(int, 0); else if (D == 3) TGN_GROUP_PAIRS(int, 3); else TGN_GROUP_PAIRS(int, 6);
            }
#undef TGN_GROUP_PAIRS
            return check_launch("group_points_pairs_kernel");
        }
    }
    // default: the row-piece kernel for wide rows (a bounded grid means "runs beside something that owns most of every
    // CU" -- the FPS level-1 workgroups, which leave one <= 48-VGPR wave per SIMD and ~95 KiB of LDS: 4 of its waves fit)
    if (impl == 0) impl = rows_ok ? 7 : v2_ok ? 2 : 1;
    if (impl == 7 && !rows_ok) impl = 2;
    if (impl == 2 && !v2_ok) impl = 1;
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
    if (impl == 7) {
        if (store_policy < 0) store_policy = 2;   // nt, see the pairs kernel above
        // one wave per workgroup; R rows per LDS image (multiple of 4, about 8 KiB, at most 20 rows)
        int R = (int)(2176 / C) / 4 * 4;
        if (R < 4) R = 4;
        if (R > 20) R = 20;
        if (R > K) R = K;
        const size_t lds = (size_t)(2 * (32 + R * C) + 64 * 3 + 64 * 5) * sizeof(float);
        const bool wide_dma = D % 4 == 0;   // 16-byte LDS-DMA pieces
        long long qx = B >= 8 ? (long long)((B + 7) / 8) * S : (queries + 7) / 8;
        long long nb = qx;                       // workgroups (= waves) per XCD
        long long per_cu = (long long)(160 * 1024) / (long long)lds;   // resident workgroups per CU (LDS-bound)
        if (per_cu > 16) per_cu = 16;
        if (per_cu < 1) per_cu = 1;
        long long capw = 32 * per_cu;            // ... per XCD: the grid is exactly what is resident
        if (max_blocks > 0 && (long long)max_blocks * 4 / 8 < capw) capw = (long long)max_blocks * 4 / 8 > 0 ? (long long)max_blocks * 4 / 8 : 1;
        if (nb > capw) nb = capw;
#define TGN_GROUP_ROWS(IT, POL)                                                                                       \
    do {                                                                                                              \
        auto kfn = wide_dma ? group_points_rows_kernel<IT, POL, 4> : group_points_rows_kernel<IT, POL, 1>;            \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kfn, dim3((unsigned)(nb * 8)), dim3(64), lds, st, queries, qx, N, S, K, D, R, xyz, new_xyz, pts, \
                           (const IT *)idx, xyz_first, out, err);                                                     \
    } while (0)
        if (idx_is_int64) {
            if (store_policy == 0) TGN_GROUP_ROWS(long long, 0); else if (store_policy == 2) TGN_GROUP_ROWS(long long, 2); else TGN_GROUP_ROWS(long long, 16);
        } else {
            if (store_policy == 0) TGN_GROUP_ROWS(int, 0); else if (store_policy == 2) TGN_GROUP_ROWS(int, 2); else TGN_GROUP_ROWS(int, 16);
        }
#undef TGN_GROUP_ROWS
        return check_launch("group_points_rows_kernel");
    }
    // staged 16-B stores: per-XCD query ranges (whole scans when B >= 8), at most 8 blocks per CU resident
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
    if (store_policy < 0) store_policy = 16;   // write-through (profiles/r01_store_bench.txt)
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

// ball_query.hip -- pointnet2_utils.query_ball_point (pointnet2_utils.py:120-144) for gfx950.
//
// Semantics (bit-exact with the reference on the same scans): for every query, the first `nsample`
// point indices IN ASCENDING INDEX ORDER whose expanded-form squared distance
//     d = ((-2*fma(z1,z2,fma(y1,y2,x1*x2))) + |q|^2) + |p|^2          (pointnet2_utils.py:38-41)
// is not greater than fp32(radius^2); short rows are padded with the first hit; a row without a
// hit is filled with N.  The reference gets there with a (B,S,N) int64 matrix, a mask and a SORT
// over N (786 MB of temporaries per 24k scan at S=4096); nothing of that is materialised here.
//
// Two kernels:
//  * scan : one wave per query walks the cloud in index order, 64 candidates per step, compacts
//           hits with ballot + mbcnt and stops as soon as nsample hits are found.  O(S*N), exact by
//           construction; used for small clouds, huge radii and non-finite input.
//  * grid : a per-cloud uniform grid (cell >= radius) is built in LDS by one workgroup per cloud
//           (bbox reduce -> cell histogram with LDS atomics -> scan -> scatter of (x,y,z,index)
//           records, so a cell's points are one contiguous, coalesced 16-B stream).  A query wave then
//           visits the <= 9 contiguous runs of its 3x3x3 neighbourhood, applies the SAME expanded-form
//           test to each candidate (the grid only prunes, it never decides), collects hits in LDS and
//           rank-selects the nsample smallest indices.  ~100-200 candidates per query instead of N.
//
// Why the grid cannot change a result: a point passes the reference test only if d_fp <= r2 where
// |d_fp - d_true| <= 1e-5 * M^2 (M = largest |coordinate|; the bound is ~2.5x the worst-case fp32
// rounding of the expanded form), so its true distance is <= sqrt(r2 + 1e-5*M^2) =: r_eff.  Cells are
// r_eff * 1.001 wide or wider and cell coordinates of points and queries come from the same fp32
// expression, so such a point is at most one cell away on every axis.  Clouds containing non-finite
// coordinates (NaN compares as "inside" in the reference) take the scan path.
#include "tgn_common.h"

#include <stdlib.h>

namespace tgn {

// ------------------------------------------------------------------------------------------------
// scan path
// ------------------------------------------------------------------------------------------------
template <typename IdxT>
__device__ __forceinline__ void ball_scan_row(int N, int K, float r2, const float *__restrict__ pts, float cx,
                                              float cy, float cz, IdxT *__restrict__ row, int lane) {
    const float s1 = sumsq3(cx, cy, cz);
    int cnt = 0;
    int first = N;
    for (int basek = 0; basek < N && cnt < K; basek += kWave) {
        const int k = basek + lane;
        bool hit = false;
        if (k < N) {
            const float px = pts[(size_t)k * 3 + 0], py = pts[(size_t)k * 3 + 1], pz = pts[(size_t)k * 3 + 2];
            const float d = sqdist_expanded(cx, cy, cz, s1, px, py, pz, sumsq3(px, py, pz));
            hit = !(d > r2);  // the reference masks `sqrdists > radius**2` (pointnet2_utils.py:135)
        }
        const unsigned long long mask = __ballot(hit);
        if (mask) {
            const int pos = cnt + mbcnt(mask);
            if (hit && pos < K) row[pos] = (IdxT)k;
            if (cnt == 0) first = basek + __builtin_ctzll(mask);
            cnt += __popcll(mask);
        }
    }
    if (cnt > K) cnt = K;
    for (int j = cnt + lane; j < K; j += kWave) row[j] = (IdxT)first;  // pad with the first hit (:138-141)
}

template <typename IdxT>
__global__ __launch_bounds__(256) void ball_query_scan_kernel(int B, int N, int S, int K, float r2,
                                                               const float *__restrict__ xyz,
                                                               const float *__restrict__ new_xyz,
                                                               IdxT *__restrict__ out) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wpb = blockDim.x / kWave;
    const long long total = (long long)B * S;
    const unsigned nb = gridDim.x;  // a multiple of 8; XCD-aware order, see ball_grid_query_kernel
    const unsigned lb = (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3);
    for (long long q = (long long)lb * wpb + __builtin_amdgcn_readfirstlane(threadIdx.x / kWave); q < total; q += (long long)nb * wpb) {
        const int b = (int)(q / S);
        ball_scan_row<IdxT>(N, K, r2, xyz + (size_t)b * N * 3, new_xyz[q * 3 + 0], new_xyz[q * 3 + 1],
                            new_xyz[q * 3 + 2], out + q * K, lane);
    }
}

// ------------------------------------------------------------------------------------------------
// grid path
// ------------------------------------------------------------------------------------------------
constexpr int kGridCells = 16384;   // cells per cloud (LDS histogram: 64 KiB)
constexpr int kGridThreads = 1024;
constexpr int kHitCap = 512;        // per-wave hit buffer (indices)
constexpr int kGridMaxK = 256;
constexpr int kGridPermCap = 32768;   // clouds up to here sort through an LDS permutation of 16-bit point numbers

struct GridHeader {   // one per cloud, 64 bytes
    float lo[3];
    float inv_h;
    int g[3];
    int use_scan;     // 1: this cloud must take the scan path (non-finite data, degenerate grid)
    int pad[8];
};

// Per-cloud workspace: header | cell_start[kGridCells+1] (padded to 16 B) | N records float4(x, y, z, |p|^2) sorted by cell |
// N int32 point indices in the same order.  The squared norm is the `sumsq3` the distance test needs (same expression, computed
// once per point here instead of once per candidate and query); the index moved out of the record to make room for it.
constexpr size_t kGridRecOff = sizeof(GridHeader) + (size_t)((kGridCells + 1 + 3) / 4 * 4) * sizeof(int);
__host__ __device__ inline size_t grid_idx_off(int N) { return kGridRecOff + (size_t)N * sizeof(float4); }
__host__ __device__ inline size_t grid_cloud_bytes(int N) { return grid_idx_off(N) + (size_t)((N + 3) / 4 * 4) * sizeof(int); }

__device__ __forceinline__ int cell_coord(float p, float lo, float inv_h, int g) {
    // identical expression for points and queries; clamped to [-1, g] so far-away queries stay comparable
    float t = (p - lo) * inv_h;
    t = fminf(fmaxf(t, -1.0f), (float)g);
    return (int)floorf(t);
}

__device__ __forceinline__ float wave_min_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__global__ __launch_bounds__(kGridThreads) void ball_grid_build_kernel(int N, float r2, const float *__restrict__ xyz,
                                                                        unsigned char *__restrict__ ws) {
    __shared__ int cnt[kGridCells];
    extern __shared__ unsigned short perm[];        // sorted position -> point: 2 N bytes of dynamic LDS for clouds of <= kGridPermCap points (step 4)
    __shared__ float red[7][kGridThreads / kWave];
    __shared__ int wave_tot[kGridThreads / kWave];
    __shared__ GridHeader hdr_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *__restrict__ pts = xyz + (size_t)b * N * 3;
    unsigned char *base = ws + (size_t)b * grid_cloud_bytes(N);
    GridHeader *hdr = (GridHeader *)base;
    int *cell_start = (int *)(base + sizeof(GridHeader));
    float4 *rec = (float4 *)(base + kGridRecOff);
    int *ridx = (int *)(base + grid_idx_off(N));

    // 1. bounding box, largest |coordinate|, non-finite census
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float bad = 0.0f;
    for (int k = tid; k < N; k += kGridThreads) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pts[(size_t)k * 3 + a];
            if (!(fabsf(v) <= 3.0e38f)) bad = 1.0f;  // NaN or inf
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float l = wave_min_f32(lo[a]), h = wave_max_f32(hi[a]);
        if (lane == 0) {
            red[a][wave] = l;
            red[3 + a][wave] = h;
        }
    }
    {
        const float bb = wave_max_f32(bad);
        if (lane == 0) red[6][wave] = bb;
    }
    for (int i = tid; i < kGridCells; i += kGridThreads) cnt[i] = 0;
    __syncthreads();
    if (tid == 0) {
        GridHeader h;
        float ext[3], m = 0.0f, any_bad = 0.0f;
        for (int a = 0; a < 3; ++a) {
            float l = INFINITY, u = -INFINITY;
            for (int w = 0; w < kGridThreads / kWave; ++w) {
                l = fminf(l, red[a][w]);
                u = fmaxf(u, red[3 + a][w]);
            }
            h.lo[a] = l;
            ext[a] = u - l;
            m = fmaxf(m, fmaxf(fabsf(l), fabsf(u)));
        }
        for (int w = 0; w < kGridThreads / kWave; ++w) any_bad = fmaxf(any_bad, red[6][w]);
        // queries may lie outside the points' box; their coordinates are bounded by the caller's data too,
        // but to stay safe the margin uses 4*M^2 (|q| up to 2M from the box still covered)
        const float r_eff = sqrtf(fmaxf(r2, 0.0f) + 4.0e-5f * m * m) * 1.001f + 1e-30f;
        float hcell = r_eff;
        int g[3];
        for (int it = 0; it < 64; ++it) {
            const float inv = 1.0f / hcell;
            long long cells = 1;
            for (int a = 0; a < 3; ++a) {
                const float t = ext[a] * inv;   // same expression as cell_coord(hi) -> floor(t) = g-1
                g[a] = (t < 1.0e6f) ? (int)floorf(t) + 1 : 1000001;
                cells *= g[a];
            }
            if (cells <= kGridCells) break;
            hcell *= 1.1f;
        }
        h.inv_h = 1.0f / hcell;
        long long cells = 1;
        for (int a = 0; a < 3; ++a) {
            h.g[a] = g[a];
            cells *= g[a];
        }
        // degenerate grids (everything in a handful of cells) cannot prune: the early-exit scan is better
        h.use_scan = (any_bad > 0.0f || !(r2 >= 0.0f) || cells > kGridCells || cells < 8 || N == 0) ? 1 : 0;
        for (int i = 0; i < 8; ++i) h.pad[i] = 0;
        hdr_s = h;
        *hdr = h;
    }
    __syncthreads();
    const GridHeader h = hdr_s;
    if (h.use_scan) return;

    // 2. histogram
    for (int k = tid; k < N; k += kGridThreads) {
        const int cx = cell_coord(pts[(size_t)k * 3 + 0], h.lo[0], h.inv_h, h.g[0]);
        const int cy = cell_coord(pts[(size_t)k * 3 + 1], h.lo[1], h.inv_h, h.g[1]);
        const int cz = cell_coord(pts[(size_t)k * 3 + 2], h.lo[2], h.inv_h, h.g[2]);
        atomicAdd(&cnt[(cz * h.g[1] + cy) * h.g[0] + cx], 1);
    }
    __syncthreads();
    // 3. exclusive scan of the histogram: 16 cells per thread, wave scan, block scan
    constexpr int PER = kGridCells / kGridThreads;
    int local[PER];
    int sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        local[i] = sum;
        sum += cnt[tid * PER + i];
    }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == kWave - 1) wave_tot[wave] = incl;
    __syncthreads();
    int wave_base = 0;
    for (int w = 0; w < wave; ++w) wave_base += wave_tot[w];
    const int thread_base = wave_base + incl - sum;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int v = thread_base + local[i];
        cell_start[tid * PER + i] = v;
        cnt[tid * PER + i] = v;  // running insert position
    }
    if (tid == kGridThreads - 1) cell_start[kGridCells] = thread_base + sum;
    __syncthreads();
    // 4. scatter (order inside a cell is arbitrary: the query kernel rank-selects by index).
    // Clouds of <= 32 768 points: the scatter goes into an LDS permutation (2 bytes per point) and the records are then written
    // in sorted order -- whole lines, 16 B per lane -- with the coordinates gathered from the cloud (288 KB at 24 000 points: L2 /
    // L1 hits).  Scattering the 16-byte records themselves wrote 225 MB per 256 scans for 98 MB of records: partially filled
    // lines evicted and fetched again (profiles/r05_pmc_traffic.json).
    if (N <= kGridPermCap) {
        for (int k = tid; k < N; k += kGridThreads) {
            const int cx = cell_coord(pts[(size_t)k * 3 + 0], h.lo[0], h.inv_h, h.g[0]);
            const int cy = cell_coord(pts[(size_t)k * 3 + 1], h.lo[1], h.inv_h, h.g[1]);
            const int cz = cell_coord(pts[(size_t)k * 3 + 2], h.lo[2], h.inv_h, h.g[2]);
            perm[atomicAdd(&cnt[(cz * h.g[1] + cy) * h.g[0] + cx], 1)] = (unsigned short)k;
        }
        __syncthreads();
#pragma unroll 4
        for (int pos = tid; pos < N; pos += kGridThreads) {
            const int k = perm[pos];
            const float px = pts[(size_t)k * 3 + 0], py = pts[(size_t)k * 3 + 1], pz = pts[(size_t)k * 3 + 2];
            rec[pos] = make_float4(px, py, pz, sumsq3(px, py, pz));
            ridx[pos] = k;
        }
        return;
    }
    for (int k = tid; k < N; k += kGridThreads) {
        const float px = pts[(size_t)k * 3 + 0], py = pts[(size_t)k * 3 + 1], pz = pts[(size_t)k * 3 + 2];
        const int cx = cell_coord(px, h.lo[0], h.inv_h, h.g[0]);
        const int cy = cell_coord(py, h.lo[1], h.inv_h, h.g[1]);
        const int cz = cell_coord(pz, h.lo[2], h.inv_h, h.g[2]);
        const int pos = atomicAdd(&cnt[(cz * h.g[1] + cy) * h.g[0] + cx], 1);
        rec[pos] = make_float4(px, py, pz, sumsq3(px, py, pz));
        ridx[pos] = k;
    }
}

// Rank-select: entry e of buf[0..H) gets rank = #entries smaller than it (indices are distinct).
// Entries with rank < K are written to dst[rank].  Each lane owns entries lane, lane+64, ...
template <int OWN, typename DstT>
__device__ __forceinline__ void rank_select_n(const int *buf, int H, int K, DstT *dst, int lane) {
    int mine[OWN], rank[OWN];
#pragma unroll
    for (int o = 0; o < OWN; ++o) {
        const int e = lane + o * kWave;
        mine[o] = e < H ? buf[e] : 0x7FFFFFFF;
        rank[o] = 0;
    }
    for (int i = 0; i < H; ++i) {
        const int v = buf[i];  // LDS broadcast read
#pragma unroll
        for (int o = 0; o < OWN; ++o) rank[o] += (v < mine[o]) ? 1 : 0;
    }
#pragma unroll
    for (int o = 0; o < OWN; ++o) {
        const int e = lane + o * kWave;
        if (e < H && rank[o] < K) dst[rank[o]] = (DstT)mine[o];
    }
}

template <typename DstT>
__device__ __forceinline__ void rank_select(const int *buf, int H, int K, DstT *dst, int lane) {
    if (H <= kWave)  // wave-uniform
        rank_select_n<1, DstT>(buf, H, K, dst, lane);
    else if (H <= 2 * kWave)
        rank_select_n<2, DstT>(buf, H, K, dst, lane);
    else
        rank_select_n<kHitCap / kWave, DstT>(buf, H, K, dst, lane);
}

template <typename IdxT>
__global__ __launch_bounds__(256) void ball_grid_query_kernel(int B, int N, int S, int K, float r2,
                                                               const float *__restrict__ xyz,
                                                               const float *__restrict__ new_xyz,
                                                               const unsigned char *__restrict__ ws,
                                                               IdxT *__restrict__ out) {
    __shared__ int hits_s[4][kHitCap];
    __shared__ int keep_s[4][kGridMaxK];
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    int *hits = hits_s[wv];
    int *keep = keep_s[wv];
    const long long total = (long long)B * S;
    const size_t cloud_bytes = grid_cloud_bytes(N);
    // XCD-aware block order (hardware block i runs on XCD i % 8; gridDim.x is a multiple of 8): every XCD walks one
    // contiguous range of queries, so a cloud's grid (cell table + records, ~0.45 MB at N = 24000, re-read ~60x by
    // its S queries) is pulled into ONE XCD's L2 instead of all eight.  Speed only.
    const unsigned nb = gridDim.x;
    const unsigned lb = (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3);
    for (long long q = (long long)lb * 4 + wv; q < total; q += (long long)nb * 4) {
        const int b = (int)(q / S);
        const unsigned char *base = ws + (size_t)b * cloud_bytes;
        const GridHeader *hdr = (const GridHeader *)base;
        const float cx = new_xyz[q * 3 + 0], cy = new_xyz[q * 3 + 1], cz = new_xyz[q * 3 + 2];
        IdxT *__restrict__ row = out + q * K;
        const bool q_finite = fabsf(cx) <= 3.0e38f && fabsf(cy) <= 3.0e38f && fabsf(cz) <= 3.0e38f;
        if (hdr->use_scan || !q_finite) {  // wave-uniform
            ball_scan_row<IdxT>(N, K, r2, xyz + (size_t)b * N * 3, cx, cy, cz, row, lane);
            continue;
        }
        const int *__restrict__ cell_start = (const int *)(base + sizeof(GridHeader));
        const float4 *__restrict__ rec = (const float4 *)(base + kGridRecOff);
        const int *__restrict__ ridx = (const int *)(base + grid_idx_off(N));
        const int gx = hdr->g[0], gy = hdr->g[1], gz = hdr->g[2];
        const float inv_h = hdr->inv_h;
        const int qx = cell_coord(cx, hdr->lo[0], inv_h, gx);
        const int qy = cell_coord(cy, hdr->lo[1], inv_h, gy);
        const int qz = cell_coord(cz, hdr->lo[2], inv_h, gz);
        const int x0 = max(qx - 1, 0), x1 = min(qx + 1, gx - 1);
        // lanes 0..8: the (dy,dz) runs of up to three x-adjacent cells, contiguous in cell order
        int rs = 0, re = 0;
        if (lane < 9 && x0 <= x1) {
            const int yy = qy + (lane % 3) - 1, zz = qz + (lane / 3) - 1;
            if (yy >= 0 && yy < gy && zz >= 0 && zz < gz) {
                const int c0 = (zz * gy + yy) * gx;
                rs = cell_start[c0 + x0];
                re = cell_start[c0 + x1 + 1];
            }
        }
        const float s1 = sumsq3(cx, cy, cz);
        int H = 0;       // entries in the buffer
        // The <= 9 runs are walked as ONE flattened candidate list, 64 candidates per step (a run holds ~15
        // records on a scan surface: one step per run would leave three quarters of the lanes idle and pay nine
        // dependent-load latencies instead of ~three).  Lane r < 9 holds run r's [rs, re); pre[r] = records before it.
        const int len = re - rs;
        int pre = len;  // inclusive scan over lanes 0..15 (one DPP row), then made exclusive
        pre += (int)dpp_or_zero<0x111, 0xF>((unsigned)pre);
        pre += (int)dpp_or_zero<0x112, 0xF>((unsigned)pre);
        pre += (int)dpp_or_zero<0x114, 0xF>((unsigned)pre);
        pre += (int)dpp_or_zero<0x118, 0xF>((unsigned)pre);
        const int T = __builtin_amdgcn_readlane(pre, 8);  // lanes 9.. have len 0
        pre -= len;
        const int delta = rs - pre;  // record index = flattened index + delta of its run
        int ends[9];                 // wave-uniform run ends in the flattened list
#pragma unroll
        for (int r = 0; r < 9; ++r) ends[r] = __builtin_amdgcn_readlane(pre + len, r);
        for (int g0 = 0; g0 < T; g0 += kWave) {
            const int g = g0 + lane;
            int r = 0;
#pragma unroll
            for (int t = 0; t < 8; ++t) r += (g >= ends[t]) ? 1 : 0;
            const int j = g + __shfl(delta, r);
            bool hit = false;
            int pidx = 0;
            if (g < T) {
                const float4 p = rec[j];
                const float d = sqdist_expanded(cx, cy, cz, s1, p.x, p.y, p.z, p.w);
                hit = !(d > r2);
                pidx = ridx[j];
            }
            const unsigned long long mask = __ballot(hit);
            if (mask == 0) continue;
            const int nh = __popcll(mask);
            if (H + nh > kHitCap) {
                // compact: only the K smallest indices can matter
                rank_select<int>(hits, H, K, keep, lane);
                const int nk = H < K ? H : K;
                for (int i = lane; i < nk; i += kWave) hits[i] = keep[i];
                H = nk;
            }
            if (hit) hits[H + mbcnt(mask)] = pidx;
            H += nh;
        }
        if (H == 0) {
            for (int j = lane; j < K; j += kWave) row[j] = (IdxT)N;  // no hit at all -> N (:136-141)
            continue;
        }
        rank_select<IdxT>(hits, H, K, row, lane);
        if (H < K) {
            // first hit = smallest index in the buffer
            int mn = 0x7FFFFFFF;
            for (int i = lane; i < H; i += kWave) mn = min(mn, hits[i]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mn = min(mn, __shfl_xor(mn, o));
            for (int j = H + lane; j < K; j += kWave) row[j] = (IdxT)mn;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// grid path, bitmap selection (clouds of up to 32 768 points: every Shape-A / Shape-B level)
// ------------------------------------------------------------------------------------------------
// The answer of a query is "the nsample smallest indices among its hits, ascending".  ball_grid_query_kernel gets there by
// rank-selecting the hit list (every lane compares its hit with every other one: ~3 instructions per hit and lane-slot,
// a third of the kernel).  Here the hits are ALSO dropped into a bitmap over the cloud's index range (one LDS `or` per
// candidate step); the rank of a hit is then the number of set bits below it:
//   * lane l owns the words of indices [l*W*32, (l+1)*W*32), W = 4*nquad: it counts its bits (one v_bcnt per word), a wave
//     scan over the lane totals gives every 128-index group its base rank (gbase);
//   * the lane that holds hit v reads gbase[v >> 7] and the group's four words and counts the bits below v: ~20
//     instructions whatever the number of hits; ranks < nsample are stored, rank 0 is the pad value (pointnet2_utils.py:138-141);
//   * the owners wipe their words (three 16-byte LDS stores per lane at N = 24 000).
// Candidates are walked one x-run per 16-lane row (a run holds ~15 records on a scan surface) instead of as one flattened
// list: the run a lane serves is a constant of the step, so there is no per-lane search for "which run does candidate g
// belong to" (16 of the ~60 instructions of a step before).  More hits than the list holds (kBmHitCap) are handled by
// walking the candidates a second time and ranking each hit straight from the bitmap.
// Same cells, same test, same arithmetic as above: the results are identical.
constexpr int kBmMaxN = 32768;
constexpr int kBmHitCap = 256;

// Cost: the kernel is bound by vector-ALU issue (a wave64 instruction holds its SIMD for 4 cycles; the rank-select kernel
// spends ~430 of them per query), so this one is written for few instructions on the common path:
//   * the candidates are one flattened list over the <= 9 runs; the first 192 of them (three per lane) are requested at
//     once, so the common query never loops and pays one record-load latency;
//   * records, cell table and the output row go through buffer descriptors (32-bit offsets, out-of-range lanes read 0);
//   * the cell-table look-ups of the wave's NEXT query are issued before the current one is processed.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ball_rsrc(const void *base, unsigned bytes) {   // wave-uniform base
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

struct BallRuns {      // what a query needs to start: lanes 0..8 hold the record range of one x-run each
    float cx, cy, cz;  // wave-uniform
    int rs, re;
    int b;             // scan (cloud) of the query
    bool scan;         // wave-uniform: the cloud has no grid / the query is not finite -> index-order scan
};

template <typename IdxT>
__global__ __launch_bounds__(256) void ball_grid_query_bitmap_kernel(int B, int N, int S, int K, float r2,
                                                                      const float *__restrict__ xyz,
                                                                      const float *__restrict__ new_xyz,
                                                                      const unsigned char *__restrict__ ws,
                                                                      IdxT *__restrict__ out, long long q_per_xcd) {
    extern __shared__ __attribute__((aligned(16))) unsigned bm_dyn[];   // per wave: bitmap [nquad*256], gbase [nquad*64], hits [kBmHitCap]
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int nquad = (N + 8191) >> 13;   // 16-byte quads (128 indices) per lane: lane l owns quads [l*nquad, (l+1)*nquad)
    const int per_wave = nquad * 256 + nquad * 64 + kBmHitCap;
    unsigned *bm = bm_dyn + wv * per_wave;
    unsigned *gbase = bm + nquad * 256;
    int *hits = (int *)(gbase + nquad * 64);
    for (int t = 0; t < nquad; ++t) *(u32x4 *)&bm[(lane * nquad + t) * 4] = u32x4{0u, 0u, 0u, 0u};
    const long long total = (long long)B * S;
    const size_t cloud_bytes = grid_cloud_bytes(N);
    const size_t rec_off = kGridRecOff;
    const int dy = lane % 3 - 1, dz = lane / 3 - 1;   // lanes 0..8: the (dy, dz) x-run this lane looks up
    // XCD x (hardware block i runs on XCD i % 8) owns the contiguous query range [x*q_per_xcd, (x+1)*q_per_xcd) -- whole
    // clouds when there are at least 8 -- and its blocks, no more than are resident at a time, walk it together: a cloud's
    // grid (cell table + records, 0.45 MB at N = 24 000, read ~20 times over by its queries) is pulled into ONE L2 once
    // (PMC: 963 MB of HBM reads per level-1 launch when every wave strode over 16 clouds, 4.4x the algorithmic bytes).
    const unsigned xcd = blockIdx.x & 7u;
    const long long stride = (long long)(gridDim.x >> 3) * 4;
    long long q_end = (long long)(xcd + 1) * q_per_xcd;
    if (q_end > total) q_end = total;

    // (the scan a query belongs to is tracked incrementally: a 64-bit division per query costs ~100 instructions)
    auto lookup = [&](long long q, int bq) -> BallRuns {   // issues the two cell-table loads of query q (lanes 0..8) of scan bq
        BallRuns r;
        r.b = bq;
        const unsigned char *base = ws + (size_t)r.b * cloud_bytes;
        const GridHeader *hdr = (const GridHeader *)base;
        r.cx = new_xyz[q * 3 + 0];
        r.cy = new_xyz[q * 3 + 1];
        r.cz = new_xyz[q * 3 + 2];
        const bool q_finite = fabsf(r.cx) <= 3.0e38f && fabsf(r.cy) <= 3.0e38f && fabsf(r.cz) <= 3.0e38f;
        r.scan = hdr->use_scan || !q_finite;
        r.rs = r.re = 0;
        if (r.scan) return r;
        const __amdgpu_buffer_rsrc_t rs_cells = ball_rsrc(base + sizeof(GridHeader), (kGridCells + 1) * 4u);
        const int gx = hdr->g[0], gy = hdr->g[1], gz = hdr->g[2];
        const float inv_h = hdr->inv_h;
        const int qx = cell_coord(r.cx, hdr->lo[0], inv_h, gx);
        const int qy = cell_coord(r.cy, hdr->lo[1], inv_h, gy);
        const int qz = cell_coord(r.cz, hdr->lo[2], inv_h, gz);
        const int x0 = max(qx - 1, 0), x1 = min(qx + 1, gx - 1);
        const int yy = qy + dy, zz = qz + dz;
        const bool ok = lane < 9 && x0 <= x1 && yy >= 0 && yy < gy && zz >= 0 && zz < gz;
        const unsigned c0 = (unsigned)((zz * gy + yy) * gx);
        // lanes without a run read out of range: the hardware returns 0 for both ends -> an empty run
        r.rs = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_cells, ok ? (c0 + (unsigned)x0) * 4u : 0xFFFFFFF0u, 0, 0);
        r.re = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_cells, ok ? (c0 + (unsigned)x1 + 1u) * 4u : 0xFFFFFFF0u, 0, 0);
        return r;
    };

    long long q = (long long)xcd * q_per_xcd + (long long)(blockIdx.x >> 3) * 4 + wv;
    if (q >= q_end) return;
    int bcur = __builtin_amdgcn_readfirstlane((int)(q / S));
    int rcur = __builtin_amdgcn_readfirstlane((int)(q - (long long)bcur * S));     // q = bcur*S + rcur
    const int db = __builtin_amdgcn_readfirstlane((int)(stride / S));
    const int dr = __builtin_amdgcn_readfirstlane((int)(stride - (long long)db * S));
    BallRuns cur = lookup(q, bcur);
    for (; q < q_end; q += stride) {
        const long long qn = q + stride;
        bcur += db;
        rcur += dr;
        if (rcur >= S) {
            rcur -= S;
            ++bcur;
        }   // (bcur, rcur) now describe qn
        if (cur.scan) {  // wave-uniform
            ball_scan_row<IdxT>(N, K, r2, xyz + (size_t)cur.b * N * 3, cur.cx, cur.cy, cur.cz, out + q * K, lane);
            if (qn < q_end) cur = lookup(qn, bcur);
            continue;
        }
        const __amdgpu_buffer_rsrc_t rs_rec = ball_rsrc(ws + (size_t)cur.b * cloud_bytes + rec_off, (unsigned)N * 16u);
        const __amdgpu_buffer_rsrc_t rs_idx = ball_rsrc(ws + (size_t)cur.b * cloud_bytes + grid_idx_off(N), (unsigned)N * 4u);
        const __amdgpu_buffer_rsrc_t rs_out = ball_rsrc(out + q * K, (unsigned)K * (unsigned)sizeof(IdxT));
        auto put = [&](int r, int v) {   // orow[r] = v
            if constexpr (sizeof(IdxT) == 8) {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{(unsigned)v, 0u}, rs_out, (unsigned)r * 8u, 0, 0);   // indices are >= 0
            } else {
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)v, rs_out, (unsigned)r * 4u, 0, 0);
            }
        };
        const float cx = cur.cx, cy = cur.cy, cz = cur.cz;
        const float s1 = sumsq3(cx, cy, cz);
        // The <= 9 runs are walked as ONE flattened candidate list (on a scan surface the candidates sit in ~3 long runs,
        // not 9 short ones: fixed lane groups per run leave most lanes idle and the rest looping).  Lane r < 9 holds run
        // r's [rs, re); pre = records before it (inclusive scan over one DPP row); candidate g lives in run
        // #{t : ends[t] <= g} at record g + delta[run].  The first 192 candidates are requested at once.
        const int len = cur.re - cur.rs;
        int pre = len;
        pre += (int)dpp_or_zero<0x111, 0xF>((unsigned)pre);
        pre += (int)dpp_or_zero<0x112, 0xF>((unsigned)pre);
        pre += (int)dpp_or_zero<0x114, 0xF>((unsigned)pre);
        pre += (int)dpp_or_zero<0x118, 0xF>((unsigned)pre);
        const int T = __builtin_amdgcn_readlane(pre, 8);   // lanes 9.. have len 0
        const int delta = cur.rs - (pre - len);
        int ends[8];                                        // wave-uniform run ends in the flattened list
#pragma unroll
        for (int r = 0; r < 8; ++r) ends[r] = __builtin_amdgcn_readlane(pre, r);
        auto record_of = [&](int g) -> int {               // flattened candidate -> record index
            int r = 0;
#pragma unroll
            for (int t = 0; t < 8; ++t) r += (g >= ends[t]) ? 1 : 0;
            return g + __shfl(delta, r);
        };
        u32x4 p[3];
        int pi[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int g = 64 * i + lane;
            const int j = record_of(g);
            p[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_rec, g < T ? (unsigned)j * 16u : 0xFFFFFFF0u, 0, 0);
            pi[i] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_idx, g < T ? (unsigned)j * 4u : 0xFFFFFFF0u, 0, 0);
        }
        // the next query's cell-table look-ups go out behind them and land while this query is processed
        BallRuns nxt;
        nxt.scan = true;
        nxt.rs = nxt.re = nxt.b = 0;
        nxt.cx = nxt.cy = nxt.cz = 0.0f;
        if (qn < q_end) nxt = lookup(qn, bcur);

        int H = 0;
        auto test = [&](const u32x4 &pp) -> bool {
            const float px = __uint_as_float(pp[0]), py = __uint_as_float(pp[1]), pz = __uint_as_float(pp[2]);
            const float d = sqdist_expanded(cx, cy, cz, s1, px, py, pz, __uint_as_float(pp[3]));
            return !(d > r2);
        };
        auto record = [&](bool hit, int pidx) {   // pass 0: set the bit, list the hit
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
            if (hit) {
                atomicOr(&bm[(unsigned)pidx >> 5], 1u << (pidx & 31));
                const int pos = H + mbcnt(mask);
                if (pos < kBmHitCap) hits[pos] = pidx;
            }
            H += __popcll(mask);
        };
#pragma unroll
        for (int i = 0; i < 3; ++i) record(64 * i + lane < T && test(p[i]), pi[i]);
        // rank of index v = number of set bits below it (valid once gbase is up to date)
        auto rank_of = [&](int v) -> int {
            const unsigned g = (unsigned)v >> 7, wsel = ((unsigned)v >> 5) & 3u, below = (1u << (v & 31)) - 1u;
            const u32x4 w = *(const u32x4 *)&bm[g * 4u];
            int r = (int)gbase[g];
            r += __popc(w[0] & (wsel > 0u ? ~0u : below));
            r += __popc(w[1] & (wsel > 1u ? ~0u : wsel == 1u ? below : 0u));
            r += __popc(w[2] & (wsel > 2u ? ~0u : wsel == 2u ? below : 0u));
            r += __popc(w[3] & (wsel == 3u ? below : 0u));
            return r;
        };
        int first = 0x7FFFFFFF;   // the hit of rank 0, in the lane that holds it
        // the candidates beyond the first 192 (pass 0), or every candidate again (pass 1: the hit list overflowed and each
        // hit is ranked straight from the bitmap)
        auto walk = [&](const int pass) {
            for (int g0 = pass == 0 ? 192 : 0; g0 < T; g0 += kWave) {
                const int g = g0 + lane;
                const int j = record_of(g);
                const u32x4 pp = __builtin_amdgcn_raw_buffer_load_b128(rs_rec, g < T ? (unsigned)j * 16u : 0xFFFFFFF0u, 0, 0);
                const int pidx = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_idx, g < T ? (unsigned)j * 4u : 0xFFFFFFF0u, 0, 0);
                const bool hit = g < T && test(pp);
                if (pass == 0) {
                    record(hit, pidx);
                } else if (hit) {
                    const int r = rank_of(pidx);
                    if (r < K) put(r, pidx);
                    if (r == 0) first = pidx;
                }
            }
        };
        if (T > 192) walk(0);
        if (H == 0) {   // wave-uniform; the bitmap is still clean
            for (int j = lane; j < K; j += kWave) put(j, N);  // no hit at all -> N (:136-141)
            cur = nxt;
            continue;
        }
        wave_lds_fence();   // the bits were set by whichever lanes held the hits
        {   // base rank of every 128-index group
            unsigned c[4] = {0u, 0u, 0u, 0u};   // nquad <= 4
            unsigned tot = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < nquad) {
                    const u32x4 w = *(const u32x4 *)&bm[(lane * nquad + t) * 4];
                    c[t] = (unsigned)(__popc(w[0]) + __popc(w[1]) + __popc(w[2]) + __popc(w[3]));
                    tot += c[t];
                }
            }
            unsigned incl = tot;   // inclusive scan over the 64 lanes
            incl += dpp_or_zero<0x111, 0xF>(incl);
            incl += dpp_or_zero<0x112, 0xF>(incl);
            incl += dpp_or_zero<0x114, 0xF>(incl);
            incl += dpp_or_zero<0x118, 0xF>(incl);
            incl += dpp_or_zero<0x142, 0xA>(incl);   // row_bcast:15 into rows 1, 3
            incl += dpp_or_zero<0x143, 0xC>(incl);   // row_bcast:31 into rows 2, 3
            unsigned run_ = incl - tot;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < nquad) {
                    gbase[lane * nquad + t] = run_;
                    run_ += c[t];
                }
            }
        }
        wave_lds_fence();   // gbase and the hit list cross lanes
        if (H <= kBmHitCap) {
            for (int i = lane; i < H; i += kWave) {
                const int v = hits[i];
                const int r = rank_of(v);
                if (r < K) put(r, v);
                if (r == 0) first = v;
            }
        } else {
            walk(1);
        }
        if (H < K) {   // pad with the first hit = the smallest index (:138-141)
            const unsigned long long fm = __ballot(first != 0x7FFFFFFF);
            const int fv = __builtin_amdgcn_readlane(first, (int)__builtin_ctzll(fm));
            for (int j = H + lane; j < K; j += kWave) put(j, fv);
        }
        wave_lds_fence();   // every rank has been read: the bitmap can be wiped for the next query
        for (int t = 0; t < nquad; ++t) *(u32x4 *)&bm[(lane * nquad + t) * 4] = u32x4{0u, 0u, 0u, 0u};
        wave_lds_fence();
        cur = nxt;
    }
}


// ------------------------------------------------------------------------------------------------
// grid path, chunked bitmap kernel (round 6; clouds of up to 32 768 points -- the default)
// ------------------------------------------------------------------------------------------------
// ball_grid_query_bitmap_kernel issues ~370 vector instructions per query and runs at the VALU issue limit (one wave64 instruction
// per 4 cycles and SIMD): every instruction removed is time.  Same cells, same test, same arithmetic, same results; what changed
// is where the instructions go (profiles/r06_ball_sq_by_stage.txt):
//   * ONE wave per workgroup, its LDS at fixed offsets: every LDS address is a register plus an instruction offset;
//   * a wave takes CHUNKS of 16 consecutive queries of one cloud.  The per-query bookkeeping that was wave-uniform work on the
//     vector ALU -- three cell coordinates, nine cell-table look-ups, the prefix sum of the nine run lengths -- is done once per
//     chunk with a lane per (query, dy): 16 x 4 lanes, three passes (dz); ends and record offsets of the runs go to a 64-byte
//     block per query in LDS, next to the query's coordinates and squared norm;
//   * candidate g of the flattened list finds its run by a three-step binary search over that block's ends in LDS (three
//     dependent ds_read_u16 + 8 vector instructions) plus one compare for the last run, instead of eight compare/add pairs on
//     nine v_readlane'd ends per 64 candidates;
//   * the record carries |p|^2 (ball_grid_build_kernel), the test is 3 fma + 3 add + 1 compare;
//   * ranks come from a per-WORD exclusive prefix of the bitmap's popcounts (u16, built by the 64 word owners with chained
//     v_bcnt + one wave scan): rank(v) = base[v >> 5] + bcnt(word & below(v)) -- two LDS reads and ~8 instructions per 64 hits,
//     where the per-128-index bases needed four masked popcounts (~33).
constexpr int kChunkQ = 16;            // queries per chunk
constexpr int kQBlk = 64;              // bytes per query block: 9 run entries (end u16 | delta i16 << 16), pad, (cx, cy, cz, |c|^2) at +48

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int dpp_zero_i(int v) {   // lanes without a source / outside the masks receive 0
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, BANK_MASK, false);
}

template <typename IdxT, int NQ>
__global__ __launch_bounds__(64) void ball_grid_query_chunk_kernel(int B, int N, int S, int K, float r2,
                                                                    const float *__restrict__ xyz,
                                                                    const float *__restrict__ new_xyz,
                                                                    const unsigned char *__restrict__ ws,
                                                                    IdxT *__restrict__ out, int chunks_per_xcd) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    // LDS of the wave (bytes): bitmap | per-word base ranks (u16) | hit list (u16) | query blocks | pad value
    constexpr int kOffBase = NQ * 1024, kOffHits = kOffBase + NQ * 512, kOffQ = kOffHits + kBmHitCap * 2,
                  kOffFirst = kOffQ + kChunkQ * kQBlk, kLdsBytes = kOffFirst + 16;
    __shared__ __attribute__((aligned(16))) unsigned char lds[kLdsBytes];
    unsigned *bm = (unsigned *)lds;
    unsigned short *wbase = (unsigned short *)(lds + kOffBase);
    unsigned short *hits = (unsigned short *)(lds + kOffHits);
    const int lane = threadIdx.x;
    const int ql = lane >> 2, dyi = lane & 3;   // chunk stage: lane = (query of the chunk, dy + 1); dyi == 3 idles
#pragma unroll
    for (int t = 0; t < NQ; ++t) *(u32x4 *)&bm[(lane * NQ + t) * 4] = u32x4{0u, 0u, 0u, 0u};
    const size_t cloud_bytes = grid_cloud_bytes(N);
    const int cpc = (S + kChunkQ - 1) / kChunkQ;          // chunks per cloud (a chunk never straddles two clouds)
    const long long total_chunks = (long long)B * cpc;
    // XCD x (hardware block i runs on XCD i % 8) owns the contiguous chunk range [x * chunks_per_xcd, (x + 1) * chunks_per_xcd) --
    // whole clouds when there are at least 8 -- and its waves, no more than are resident at a time, walk it together: a cloud's
    // grid (cell table + records, 0.55 MB at N = 24 000, read ~20 times over by its queries) is pulled into ONE L2 once
    const unsigned xcd = blockIdx.x & 7u;
    const int wstride = (int)(gridDim.x >> 3);
    long long c_end = (long long)(xcd + 1) * chunks_per_xcd;
    if (c_end > total_chunks) c_end = total_chunks;
    long long c = (long long)xcd * chunks_per_xcd + (long long)(blockIdx.x >> 3);
    if (c >= c_end) return;
    int bcur = __builtin_amdgcn_readfirstlane((int)(c / cpc));
    int ccur = __builtin_amdgcn_readfirstlane((int)(c - (long long)bcur * cpc));   // c = bcur * cpc + ccur
    const int db = wstride / cpc, dc = wstride - db * cpc;

    for (; c < c_end; c += wstride) {
        const int b = bcur, q0 = ccur * kChunkQ;          // first query of the chunk within its cloud
        bcur += db;
        ccur += dc;
        if (ccur >= cpc) {
            ccur -= cpc;
            ++bcur;
        }
        const int nq = min(kChunkQ, S - q0);
        const long long qg0 = (long long)b * S + q0;      // global number of the chunk's first query
        const unsigned char *base = ws + (size_t)b * cloud_bytes;
        const GridHeader *hdr = (const GridHeader *)base;
        const float *__restrict__ cloud = xyz + (size_t)b * N * 3;
        // ---- chunk stage: the 16 queries' coordinates, cells, runs ------------------------------------------------------
        const __amdgpu_buffer_rsrc_t rs_q = ball_rsrc(new_xyz + qg0 * 3, (unsigned)nq * 12u);
        const float cx = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_q, (unsigned)ql * 12u + 0u, 0, 0));
        const float cy = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_q, (unsigned)ql * 12u + 4u, 0, 0));
        const float cz = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_q, (unsigned)ql * 12u + 8u, 0, 0));
        const bool fin = fabsf(cx) <= 3.0e38f && fabsf(cy) <= 3.0e38f && fabsf(cz) <= 3.0e38f;
        // queries that take the index-order scan: the cloud has no grid, or the query is not finite (bit 4 * query)
        unsigned long long scanq = __builtin_amdgcn_ballot_w64(!fin && dyi == 0 && ql < nq);
        if (hdr->use_scan) scanq = 0x1111111111111111ull;
        {
            const __amdgpu_buffer_rsrc_t rs_cells = ball_rsrc(base + sizeof(GridHeader), (kGridCells + 1) * 4u);
            const int gx = hdr->g[0], gy = hdr->g[1], gz = hdr->g[2];
            const float inv_h = hdr->inv_h;
            const int qx = cell_coord(cx, hdr->lo[0], inv_h, gx);
            const int qy = cell_coord(cy, hdr->lo[1], inv_h, gy);
            const int qz = cell_coord(cz, hdr->lo[2], inv_h, gz);
            const int x0 = max(qx - 1, 0), x1p = min(qx + 1, gx - 1) + 1;
            const int yy = qy + dyi - 1;
            const bool ok_l = fin && ql < nq && dyi < 3 && x0 < x1p && (unsigned)yy < (unsigned)gy && !hdr->use_scan;
            int rs[3], len[3];
#pragma unroll
            for (int ps = 0; ps < 3; ++ps) {
                const int zz = qz + ps - 1;
                const bool ok = ok_l && (unsigned)zz < (unsigned)gz;
                const unsigned c0 = (unsigned)((zz * gy + yy) * gx);
                // lanes without a run read out of range: the hardware returns 0 for both ends -> an empty run
                rs[ps] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_cells, ok ? (c0 + (unsigned)x0) * 4u : 0xFFFFFFF0u, 0, 0);
                len[ps] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_cells, ok ? (c0 + (unsigned)x1p) * 4u : 0xFFFFFFF0u, 0, 0) - rs[ps];
            }
            // inclusive prefix of the nine run lengths of a query, run r = 3 * pass + dyi.  Inside the quad: the length one and
            // two lanes to the left (lane 3 of a quad has no run, its length is 0: it stands in for "nothing to the left"),
            // then the totals of the earlier passes (lane 2 of the quad holds a pass total)
            int e[3];
#pragma unroll
            for (int ps = 0; ps < 3; ++ps)
                e[ps] = len[ps] + dpp_zero_i<0x93, 0xF, 0xF>(len[ps])      // quad_perm [3,0,1,2]
                        + dpp_zero_i<0x4F, 0xF, 0xF>(len[ps]);             // quad_perm [3,3,0,1]
            const int t0 = dpp_zero_i<0xAA, 0xF, 0xF>(e[0]), t1 = dpp_zero_i<0xAA, 0xF, 0xF>(e[1]);   // quad_perm [2,2,2,2]
            e[1] += t0;
            e[2] += t0 + t1;
            // the query block: entry r = end | (record of the run's first candidate - its position in the list) << 16; the
            // ninth entry's end is T, the number of candidates
            unsigned char *qb = lds + kOffQ + ql * kQBlk;
#pragma unroll
            for (int ps = 0; ps < 3; ++ps) {
                const int delta = rs[ps] - (e[ps] - len[ps]);
                const unsigned ent = ((unsigned)delta << 16) | (unsigned)e[ps];
                *(unsigned *)(qb + (dyi < 3 ? (ps * 3 + dyi) * 4 : 40)) = ent;    // (lane 3 of the quad: a dummy word)
            }
            *(float4 *)(qb + 48) = make_float4(cx, cy, cz, sumsq3(cx, cy, cz));   // (the four lanes of a quad write the same words)
        }
        wave_lds_fence();
        // ---- the queries of the chunk ---------------------------------------------------------------------------------------
        const __amdgpu_buffer_rsrc_t rs_rec = ball_rsrc(base + kGridRecOff, (unsigned)N * 16u);
        const __amdgpu_buffer_rsrc_t rs_idx = ball_rsrc(base + grid_idx_off(N), (unsigned)N * 4u);
        for (int i = 0; i < nq; ++i) {
            IdxT *__restrict__ orow = out + (qg0 + i) * K;
            const unsigned char *qb = lds + kOffQ + i * kQBlk;            // wave-uniform
            if ((scanq >> (4 * i)) & 1ull) {
                const float4 qc = *(const float4 *)(qb + 48);
                ball_scan_row<IdxT>(N, K, r2, cloud, qc.x, qc.y, qc.z, orow, lane);
                continue;
            }
            const __amdgpu_buffer_rsrc_t rs_out = ball_rsrc(orow, (unsigned)K * (unsigned)sizeof(IdxT));
            auto put = [&](int r, int v) {   // orow[r] = v
                if constexpr (sizeof(IdxT) == 8)
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{(unsigned)v, 0u}, rs_out, (unsigned)r * 8u, 0, 0);   // indices are >= 0
                else
                    __builtin_amdgcn_raw_buffer_store_b32((unsigned)v, rs_out, (unsigned)r * 4u, 0, 0);
            };
            const float4 qc = *(const float4 *)(qb + 48);                 // broadcast read: (cx, cy, cz, |c|^2)
            const int T = __builtin_amdgcn_readfirstlane((int)*(const unsigned short *)(qb + 32));
            const int e3 = (int)*(const unsigned short *)(qb + 12), e7 = (int)*(const unsigned short *)(qb + 28);
            const int d8 = (int)*(const short *)(qb + 34);
            // flattened candidate g -> byte offset of its record.  run(g) = #{t < 8 : end[t] <= g}: three binary-search steps
            // over ends 0..6 in the query block, one compare against end 7 for the last run.  Written for three candidates at a
            // time, step by step, so that the three dependent LDS reads of each overlap (NB of three batches); the empty asm
            // statements keep the compiler from sinking a batch's search into a branch of its own (it then runs them one
            // after the other, each waiting for its own reads).
            // Candidates beyond T compute an offset behind the last run: whatever they read is masked by `g < T`.
            auto record_off3 = [&](const int (&g)[3], unsigned (&off)[3]) {
                const unsigned qo = (unsigned)(kOffQ + i * kQBlk);
                unsigned a[3];
                int t[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) a[k] = g[k] >= e3 ? qo + 16u : qo;           // run in 4..7 | 0..3
#pragma unroll
                for (int k = 0; k < 3; ++k) t[k] = (int)*(const unsigned short *)(lds + a[k] + 4);
#pragma unroll
                for (int k = 0; k < 3; ++k) a[k] += g[k] >= t[k] ? 8u : 0u;
#pragma unroll
                for (int k = 0; k < 3; ++k) t[k] = (int)*(const unsigned short *)(lds + a[k]);
#pragma unroll
                for (int k = 0; k < 3; ++k) a[k] += g[k] >= t[k] ? 4u : 0u;
#pragma unroll
                for (int k = 0; k < 3; ++k) t[k] = (int)*(const short *)(lds + a[k] + 2);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    asm volatile("" : "+v"(t[k]));
                    off[k] = (unsigned)(g[k] + (g[k] >= e7 ? d8 : t[k])) << 4;
                }
            };
            auto dist = [&](const u32x4 &pp) -> float {
                float d = sqdist_expanded(qc.x, qc.y, qc.z, qc.w, __uint_as_float(pp[0]), __uint_as_float(pp[1]),
                                          __uint_as_float(pp[2]), __uint_as_float(pp[3]));
                asm volatile("" : "+v"(d));     // evaluated in every lane: no branch around six instructions
                return d;
            };
            u32x4 p[3];
            int pi[3];
            {
                const int g[3] = {lane, lane + 64, lane + 128};
                unsigned off[3];
                record_off3(g, off);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    p[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_rec, off[k], 0, 0);
                    pi[k] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_idx, off[k] >> 2, 0, 0);
                }
            }
            int H = 0;
            // (the hit mask is the AND of two compare results in scalar registers; a ballot of the combined predicate makes the
            // compiler rebuild the mask from a 0/1 select)
            auto record = [&](int g, float d, int pidx, bool capped) {   // set the bit, list the hit
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(g < T) & __builtin_amdgcn_ballot_w64(!(d > r2));
                const bool hit = (g < T) & !(d > r2);
                if (hit) {
                    atomicOr(&bm[(unsigned)pidx >> 5], 1u << (pidx & 31));
                    const int pos = H + mbcnt(mask);
                    if (!capped || pos < kBmHitCap) hits[pos] = (unsigned short)pidx;
                }
                H += __popcll(mask);
            };
#pragma unroll
            for (int k = 0; k < 3; ++k) record(64 * k + lane, dist(p[k]), pi[k], false);   // (192 <= kBmHitCap: no cap test)
            // rank of index v = number of set bits below it (valid once the word bases are up to date)
            auto rank_of = [&](int v) -> int {
                const unsigned w = bm[(unsigned)v >> 5];
                const unsigned below = (1u << ((unsigned)v & 31u)) - 1u;   // the low v % 32 bits
                return (int)wbase[(unsigned)v >> 5] + __popc(w & below);
            };
            // the candidates beyond the first 192 (pass 0), or every candidate again (pass 1: the hit list overflowed and each
            // hit is ranked straight from the bitmap)
            auto walk = [&](const int pass) {
                for (int g0 = pass == 0 ? 192 : 0; g0 < T; g0 += 3 * kWave) {
                    const int g[3] = {g0 + lane, g0 + lane + 64, g0 + lane + 128};
                    unsigned off[3];
                    record_off3(g, off);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        if (g0 + 64 * k >= T) break;     // wave-uniform
                        const u32x4 pp = __builtin_amdgcn_raw_buffer_load_b128(rs_rec, off[k], 0, 0);
                        const int pidx = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_idx, off[k] >> 2, 0, 0);
                        const float d = dist(pp);
                        if (pass == 0) {
                            record(g[k], d, pidx, true);
                        } else if ((g[k] < T) & !(d > r2)) {
                            const int r = rank_of(pidx);
                            if (r < K) put(r, pidx);
                            if (r == 0) *(int *)(lds + kOffFirst) = pidx;
                        }
                    }
                }
            };
            if (T > 192) walk(0);
            if (H == 0) {   // wave-uniform; the bitmap is still clean
                for (int j = lane; j < K; j += kWave) put(j, N);  // no hit at all -> N (pointnet2_utils.py:136-141)
                continue;
            }
            wave_lds_fence();   // the bits were set by whichever lanes held the hits
            {   // exclusive prefix of the popcounts of the bitmap's words: the owner of words [lane * 4 NQ, (lane + 1) * 4 NQ)
                // chains its v_bcnt, a wave scan turns the lane totals into lane bases, the bases are written as u16 pairs
                unsigned pre[4 * NQ];
                unsigned run = 0;
#pragma unroll
                for (int t = 0; t < NQ; ++t) {
                    const u32x4 w = *(const u32x4 *)&bm[(lane * NQ + t) * 4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        pre[t * 4 + u] = run;
                        run += (unsigned)__popc(w[u]);        // (one v_bcnt_u32_b32 with its accumulate operand)
                    }
                }
                unsigned incl = run;   // inclusive scan over the 64 lanes
                incl += dpp_or_zero<0x111, 0xF>(incl);
                incl += dpp_or_zero<0x112, 0xF>(incl);
                incl += dpp_or_zero<0x114, 0xF>(incl);
                incl += dpp_or_zero<0x118, 0xF>(incl);
                incl += dpp_or_zero<0x142, 0xA>(incl);   // row_bcast:15 into rows 1, 3
                incl += dpp_or_zero<0x143, 0xC>(incl);   // row_bcast:31 into rows 2, 3
                const unsigned lb = incl - run, lb2 = (lb << 16) | lb;
#pragma unroll
                for (int t = 0; t < NQ; ++t) {
                    const unsigned a = ((pre[t * 4 + 1] << 16) + pre[t * 4 + 0]) + lb2, bq = ((pre[t * 4 + 3] << 16) + pre[t * 4 + 2]) + lb2;
                    *(u32x2 *)(lds + kOffBase + (lane * NQ + t) * 8) = u32x2{a, bq};
                }
            }
            wave_lds_fence();   // the bases and the hit list cross lanes
            const bool pad = H < K;   // wave-uniform: the row is padded with its first entry, the hit of rank 0
            if (H <= kBmHitCap) {
                for (int k = lane; k < H; k += kWave) {
                    const int v = hits[k];
                    const int r = rank_of(v);
                    if (r < K) put(r, v);
                    if (pad && r == 0) *(int *)(lds + kOffFirst) = v;
                }
            } else {
                walk(1);
            }
            if (pad) {   // pad with the first hit = the smallest index (pointnet2_utils.py:138-141)
                wave_lds_fence();
                const int fv = *(const int *)(lds + kOffFirst);
                for (int j = H + lane; j < K; j += kWave) put(j, fv);
            }
            wave_lds_fence();   // every rank has been read: the bitmap can be wiped for the next query
#pragma unroll
            for (int t = 0; t < NQ; ++t) *(u32x4 *)&bm[(lane * NQ + t) * 4] = u32x4{0u, 0u, 0u, 0u};
            wave_lds_fence();
        }
        wave_lds_fence();   // the query blocks are rewritten by the next chunk
    }
}

static bool use_grid(int N, int S, int K) {
    // the grid pays off once the per-query scan (N/64 steps) clearly exceeds a neighbourhood visit
    return N >= 2048 && (long long)S * 8 >= N / 64 && K <= kGridMaxK;
}

}  // namespace tgn

using namespace tgn;

TGN_API size_t tgn_ball_query_workspace_bytes(int B, int N, int S) {
    if (B <= 0 || N <= 0 || S <= 0) return 0;
    return (size_t)B * grid_cloud_bytes(N);
}

static int ball_query_impl(int B, int N, int S, int nsample, float r2, const float *xyz, const float *new_xyz, void *idx,
                           int idx_is_int64, void *workspace, size_t workspace_bytes, hipStream_t st, bool build, bool query) {
    if (B < 0 || N < 0 || S < 0 || nsample < 0) {
        set_error("tgn_ball_query: negative size");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    const long long total = (long long)B * S;
    if (total == 0 || nsample == 0) return TGN_OK;
    if (!xyz || (query && (!new_xyz || !idx))) {
        set_error("tgn_ball_query: null pointer");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    long long blocks = ((total + 3) / 4 + 7) / 8 * 8;  // a multiple of the 8 XCDs (ball_grid_query_kernel)
    if (blocks > 256 * 64) blocks = 256 * 64;
    const bool grid = use_grid(N, S, nsample) && workspace && workspace_bytes >= (size_t)B * grid_cloud_bytes(N);
    if (grid) {
        if (build) {
            const size_t perm_bytes = N <= kGridPermCap ? ((size_t)N * 2 + 15) / 16 * 16 : 0;   // the LDS permutation of step 4
            // 64 KiB cell table + up to 64 KiB permutation: past the 64-KiB default of dynamic + static LDS.  The attribute belongs to
            // the (function, device) pair, so it is set on every launch (a host-side table look-up; a once-per-process flag would
            // leave a second GPU of the process without it)
            if (perm_bytes && hipFuncSetAttribute((const void *)ball_grid_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  kGridPermCap * 2) != hipSuccess) {
                (void)hipGetLastError();
                set_error("tgn_ball_query: cannot raise the grid build kernel's dynamic LDS limit to %d bytes", kGridPermCap * 2);
                return TGN_ERR_LAUNCH;
            }
            hipLaunchKernelGGL(ball_grid_build_kernel, dim3(B), dim3(kGridThreads), perm_bytes, st, N, r2, xyz,
                               (unsigned char *)workspace);
            if (int rc = check_launch("ball_grid_build_kernel")) return rc;
        }
        if (!query) return TGN_OK;
        const int bitmap_ok = tuning(kTuneBallBitmap);   // 2: the chunked bitmap kernel; 1: round 2's bitmap kernel; 0: rank-select (experiments)
        if (bitmap_ok >= 2 && N <= kBmMaxN) {
            const int nquad = (N + 8191) >> 13;
            const int cpc = (S + kChunkQ - 1) / kChunkQ;
            const long long total_chunks = (long long)B * cpc;
            const long long cpx = B >= 8 ? (long long)((B + 7) / 8) * cpc : (total_chunks + 7) / 8;
            long long wpx = cpx < 1024 ? cpx : 1024;   // waves per XCD: what is resident at a time (32 CUs x <= 32), not more
            const unsigned nblk = (unsigned)(wpx * 8);
#define TGN_BALL_CHUNK(T_, NQ_)                                                                                            \
    hipLaunchKernelGGL((ball_grid_query_chunk_kernel<T_, NQ_>), dim3(nblk), dim3(64), 0, st, B, N, S, nsample, r2, xyz, \
                       new_xyz, (const unsigned char *)workspace, (T_ *)idx, (int)cpx)
            if (idx_is_int64) {
                if (nquad == 1) TGN_BALL_CHUNK(long long, 1);
                else if (nquad == 2) TGN_BALL_CHUNK(long long, 2);
                else if (nquad == 3) TGN_BALL_CHUNK(long long, 3);
                else TGN_BALL_CHUNK(long long, 4);
            } else {
                if (nquad == 1) TGN_BALL_CHUNK(int, 1);
                else if (nquad == 2) TGN_BALL_CHUNK(int, 2);
                else if (nquad == 3) TGN_BALL_CHUNK(int, 3);
                else TGN_BALL_CHUNK(int, 4);
            }
#undef TGN_BALL_CHUNK
            return check_launch("ball_grid_query_chunk_kernel");
        }
        if (bitmap_ok && N <= kBmMaxN) {
            const int nquad = (N + 8191) >> 13;
            const size_t lds = (size_t)4 * (nquad * 256 + nquad * 64 + kBmHitCap) * sizeof(unsigned);   // <= 24 KiB per workgroup
            const long long q_per_xcd = B >= 8 ? (long long)((B + 7) / 8) * S : ((total + 7) / 8 + 3) / 4 * 4;
            long long nbx = (q_per_xcd + 3) / 4;   // blocks per XCD: what is resident at a time (32 CUs x 8), not more
            if (nbx > 256) nbx = 256;
            const unsigned grid = (unsigned)(nbx * 8);
            if (idx_is_int64)
                hipLaunchKernelGGL((ball_grid_query_bitmap_kernel<long long>), dim3(grid), dim3(256), lds, st, B, N, S,
                                   nsample, r2, xyz, new_xyz, (const unsigned char *)workspace, (long long *)idx, q_per_xcd);
            else
                hipLaunchKernelGGL((ball_grid_query_bitmap_kernel<int>), dim3(grid), dim3(256), lds, st, B, N, S, nsample,
                                   r2, xyz, new_xyz, (const unsigned char *)workspace, (int *)idx, q_per_xcd);
            return check_launch("ball_grid_query_bitmap_kernel");
        }
        if (idx_is_int64)
            hipLaunchKernelGGL((ball_grid_query_kernel<long long>), dim3((unsigned)blocks), dim3(256), 0, st, B, N, S,
                               nsample, r2, xyz, new_xyz, (const unsigned char *)workspace, (long long *)idx);
        else
            hipLaunchKernelGGL((ball_grid_query_kernel<int>), dim3((unsigned)blocks), dim3(256), 0, st, B, N, S, nsample,
                               r2, xyz, new_xyz, (const unsigned char *)workspace, (int *)idx);
        return check_launch("ball_grid_query_kernel");
    }
    if (!query) return TGN_OK;
    if (idx_is_int64)
        hipLaunchKernelGGL((ball_query_scan_kernel<long long>), dim3((unsigned)blocks), dim3(256), 0, st, B, N, S,
                           nsample, r2, xyz, new_xyz, (long long *)idx);
    else
        hipLaunchKernelGGL((ball_query_scan_kernel<int>), dim3((unsigned)blocks), dim3(256), 0, st, B, N, S, nsample, r2,
                           xyz, new_xyz, (int *)idx);
    return check_launch("ball_query_scan_kernel");
}

TGN_API int tgn_ball_query(int B, int N, int S, int nsample, float r2, const float *xyz, const float *new_xyz,
                           void *idx, int idx_is_int64, void *workspace, size_t workspace_bytes,
                           tgn_stream_t stream) {
    return ball_query_impl(B, N, S, nsample, r2, xyz, new_xyz, idx, idx_is_int64, workspace, workspace_bytes,
                           (hipStream_t)stream, true, true);
}

// The two halves of tgn_ball_query for callers that schedule them apart: the grid depends on the cloud and the radius
// only, not on the queries, so a planner can build it while the sampling that produces the queries is still running.
TGN_API int tgn_ball_query_build(int B, int N, int S, int nsample, float r2, const float *xyz, void *workspace,
                                 size_t workspace_bytes, tgn_stream_t stream) {
    return ball_query_impl(B, N, S, nsample, r2, xyz, nullptr, nullptr, 0, workspace, workspace_bytes, (hipStream_t)stream,
                           true, false);
}

TGN_API int tgn_ball_query_prebuilt(int B, int N, int S, int nsample, float r2, const float *xyz, const float *new_xyz,
                                    void *idx, int idx_is_int64, void *workspace, size_t workspace_bytes,
                                    tgn_stream_t stream) {
    return ball_query_impl(B, N, S, nsample, r2, xyz, new_xyz, idx, idx_is_int64, workspace, workspace_bytes,
                           (hipStream_t)stream, false, true);
}

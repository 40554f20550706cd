// neighbors.hip -- neighbour searches: kNN, three_nn, square_distance (gfx950).  Ball query: ball_query.hip.
//
// kNN        : pointops.knnquery (pointops.py:30-45 -> knnquery_cuda_kernel.cu:65-108), heap semantics
//              preserved exactly (insertion history decides the order of equal distances).
// three_nn   : the square_distance + full sort + [:3] of PointNetFeaturePropagation
//              (pointnet2_utils.py:333-335) as a running top-3 by (distance, index).
#include "tgn_common.h"
#include <stdlib.h>

namespace tgn {

// ---------------------------------------------------------------------------------------------
// kNN, exact heap semantics.  One thread per query; the heap lives in LDS (column per thread) so the
// data-dependent sift never touches scratch memory.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void knn_reheap(float *hd, int *hi, int stride, int k) {
    // knnquery_cuda_kernel.cu:21-36
    int root = 0;
    int child = 1;
    while (child < k) {
        if (child + 1 < k && hd[(child + 1) * stride] > hd[child * stride]) child++;
        if (hd[root * stride] > hd[child * stride]) return;
        const float td = hd[root * stride];
        hd[root * stride] = hd[child * stride];
        hd[child * stride] = td;
        const int ti = hi[root * stride];
        hi[root * stride] = hi[child * stride];
        hi[child * stride] = ti;
        root = child;
        child = root * 2 + 1;
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void knn_heap_kernel(int b, int m, int nsample, const float *__restrict__ xyz,
                                                       const float *__restrict__ new_xyz,
                                                       const int *__restrict__ offset,
                                                       const int *__restrict__ new_offset, int *__restrict__ idx,
                                                       float *__restrict__ dist2, const int *__restrict__ only) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *hd = (float *)smem + threadIdx.x;
    int *hi = (int *)smem + (size_t)nsample * NT + threadIdx.x;
    int pt = blockIdx.x * NT + threadIdx.x;
    if (only) {  // second pass of the wave kernel: only[0] = count, only[1..] = the queries to redo exactly
        if (pt >= only[0]) return;
        pt = only[1 + pt];
    }
    if (pt >= m) return;
    int bt = 0;  // get_bt_idx, knnquery_cuda_kernel.cu:51-62 (bounded by b here)
    while (bt < b - 1 && !(pt < new_offset[bt])) ++bt;
    const int start = bt == 0 ? 0 : offset[bt - 1];
    const int end = offset[bt];
    const float qx = new_xyz[(size_t)pt * 3 + 0], qy = new_xyz[(size_t)pt * 3 + 1], qz = new_xyz[(size_t)pt * 3 + 2];
    for (int i = 0; i < nsample; ++i) {
        hd[i * NT] = 1e10f;
        hi[i * NT] = start;
    }
    float root = 1e10f;
    for (int i = start; i < end; ++i) {
        const float x = xyz[(size_t)i * 3 + 0], y = xyz[(size_t)i * 3 + 1], z = xyz[(size_t)i * 3 + 2];
        const float ex = qx - x, ey = qy - y, ez = qz - z;
        const float d2 = dist_direct_nofma(ex, ey, ez);  // knnquery_cuda_kernel.cu:96
        if (d2 < root) {
            hd[0] = d2;
            hi[0] = i;
            knn_reheap(hd, hi, NT, nsample);
            root = hd[0];
        }
    }
    // heap_sort, knnquery_cuda_kernel.cu:39-48
    for (int i = nsample - 1; i > 0; --i) {
        const float td = hd[0];
        hd[0] = hd[i * NT];
        hd[i * NT] = td;
        const int ti = hi[0];
        hi[0] = hi[i * NT];
        hi[i * NT] = ti;
        knn_reheap(hd, hi, NT, i);
    }
    for (int i = 0; i < nsample; ++i) {
        idx[(size_t)pt * nsample + i] = hi[i * NT];
        dist2[(size_t)pt * nsample + i] = hd[i * NT];
    }
}

// ---------------------------------------------------------------------------------------------
// kNN, wave-parallel.  A wave owns kKnnQ consecutive queries and streams the segment once, 64 candidates
// per step (lane = candidate).  Per query the k+1 smallest (d2, index) so far are kept SORTED ACROSS LANES
// (lane i = i-th smallest); a candidate passing the `d2 < (k+1)-th smallest` test is inserted with one
// wave_shr:1 DPP shift.  The same arithmetic as the heap kernel gives the same k smallest distances; the
// heap's result differs only when two of those distances tie bit-for-bit (then the heap's insertion history
// decides which tied point survives and in what order): such queries -- equal neighbours inside the list,
// or k-th == (k+1)-th -- are appended to `redo` and recomputed by the exact heap kernel.  ~30x the
// throughput of one-thread-per-query on (24000 x 24000, k=36).
// ---------------------------------------------------------------------------------------------
constexpr int kKnnQ = 4;

__device__ __forceinline__ float dpp_wave_shr1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ int dpp_wave_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, false); }

__global__ __launch_bounds__(256) void knn_wave_kernel(int b, int m, int k, const float *__restrict__ xyz,
                                                        const float *__restrict__ new_xyz,
                                                        const int *__restrict__ offset,
                                                        const int *__restrict__ new_offset, int *__restrict__ idx,
                                                        float *__restrict__ dist2, int *__restrict__ redo) {
    const int lane = threadIdx.x & (kWave - 1);
    const int q0 = (blockIdx.x * (blockDim.x / kWave) + __builtin_amdgcn_readfirstlane(threadIdx.x / kWave)) * kKnnQ;
    if (q0 >= m) return;
    float qx[kKnnQ], qy[kKnnQ], qz[kKnnQ], ld[kKnnQ], tau[kKnnQ];
    int li[kKnnQ], st[kKnnQ], en[kKnnQ];
    int lo = 0x7FFFFFFF, hi = 0;
#pragma unroll
    for (int j = 0; j < kKnnQ; ++j) {
        const int q = min(q0 + j, m - 1);
        int bt = 0;  // get_bt_idx, knnquery_cuda_kernel.cu:51-62
        while (bt < b - 1 && !(q < new_offset[bt])) ++bt;
        st[j] = bt == 0 ? 0 : offset[bt - 1];
        en[j] = q0 + j < m ? offset[bt] : st[j];  // padding queries scan nothing
        qx[j] = new_xyz[(size_t)q * 3 + 0];
        qy[j] = new_xyz[(size_t)q * 3 + 1];
        qz[j] = new_xyz[(size_t)q * 3 + 2];
        ld[j] = 1e10f;  // knnquery_cuda_kernel.cu:88-91
        li[j] = st[j];
        tau[j] = 1e10f;
        lo = min(lo, st[j]);
        hi = max(hi, en[j]);
    }
    for (int base = lo; base < hi; base += kWave) {
        const int i = base + lane;
        float x = 0.f, y = 0.f, z = 0.f;
        if (i < hi) {
            x = xyz[(size_t)i * 3 + 0];
            y = xyz[(size_t)i * 3 + 1];
            z = xyz[(size_t)i * 3 + 2];
        }
#pragma unroll
        for (int j = 0; j < kKnnQ; ++j) {
            const float ex = qx[j] - x, ey = qy[j] - y, ez = qz[j] - z;
            const float d2 = dist_direct_nofma(ex, ey, ez);  // knnquery_cuda_kernel.cu:96
            unsigned long long mask = __ballot(i >= st[j] && i < en[j] && d2 < tau[j]);
            while (mask) {  // wave-uniform; candidates in ascending index order
                const int src = __builtin_ctzll(mask);
                mask &= mask - 1;
                const float dn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d2), src));
                if (!(dn < tau[j])) continue;  // the threshold may have dropped meanwhile
                const int in = base + src;
                // entries <= dn stay in front (an equal, earlier candidate keeps its place)
                const int pos = __popcll(__ballot(lane <= k && ld[j] <= dn));
                const float sd = dpp_wave_shr1(ld[j]);
                const int si = dpp_wave_shr1(li[j]);
                ld[j] = lane < pos ? ld[j] : (lane == pos ? dn : sd);
                li[j] = lane < pos ? li[j] : (lane == pos ? in : si);
                tau[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ld[j]), k));
            }
        }
    }
#pragma unroll
    for (int j = 0; j < kKnnQ; ++j) {
        const int q = q0 + j;
        if (q >= m) break;
        if (lane < k) {
            idx[(size_t)q * k + lane] = li[j];
            dist2[(size_t)q * k + lane] = ld[j];
        }
        // exact ties decide nothing here; the heap kernel settles them
        const float nd = __shfl_down(ld[j], 1);
        const int ni = __shfl_down(li[j], 1);
        const bool tie = lane < k && ld[j] == nd && li[j] != ni;
        if (__any(tie) && lane == 0) redo[1 + atomicAdd(&redo[0], 1)] = q;
    }
}

// ---------------------------------------------------------------------------------------------
// Grid kNN for large segments.  The brute-force kernel above evaluates every (query, point) pair (576 M at 24 000^2);
// on a scan surface the k nearest neighbours sit in the 3x3x3 cells around the query once the cell size is about the
// k-neighbour radius.  Per segment a uniform grid is built in LDS (bounding box -> histogram with LDS atomics -> scan ->
// scatter of 16-byte (x, y, z, index) records sorted by cell); a query wave scans the (2r+1)^2 x-runs of cells of the
// block of radius r around its cell with the same insertion into the lane-sorted (k+1)-list, and is done when the
// (k+1)-th distance lies inside the radius the block is guaranteed to cover (r cells, minus a margin for the rounding of
// the cell coordinates); otherwise the block radius doubles and the query starts over.  Distances are the same
// unfused expression on the same operands, candidates merely arrive in another order: wherever the k+1 smallest
// distances are distinct the result is the brute-force one, and queries with exact ties go to the heap kernel just
// like there.  Degenerate or tiny segments (flagged by the build) are scanned linearly by their query waves.
// ---------------------------------------------------------------------------------------------
constexpr int kKnnCells = 16384;
constexpr int kKnnBuildThreads = 1024;

__device__ __forceinline__ float wave_min_f32_x(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max_f32_x(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

struct KnnGridHeader {  // one per segment, 64 bytes
    float lo[3];
    float inv_h;
    int g[3];
    int use_scan;
    float h;
    int pad[7];
};

__host__ __device__ inline size_t knn_grid_off_headers(int m) { return (((size_t)m + 1) * sizeof(int) + 63) / 64 * 64; }
__host__ __device__ inline size_t knn_grid_off_cells(int b, int m) { return knn_grid_off_headers(m) + (size_t)b * sizeof(KnnGridHeader); }
__host__ __device__ inline size_t knn_grid_off_records(int b, int m) {
    return knn_grid_off_cells(b, m) + (size_t)b * (kKnnCells + 4) * sizeof(int);
}

__device__ __forceinline__ int knn_cell(float p, float lo, float inv_h, int g) {
    float t = (p - lo) * inv_h;           // the same expression for points and queries
    t = fminf(fmaxf(t, 0.0f), (float)(g - 1));  // NaN -> 0; queries outside the box go to the border cell
    return (int)t;
}

__global__ __launch_bounds__(kKnnBuildThreads) void knn_grid_build_kernel(int b_total, int m, int k, float scale, const float *__restrict__ xyz,
                                                                          const int *__restrict__ offset,
                                                                          unsigned char *__restrict__ ws) {
    __shared__ int cnt[kKnnCells];
    __shared__ float red[7][kKnnBuildThreads / kWave];
    __shared__ int wave_tot[kKnnBuildThreads / kWave];
    __shared__ KnnGridHeader hdr_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int st = b ? offset[b - 1] : 0, n = offset[b] - st;
    const float *__restrict__ pts = xyz + (size_t)st * 3;
    KnnGridHeader *hdr = (KnnGridHeader *)(ws + knn_grid_off_headers(m)) + b;
    int *cell_start = (int *)(ws + knn_grid_off_cells(b_total, m)) + (size_t)b * (kKnnCells + 4);
    float4 *rec = (float4 *)(ws + knn_grid_off_records(b_total, m)) + st;

    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float bad = 0.0f;
    for (int i = tid; i < n; i += kKnnBuildThreads) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pts[(size_t)i * 3 + a];
            if (!(fabsf(v) <= 1.0e18f)) bad = 1.0f;  // NaN, Inf or so large that squared distances overflow
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float l = wave_min_f32_x(lo[a]), h = wave_max_f32_x(hi[a]);
        if (lane == 0) {
            red[a][wave] = l;
            red[3 + a][wave] = h;
        }
    }
    {
        const float bb = wave_max_f32_x(bad);
        if (lane == 0) red[6][wave] = bb;
    }
    for (int i = tid; i < kKnnCells; i += kKnnBuildThreads) cnt[i] = 0;
    __syncthreads();
    if (tid == 0) {
        KnnGridHeader h;
        float ext[3], any_bad = 0.0f;
        for (int a = 0; a < 3; ++a) {
            float l = INFINITY, u = -INFINITY;
            for (int w = 0; w < kKnnBuildThreads / kWave; ++w) {
                l = fminf(l, red[a][w]);
                u = fmaxf(u, red[3 + a][w]);
            }
            h.lo[a] = l;
            ext[a] = u - l;
        }
        for (int w = 0; w < kKnnBuildThreads / kWave; ++w) any_bad = fmaxf(any_bad, red[6][w]);
        // cell size ~ radius that holds k points of a SURFACE of area xy + yz + zx (half the box surface: a thin sheet
        // gives its own area); too small a guess only costs a second, larger block for some queries
        const float area = ext[0] * ext[1] + ext[1] * ext[2] + ext[0] * ext[2];
        float hcell = scale * sqrtf(fmaxf(area, 0.0f) * (float)(k + 1) / (3.14159265f * fmaxf((float)n, 1.0f)));
        const float emax = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
        hcell = fmaxf(hcell, emax * (1.0f / 4096.0f));
        int g[3] = {1, 1, 1};
        bool ok = any_bad == 0.0f && n >= 512 && hcell > 0.0f && hcell < INFINITY;
        if (ok) {
            for (int it = 0; it < 80; ++it) {
                const float inv = 1.0f / hcell;
                long long cells = 1;
                for (int a = 0; a < 3; ++a) {
                    const float t = ext[a] * inv;
                    g[a] = (t < 1.0e6f) ? (int)t + 1 : 1000001;
                    cells *= g[a];
                }
                if (cells <= kKnnCells) break;
                hcell *= 1.1f;
            }
            long long cells = (long long)g[0] * g[1] * g[2];
            ok = cells <= kKnnCells && cells >= 8;
        }
        h.h = hcell;
        h.inv_h = ok ? 1.0f / hcell : 0.0f;
        for (int a = 0; a < 3; ++a) h.g[a] = ok ? g[a] : 1;
        h.use_scan = ok ? 0 : 1;
        for (int i = 0; i < 7; ++i) h.pad[i] = 0;
        hdr_s = h;
        *hdr = h;
    }
    __syncthreads();
    const KnnGridHeader h = hdr_s;
    if (h.use_scan) return;
    for (int i = tid; i < n; i += kKnnBuildThreads) {
        const int cx = knn_cell(pts[(size_t)i * 3 + 0], h.lo[0], h.inv_h, h.g[0]);
        const int cy = knn_cell(pts[(size_t)i * 3 + 1], h.lo[1], h.inv_h, h.g[1]);
        const int cz = knn_cell(pts[(size_t)i * 3 + 2], h.lo[2], h.inv_h, h.g[2]);
        atomicAdd(&cnt[(cz * h.g[1] + cy) * h.g[0] + cx], 1);
    }
    __syncthreads();
    constexpr int PER = kKnnCells / kKnnBuildThreads;
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
        cnt[tid * PER + i] = v;
    }
    if (tid == kKnnBuildThreads - 1) cell_start[kKnnCells] = thread_base + sum;
    __syncthreads();
    for (int i = tid; i < n; i += kKnnBuildThreads) {
        const float px = pts[(size_t)i * 3 + 0], py = pts[(size_t)i * 3 + 1], pz = pts[(size_t)i * 3 + 2];
        const int cx = knn_cell(px, h.lo[0], h.inv_h, h.g[0]);
        const int cy = knn_cell(py, h.lo[1], h.inv_h, h.g[1]);
        const int cz = knn_cell(pz, h.lo[2], h.inv_h, h.g[2]);
        const int pos = atomicAdd(&cnt[(cz * h.g[1] + cy) * h.g[0] + cx], 1);
        rec[pos] = make_float4(px, py, pz, __int_as_float(st + i));  // the packed (global) point index
    }
}

// insert candidate (dn, in) into the lane-sorted list (entries <= dn stay in front), as in knn_wave_kernel
__device__ __forceinline__ void knn_insert(float &ld, int &li, float &tau, float dn, int in, int k, int lane) {
    // lanes 0..k hold the list: the lane mask is applied to the ballot as a (wave-uniform) constant, and the shifted-in
    // value of lane 0 is never used, so the DPP moves need no defined "old" operand
    const unsigned long long kmask = (k >= 63) ? ~0ull : ((2ull << k) - 1ull);
    const int pos = __popcll(__builtin_amdgcn_ballot_w64(ld <= dn) & kmask);
    const float sd = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ld), __float_as_int(ld), 0x138, 0xF, 0xF, false));
    const int si = __builtin_amdgcn_update_dpp(li, li, 0x138, 0xF, 0xF, false);
    ld = lane < pos ? ld : (lane == pos ? dn : sd);
    li = lane < pos ? li : (lane == pos ? in : si);
    tau = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ld), k));
}

__global__ __launch_bounds__(256) void knn_grid_query_kernel(int b, int m, int k, const float *__restrict__ xyz,
                                                              const float *__restrict__ new_xyz,
                                                              const int *__restrict__ offset,
                                                              const int *__restrict__ new_offset,
                                                              const unsigned char *__restrict__ ws, int *__restrict__ idx,
                                                              float *__restrict__ dist2, int *__restrict__ redo) {
    const int lane = threadIdx.x & (kWave - 1);
    const int q = blockIdx.x * (blockDim.x / kWave) + __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    if (q >= m) return;
    int bt = 0;  // get_bt_idx, knnquery_cuda_kernel.cu:51-62
    while (bt < b - 1 && !(q < new_offset[bt])) ++bt;
    const int st = bt == 0 ? 0 : offset[bt - 1], en = offset[bt];
    const float qx = new_xyz[(size_t)q * 3 + 0], qy = new_xyz[(size_t)q * 3 + 1], qz = new_xyz[(size_t)q * 3 + 2];
    const KnnGridHeader *hdr = (const KnnGridHeader *)(ws + knn_grid_off_headers(m)) + bt;
    float ld = 1e10f, tau = 1e10f;  // knnquery_cuda_kernel.cu:88-91
    int li = st;
    const bool q_ok = fabsf(qx) <= 1.0e18f && fabsf(qy) <= 1.0e18f && fabsf(qz) <= 1.0e18f;
    if (hdr->use_scan || !q_ok) {
        for (int base = st; base < en; base += kWave) {  // linear scan of the segment, ascending index
            const int i = base + lane;
            float d2 = 0.0f;
            if (i < en) {
                const float ex = qx - xyz[(size_t)i * 3 + 0], ey = qy - xyz[(size_t)i * 3 + 1], ez = qz - xyz[(size_t)i * 3 + 2];
                d2 = dist_direct_nofma(ex, ey, ez);
            }
            unsigned long long mask = __ballot(i < en && d2 < tau);
            while (mask) {
                const int src = __builtin_ctzll(mask);
                mask &= mask - 1;
                const float dn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d2), src));
                if (!(dn < tau)) continue;
                knn_insert(ld, li, tau, dn, base + src, k, lane);
            }
        }
    } else {
        const int *__restrict__ cell_start = (const int *)(ws + knn_grid_off_cells(b, m)) + (size_t)bt * (kKnnCells + 4);
        const float4 *__restrict__ rec = (const float4 *)(ws + knn_grid_off_records(b, m)) + st;
        const int gx = hdr->g[0], gy = hdr->g[1], gz = hdr->g[2];
        const float inv_h = hdr->inv_h, h = hdr->h;
        const int cx = knn_cell(qx, hdr->lo[0], inv_h, gx), cy = knn_cell(qy, hdr->lo[1], inv_h, gy),
                  cz = knn_cell(qz, hdr->lo[2], inv_h, gz);
        // a query outside the box sits in a border cell although it is farther away: what a block of radius r covers
        // is measured from the box then (all points are inside it)
        const float ox = fmaxf(fmaxf(hdr->lo[0] - qx, qx - (hdr->lo[0] + (float)gx * h)), 0.0f);
        const float oy = fmaxf(fmaxf(hdr->lo[1] - qy, qy - (hdr->lo[1] + (float)gy * h)), 0.0f);
        const float oz = fmaxf(fmaxf(hdr->lo[2] - qz, qz - (hdr->lo[2] + (float)gz * h)), 0.0f);
        const float outside = fmaxf(ox, fmaxf(oy, oz));
        // blocks of radius 1, 2, 4, ...: a wider block only adds its SHELL (the cells of the previous block are not
        // visited again), so the list and its threshold carry over
        for (int r = 1, rp = -1;; rp = r, r *= 2) {
            const int x0 = max(cx - r, 0), x1 = min(cx + r, gx - 1);
            const int side = 2 * r + 1, nruns = side * side;
            for (int t0 = 0; t0 < nruns; t0 += kWave) {
                // lane t: one (dy, dz) run of the block; inside the previous block's (dy, dz) range only the
                // two x-pieces left and right of it are new
                int rs[2] = {0, 0}, re[2] = {0, 0};
                const int t = t0 + lane;
                if (t < nruns) {
                    const int tt = (t + nruns / 2) % nruns;  // the query's own run first: the threshold drops at once
                    const int dy = (tt % side) - r, dz = (tt / side) - r;
                    const int yy = cy + dy, zz = cz + dz;
                    if (yy >= 0 && yy < gy && zz >= 0 && zz < gz) {
                        const int c0 = (zz * gy + yy) * gx;
                        if (rp >= 0 && abs(dy) <= rp && abs(dz) <= rp) {
                            const int l1 = min(cx - rp - 1, gx - 1), r0 = max(cx + rp + 1, 0);
                            if (x0 <= l1) {
                                rs[0] = cell_start[c0 + x0];
                                re[0] = cell_start[c0 + l1 + 1];
                            }
                            if (r0 <= x1) {
                                rs[1] = cell_start[c0 + r0];
                                re[1] = cell_start[c0 + x1 + 1];
                            }
                        } else {
                            rs[0] = cell_start[c0 + x0];
                            re[0] = cell_start[c0 + x1 + 1];
                        }
                    }
                }
#pragma unroll
                for (int part = 0; part < 2; ++part) {
                    unsigned long long live = __ballot(re[part] > rs[part]);
                    while (live) {  // wave-uniform: the non-empty runs of this chunk
                        const int rl = __builtin_ctzll(live);
                        live &= live - 1;
                        const int a = __builtin_amdgcn_readlane(rs[part], rl), e = __builtin_amdgcn_readlane(re[part], rl);
                        for (int j0 = a; j0 < e; j0 += kWave) {
                            const int j = j0 + lane;
                            float d2 = 0.0f;
                            int pi = 0;
                            if (j < e) {
                                const float4 p = rec[j];
                                const float ex = qx - p.x, ey = qy - p.y, ez = qz - p.z;
                                d2 = dist_direct_nofma(ex, ey, ez);  // knnquery_cuda_kernel.cu:96
                                pi = __float_as_int(p.w);
                            }
                            unsigned long long mask = __ballot(j < e && d2 < tau);
                            while (mask) {
                                const int src = __builtin_ctzll(mask);
                                mask &= mask - 1;
                                const float dn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d2), src));
                                if (!(dn < tau)) continue;
                                knn_insert(ld, li, tau, dn, __builtin_amdgcn_readlane(pi, src), k, lane);
                            }
                        }
                    }
                }
            }
            const bool whole = x0 == 0 && x1 == gx - 1 && cy - r <= 0 && cy + r >= gy - 1 && cz - r <= 0 && cz + r >= gz - 1;
            if (whole) break;
            // every point within (r cells) of the query's cell boundary has been seen; 0.999 absorbs the rounding of
            // the cell coordinates, `outside` the distance of a query beyond the box to its (border) cell
            const float cover = (float)r * h * 0.999f - outside;
            if (cover > 0.0f && tau < cover * cover) break;
        }
    }
    if (lane < k) {
        idx[(size_t)q * k + lane] = li;
        dist2[(size_t)q * k + lane] = ld;
    }
    const float nd = __shfl_down(ld, 1);
    const int ni = __shfl_down(li, 1);
    const bool tie = lane < k && ld == nd && li != ni;
    if (__any(tie) && lane == 0) redo[1 + atomicAdd(&redo[0], 1)] = q;
}

// ---------------------------------------------------------------------------------------------
// three_nn.  One thread per query; support points staged through LDS as (x,y,z,|p|^2).
// ---------------------------------------------------------------------------------------------
constexpr int kTnnTile = 1024;

template <typename IdxT>
__global__ __launch_bounds__(256) void three_nn_kernel(int B, int N, int S, const float *__restrict__ xyz1,
                                                        const float *__restrict__ xyz2, float *__restrict__ dist,
                                                        IdxT *__restrict__ idx) {
    __shared__ float4 tile[kTnnTile];
    const int b = blockIdx.y;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = q < N;
    float cx = 0.f, cy = 0.f, cz = 0.f, s1 = 0.f;
    if (active) {
        const float *c = xyz1 + ((size_t)b * N + q) * 3;
        cx = c[0];
        cy = c[1];
        cz = c[2];
        s1 = sumsq3(cx, cy, cz);
    }
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    int i0 = 0, i1 = 0, i2 = 0;
    for (int t0 = 0; t0 < S; t0 += kTnnTile) {
        const int cnt = min(kTnnTile, S - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const float *p = xyz2 + ((size_t)b * S + t0 + i) * 3;
            const float px = p[0], py = p[1], pz = p[2];
            tile[i] = make_float4(px, py, pz, sumsq3(px, py, pz));
        }
        __syncthreads();
        if (active) {
            for (int i = 0; i < cnt; ++i) {
                const float4 p = tile[i];
                const float d = sqdist_expanded(cx, cy, cz, s1, p.x, p.y, p.z, p.w);
                // strict '<' in ascending index order keeps the earlier index on ties
                if (d < d2) {
                    const int k = t0 + i;
                    if (d < d1) {
                        d2 = d1;
                        i2 = i1;
                        if (d < d0) {
                            d1 = d0;
                            i1 = i0;
                            d0 = d;
                            i0 = k;
                        } else {
                            d1 = d;
                            i1 = k;
                        }
                    } else {
                        d2 = d;
                        i2 = k;
                    }
                }
            }
        }
    }
    if (active) {
        const size_t o = ((size_t)b * N + q) * 3;
        dist[o + 0] = d0;
        dist[o + 1] = d1;
        dist[o + 2] = d2;
        idx[o + 0] = (IdxT)i0;
        idx[o + 1] = (IdxT)i1;
        idx[o + 2] = (IdxT)i2;
    }
}

// Few queries (the coarse levels of a feature-propagation stack: 4 096 - 8 192 queries against 256 - 512 support points): one query per
// thread leaves a wave alone on its SIMD, paying an LDS round trip and a dependent chain per pair (~285 cycles each).  Here the FOUR
// waves of a workgroup share 64 queries and scan a quarter of the support cloud each -- in ascending index ranges, wave w below wave
// w + 1 --, park their top three in LDS, and wave 0 merges the twelve candidates by the same strict-'<' insertion in wave order: the
// earlier index still wins ties, so the result is the one-thread scan's, bit for bit.  Single tile: S <= kTnnTile.
template <typename IdxT>
__global__ __launch_bounds__(256) void three_nn_split_kernel(int B, int N, int S, const float *__restrict__ xyz1,
                                                              const float *__restrict__ xyz2, float *__restrict__ dist,
                                                              IdxT *__restrict__ idx) {
    __shared__ float4 tile[kTnnTile];
    __shared__ float pd[4][3][kWave];
    __shared__ int pi[4][3][kWave];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int q = blockIdx.x * kWave + lane;
    const bool active = q < N;
    float cx = 0.f, cy = 0.f, cz = 0.f, s1 = 0.f;
    if (active) {
        const float *c = xyz1 + ((size_t)b * N + q) * 3;
        cx = c[0];
        cy = c[1];
        cz = c[2];
        s1 = sumsq3(cx, cy, cz);
    }
    for (int i = threadIdx.x; i < S; i += blockDim.x) {
        const float *p = xyz2 + ((size_t)b * S + i) * 3;
        const float px = p[0], py = p[1], pz = p[2];
        tile[i] = make_float4(px, py, pz, sumsq3(px, py, pz));
    }
    __syncthreads();
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    int i0 = 0, i1 = 0, i2 = 0;
    auto insert = [&](float d, int k) {   // strict '<' in ascending index order keeps the earlier index on ties
        if (d < d2) {
            if (d < d1) {
                d2 = d1;
                i2 = i1;
                if (d < d0) {
                    d1 = d0;
                    i1 = i0;
                    d0 = d;
                    i0 = k;
                } else {
                    d1 = d;
                    i1 = k;
                }
            } else {
                d2 = d;
                i2 = k;
            }
        }
    };
    const int part = (S + 3) / 4, lo = wv * part, hi = min(S, lo + part);
    for (int i = lo; i < hi; ++i) {
        const float4 p = tile[i];
        insert(sqdist_expanded(cx, cy, cz, s1, p.x, p.y, p.z, p.w), i);
    }
    pd[wv][0][lane] = d0, pd[wv][1][lane] = d1, pd[wv][2][lane] = d2;
    pi[wv][0][lane] = i0, pi[wv][1][lane] = i1, pi[wv][2][lane] = i2;
    __syncthreads();
    if (wv == 0 && active) {   // (my own three are already in place: the other nine follow in wave order)
        for (int w = 1; w < 4; ++w)
#pragma unroll
            for (int r = 0; r < 3; ++r) insert(pd[w][r][lane], pi[w][r][lane]);
        const size_t o = ((size_t)b * N + q) * 3;
        dist[o + 0] = d0;
        dist[o + 1] = d1;
        dist[o + 2] = d2;
        idx[o + 0] = (IdxT)i0;
        idx[o + 1] = (IdxT)i1;
        idx[o + 2] = (IdxT)i2;
    }
}

// ---------------------------------------------------------------------------------------------
// square_distance (pointnet2_utils.py:20-41), C = 3.  HBM-bound on the (B,N,M) store: a thread owns
// one dst column for kSqdRows src rows, so every store instruction writes 256 contiguous bytes per wave.
// ---------------------------------------------------------------------------------------------
constexpr int kSqdRows = 16;

__global__ __launch_bounds__(256) void square_distance_kernel(int N, int M, const float *__restrict__ src,
                                                               const float *__restrict__ dst, float *__restrict__ out) {
    __shared__ float4 rows[kSqdRows];
    const int b = blockIdx.z;
    const int i0 = blockIdx.y * kSqdRows;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x < kSqdRows && i0 + threadIdx.x < N) {
        const float *s = src + ((size_t)b * N + i0 + threadIdx.x) * 3;
        rows[threadIdx.x] = make_float4(s[0], s[1], s[2], sumsq3(s[0], s[1], s[2]));
    }
    __syncthreads();
    if (j >= M) return;
    const float *d = dst + ((size_t)b * M + j) * 3;
    const float dx = d[0], dy = d[1], dz = d[2];
    const float s2 = sumsq3(dx, dy, dz);
    const int cnt = min(kSqdRows, N - i0);
    for (int r = 0; r < cnt; ++r) {
        const float4 s = rows[r];
        out[((size_t)b * N + i0 + r) * M + j] = sqdist_expanded(s.x, s.y, s.z, s.w, dx, dy, dz, s2);
    }
}

}  // namespace tgn

using namespace tgn;

TGN_API int tgn_square_distance(int B, int N, int M, const float *src, const float *dst, float *out,
                                tgn_stream_t stream) {
    if (B <= 0 || N <= 0 || M <= 0) return TGN_OK;
    if (!src || !dst || !out) {
        set_error("tgn_square_distance: null pointer");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    if (B > 65535 || (N + kSqdRows - 1) / kSqdRows > 65535) {
        set_error("tgn_square_distance: shape exceeds the launch grid");
        return TGN_ERR_UNSUPPORTED;
    }
    dim3 grid((M + 255) / 256, (N + kSqdRows - 1) / kSqdRows, B);
    hipLaunchKernelGGL(square_distance_kernel, grid, dim3(256), 0, (hipStream_t)stream, N, M, src, dst, out);
    return check_launch("square_distance_kernel");
}

// ---------------------------------------------------------------------------------------------
// kNN, exact heap, one WAVE per query: the 64 lanes evaluate 64 candidates per step and ballot the
// `d2 < root` test (knnquery_cuda_kernel.cu:97); the passing candidates are then pushed, in index order, into
// the reference's heap (verbatim reheap / heap_sort, arrays in LDS) by lane 0.  Identical results to the
// thread-per-query kernel -- the same insertions in the same order -- at a fraction of its latency, which is
// what matters when only a handful of tie queries are redone.
// ---------------------------------------------------------------------------------------------
constexpr int kKnnHeapMax = 128;

__global__ __launch_bounds__(256) void knn_heap_wave_kernel(int b, int m, int nsample, const float *__restrict__ xyz,
                                                             const float *__restrict__ new_xyz,
                                                             const int *__restrict__ offset,
                                                             const int *__restrict__ new_offset,
                                                             int *__restrict__ idx, float *__restrict__ dist2,
                                                             const int *__restrict__ only) {
    __shared__ float hd_s[4][kKnnHeapMax];
    __shared__ int hi_s[4][kKnnHeapMax];
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    float *hd = hd_s[wv];
    int *hi = hi_s[wv];
    const int total = only ? only[0] : m;
    for (int w = blockIdx.x * 4 + wv; w < total; w += gridDim.x * 4) {
        const int pt = only ? only[1 + w] : w;
        int bt = 0;
        while (bt < b - 1 && !(pt < new_offset[bt])) ++bt;
        const int start = bt == 0 ? 0 : offset[bt - 1];
        const int end = offset[bt];
        const float qx = new_xyz[(size_t)pt * 3 + 0], qy = new_xyz[(size_t)pt * 3 + 1], qz = new_xyz[(size_t)pt * 3 + 2];
        for (int i = lane; i < nsample; i += kWave) {
            hd[i] = 1e10f;
            hi[i] = start;
        }
        float root = 1e10f;  // wave-uniform copy of hd[0]
        for (int base = start; base < end; base += kWave) {
            const int i = base + lane;
            float d2 = INFINITY;
            if (i < end) {
                const float ex = qx - xyz[(size_t)i * 3 + 0], ey = qy - xyz[(size_t)i * 3 + 1],
                            ez = qz - xyz[(size_t)i * 3 + 2];
                d2 = dist_direct_nofma(ex, ey, ez);
            }
            unsigned long long mask = __ballot(i < end && d2 < root);
            while (mask) {  // wave-uniform
                const int src = __builtin_ctzll(mask);
                mask &= mask - 1;
                const float dn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d2), src));
                if (!(dn < root)) continue;
                if (lane == 0) {
                    hd[0] = dn;
                    hi[0] = base + src;
                    knn_reheap(hd, hi, 1, nsample);
                }
                root = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hd[0])));
            }
        }
        if (lane == 0) {  // heap_sort, knnquery_cuda_kernel.cu:39-48
            for (int i = nsample - 1; i > 0; --i) {
                const float td = hd[0];
                hd[0] = hd[i];
                hd[i] = td;
                const int ti = hi[0];
                hi[0] = hi[i];
                hi[i] = ti;
                knn_reheap(hd, hi, 1, i);
            }
        }
        for (int i = lane; i < nsample; i += kWave) {
            idx[(size_t)pt * nsample + i] = hi[i];
            dist2[(size_t)pt * nsample + i] = hd[i];
        }
    }
}

static int knn_heap_launch(int b, int m, int nsample, const float *xyz, const float *new_xyz, const int *offset,
                           const int *new_offset, int *idx, float *dist2, const int *only, hipStream_t st) {
    if (nsample <= kKnnHeapMax) {
        // redo pass: the list is usually empty or a few queries; 64 blocks cover up to 256 of them per sweep
        const int blocks = only ? 64 : (m + 3) / 4 < 256 * 16 ? (m + 3) / 4 : 256 * 16;
        hipLaunchKernelGGL(knn_heap_wave_kernel, dim3(blocks), dim3(256), 0, st, b, m, nsample, xyz, new_xyz, offset,
                           new_offset, idx, dist2, only);
        return check_launch("knn_heap_wave_kernel");
    }
    // heap columns in LDS: nsample * 8 B per thread; keep a block under 64 KiB
#define TGN_KNN_HEAP(NT_)                                                                                             \
    hipLaunchKernelGGL((knn_heap_kernel<NT_>), dim3((m + NT_ - 1) / NT_), dim3(NT_), (size_t)nsample * NT_ * 8, st, b, \
                       m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, only)
    if (nsample <= 32)
        TGN_KNN_HEAP(256);
    else if (nsample <= 64)
        TGN_KNN_HEAP(128);
    else if (nsample <= 128)
        TGN_KNN_HEAP(64);
    else {
        set_error("tgn_knnquery: nsample %d > 128 unsupported (the reference's limit is 100)", nsample);
        return TGN_ERR_UNSUPPORTED;
    }
#undef TGN_KNN_HEAP
    return check_launch("knn_heap_kernel");
}

namespace tgn {
__global__ void knn_zero_word_kernel(int *w) { *w = 0; }
// The redo counter is cleared by a one-thread KERNEL, not by hipMemsetAsync: under stream capture the memset becomes a
// memset node of the HIP graph, and replays of graphs holding such a node on a block of torch's private pool faulted on
// ROCm 7.2 whenever an eager allocation happened between two replays (tools/experiments/pt_capture_parts.py).
static int zero_redo(int *redo, hipStream_t st) {
    if (tuning(kTuneKnnMemset)) return hipMemsetAsync(redo, 0, sizeof(int), st) == hipSuccess ? 0 : 1;
    hipLaunchKernelGGL(knn_zero_word_kernel, dim3(1), dim3(1), 0, st, redo);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
}  // namespace tgn

TGN_API size_t tgn_knnquery_workspace_bytes(int m) { return m > 0 ? ((size_t)m + 1) * sizeof(int) : 0; }

TGN_API int tgn_knnquery_ws(int b, int m, int nsample, const float *xyz, const float *new_xyz, const int *offset,
                            const int *new_offset, int *idx, float *dist2, void *workspace, size_t workspace_bytes,
                            tgn_stream_t stream) {
    if (m <= 0 || nsample <= 0) return TGN_OK;
    if (b <= 0 || !xyz || !new_xyz || !offset || !new_offset || !idx || !dist2) {
        set_error("tgn_knnquery: bad argument");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool wave_path = nsample <= kWave - 1 && workspace && workspace_bytes >= tgn_knnquery_workspace_bytes(m);
    if (!wave_path) return knn_heap_launch(b, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, nullptr, st);
    int *redo = (int *)workspace;
    if (zero_redo(redo, st)) {
        set_error("tgn_knnquery: clearing the redo counter failed");
        return TGN_ERR_LAUNCH;
    }
    const int qpb = 4 * kKnnQ;  // queries per 256-thread block
    hipLaunchKernelGGL(knn_wave_kernel, dim3((m + qpb - 1) / qpb), dim3(256), 0, st, b, m, nsample, xyz, new_xyz, offset,
                       new_offset, idx, dist2, redo);
    if (int rc = check_launch("knn_wave_kernel")) return rc;
    // queries with bit-exact distance ties: exact heap order (usually none; the grid is sized for all of them)
    return knn_heap_launch(b, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, redo, st);
}

TGN_API size_t tgn_knnquery_grid_workspace_bytes(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    return knn_grid_off_records(b, m) + (size_t)n * sizeof(float4);
}

TGN_API int tgn_knnquery_grid(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, const int *offset,
                              const int *new_offset, int *idx, float *dist2, void *workspace, size_t workspace_bytes,
                              tgn_stream_t stream) {
    if (m <= 0 || nsample <= 0) return TGN_OK;
    if (b <= 0 || n < 0 || !xyz || !new_xyz || !offset || !new_offset || !idx || !dist2) {
        set_error("tgn_knnquery_grid: bad argument");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    // the grids only pay for big segments and need the (k+1)-list in one wave; otherwise the plain kernels
    if (nsample > kWave - 1 || !workspace || workspace_bytes < tgn_knnquery_grid_workspace_bytes(b, n, m) || b > 4096)
        return tgn_knnquery_ws(b, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, workspace, workspace_bytes, stream);
    hipStream_t st = (hipStream_t)stream;
    int *redo = (int *)workspace;
    if (zero_redo(redo, st)) {
        set_error("tgn_knnquery_grid: clearing the redo counter failed");
        return TGN_ERR_LAUNCH;
    }
    const float scale = 0.001f * (float)tuning(kTuneKnnGridScale);  // cell size relative to the estimated k-neighbour radius
    hipLaunchKernelGGL(knn_grid_build_kernel, dim3(b), dim3(kKnnBuildThreads), 0, st, b, m, nsample, scale, xyz, offset,
                       (unsigned char *)workspace);
    if (int rc = check_launch("knn_grid_build_kernel")) return rc;
    hipLaunchKernelGGL(knn_grid_query_kernel, dim3((m + 3) / 4), dim3(256), 0, st, b, m, nsample, xyz, new_xyz, offset,
                       new_offset, (const unsigned char *)workspace, idx, dist2, redo);
    if (int rc = check_launch("knn_grid_query_kernel")) return rc;
    return knn_heap_launch(b, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, redo, st);
}

TGN_API int tgn_knnquery(int b, int m, int nsample, const float *xyz, const float *new_xyz, const int *offset,
                         const int *new_offset, int *idx, float *dist2, tgn_stream_t stream) {
    return tgn_knnquery_ws(b, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, nullptr, 0, stream);
}

// Reference ABI (knnquery_cuda_kernel.h:13) has no segment count: like the reference's get_bt_idx the
// search simply walks new_offset until it finds the query's segment, so pass "unbounded".
TGN_API void knnquery_cuda_launcher(int m, int nsample, const float *xyz, const float *new_xyz, const int *offset,
                                    const int *new_offset, int *idx, float *dist2) {
    (void)tgn_knnquery(1 << 30, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2,
                       (tgn_stream_t)default_stream());
}

TGN_API int tgn_three_nn(int B, int N, int S, const float *xyz1, const float *xyz2, float *dist, void *idx,
                         int idx_is_int64, tgn_stream_t stream) {
    if (B <= 0 || N <= 0) return TGN_OK;
    if (S < 0 || !xyz1 || !xyz2 || !dist || !idx) {
        set_error("tgn_three_nn: bad argument");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    if (S <= kTnnTile && S >= 16 && (long long)B * N <= 32768) {   // few queries: four waves per 64 queries (three_nn_split_kernel)
        dim3 grid4((N + kWave - 1) / kWave, B);
        if (idx_is_int64)
            hipLaunchKernelGGL((three_nn_split_kernel<long long>), grid4, dim3(256), 0, (hipStream_t)stream, B, N, S, xyz1, xyz2, dist,
                               (long long *)idx);
        else
            hipLaunchKernelGGL((three_nn_split_kernel<int>), grid4, dim3(256), 0, (hipStream_t)stream, B, N, S, xyz1, xyz2, dist,
                               (int *)idx);
        return check_launch("three_nn_split_kernel");
    }
    dim3 grid((N + 255) / 256, B);
    if (idx_is_int64)
        hipLaunchKernelGGL((three_nn_kernel<long long>), grid, dim3(256), 0, (hipStream_t)stream, B, N, S, xyz1, xyz2,
                           dist, (long long *)idx);
    else
        hipLaunchKernelGGL((three_nn_kernel<int>), grid, dim3(256), 0, (hipStream_t)stream, B, N, S, xyz1, xyz2, dist,
                           (int *)idx);
    return check_launch("three_nn_kernel");
}

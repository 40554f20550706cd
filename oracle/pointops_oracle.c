/*
 * oracle/pointops_oracle.c -- CPU restatement of the reference's point-cloud
 * sampling / grouping hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The shipped path is the HIP
 * library under toothgroupnetwork_amd/csrc and fails loudly when it is missing.
 *
 * Every function cites the reference lines it restates (paths relative to the
 * reference checkout, limhoyeon/ToothGroupNetwork):
 *   P  = external_libs/pointops/src
 *   U  = external_libs/pointnet2_utils/pointnet2_utils.py
 *   PY = external_libs/pointops/functions/pointops.py
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - ball query / three_nn / square_distance / index_points / FPS (torch-CPU
 *     semantics, start index 0): PINNED against outputs of the reference's own
 *     torch functions executed on CPU in the build container
 *     (tests/golden/make_golden.py -> tests/golden/ npz files).
 *   - kNN, grouping, interpolation, subtraction, aggregation and the tree tie order
 *     of FPS restate CUDA-only kernels: PINNED on the GPU box against the reference's
 *     own kernels, compiled for gfx950 where they lie (oracle/Makefile `ref` ->
 *     oracle/_ref/libpointops_ref.so; tests/test_gpu_parity.py::*_vs_reference_kernel*),
 *     and against independent brute-force numpy definitions on the CPU.
 *   - PARITY UNPINNED: only nvcc's own FMA contraction of the FPS / kNN distance
 *     (the TGN_FPS_FMA flag of the "cuda-compat" mode) -- there is no nvcc and no
 *     CUDA device here, and hipcc contracts the same source line differently.
 *
 * Arithmetic contract (build with -ffp-contract=off; fused ops are written as
 * fmaf() explicitly so the compiler never chooses):
 *   FPS canonical  : d = ((dx*dx) + (dy*dy)) + (dz*dz), first index wins ties
 *                    (torch-CPU form of U:103-118, start forced to 0 as P/sampling/
 *                    sampling_cuda_kernel.cu:39 does).
 *   FPS cuda-compat: d = fma(dz,dz, fma(dy,dy, dx*dx)); ties resolved by the
 *                    shared-memory tree of sampling_cuda_kernel.cu:5-10,64-123.
 *   square_distance: dot = fma(z1,z2, fma(y1,y2, x1*x2));
 *                    d = ((-2*dot) + ((x1*x1+y1*y1)+z1*z1)) + ((x2*x2+y2*y2)+z2*z2)
 *                    (bit-identical to torch-CPU U:20-41 on the build host).
 *   kNN            : d2 = ((qx-x)*(qx-x) + (qy-y)*(qy-y)) + (qz-z)*(qz-z), no fusion
 *                    (knnquery_cuda_kernel.cu:96).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* P/cuda_utils.h:11-14 -- largest power of two <= work_size, clamped to [1,1024]. */
ORACLE_API int oracle_opt_n_threads(int work_size) {
    if (work_size < 1) return 1;
    int pow_2 = (int)(log((double)work_size) / log(2.0));
    int t = 1 << pow_2;
    if (t > 1024) t = 1024;
    if (t < 1) t = 1;
    return t;
}

static inline float fps_dist_canonical(float x1, float y1, float z1, float x2, float y2, float z2) {
    float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
    return ((dx * dx) + (dy * dy)) + (dz * dz);
}

static inline float fps_dist_fma(float x1, float y1, float z1, float x2, float y2, float z2) {
    float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

/* IEEE minNum, what CUDA min(float,float) and torch's masked update both give
 * for the non-NaN case; a NaN distance never replaces tmp. */
static inline float min_nn(float d, float t) { return (d < t) ? d : t; }

/*
 * Canonical FPS on one packed segment [start_n, end_n) -> idx[start_m, end_m).
 * Restates sampling_cuda_kernel.cu:39-59,125-127 (first sample = first point of the
 * segment, running minimum in tmp, strict '>' argmax) with the torch-CPU arithmetic
 * and first-index tie-break of U:103-118.
 */
static void fps_segment_canonical(const float *xyz, int start_n, int end_n, int start_m, int end_m,
                                  float *tmp, int *idx, int use_fma) {
    if (end_m <= start_m) return;
    idx[start_m] = start_n;
    int old = start_n;
    for (int j = start_m + 1; j < end_m; j++) {
        float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
        int besti = start_n;
        float best = -1.0f;
        for (int k = start_n; k < end_n; k++) {
            float d = use_fma ? fps_dist_fma(x1, y1, z1, xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2])
                              : fps_dist_canonical(x1, y1, z1, xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2]);
            float d2 = min_nn(d, tmp[k]);
            tmp[k] = d2;
            if (d2 > best) { best = d2; besti = k; }
        }
        old = besti;
        idx[j] = old;
    }
}

/*
 * "cuda-compat" FPS: simulates the reference block exactly -- block_size threads stride
 * over the segment (sampling_cuda_kernel.cu:49-59), then the shared-memory tree
 * (:64-123) where __update (:5-10) keeps the LOWER slot on ties.
 */
static void fps_segment_cuda_compat(const float *xyz, int start_n, int end_n, int start_m, int end_m,
                                    float *tmp, int *idx, int block_size, int old0, int use_fma) {
    if (end_m <= start_m) return;
    float *dists = (float *)malloc(sizeof(float) * (size_t)block_size);
    int *dists_i = (int *)malloc(sizeof(int) * (size_t)block_size);
    idx[start_m] = start_n;
    int old = old0; /* sampling_cuda_kernel.cu:24,32: bid==0 -> 0, else offset[bid-1] == start_n */
    for (int j = start_m + 1; j < end_m; j++) {
        float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
        for (int tid = 0; tid < block_size; tid++) {
            int besti = start_n;
            float best = -1.0f;
            for (int k = start_n + tid; k < end_n; k += block_size) {
                float d = use_fma ? fps_dist_fma(x1, y1, z1, xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2])
                                  : fps_dist_canonical(x1, y1, z1, xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2]);
                float d2 = min_nn(d, tmp[k]);
                tmp[k] = d2;
                besti = d2 > best ? k : besti;
                best = d2 > best ? d2 : best;
            }
            dists[tid] = best;
            dists_i[tid] = besti;
        }
        for (int half = block_size / 2; half >= 1; half /= 2) {
            for (int tid = 0; tid < half; tid++) {
                float v1 = dists[tid], v2 = dists[tid + half];
                int i1 = dists_i[tid], i2 = dists_i[tid + half];
                dists[tid] = v1 > v2 ? v1 : v2; /* max(v1, v2) */
                dists_i[tid] = v2 > v1 ? i2 : i1;
            }
        }
        old = dists_i[0];
        idx[j] = old;
    }
    free(dists);
    free(dists_i);
}

/*
 * pointops.furthestsampling (PY:10-27 -> sampling_cuda.cpp:8-16 -> sampling_cuda_kernel.cu:14-171).
 * xyz (n,3) packed, offset/new_offset (b) cumulative ends, idx (new_offset[b-1]) out.
 * mode bit 0: arithmetic -- 0 = unfused ((dx*dx)+(dy*dy))+(dz*dz), 1 = FMA chain (nvcc-style contraction);
 * mode bit 1: tie order -- 0 = first index, 2 = the reference's shared-memory tree.
 *   mode 0 = canonical (torch-CPU semantics); mode 3 = "cuda-compat"; mode 2 = the reference source
 *   compiled without contraction (what oracle/_ref is, see oracle/Makefile).
 * block_size: only with the tree; 0 -> opt_n_threads(n_max) like the reference launcher.
 * Clouds are independent; with OpenMP they are spread over the host cores.
 */
ORACLE_API int oracle_furthestsampling(int b, const float *xyz, const int *offset, const int *new_offset,
                                       int *idx, int mode, int block_size) {
    if (b <= 0) return 0;
    int n_total = offset[b - 1];
    int n_max = offset[0];
    for (int i = 1; i < b; i++) {
        int c = offset[i] - offset[i - 1];
        if (c > n_max) n_max = c;
    }
    if ((mode & 2) && block_size <= 0) block_size = oracle_opt_n_threads(n_max);
    float *tmp = (float *)malloc(sizeof(float) * (size_t)(n_total > 0 ? n_total : 1));
    for (int i = 0; i < n_total; i++) tmp[i] = 1e10f; /* PY:22 */
#pragma omp parallel for schedule(dynamic, 1)
    for (int bid = 0; bid < b; bid++) {
        int start_n = bid == 0 ? 0 : offset[bid - 1];
        int end_n = offset[bid];
        int start_m = bid == 0 ? 0 : new_offset[bid - 1];
        int end_m = new_offset[bid];
        if (!(mode & 2))
            fps_segment_canonical(xyz, start_n, end_n, start_m, end_m, tmp, idx, mode & 1);
        else
            fps_segment_cuda_compat(xyz, start_n, end_n, start_m, end_m, tmp, idx, block_size,
                                    bid == 0 ? 0 : offset[bid - 1], mode & 1);
    }
    free(tmp);
    return 0;
}

/* ---- kNN: knnquery_cuda_kernel.cu:21-48 (reheap, heap_sort), :51-62, :65-108 ---- */
static void knn_reheap(float *dist, int *idx, int k) {
    int root = 0;
    int child = root * 2 + 1;
    while (child < k) {
        if (child + 1 < k && dist[child + 1] > dist[child]) child++;
        if (dist[root] > dist[child]) return;
        float td = dist[root]; dist[root] = dist[child]; dist[child] = td;
        int ti = idx[root]; idx[root] = idx[child]; idx[child] = ti;
        root = child;
        child = root * 2 + 1;
    }
}

static void knn_heap_sort(float *dist, int *idx, int k) {
    for (int i = k - 1; i > 0; i--) {
        float td = dist[0]; dist[0] = dist[i]; dist[i] = td;
        int ti = idx[0]; idx[0] = idx[i]; idx[i] = ti;
        knn_reheap(dist, idx, i);
    }
}

/*
 * pointops.knnquery native half (PY:30-45 minus the sqrt).  b = number of segments.
 * idx (m,nsample) int32, dist2 (m,nsample) fp32 squared distances.
 * nsample <= 100 in the reference (local arrays of 100, knnquery_cuda_kernel.cu:86-87);
 * the oracle accepts any nsample.
 */
ORACLE_API int oracle_knnquery(int b, int m, int nsample, const float *xyz, const float *new_xyz,
                               const int *offset, const int *new_offset, int *idx, float *dist2) {
    if (nsample <= 0) return 0;
#pragma omp parallel
    {
        float *best_dist = (float *)malloc(sizeof(float) * (size_t)nsample);
        int *best_idx = (int *)malloc(sizeof(int) * (size_t)nsample);
#pragma omp for schedule(static)
        for (int pt = 0; pt < m; pt++) {
            int bt = 0; /* get_bt_idx :51-62 */
            while (bt < b - 1 && !(pt < new_offset[bt])) bt++;
            int start = bt == 0 ? 0 : offset[bt - 1];
            int end = offset[bt];
            float qx = new_xyz[pt * 3 + 0], qy = new_xyz[pt * 3 + 1], qz = new_xyz[pt * 3 + 2];
            for (int i = 0; i < nsample; i++) { best_dist[i] = 1e10f; best_idx[i] = start; }
            for (int i = start; i < end; i++) {
                float x = xyz[i * 3 + 0], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
                float d2 = ((qx - x) * (qx - x) + (qy - y) * (qy - y)) + (qz - z) * (qz - z);
                if (d2 < best_dist[0]) {
                    best_dist[0] = d2;
                    best_idx[0] = i;
                    knn_reheap(best_dist, best_idx, nsample);
                }
            }
            knn_heap_sort(best_dist, best_idx, nsample);
            for (int i = 0; i < nsample; i++) {
                idx[(size_t)pt * nsample + i] = best_idx[i];
                dist2[(size_t)pt * nsample + i] = best_dist[i];
            }
        }
        free(best_dist);
        free(best_idx);
    }
    return 0;
}

/* ---- grouping: grouping_cuda_kernel.cu:5-25 ---- */
ORACLE_API int oracle_grouping_forward(int m, int nsample, int c, const float *input, const int *idx,
                                       float *output) {
    for (int64_t r = 0; r < (int64_t)m * nsample; r++)
        for (int ci = 0; ci < c; ci++) output[r * c + ci] = input[(int64_t)idx[r] * c + ci];
    return 0;
}

ORACLE_API int oracle_grouping_backward(int m, int nsample, int c, const float *grad_output, const int *idx,
                                        float *grad_input /* (n,c) pre-zeroed */) {
    for (int64_t r = 0; r < (int64_t)m * nsample; r++)
        for (int ci = 0; ci < c; ci++) grad_input[(int64_t)idx[r] * c + ci] += grad_output[r * c + ci];
    return 0;
}

/* ---- interpolation: interpolation_cuda_kernel.cu:5-33 (output pre-zeroed, sequential i) ---- */
ORACLE_API int oracle_interpolation_forward(int n, int c, int k, const float *input, const int *idx,
                                            const float *weight, float *output) {
    for (int64_t ni = 0; ni < n; ni++)
        for (int ci = 0; ci < c; ci++) {
            float acc = output[ni * c + ci];
            for (int i = 0; i < k; i++)
                acc += input[(int64_t)idx[ni * k + i] * c + ci] * weight[ni * k + i];
            output[ni * c + ci] = acc;
        }
    return 0;
}

ORACLE_API int oracle_interpolation_backward(int n, int c, int k, const float *grad_output, const int *idx,
                                             const float *weight, float *grad_input /* (m,c) pre-zeroed */) {
    for (int64_t ni = 0; ni < n; ni++)
        for (int ci = 0; ci < c; ci++)
            for (int i = 0; i < k; i++)
                grad_input[(int64_t)idx[ni * k + i] * c + ci] += grad_output[ni * c + ci] * weight[ni * k + i];
    return 0;
}

/* ---- subtraction: subtraction_cuda_kernel.cu:5-30 ---- */
ORACLE_API int oracle_subtraction_forward(int n, int nsample, int c, const float *input1, const float *input2,
                                          const int *idx, float *output) {
    for (int64_t ni = 0; ni < n; ni++)
        for (int j = 0; j < nsample; j++)
            for (int ci = 0; ci < c; ci++)
                output[(ni * nsample + j) * c + ci] =
                    input1[ni * c + ci] - input2[(int64_t)idx[ni * nsample + j] * c + ci];
    return 0;
}

ORACLE_API int oracle_subtraction_backward(int n, int nsample, int c, const int *idx, const float *grad_output,
                                           float *grad_input1, float *grad_input2 /* both pre-zeroed */) {
    for (int64_t ni = 0; ni < n; ni++)
        for (int j = 0; j < nsample; j++)
            for (int ci = 0; ci < c; ci++) {
                float g = grad_output[(ni * nsample + j) * c + ci];
                grad_input1[ni * c + ci] += g;
                grad_input2[(int64_t)idx[ni * nsample + j] * c + ci] += -g;
            }
    return 0;
}

/* ---- aggregation: aggregation_cuda_kernel.cu:5-39 ---- */
ORACLE_API int oracle_aggregation_forward(int n, int nsample, int c, int w_c, const float *input,
                                          const float *position, const float *weight, const int *idx,
                                          float *output /* pre-zeroed */) {
    for (int64_t ni = 0; ni < n; ni++)
        for (int ci = 0; ci < c; ci++) {
            int wci = ci % w_c;
            float acc = output[ni * c + ci];
            for (int j = 0; j < nsample; j++) {
                int64_t ii = ni * nsample + j;
                acc += (input[(int64_t)idx[ii] * c + ci] + position[ii * c + ci]) * weight[ii * w_c + wci];
            }
            output[ni * c + ci] = acc;
        }
    return 0;
}

ORACLE_API int oracle_aggregation_backward(int n, int nsample, int c, int w_c, const float *input,
                                           const float *position, const float *weight, const int *idx,
                                           const float *grad_output, float *grad_input, float *grad_position,
                                           float *grad_weight /* grad_input, grad_weight pre-zeroed */) {
    for (int64_t ni = 0; ni < n; ni++)
        for (int ci = 0; ci < c; ci++) {
            int wci = ci % w_c;
            float go = grad_output[ni * c + ci];
            for (int j = 0; j < nsample; j++) {
                int64_t ii = ni * nsample + j;
                float w = weight[ii * w_c + wci];
                grad_input[(int64_t)idx[ii] * c + ci] += go * w;
                grad_position[ii * c + ci] = go * w;
                grad_weight[ii * w_c + wci] += go * (input[(int64_t)idx[ii] * c + ci] + position[ii * c + ci]);
            }
        }
    return 0;
}

/* ---- square_distance (U:20-41), one pair, torch-CPU bit pattern ---- */
static inline float sqdist_expanded(float x1, float y1, float z1, float s1, float x2, float y2, float z2,
                                    float s2) {
    float dot = fmaf(z1, z2, fmaf(y1, y2, x1 * x2));
    return ((-2.0f * dot) + s1) + s2;
}

static inline float sumsq3(float x, float y, float z) { return ((x * x) + (y * y)) + (z * z); }

ORACLE_API int oracle_square_distance(int B, int N, int M, const float *src /* (B,N,3) */,
                                      const float *dst /* (B,M,3) */, float *out /* (B,N,M) */) {
    for (int b = 0; b < B; b++)
        for (int i = 0; i < N; i++) {
            const float *s = src + ((int64_t)b * N + i) * 3;
            float s1 = sumsq3(s[0], s[1], s[2]);
            for (int j = 0; j < M; j++) {
                const float *d = dst + ((int64_t)b * M + j) * 3;
                out[((int64_t)b * N + i) * M + j] =
                    sqdist_expanded(s[0], s[1], s[2], s1, d[0], d[1], d[2], sumsq3(d[0], d[1], d[2]));
            }
        }
    return 0;
}

/*
 * query_ball_point (U:120-144): the first nsample indices, in ascending index order,
 * with sqrdist <= r2 (the reference masks `sqrdists > radius**2`, so NaN distances
 * count as inside); short rows are padded with the row's first hit; a row with no hit
 * at all is filled with N (what the sort of an all-N row yields, U:136-141).
 * r2 is passed as the float the comparison is carried out in (see host wrapper).
 */
ORACLE_API int oracle_ball_query(int B, int N, int S, int nsample, float r2, const float *xyz /* (B,N,3) */,
                                 const float *new_xyz /* (B,S,3) */, int64_t *group_idx /* (B,S,nsample) */) {
    float *pn = (float *)malloc(sizeof(float) * (size_t)(B > 0 && N > 0 ? (size_t)B * N : 1));
    for (int64_t i = 0; i < (int64_t)B * N; i++) pn[i] = sumsq3(xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]);
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < (int64_t)B * S; q++) {
        int b = (int)(q / S);
        const float *c = new_xyz + q * 3;
        float s1 = sumsq3(c[0], c[1], c[2]);
        int64_t *row = group_idx + q * nsample;
        int cnt = 0;
        for (int k = 0; k < N && cnt < nsample; k++) {
            const float *p = xyz + ((int64_t)b * N + k) * 3;
            float d = sqdist_expanded(c[0], c[1], c[2], s1, p[0], p[1], p[2], pn[(int64_t)b * N + k]);
            if (!(d > r2)) row[cnt++] = k;
        }
        int64_t first = cnt > 0 ? row[0] : (int64_t)N;
        for (int j = cnt; j < nsample; j++) row[j] = first;
    }
    free(pn);
    return 0;
}

/*
 * three_nn of PointNetFeaturePropagation (U:333-335): the 3 smallest entries of
 * square_distance(xyz1, xyz2) per row of xyz1, ascending by (distance, index)
 * (torch.sort on ties is unspecified; (d, idx) is the canonical order used here).
 * Needs S >= 3 like the reference slice; for S < 3 the tail is (inf, 0).
 */
ORACLE_API int oracle_three_nn(int B, int N, int S, const float *xyz1 /* (B,N,3) queries */,
                               const float *xyz2 /* (B,S,3) support */, float *dist /* (B,N,3) */,
                               int64_t *idx /* (B,N,3) */) {
    float *sn = (float *)malloc(sizeof(float) * (size_t)(B > 0 && S > 0 ? (size_t)B * S : 1));
    for (int64_t i = 0; i < (int64_t)B * S; i++) sn[i] = sumsq3(xyz2[i * 3], xyz2[i * 3 + 1], xyz2[i * 3 + 2]);
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < (int64_t)B * N; q++) {
        int b = (int)(q / N);
        const float *c = xyz1 + q * 3;
        float s1 = sumsq3(c[0], c[1], c[2]);
        float bd[3] = {INFINITY, INFINITY, INFINITY};
        int64_t bi[3] = {0, 0, 0};
        for (int k = 0; k < S; k++) {
            const float *p = xyz2 + ((int64_t)b * S + k) * 3;
            float d = sqdist_expanded(c[0], c[1], c[2], s1, p[0], p[1], p[2], sn[(int64_t)b * S + k]);
            /* strict '<' while scanning ascending k keeps the earlier index on ties */
            if (d < bd[2]) {
                if (d < bd[1]) {
                    bd[2] = bd[1]; bi[2] = bi[1];
                    if (d < bd[0]) { bd[1] = bd[0]; bi[1] = bi[0]; bd[0] = d; bi[0] = k; }
                    else { bd[1] = d; bi[1] = k; }
                } else { bd[2] = d; bi[2] = k; }
            }
        }
        for (int j = 0; j < 3; j++) { dist[q * 3 + j] = bd[j]; idx[q * 3 + j] = bi[j]; }
    }
    free(sn);
    return 0;
}

/*
 * three_interpolate of PointNetFeaturePropagation (U:337-340):
 * w = 1/(d + 1e-8), normalised by ((w0+w1)+w2); out = ((f0*w0 + f1*w1) + f2*w2).
 */
ORACLE_API int oracle_three_interpolate(int B, int N, int S, int C, const float *points2 /* (B,S,C) */,
                                        const float *dist /* (B,N,3) */, const int64_t *idx /* (B,N,3) */,
                                        float *out /* (B,N,C) */) {
    for (int64_t q = 0; q < (int64_t)B * N; q++) {
        int b = (int)(q / N);
        float r0 = 1.0f / (dist[q * 3 + 0] + 1e-8f);
        float r1 = 1.0f / (dist[q * 3 + 1] + 1e-8f);
        float r2 = 1.0f / (dist[q * 3 + 2] + 1e-8f);
        float norm = (r0 + r1) + r2;
        float w0 = r0 / norm, w1 = r1 / norm, w2 = r2 / norm;
        const float *f0 = points2 + ((int64_t)b * S + idx[q * 3 + 0]) * C;
        const float *f1 = points2 + ((int64_t)b * S + idx[q * 3 + 1]) * C;
        const float *f2 = points2 + ((int64_t)b * S + idx[q * 3 + 2]) * C;
        for (int ci = 0; ci < C; ci++) out[q * C + ci] = ((f0[ci] * w0) + (f1[ci] * w1)) + (f2[ci] * w2);
    }
    return 0;
}

/*
 * The grouping step of sample_and_group (U:162-169) / PointNetSetAbstractionMsg
 * (U:281-285): out[b,s,k,:] = concat of (xyz[b,idx]-new_xyz[b,s]) and points[b,idx].
 * xyz_first != 0 -> [rel_xyz, feat] (sample_and_group); 0 -> [feat, rel_xyz] (Msg).
 * An index equal to N (empty ball) is out of range in the reference (it would raise);
 * the oracle rejects it with a non-zero return.
 */
ORACLE_API int oracle_group_points(int B, int N, int S, int K, int D, const float *xyz, const float *new_xyz,
                                   const float *points /* (B,N,D) or NULL */, const int64_t *idx,
                                   int xyz_first, float *out /* (B,S,K,3+D) */) {
    int C = 3 + (points ? D : 0);
    for (int64_t r = 0; r < (int64_t)B * S * K; r++) {
        int64_t q = r / K;
        int b = (int)(q / S);
        int64_t k = idx[r];
        if (k < 0 || k >= N) return 1;
        const float *p = xyz + ((int64_t)b * N + k) * 3;
        const float *c = new_xyz + q * 3;
        float *o = out + r * C;
        int xo = xyz_first ? 0 : (points ? D : 0);
        int fo = xyz_first ? 3 : 0;
        o[xo + 0] = p[0] - c[0];
        o[xo + 1] = p[1] - c[1];
        o[xo + 2] = p[2] - c[2];
        if (points)
            for (int ci = 0; ci < D; ci++) o[fo + ci] = points[((int64_t)b * N + k) * D + ci];
    }
    return 0;
}

ORACLE_API int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORACLE_API void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

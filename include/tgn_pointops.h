/*
 * tgn_pointops.h -- C ABI of libtgn_pointops.so, the MI355X (gfx950) implementation of
 * ToothGroupNetwork's point-cloud sampling / grouping hot path.
 *
 * Plain pointers and sizes only (device pointers unless stated), no torch types.  The host
 * side (toothgroupnetwork_amd/_lib.py) binds these with ctypes and passes tensor.data_ptr()
 * and the current HIP stream.
 *
 * Section 1 re-exports, name for name and argument for argument, the `extern "C"` launchers the
 * reference's pybind layer binds (limhoyeon/ToothGroupNetwork, external_libs/pointops/src):
 * a maintainer can link the reference's *_cuda.cpp wrappers against this library unchanged.
 * They run on the stream set with tgn_set_default_stream() (the reference uses the null
 * stream, e.g. sampling_cuda_kernel.cu:136) and report errors through tgn_last_error().
 *
 * Section 2 holds the stream-aware forms of the same operators (trailing stream, int status)
 * and Section 3 the operators the reference composes out of torch kernels
 * (external_libs/pointnet2_utils/pointnet2_utils.py) that are single HIP kernels here.
 *
 * All functions are thread-safe as long as callers own their buffers; scratch is caller-provided.
 * Status: 0 = ok, non-zero = error (tgn_last_error() gives the text).  Layouts are row-major
 * contiguous fp32 / int32 exactly as the reference's kernels expect.
 */
#ifndef TGN_POINTOPS_H
#define TGN_POINTOPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *tgn_stream_t; /* a hipStream_t; NULL = the null stream */

#define TGN_OK 0
#define TGN_ERR_INVALID_ARGUMENT 1
#define TGN_ERR_LAUNCH 2
#define TGN_ERR_UNSUPPORTED 3

/* flags of tgn_furthestsampling* */
#define TGN_FPS_FMA 1         /* d = fma(dz,dz,fma(dy,dy,dx*dx)): nvcc's contraction of sampling_cuda_kernel.cu:54 */
#define TGN_FPS_LOCAL_INDEX 2 /* write indices relative to the cloud's first point (pointnet2_utils.py:96) */
#define TGN_FPS_INDEX64 4     /* idx is int64_t* instead of int32_t* */
#define TGN_FPS_TREE_TIES 8   /* equal distances resolved like the CUDA kernel's shared-memory tree (:5-10,64-123) */
#define TGN_FPS_CUDA_COMPAT (TGN_FPS_FMA | TGN_FPS_TREE_TIES)
#define TGN_FPS_LOW_VALU 16   /* scheduling hint, same results: clouds of 2048 - 4096 points also take the bucket-skipping kernel.
                                 Alone it is ~15 % slower there than the plain register-resident kernel, but it issues a tenth of
                                 the vector instructions -- the right choice when the launch runs beside ALU-bound work */

#define TGN_FPS_THROUGHPUT 32 /* scheduling hint, same results: clouds of 4 097 - 32 768 points run the owner-wave kernel out of a
                                 cell-sorted, L2-resident workspace (tgn_fps_throughput_workspace_bytes; the *_ws / *_prefix entry
                                 points) instead of out of registers: 58 VGPRs per lane, so FOUR workgroups share a CU where the
                                 register-resident kernel allows one.  A single cloud takes 1.5x as long; a batch of several
                                 clouds per CU finishes sooner (profiles/r06_fps_throughput.txt).  Ignored without a workspace. */

const char *tgn_version(void);
const char *tgn_last_error(void);     /* per host thread */
/* Stream of the section-1 entry points (which have no stream argument); per host thread, default NULL. */
void tgn_set_default_stream(tgn_stream_t stream);
/*
 * FPS arithmetic of furthestsampling_cuda_launcher (which has no flags argument): any of TGN_FPS_TREE_TIES |
 * TGN_FPS_FMA.  Process-wide; initialised from the environment (TGN_FPS_TIES=first|tree, TGN_FPS_FMA=0|1), default 0 =
 * first-index ties, unfused distance.  TGN_FPS_TREE_TIES reproduces the tie order of the reference kernel's
 * shared-memory tree (sampling_cuda_kernel.cu:5-10,64-123) and is pinned against that kernel (oracle/_ref).
 */
void tgn_set_fps_mode(int flags);
int tgn_get_fps_mode(void);
/*
 * Kernel-variant switches for experiments, A/B runs and the parity tests that must reach every variant.  One table of
 * atomics inside the library, read with a relaxed load on the launch paths (no getenv there); the legacy TGN_* environment
 * names seed it once, when the library is loaded.  Thread-safe; a change applies to launches enqueued after the call.
 *   "fps_plain"          1 = plain register-resident / streaming FPS kernels, no bucket skipping        (TGN_FPS_V1)
 *   "fps_config"         NT * 256 + P forces an instantiated plain-kernel shape, 0 = pick               (TGN_FPS_CONFIG=NT,P)
 *   "fps_bucket_config"  NT * 256 + P forces a bucket-kernel shape, 0 = pick                            (TGN_FPS_BUCKET_CONFIG=NT,P)
 *   "fps_cell_bits"      4 (default) or 5: bits per axis of the bucket kernel's Z-order cell code       (TGN_FPS_CELL_BITS)
 *   "fps_bucket_min"     smallest cloud the bucket kernel takes, -1 = built-in thresholds               (TGN_FPS_BUCKET_MIN)
 *   "ball_bitmap"        0 = rank-select ball-query kernel instead of the bitmap one                    (TGN_BALL_BITMAP)
 *   "sa_tile"            0 = pick, 128 / 256 = force the workgroup tile of tgn_sa_mlp2_max_bf16x3                (TGN_SA_TILE)
 *   "gather_v4"          gather family: bit 0 forward kernels with 16-byte lanes, bit 1 backward kernels with 16-byte lanes,
 *                        bit 2 subtraction / aggregation backward with owner-side sums on dword lanes (default 5)    (TGN_GATHER_V4)
 *   "knn_memset"         1 = clear the kNN redo counter with hipMemsetAsync (reproduces a graph fault)  (TGN_KNN_MEMSET)
 *   "knn_grid_scale"     kNN grid cell, per mille of the estimated k-neighbour radius (1000)            (TGN_KNN_GRID_SCALE)
 * tgn_set_tuning returns TGN_ERR_INVALID_ARGUMENT for an unknown key; tgn_get_tuning returns `fallback` for one.
 */
int tgn_set_tuning(const char *key, int value);
int tgn_get_tuning(const char *key, int fallback);

/* ------------------------------------------------------------------------------------------
 * 1. The reference's C ABI, verbatim (each line cites the declaration it replaces).
 * ---------------------------------------------------------------------------------------- */
/* sampling/sampling_cuda_kernel.h:13 ; n = max points per cloud ; tmp (n_total) pre-filled 1e10 */
void furthestsampling_cuda_launcher(int b, int n, const float *xyz, const int *offset, const int *new_offset,
                                    float *tmp, int *idx);
/* knnquery/knnquery_cuda_kernel.h:13 ; dist2 = squared distances, ascending */
void knnquery_cuda_launcher(int m, int nsample, const float *xyz, const float *new_xyz, const int *offset,
                            const int *new_offset, int *idx, float *dist2);
/* grouping/grouping_cuda_kernel.h:14-15 */
void grouping_forward_cuda_launcher(int m, int nsample, int c, const float *input, const int *idx, float *output);
void grouping_backward_cuda_launcher(int m, int nsample, int c, const float *grad_output, const int *idx,
                                     float *grad_input);
/* interpolation/interpolation_cuda_kernel.h:14-15 ; output / grad_input pre-zeroed, accumulated into */
void interpolation_forward_cuda_launcher(int n, int c, int k, const float *input, const int *idx,
                                         const float *weight, float *output);
void interpolation_backward_cuda_launcher(int n, int c, int k, const float *grad_output, const int *idx,
                                          const float *weight, float *grad_input);
/* subtraction/subtraction_cuda_kernel.h:14-15 */
void subtraction_forward_cuda_launcher(int n, int nsample, int c, const float *input1, const float *input2,
                                       const int *idx, float *output);
void subtraction_backward_cuda_launcher(int n, int nsample, int c, const int *idx, const float *grad_output,
                                        float *grad_input1, float *grad_input2);
/* aggregation/aggregation_cuda_kernel.h:14-15 */
void aggregation_forward_cuda_launcher(int n, int nsample, int c, int w_c, const float *input,
                                       const float *position, const float *weight, const int *idx, float *output);
void aggregation_backward_cuda_launcher(int n, int nsample, int c, int w_c, const float *input,
                                        const float *position, const float *weight, const int *idx,
                                        const float *grad_output, float *grad_input, float *grad_position,
                                        float *grad_weight);

/* ------------------------------------------------------------------------------------------
 * 2. Stream-aware forms of the same operators.
 * ---------------------------------------------------------------------------------------- */
/*
 * Farthest point sampling over b packed clouds (pointops.py:10-27).  n_max = largest cloud.
 * tmp may be NULL unless a cloud exceeds tgn_fps_resident_capacity() points (then it must hold
 * offset[b-1] floats; contents on entry are ignored).  new_xyz (m,3) is optional (NULL to skip): the
 * sampled coordinates, i.e. xyz[idx] (pointnet2_utils.py:276 / blocks.py:70).
 */
int tgn_furthestsampling(int b, int n_max, const float *xyz, const int *offset, const int *new_offset,
                         float *tmp, void *idx, float *new_xyz, int flags, tgn_stream_t stream);
/* Dense batch (B,N,3) -> (B,S): the offsets of pointnet2_utils.py:87-96 are implicit. */
int tgn_furthestsampling_dense(int B, int N, int S, const float *xyz, float *tmp, void *idx, float *new_xyz,
                               int flags, tgn_stream_t stream);
int tgn_fps_resident_capacity(void);
/*
 * Clouds larger than tgn_fps_resident_capacity() (raw scans, preprocess_data.py:55-56) run the bucket-skipping
 * kernel out of a cell-sorted workspace of tgn_fps_workspace_bytes(b, n_max) bytes (20 B per point; 0 when every
 * cloud fits the register-resident kernels; clouds above 262 144 points fall back to streaming through the same
 * buffer used as the reference's tmp array, which then must hold 4 B per point of the batch).
 */
size_t tgn_fps_workspace_bytes(int b, int n_max);
/* Workspace of the TGN_FPS_THROUGHPUT form (20 B per point, whatever the cloud size up to 32 768 points; 0 above). */
size_t tgn_fps_throughput_workspace_bytes(int b, int n_max);
int tgn_furthestsampling_ws(int b, int n_max, const float *xyz, const int *offset, const int *new_offset,
                            void *workspace, size_t workspace_bytes, void *idx, float *new_xyz, int flags,
                            tgn_stream_t stream);
int tgn_furthestsampling_dense_ws(int B, int N, int S, const float *xyz, void *workspace, size_t workspace_bytes,
                                  void *idx, float *new_xyz, int flags, tgn_stream_t stream);
/*
 * FPS of an FPS result is the identity: if a cloud IS the sequence p_0, p_1, ... produced by farthest point sampling
 * (canonical first-index tie order, same arithmetic flags), sampling it again returns positions 0, 1, 2, ... -- which
 * is what the reference's set-abstraction / transition-down chains compute at every level after the first
 * (pointnet2_utils.py:160 on the previous level's new_xyz; blocks.py:69-70).  It holds as long as every winning
 * distance of the producing run was > 0 and < 1e10 (no exhausted cloud, no NaN/Inf point); the winning distance never
 * increases, so the first and the last one decide.
 *   prefix_out[b] (int32 per cloud, optional): this result's samples 0 .. prefix_out[b]-1 carry the property -- the
 *                 requested sample count when the run stayed inside (0, 1e10), 1 (nothing claimed) otherwise.
 *   prefix_in[b]  (int32 per cloud, optional): cloud b of xyz is such a sequence up to prefix_in[b] samples; when that
 *                 covers the requested sample count (and TGN_FPS_TREE_TIES is not set) the kernel writes the identity
 *                 (indices, new_xyz, prefix_out) and returns -- decided on the device, per cloud, no host sync.
 * The caller vouches for provenance: prefix_in must come from the prefix_out of the launch that produced xyz.
 * prefix_ref (optional, same layout as xyz): the new_xyz that launch wrote; the kernel then takes the shortcut for a
 * cloud only if its coordinates equal prefix_ref bit for bit -- provenance by content, for callers that cannot
 * vouch for it (the model gathers p[idx] itself, blocks.py:70, or hands tensors through arbitrary code).
 */
int tgn_furthestsampling_prefix(int b, int n_max, const float *xyz, const int *offset, const int *new_offset,
                                void *workspace, size_t workspace_bytes, void *idx, float *new_xyz, const int *prefix_in,
                                const float *prefix_ref, int *prefix_out, int flags, tgn_stream_t stream);
int tgn_furthestsampling_dense_prefix(int B, int N, int S, const float *xyz, void *workspace, size_t workspace_bytes,
                                      void *idx, float *new_xyz, const int *prefix_in, const float *prefix_ref,
                                      int *prefix_out, int flags, tgn_stream_t stream);

/* kNN (pointops.py:30-45): b segments; idx (m,nsample) int32; dist2 (m,nsample) squared, ascending. */
int tgn_knnquery(int b, int m, int nsample, const float *xyz, const float *new_xyz, const int *offset,
                 const int *new_offset, int *idx, float *dist2, tgn_stream_t stream);
/*
 * Same result, ~30x faster: with tgn_knnquery_workspace_bytes(m) bytes of device scratch the wave-parallel
 * kernel runs first and only queries whose k+1 smallest distances tie bit-for-bit are redone by the exact
 * heap kernel (whose insertion history decides the order of tied neighbours, knnquery_cuda_kernel.cu:21-48).
 */
size_t tgn_knnquery_workspace_bytes(int m);
int tgn_knnquery_ws(int b, int m, int nsample, const float *xyz, const float *new_xyz, const int *offset,
                    const int *new_offset, int *idx, float *dist2, void *workspace, size_t workspace_bytes,
                    tgn_stream_t stream);
/*
 * The same result again through per-segment uniform grids (cell size ~ the k-neighbour radius of a scan surface): a
 * query looks at the 3x3x3 cells around it instead of the whole segment and widens the block only if the (k+1)-th
 * distance does not lie inside the covered radius.  n = total number of points (rows of xyz).  Pays from a few
 * thousand points per segment up; small or degenerate segments are scanned linearly inside the same launch, and
 * nsample > 63 or too small a workspace fall back to tgn_knnquery_ws.  (24 000 x 24 000, k = 36: see DESIGN.md.)
 */
size_t tgn_knnquery_grid_workspace_bytes(int b, int n, int m);
int tgn_knnquery_grid(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, const int *offset,
                      const int *new_offset, int *idx, float *dist2, void *workspace, size_t workspace_bytes,
                      tgn_stream_t stream);

int tgn_grouping_forward(int m, int nsample, int c, const float *input, const int *idx, float *output,
                         tgn_stream_t stream);
int tgn_grouping_backward(int m, int nsample, int c, const float *grad_output, const int *idx, float *grad_input,
                          tgn_stream_t stream);
int tgn_interpolation_forward(int n, int c, int k, const float *input, const int *idx, const float *weight,
                              float *output, tgn_stream_t stream);
int tgn_interpolation_backward(int n, int c, int k, const float *grad_output, const int *idx, const float *weight,
                               float *grad_input, tgn_stream_t stream);
int tgn_subtraction_forward(int n, int nsample, int c, const float *input1, const float *input2, const int *idx,
                            float *output, tgn_stream_t stream);
int tgn_subtraction_backward(int n, int nsample, int c, const int *idx, const float *grad_output,
                             float *grad_input1, float *grad_input2, tgn_stream_t stream);
int tgn_aggregation_forward(int n, int nsample, int c, int w_c, const float *input, const float *position,
                            const float *weight, const int *idx, float *output, tgn_stream_t stream);
int tgn_aggregation_backward(int n, int nsample, int c, int w_c, const float *input, const float *position,
                             const float *weight, const int *idx, const float *grad_output, float *grad_input,
                             float *grad_position, float *grad_weight, tgn_stream_t stream);

/*
 * Point-Transformer vector attention (models/modules/cbl_point_transformer/blocks.py:31-44), packed layout.
 *
 * tgn_pt_attention_forward: the whole PointTransformerLayer after its three input projections, eval mode:
 *     p_r = linear_p(p[idx] - p)                      Linear(3,3) + BatchNorm1d(3) folded into (Wp1, bp1), ReLU, Linear(3,c) = (Wp2, bp2)
 *     w   = linear_w(x_k[idx] - x_q + p_r)            BatchNorm1d(c) as (a1, t1), ReLU, Linear(c,g) + BatchNorm1d(g) folded into
 *                                                     (Ww1, bw1), ReLU, Linear(g,g) = (Ww2, bw2);  g = c / share_planes
 *     out = sum_j (x_v[idx_j] + p_r_j) * softmax_j(w)[.., ch % g]
 *   p (n,3), x_q / x_k / x_v (n,c), idx (n,nsample) int32 neighbour rows (the kNN of p among p), out (n,c).
 *   post_scale / post_shift (c each, or both NULL): out = relu(out * post_scale + post_shift) -- the BatchNorm + ReLU the block
 *   applies to the layer's output (blocks.py:151), folded.
 *   One kernel, nothing of size n*nsample*c is written.  nsample <= 64, c % 4 == 0, g in {4,8,16,32,64}.
 * tgn_pt_softmax_aggregate_{forward,backward}: the trainable tail alone (blocks.py:41-43): softmax over the
 *   neighbours of logit (n,nsample,g) -> sm (kept for backward), out[n,ch] = sum_j (x_v[idx[n,j],ch] + p_r[n,j,ch]) *
 *   sm[n,j,ch % g]; the backward writes grad_pr (n,nsample,c), grad_logit (n,nsample,g) and ACCUMULATES into grad_xv
 *   (n_v,c; pre-zeroed).  This is the reference's `aggregation` (aggregation_cuda_kernel.cu:5-39) with the softmax
 *   fused in front.
 */
int tgn_pt_attention_forward(int n, int nsample, int c, int g, const float *p, const float *xq, const float *xk,
                             const float *xv, const int *idx, const float *Wp1, const float *bp1, const float *Wp2,
                             const float *bp2, const float *a1, const float *t1, const float *Ww1, const float *bw1,
                             const float *Ww2, const float *bw2, const float *post_scale, const float *post_shift, float *out,
                             tgn_stream_t stream);
int tgn_pt_softmax_aggregate_forward(int n, int nsample, int c, int g, const float *xv, const float *pr, const float *logit,
                                     const int *idx, float *sm, float *out, tgn_stream_t stream);
int tgn_pt_softmax_aggregate_backward(int n, int nsample, int c, int g, const float *xv, const float *pr, const float *sm,
                                      const int *idx, const float *grad_out, float *grad_xv, float *grad_pr,
                                      float *grad_logit, tgn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * 3. pointnet2_utils operators as single kernels (dense (B,N,*) layout).
 * ---------------------------------------------------------------------------------------- */
/* square_distance (pointnet2_utils.py:20-41) for 3-D points: src (B,N,3), dst (B,M,3) -> out (B,N,M). */
int tgn_square_distance(int B, int N, int M, const float *src, const float *dst, float *out, tgn_stream_t stream);
/*
 * query_ball_point (pointnet2_utils.py:120-144): first nsample indices in ascending index order with
 * square_distance(new_xyz, xyz) <= r2 (expanded-form arithmetic of pointnet2_utils.py:20-41), padded
 * with the first hit; a row without any hit is filled with N.  idx: (B,S,nsample) int32 or int64.
 * workspace: tgn_ball_query_workspace_bytes(B,N,S) bytes of device scratch (may be NULL if 0).
 */
size_t tgn_ball_query_workspace_bytes(int B, int N, int S);
int tgn_ball_query(int B, int N, int S, int nsample, float r2, const float *xyz, const float *new_xyz, void *idx,
                   int idx_is_int64, void *workspace, size_t workspace_bytes, tgn_stream_t stream);
/*
 * The same in two launches, for planners: _build fills the workspace (per-cloud uniform grid: depends on xyz and r2 only,
 * not on the queries -- it can run while the sampling that produces new_xyz is still in flight), _prebuilt answers the
 * queries from a workspace filled by _build with the same (B, N, S, nsample, r2, xyz).  Same results as tgn_ball_query.
 */
/* out[r, 0..ncols) = in[r, first..first+ncols) for `rows` rows of `stride` floats: the xyz block of (N, 6) scan rows
 * (gen_utils.py:138, pointnet_pp_model.py:16-20) without torch's strided-copy kernel (10x slower beside a running FPS launch) */
int tgn_slice_columns(long long rows, int stride, int first, int ncols, const float *in, float *out, tgn_stream_t stream);
/* scheduling spacer: a one-wave kernel that idles for about `microseconds` on `stream` (planners: hold one stream's work
 * back behind another's start without a host round trip) */
int tgn_stream_delay(int microseconds, tgn_stream_t stream);
int tgn_ball_query_build(int B, int N, int S, int nsample, float r2, const float *xyz, void *workspace,
                         size_t workspace_bytes, tgn_stream_t stream);
int tgn_ball_query_prebuilt(int B, int N, int S, int nsample, float r2, const float *xyz, const float *new_xyz, void *idx,
                            int idx_is_int64, void *workspace, size_t workspace_bytes, tgn_stream_t stream);
/*
 * Grouping of sample_and_group (pointnet2_utils.py:162-169, xyz_first=1: [xyz[idx]-new_xyz, points[idx]])
 * and of PointNetSetAbstractionMsg (pointnet2_utils.py:281-285, xyz_first=0: [points[idx], xyz[idx]-new_xyz]).
 * points (B,N,D) may be NULL (D ignored).  out: (B,S,K,3+D).
 */
int tgn_group_points(int B, int N, int S, int K, int D, const float *xyz, const float *new_xyz, const float *points,
                     const void *idx, int idx_is_int64, int xyz_first, float *out, tgn_stream_t stream);
/*
 * The same with the launch knobs exposed.  impl: 0 = choose, 1 = 4-B stores, 2 = 16-B stores through LDS (needs
 * K*(3+D) % 4 == 0).  store_policy: cache-policy bits of the 16-B output stores (0 plain, 2 nt, 16 sc1 = write-through,
 * the output lines do not stay in L2; -1 = the kernel's default: nt for the LDS-image kernels, sc1 for the others).  max_blocks: upper bound on the grid (0 = none) for callers
 * that overlap the grouping with a kernel that needs most of every CU (the FPS level-1 workgroups).
 */
int tgn_group_points_ex(int B, int N, int S, int K, int D, const float *xyz, const float *new_xyz,
                        const float *points, const void *idx, int idx_is_int64, int xyz_first, float *out, int impl,
                        int store_policy, int max_blocks, tgn_stream_t stream);
/*
 * Fused first layer of a set-abstraction shared MLP, eval mode (pointnet2_utils.py:229-236, 281-294): the 1x1
 * convolution commutes with the gather, so the caller transforms the POINTS once (A = scale*(W_p*points + W_x*xyz),
 * (B,N,C)) and folds bias / BatchNorm / the centre term into Cst (B,S,C); this writes
 *   out[b,s,k,:] = act(A[b, idx[b,s,k], :] + Cst[b,s,:])   (B,S,K,C)      -- the grouped (B,S,K,3+D) tensor never exists
 * and the _max form reduces over k as well (single-layer MLPs): out (B,S,C).  relu != 0 applies max(.,0).
 */
int tgn_sa_first_layer(int B, int N, int S, int K, int C, const float *A, const float *Cst, const void *idx,
                       int idx_is_int64, int relu, float *out, tgn_stream_t stream);
int tgn_sa_first_layer_max(int B, int N, int S, int K, int C, const float *A, const float *Cst, const void *idx,
                           int idx_is_int64, int relu, float *out, tgn_stream_t stream);
/*
 * The three kernels of the fused set-abstraction level (eval mode, BatchNorm folded; pointnet2_utils.py:229-236,
 * 281-294).  With scale = gamma/sqrt(var+eps), shift = beta - mean*scale and W = the first 1x1 convolution:
 *
 * tgn_sa_point_transform: A[m,:] = [points[m,:], xyz[m,:]] * Wt for the M = B*N points of a batch -- the per-POINT half
 *   of the layer (the convolution commutes with the gather).  Wt: (D+3, C1) row-major, rows ordered
 *   [feature channels..., x, y, z], columns already multiplied by scale.  Hand-written fp32-MFMA GEMM
 *   (v_mfma_f32_32x32x2_f32: exact fp32 fma chain).
 * tgn_sa_gather_max: out[b,s,c] = act( max_k A[b, idx[b,s,k], c] - (Wxs[0,c]*cx + Wxs[1,c]*cy + Wxs[2,c]*cz) + b2[c] )
 *   with (cx,cy,cz) = new_xyz[b,s], Wxs = the x,y,z rows of Wt, b2 = shift + scale*bias: the whole single-layer
 *   set-abstraction level, (B,S,C1) out, nothing of size S*K written.  nsample <= 64, C1 % 4 == 0.
 * tgn_sa_direct_max: the same result without the per-point tensor for narrow inputs (3+D <= 16, C1 % 32 == 0,
 *   C1 <= 256, nsample <= 64; tgn_sa_direct_supported): a wave owns a query and contracts its K gathered rows
 *   [x-c, f] (D+3) with Wd on the matrix cores.  Wd: (16, C1), rows [x, y, z, f0.., zero padding], scale folded.
 */
int tgn_sa_point_transform(long long M, int D, int C1, const float *xyz, const float *points, const float *Wt, float *A,
                           tgn_stream_t stream);
int tgn_sa_gather_max(int B, int N, int S, int K, int C1, const float *A, const float *new_xyz, const float *Wxs,
                      const float *b2, const void *idx, int idx_is_int64, int relu, float *out, tgn_stream_t stream);
/* First layer of a MULTI-layer shared MLP: out[b,s,k,c] = act(A[b,idx[b,s,k],c] - Wxs[:,c].centre + b2[c]), (B,S,K,C1). */
int tgn_sa_gather_act(int B, int N, int S, int K, int C1, const float *A, const float *new_xyz, const float *Wxs,
                      const float *b2, const void *idx, int idx_is_int64, int relu, float *out, tgn_stream_t stream);
int tgn_sa_direct_supported(int K, int D, int C1);
int tgn_sa_direct_max(int B, int N, int S, int K, int D, int C1, const float *xyz, const float *new_xyz,
                      const float *points, const float *Wd, const float *b2, const void *idx, int idx_is_int64, int relu,
                      float *out, tgn_stream_t stream);
/*
 * tgn_sa_mlp2_max: a WHOLE set-abstraction level with a TWO-layer shared MLP (what every level of the reference networks
 * has: pointnet_pp.py:13-15, tsg_centroid_module.py:10-12, tsg_seg_module.py:11-28), eval mode, both BatchNorms folded:
 *   out[b,s,:] = max_k relu(W2 * relu(W1 * [x[idx]-c, f[idx]] + b1) + b2)        (pointnet2_utils.py:281-294)   (B,S,C2)
 * in ONE kernel: neither the grouped tensor nor the (B,S,K,C1) / (B,S,K,C2) layer outputs are written.  The second layer
 * runs on the fp32 matrix cores (exact fp32).  First layer, two forms:
 *   commuted (A1 != NULL): A1 (B,N,C1p) = tgn_sa_point_transform of the level's points with the first layer's folded
 *     weights; W1 = Wxs (3,C1p), the x,y,z rows of those weights; xyz / points unused (may be NULL);
 *   direct (A1 == NULL, 3+D <= 16; tgn_sa_mlp2_direct_supported): W1 = Wd (16,C1p), rows [x, y, z, f0.., zero padding].
 * C1p = the first layer's width padded with zero columns to a multiple of 16 (b1, W1, A1 padded alike);
 * W2f (C1p/8, C2, 8): W2f[kb][c][i] = scale2[c] * W2[c, 8*kb + i] (zero for padded k); b2 (C2) = shift2 + scale2*bias2.
 * nsample <= 64; idx int32 or int64, local to each cloud, out-of-range handled like tgn_group_points.
 * out_stride: floats between consecutive rows of `out` (0 = C2): a multi-scale level lets every (radius, nsample) branch write
 * its columns of the concatenated (B,S,sum C2) tensor directly (pointnet2_utils.py:296-298 concatenates them).
 */
int tgn_sa_mlp2_direct_supported(int K, int D);
int tgn_sa_mlp2_max(int B, int N, int S, int K, int D, int C1p, int C2, const float *A1, const float *xyz,
                    const float *points, const float *new_xyz, const float *W1, const float *b1, const void *idx,
                    int idx_is_int64, const float *W2f, const float *b2, float *out, int out_stride, tgn_stream_t stream);
/*
 * The same level with the second layer on the BF16 matrix cores at fp32 accuracy ("bf16x3").  v_mfma_f32_32x32x16_bf16 runs at
 * 16x the rate of the fp32-input MFMA; both operands are written as sums of three bf16 numbers (24 mantissa bits) and a product is
 * six bf16 MFMAs with fp32 accumulation -- a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1, the dropped terms are below 2^-24 of the
 * product -- i.e. fp32-class rounding at up to 16 / 6 = 2.7x the fp32-MFMA rate.  Not bit-identical to tgn_sa_mlp2_max; held to the
 * same elementwise 1e-5 bound against the float64 oracle.
 *   tgn_sa_mlp2_split_bytes(C1p, C2)    size of the split weight image;
 *   tgn_sa_mlp2_split_weights           W2f (C1p/8, C2, 8) fp32 -> image (device to device, once per weight matrix): per 128-column
 *                                       tile and 16-wide K tile the 12 KiB the kernel's LDS tile holds, fetched by LDS-DMA;
 *   tgn_sa_mlp2_max_bf16x3              tgn_sa_mlp2_max with W2s (the image) in place of W2f; every other argument as above.
 */
size_t tgn_sa_mlp2_split_bytes(int C1p, int C2);
/* tgn_sa_point_transform on the same scheme: Wts = tgn_sa_mlp2_split_weights(Kp, C1, Wtf) with Wtf (Kp/8, C1, 8),
 * Wtf[kb][c][i] = Wt[8 kb + i][c] (rows [features..., x, y, z], zero rows past D + 3), Kp = D + 3 rounded up to a multiple of 16. */
int tgn_sa_point_transform_bf16x3(long long M, int D, int Kp, int C1, const float *xyz, const float *points, const void *Wts,
                                  float *A, tgn_stream_t stream);
int tgn_sa_mlp2_split_weights(int C1p, int C2, const float *W2f, void *W2s, tgn_stream_t stream);
int tgn_sa_mlp2_max_bf16x3(int B, int N, int S, int K, int D, int C1p, int C2, const float *A1, const float *xyz,
                           const float *points, const float *new_xyz, const float *W1, const float *b1, const void *idx,
                           int idx_is_int64, const void *W2s, const float *b2, float *out, int out_stride, tgn_stream_t stream);
/*
 * tgn_sa_all_mlp2_max: PointNetSetAbstraction with group_all=True -- the only form of that module a reference model builds
 * (models/modules/tsg_seg_module.py:28: 515 -> [256, 512] over the 256 points of the last level) -- eval mode, BatchNorms folded:
 *   out[b,:] = max_n relu(W2 * relu(W1 * [x_n, f_n] + b1) + b2)     (pointnet2_utils.py:178-195 + 229-236)     (B,C2)
 * The whole cloud is one group, there is no centre and no index tensor: chunks of 64 consecutive points run through the
 * two-layer kernel of tgn_sa_mlp2_max, a second small launch takes the maximum over a cloud's chunks.  Operands as for
 * tgn_sa_mlp2_max, with the group_all channel order [x, y, z, features] (pointnet2_utils.py:190): commuted form A1 (B,N,C1p) =
 * tgn_sa_point_transform (Wd unused, may be NULL), direct form A1 == NULL with xyz, points, Wd (16,C1p).
 * part: workspace of tgn_sa_all_chunks(N) * B * C2 floats (may be NULL when tgn_sa_all_chunks(N) == 1).
 */
int tgn_sa_all_chunks(int N);
int tgn_sa_all_mlp2_max(int B, int N, int D, int C1p, int C2, const float *A1, const float *xyz, const float *points,
                        const float *Wd, const float *b1, const float *W2f, const float *b2, float *part, float *out,
                        int out_stride, tgn_stream_t stream);
/*
 * Weight gradient of a linear layer with few columns and many (or not so many) rows -- the training path of the Point-Transformer
 * mirrors; blocks.py:19-30 declares the layers:
 *   dW[o][i] = sum_r gy[r][o] * x[r][i]      (cout, cin)         db[o] = sum_r gy[r][o]   (optional, NULL to skip)
 * a contraction over the ROWS with a small result, which the BLAS libraries walk with one or two tiles (94 us for a 256 x 256
 * gradient over 375 rows).  One wave per (row slice, 32 x 32 output tile) on the fp32 matrix cores straight from global memory,
 * then one reduction over the slices.  x (rows, cin), gy (rows, cout), row-major fp32; workspace of
 * tgn_linear_wgrad_workspace_bytes(rows, cin, cout) bytes.
 */
size_t tgn_linear_wgrad_workspace_bytes(long long rows, int cin, int cout);
int tgn_linear_wgrad(long long rows, int cin, int cout, const float *x, const float *gy, float *dW, float *db, void *workspace,
                     size_t workspace_bytes, tgn_stream_t stream);

/*
 * Training-mode BatchNorm1d over the rows of x (rows, C), optionally fused with the ReLU that follows it: the normalisations of
 * the Point-Transformer training path (blocks.py:37,40 via nn.BatchNorm1d; this package normalises the flattened (n * nsample, c)
 * rows).  forward: batch statistics in double, y = [relu]((x - mean) * invstd * gamma + beta), save_mean / save_invstd for the
 * backward pass, running statistics updated like nn.BatchNorm1d (running = (1 - momentum) running + momentum batch, variance
 * unbiased; NULL running pointers: not tracked), *num_batches_tracked += 1 when given.  backward: dx, dgamma, dbeta; with relu
 * the mask is y > 0 (y = the forward output).  workspace: tgn_bn_rows_workspace_bytes(C) bytes, ZERO before the first use -- the
 * kernels leave it zeroed, so one buffer serves every call of a layer on one stream.  rows >= 2, 1 <= C <= 1024, fp32, contiguous.
 */
size_t tgn_bn_rows_workspace_bytes(int C);
int tgn_bn_rows_forward(long long rows, int C, const float *x, const float *gamma, const float *beta, float eps, float momentum,
                        float *running_mean, float *running_var, long long *num_batches_tracked, int relu, float *y,
                        float *save_mean, float *save_invstd, void *workspace, tgn_stream_t stream);
int tgn_bn_rows_backward(long long rows, int C, const float *x, const float *y, const float *dy, const float *gamma,
                         const float *save_mean, const float *save_invstd, int relu, float *dx, float *dgamma, float *dbeta,
                         void *workspace, tgn_stream_t stream);
/* index_points (pointnet2_utils.py:44-61): out[b,j,:] = points[b, idx[b,j], :], idx flattened to (B,M). */
int tgn_gather_points(int B, int N, int M, int C, const float *points, const void *idx, int idx_is_int64, float *out,
                      tgn_stream_t stream);
/* backward of tgn_gather_points / tgn_group_points feature part: grad_points[b, idx[b,j], :] += grad_out[b,j,:] */
int tgn_scatter_add_points(int B, int N, int M, int C, const float *grad_out, const void *idx, int idx_is_int64,
                           float *grad_points, tgn_stream_t stream);
/*
 * three nearest support points (pointnet2_utils.py:333-335): xyz1 (B,N,3) queries, xyz2 (B,S,3) support;
 * dist (B,N,3) expanded-form squared distances ascending by (dist, index); idx int32 or int64.
 */
int tgn_three_nn(int B, int N, int S, const float *xyz1, const float *xyz2, float *dist, void *idx, int idx_is_int64,
                 tgn_stream_t stream);
/* inverse-distance weighted sum (pointnet2_utils.py:337-340); weight (B,N,3) is also written if non-NULL. */
int tgn_three_interpolate(int B, int N, int S, int C, const float *points2, const float *dist, const void *idx,
                          int idx_is_int64, float *out, float *weight, tgn_stream_t stream);
/* the same with the epilogue of the eval-mode feature propagation fused in (pointnet2_utils.py:337-347, first convolution
 * commuted onto the coarse points): out = [relu](interpolation (+ add)); add (B,N,C) may be NULL and may be `out` itself. */
int tgn_three_interpolate_ex(int B, int N, int S, int C, const float *points2, const float *dist, const void *idx,
                             int idx_is_int64, const float *add, int relu, float *out, float *weight, tgn_stream_t stream);
/*
 * Gather indices follow torch's advanced indexing (pointnet2_utils.py:56-60): negative values wrap (k + N); an
 * index still outside [0,N) -- where the reference raises, e.g. an empty ball yields index N -- makes
 * tgn_gather_points write a zero row and tgn_group_points / tgn_sa_* read point 0, and latches a flag in the
 * current device's memory -- one flag per (device, stream), so host threads that drive different streams do not see or
 * clear each other's.  This returns 1 and clears the flag if that happened in a launch on `stream` since the last call;
 * it synchronises `stream`.  (The Python operators call it and raise IndexError.)
 */
int tgn_take_index_error(tgn_stream_t stream);
/* The same over every stream of the current device (synchronises the DEVICE, ORs and clears all of its flags): for planners
 * that launch unchecked on streams of their own (HotPath), graphs replayed on another stream than they were captured on
 * (a graph keeps the capture stream's flag), and TGN_INDEX_CHECK=off sections. */
int tgn_take_index_error_device(void);
/* Clears the flag in stream order without synchronising: what a checked operator issues in front of its own launch, so
 * that a bit latched by an earlier UNCHECKED launch is not attributed to it. */
int tgn_clear_index_error(tgn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * 4. Mesh input of the preprocess path (HOST pointers, CPU code): gen_utils.read_txt_obj_ls (gen_utils.py:207-233).
 * ---------------------------------------------------------------------------------------- */
/*
 * Text-OBJ reader with the reference loop's semantics (gen_utils.py:211-226): only lines whose first token is exactly
 * "v" or "f" count, reading stops at the first blank line, "a//b" face references are cut at "//", anything int() /
 * float() would reject is an error.  vertices (cap_v,3) double, faces (cap_f,3) int64, 1-based as in the file.
 */
int tgn_obj_count(const char *path, long long *n_vertices, long long *n_faces);
int tgn_obj_read(const char *path, double *vertices, long long *faces, long long cap_v, long long cap_f,
                 long long *n_vertices, long long *n_faces);
/*
 * Vertex normals as open3d's compute_vertex_normals() (gen_utils.py:228-233): area-weighted sum of the triangle
 * cross products, normalised, (0,0,1) where undefined.  triangles are ZERO-based.  Parity unpinned (no open3d here).
 */
int tgn_vertex_normals(const double *vertices, long long nv, const long long *triangles, long long nf, double *normals);
/*
 * One scan of the preprocess loop in one call that never holds the interpreter lock (preprocess_data.py:37-52): reads the
 * ground-truth json ({"jaw": "upper"|"lower", "labels": [FDI numbers]}), remaps the labels to 0..16 (:39-44), reads the OBJ
 * with tgn_obj_read's semantics, computes the vertex normals, centres the vertices and maps [y_min, y_max] to [-1, 1]
 * (:48-50, with numpy's summation order), and holds the (n, 7) float64 rows [x y z nx ny nz label] behind *handle.
 * jaw receives the json's "jaw" string (NUL-terminated, at most jaw_cap - 1 bytes).
 * TGN_ERR_UNSUPPORTED: the json is not the plain shape above (floats among the labels, escapes, missing keys) -- the caller
 * falls back to a general json parser; TGN_ERR_INVALID_ARGUMENT: what the reference raises on (unreadable file, bad OBJ
 * token, label count != vertex count).
 */
int tgn_scan_open(const char *obj_path, const char *json_path, double y_min, double y_max, void **handle,
                  long long *n_vertices, char *jaw, int jaw_cap);
/* Copies the rows to labeled (n,7) double and, when xyz32 is not NULL, the float32 coordinates to xyz32 (n,3); frees the
 * handle (both pointers NULL: only frees). */
int tgn_scan_take(void *handle, double *labeled, float *xyz32);
/* The loader recycles its per-scan scratch (file text, vertices, faces, normals, rows: ~25 MB for a 100 000-vertex scan) through a pool of
 * at most 64 objects; this frees them (returns how many).  Safe at any time: tgn_scan_open allocates afresh when the pool is empty. */
int tgn_scan_pool_trim(void);

#ifdef __cplusplus
}
#endif
#endif /* TGN_POINTOPS_H */

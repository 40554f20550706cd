// pt_attention.hip -- the Point-Transformer vector attention of models/modules/cbl_point_transformer/blocks.py:31-44
// ("subtraction attention"), SURVEY.md 8(f)2.
//
// The reference runs it as ~25 torch kernels over (n, nsample, c) tensors: two kNN searches, two queryandgroup
// gathers, linear_p on the relative coordinates, x_k - x_q + p_r, linear_w, a softmax over the neighbours and the
// share_planes-broadcast weighted sum.  Here:
//   * tgn_pt_attention_forward: the WHOLE layer after the three input projections, eval mode (BatchNorm folded), as one
//     kernel -- a wave owns a point, lane j owns neighbour j: it gathers that neighbour's key / value rows and its
//     relative coordinates, evaluates both small MLPs in registers, the softmax runs across the lanes (DPP
//     reductions), the weighted sum likewise.  Nothing of size n*nsample*c touches memory: per point the kernel reads
//     2*nsample*c*4 B of gathered rows (L2) and writes c*4 B.
//   * tgn_pt_softmax_aggregate_{forward,backward}: the trainable tail (softmax over the neighbours + sum_j (x_v[idx_j] +
//     p_r_j) * w_j with the share_planes broadcast, blocks.py:41-43) for training, where the learned layers in between
//     need batch statistics and stay torch modules.  This is the reference's `aggregation` operator
//     (aggregation_cuda_kernel.cu:5-39) with the softmax fused in front and without its per-element atomics on
//     grad_weight.
#include "tgn_common.h"

namespace tgn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// sum / max over the 64 lanes of a wave (DPP row shifts + row broadcasts; result uniform)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v, float identity) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum_f32(float v) {
    v += dpp_f32<0x111, 0xF>(v, 0.0f);
    v += dpp_f32<0x112, 0xF>(v, 0.0f);
    v += dpp_f32<0x114, 0xF>(v, 0.0f);
    v += dpp_f32<0x118, 0xF>(v, 0.0f);   // lane 15 of each row: the row sum
    v += dpp_f32<0x142, 0xA>(v, 0.0f);   // row_bcast:15 into rows 1,3
    v += dpp_f32<0x143, 0xC>(v, 0.0f);   // row_bcast:31 into rows 2,3: lane 63 holds the wave sum
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max_f32x(float v) {
    v = fmaxf(v, dpp_f32<0x111, 0xF>(v, -INFINITY));
    v = fmaxf(v, dpp_f32<0x112, 0xF>(v, -INFINITY));
    v = fmaxf(v, dpp_f32<0x114, 0xF>(v, -INFINITY));
    v = fmaxf(v, dpp_f32<0x118, 0xF>(v, -INFINITY));
    v = fmaxf(v, dpp_f32<0x142, 0xA>(v, -INFINITY));
    v = fmaxf(v, dpp_f32<0x143, 0xC>(v, -INFINITY));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

struct PtParams {   // folded parameters of one PointTransformerLayer (device pointers)
    const float *Wp1, *bp1;   // (3,3), (3): linear_p[0] with linear_p[1] (BatchNorm1d(3)) folded
    const float *Wp2, *bp2;   // (c,3), (c): linear_p[3]
    const float *a1, *t1;     // (c), (c): linear_w[0] (BatchNorm1d(c)) as scale / shift
    const float *Ww1, *bw1;   // (g,c), (g): linear_w[2] with linear_w[3] (BatchNorm1d(g)) folded
    const float *Ww2, *bw2;   // (g,g), (g): linear_w[5]
    const float *post_s, *post_t;   // optional (c), (c): out = relu(out * post_s + post_t) -- the block's bn2 + ReLU (blocks.py:151)
};

// One wave per point, lane j = neighbour j (nsample <= 64), G = c / share_planes weight channels.
template <int G>
__global__ __launch_bounds__(256) void pt_attention_fwd_kernel(int n, int nsample, int c, const float *__restrict__ p,
                                                                const float *__restrict__ xq, const float *__restrict__ xk,
                                                                const float *__restrict__ xv, const int *__restrict__ idx,
                                                                PtParams P, float *__restrict__ out) {
    const unsigned lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool act = lane < (unsigned)nsample;
    for (int pt = blockIdx.x * 4 + wv; pt < n; pt += gridDim.x * 4) {
        const int nbr = act ? idx[(size_t)pt * nsample + lane] : 0;
        // relative coordinates (queryandgroup: xyz[idx] - new_xyz, pointops.py:89-91) and the first half of linear_p
        const float r0 = p[(size_t)nbr * 3 + 0] - p[(size_t)pt * 3 + 0];
        const float r1 = p[(size_t)nbr * 3 + 1] - p[(size_t)pt * 3 + 1];
        const float r2 = p[(size_t)nbr * 3 + 2] - p[(size_t)pt * 3 + 2];
        float h[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            h[i] = fmaxf(((P.Wp1[i * 3 + 0] * r0 + P.Wp1[i * 3 + 1] * r1) + P.Wp1[i * 3 + 2] * r2) + P.bp1[i], 0.0f);
        const float *__restrict__ krow = xk + (size_t)nbr * c;
        const float *__restrict__ vrow = xv + (size_t)nbr * c;
        const float *__restrict__ qrow = xq + (size_t)pt * c;
        // ---- pass 1: w = x_k - x_q + p_r -> linear_w -> one logit per weight channel
        float lg[G];
#pragma unroll
        for (int g = 0; g < G; ++g) lg[g] = 0.0f;
        for (int ch = 0; ch < c; ch += 4) {
            const f32x4 k4 = *(const f32x4 *)(krow + ch);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cc = ch + i;   // wave-uniform: the parameter reads below are scalar loads
                const float pr = ((P.Wp2[cc * 3 + 0] * h[0] + P.Wp2[cc * 3 + 1] * h[1]) + P.Wp2[cc * 3 + 2] * h[2]) + P.bp2[cc];
                const float w = (k4[i] - qrow[cc]) + pr;
                const float u = fmaxf(P.a1[cc] * w + P.t1[cc], 0.0f);
#pragma unroll
                for (int g = 0; g < G; ++g) lg[g] += P.Ww1[(size_t)g * c + cc] * u;
            }
        }
        float hid[G], sm[G];
#pragma unroll
        for (int g = 0; g < G; ++g) hid[g] = fmaxf(lg[g] + P.bw1[g], 0.0f);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float a = P.bw2[g];
#pragma unroll
            for (int g2 = 0; g2 < G; ++g2) a += P.Ww2[g * G + g2] * hid[g2];
            // softmax over the neighbours (= over the lanes), blocks.py:41
            const float lgt = act ? a : -INFINITY;
            const float mx = wave_max_f32x(lgt);
            const float e = act ? __expf(lgt - mx) : 0.0f;
            sm[g] = e / wave_sum_f32(e);
        }
        // ---- pass 2: out[ch] = sum_j (x_v[idx_j, ch] + p_r_j[ch]) * sm_j[ch % G]   (blocks.py:42-43)
        for (int ch0 = 0; ch0 < c; ch0 += G) {
#pragma unroll
            for (int g = 0; g < G; g += 4) {
                const f32x4 v4 = *(const f32x4 *)(vrow + ch0 + g);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cc = ch0 + g + i;
                    const float pr = ((P.Wp2[cc * 3 + 0] * h[0] + P.Wp2[cc * 3 + 1] * h[1]) + P.Wp2[cc * 3 + 2] * h[2]) + P.bp2[cc];
                    const float s = wave_sum_f32(act ? (v4[i] + pr) * sm[g + i] : 0.0f);
                    if (lane == 0) out[(size_t)pt * c + cc] = P.post_s ? fmaxf(s * P.post_s[cc] + P.post_t[cc], 0.0f) : s;
                }
            }
        }
    }
}

// ---- trainable tail: softmax over the neighbours + share_planes aggregation ---------------------------------------
// forward: sm[n,j,g] = softmax_j(logit[n,j,g]);  out[n,ch] = sum_j (xv[idx[n,j],ch] + pr[n,j,ch]) * sm[n,j,ch % g_]
// One wave per point; lanes run along the channels (coalesced row reads), the neighbours are a loop.
__global__ __launch_bounds__(256) void pt_softmax_aggregate_fwd_kernel(int n, int nsample, int c, int g_,
                                                                        const float *__restrict__ xv,
                                                                        const float *__restrict__ pr,
                                                                        const float *__restrict__ logit,
                                                                        const int *__restrict__ idx, float *__restrict__ sm,
                                                                        float *__restrict__ out) {
    extern __shared__ float sm_s[];   // [4][nsample * g_]: the point's softmax weights
    const unsigned lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *sw = sm_s + (size_t)wv * nsample * g_;
    for (int pt = blockIdx.x * 4 + wv; pt < n; pt += gridDim.x * 4) {
        const float *__restrict__ lrow = logit + (size_t)pt * nsample * g_;
        float *__restrict__ srow = sm + (size_t)pt * nsample * g_;
        for (int g0 = (int)lane; g0 < g_; g0 += 64) {   // softmax over the neighbours, one weight channel per lane
            float mx = -INFINITY;
            for (int j = 0; j < nsample; ++j) mx = fmaxf(mx, lrow[j * g_ + g0]);
            float s = 0.0f;
            for (int j = 0; j < nsample; ++j) {
                const float e = expf(lrow[j * g_ + g0] - mx);   // expf, not __expf: the training path tracks torch.exp
                sw[j * g_ + g0] = e;                            // (parked: one exponential per weight, not two)
                s += e;
            }
            const float inv = 1.0f / s;
            for (int j = 0; j < nsample; ++j) {
                const float w = sw[j * g_ + g0] * inv;
                sw[j * g_ + g0] = w;
                srow[j * g_ + g0] = w;   // kept for the backward pass
            }
        }
        // sw was written by other lanes of this wave: the hardware completes a wave's LDS operations in order, the fence
        // and the wave barrier keep the COMPILER from moving the reads above the writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the point's neighbour indices once, one per lane (nsample <= 64): the loop below then has no load that depends on a load,
        // and four neighbours' rows are requested before the first is used (deep stages: 93 points x 512 channels were 192 dependent
        // round trips per wave, 56 us a launch)
        const int nbv = lane < (unsigned)nsample ? idx[(size_t)pt * nsample + lane] : 0;
        for (int ch = (int)lane; ch < c; ch += 64) {
            const int g = ch % g_;
            float acc = 0.0f;
            int j = 0;
            for (; nsample <= 64 && j + 4 <= nsample; j += 4) {
                float a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int nb = __builtin_amdgcn_readlane(nbv, j + u);
                    a[u] = xv[(size_t)nb * c + ch];
                    b[u] = pr[((size_t)pt * nsample + j + u) * c + ch];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc += (a[u] + b[u]) * sw[(j + u) * g_ + g];
            }
            for (; j < nsample; ++j) {   // (the tail, and every neighbour when there are more than a wave has lanes)
                const int nb = nsample <= 64 ? __builtin_amdgcn_readlane(nbv, j) : idx[(size_t)pt * nsample + j];
                acc += (xv[(size_t)nb * c + ch] + pr[((size_t)pt * nsample + j) * c + ch]) * sw[j * g_ + g];
            }
            out[(size_t)pt * c + ch] = acc;
        }
        __builtin_amdgcn_wave_barrier();   // the next point's weights overwrite sw
    }
}

// backward: given go = dL/dout (n,c):
//   d_pr[n,j,ch]   = go[n,ch] * sm[n,j,g]
//   d_xv[idx,ch]  += go[n,ch] * sm[n,j,g]                                   (atomic scatter)
//   d_sm[n,j,g]    = sum_{ch % g_ == g} go[n,ch] * (xv[idx,ch] + pr[n,j,ch])
//   d_logit[n,j,g] = sm[n,j,g] * (d_sm[n,j,g] - sum_j' sm[n,j',g] * d_sm[n,j',g])
__global__ __launch_bounds__(256) void pt_softmax_aggregate_bwd_kernel(int n, int nsample, int c, int g_,
                                                                        const float *__restrict__ xv,
                                                                        const float *__restrict__ pr,
                                                                        const float *__restrict__ sm,
                                                                        const int *__restrict__ idx,
                                                                        const float *__restrict__ go, float *__restrict__ d_xv,
                                                                        float *__restrict__ d_pr, float *__restrict__ d_logit) {
    extern __shared__ float dsm_s[];   // [4][nsample * g_]
    const unsigned lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *dsm = dsm_s + (size_t)wv * nsample * g_;
    const int s_ = c / g_;
    // merged form: needs a lane's weight channel to be lane % g_ in every 64-channel chunk (g_ divides 64) and one index per lane
    const bool merged = nsample <= 64 && (64 % g_) == 0;
    for (int pt = blockIdx.x * 4 + wv; pt < n; pt += gridDim.x * 4) {
        const float *__restrict__ srow = sm + (size_t)pt * nsample * g_;
        if (merged) {
            // ONE pass over (neighbour, channel): d_pr, the scatter into d_xv, and -- from the same loads of xv and pr -- the products
            // go * (xv + pr), summed over the channels that share a weight (lanes l, l + g_, l + 2 g_, ...: xor shuffles) into d_sm.
            // (The two-pass form read xv and pr a second time, 16 bytes at a time: lanes over (j, g) pairs.)
            const int nbv = lane < (unsigned)nsample ? idx[(size_t)pt * nsample + lane] : 0;
            for (int e = (int)lane; e < nsample * g_; e += 64) dsm[e] = 0.0f;
            // lanes e % 64 zero the entries, lanes < g_ accumulate into them: a cross-lane LDS hand-off like the ones below
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int c0 = 0; c0 < c; c0 += 64) {   // wave-uniform trip count: the shuffles need every lane
                const int ch = c0 + (int)lane;
                const bool in = ch < c;
                const int g = (int)lane % g_;
                const float gch = in ? go[(size_t)pt * c + ch] : 0.0f;
                for (int j0 = 0; j0 < nsample; j0 += 4) {
                    float a[4], b[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = j0 + u;
                        const bool ok = in && j < nsample;
                        const int nb = __builtin_amdgcn_readlane(nbv, j < nsample ? j : 0);
                        a[u] = ok ? xv[(size_t)nb * c + ch] : 0.0f;
                        b[u] = ok ? pr[((size_t)pt * nsample + j) * c + ch] : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = j0 + u;
                        if (j < nsample) {   // wave-uniform
                            const int nb = __builtin_amdgcn_readlane(nbv, j);
                            if (in) {
                                const float w = gch * srow[j * g_ + g];
                                d_pr[((size_t)pt * nsample + j) * c + ch] = w;
                                atomicAdd(d_xv + (size_t)nb * c + ch, w);
                            }
                            float t = gch * (a[u] + b[u]);
                            for (int off = g_; off < 64; off <<= 1) t += __shfl_xor(t, off);
                            if ((int)lane < g_) dsm[j * g_ + (int)lane] += t;
                        }
                    }
                }
            }
        } else {
        for (int ch = (int)lane; ch < c; ch += 64) {
            const int g = ch % g_;
            const float gch = go[(size_t)pt * c + ch];
            for (int j = 0; j < nsample; ++j) {
                const int nb = idx[(size_t)pt * nsample + j];
                const float w = gch * srow[j * g_ + g];
                d_pr[((size_t)pt * nsample + j) * c + ch] = w;
                atomicAdd(d_xv + (size_t)nb * c + ch, w);
            }
        }
        // d_sm: lanes over (j, g) pairs, each sums its share_planes channels
        for (int e = (int)lane; e < nsample * g_; e += 64) {
            const int j = e / g_, g = e - j * g_;
            const int nb = idx[(size_t)pt * nsample + j];
            float acc = 0.0f;
            for (int s = 0; s < s_; ++s) {
                const int ch = s * g_ + g;
                acc += go[(size_t)pt * c + ch] * (xv[(size_t)nb * c + ch] + pr[((size_t)pt * nsample + j) * c + ch]);
            }
            dsm[e] = acc;
        }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // dsm crosses lanes: see the forward kernel
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int g0 = (int)lane; g0 < g_; g0 += 64) {
            float dot = 0.0f;
            for (int j = 0; j < nsample; ++j) dot += srow[j * g_ + g0] * dsm[j * g_ + g0];
            for (int j = 0; j < nsample; ++j)
                d_logit[((size_t)pt * nsample + j) * g_ + g0] = srow[j * g_ + g0] * (dsm[j * g_ + g0] - dot);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace tgn

using namespace tgn;

TGN_API int tgn_pt_attention_forward(int n, int nsample, int c, int g, const float *p, const float *xq, const float *xk,
                                     const float *xv, const int *idx, const float *Wp1, const float *bp1, const float *Wp2,
                                     const float *bp2, const float *a1, const float *t1, const float *Ww1, const float *bw1,
                                     const float *Ww2, const float *bw2, const float *post_scale, const float *post_shift, float *out,
                                     tgn_stream_t stream) {
    if (n <= 0) return TGN_OK;
    if (!post_scale != !post_shift) {
        set_error("tgn_pt_attention_forward: post_scale and post_shift come together");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    if (!p || !xq || !xk || !xv || !idx || !out || !Wp1 || !bp1 || !Wp2 || !bp2 || !a1 || !t1 || !Ww1 || !bw1 || !Ww2 || !bw2) {
        set_error("tgn_pt_attention_forward: null pointer");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    if (nsample < 1 || nsample > 64 || c < 4 || (c & 3) || g < 4 || c % g || (((uintptr_t)xk | (uintptr_t)xv) & 15)) {
        set_error("tgn_pt_attention_forward: needs nsample <= 64, c %% 4 == 0, weight channels in {4,8,16,32,64} dividing c, "
                  "16-byte aligned x_k / x_v");
        return TGN_ERR_UNSUPPORTED;
    }
    const PtParams P{Wp1, bp1, Wp2, bp2, a1, t1, Ww1, bw1, Ww2, bw2, post_scale, post_shift};
    long long blocks = ((long long)n + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipStream_t st = (hipStream_t)stream;
#define TGN_PT(GG) hipLaunchKernelGGL((pt_attention_fwd_kernel<GG>), dim3((unsigned)blocks), dim3(256), 0, st, n, nsample, c, p, xq, xk, xv, idx, P, out)
    switch (g) {
        case 4: TGN_PT(4); break;
        case 8: TGN_PT(8); break;
        case 16: TGN_PT(16); break;
        case 32: TGN_PT(32); break;
        case 64: TGN_PT(64); break;
        default:
            set_error("tgn_pt_attention_forward: weight channels must be 4, 8, 16, 32 or 64 (got %d)", g);
            return TGN_ERR_UNSUPPORTED;
    }
#undef TGN_PT
    return check_launch("pt_attention_fwd_kernel");
}

TGN_API int tgn_pt_softmax_aggregate_forward(int n, int nsample, int c, int g, const float *xv, const float *pr,
                                             const float *logit, const int *idx, float *sm, float *out,
                                             tgn_stream_t stream) {
    if (n <= 0) return TGN_OK;
    if (!xv || !pr || !logit || !idx || !sm || !out || nsample < 1 || g < 1 || c % g) {
        set_error("tgn_pt_softmax_aggregate_forward: bad argument");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    const size_t lds = (size_t)4 * nsample * g * sizeof(float);
    if (lds > 64 * 1024) {
        set_error("tgn_pt_softmax_aggregate_forward: nsample * weight channels too large");
        return TGN_ERR_UNSUPPORTED;
    }
    long long blocks = ((long long)n + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(pt_softmax_aggregate_fwd_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, n, nsample,
                       c, g, xv, pr, logit, idx, sm, out);
    return check_launch("pt_softmax_aggregate_fwd_kernel");
}

TGN_API int tgn_pt_softmax_aggregate_backward(int n, int nsample, int c, int g, const float *xv, const float *pr,
                                              const float *sm, const int *idx, const float *grad_out, float *grad_xv,
                                              float *grad_pr, float *grad_logit, tgn_stream_t stream) {
    if (n <= 0) return TGN_OK;
    if (!xv || !pr || !sm || !idx || !grad_out || !grad_xv || !grad_pr || !grad_logit || nsample < 1 || g < 1 || c % g) {
        set_error("tgn_pt_softmax_aggregate_backward: bad argument");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    const size_t lds = (size_t)4 * nsample * g * sizeof(float);
    if (lds > 64 * 1024) {
        set_error("tgn_pt_softmax_aggregate_backward: nsample * weight channels too large");
        return TGN_ERR_UNSUPPORTED;
    }
    long long blocks = ((long long)n + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(pt_softmax_aggregate_bwd_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, n, nsample, c,
                       g, xv, pr, sm, idx, grad_out, grad_xv, grad_pr, grad_logit);
    return check_launch("pt_softmax_aggregate_bwd_kernel");
}

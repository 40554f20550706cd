// fps_common.h -- pieces shared by the FPS kernels (fps.hip: register-resident / streaming; fps_bucket.hip:
// register-resident with exact bucket skipping).
#pragma once
#include "tgn_common.h"

namespace tgn {

struct FpsArgs {
    const float *xyz;
    const int *offset;      // nullptr => dense batch: cloud i = [i*n_uniform, (i+1)*n_uniform)
    const int *new_offset;
    int n_uniform, m_uniform;
    void *idx;
    float *new_xyz;         // optional (m,3)
    float *tmp;             // only used by the streaming kernel
    void *ws;               // workspace of the large-cloud bucket kernel (tgn_fps_workspace_bytes)
    size_t ws_bytes;
    int n_max;              // largest cloud of the batch (workspace stride)
    int flags;
    int ref_log2_block;     // log2 of the reference's block size (cuda-compat tie order)
};

__device__ __forceinline__ void fps_segment(const FpsArgs &a, int bid, int &start_n, int &n, int &start_m, int &m) {
    if (a.offset) {
        start_n = bid ? a.offset[bid - 1] : 0;
        n = a.offset[bid] - start_n;
        start_m = bid ? a.new_offset[bid - 1] : 0;
        m = a.new_offset[bid] - start_m;
    } else {
        start_n = bid * a.n_uniform;
        n = a.n_uniform;
        start_m = bid * a.m_uniform;
        m = a.m_uniform;
    }
}

__device__ __forceinline__ void fps_emit(const FpsArgs &a, int row, int start_n, int k_local, float x, float y,
                                         float z) {
    long long v = (a.flags & TGN_FPS_LOCAL_INDEX) ? (long long)k_local : (long long)start_n + k_local;
    if (a.flags & TGN_FPS_INDEX64)
        ((long long *)a.idx)[row] = v;
    else
        ((int *)a.idx)[row] = (int)v;
    if (a.new_xyz) {
        a.new_xyz[(size_t)row * 3 + 0] = x;
        a.new_xyz[(size_t)row * 3 + 1] = y;
        a.new_xyz[(size_t)row * 3 + 2] = z;
    }
}

// cuda-compat tie order: (bit-reversed reference thread id, position within that thread).
__device__ __forceinline__ unsigned compat_key(int k, int log2bs) {
    unsigned t = (unsigned)k & ((1u << log2bs) - 1u);
    unsigned r = log2bs ? (__brev(t) >> (32 - log2bs)) : 0u;
    return (r << 21) | ((unsigned)k >> log2bs);
}
__device__ __forceinline__ int compat_index(unsigned key, int log2bs) {
    unsigned r = key >> 21;
    unsigned t = log2bs ? (__brev(r) >> (32 - log2bs)) : 0u;
    return (int)(((key & 0x1FFFFFu) << log2bs) | t);
}

__device__ __forceinline__ unsigned long long fps_pack(float best, unsigned key) {
    // best < 0 <=> this lane saw no real point: 0 loses against every real candidate
    return best < 0.0f ? 0ull : pack64(__float_as_uint(best), 0xFFFFFFFFu - key);
}

// Block-wide max of the packed keys; one barrier; result uniform in every wave.
template <int NW>
__device__ __forceinline__ unsigned long long fps_block_max(unsigned long long pk, unsigned long long (*slots)[NW],
                                                            int parity, int wave, int lane) {
    unsigned long long wmax = wave_max_u64(pk);
    if constexpr (NW == 1) {
        return wmax;
    } else {
        if (lane == 0) slots[parity][wave] = wmax;
        __syncthreads();
        unsigned long long v = lane < NW ? slots[parity][lane] : 0ull;
        return row0_max_u64(v);
    }
}

// fps_bucket.hip: launches the bucket-skipping kernel when a shape covers n_max; returns -1 if none does.
int fps_bucket_launch(int mode, int b, int n_max, const FpsArgs &a, hipStream_t stream);
// large clouds through a cell-sorted workspace; -1 if the workspace is missing / too small / cloud too large
int fps_bucket_stream_launch(int mode, int b, int n_max, const FpsArgs &a, hipStream_t stream);
size_t fps_stream_workspace_bytes(int b, int n_max);

}  // namespace tgn

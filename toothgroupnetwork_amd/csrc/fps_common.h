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
    // FPS of an FPS prefix is the identity (fps_prefix_* below): per-cloud certificates, both optional
    const int *prefix_in;   // prefix_in[cloud] >= m: the cloud is known to BE an FPS sequence -> samples are 0..m-1
    int *prefix_out;        // number of leading samples of THIS result that carry the property on
    const float *prefix_ref;  // optional: the coordinates the certificate was issued for (same layout as xyz); the
                              // shortcut is taken only if the cloud equals them bit for bit
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

// Farthest point sampling of a cloud that is itself an FPS sequence p_0, p_1, ... (same arithmetic, first-index ties)
// returns positions 0, 1, 2, ...: p_j attains the maximum of the running minimum distance over the whole original
// cloud, hence over the subset, and every earlier position is an already-picked sample at distance exactly 0.  That
// needs the winning distance d_j of iteration j to be > 0 (not exhausted: no duplicates picked) and < 1e10 (a point
// with NaN/Inf coordinates never leaves its initial 1e10 and is picked again and again).  The kernels record, per
// cloud, the first iteration that violates it (FpsPrefixCert); a later launch that is handed this certificate (and does not use
// the tree tie order) emits the identity without running; with prefix_ref the kernel first checks that the cloud
// really is the sequence the certificate was issued for.  vbits = bit pattern of d_j (>= 0: ordered like unsigned).
// The winning distance never increases from one iteration to the next (every running minimum only shrinks), so the
// property follows from two wave-uniform values: the largest d_j seen (= d_1; >= 1e10 means a NaN/Inf point or a clamped tie) and the
// last one (> 0: no iteration was exhausted).
// Kernels are instantiated with MODE bit 2 (kFpsModeCert) only when a certificate is wanted: even these few scalar
// instructions (and two more live SGPRs) cost the 24 000-point kernel 2 % per iteration.
constexpr int kFpsModeCert = 4;
// (Round 3: the count of iterations with d_j > 0 is not kept any more -- d_j never increases, so the LAST winning distance being
// positive says all of them were; a launch that ran into exhaustion claims nothing instead of its exact prefix length.  One scalar
// max per iteration is left.)
struct FpsPrefixCert {
    unsigned mx = 0u, last = 0u;
    __device__ __forceinline__ void update(unsigned vbits) {
        mx = vbits > mx ? vbits : mx;
        last = vbits;
    }
    __device__ __forceinline__ int value(int m) const {
        return (m <= 1 || (mx < 0x501502F9u /* 1e10f */ && last != 0u)) ? m : 1;
    }
};
template <int NT>
__device__ __forceinline__ bool fps_prefix_shortcut(const FpsArgs &a, int cloud, int start_n, int n, int start_m, int m) {
    if (!a.prefix_in || (a.flags & TGN_FPS_TREE_TIES)) return false;
    const int c = a.prefix_in[cloud];  // block-uniform
    if (c < m || m > n) return false;
    const float *__restrict__ base = a.xyz + (size_t)start_n * 3;
    if (a.prefix_ref) {
        // provenance by content: the caller only believes that xyz is the sequence the certificate belongs to
        const unsigned *__restrict__ u = (const unsigned *)base;
        const unsigned *__restrict__ r = (const unsigned *)(a.prefix_ref + (size_t)start_n * 3);
        int same = 1;
        for (int i = threadIdx.x; i < 3 * n; i += NT) same &= (u[i] == r[i]) ? 1 : 0;
        if (!__syncthreads_and(same)) return false;
    }
    for (int j = threadIdx.x; j < m; j += NT)
        fps_emit(a, start_m + j, start_n, j, base[(size_t)j * 3 + 0], base[(size_t)j * 3 + 1], base[(size_t)j * 3 + 2]);
    if (a.prefix_out && threadIdx.x == 0) a.prefix_out[cloud] = m;
    return true;
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
        slots[parity][wave] = wmax;   // wave-uniform: every lane stores it (no exec juggling on the way to the barrier)
        __syncthreads();
        unsigned long long v = lane < NW ? slots[parity][lane] : 0ull;
        return row0_max_u64(v);
    }
}

// ---- wave reductions written as v_*_dpp instructions (shared by all FPS kernels) ------------------------------
// 64-lane max / min of an fp32 value; wave-uniform result.  Written as six v_max_f32_dpp / v_min_f32_dpp
// (row_shr 1,2,4,8 then row_bcast 15,31): lanes without a DPP source are write-disabled and keep their value.
// hipcc expands the same reduction from builtins into 5 instructions per step (identity mov, dpp mov, two
// canonicalising v_max, v_max), a ~350-cycle dependent chain; this is ~60.  The s_nop 1 before each step is
// the VALU-write -> DPP-read hazard (2 wait states) that the assembler does not insert inside asm blocks.
#define TGN_DPP_REDUCE(OP)                                                                \
    asm volatile("s_nop 1\n\t" OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"   \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"   \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"   \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"   \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
                 "s_nop 1"                                                                    \
                 : "+v"(v))
__device__ __forceinline__ float wave_max_f32_dpp(float v) {
    TGN_DPP_REDUCE("v_max_f32_dpp");
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_min_f32_dpp(float v) {
    TGN_DPP_REDUCE("v_min_f32_dpp");
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ unsigned wave_min_u32_dpp(unsigned v) {
    TGN_DPP_REDUCE("v_min_u32_dpp");
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32_shfl(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)v, o);
        v = t < v ? t : v;
    }
    return v;
}


// Block-wide argmax of (value, tie key): value = max (values >= 0, or -1 for "none"), smallest tie key among equal
// values.  One LDS slot per wave, one barrier.  Returns the winning tie key (0xFFFFFFFF if no lane had a value).
// The common case (a single lane / a single wave holds the maximum) costs one 6-instruction DPP max, a ballot and a
// readlane per level; exact ties fall back to a key minimum.
// HIP's __ballot(int) compiles to select(0/1) + compare-not-zero around the lane mask the predicate already is
// (two to three extra instructions on a wave that issues one per ~5 cycles); the builtin takes the i1 directly, and
// lane-range restrictions are applied to the 64-bit result as constants.
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

template <int NW>
__device__ __forceinline__ unsigned fps_block_argmax(float best, unsigned key, unsigned long long (*slots)[NW],
                                                     int parity, int wave, int lane, unsigned &vbits) {
    static_assert(NW == 1 || NW == 4 || NW == 8 || NW == 16, "wave count");
    const float wm = wave_max_f32_dpp(best);
    const unsigned long long eq = wm >= 0.0f ? ballot64(best == wm) : 0ull;  // wm is wave-uniform
    unsigned wkey = 0xFFFFFFFFu;
    if (eq) {
        if (__popcll(eq) == 1)
            wkey = (unsigned)__builtin_amdgcn_readlane((int)key, __builtin_ctzll(eq));
        else
            wkey = wave_min_u32_dpp(((eq >> lane) & 1ull) ? key : 0xFFFFFFFFu);
    }
    if constexpr (NW == 1) {
        vbits = wm < 0.0f ? 0u : __float_as_uint(wm);
        return wkey;
    } else {
        // distances are >= 0: their bit patterns order like unsigned integers
        // (every lane stores the same word: `if (lane == 0)` would be an exec save / branch / restore on the way to the barrier)
        slots[parity][wave] = pack64(wm < 0.0f ? 0u : __float_as_uint(wm), wkey);
        __syncthreads();
        // every lane reads slot lane % NW (no exec juggling); the max over lanes 0..NW-1 lands in lane NW-1 after
        // log2(NW) row_shr steps (one asm statement: between statements the compiler adds wait states of its own)
        const unsigned long long v = slots[parity][lane & (NW - 1)];
        const unsigned vb = (unsigned)(v >> 32), vk = (unsigned)v;
        unsigned mb = vb;
        if constexpr (NW == 4)
            asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                         : "+v"(mb));
        else if constexpr (NW == 8)
            asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                         : "+v"(mb));
        else
            asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                         : "+v"(mb));
        mb = (unsigned)__builtin_amdgcn_readlane((int)mb, NW - 1);
        vbits = mb;
        const unsigned long long wmask = ballot64(vb == mb) & ((1ull << NW) - 1ull);
        if (__popcll(wmask) == 1) return (unsigned)__builtin_amdgcn_readlane((int)vk, __builtin_ctzll(wmask));
        return wave_min_u32_dpp(((wmask >> lane) & 1ull) ? vk : 0xFFFFFFFFu);
    }
}

// fps_bucket.hip: launches the bucket-skipping kernel when a shape covers n_max; returns -1 if none does.
int fps_bucket_launch(int mode, int b, int n_max, const FpsArgs &a, hipStream_t stream);
// large clouds through a cell-sorted workspace; -1 if the workspace is missing / too small / cloud too large
int fps_bucket_stream_launch(int mode, int b, int n_max, const FpsArgs &a, hipStream_t stream);
size_t fps_stream_workspace_bytes(int b, int n_max);
// TGN_FPS_THROUGHPUT: the owner-wave kernel with 8 waves and one metadata group for clouds of up to 32 768 points; -1 if the
// workspace is missing / too small or the cloud too large
int fps_bucket_owner_small_launch(int mode, int b, int n_max, const FpsArgs &a, hipStream_t stream);

}  // namespace tgn

// tgn_common.h -- shared device helpers and host-side error plumbing for libtgn_pointops.so.
// gfx950 (MI355X / CDNA4) only: 64-lane wavefronts, DPP row operations, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tgn_pointops.h"

#define TGN_API extern "C" __attribute__((visibility("default")))

#pragma clang fp contract(off)

namespace tgn {

constexpr int kWave = 64;

void set_error(const char *fmt, ...);
hipStream_t default_stream();
// error word of the gather family for launches on `stream` of the current device (capi.hip): one word per (device, stream), so
// that host threads driving different streams never clear or take each other's bits; nullptr if it cannot be allocated
int *index_error_word(hipStream_t stream);

// Kernel-variant switches (capi.hip; include/tgn_pointops.h: tgn_set_tuning).  One table of atomics, read with a relaxed load on
// the launch paths -- no getenv() there.  The legacy TGN_* environment names seed the table ONCE, when the library is loaded.
enum Tuning {
    kTuneFpsPlain = 0,      // "fps_plain": 1 = the plain register-resident / streaming FPS kernels, no bucket skipping
    kTuneFpsConfig,         // "fps_config": NT * 256 + P forces an instantiated plain-kernel shape (0 = pick)
    kTuneFpsBucketConfig,   // "fps_bucket_config": NT * 256 + P forces a bucket-kernel shape (0 = pick)
    kTuneFpsCellBits,       // "fps_cell_bits": 4 (12-bit cell codes, default) or 5 (round 1's 15-bit codes)
    kTuneFpsBucketMin,      // "fps_bucket_min": smallest cloud the bucket kernel takes (-1 = the built-in thresholds)
    kTuneBallBitmap,        // "ball_bitmap": 0 = the rank-select ball-query kernel
    kTuneKnnMemset,         // "knn_memset": 1 = clear the redo counter with hipMemsetAsync (reproduces the graph-replay fault)
    kTuneKnnGridScale,      // "knn_grid_scale": kNN grid cell size, per mille of the estimated k-neighbour radius (1000)
    kTuneSaTile,            // "sa_tile": 0 = pick, 128 / 256 = force the workgroup tile of tgn_sa_mlp2_max_bf16x3
    kTuneGatherV4,          // "gather_v4": gather-family variants, bit 0: forward kernels with 16-byte lanes; bit 1: backward kernels with
                            // 16-byte lanes; bit 2: subtraction / aggregation backward with dword lanes and owner-side sums (wins over bit 1)
    kTuneFpsLean,           // "fps_lean": 0 = fps_resident_kernel for small clouds too, 1 = fps_lean_kernel for 257 .. 2048 points, 2 = up to 4096
    kTuneCount
};
int tuning(Tuning t);

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return TGN_ERR_LAUNCH;
    }
    return TGN_OK;
}

// ---- wave-level primitives -----------------------------------------------------------------
// DPP control words (gfx9 family): row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_or_zero(unsigned v) {
    // lanes without a valid source (or outside ROW_MASK) receive 0
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}

__device__ __forceinline__ unsigned long long pack64(unsigned hi, unsigned lo) {
    return ((unsigned long long)hi << 32) | lo;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_max_step_u64(unsigned long long v) {
    unsigned lo = dpp_or_zero<CTRL, ROW_MASK>((unsigned)v);
    unsigned hi = dpp_or_zero<CTRL, ROW_MASK>((unsigned)(v >> 32));
    unsigned long long o = pack64(hi, lo);
    return o > v ? o : v;
}

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int lane) {
    unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
    unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
    return pack64(hi, lo);
}

// Max of an unsigned 64-bit key over the 64 lanes of a wave; result is wave-uniform (SGPR pair).
// 0 is the identity: keys are built so that every real candidate is > 0.
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    v = dpp_max_step_u64<0x111, 0xF>(v);  // row_shr:1
    v = dpp_max_step_u64<0x112, 0xF>(v);  // row_shr:2
    v = dpp_max_step_u64<0x114, 0xF>(v);  // row_shr:4
    v = dpp_max_step_u64<0x118, 0xF>(v);  // row_shr:8  -> lane 15 of each row holds the row max
    v = dpp_max_step_u64<0x142, 0xA>(v);  // row_bcast:15 into rows 1,3
    v = dpp_max_step_u64<0x143, 0xC>(v);  // row_bcast:31 into rows 2,3 -> lane 63 holds the wave max
    return readlane_u64(v, 63);
}

// Max over the first 16 lanes (one DPP row); result uniform.
__device__ __forceinline__ unsigned long long row0_max_u64(unsigned long long v) {
    v = dpp_max_step_u64<0x111, 0xF>(v);
    v = dpp_max_step_u64<0x112, 0xF>(v);
    v = dpp_max_step_u64<0x114, 0xF>(v);
    v = dpp_max_step_u64<0x118, 0xF>(v);
    return readlane_u64(v, 15);
}

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Number of set bits of `mask` strictly below this lane.
__device__ __forceinline__ int mbcnt(unsigned long long mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// ---- arithmetic contracts (DESIGN.md "Arithmetic contract") ----------------------------------
// hipcc defaults to -ffp-contract=fast, and HIP's __fmul_rn/__fadd_rn are plain operators that it
// happily fuses (verified in the ISA), so contraction is switched OFF for every translation unit
// (Makefile: -ffp-contract=off, plus the pragma at the top of this header).  Wherever the contract
// calls for a fused multiply-add it is written explicitly as __builtin_fmaf.
__device__ __forceinline__ float dist_direct_nofma(float dx, float dy, float dz) {
    return ((dx * dx) + (dy * dy)) + (dz * dz);
}
__device__ __forceinline__ float dist_direct_fma(float dx, float dy, float dz) {
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}
__device__ __forceinline__ float sumsq3(float x, float y, float z) { return ((x * x) + (y * y)) + (z * z); }
// square_distance (pointnet2_utils.py:20-41) as torch-CPU evaluates it; (x1,y1,z1,s1) = src row.
__device__ __forceinline__ float sqdist_expanded(float x1, float y1, float z1, float s1, float x2, float y2, float z2,
                                                 float s2) {
    const float dot = __builtin_fmaf(z1, z2, __builtin_fmaf(y1, y2, x1 * x2));
    return ((-2.0f * dot) + s1) + s2;
}
// IEEE minNum as ONE v_min_f32: the compiler otherwise prepends a canonicalising v_max to every
// loop-carried operand of fminf (seen in the ISA).  NaN operand -> the other operand, like CUDA min().
__device__ __forceinline__ float vmin_f32(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <typename T>
__device__ __forceinline__ T idx_load(const void *p, long long i, bool is64) {
    return is64 ? (T)((const long long *)p)[i] : (T)((const int *)p)[i];
}


// LDS words written by some lanes of a wave and read by OTHER lanes of the same wave: the hardware completes one wave's
// LDS operations in order, but the compiler's memory model does not know that -- a wavefront-scope release / acquire pair
// around a wave barrier (no instructions) keeps it from moving the reads above the writes.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace tgn

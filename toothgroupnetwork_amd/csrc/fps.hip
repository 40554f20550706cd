// fps.hip -- farthest point sampling for gfx950.
//
// Replaces pointops.furthestsampling (external_libs/pointops/functions/pointops.py:10-27 ->
// src/sampling/sampling_cuda_kernel.cu:14-171) and the dense adaptor farthest_point_sample
// (external_libs/pointnet2_utils/pointnet2_utils.py:64-98).
//
// Design (DESIGN.md "FPS"): one workgroup per cloud.  The cloud's coordinates and its running
// minimum distance live in VGPRs for the whole kernel (P points per lane, statically indexed),
// so an iteration touches no HBM at all: the reference re-streams 20 B per point per iteration.
// Each iteration is: P fused update/argmax steps per lane -> 64-lane DPP max on a packed
// (distance bits, ~tie key) 64-bit word -> one LDS slot per wave -> ONE barrier -> 16-lane DPP
// max -> the winner's coordinates are fetched with a wave-uniform load.
//
// The packed key makes the argmax a plain unsigned max: distances are non-negative floats, whose
// bit patterns order like unsigned integers; the low word is (0xFFFFFFFF - tie_key) so the
// smallest tie key wins among equal distances.
//   canonical mode : tie_key = point index (first index wins, torch-CPU semantics of
//                    pointnet2_utils.py:103-118) and d = ((dx*dx)+(dy*dy))+(dz*dz), no fusion.
//   TGN_FPS_TREE_TIES: tie_key orders points the way the reference's shared-memory tree does
//                    (sampling_cuda_kernel.cu:5-10,64-123: lower slot wins => bit-reversed thread id,
//                    then lowest index within a thread).
//   TGN_FPS_FMA    : d = fma(dz,dz,fma(dy,dy,dx*dx)), the contraction nvcc applies to :54.
//   Both together = "cuda-compat"; TREE_TIES alone = the reference source compiled without contraction,
//   which is what oracle/_ref is and what the GPU tests pin this kernel against.
#include "fps_common.h"

#include <atomic>
#include <stdlib.h>

#include <stdlib.h>
#include <string.h>

namespace tgn {

// ---------------------------------------------------------------------------------------------
// Register-resident kernel: n <= NT*P.
// ---------------------------------------------------------------------------------------------
template <int NT, int P, int MODE>
__global__ __launch_bounds__(NT) void fps_resident_kernel(FpsArgs a) {
    constexpr bool FMA = (MODE & 1) != 0, TREE = (MODE & 2) != 0, CERT = (MODE & kFpsModeCert) != 0;
    constexpr int NW = NT / kWave;
    // small clouds also keep a copy of the coordinates in LDS: the winner's coordinates are then one LDS read away
    // instead of a dependent global load (~300+ cycles of an iteration that is all latency)
    constexpr bool LDS_XYZ = NT * P <= 8192;
    __shared__ unsigned long long slots[2][NW];
    __shared__ float sxyz[LDS_XYZ ? NT * P * 3 : 1];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);  // wave-uniform (SGPR)
    int start_n, n, start_m, m;
    fps_segment(a, blockIdx.x, start_n, n, start_m, m);
    if (m <= 0) return;
    if (fps_prefix_shortcut<NT>(a, blockIdx.x, start_n, n, start_m, m)) return;
    const float *__restrict__ base = a.xyz + (size_t)start_n * 3;
    FpsPrefixCert cert;

    float x[P], y[P], z[P], d[P];
#pragma unroll
    for (int s = 0; s < P; ++s) {
        const int k = s * NT + tid;
        const bool valid = k < n;
        x[s] = valid ? base[(size_t)k * 3 + 0] : 0.0f;
        y[s] = valid ? base[(size_t)k * 3 + 1] : 0.0f;
        z[s] = valid ? base[(size_t)k * 3 + 2] : 0.0f;
        d[s] = valid ? 1e10f : -1.0f;  // pointops.py:22 ; padding can never win (real distances are >= 0)
        if constexpr (LDS_XYZ) {
            sxyz[k * 3 + 0] = x[s];
            sxyz[k * 3 + 1] = y[s];
            sxyz[k * 3 + 2] = z[s];
        }
    }
    if constexpr (LDS_XYZ) __syncthreads();

    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    if (n > 0) {
        qx = base[0];
        qy = base[1];
        qz = base[2];
    }
    if (tid == 0) fps_emit(a, start_m, start_n, 0, qx, qy, qz);  // sampling_cuda_kernel.cu:39

    for (int j = 1; j < m; ++j) {
        float best = -1.0f;
        unsigned bkey = 0;
#pragma unroll
        for (int s = 0; s < P; ++s) {
            const float dx = x[s] - qx, dy = y[s] - qy, dz = z[s] - qz;
            const float dd = FMA ? dist_direct_fma(dx, dy, dz) : dist_direct_nofma(dx, dy, dz);
            const float nd = vmin_f32(dd, d[s]);  // min(d, tmp[k]) sampling_cuda_kernel.cu:55
            d[s] = nd;
            if constexpr (TREE) {
                const unsigned key = compat_key(s * NT + tid, a.ref_log2_block);
                const bool take = nd > best || (nd == best && key < bkey);
                best = take ? nd : best;
                bkey = take ? key : bkey;
            } else {
                // ascending s == ascending point index within the lane; strict '>' keeps the first
                const bool take = nd > best;
                best = take ? nd : best;
                bkey = take ? (unsigned)s : bkey;
            }
        }
        if constexpr (!TREE) bkey = bkey * NT + tid;
        unsigned vbits;
        const unsigned key = fps_block_argmax<NW>(best, bkey, slots, j & 1, wave, lane, vbits);
        if constexpr (CERT) cert.update(vbits);
        int k = key == 0xFFFFFFFFu ? 0 : (TREE ? compat_index(key, a.ref_log2_block) : (int)key);
        k = __builtin_amdgcn_readfirstlane(k);
        if (n > 0) {
            if constexpr (LDS_XYZ) {
                qx = sxyz[k * 3 + 0];
                qy = sxyz[k * 3 + 1];
                qz = sxyz[k * 3 + 2];
            } else {
                qx = base[(size_t)k * 3 + 0];
                qy = base[(size_t)k * 3 + 1];
                qz = base[(size_t)k * 3 + 2];
            }
        }
        if (tid == 0) fps_emit(a, start_m + j, start_n, k, qx, qy, qz);
    }
    if (a.prefix_out && tid == 0) a.prefix_out[blockIdx.x] = CERT ? cert.value(m) : 1;  // not tracked: no claim
}

// ---------------------------------------------------------------------------------------------------------------------------
// Lean register-resident kernel for small clouds (round 6): the same operator, arithmetic and tie orders as fps_resident_kernel
// (first-index ties; the tree order costs it a key per point and two more instructions per selection step, and resolves ties
// between lanes / waves in rare wave-uniform branches), written for the two things that bound these launches in
// the phased HotPath schedule: a lone wave issues about one instruction per 5 cycles WHATEVER its kind, so an iteration costs
// what the wave that holds the winner ISSUES (fps_resident_kernel<64,16>: ~350 instructions per iteration); and the ball
// queries that run beside FPS levels 2-3 are occupancy-bound by LDS (6 KB per wave), so every KB this kernel holds is theirs.
//   * two points per instruction: the coordinates and running minima are fp32 PAIRS (v_pk_add_f32 / v_pk_mul_f32: each half an
//     ordinary IEEE fp32 operation, nothing fused unless TGN_FPS_FMA asks for it);
//   * the loop tracks the VALUE of the lane's maximum only (one v_max3_f32 per pair).  Points are dealt lane-major (lane t owns
//     points t P .. t P + P - 1), so "first index wins" = first lane, then first slot inside it: a ballot finds the lane, and every
//     lane looks up the first slot equal to its own maximum AND that slot's coordinates with a compare + four selects per slot --
//     once per iteration, not per point, and independent of the wave reduction, whose DPP wait states they fill (the reduction
//     is written with update_dpp builtins on the integer view of the non-negative distances so that the compiler schedules it);
//   * one 32-byte record per wave (value, key, x, y, z), one barrier, one LDS round trip: the winner's coordinates come with the
//     record -- no copy of the cloud in LDS (fps_resident_kernel keeps 12 B per point there: 48 KB at 4096 points, which cost
//     the level-1 ball query beside it a third of its waves, profiles/r06_fps_lean.txt);
//   * result rows parked in a 64-row LDS buffer that wave 0 flushes by itself (no block barrier).
// LDS: 1.5 KB whatever the cloud.  profiles/r06_fps_lean.txt has the timings.
// ---------------------------------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float vmax3_f32(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// one step of a wave reduction as v_mov_dpp(identity) + max: the compiler folds the pair into v_max_*_dpp (the identity as `old`
// is the pattern its DPP combiner knows), keeps the wait states the hazard needs and fills them with independent work
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_max_i32(int v) {
    const int t = __builtin_amdgcn_update_dpp((int)0x80000000, v, CTRL, ROW_MASK, 0xf, false);
    return t > v ? t : v;
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_max_u32(unsigned v) {
    const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
    return t > v ? t : v;
}

template <int NT, int P, int MODE>
__global__ __launch_bounds__(NT) void fps_lean_kernel(FpsArgs a) {
    static_assert(P % 2 == 0 && P >= 2, "points come in pairs");
    constexpr bool FMA = (MODE & 1) != 0, TREE = (MODE & 2) != 0, CERT = (MODE & kFpsModeCert) != 0;
    constexpr int NW = NT / kWave, H = P / 2;
    static_assert(NW == 1 || NW == 2 || NW == 4 || NW == 8 || NW == 16, "wave count");
    __shared__ uint4 rec[2][NW][2];     // {value bits, key, x, y} {z, -, -, -}
    __shared__ float4 outbuf[kWave];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    int start_n, n, start_m, m;
    fps_segment(a, blockIdx.x, start_n, n, start_m, m);
    if (m <= 0) return;
    if (fps_prefix_shortcut<NT>(a, blockIdx.x, start_n, n, start_m, m)) return;
    const float *__restrict__ base = a.xyz + (size_t)start_n * 3;
    FpsPrefixCert cert;

    f32x2 x[H], y[H], z[H], d[H];
#pragma unroll
    for (int i = 0; i < H; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = tid * P + 2 * i + u;
            const bool valid = k < n;
            x[i][u] = valid ? base[(size_t)k * 3 + 0] : 0.0f;
            y[i][u] = valid ? base[(size_t)k * 3 + 1] : 0.0f;
            z[i][u] = valid ? base[(size_t)k * 3 + 2] : 0.0f;
            d[i][u] = valid ? 1e10f : -1.0f;   // pointops.py:22 ; padding can never win (real distances are >= 0)
        }
    }
    // tree tie order (sampling_cuda_kernel.cu:5-10,64-123): the smallest compat_key wins among equal distances, not the smallest
    // index -- the keys of a lane's points are kept beside them (P more registers, this mode only)
    const int log2bs = a.ref_log2_block;
    unsigned tk[TREE ? P : 1];
    if constexpr (TREE) {
#pragma unroll
        for (int s = 0; s < P; ++s) tk[s] = compat_key(tid * P + s, log2bs);
    }
    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    if (n > 0) {
        qx = base[0];
        qy = base[1];
        qz = base[2];
    }
    if (tid == 0) outbuf[0] = make_float4(__int_as_float(0), qx, qy, qz);   // row 0: sampling_cuda_kernel.cu:39

    for (int j = 1; j < m; ++j) {
        const f32x2 q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
        float best = -1.0f;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const f32x2 dx = x[i] - q2x, dy = y[i] - q2y, dz = z[i] - q2z;
            f32x2 dd;
            if constexpr (FMA)
                dd = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
            else
                dd = ((dx * dx) + (dy * dy)) + (dz * dz);
            const float n0 = vmin_f32(dd[0], d[i][0]), n1 = vmin_f32(dd[1], d[i][1]);   // min(d, tmp[k]) sampling_cuda_kernel.cu:55
            d[i][0] = n0;
            d[i][1] = n1;
            best = vmax3_f32(best, n0, n1);
        }
        // every lane: the first slot that holds ITS maximum, and that point's coordinates (descending chain: the smallest slot is
        // written last).  `best` is -1 or a non-negative, non-NaN float: its bit pattern orders like a signed integer.  The six
        // reduction steps and the P - 1 selection steps are independent chains, written interleaved: the selects sit in the wait
        // states the DPP hazard asks for.
        int slot = P - 1;
        float cx = x[H - 1][1], cy = y[H - 1][1], cz = z[H - 1][1];
        unsigned ksel = 0xFFFFFFFFu;   // (tree order) the key of the selected slot
        if constexpr (TREE) ksel = d[H - 1][1] == best ? tk[P - 1] : 0xFFFFFFFFu;
        int wv = __float_as_int(best);
        auto select_step = [&](int s2) {
            bool c = d[s2 >> 1][s2 & 1] == best;
            if constexpr (TREE) {
                c = c && tk[s2] < ksel;
                ksel = c ? tk[s2] : ksel;
            }
            slot = c ? s2 : slot;
            cx = c ? x[s2 >> 1][s2 & 1] : cx;
            cy = c ? y[s2 >> 1][s2 & 1] : cy;
            cz = c ? z[s2 >> 1][s2 & 1] : cz;
        };
        constexpr int kSel = P - 1;                       // selection steps, dealt over the six reduction steps
        int s2 = P - 2;
#define TGN_LEAN_STEP(CTRL, MASK, STEP)                                                    \
        wv = dpp_max_i32<CTRL, MASK>(wv);                                                  \
        __builtin_amdgcn_sched_barrier(0);   /* (the scheduler otherwise moves the whole selection behind the reduction) */ \
        _Pragma("unroll") for (int t = (kSel * (STEP)) / 6; t < (kSel * ((STEP) + 1)) / 6; ++t) select_step(s2--);       \
        __builtin_amdgcn_sched_barrier(0);
        TGN_LEAN_STEP(0x111, 0xf, 0)   // row_shr:1
        TGN_LEAN_STEP(0x112, 0xf, 1)   // row_shr:2
        TGN_LEAN_STEP(0x114, 0xf, 2)   // row_shr:4
        TGN_LEAN_STEP(0x118, 0xf, 3)   // row_shr:8
        TGN_LEAN_STEP(0x142, 0xa, 4)   // row_bcast:15 into rows 1 and 3
        TGN_LEAN_STEP(0x143, 0xc, 5)   // row_bcast:31 into rows 2 and 3
#undef TGN_LEAN_STEP
        const int wmi = __builtin_amdgcn_readlane(wv, 63);
        // the lanes that hold the wave's maximum (none if the wave has no point at all); no branch on `eq` or on the sign: the
        // selection chain above would sink into it, behind the reduction
        const unsigned long long eq = ballot64(__float_as_int(best) == wmi) & (wmi >= 0 ? ~0ull : 0ull);
        int L = eq ? (int)__builtin_ctzll(eq) : 0;   // lane-major points: the first lane holds the smallest index
        if constexpr (TREE) {
            if (__builtin_expect(__popcll(eq) > 1, 0)) {   // (wave-uniform, rare) several lanes tie: the smallest tree key among them
                const unsigned kl = ((eq >> lane) & 1ull) ? ksel : 0xFFFFFFFFu;
                const unsigned kmin = wave_min_u32_dpp(kl);
                L = (int)__builtin_ctzll(ballot64(kl == kmin) | (1ull << 63));
            }
        }
        const int sL = __builtin_amdgcn_readlane(slot, L);
        const int ix = __builtin_amdgcn_readlane(__float_as_int(cx), L), iy = __builtin_amdgcn_readlane(__float_as_int(cy), L),
                  iz = __builtin_amdgcn_readlane(__float_as_int(cz), L);
        unsigned key = eq ? (unsigned)((wave * kWave + L) * P + sL) : 0xFFFFFFFFu;
        float kx = __int_as_float(eq ? ix : 0), ky = __int_as_float(eq ? iy : 0), kz = __int_as_float(eq ? iz : 0);
        unsigned vbits = wmi < 0 ? 0u : (unsigned)wmi;   // distances are >= 0: their bit patterns order like unsigned integers
        if constexpr (NW > 1) {
            // (every lane writes the same 20 bytes: no exec juggling before the barrier)
            rec[j & 1][wave][0] = make_uint4(vbits, key, __float_as_uint(kx), __float_as_uint(ky));
            rec[j & 1][wave][1].x = __float_as_uint(kz);
            __syncthreads();
            const uint4 r0 = rec[j & 1][lane & (NW - 1)][0];
            const unsigned r1 = rec[j & 1][lane & (NW - 1)][1].x;
            unsigned mb = r0.x;
            mb = dpp_max_u32<0x111>(mb);
            if constexpr (NW >= 4) mb = dpp_max_u32<0x112>(mb);
            if constexpr (NW >= 8) mb = dpp_max_u32<0x114>(mb);
            if constexpr (NW >= 16) mb = dpp_max_u32<0x118>(mb);
            mb = (unsigned)__builtin_amdgcn_readlane((int)mb, NW - 1);
            // waves own ascending index ranges: of several waves with the same maximum the first holds the smallest index
            const unsigned long long wmask = ballot64(r0.x == mb) & ((1ull << NW) - 1ull);
            int w = (int)__builtin_ctzll(wmask);
            if constexpr (TREE) {
                if (__builtin_expect(__popcll(wmask) > 1, 0)) {   // (rare) equal maxima in several waves: the smallest tree key
                    const unsigned kk = (((wmask >> lane) & 1ull) && r0.y != 0xFFFFFFFFu) ? compat_key((int)r0.y, log2bs) : 0xFFFFFFFFu;
                    const unsigned kmin = wave_min_u32_dpp(kk);
                    if (kmin != 0xFFFFFFFFu) w = (int)__builtin_ctzll(ballot64(kk == kmin) | (1ull << 63));
                }
            }
            key = (unsigned)__builtin_amdgcn_readlane((int)r0.y, w);
            kx = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)r0.z, w));
            ky = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)r0.w, w));
            kz = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)r1, w));
            vbits = mb;
        }
        if constexpr (CERT) cert.update(vbits);
        const int k = key == 0xFFFFFFFFu ? 0 : (int)key;   // (no candidate: an empty cloud; its rows are index 0, coordinates 0)
        qx = kx;
        qy = ky;
        qz = kz;
        if (wave == 0) outbuf[j & (kWave - 1)] = make_float4(__int_as_float(k), qx, qy, qz);   // every lane of wave 0: the same 16 bytes
        if ((j & (kWave - 1)) == kWave - 1 && wave == 0) {   // wave-uniform: wave 0 flushes a full chunk of 64 rows by itself
            wave_lds_fence();
            const float4 o = outbuf[lane];
            fps_emit(a, start_m + j - (kWave - 1) + lane, start_n, __float_as_int(o.x), o.y, o.z, o.w);
            wave_lds_fence();
        }
    }
    if (wave == 0 && ((m - 1) & (kWave - 1)) != kWave - 1) {   // rows of the last, partial chunk
        wave_lds_fence();
        const int cb = ((m - 1) / kWave) * kWave;
        if (cb + lane <= m - 1) {
            const float4 o = outbuf[lane];
            fps_emit(a, start_m + cb + lane, start_n, __float_as_int(o.x), o.y, o.z, o.w);
        }
    }
    if (a.prefix_out && tid == 0) a.prefix_out[blockIdx.x] = CERT ? cert.value(m) : 1;   // not tracked: no claim
}

// lean shapes: (threads, points per lane), ordered by capacity
#define TGN_FPS_LEAN_CONFIGS(X) X(64, 8) X(256, 4) X(256, 8) X(512, 8)

// ---------------------------------------------------------------------------------------------
// Streaming fallback for clouds larger than the register file of one CU: coordinates and the
// running minimum are re-read from memory (L2 / Infinity Cache resident) every iteration, as the
// reference does, but with the wave-reduced argmax and one barrier per iteration.
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(1024) void fps_streaming_kernel(FpsArgs a) {
    constexpr bool FMA = (MODE & 1) != 0, TREE = (MODE & 2) != 0, CERT = (MODE & kFpsModeCert) != 0;
    constexpr int NT = 1024, NW = NT / kWave;
    __shared__ unsigned long long slots[2][NW];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);  // wave-uniform (SGPR)
    int start_n, n, start_m, m;
    fps_segment(a, blockIdx.x, start_n, n, start_m, m);
    if (m <= 0) return;
    if (fps_prefix_shortcut<NT>(a, blockIdx.x, start_n, n, start_m, m)) return;
    const float *__restrict__ base = a.xyz + (size_t)start_n * 3;
    FpsPrefixCert cert;
    float *__restrict__ tmp = a.tmp + start_n;
    for (int k = tid; k < n; k += NT) tmp[k] = 1e10f;

    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    if (n > 0) {
        qx = base[0];
        qy = base[1];
        qz = base[2];
    }
    if (tid == 0) fps_emit(a, start_m, start_n, 0, qx, qy, qz);

    for (int j = 1; j < m; ++j) {
        float best = -1.0f;
        unsigned bkey = 0;
        for (int k = tid; k < n; k += NT) {
            const float dx = base[(size_t)k * 3 + 0] - qx, dy = base[(size_t)k * 3 + 1] - qy,
                        dz = base[(size_t)k * 3 + 2] - qz;
            const float dd = FMA ? dist_direct_fma(dx, dy, dz) : dist_direct_nofma(dx, dy, dz);
            const float nd = vmin_f32(dd, tmp[k]);
            tmp[k] = nd;
            const unsigned key = TREE ? compat_key(k, a.ref_log2_block) : (unsigned)k;
            const bool take = nd > best || (TREE && nd == best && key < bkey);
            best = take ? nd : best;
            bkey = take ? key : bkey;
        }
        unsigned vbits;
        const unsigned key = fps_block_argmax<NW>(best, bkey, slots, j & 1, wave, lane, vbits);
        if constexpr (CERT) cert.update(vbits);
        int k = key == 0xFFFFFFFFu ? 0 : (TREE ? compat_index(key, a.ref_log2_block) : (int)key);
        k = __builtin_amdgcn_readfirstlane(k);
        if (n > 0) {
            qx = base[(size_t)k * 3 + 0];
            qy = base[(size_t)k * 3 + 1];
            qz = base[(size_t)k * 3 + 2];
        }
        if (tid == 0) fps_emit(a, start_m + j, start_n, k, qx, qy, qz);
    }
    if (a.prefix_out && tid == 0) a.prefix_out[blockIdx.x] = CERT ? cert.value(m) : 1;  // not tracked: no claim
}

// ---------------------------------------------------------------------------------------------
// Host side: pick the (threads, points-per-lane) shape whose capacity covers the largest cloud.
// ---------------------------------------------------------------------------------------------
struct FpsConfig {
    int nt, p;
};

// Ordered by capacity; of two shapes with the same capacity the first is taken.  Fewer waves mean a cheaper hand-off:
// measured per iteration at 4096 points 256x16 0.71 us, 512x8 0.73, 1024x4 0.74; at 2048 points 256x8 0.55, 512x8 0.73.
#define TGN_FPS_CONFIGS(X) \
    X(64, 1) X(64, 2) X(64, 4) X(64, 8) X(64, 16) X(256, 8) X(256, 16) X(512, 8) \
    X(512, 16) X(512, 24) X(512, 32) X(1024, 24) X(512, 48) X(512, 56)

static const FpsConfig kConfigs[] = {
#define X(NT_, P_) {NT_, P_},
    TGN_FPS_CONFIGS(X)
#undef X
};
constexpr int kNumConfigs = sizeof(kConfigs) / sizeof(kConfigs[0]);

static int fps_capacity() {
    int c = 0;
    for (int i = 0; i < kNumConfigs; ++i) c = kConfigs[i].nt * kConfigs[i].p > c ? kConfigs[i].nt * kConfigs[i].p : c;
    return c;
}

static bool fps_pick(int n_max, FpsConfig &out) {
    // experiments: tgn_set_tuning("fps_config", NT * 256 + P) forces an instantiated shape when it is large enough
    if (const int forced = tuning(kTuneFpsConfig)) {
        const int nt = forced >> 8, p = forced & 255;
        for (int i = 0; i < kNumConfigs; ++i)
            if (kConfigs[i].nt == nt && kConfigs[i].p == p && nt * p >= n_max) {
                out = kConfigs[i];
                return true;
            }
    }
    int best = -1;
    for (int i = 0; i < kNumConfigs; ++i) {
        const int cap = kConfigs[i].nt * kConfigs[i].p;
        if (cap >= n_max && (best < 0 || cap < kConfigs[best].nt * kConfigs[best].p)) best = i;
    }
    if (best < 0) return false;
    out = kConfigs[best];
    return true;
}

template <int MODE>
static int fps_launch(int b, int n_max, const FpsArgs &a, hipStream_t stream) {
    FpsConfig cfg;
    if ((a.flags & TGN_FPS_THROUGHPUT) && a.ws && n_max > 4096) {   // several workgroups per CU out of the L2-resident workspace
        const int rc = fps_bucket_owner_small_launch(MODE, b, n_max, a, stream);
        if (rc >= 0) return rc;
    }
    {
        if (!tuning(kTuneFpsPlain)) {   // experiments: "fps_plain" forces the plain (no skipping) kernels
            const int rc = fps_bucket_launch(MODE, b, n_max, a, stream);
            if (rc >= 0) return rc;
        }
    }
    {
        // small clouds: the lean kernel ("fps_lean": 0 = off, 1 = clouds of 257 .. 2048 points, 2 = up to 4096)
        const int lean = tuning(kTuneFpsLean);
        if (lean && n_max > 256 && n_max <= (lean >= 2 ? 4096 : 2048) && !tuning(kTuneFpsConfig)) {
#define X(NT_, P_)                                                                                      \
    if (n_max <= NT_ * P_) {                                                                            \
        hipLaunchKernelGGL((fps_lean_kernel<NT_, P_, MODE>), dim3(b), dim3(NT_), 0, stream, a);       \
        return check_launch("fps_lean_kernel");                                                         \
    }
            TGN_FPS_LEAN_CONFIGS(X)
#undef X
        }
    }
    if (fps_pick(n_max, cfg)) {
#define X(NT_, P_)                                                                                      \
    if (cfg.nt == NT_ && cfg.p == P_) {                                                                 \
        hipLaunchKernelGGL((fps_resident_kernel<NT_, P_, MODE>), dim3(b), dim3(NT_), 0, stream, a);   \
        return check_launch("fps_resident_kernel");                                                     \
    }
        TGN_FPS_CONFIGS(X)
#undef X
    }
    {
        const int rc = fps_bucket_stream_launch(MODE, b, n_max, a, stream);
        if (rc >= 0) return rc;
    }
    if (!a.tmp) {
        set_error("tgn_furthestsampling: cloud of %d points exceeds the resident capacity (%d) and tmp is NULL",
                  n_max, fps_capacity());
        return TGN_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL((fps_streaming_kernel<MODE>), dim3(b), dim3(1024), 0, stream, a);
    return check_launch("fps_streaming_kernel");
}

static int ilog2_floor(int v) {
    int l = 0;
    while ((1 << (l + 1)) <= v) ++l;
    return l;
}

// cuda_utils.h:11-14 opt_n_threads: largest power of two <= n, clamped to [1,1024]
static int ref_log2_block(int n_max) {
    if (n_max < 1) return 0;
    int l = ilog2_floor(n_max);
    return l > 10 ? 10 : l;
}

static int fps_dispatch(int b, int n_max, FpsArgs a, hipStream_t stream) {
    if (b <= 0) return TGN_OK;
    if (!a.xyz || !a.idx) {
        set_error("tgn_furthestsampling: null xyz/idx");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    if (n_max < 0) {
        set_error("tgn_furthestsampling: negative n_max");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    a.ref_log2_block = ref_log2_block(n_max);
    const bool tree = (a.flags & TGN_FPS_TREE_TIES) != 0, fma = (a.flags & TGN_FPS_FMA) != 0;
    if (tree) return fma ? fps_launch<3>(b, n_max, a, stream) : fps_launch<2>(b, n_max, a, stream);
    if (a.prefix_out)  // certificate wanted: the tracking variants (the tree tie order never carries the property)
        return fma ? fps_launch<1 | kFpsModeCert>(b, n_max, a, stream) : fps_launch<kFpsModeCert>(b, n_max, a, stream);
    return fma ? fps_launch<1>(b, n_max, a, stream) : fps_launch<0>(b, n_max, a, stream);
}

}  // namespace tgn

using namespace tgn;

TGN_API int tgn_fps_resident_capacity(void) { return fps_capacity(); }

TGN_API int tgn_furthestsampling(int b, int n_max, const float *xyz, const int *offset, const int *new_offset,
                                 float *tmp, void *idx, float *new_xyz, int flags, tgn_stream_t stream) {
    if (b > 0 && (!offset || !new_offset)) {
        set_error("tgn_furthestsampling: null offsets");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    FpsArgs a{xyz, offset, new_offset, 0, 0, idx, new_xyz, tmp, nullptr, 0, n_max, flags, 0};
    return fps_dispatch(b, n_max, a, (hipStream_t)stream);
}

TGN_API size_t tgn_fps_throughput_workspace_bytes(int b, int n_max) {
    if (n_max <= 4096 || n_max > 32768) return 0;
    return fps_stream_workspace_bytes(b, n_max);
}

TGN_API size_t tgn_fps_workspace_bytes(int b, int n_max) {
    if (n_max <= fps_capacity()) return 0;
    return fps_stream_workspace_bytes(b, n_max);
}

TGN_API int tgn_furthestsampling_ws(int b, int n_max, const float *xyz, const int *offset, const int *new_offset,
                                    void *workspace, size_t workspace_bytes, void *idx, float *new_xyz, int flags,
                                    tgn_stream_t stream) {
    if (b > 0 && (!offset || !new_offset)) {
        set_error("tgn_furthestsampling_ws: null offsets");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    // clouds beyond both the register capacity and the workspace kernel (262 144 points) stream through `workspace`
    // as the reference's tmp array if it is large enough for that (4 B per point of the whole batch)
    FpsArgs a{xyz, offset, new_offset, 0, 0, idx, new_xyz, (float *)workspace, workspace, workspace_bytes, n_max, flags, 0};
    return fps_dispatch(b, n_max, a, (hipStream_t)stream);
}

TGN_API int tgn_furthestsampling_dense(int B, int N, int S, const float *xyz, float *tmp, void *idx, float *new_xyz,
                                       int flags, tgn_stream_t stream) {
    if (N < 0 || S < 0) {
        set_error("tgn_furthestsampling_dense: negative size");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    FpsArgs a{xyz, nullptr, nullptr, N, S, idx, new_xyz, tmp, nullptr, 0, N, flags, 0};
    return fps_dispatch(B, N, a, (hipStream_t)stream);
}

TGN_API int tgn_furthestsampling_dense_ws(int B, int N, int S, const float *xyz, void *workspace, size_t workspace_bytes,
                                          void *idx, float *new_xyz, int flags, tgn_stream_t stream) {
    if (N < 0 || S < 0) {
        set_error("tgn_furthestsampling_dense_ws: negative size");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    FpsArgs a{xyz, nullptr, nullptr, N, S, idx, new_xyz, (float *)workspace, workspace, workspace_bytes, N, flags, 0};
    return fps_dispatch(B, N, a, (hipStream_t)stream);
}

// FPS with the prefix certificate (fps_common.h): prefix_in says which clouds ARE FPS sequences already, prefix_out
// receives the same statement about this call's result.  Either may be null.
TGN_API int tgn_furthestsampling_prefix(int b, int n_max, const float *xyz, const int *offset, const int *new_offset,
                                        void *workspace, size_t workspace_bytes, void *idx, float *new_xyz,
                                        const int *prefix_in, const float *prefix_ref, int *prefix_out, int flags,
                                        tgn_stream_t stream) {
    if (b > 0 && (!offset || !new_offset)) {
        set_error("tgn_furthestsampling_prefix: null offsets");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    FpsArgs a{xyz, offset, new_offset, 0, 0, idx, new_xyz, (float *)workspace, workspace, workspace_bytes, n_max, flags, 0,
              prefix_in, prefix_out, prefix_ref};
    return fps_dispatch(b, n_max, a, (hipStream_t)stream);
}

TGN_API int tgn_furthestsampling_dense_prefix(int B, int N, int S, const float *xyz, void *workspace,
                                              size_t workspace_bytes, void *idx, float *new_xyz, const int *prefix_in,
                                              const float *prefix_ref, int *prefix_out, int flags, tgn_stream_t stream) {
    if (N < 0 || S < 0) {
        set_error("tgn_furthestsampling_dense_prefix: negative size");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    FpsArgs a{xyz, nullptr, nullptr, N, S, idx, new_xyz, (float *)workspace, workspace, workspace_bytes, N, flags, 0,
              prefix_in, prefix_out, prefix_ref};
    return fps_dispatch(B, N, a, (hipStream_t)stream);
}

// Tie order / contraction of the reference-signature entry point (it has no flags argument): TGN_FPS_TREE_TIES and
// TGN_FPS_FMA bits, initialised from the environment (TGN_FPS_TIES=first|tree, TGN_FPS_FMA=0|1) and settable at run time.
static std::atomic<int> g_fps_legacy_flags{-1};

static int fps_legacy_flags() {
    int f = g_fps_legacy_flags.load(std::memory_order_relaxed);
    if (f < 0) {
        f = 0;
        const char *t = getenv("TGN_FPS_TIES");
        if (t && (t[0] == 't' || t[0] == 'T')) f |= TGN_FPS_TREE_TIES;
        const char *m = getenv("TGN_FPS_FMA");
        if (m && m[0] == '1') f |= TGN_FPS_FMA;
        g_fps_legacy_flags.store(f, std::memory_order_relaxed);
    }
    return f;
}

TGN_API void tgn_set_fps_mode(int flags) {
    g_fps_legacy_flags.store(flags & (TGN_FPS_TREE_TIES | TGN_FPS_FMA), std::memory_order_relaxed);
}
TGN_API int tgn_get_fps_mode(void) { return fps_legacy_flags(); }

// Reference ABI (sampling_cuda_kernel.h:13): global int32 indices, default stream; arithmetic per tgn_set_fps_mode
// (default: the canonical first-index tie order, unfused distance).
TGN_API void furthestsampling_cuda_launcher(int b, int n, const float *xyz, const int *offset, const int *new_offset,
                                            float *tmp, int *idx) {
    (void)tgn_furthestsampling(b, n, xyz, offset, new_offset, tmp, idx, nullptr, fps_legacy_flags(),
                               (tgn_stream_t)default_stream());
}

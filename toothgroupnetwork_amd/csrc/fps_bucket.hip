// fps_bucket.hip -- farthest point sampling with EXACT bucket skipping (gfx950).
//
// Same operator, same results as fps_resident_kernel (fps.hip): the cloud and its running minimum
// distances live in VGPRs of one workgroup for the whole kernel.  What changes is how much of the
// cloud an iteration touches.  FPS's update `tmp[k] = min(tmp[k], |p_k - q|^2)` is a no-op for every
// point that is farther from the new sample q than its current minimum distance, and after the first
// few hundred samples that is almost the whole cloud.  So:
//
//   * once per cloud the points are ordered along a Z-order curve (5+5+5-bit cell code | 15-bit index,
//     one in-LDS bitonic sort by the workgroup) and dealt out so that every (wave, register slot) pair
//     -- a "bucket" of 64 points, one per lane -- is a compact patch of the scan;
//   * lane s of a wave keeps bucket s's bounding box, its current largest minimum distance `bmax` and the
//     tie key of the point that attains it;
//   * per iteration the P metadata lanes of every wave evaluate, with the SAME operation order as the
//     point update, the distance L from q to each bucket's box.  Every fp32 operation involved (subtract,
//     square, add, and fma in FMA mode) is monotone, so L <= d_fp(p, q) holds EXACTLY for every point p in
//     the box, with no tolerance.  If L >= bmax then d_fp(p,q) >= tmp[p] for all its points: the update
//     cannot change anything and the bucket is skipped.  Only touched buckets are recomputed (8 VALU per
//     point) and get their (bmax, key) refreshed by a 64-lane DPP max.
//   * a wave's candidate is the max over its <= 64 bucket maxima (cached while the wave is untouched);
//     the block argmax is the packed 64-bit max of fps.hip.
//
// Results are bit-identical to the plain kernel and to the oracle by construction (skipped updates are
// provably no-ops); tests/test_gpu_parity.py runs every launch shape against the oracle, including
// lattices and duplicated vertices (exact ties) and NaN coordinates.
// Measured (profiles/): 24 000 -> 4096 drops from 2.3 us to 1.25 us per iteration (8.6 of 375 buckets touched on
// average; what remains is the latency of ~3 bucket updates on the busiest wave plus the block hand-off).
#include "fps_common.h"

#include <type_traits>

#include <stdlib.h>


#ifndef TGN_REFRESH_SKIP
#define TGN_REFRESH_SKIP 1
#endif
// Which wave holds Z-order bucket b = 8*s + i in its register slot s.  Plain round robin (i) sends buckets b, b+8, b+16,
// b+64 ... to the SAME wave, and those are exactly the index distances of spatial neighbours along a Z-order curve: the
// ~9 buckets one sample touches then pile up on a few waves and everyone else waits at the barrier (wave 0 spends 38 % of
// an iteration there).  XOR-folding the slot number into the wave number keeps every group of 8 consecutive buckets on
// 8 different waves and moves the power-of-two neighbours apart as well.
#ifndef TGN_FPS_PERM
#define TGN_FPS_PERM 1
#endif
// (applied where the sorted order is written -- tab[] is stored in (slot, wave) order -- so the loop's look-ups are unchanged)
template <int NW>
__device__ __forceinline__ unsigned fps_bucket_slot(unsigned pos) {   // sorted position -> position in (slot, wave, lane) order
    const unsigned b = pos >> 6, sl = b / NW, i = b & (NW - 1);
#if TGN_FPS_PERM == 1
    const unsigned w = i ^ ((sl ^ (sl >> 3)) & (NW - 1));
#elif TGN_FPS_PERM == 2
    const unsigned w = (i + 3u * sl + (sl >> 3)) & (NW - 1);
#else
    const unsigned w = i;
#endif
    return ((sl * NW + w) << 6) | (pos & 63u);
}

namespace tgn {

// deposit a wave-uniform value into one lane of a per-lane register (the metadata lane of a bucket)
// (one v_writelane_b32: a select would need a lane mask per slot -- 2 x 48 SGPRs, spilled to VGPR lanes)
#define TGN_WRITELANE_F32(reg, val_uniform, lane_const)                                              \
    do {                                                                                               \
        const int wl_bits_ = __builtin_amdgcn_readfirstlane(__float_as_int(val_uniform));              \
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(reg) : "s"(wl_bits_), "i"(lane_const));      \
    } while (0)
__device__ __forceinline__ unsigned spread5(unsigned v) {  // abcde -> a00b00c00d00e
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6) | ((v & 16u) << 8);
}

constexpr int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

template <int NT, int P, int MODE, bool DBG = false, int CB = 4>
__global__ __launch_bounds__(NT) void fps_bucket_kernel(FpsArgs a) {
    constexpr bool FMA = (MODE & 1) != 0, TREE = (MODE & 2) != 0, CERT = (MODE & kFpsModeCert) != 0;
    constexpr int NW = NT / kWave;
    constexpr int CAP = NT * P;
    static_assert(P <= kWave, "bucket metadata lives in lanes 0..P-1");
    static_assert(CAP <= 32768, "15-bit local indices");
    // Set-up: 32 768 Z-order cells, two 16-bit counters per word (a cloud has < 65 536 points); once the points are
    // placed the same 64 KiB hold the result staging buffer and the bucket arg-max planes.
    // CB = bits per axis of the Z-order cell code.  4 (default): 4096 cells, 8 KiB of counters; 5: 32 768 cells, 64 KiB.
    // The order only decides which bucket a point lands in, never a result, and 12 bits are as good as 15 for buckets of
    // 64 points (24 000 -> 4096: 3.76 vs 3.75 ms per 256 scans) -- but the workgroup then holds 63 KiB of LDS instead of
    // 116, which is what lets four waves of the row-piece grouping kernel (19 KiB each) run beside it.
    constexpr int kCellWords = (1 << (3 * CB)) / 2;
    constexpr int kAliasWords = NT * 4 + 4 * NW * P;
    constexpr int kLdsWords = kCellWords > kAliasWords ? kCellWords : kAliasWords;
    __shared__ unsigned cells[kLdsWords];
    __shared__ unsigned short tab[CAP];  // sorted position -> local point index (0xFFFF = padding)
    __shared__ float red[6][NW];
    __shared__ int wave_tot[NW];
    __shared__ float4 rec[2][NW][2];  // per wave: {value, tie key} and {x, y, z} of its candidate
    float4 *outbuf = (float4 *)cells;                                  // [NT]: results of the current chunk of NT iterations
    float (*bmeta)[NW][P] = (float (*)[NW][P])(cells + NT * 4);        // [4][NW][P]: per bucket x, y, z, tie key of its arg-max
    static_assert(kCellWords % NT == 0, "prefix scan: whole words per thread");

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);  // wave-uniform (SGPR)
    int start_n, n, start_m, m;
    fps_segment(a, blockIdx.x, start_n, n, start_m, m);
    if (m <= 0) return;
    if (fps_prefix_shortcut<NT>(a, blockIdx.x, start_n, n, start_m, m)) return;
    FpsPrefixCert cert;  // first iteration whose winning distance breaks the FPS-prefix property (fps_common.h)
    const float *__restrict__ base = a.xyz + (size_t)start_n * 3;
    const int log2bs = a.ref_log2_block;

    // ---- 1. bounding box of the cloud (finite values only; NaN is ignored by fmin/fmax) ----------------
    {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        #pragma unroll 4
        for (int k = tid; k < n; k += NT) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = base[(size_t)k * 3 + c];
                if (fabsf(v) <= 3.0e38f) {
                    lo[c] = fminf(lo[c], v);
                    hi[c] = fmaxf(hi[c], v);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float l = wave_min_f32_dpp(lo[c]), h = wave_max_f32_dpp(hi[c]);
            if (lane == 0) {
                red[c][wave] = l;
                red[3 + c][wave] = h;
            }
        }
    }
    __syncthreads();
    float glo[3], gscale[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float l = INFINITY, h = -INFINITY;
        for (int w = 0; w < NW; ++w) {
            l = fminf(l, red[c][w]);
            h = fmaxf(h, red[3 + c][w]);
        }
        const float ext = h - l;
        glo[c] = l;
        gscale[c] = (ext > 0.0f && ext < 3.0e38f) ? (float)(1 << CB) / ext : 0.0f;
    }

    // ---- 2. counting sort by 15-bit Z-order cell (LDS atomics; the order inside a cell is arbitrary -- it only decides
    //         which of two neighbouring buckets a point lands in, never a result) -------------------------------------
    auto cell_of = [&](int i) -> unsigned {
        unsigned cc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t = (base[(size_t)i * 3 + c] - glo[c]) * gscale[c];
            t = fminf(fmaxf(t, 0.0f), (float)((1 << CB) - 1));  // NaN -> 0
            cc[c] = (unsigned)(int)t;
        }
        return spread5(cc[0]) | (spread5(cc[1]) << 1) | (spread5(cc[2]) << 2);
    };
    for (int i = tid; i < kCellWords; i += NT) cells[i] = 0u;
    for (int i = n + tid; i < CAP; i += NT) tab[fps_bucket_slot<NW>((unsigned)i)] = (unsigned short)0xFFFF;
    __syncthreads();
    #pragma unroll 4
    for (int i = tid; i < n; i += NT) {
        const unsigned code = cell_of(i);
        atomicAdd(&cells[code >> 1], 1u << ((code & 1u) << 4));
    }
    __syncthreads();
    {   // exclusive prefix over the 32 768 counters: thread t owns words [t*W, (t+1)*W)
        constexpr int W = kCellWords / NT;
        unsigned sum = 0;
#pragma unroll 4
        for (int i = 0; i < W; ++i) {
            const unsigned w = cells[tid * W + i];
            sum += (w & 0xFFFFu) + (w >> 16);
        }
        unsigned incl = sum;  // inclusive scan over the wave, then over the waves
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const unsigned t = (unsigned)__shfl_up((int)incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == kWave - 1) wave_tot[wave] = (int)incl;
        __syncthreads();
        unsigned run = incl - sum;
        for (int w = 0; w < wave; ++w) run += (unsigned)wave_tot[w];
#pragma unroll 4
        for (int i = 0; i < W; ++i) {
            const unsigned w = cells[tid * W + i];
            const unsigned lo = w & 0xFFFFu, hi = w >> 16;
            cells[tid * W + i] = run | ((run + lo) << 16);  // start offsets (< 65 536 each)
            run += lo + hi;
        }
    }
    __syncthreads();
    #pragma unroll 4
    for (int i = tid; i < n; i += NT) {
        const unsigned code = cell_of(i);
        const unsigned sh = (code & 1u) << 4;
        const unsigned old = atomicAdd(&cells[code >> 1], 1u << sh);  // cursor of the cell; never carries into its neighbour
        tab[fps_bucket_slot<NW>((old >> sh) & 0xFFFFu)] = (unsigned short)i;
    }
    __syncthreads();

    // ---- 5. my P points.  Z-order bucket b (sorted positions [64b, 64b+64)) goes to wave b % NW, slot b / NW:
    //         neighbouring buckets -- the ones a new sample touches together -- sit in DIFFERENT waves, so
    //         their updates run concurrently instead of queueing on one wave. ---------------------------------
    float x[P], y[P], z[P], d[P];
#pragma unroll
    for (int s = 0; s < P; ++s) {
        const unsigned o = tab[(s * NW + wave) * kWave + lane];
        const bool valid = o != 0xFFFFu;
        x[s] = valid ? base[(size_t)o * 3 + 0] : 0.0f;
        y[s] = valid ? base[(size_t)o * 3 + 1] : 0.0f;
        z[s] = valid ? base[(size_t)o * 3 + 2] : 0.0f;
        d[s] = valid ? 1e10f : -1.0f;  // pointops.py:22 ; padding never wins (real distances are >= 0)
    }
    // ---- 6. bucket metadata.  Lane s of a wave keeps bucket s's box and largest min-distance in VGPRs (read
    //         every iteration); the arg-max point of the bucket (its coordinates and tie key) sits in LDS, written
    //         by the winning lane itself with one 16-B store and read only when the wave's candidate changes. ----
    float blo0 = INFINITY, blo1 = INFINITY, blo2 = INFINITY, bhi0 = -INFINITY, bhi1 = -INFINITY, bhi2 = -INFINITY;
    float bmax = -1.0f;
#pragma unroll
    for (int s = 0; s < P; ++s) {
        const bool valid = d[s] >= 0.0f;
        // NaN coordinates are left out of the box: such a point can never be updated anyway (min ignores NaN)
        const float l0 = wave_min_f32_dpp(valid ? x[s] : INFINITY), h0 = wave_max_f32_dpp(valid ? x[s] : -INFINITY);
        const float l1 = wave_min_f32_dpp(valid ? y[s] : INFINITY), h1 = wave_max_f32_dpp(valid ? y[s] : -INFINITY);
        const float l2 = wave_min_f32_dpp(valid ? z[s] : INFINITY), h2 = wave_max_f32_dpp(valid ? z[s] : -INFINITY);
        const float any = wave_max_f32_dpp(d[s]);
        TGN_WRITELANE_F32(blo0, l0, s);
        TGN_WRITELANE_F32(blo1, l1, s);
        TGN_WRITELANE_F32(blo2, l2, s);
        TGN_WRITELANE_F32(bhi0, h0, s);
        TGN_WRITELANE_F32(bhi1, h1, s);
        TGN_WRITELANE_F32(bhi2, h2, s);
        TGN_WRITELANE_F32(bmax, any, s);  // 1e10 if the bucket holds a real point, else -1
        // initial arg-max of the bucket: all real points sit at 1e10, the smallest tie key wins
        const unsigned o = tab[(s * NW + wave) * kWave + lane];
        const unsigned kl = valid ? (TREE ? compat_key((int)o, log2bs) : o) : 0xFFFFFFFFu;
        const unsigned kmin = wave_min_u32_dpp(kl);
        const bool win = valid && kl == kmin;
        if (win) {
            bmeta[0][wave][s] = x[s];
            bmeta[1][wave][s] = y[s];
            bmeta[2][wave][s] = z[s];
            bmeta[3][wave][s] = __uint_as_float(o);   // the ORIGINAL index in every mode: tie keys are derived from it where a tie is
                                                      // resolved (tree order: compat_key), nowhere else -- the common path of the tree
                                                      // mode is then the default mode's (round 5: 0.90 against 0.81 us per iteration)
        }
    }

    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    if (n > 0) {
        qx = base[0];
        qy = base[1];
        qz = base[2];
    }
    // Results are parked in LDS (one 16-B record per row, written by one lane) and flushed NT rows at a time,
    // coalesced: the per-iteration global store sequence of one lane was ~20 instructions on the critical wave.
    if (tid == 0) outbuf[0] = make_float4(__int_as_float(0), qx, qy, qz);  // row 0: sampling_cuda_kernel.cu:39

    // cached wave candidate (wave-uniform): value, tie key, coordinates
    float wm = -1.0f, wx = 0.0f, wy = 0.0f, wz = 0.0f;
    unsigned wkey = 0, pub_bits = 0u, pub_key = 0xFFFFFFFFu;
    int wslot = -1;     // slot of the bucket that currently is this wave's candidate
    // hand-off records: byte offsets of this wave's record (writer) and of record lane % NW (reader), parity 0
    constexpr unsigned kRecParity = NW * 32u;
    char *recb = (char *)rec;
    unsigned wr_off = (unsigned)wave * 32u, rd_off = (unsigned)(lane & (NW - 1)) * 32u;
    bool dirty = true;  // bucket maxima only shrink: the candidate changes only when ITS bucket's arg-max changes

    // optional instrumentation (flag 0x100 + tmp): touched-bucket census and per-phase cycles of one wave
    // compile-time switch: even never-taken `if (dbg)` branches cost a lone wave ~10 cycles each per iteration
    constexpr bool dbg = DBG;
    // The hand-off record is written by EVERY lane of the wave (same address, same data) in the 8-wave kernels: the
    // lane-0 form costs an exec save / branch / restore on the chain everybody waits for (3.39 -> 3.32 ms for 24 000 -> 4096;
    // the 4-wave kernels, which run beside the level-1 ball query, measured no better with it and keep lane 0)
    constexpr bool kRecAllLanes = NT >= 512;
    unsigned long long st_skip = 0, st_touched = 0, st_waves = 0, cyA = 0, cyU = 0, cyB = 0, cyC = 0, cyC1 = 0, cyC2 = 0;
    unsigned long long dbg_sum_max = 0, dbg_crit_is_winner = 0, dbg_crit_dirty = 0, dbg_crit_touched = 0, dbg_sum_winner = 0;
    unsigned dbg_prev_winner = 0;
    bool dbg_dirty = false;

    for (int j = 1; j < m; ++j) {
        long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        if (dbg) t0 = clock64();
        // ---- A. which of my buckets can the new sample change? (monotone lower bound, exact) -------------
        // (x, y) as packed fp32 pairs: a lone wave pays per instruction, not per lane-operation
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 qxy = {qx, qy};
        const f2 lo = f2{blo0, blo1} - qxy, hi = qxy - f2{bhi0, bhi1};
        const float ex = fmaxf(fmaxf(lo.x, hi.x), 0.0f);
        const float ey = fmaxf(fmaxf(lo.y, hi.y), 0.0f);
        const float ez = fmaxf(fmaxf(blo2 - qz, qz - bhi2), 0.0f);
        const float L = FMA ? dist_direct_fma(ex, ey, ez) : dist_direct_nofma(ex, ey, ez);
        // lane mask of the compare, restricted to the P metadata lanes by a constant (a ballot of `lane < P && ...`
        // makes the compiler rebuild the mask through a 0/1 select)
        constexpr unsigned long long kSlotMask = P >= 64 ? ~0ull : ((1ull << P) - 1ull);
        const unsigned long long mask = ballot64(!(L >= bmax)) & kSlotMask;
        if (dbg) {
            t1 = clock64();
            st_touched += __popcll(mask);
            st_waves += mask ? 1 : 0;
        }
        if (mask) {  // wave-uniform
            // A lone wave issues roughly one instruction per 5 cycles whatever its kind, so the mask is walked
            // hierarchically (groups of 8 slots, 32-bit tests: 2 scalar instructions per test) rather than bit by bit.
            const unsigned mlo = (unsigned)mask, mhi = (unsigned)(mask >> 32);
#pragma unroll
            for (int g = 0; g < (P + 7) / 8; ++g) {
                const unsigned mg = ((g < 4 ? mlo : mhi) >> ((g & 3) * 8)) & 0xFFu;
                if (__builtin_expect(mg != 0u, 0)) {  // wave-uniform; out of line: the chain of tests falls through (a taken
                                                      // branch costs a lone wave ~25 cycles, and 5 of 6 groups / 7 of 8 slots are clear)
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int s = g * 8 + u;  // compile-time after unrolling
                        if (s < P && __builtin_expect((mg & (1u << u)) != 0u, 0)) {  // wave-uniform
                            const float dx = x[s] - qx, dy = y[s] - qy, dz = z[s] - qz;
                            const float dd = FMA ? dist_direct_fma(dx, dy, dz) : dist_direct_nofma(dx, dy, dz);
                            // Distances only shrink: unless a point that HELD the bucket's maximum gets closer, the
                            // bucket's maximum and arg-max are unchanged and the 64-lane refresh is skipped.
                            const float bold = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bmax), s));
                            const bool unchanged = TGN_REFRESH_SKIP && ballot64(d[s] == bold && dd < d[s]) == 0;  // wave-uniform
                            const float nd = vmin_f32(dd, d[s]);  // min(d, tmp[k]) sampling_cuda_kernel.cu:55
                            d[s] = nd;
                            if (dbg && unchanged) ++st_skip;
                            if (!unchanged) {
                            // the lane's tie key (original index), wanted by the winner only: issued first so that the
                            // LDS round trip hides behind the reduction
                            const unsigned o = tab[(s * NW + wave) * kWave + lane];
                            if (s == wslot) dirty = true;
                            const float mx = wave_max_f32_dpp(nd);
                            const unsigned long long eq = ballot64(nd == mx);
                            bool win = nd == mx;
                            if (__builtin_expect(__popcll(eq) != 1, 0)) {  // exact tie inside the bucket (rare): the smallest tie key wins
                                const unsigned kl = win ? (TREE ? compat_key((int)o, log2bs) : o) : 0xFFFFFFFFu;
                                const unsigned kmin = __builtin_amdgcn_readfirstlane(wave_min_u32_shfl(kl));
                                win = win && kl == kmin;
                            }
                            if (win) {  // four 4-B stores into separate planes: one 16-B store would pin (x,y,z,key)
                                        // of every slot to an aligned register quadruple and double the footprint
                                bmeta[0][wave][s] = x[s];
                                bmeta[1][wave][s] = y[s];
                                bmeta[2][wave][s] = z[s];
                                bmeta[3][wave][s] = __uint_as_float(o);
                            }
                            {  // bmax[lane s] = mx: one v_writelane (a select would keep 48 hoisted lane masks alive in SGPRs)
                                const int mxs = __builtin_amdgcn_readfirstlane(__float_as_int(mx));
                                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(bmax) : "s"(mxs), "i"(s));
                            }
                            }
                        }
                    }
                }
            }
        }
        if (dbg) t2 = clock64();
        // ---- B. wave candidate = max over my bucket maxima (recomputed only if a bucket changed) ---------
        if (dbg) dbg_dirty = dirty;
        if (dirty) {
            // the arg-max records of ALL my buckets (coordinates, tie key) are requested BEFORE the reduction that says which
            // one is wanted: the LDS round trip then hides behind the six DPP steps instead of following them (this is the
            // winner's wave, the one everybody else is waiting for at the barrier)
            const int ml = lane < P ? lane : 0;
            const float4 pm = make_float4(bmeta[0][wave][ml], bmeta[1][wave][ml], bmeta[2][wave][ml], bmeta[3][wave][ml]);
            const float v = lane < P ? bmax : -1.0f;
            wm = wave_max_f32_dpp(v);
            const unsigned long long cm = wm >= 0.0f ? (ballot64(v == wm) & kSlotMask) : 0ull;
            int sl = cm ? __builtin_ctzll(cm) : 0;
            if (__builtin_expect(__popcll(cm) > 1, 0)) {
                const bool cand = ((cm >> lane) & 1ull) != 0ull;
                const unsigned kl = cand ? (TREE ? compat_key((int)__float_as_uint(pm.w), log2bs) : __float_as_uint(pm.w)) : 0xFFFFFFFFu;
                const unsigned kmin = __builtin_amdgcn_readfirstlane(wave_min_u32_shfl(kl));
                sl = __builtin_ctzll(ballot64(kl == kmin));
            }
            wkey = (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(pm.w), sl);
            wslot = sl;
            wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pm.x), sl));
            wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pm.y), sl));
            wz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pm.z), sl));
            pub_bits = wm < 0.0f ? 0u : __float_as_uint(wm);  // what the hand-off publishes (a wave without points: 0, ~0)
            pub_key = wm < 0.0f ? 0xFFFFFFFFu : wkey;
            dirty = false;
        }
        if (dbg) t3 = clock64();
        // ---- C. block argmax over the wave candidates: one LDS record per wave, ONE barrier ------------------
        unsigned kwin;
        if constexpr (NW == 1) {
            kwin = wm < 0.0f ? 0xFFFFFFFFu : wkey;
            if constexpr (CERT) cert.update(wm < 0.0f ? 0u : __float_as_uint(wm));
            qx = wx;
            qy = wy;
            qz = wz;
        } else {
            if (kRecAllLanes || lane == 0) {  // {value bits, tie key} and {x, y, z}: 8 + 12 bytes of the wave's 32-byte record
                *(uint2 *)(recb + wr_off) = make_uint2(pub_bits, pub_key);
                *(float3 *)(recb + wr_off + 16) = make_float3(wx, wy, wz);
            }
            long long tc1 = 0, tc2 = 0;
            if (dbg) {
                tc1 = clock64();
                // (instrumented build) my pre-barrier time, my touched / refreshed bucket counts and whether I searched for a
                // new candidate, into the spare bytes of my record
                if (lane == 0)
                    *(uint2 *)(recb + wr_off + 8) = make_uint2((unsigned)(tc1 - t0), (unsigned)__popcll(mask) | (dbg_dirty ? 0x100u : 0u));
            }
            __syncthreads();
            if (dbg) {
                tc2 = clock64();
                cyC1 += tc1 - t3;
                cyC2 += tc2 - tc1;
            }
            // distances are >= 0: their bit patterns order like unsigned integers
            // every lane reads record lane % NW (no exec juggling); lanes 0..NW-1 are the ones that count
            const uint2 r0 = *(const uint2 *)(recb + rd_off);
            const float3 r1 = *(const float3 *)(recb + rd_off + 16);  // same LDS round trip
            if (dbg && wave == 0) {   // who was the slowest wave of this iteration, and was it last iteration's winner?
                const uint2 dd_ = *(const uint2 *)(recb + rd_off + 8);
                unsigned tmax = 0, wmax_ = 0, info = 0;
                for (int w = 0; w < NW; ++w) {
                    const unsigned tw = (unsigned)__builtin_amdgcn_readlane((int)dd_.x, w);
                    const unsigned iw = (unsigned)__builtin_amdgcn_readlane((int)dd_.y, w);
                    if (tw > tmax) {
                        tmax = tw;
                        wmax_ = (unsigned)w;
                        info = iw;
                    }
                }
                dbg_sum_max += tmax;
                dbg_crit_is_winner += (wmax_ == dbg_prev_winner) ? 1 : 0;
                dbg_crit_dirty += (info >> 8) & 1u;
                dbg_crit_touched += info & 0xFFu;
                const unsigned tw_ = (unsigned)__builtin_amdgcn_readlane((int)dd_.x, (int)dbg_prev_winner);
                dbg_sum_winner += tw_;
            }
            rd_off ^= kRecParity;  // double-buffered by iteration parity: one barrier per iteration is enough
            wr_off ^= kRecParity;
            const unsigned vb = r0.x;
            unsigned mb = vb;
            // max over lanes 0..NW-1 lands in lane NW-1 after log2(NW) row_shr steps
            // (one asm statement per wave count: between statements the compiler adds wait states of its own)
            static_assert(NW == 4 || NW == 8, "wave count");
            if constexpr (NW == 4)
                asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                             : "+v"(mb));
            else
                asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                             : "+v"(mb));
            mb = (unsigned)__builtin_amdgcn_readlane((int)mb, NW - 1);
            if constexpr (CERT) cert.update(mb);
            const unsigned long long wmask = ballot64(vb == mb) & ((1ull << NW) - 1ull);  // lanes 0..NW-1 hold the records
            int wl = __builtin_ctzll(wmask);
            if (__builtin_expect(__popcll(wmask) > 1, 0)) {  // equal maxima in several waves (rare): the smallest tie key wins
                const unsigned kk = ((wmask >> lane) & 1ull) ? (TREE ? compat_key((int)r0.y, log2bs) : r0.y) : 0xFFFFFFFFu;
                const unsigned kmin = __builtin_amdgcn_readfirstlane(wave_min_u32_shfl(kk));
                wl = __builtin_ctzll(ballot64(kk == kmin));
            }
            if (dbg) dbg_prev_winner = (unsigned)wl;
            kwin = (unsigned)__builtin_amdgcn_readlane((int)r0.y, wl);
            qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1.x), wl));
            qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1.y), wl));
            qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1.z), wl));
        }
        int k = kwin == 0xFFFFFFFFu ? 0 : (int)kwin;      // (records carry original indices in every mode)
        k = __builtin_amdgcn_readfirstlane(k);
        if (tid == 0) outbuf[j & (NT - 1)] = make_float4(__int_as_float(k), qx, qy, qz);
        if ((j & (NT - 1)) == NT - 1) {  // wave-uniform
            __syncthreads();
            const float4 o = outbuf[tid];
            fps_emit(a, start_m + j - (NT - 1) + tid, start_n, __float_as_int(o.x), o.y, o.z, o.w);
            __syncthreads();
        }
        if (dbg) {
            const long long t4 = clock64();
            cyA += t1 - t0;
            cyU += t2 - t1;
            cyB += t3 - t2;
            cyC += t4 - t3;
        }
    }
    if (((m - 1) & (NT - 1)) != NT - 1) {  // rows of the last, partial chunk
        __syncthreads();
        const int cb = ((m - 1) / NT) * NT;
        if (cb + tid <= m - 1) {
            const float4 o = outbuf[tid];
            fps_emit(a, start_m + cb + tid, start_n, __float_as_int(o.x), o.y, o.z, o.w);
        }
    }
    if (a.prefix_out && tid == 0) a.prefix_out[blockIdx.x] = CERT ? cert.value(m) : 1;  // not tracked: no claim
    if (dbg && lane == 0) {
        unsigned long long *st = (unsigned long long *)a.tmp;
        atomicAdd(&st[0], st_touched);
        atomicAdd(&st[1], st_waves);
        if (wave == 0 && blockIdx.x == 0) {
            st[2] = cyA;
            st[3] = cyU;
            st[4] = cyB;
            st[5] = cyC;
            st[6] = (unsigned long long)(m - 1);
            st[7] = cyC1;
            st[8] = cyC2;
            st[10] = dbg_sum_max;
            st[11] = dbg_crit_is_winner;
            st[12] = dbg_crit_dirty;
            st[13] = dbg_crit_touched;
            st[14] = dbg_sum_winner;
        }
        atomicAdd(&st[9], st_skip);
        if (false) {
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Large clouds (raw scans of 10^5 points, preprocess_data.py:55-56 resamples N_raw -> 24 000): the same exact bucket
// skipping with the points in a cell-sorted, L2-resident workspace instead of registers.  (The streaming fallback of
// fps.hip re-reads 20 B per point per iteration: 2 MB at 10^5 points, 18-36 us per iteration measured.)
// ---------------------------------------------------------------------------------------------------------------
constexpr int kStreamThreads = 1024;
constexpr int kStreamMaxBuckets = 4096;           // 262 144 points

__host__ __device__ inline size_t fps_stream_cloud_bytes(int n_max) {
    const size_t npad = ((size_t)n_max + 63) / 64 * 64;
    return npad * (sizeof(float4) + sizeof(int));
}

// Every wave OWNS its buckets, as in the register-resident kernel: the bucket boxes, maxima and arg-max points (coordinates,
// tie key) of a wave live in its own VGPR lanes (G groups of 64 buckets), only the 64 points of a bucket stay in the
// workspace.  An iteration is: G box tests on the metadata lanes, the records of the few touched buckets requested from L2
// (the first two of a group in flight together), update + refresh in registers, the wave's candidate from registers, ONE
// hand-off record per wave and ONE barrier.  (Round 2's form kept the bucket planes in LDS, collected the touched buckets
// in an LDS list for the waves to share, and read the winner's coordinates back from memory: three barriers and a
// dependent global read per iteration, 1.97 us per iteration at 108 000 points against 1.06 now.)
//   workspace per cloud: rec[NBpad] float4 (x, y, z, original index as bits; read-only after set-up) +
//                        dist[NBpad] float (running minimum, one 256-B line per bucket)
//   bucket b (sorted positions [64b, 64b+64)) -> slot b / NW of wave (b % NW) ^ fold(slot): Z-order neighbours (index
//   distances 1, 2, 4 ... and NW, 2 NW, ...) sit in different waves and are updated concurrently.
// ---------------------------------------------------------------------------------------------------------------
template <int NW>
__device__ __forceinline__ int fps_owner_bucket(int wave, int slot) {
    return slot * NW + (wave ^ ((slot ^ (slot >> 4)) & (NW - 1)));
}

template <int MODE, int G, int NT = kStreamThreads, int CB = 5>
__global__ __launch_bounds__(NT) void fps_bucket_owner_kernel(FpsArgs a) {
    constexpr bool FMA = (MODE & 1) != 0, TREE = (MODE & 2) != 0, CERT = (MODE & kFpsModeCert) != 0;
    constexpr int NW = NT / kWave, kCells = 1 << (3 * CB);
    static_assert((NW == 16 || NW == 8) && G >= 1 && G <= 4 && kCells % NT == 0 && kCells >= 4 * NT, "NW waves x G x 64 buckets");
    __shared__ unsigned lds[kCells];         // (128 KiB at CB = 5) cell histogram during set-up, then the parked result rows
    __shared__ float red[6][NW];
    __shared__ int wave_tot[NW];
    __shared__ float4 hand[2][NW][2];        // per wave: {value bits, tie key} and {x, y, z} of its candidate
    float4 *outbuf = (float4 *)lds;          // [NT]

    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    int start_n, n, start_m, m;
    fps_segment(a, blockIdx.x, start_n, n, start_m, m);
    if (m <= 0) return;
    if (fps_prefix_shortcut<NT>(a, blockIdx.x, start_n, n, start_m, m)) return;
    FpsPrefixCert cert;
    const float *__restrict__ base = a.xyz + (size_t)start_n * 3;
    const int log2bs = a.ref_log2_block;
    unsigned char *wsb = (unsigned char *)a.ws + (size_t)blockIdx.x * fps_stream_cloud_bytes(a.n_max);
    const int NB = (n + 63) / 64;
    const int npad = NB * 64;
    float4 *__restrict__ rec = (float4 *)wsb;
    float *__restrict__ dist = (float *)(wsb + (((size_t)a.n_max + 63) / 64 * 64) * sizeof(float4));

    // ---- set-up 1: bounding box -------------------------------------------------------------------------------
    {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll 4
        for (int k = tid; k < n; k += NT) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = base[(size_t)k * 3 + c];
                if (fabsf(v) <= 3.0e38f) {
                    lo[c] = fminf(lo[c], v);
                    hi[c] = fmaxf(hi[c], v);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float l = wave_min_f32_dpp(lo[c]), h = wave_max_f32_dpp(hi[c]);
            if (lane == 0) {
                red[c][wave] = l;
                red[3 + c][wave] = h;
            }
        }
    }
    for (int i = tid; i < kCells; i += NT) lds[i] = 0;
    __syncthreads();
    float glo[3], gscale[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float l = INFINITY, h = -INFINITY;
        for (int w = 0; w < NW; ++w) {
            l = fminf(l, red[c][w]);
            h = fmaxf(h, red[3 + c][w]);
        }
        const float ext = h - l;
        glo[c] = l;
        gscale[c] = (ext > 0.0f && ext < 3.0e38f) ? (float)(1 << CB) / ext : 0.0f;
    }
    auto cell_of = [&](int i) -> unsigned {
        unsigned cc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t = (base[(size_t)i * 3 + c] - glo[c]) * gscale[c];
            t = fminf(fmaxf(t, 0.0f), (float)((1 << CB) - 1));  // NaN -> 0
            cc[c] = (unsigned)(int)t;
        }
        return spread5(cc[0]) | (spread5(cc[1]) << 1) | (spread5(cc[2]) << 2);
    };
    // ---- set-up 2: counting sort by Z-order cell into the workspace ---------------------------------------------
#pragma unroll 4
    for (int i = tid; i < n; i += NT) atomicAdd(&lds[cell_of(i)], 1u);
    __syncthreads();
    {
        constexpr int PER = kCells / NT;  // cells per thread
        unsigned local[PER];
        unsigned sum = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            local[i] = sum;
            sum += lds[tid * PER + i];
        }
        unsigned incl = sum;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const unsigned v = (unsigned)__shfl_up((int)incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == kWave - 1) wave_tot[wave] = (int)incl;
        __syncthreads();
        unsigned wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += (unsigned)wave_tot[w];
        const unsigned tbase = wbase + incl - sum;
#pragma unroll
        for (int i = 0; i < PER; ++i) lds[tid * PER + i] = tbase + local[i];  // running insert position
    }
    __syncthreads();
#pragma unroll 4
    for (int i = tid; i < n; i += NT) {
        const unsigned pos = atomicAdd(&lds[cell_of(i)], 1u);
        rec[pos] = make_float4(base[(size_t)i * 3 + 0], base[(size_t)i * 3 + 1], base[(size_t)i * 3 + 2], __int_as_float(i));
        dist[pos] = 1e10f;  // pointops.py:22
    }
    for (int i = n + tid; i < npad; i += NT) {
        rec[i] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(0x7FFFFFFF));
        dist[i] = -1.0f;  // padding never wins (real distances are >= 0)
    }
    __syncthreads();  // workspace complete and visible to the whole workgroup; histogram no longer needed

    // ---- set-up 3: the metadata of my buckets into my lanes -----------------------------------------------------------
    float blo0[G], blo1[G], blo2[G], bhi0[G], bhi1[G], bhi2[G], bmax[G], ax[G], ay[G], az[G];
    unsigned akey[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        blo0[g] = blo1[g] = blo2[g] = INFINITY;
        bhi0[g] = bhi1[g] = bhi2[g] = -INFINITY;
        bmax[g] = -1.0f;
        ax[g] = ay[g] = az[g] = 0.0f;
        akey[g] = 0xFFFFFFFFu;
        for (int l = 0; l < kWave; ++l) {   // wave-uniform trip
            const int bk = fps_owner_bucket<NW>(wave, g * kWave + l);
            if (bk >= NB) continue;
            const float4 r = rec[bk * kWave + lane];
            const float dv = dist[bk * kWave + lane];
            const bool valid = dv >= 0.0f;
            // NaN coordinates are left out of the box: such a point can never be updated anyway (min ignores NaN)
            const float l0 = wave_min_f32_dpp(valid ? r.x : INFINITY), h0 = wave_max_f32_dpp(valid ? r.x : -INFINITY);
            const float l1 = wave_min_f32_dpp(valid ? r.y : INFINITY), h1 = wave_max_f32_dpp(valid ? r.y : -INFINITY);
            const float l2 = wave_min_f32_dpp(valid ? r.z : INFINITY), h2 = wave_max_f32_dpp(valid ? r.z : -INFINITY);
            const float mx = wave_max_f32_dpp(dv);   // 1e10: the bucket holds a real point
            // initial arg-max of the bucket: all real points sit at 1e10, the smallest tie key wins
            const unsigned o = __float_as_uint(r.w);
            const unsigned kl = valid ? (TREE ? compat_key((int)o, log2bs) : o) : 0xFFFFFFFFu;
            const unsigned kmin = wave_min_u32_dpp(kl);
            const int wl = __builtin_ctzll(ballot64(kl == kmin) | (1ull << 63));
            const float cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.x), wl));
            const float cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.y), wl));
            const float cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.z), wl));
            if (lane == l) {   // (a v_writelane with a lane number in an SGPR needs M0 on gfx9: the masked moves are as short)
                blo0[g] = l0;
                blo1[g] = l1;
                blo2[g] = l2;
                bhi0[g] = h0;
                bhi1[g] = h1;
                bhi2[g] = h2;
                bmax[g] = mx;
                ax[g] = cx;
                ay[g] = cy;
                az[g] = cz;
                akey[g] = kmin;
            }
        }
    }

    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    if (n > 0) {
        qx = base[0];
        qy = base[1];
        qz = base[2];
    }
    if (tid == 0) outbuf[0] = make_float4(__int_as_float(0), qx, qy, qz);  // row 0: sampling_cuda_kernel.cu:39

    // cached wave candidate (wave-uniform)
    float wm = -1.0f, wx = 0.0f, wy = 0.0f, wz = 0.0f;
    unsigned wkey = 0, pub_bits = 0u, pub_key = 0xFFFFFFFFu;
    int wslot = -1;
    constexpr unsigned kRecParity = NW * 32u;
    char *recb = (char *)hand;
    unsigned wr_off = (unsigned)wave * 32u, rd_off = (unsigned)(lane & (NW - 1)) * 32u;
    bool dirty = true;

    for (int j = 1; j < m; ++j) {
        // ---- A. which of my buckets can the new sample change? (monotone lower bound, exact) -------------
        unsigned long long mask[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float ex = fmaxf(fmaxf(blo0[g] - qx, qx - bhi0[g]), 0.0f);
            const float ey = fmaxf(fmaxf(blo1[g] - qy, qy - bhi1[g]), 0.0f);
            const float ez = fmaxf(fmaxf(blo2[g] - qz, qz - bhi2[g]), 0.0f);
            const float L = FMA ? dist_direct_fma(ex, ey, ez) : dist_direct_nofma(ex, ey, ez);
            mask[g] = ballot64(!(L >= bmax[g]));   // lanes without a bucket: bmax = -1, never set
        }
        // ---- B. the touched buckets: records from L2.  The first two of a group are requested together (one latency for
        //         both), the rare third and later ones one at a time.  The running minima go back unconditionally: with a
        //         conditional store the wait in front of the second record's use could not be counted exactly. ----------
        auto process = [&](auto gtag, int l, int p, const float4 &r, float dv) __attribute__((always_inline)) {
            constexpr int g = decltype(gtag)::value;
            const float dx = r.x - qx, dy = r.y - qy, dz = r.z - qz;
            const float dd = FMA ? dist_direct_fma(dx, dy, dz) : dist_direct_nofma(dx, dy, dz);
            const bool closer = dd < dv;
            const float nd = vmin_f32(dd, dv);  // min(d, tmp[k]) sampling_cuda_kernel.cu:55
            dist[p] = nd;
            const float bold = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bmax[g]), l));
            // Distances only shrink: unless a point that HELD the bucket's maximum gets closer, the bucket's maximum and
            // arg-max are unchanged and the 64-lane refresh is skipped.
            if (ballot64(dv == bold && closer) != 0ull) {  // wave-uniform
                const float mx = wave_max_f32_dpp(nd);
                const unsigned o = __float_as_uint(r.w);
                const unsigned ko = TREE ? compat_key((int)o, log2bs) : o;
                const unsigned long long eq = ballot64(nd == mx);
                int wl = __builtin_ctzll(eq | (1ull << 63));
                if (__builtin_expect(__popcll(eq) != 1, 0)) {  // exact tie inside the bucket (rare): the smallest tie key wins
                    const unsigned kl = nd == mx ? ko : 0xFFFFFFFFu;
                    const unsigned kmin = __builtin_amdgcn_readfirstlane(wave_min_u32_shfl(kl));
                    wl = __builtin_ctzll(ballot64(kl == kmin) | (1ull << 63));
                }
                const float cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.x), wl));
                const float cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.y), wl));
                const float cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.z), wl));
                const int ck = __builtin_amdgcn_readlane((int)ko, wl);
                if (lane == l) {   // one exec-masked block of five scalar-source moves
                    bmax[g] = mx;
                    ax[g] = cx;
                    ay[g] = cy;
                    az[g] = cz;
                    akey[g] = (unsigned)ck;
                }
                if (g * kWave + l == wslot) dirty = true;
            }
        };
        auto walk = [&](auto gtag) __attribute__((always_inline)) {
            constexpr int g = decltype(gtag)::value;
            unsigned long long mm = mask[g];
            if (__builtin_expect(mm != 0ull, 0)) {  // wave-uniform
                const int l0 = __builtin_ctzll(mm);
                mm &= mm - 1ull;
                const int p0 = fps_owner_bucket<NW>(wave, g * kWave + l0) * kWave + lane;
                const float4 r0 = rec[p0];
                const float d0 = dist[p0];
                if (mm == 0ull) {
                    process(gtag, l0, p0, r0, d0);
                } else {
                    const int l1 = __builtin_ctzll(mm);
                    mm &= mm - 1ull;
                    const int p1 = fps_owner_bucket<NW>(wave, g * kWave + l1) * kWave + lane;
                    const float4 r1 = rec[p1];
                    const float d1 = dist[p1];
                    process(gtag, l0, p0, r0, d0);
                    process(gtag, l1, p1, r1, d1);
                    while (mm != 0ull) {
                        const int l2 = __builtin_ctzll(mm);
                        mm &= mm - 1ull;
                        const int p2 = fps_owner_bucket<NW>(wave, g * kWave + l2) * kWave + lane;
                        const float4 r2 = rec[p2];
                        const float d2 = dist[p2];
                        process(gtag, l2, p2, r2, d2);
                    }
                }
            }
        };
        walk(std::integral_constant<int, 0>{});
        if constexpr (G > 1) walk(std::integral_constant<int, 1>{});
        if constexpr (G > 2) walk(std::integral_constant<int, 2>{});
        if constexpr (G > 3) walk(std::integral_constant<int, 3>{});
        // ---- C. wave candidate = max over my bucket maxima (recomputed only if its bucket changed) ---------
        if (dirty) {
            float v = bmax[0], cx = ax[0], cy = ay[0], cz = az[0];
            unsigned kk = akey[0];
            int gs = 0;
#pragma unroll
            for (int g = 1; g < G; ++g) {
                const bool better = bmax[g] > v || (bmax[g] == v && akey[g] < kk);
                v = better ? bmax[g] : v;
                kk = better ? akey[g] : kk;
                cx = better ? ax[g] : cx;
                cy = better ? ay[g] : cy;
                cz = better ? az[g] : cz;
                gs = better ? g : gs;
            }
            wm = wave_max_f32_dpp(v);
            const unsigned long long cm = wm >= 0.0f ? ballot64(v == wm) : 0ull;
            int sl = cm ? __builtin_ctzll(cm) : 0;
            if (__builtin_expect(__popcll(cm) > 1, 0)) {
                const unsigned kl = ((cm >> lane) & 1ull) ? kk : 0xFFFFFFFFu;
                const unsigned kmin = __builtin_amdgcn_readfirstlane(wave_min_u32_shfl(kl));
                sl = __builtin_ctzll(ballot64(kl == kmin) | (1ull << 63));
            }
            wkey = (unsigned)__builtin_amdgcn_readlane((int)kk, sl);
            wslot = __builtin_amdgcn_readlane(gs, sl) * kWave + sl;
            wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cx), sl));
            wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cy), sl));
            wz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cz), sl));
            pub_bits = wm < 0.0f ? 0u : __float_as_uint(wm);  // a wave without points publishes (0, ~0)
            pub_key = wm < 0.0f ? 0xFFFFFFFFu : wkey;
            dirty = false;
        }
        // ---- D. block argmax over the wave candidates: one LDS record per wave, ONE barrier ------------------
        *(uint2 *)(recb + wr_off) = make_uint2(pub_bits, pub_key);          // every lane stores the same bytes (no exec juggling)
        *(float3 *)(recb + wr_off + 16) = make_float3(wx, wy, wz);
        __syncthreads();
        const uint2 r0 = *(const uint2 *)(recb + rd_off);        // every lane reads record lane % NW
        const float3 r1 = *(const float3 *)(recb + rd_off + 16);
        rd_off ^= kRecParity;  // double-buffered by iteration parity: one barrier per iteration is enough
        wr_off ^= kRecParity;
        const unsigned vb = r0.x;
        unsigned mb = vb;
        if constexpr (NW == 8)
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
        if constexpr (CERT) cert.update(mb);
        const unsigned long long wmask = ballot64(vb == mb) & ((1ull << NW) - 1ull);  // lanes 0..NW-1 hold the records
        int wl = __builtin_ctzll(wmask | (1ull << 63));
        if (__builtin_expect(__popcll(wmask) > 1, 0)) {  // equal maxima in several waves (rare): the smallest tie key wins
            const unsigned kq = ((wmask >> lane) & 1ull) ? r0.y : 0xFFFFFFFFu;
            const unsigned kmin = __builtin_amdgcn_readfirstlane(wave_min_u32_shfl(kq));
            wl = __builtin_ctzll(ballot64(kq == kmin) | (1ull << 63));
        }
        const unsigned kwin = (unsigned)__builtin_amdgcn_readlane((int)r0.y, wl);
        qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1.x), wl));
        qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1.y), wl));
        qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1.z), wl));
        int k = kwin == 0xFFFFFFFFu ? 0 : (TREE ? compat_index(kwin, log2bs) : (int)kwin);
        k = __builtin_amdgcn_readfirstlane(k);
        if (tid == 0) outbuf[j & (NT - 1)] = make_float4(__int_as_float(k), qx, qy, qz);
        if ((j & (NT - 1)) == NT - 1) {  // wave-uniform
            __syncthreads();
            const float4 o = outbuf[tid];
            fps_emit(a, start_m + j - (NT - 1) + tid, start_n, __float_as_int(o.x), o.y, o.z, o.w);
            __syncthreads();
        }
    }
    if (((m - 1) & (NT - 1)) != NT - 1) {  // rows of the last, partial chunk
        __syncthreads();
        const int cb = ((m - 1) / NT) * NT;
        if (cb + tid <= m - 1) {
            const float4 o = outbuf[tid];
            fps_emit(a, start_m + cb + tid, start_n, __float_as_int(o.x), o.y, o.z, o.w);
        }
    }
    if (a.prefix_out && tid == 0) a.prefix_out[blockIdx.x] = CERT ? cert.value(m) : 1;  // not tracked: no claim
}

template <int G>
static void fps_owner_launch_g(int mode, int b, const FpsArgs &a, hipStream_t stream) {
    switch (mode) {
        case 0: hipLaunchKernelGGL((fps_bucket_owner_kernel<0, G>), dim3(b), dim3(kStreamThreads), 0, stream, a); break;
        case 1: hipLaunchKernelGGL((fps_bucket_owner_kernel<1, G>), dim3(b), dim3(kStreamThreads), 0, stream, a); break;
        case 2: hipLaunchKernelGGL((fps_bucket_owner_kernel<2, G>), dim3(b), dim3(kStreamThreads), 0, stream, a); break;
        case 3: hipLaunchKernelGGL((fps_bucket_owner_kernel<3, G>), dim3(b), dim3(kStreamThreads), 0, stream, a); break;
        case 4: hipLaunchKernelGGL((fps_bucket_owner_kernel<4, G>), dim3(b), dim3(kStreamThreads), 0, stream, a); break;
        default: hipLaunchKernelGGL((fps_bucket_owner_kernel<5, G>), dim3(b), dim3(kStreamThreads), 0, stream, a); break;
    }
}

size_t fps_stream_workspace_bytes(int b, int n_max) {
    if (n_max <= 0 || b <= 0 || n_max > kStreamMaxBuckets * kWave) return 0;
    return (size_t)b * fps_stream_cloud_bytes(n_max);
}

int fps_bucket_stream_launch(int mode, int b, int n_max, const FpsArgs &a, hipStream_t stream) {
    if (!a.ws || n_max > kStreamMaxBuckets * kWave || a.ws_bytes < fps_stream_workspace_bytes(b, n_max)) return -1;
    const int slots = ((n_max + 63) / 64 + 15) / 16;   // buckets per wave
    if (slots <= 64) fps_owner_launch_g<1>(mode, b, a, stream);
    else if (slots <= 128) fps_owner_launch_g<2>(mode, b, a, stream);
    else if (slots <= 192) fps_owner_launch_g<3>(mode, b, a, stream);
    else fps_owner_launch_g<4>(mode, b, a, stream);
    return check_launch("fps_bucket_owner_kernel");
}

// TGN_FPS_THROUGHPUT: 8 waves x 64 bucket lanes = 512 buckets = 32 768 points, 4096 Z-order cells (16 KiB of LDS), 58 VGPRs:
// four workgroups per CU
constexpr int kOwnerSmallThreads = 512, kOwnerSmallMax = kOwnerSmallThreads / kWave * kWave * kWave;
int fps_bucket_owner_small_launch(int mode, int b, int n_max, const FpsArgs &a, hipStream_t stream) {
    if (!a.ws || n_max > kOwnerSmallMax || n_max <= 0 || a.ws_bytes < fps_stream_workspace_bytes(b, n_max)) return -1;
    switch (mode) {
        case 0: hipLaunchKernelGGL((fps_bucket_owner_kernel<0, 1, kOwnerSmallThreads, 4>), dim3(b), dim3(kOwnerSmallThreads), 0, stream, a); break;
        case 1: hipLaunchKernelGGL((fps_bucket_owner_kernel<1, 1, kOwnerSmallThreads, 4>), dim3(b), dim3(kOwnerSmallThreads), 0, stream, a); break;
        case 2: hipLaunchKernelGGL((fps_bucket_owner_kernel<2, 1, kOwnerSmallThreads, 4>), dim3(b), dim3(kOwnerSmallThreads), 0, stream, a); break;
        case 3: hipLaunchKernelGGL((fps_bucket_owner_kernel<3, 1, kOwnerSmallThreads, 4>), dim3(b), dim3(kOwnerSmallThreads), 0, stream, a); break;
        case 4: hipLaunchKernelGGL((fps_bucket_owner_kernel<4, 1, kOwnerSmallThreads, 4>), dim3(b), dim3(kOwnerSmallThreads), 0, stream, a); break;
        default: hipLaunchKernelGGL((fps_bucket_owner_kernel<5, 1, kOwnerSmallThreads, 4>), dim3(b), dim3(kOwnerSmallThreads), 0, stream, a); break;
    }
    return check_launch("fps_bucket_owner_kernel<small>");
}

#define TGN_FPS_BUCKET_CONFIGS(X) X(256, 8) X(256, 16) X(512, 16) X(512, 24) X(512, 32) X(512, 48) X(512, 56)

template <int MODE>
static int bucket_launch_mode(int b, int n_max, const FpsArgs &a, hipStream_t stream) {
    int nt = 0, p = 0;
    if (const int forced = tuning(kTuneFpsBucketConfig)) {  // experiments: force a shape
        nt = forced >> 8, p = forced & 255;
        if (nt * p < n_max) nt = p = 0;
    }
    if (!nt) {
        int best = 1 << 30;
#define X(NT_, P_)                                             \
    if (NT_ * P_ >= n_max && NT_ * P_ < best) {                \
        best = NT_ * P_;                                       \
        nt = NT_;                                              \
        p = P_;                                                \
    }
        TGN_FPS_BUCKET_CONFIGS(X)
#undef X
        // beside the ball queries of the phased schedule (TGN_FPS_LOW_VALU) a 2049..4096-point cloud is better off on EIGHT waves
        // with half-empty slot sets than on four full ones: 0.81 against 0.86 ms for 4096 -> 1024 beside the level-1 query, and the
        // step 4.51 against 4.55 ms (profiles/r06_phase2_experiments.txt; it lost in round 2, when the query beside it issued
        // 2.4 times the vector instructions)
        if ((a.flags & TGN_FPS_LOW_VALU) && nt == 256 && p == 16) nt = 512;
    }
    if constexpr (MODE == 0) {   // experiments: "fps_cell_bits" = 5 selects the 15-bit cell codes of round 1 (116 KiB of LDS)
        const int cell_bits = tuning(kTuneFpsCellBits);
#define X(NT_, P_)                                                                                              \
    if (cell_bits == 5 && nt == NT_ && p == P_ && !(a.flags & 0x100)) {                                          \
        hipLaunchKernelGGL((fps_bucket_kernel<NT_, P_, MODE, false, 5>), dim3(b), dim3(NT_), 0, stream, a);     \
        return check_launch("fps_bucket_kernel<cb5>");                                                           \
    }
        TGN_FPS_BUCKET_CONFIGS(X)
#undef X
    }
    if constexpr (MODE == 0) {
        if ((a.flags & 0x100) && a.tmp && nt == 512 && p == 48) {  // instrumented build (tools/fps_stats.py)
            hipLaunchKernelGGL((fps_bucket_kernel<512, 48, 0, true>), dim3(b), dim3(512), 0, stream, a);
            return check_launch("fps_bucket_kernel<dbg>");
        }
    }
#define X(NT_, P_)                                                                                   \
    if (nt == NT_ && p == P_) {                                                                      \
        hipLaunchKernelGGL((fps_bucket_kernel<NT_, P_, MODE>), dim3(b), dim3(NT_), 0, stream, a);    \
        return check_launch("fps_bucket_kernel");                                                    \
    }
    TGN_FPS_BUCKET_CONFIGS(X)
#undef X
    return -1;
}

int fps_bucket_launch(int mode, int b, int n_max, const FpsArgs &a, hipStream_t stream) {
    // Measured (profiles/r01_fps_bucket_sweep.txt): 24 000 points 0.89 vs 2.25 us per iteration, 6000 points 0.79 vs
    // 0.98, 4096 points 0.78 vs 0.72 -- there the plain kernel wins (8 points per lane: its whole iteration is already
    // fixed cost).  tgn_set_tuning("fps_bucket_min") overrides.
    int min_n = (a.flags & TGN_FPS_LOW_VALU) ? 2048 : 4097;
    if (const int forced = tuning(kTuneFpsBucketMin); forced >= 0) min_n = forced;
    if (n_max < min_n) return -1;
    switch (mode) {  // bit 0 FMA, bit 1 tree ties, bit 2 certificate tracking (never with tree ties)
        case 0: return bucket_launch_mode<0>(b, n_max, a, stream);
        case 1: return bucket_launch_mode<1>(b, n_max, a, stream);
        case 2: return bucket_launch_mode<2>(b, n_max, a, stream);
        case 3: return bucket_launch_mode<3>(b, n_max, a, stream);
        case 4: return bucket_launch_mode<4>(b, n_max, a, stream);
        default: return bucket_launch_mode<5>(b, n_max, a, stream);
    }
}

}  // namespace tgn

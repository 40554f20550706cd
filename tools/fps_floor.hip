// fps_floor.hip -- the latency floor of one farthest-point-sampling iteration on gfx950.
//
// FPS (sampling_cuda_kernel.cu:42-127) is a serial chain: sample j+1 is the block-wide arg-max of distances that depend on
// sample j.  The production kernel (toothgroupnetwork_amd/csrc/fps_bucket.hip) keeps the cloud in registers and skips every
// bucket the new sample cannot change, so what is left per iteration is a dependent chain of a few dozen instructions issued
// by lone waves, two LDS round trips and one barrier.  This program times that chain with the data work removed, with the
// same primitives (fps_common.h: the 6-step v_max_f32_dpp reduction, ballot, the 32-byte hand-off records, double-buffered by
// parity, one s_barrier, the 3-step v_max_u32_dpp over the 8 records, readlane broadcast), 8 waves per workgroup, one
// workgroup per CU:
//
//   chain      box test of the new sample against the wave's bucket boxes (48 metadata lanes) -> ballot
//              -> the wave's candidate: 6-step DPP max over its bucket maxima, ballot, ctz, 4 readlanes
//              -> LDS record {value, key, x, y, z} -> s_barrier -> read the 8 records -> 3-step DPP max -> readlane
//              -> ballot / ctz / 3 readlanes: the next sample's coordinates in SGPRs.                 EVERY wave, every iteration.
//   chain+1    what the algorithm cannot avoid on top: the bucket that CONTAINS the new sample changes (its own distance
//              becomes 0), so the wave that owns it -- the previous winner -- recomputes those 64 distances, their maximum
//              (6-step DPP, ballot), stores the arg-max point (4 LDS words) and the new bucket maximum (v_writelane), and
//              only THAT wave searches for a new candidate; the other seven publish their cached one and wait at the barrier.
//
// The winner is data dependent (the farthest candidate from the current sample; its coordinates become the next sample), so
// nothing can be hoisted; there is no running minimum, so the values never collapse into ties.  Output: microseconds per
// iteration for both forms, for one workgroup alone and for 256 (one per CU), and one JSON line bench.py reads.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I toothgroupnetwork_amd/csrc tools/fps_floor.hip -o tools/_bin/fps_floor
#include "fps_common.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

using namespace tgn;

namespace tgn {   // (symbols fps_common.h / tgn_common.h declare and the library defines in capi.hip: unused here)
void set_error(const char *, ...) {}
}

constexpr int NT = 512, NW = NT / kWave, P = 48;

template <bool PLUS_ONE>
__global__ __launch_bounds__(NT) void fps_floor_kernel(const float *__restrict__ boxes, const float *__restrict__ pts, int iters,
                                                        float *__restrict__ out, unsigned long long *__restrict__ cycles) {
    __shared__ float4 rec[2][NW][2];
    __shared__ float bmeta[4][NW][P];
    __shared__ float4 outbuf[NT];
    __shared__ unsigned pad[12 * 1024];   // what the production kernel holds (63 KiB): one workgroup per CU, as there
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    if (tid == 0) pad[blockIdx.x & 1023] = 0u;
    // bucket boxes of this wave: lane s < P holds box s (lo corner, hi corner) and its current maximum
    const float *bp = boxes + ((size_t)(blockIdx.x & 7) * NW + wave) * kWave * 6 + lane * 6;
    float blo0 = bp[0], blo1 = bp[1], blo2 = bp[2], bhi0 = bp[3], bhi1 = bp[4], bhi2 = bp[5];
    float bmax = lane < P ? 1.0f + 0.001f * lane : -1.0f;
    // one bucket of 64 points in registers (chain+1 refreshes it)
    const float *pp = pts + ((size_t)(blockIdx.x & 7) * NW + wave) * kWave * 3 + lane * 3;
    const float x0 = pp[0], y0 = pp[1], z0 = pp[2];
    if (lane < P) {
        bmeta[0][wave][lane] = blo0;
        bmeta[1][wave][lane] = blo1;
        bmeta[2][wave][lane] = blo2;
        bmeta[3][wave][lane] = __uint_as_float((unsigned)(wave * kWave + lane));
    }
    __syncthreads();
    float qx = 0.1f, qy = -0.2f, qz = 0.3f;
    float wm = -1.0f, wx = 0.0f, wy = 0.0f, wz = 0.0f;
    unsigned wkey = 0, pub_bits = 0u, pub_key = 0xFFFFFFFFu;
    constexpr unsigned kRecParity = NW * 32u;
    char *recb = (char *)rec;
    unsigned wr_off = (unsigned)wave * 32u, rd_off = (unsigned)(lane & (NW - 1)) * 32u;
    constexpr unsigned long long kSlotMask = (1ull << P) - 1ull;
    int owner = 0;          // the wave whose candidate won the previous iteration (wave-uniform)
    unsigned touched = 0;   // keeps the box test alive
    const long long c0 = clock64();
    for (int j = 1; j < iters; ++j) {
        // ---- A. box test: exact lower bound of the distance from the new sample to each bucket's box -------------------
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 qxy = {qx, qy};
        const f2 lo = f2{blo0, blo1} - qxy, hi = qxy - f2{bhi0, bhi1};
        const float ex = fmaxf(fmaxf(lo.x, hi.x), 0.0f);
        const float ey = fmaxf(fmaxf(lo.y, hi.y), 0.0f);
        const float ez = fmaxf(fmaxf(blo2 - qz, qz - bhi2), 0.0f);
        const float L = dist_direct_nofma(ex, ey, ez);
        const unsigned long long mask = ballot64(!(L >= bmax)) & kSlotMask;
        touched += mask ? 1u : 0u;
        bool search;
        if constexpr (PLUS_ONE) {
            search = wave == owner;      // wave-uniform
            if (search) {
                // ---- U. the bucket that holds the new sample: 64 distances, their maximum, the arg-max point, the bucket maximum
                const float dx = x0 - qx, dy = y0 - qy, dz = z0 - qz;
                const float nd = dist_direct_nofma(dx, dy, dz);
                const float mx = wave_max_f32_dpp(nd);
                const unsigned long long eq = ballot64(nd == mx);
                bool win = nd == mx;
                if (__builtin_expect(__popcll(eq) != 1, 0)) win = win && lane == __builtin_ctzll(eq);
                if (win) {
                    bmeta[0][wave][0] = x0;
                    bmeta[1][wave][0] = y0;
                    bmeta[2][wave][0] = z0;
                    bmeta[3][wave][0] = __uint_as_float((unsigned)(wave * kWave + lane));
                }
                const int mxs = __builtin_amdgcn_readfirstlane(__float_as_int(mx));
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(bmax) : "s"(mxs), "i"(0));
            }
        } else {
            search = true;
            bmax = lane < P ? L : -1.0f;     // the candidate value depends on the new sample: nothing to cache
        }
        if (search) {
            // ---- B. the wave's candidate: max over its bucket maxima, its coordinates from LDS ---------------------------
            const int ml = lane < P ? lane : 0;
            const float4 pm = make_float4(bmeta[0][wave][ml], bmeta[1][wave][ml], bmeta[2][wave][ml], bmeta[3][wave][ml]);
            const float v = lane < P ? bmax : -1.0f;
            wm = wave_max_f32_dpp(v);
            const unsigned long long cm = wm >= 0.0f ? (ballot64(v == wm) & kSlotMask) : 0ull;
            const int sl = cm ? __builtin_ctzll(cm) : 0;
            wkey = (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(pm.w), sl);
            wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pm.x), sl));
            wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pm.y), sl));
            wz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pm.z), sl));
            pub_bits = wm < 0.0f ? 0u : __float_as_uint(wm);
            pub_key = wm < 0.0f ? 0xFFFFFFFFu : wkey;
        }
        // ---- C. hand-off: one record per wave (written by every lane), ONE barrier, 8 records read, 3-step DPP max ------
        *(uint2 *)(recb + wr_off) = make_uint2(pub_bits, pub_key);
        *(float3 *)(recb + wr_off + 16) = make_float3(wx, wy, wz);
        __syncthreads();
        const uint2 r0 = *(const uint2 *)(recb + rd_off);
        const float3 r1 = *(const float3 *)(recb + rd_off + 16);
        rd_off ^= kRecParity;
        wr_off ^= kRecParity;
        const unsigned vb = r0.x;
        unsigned mb = vb;
        asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                     : "+v"(mb));
        mb = (unsigned)__builtin_amdgcn_readlane((int)mb, NW - 1);
        const unsigned long long wmask = ballot64(vb == mb) & ((1ull << NW) - 1ull);
        const int wl = wmask ? __builtin_ctzll(wmask) : 0;
        owner = wl;
        const unsigned kwin = (unsigned)__builtin_amdgcn_readlane((int)r0.y, wl);
        qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1.x), wl));
        qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1.y), wl));
        qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1.z), wl));
        if (tid == 0) outbuf[j & (NT - 1)] = make_float4(__int_as_float((int)kwin), qx, qy, qz);
    }
    const long long c1 = clock64();
    __syncthreads();
    if (tid == 0) {
        out[blockIdx.x * 4 + 0] = qx + outbuf[1].x;
        out[blockIdx.x * 4 + 1] = qy;
        out[blockIdx.x * 4 + 2] = qz;
        out[blockIdx.x * 4 + 3] = (float)touched + (float)pad[blockIdx.x & 1023];
        cycles[blockIdx.x] = (unsigned long long)(c1 - c0);
    }
}

// ---- the primitives alone ------------------------------------------------------------------------------------------------
// What NO register-resident FPS with one workgroup per cloud can do without, per iteration: a value per lane that depends on the
// previous winner (here: one subtract and one multiply), the 6-step DPP maximum of a wave, the lane that holds it (ballot, ctz), ONE
// 8-byte record per wave {value bits, key}, ONE barrier, the 8 records read back, a 3-step DPP maximum, the winning wave (compare,
// ctz), its key (one readlane), and only THEN the winner's coordinates: one 12-byte LDS read indexed by the key.  No box test, no
// bucket metadata, no coordinates in the record, no result row.  The distance between this number and `chain` is what the production
// kernel's bookkeeping costs on the chain (the box test in front, four readlanes + a 16-byte second record for the coordinates, the
// result row); the distance between `chain` and production is the data work (bucket walk and refresh).
__global__ __launch_bounds__(NT) void fps_primitive_kernel(const float *__restrict__ pts, int iters, float *__restrict__ out,
                                                            unsigned long long *__restrict__ cycles) {
    __shared__ uint2 rec[2][NW];
    __shared__ float sx[NT * 3];
    __shared__ unsigned pad[12 * 1024];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    if (tid == 0) pad[blockIdx.x & 1023] = 0u;
    const float *pp = pts + ((size_t)(blockIdx.x & 7) * NW + wave) * kWave * 3 + lane * 3;
    const float x0 = pp[0], y0 = pp[1];
    sx[tid * 3 + 0] = pp[0];
    sx[tid * 3 + 1] = pp[1];
    sx[tid * 3 + 2] = pp[2];
    __syncthreads();
    float qx = 0.1f;
    unsigned par = 0;
    const long long c0 = clock64();
    for (int j = 1; j < iters; ++j) {
        const float d = x0 - qx;
        const float v = d * d + y0 * 0.0f;                       // depends on the previous winner; >= 0
        const float wm = wave_max_f32_dpp(v);
        const unsigned long long eq = ballot64(v == wm);
        const unsigned key = (unsigned)wave * kWave + (unsigned)__builtin_ctzll(eq);
        rec[par][wave] = make_uint2(__float_as_uint(wm), key);   // (every lane writes the same 8 bytes)
        __syncthreads();
        const uint2 r0 = rec[par][lane & (NW - 1)];
        par ^= 1u;
        unsigned mb = r0.x;
        asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                     : "+v"(mb));
        mb = (unsigned)__builtin_amdgcn_readlane((int)mb, NW - 1);
        const unsigned long long wmask = ballot64(r0.x == mb) & ((1ull << NW) - 1ull);
        const unsigned kwin = (unsigned)__builtin_amdgcn_readlane((int)r0.y, (int)__builtin_ctzll(wmask));
        qx = sx[kwin * 3 + 0] + sx[kwin * 3 + 1] * 0.0f + sx[kwin * 3 + 2] * 0.0f;   // the winner's coordinates: one dependent LDS read
    }
    const long long c1 = clock64();
    __syncthreads();
    if (tid == 0) {
        out[blockIdx.x * 4 + 0] = qx + (float)pad[blockIdx.x & 1023];
        cycles[blockIdx.x] = (unsigned long long)(c1 - c0);
    }
}

static double run_primitive(int grid, int iters, const float *d_pts, float *d_out, unsigned long long *d_cyc, int reps, double *cyc_per_iter) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    std::vector<float> ms;
    for (int r = 0; r < reps + 1; ++r) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(fps_primitive_kernel, dim3(grid), dim3(NT), 0, 0, d_pts, iters, d_out, d_cyc);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float t = 0;
        hipEventElapsedTime(&t, a, b);
        if (r) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    std::vector<unsigned long long> cyc(grid);
    hipMemcpy(cyc.data(), d_cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (auto c : cyc) mx = std::max(mx, c);
    *cyc_per_iter = (double)mx / (iters - 1);
    hipEventDestroy(a);
    hipEventDestroy(b);
    return 1e3 * ms[ms.size() / 2] / (iters - 1);
}

static float frand(unsigned &s) {
    s = s * 1664525u + 1013904223u;
    return (float)(s >> 8) / 16777216.0f * 2.0f - 1.0f;
}

template <bool PLUS_ONE>
static double run(int grid, int iters, const float *d_boxes, const float *d_pts, float *d_out, unsigned long long *d_cyc, int reps,
                  double *cyc_per_iter) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    std::vector<float> ms;
    for (int r = 0; r < reps + 1; ++r) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(fps_floor_kernel<PLUS_ONE>, dim3(grid), dim3(NT), 0, 0, d_boxes, d_pts, iters, d_out, d_cyc);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float t = 0;
        hipEventElapsedTime(&t, a, b);
        if (r) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    std::vector<unsigned long long> cyc(grid);
    hipMemcpy(cyc.data(), d_cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (auto c : cyc) mx = std::max(mx, c);
    *cyc_per_iter = (double)mx / (iters - 1);
    hipEventDestroy(a);
    hipEventDestroy(b);
    return 1e3 * ms[ms.size() / 2] / (iters - 1);   // us per iteration (launch overhead of ~5 us over `iters` iterations included)
}

int main(int argc, char **argv) {
    int iters = 4096, reps = 7;
    bool json_only = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
        if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[++i]);
        if (!strcmp(argv[i], "--json")) json_only = true;
    }
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, 0) != hipSuccess) {
        fprintf(stderr, "fps_floor: no GPU\n");
        return 1;
    }
    const int ncu = pr.multiProcessorCount;
    unsigned seed = 12345u;
    std::vector<float> boxes(8 * NW * kWave * 6), pts(8 * NW * kWave * 3);
    for (size_t i = 0; i < boxes.size(); i += 6) {
        for (int c = 0; c < 3; ++c) {
            const float lo = frand(seed);
            boxes[i + c] = lo;
            boxes[i + 3 + c] = lo + 0.05f;
        }
    }
    for (auto &v : pts) v = frand(seed);
    float *d_boxes, *d_pts, *d_out;
    unsigned long long *d_cyc;
    hipMalloc(&d_boxes, boxes.size() * 4);
    hipMalloc(&d_pts, pts.size() * 4);
    hipMalloc(&d_out, 4096 * 4 * 4);
    hipMalloc(&d_cyc, 4096 * 8);
    hipMemcpy(d_boxes, boxes.data(), boxes.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_pts, pts.data(), pts.size() * 4, hipMemcpyHostToDevice);
    double cy[6];
    const double prim_1 = run_primitive(1, iters, d_pts, d_out, d_cyc, reps, &cy[4]);
    const double prim_n = run_primitive(ncu, iters, d_pts, d_out, d_cyc, reps, &cy[5]);
    const double chain_1 = run<false>(1, iters, d_boxes, d_pts, d_out, d_cyc, reps, &cy[0]);
    const double chain_n = run<false>(ncu, iters, d_boxes, d_pts, d_out, d_cyc, reps, &cy[1]);
    const double plus_1 = run<true>(1, iters, d_boxes, d_pts, d_out, d_cyc, reps, &cy[2]);
    const double plus_n = run<true>(ncu, iters, d_boxes, d_pts, d_out, d_cyc, reps, &cy[3]);
    if (!json_only) {
        printf("# tools/fps_floor (%s, %d CUs): us per FPS iteration of the dependent chain alone, %d iterations, 8 waves per workgroup\n",
               pr.name, ncu, iters - 1);
        printf("%-78s %8s %8s\n", "", "1 WG", "1 WG/CU");
        printf("%-78s %8.4f %8.4f   (s_memtime ticks/iter %.0f / %.0f)\n",
               "primitives: value -> 6-step DPP max -> 8-byte record -> barrier -> read -> 3-step DPP -> key -> xyz", prim_1, prim_n, cy[4], cy[5]);
        printf("%-78s %8.4f %8.4f   (s_memtime ticks/iter %.0f / %.0f)\n",
               "chain: box test -> 6-step DPP max -> record -> barrier -> read -> 3-step DPP -> bcast", chain_1, chain_n, cy[0], cy[1]);
        printf("%-78s %8.4f %8.4f   (s_memtime ticks/iter %.0f / %.0f)\n",
               "chain+1: + the owner wave refreshes the one bucket that holds the new sample", plus_1, plus_n, cy[2], cy[3]);
    }
    printf("{\"fps_floor\": {\"device\": \"%s\", \"cus\": %d, \"iters\": %d, \"chain_us\": %.5f, \"chain_us_one_wg\": %.5f, "
           "\"chain_plus_one_bucket_us\": %.5f, \"chain_plus_one_bucket_us_one_wg\": %.5f, \"primitive_us\": %.5f, \"primitive_us_one_wg\": %.5f}}\n",
           pr.name, ncu, iters - 1, chain_n, chain_1, plus_n, plus_1, prim_n, prim_1);
    return 0;
}

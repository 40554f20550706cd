// tools/store_bench.hip -- what does the store path of gfx950 deliver for the patterns group_points could use?
// Build: hipcc --offload-arch=gfx950 -O3 tools/store_bench.hip -o tools/store_bench ; run on the GPU box.
// Every kernel writes the same (queries x chunk) fp32 tensor (4.4 GB at the Shape-A level-2 size); a wave owns one
// query's contiguous chunk, queries are dealt to waves exactly as group_points_kernel does.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// KIND 0: 4 B per lane, 256 B per wave-instruction
// KIND 1: 16 B per lane, 1 KiB per wave-instruction
// KIND 2: 4 B per lane, nontemporal
// KIND 3: 16 B per lane, nontemporal
// KIND 4: 4 B per lane, value loaded from an L2-resident source with the same lane-contiguous pattern (gather proxy)
// KIND 5: 16 B per lane, value = 4 lane-contiguous 4-B loads transposed through LDS (the VEC=4 path's traffic)
// KIND 6: 8 B per lane
template <int KIND>
__global__ __launch_bounds__(256) void k(float *__restrict__ out, const float *__restrict__ src, long long queries,
                                        int chunk, unsigned srcmask) {
    __shared__ float stage[4][256];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nb = gridDim.x;
    const unsigned lb = (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3);
    for (long long q = (long long)lb * 4 + wv; q < queries; q += (long long)nb * 4) {
        float *__restrict__ dst = out + (size_t)q * chunk;
        const unsigned so = (unsigned)(q * 977) & srcmask;
        if (KIND == 0 || KIND == 2 || KIND == 4) {
            for (int e = lane; e < chunk; e += 64) {
                float v = (float)e;
                if (KIND == 4) v = src[(so + (unsigned)e) & srcmask];
                if (KIND == 2) __builtin_nontemporal_store(v, dst + e);
                else dst[e] = v;
            }
        } else if (KIND == 7) {
            // group_rows pattern: 32 rows of 131 floats; 3 leading floats written by a scattered pass, then 128 floats per
            // row with 8 B per lane (row start only 4-B aligned)
            for (int t = lane; t < 96; t += 64) dst[(t / 3) * 131 + t % 3] = 1.0f;
            for (int kk = 0; kk < 32; ++kk) {
                typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
                f2u v = {(float)kk, 1.0f};
                *(f2u *)(dst + kk * 131 + 3 + lane * 2) = v;
            }
        } else if (KIND == 6) {
            for (int e = lane * 2; e < chunk; e += 128) {
                float2 v = {(float)e, 1.0f};
                *(float2 *)(dst + e) = v;
            }
        } else {
            for (int e0 = 0; e0 < chunk; e0 += 256) {
                const int e = e0 + lane * 4;
                float4 v = {(float)e, 1.0f, 2.0f, 3.0f};
                if (KIND == 5) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) stage[wv][u * 64 + lane] = src[(so + (unsigned)(e0 + u * 64 + lane)) & srcmask];
                    v = *(const float4 *)&stage[wv][lane * 4];
                }
                if (e < chunk) {
                    if (KIND == 3) {
                        __builtin_nontemporal_store(v.x, dst + e);
                        __builtin_nontemporal_store(v.y, dst + e + 1);
                        __builtin_nontemporal_store(v.z, dst + e + 2);
                        __builtin_nontemporal_store(v.w, dst + e + 3);
                    } else {
                        *(float4 *)(dst + e) = v;
                    }
                }
            }
        }
    }
}

template <int KIND>
static int run(const char *name, float *out, const float *src, long long queries, int chunk, unsigned srcmask, int blocks) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    float best = 1e30f;
    for (int it = 0; it < 6; ++it) {
        CHK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, src, queries, chunk, srcmask);
        CHK(hipEventRecord(b, 0));
        CHK(hipEventSynchronize(b));
        float ms;
        CHK(hipEventElapsedTime(&ms, a, b));
        if (it > 0 && ms < best) best = ms;
    }
    const double gb = (double)queries * chunk * 4 / 1e9;
    printf("%-44s blocks=%7d  %.3f ms  %.0f GB/s\n", name, blocks, best, gb / best * 1e3);
    return 0;
}

// Staged proxies of group_points_kernel's inner loop (4 B/lane loads and stores):
// STAGE 0: + (k, c) index arithmetic; STAGE 1: + row offsets and relative coordinates read from per-wave LDS tables;
// STAGE 2: + the table set-up per query (index row load, xyz gathers); rows live in a per-cloud window of `win` floats
template <int STAGE>
__global__ __launch_bounds__(256) void k2(float *__restrict__ out, const float *__restrict__ src, const int *__restrict__ idx,
                                         long long queries, int S, int K, int C, unsigned magicC, unsigned win, unsigned nwin) {
    __shared__ unsigned sfb[4][128];
    __shared__ float srel[4][128 * 3];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nb = gridDim.x;
    const unsigned lb = (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3);
    const int total = K * C, D = C - 3;
    for (long long q = (long long)lb * 4 + wv; q < queries; q += (long long)nb * 4) {
        float *__restrict__ dst = out + (size_t)q * total;
        const unsigned cloud = (unsigned)(q / S);
        const unsigned base = (cloud % nwin) * win;
        float pf = 0.0f;
        if (STAGE == 5) {
            // prefetch this wave's share of the NEXT cloud's window (one dword per 128-B line), consumed after the loop
            const unsigned nbase = ((cloud + 1) % nwin) * win;
            const unsigned share = win / (unsigned)S;           // floats per query
            const unsigned off = (unsigned)(q % S) * share + (unsigned)lane * 32u;
            if ((unsigned)lane * 32u < share) pf = src[nbase + off];
        }
        if (STAGE >= 1 && STAGE <= 3) {
            for (int kk = lane; kk < K; kk += 64) {
                unsigned row;
                if (STAGE >= 2) row = (unsigned)idx[q * K + kk];
                else row = (unsigned)((q * 131 + kk * 7) % (win / D));
                sfb[wv][kk] = base + row * (unsigned)D;
                if (STAGE >= 2) {
                    srel[wv][kk * 3 + 0] = src[base + row * 3u + 0u] - 1.0f;
                    srel[wv][kk * 3 + 1] = src[base + row * 3u + 1u] - 1.0f;
                    srel[wv][kk * 3 + 2] = src[base + row * 3u + 2u] - 1.0f;
                } else {
                    srel[wv][kk * 3 + 0] = srel[wv][kk * 3 + 1] = srel[wv][kk * 3 + 2] = 1.0f;
                }
            }
        }
        if (STAGE == 3) {
            // (k, c) tracked incrementally: no multiply / divide per element; tables as 16-B records {row offset, rel xyz}
            unsigned kq = (unsigned)lane / (unsigned)C, c = (unsigned)lane - kq * (unsigned)C;
            const unsigned stepk = 64u / (unsigned)C, stepc = 64u - stepk * (unsigned)C;
#pragma unroll 1
            for (int e = lane; e < total; e += 64) {
                const bool isx = c < 3u;
                const float rel = srel[wv][kq * 3 + (isx ? c : 0u)];
                const unsigned rowoff = sfb[wv][kq];
                const float ld = src[rowoff + (isx ? 0u : c - 3u)];
                dst[e] = isx ? rel : ld;
                c += stepc; kq += stepk;
                if (c >= (unsigned)C) { c -= (unsigned)C; ++kq; }
            }
            continue;
        }
#pragma unroll 1
        for (int e = lane; e < total; e += 64) {
            const unsigned kq = __umulhi((unsigned)e, magicC);
            const unsigned c = (unsigned)e - kq * (unsigned)C;
            const bool isx = c < 3u;
            float rel = 1.0f;
            unsigned rowoff;
            if (STAGE >= 1 && STAGE <= 3) {
                rel = srel[wv][kq * 3 + (isx ? c : 0u)];
                rowoff = sfb[wv][kq];
            } else {
                rowoff = base + ((unsigned)(q * 131 + kq * 7) % (win / D)) * (unsigned)D;
            }
            const float ld = src[rowoff + (isx ? 0u : c - 3u)];
            if (STAGE == 4) __builtin_nontemporal_store(isx ? rel : ld, dst + e);
            else if (STAGE >= 10) {
                // buffer store with cache-policy bits: aux bit 0 = sc0, bit 1 = nt, bit 4 = sc1 (gfx94x/gfx950)
                const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, total * 4, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(isx ? rel : ld), rd, e * 4, 0, STAGE - 10);
            } else dst[e] = isx ? rel : ld;
        }
        if (STAGE == 5 && pf == 123.456f) dst[0] = pf;  // never true (source is zero): keeps the prefetch alive
    }
}

template <int STAGE>
static int run2(const char *name, float *out, const float *src, const int *idx, long long queries, int S, int K, int C,
                unsigned win, int blocks, unsigned nwin = 256) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    float best = 1e30f;
    const unsigned magicC = (unsigned)((0x100000000ULL + C - 1) / C);
    for (int it = 0; it < 6; ++it) {
        CHK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k2<STAGE>, dim3(blocks), dim3(256), 0, 0, out, src, idx, queries, S, K, C, magicC, win, nwin);
        CHK(hipEventRecord(b, 0));
        CHK(hipEventSynchronize(b));
        float ms;
        CHK(hipEventElapsedTime(&ms, a, b));
        if (it > 0 && ms < best) best = ms;
    }
    const double gb = (double)queries * K * C * 4 / 1e9;
    printf("%-44s blocks=%7d  %.3f ms  %.0f GB/s\n", name, blocks, best, gb / best * 1e3);
    return 0;
}

int main(int argc, char **argv) {
    const long long queries = 256LL * 1024;
    const int chunk = 32 * 131;  // K * C at level 2 (multiple of 4, 128-B aligned chunks)
    float *out, *src;
    const unsigned srcwords = 1u << 20;  // 4 MiB source: L2/MALL resident
    CHK(hipMalloc(&out, (size_t)queries * chunk * 4));
    CHK(hipMalloc(&src, (size_t)srcwords * 4));
    CHK(hipMemset(src, 0, (size_t)srcwords * 4));
    for (int blocks : {65536, 4096}) {
        if (run<0>("4 B/lane", out, src, queries, chunk, srcwords - 1, blocks)) return 1;
        if (run<6>("8 B/lane", out, src, queries, chunk, srcwords - 1, blocks)) return 1;
        if (run<1>("16 B/lane", out, src, queries, chunk, srcwords - 1, blocks)) return 1;
        if (run<2>("4 B/lane nontemporal", out, src, queries, chunk, srcwords - 1, blocks)) return 1;
        if (run<3>("4x4 B/lane nontemporal (16 B span)", out, src, queries, chunk, srcwords - 1, blocks)) return 1;
        if (run<7>("row pattern: 8 B/lane, rows 4-B aligned", out, src, queries, chunk, srcwords - 1, blocks)) return 1;
        if (run<4>("4 B/lane load (L2) + 4 B/lane store", out, src, queries, chunk, srcwords - 1, blocks)) return 1;
        if (run<5>("4x4 B/lane load, LDS transpose, 16 B store", out, src, queries, chunk, srcwords - 1, blocks)) return 1;
    }
    {
        // level-2 shape: 256 clouds x 1024 queries, K = 32, C = 131, rows of D = 128 floats in a 4096-row window per cloud
        const int S = 1024, K = 32, C = 131, Nrows = 4096;
        const unsigned win = (unsigned)Nrows * 128u;
        float *big;
        int *idx;
        CHK(hipMalloc(&big, (size_t)256 * win * 4));
        CHK(hipMemset(big, 0, (size_t)256 * win * 4));
        CHK(hipMalloc(&idx, (size_t)queries * K * 4));
        int *h = (int *)malloc((size_t)queries * K * 4);
        unsigned r = 12345u;
        for (long long i = 0; i < queries; ++i) {
            r = r * 1664525u + 1013904223u;
            const unsigned b0 = (r >> 8) % Nrows;
            for (int j = 0; j < K; ++j) {
                r = r * 1664525u + 1013904223u;
                h[i * K + j] = (int)((b0 + (r >> 8) % 256u) % Nrows);
            }
        }
        CHK(hipMemcpy(idx, h, (size_t)queries * K * 4, hipMemcpyHostToDevice));
        free(h);
        const int blocks = 65536;
        if (run2<0>("proxy: + (k,c) arithmetic, 512 MB source", out, big, idx, queries, S, K, C, win, blocks)) return 1;
        if (run2<1>("proxy: + LDS tables", out, big, idx, queries, S, K, C, win, blocks)) return 1;
        if (run2<2>("proxy: + per-query table set-up (= group)", out, big, idx, queries, S, K, C, win, blocks)) return 1;
        if (run2<3>("proxy: LDS tables, incremental (k,c)", out, big, idx, queries, S, K, C, win, blocks)) return 1;
        if (run2<0>("proxy: (k,c) arithmetic, 16 windows (32 MB)", out, big, idx, queries, S, K, C, win, blocks, 16)) return 1;
        if (run2<0>("proxy: (k,c) arithmetic, 64 windows (128 MB)", out, big, idx, queries, S, K, C, win, blocks, 64)) return 1;
        if (run2<4>("proxy: (k,c) arithmetic, 512 MB, nt stores", out, big, idx, queries, S, K, C, win, blocks)) return 1;
        if (run2<10>("proxy: 512 MB, buffer store aux=0", out, big, idx, queries, S, K, C, win, blocks)) return 1;
        if (run2<11>("proxy: 512 MB, buffer store sc0", out, big, idx, queries, S, K, C, win, blocks)) return 1;
        if (run2<12>("proxy: 512 MB, buffer store nt", out, big, idx, queries, S, K, C, win, blocks)) return 1;
        if (run2<13>("proxy: 512 MB, buffer store sc0 nt", out, big, idx, queries, S, K, C, win, blocks)) return 1;
        if (run2<26>("proxy: 512 MB, buffer store sc1", out, big, idx, queries, S, K, C, win, blocks)) return 1;
        if (run2<27>("proxy: 512 MB, buffer store sc1 sc0", out, big, idx, queries, S, K, C, win, blocks)) return 1;
        if (run2<28>("proxy: 512 MB, buffer store sc1 nt", out, big, idx, queries, S, K, C, win, blocks)) return 1;
        if (run2<29>("proxy: 512 MB, buffer store sc1 sc0 nt", out, big, idx, queries, S, K, C, win, blocks)) return 1;
        if (run2<5>("proxy: (k,c) arithmetic, 512 MB, prefetch", out, big, idx, queries, S, K, C, win, blocks)) return 1;
        // per-scan window size (256 distinct windows): how much of it survives in L2 beside the store stream?
        if (run2<0>("proxy: 256 windows of 0.25 MB", out, big, idx, queries, S, K, C, win / 8, blocks)) return 1;
        if (run2<0>("proxy: 256 windows of 0.5 MB", out, big, idx, queries, S, K, C, win / 4, blocks)) return 1;
        if (run2<0>("proxy: 256 windows of 1 MB", out, big, idx, queries, S, K, C, win / 2, blocks)) return 1;
        if (run2<0>("proxy: 256 windows of 2 MB", out, big, idx, queries, S, K, C, win, blocks)) return 1;
        // same loop, 4 MiB window shared by all clouds: is it the source footprint?
        if (run2<0>("proxy: (k,c) arithmetic, 4 MiB source", out, big, idx, queries, 1 << 30, K, C, 1u << 20, blocks)) return 1;
    }
    CHK(hipMemset(out, 0, 1 << 20));
    return 0;
}

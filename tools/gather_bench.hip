// tools/gather_bench.hip -- what does the LOAD path of one gfx950 CU deliver for the access patterns a grouping kernel
// can use?  (round 2: the grouping kernel without its stores still takes 1.3-1.9 ms per 4.3 GB: profiles/r02_group_sides.txt)
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o tools/gather_bench ; run on the GPU box.
//
// Every wave walks `steps` 1-KiB steps; a step gathers 256 consecutive floats of a pseudo-random 2-KiB row of a per-XCD
// window (L2 resident when the window is small, HBM/MALL when it is large) and consumes them (ds_read / VALU add).
//   WIDTH 4 : four 256-B wave-instructions per step (4 B per lane)      WIDTH 16: one 1-KiB instruction (16 B per lane,
//   PATH  0 : into VGPRs                                                          source only 4-B aligned: `shift`)
//   PATH  1 : LDS-DMA (buffer_load ... lds), ring of NBUF slots, explicit vmcnt
// Reported: GB/s, wave-instructions per microsecond per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70);
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ unsigned rnd(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <int WIDTH, int PATH, int NBUF>
__global__ __launch_bounds__(256) void k(const float *__restrict__ src, float *__restrict__ sink, unsigned win_rows,
                                        unsigned nwin, int steps, unsigned shift) {
    __shared__ __attribute__((aligned(16))) float ring[4][NBUF][256];
    const unsigned lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned x = blockIdx.x & 7u;  // XCD (observed dispatch)
    const unsigned wave_id = (blockIdx.x >> 3) * 4 + wv;
    const float *base = src + (size_t)((x + 8u * (wave_id % (nwin / 8u ? nwin / 8u : 1u))) % nwin) * win_rows * 512u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)(win_rows * 2048u), 0x00020000);
    float acc = 0.0f;
    unsigned seed = wave_id * 2654435761u + 12345u;
    auto row_off = [&](unsigned t) {   // byte offset of the 1-KiB half row gathered by step t (wave-uniform)
        const unsigned r = rnd(seed + t);
        return (r % win_rows) * 2048u + ((r >> 20) & 1u) * 1008u + shift * 4u;   // 1008: keeps +1 KiB inside the row
    };
    if (PATH == 0) {
        for (int t = 0; t < steps; ++t) {
            const unsigned off = __builtin_amdgcn_readfirstlane(row_off((unsigned)t));
            if (WIDTH == 4) {
                float v[4];
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    v[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4u + s * 256u, off, 0));
                acc += (v[0] + v[1]) + (v[2] + v[3]);
            } else {
                const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u, off, 0);
                acc += __builtin_bit_cast(float, w[0]) + __builtin_bit_cast(float, w[3]);
            }
        }
    } else {
        auto issue = [&](unsigned t) {
            float *slot = ring[wv][t % NBUF];
            const unsigned off = __builtin_amdgcn_readfirstlane(row_off(t));
            if (WIDTH == 4) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(slot + s * 64), 4,
                                                             lane * 4u + s * 256u, off, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)slot, 16, lane * 16u, off, 0, 0);
            }
        };
        constexpr int PER = WIDTH == 4 ? 4 : 1;
        for (int t = 0; t < NBUF - 1 && t < steps; ++t) issue((unsigned)t);
        for (int t = 0; t < steps; ++t) {
            if (t + NBUF - 1 < steps) {
                issue((unsigned)(t + NBUF - 1));
                wait_vmcnt<PER *(NBUF - 1)>();
            } else {
                wait_vmcnt<0>();
            }
            const f32x4 w = *(const f32x4 *)&ring[wv][t % NBUF][lane * 4];
            acc += w[0] + w[3];
        }
    }
    if (acc == 123.456f) sink[threadIdx.x] = acc;
}

template <int WIDTH, int PATH, int NBUF>
static int run(const char *name, const float *src, float *sink, unsigned win_rows, unsigned nwin, int blocks, int steps,
               unsigned shift) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CHK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((k<WIDTH, PATH, NBUF>), dim3(blocks), dim3(256), 0, 0, src, sink, win_rows, nwin, steps, shift);
        CHK(hipEventRecord(b, 0));
        CHK(hipEventSynchronize(b));
        float ms;
        CHK(hipEventElapsedTime(&ms, a, b));
        if (it > 0 && ms < best) best = ms;
    }
    const double bytes = (double)blocks * 4 * steps * 1024.0;
    const double instr = (double)blocks * 4 * steps * (WIDTH == 4 ? 4 : 1);
    printf("%-34s win %5.2f MB x%3u blocks=%5d (%2d waves/CU) shift=%u  %7.3f ms  %6.0f GB/s  %5.1f instr/us/CU\n", name,
           win_rows * 2048.0 / 1048576.0, nwin, blocks, blocks * 4 / 256, shift, best, bytes / best / 1e6,
           instr / (best * 1e3) / 256.0);
    return 0;
}

int main() {
    const unsigned rows_total = 256u * 1024u;     // 512 MB source
    float *src, *sink;
    CHK(hipMalloc(&src, (size_t)rows_total * 2048));
    CHK(hipMemset(src, 0, (size_t)rows_total * 2048));
    CHK(hipMalloc(&sink, 4096));
    const int steps = 2048;
    struct Win { unsigned rows, n; } wins[] = {{1024, 8}, {1024, 256}};   // 2 MB per XCD (L2 resident) / 256 windows (512 MB)
    for (const Win &w : wins) {
        for (int blocks : {256, 512, 1024, 2048}) {
            if (run<4, 0, 2>("4 B/lane -> VGPR", src, sink, w.rows, w.n, blocks, steps, 0)) return 1;
            if (run<16, 0, 2>("16 B/lane -> VGPR (aligned)", src, sink, w.rows, w.n, blocks, steps, 0)) return 1;
            if (run<16, 0, 2>("16 B/lane -> VGPR (4-B aligned)", src, sink, w.rows, w.n, blocks, steps, 1)) return 1;
            if (run<4, 1, 4>("4 B/lane -> LDS ring 4", src, sink, w.rows, w.n, blocks, steps, 0)) return 1;
            if (run<4, 1, 12>("4 B/lane -> LDS ring 12", src, sink, w.rows, w.n, blocks, steps, 0)) return 1;
            if (run<16, 1, 4>("16 B/lane -> LDS ring 4 (aligned)", src, sink, w.rows, w.n, blocks, steps, 0)) return 1;
            if (run<16, 1, 4>("16 B/lane -> LDS ring 4 (4-B al.)", src, sink, w.rows, w.n, blocks, steps, 3)) return 1;
            if (run<16, 1, 8>("16 B/lane -> LDS ring 8 (4-B al.)", src, sink, w.rows, w.n, blocks, steps, 3)) return 1;
            if (w.n > 8 || blocks <= 512)
                if (run<16, 1, 16>("16 B/lane -> LDS ring 16 (4-B al.)", src, sink, w.rows, w.n, blocks > 512 ? 512 : blocks, steps, 3)) return 1;
        }
    }
    return 0;
}

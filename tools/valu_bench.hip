// tools/valu_bench.hip -- VALU issue-rate microbenchmark for gfx950 (what does one wave64 instruction cost?).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_bench.hip -o tools/valu_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND>
__global__ void bench(float *out, int iters, long long *cycles) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 1.0001f, b1 = 0.9999f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    f2 q = {b0, b1};
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (KIND == 0) {  // scalar v_fma_f32 x8 independent
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
            } else if (KIND == 1) {  // v_pk_fma_f32 x8
                asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                             "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));
            } else if (KIND == 2) {  // v_min_f32 x8
                asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n"
                             "v_min_f32 %4, %4, %9\n v_min_f32 %5, %5, %9\n v_min_f32 %6, %6, %9\n v_min_f32 %7, %7, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
            } else if (KIND == 3) {  // v_cmp_gt_f32 + v_cndmask x4 pairs
                asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_gt_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc\n"
                             "v_cmp_gt_f32 vcc, %4, %5\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_gt_f32 vcc, %6, %7\n v_cndmask_b32 %6, %6, %7, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");
            } else if (KIND == 4) {  // v_max3_f32 x8
                asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                             "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
            } else if (KIND == 5) {  // v_pk_add_f32 x8
                asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                             "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));
            } else if (KIND == 6) {  // v_sub_f32 + v_mul_f32 mix x8 (plain scalar adds/muls)
                asm volatile("v_sub_f32 %0, %0, %8\n v_mul_f32 %1, %1, %9\n v_sub_f32 %2, %2, %8\n v_mul_f32 %3, %3, %9\n"
                             "v_add_f32 %4, %4, %8\n v_mul_f32 %5, %5, %9\n v_add_f32 %6, %6, %8\n v_mul_f32 %7, %7, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
            } else if (KIND == 7) {  // v_mov_b32 dpp row_shr:1 x8
                asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        }
    }
    long long t1 = clock64();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int KIND>
int run(const char *name, int waves_per_simd) {
    float *out; long long *cyc;
    CHK(hipMalloc(&out, 4)); CHK(hipMalloc(&cyc, 8));
    const int iters = 20000, threads = 64 * 4 * waves_per_simd, blocks = 256;
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 100, cyc);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a)); hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc); CHK(hipEventRecord(b));
    CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    long long h; CHK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    const double instr_per_wave = (double)iters * 32;
    const double instr_per_simd = instr_per_wave * waves_per_simd;
    printf("%-26s waves/SIMD=%d  %8.3f ms  %6.3f ns/wave-instr/SIMD  clock64 ticks/instr/SIMD=%.3f (ticks %lld; tick rate %.1f MHz)\n", name,
           waves_per_simd, ms, ms * 1e6 / instr_per_simd, (double)h / instr_per_simd, h, h / (ms * 1e3));
    hipFree(out); hipFree(cyc);
    return 0;
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<5>("v_pk_add_f32", w); run<6>("v_sub/mul/add_f32", w);
        run<2>("v_min_f32", w); run<3>("v_cmp+v_cndmask", w); run<4>("v_max3_f32", w); run<7>("v_mov_dpp", w);
    }
    return 0;
}

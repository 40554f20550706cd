// atomic_floor.hip -- what the L2 atomic units of an MI355X sustain for the scatter of the gather family's backward kernels.
//
// grouping / interpolation / subtraction / aggregation backward (grouping_cuda_kernel.cu:16-25, interpolation_cuda_kernel.cu:20-33,
// subtraction_cuda_kernel.cu:17-30, aggregation_cuda_kernel.cu:21-39) add a row of c floats into grad_input[idx[r]] for every
// (point, neighbour) pair: the targets are data dependent, so the adds are fp32 atomics (`global_atomic_add_f32`, no return).  Such a
// kernel is bound neither by HBM nor by the CUs but by the atomic ALUs of the L2 channels.  This program measures that ceiling with
// nothing else in the way: every wave adds REGISTER values (no source loads) to pseudo-random rows of an (n, c) table --
//   line   lanes cover whole 128-byte lines: one instruction = 64 consecutive floats = two rows of c = 32 (the layout gather.hip uses)
//   quad   a lane owns 4 consecutive channels and issues 4 atomics: each instruction touches every fourth dword of 8 rows
// and prints dword-atomics per second for both.  tools/secondary_bench.py prices the backward kernels against the `line` figure.
//
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/atomic_floor.hip -o tools/_bin/atomic_floor
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void scatter_kernel(float *__restrict__ table, unsigned n, unsigned c, unsigned long long rows) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned long long wave = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned long long nwaves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
    if (MODE == 0) {   // lane = channel: 64 / c rows per instruction, whole lines
        const unsigned rpw = 64u / c, sub = lane / c, ch = lane % c;
        for (unsigned long long r = wave * rpw + sub; r < rows; r += nwaves * rpw) {
            const unsigned t = hash32((unsigned)r) % n;
            atomicAdd(table + (size_t)t * c + ch, 1.0f);
        }
    } else {           // lane = 4 channels: 16 * 4 / c ... rows per instruction, four instructions per row
        const unsigned c4 = c / 4u, rpw = 64u / c4, sub = lane / c4, q = lane % c4;
        for (unsigned long long r = wave * rpw + sub; r < rows; r += nwaves * rpw) {
            const unsigned t = hash32((unsigned)r) % n;
            float *dst = table + (size_t)t * c + q * 4u;
            atomicAdd(dst + 0, 1.0f);
            atomicAdd(dst + 1, 1.0f);
            atomicAdd(dst + 2, 1.0f);
            atomicAdd(dst + 3, 1.0f);
        }
    }
}

template <int MODE>
static double run(float *table, unsigned n, unsigned c, unsigned long long rows, int reps) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    std::vector<float> ms;
    for (int r = 0; r < reps + 1; ++r) {
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL(scatter_kernel<MODE>, dim3(8192), dim3(256), 0, 0, table, n, c, rows);
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float t = 0;
        (void)hipEventElapsedTime(&t, a, b);
        if (r) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return ms[ms.size() / 2];
}

int main(int argc, char **argv) {
    unsigned n = 24000, c = 32;
    unsigned long long rows = 24000ull * 36ull;
    bool json_only = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--json")) json_only = true;
        if (!strcmp(argv[i], "--n") && i + 1 < argc) n = (unsigned)atoi(argv[++i]);
        if (!strcmp(argv[i], "--rows") && i + 1 < argc) rows = strtoull(argv[++i], nullptr, 10);
    }
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, 0) != hipSuccess) {
        fprintf(stderr, "atomic_floor: no GPU\n");
        return 1;
    }
    float *table = nullptr;
    if (hipMalloc(&table, (size_t)n * c * 4) != hipSuccess) return 1;
    (void)hipMemset(table, 0, (size_t)n * c * 4);
    const double ms_line = run<0>(table, n, c, rows, 9), ms_quad = run<1>(table, n, c, rows, 9);
    const double dwords = (double)rows * c;
    if (!json_only) {
        printf("# tools/atomic_floor (%d CUs): %llu rows of %u floats added into pseudo-random rows of a (%u, %u) fp32 table, no source loads\n",
               pr.multiProcessorCount, rows, c, n, c);
        printf("lane = channel (whole 128-B lines per instruction)   %8.1f us   %7.1f G dword-atomics/s   %6.0f GB/s of operands\n",
               1e3 * ms_line, dwords / ms_line / 1e6, 4 * dwords / ms_line / 1e6);
        printf("lane = 4 channels (every 4th dword of 8 rows)         %8.1f us   %7.1f G dword-atomics/s   %6.0f GB/s of operands\n",
               1e3 * ms_quad, dwords / ms_quad / 1e6, 4 * dwords / ms_quad / 1e6);
    }
    printf("{\"atomic_floor\": {\"rows\": %llu, \"c\": %u, \"n\": %u, \"line_us\": %.2f, \"quad_us\": %.2f, \"line_gatomics_per_s\": %.2f, "
           "\"quad_gatomics_per_s\": %.2f}}\n",
           rows, c, n, 1e3 * ms_line, 1e3 * ms_quad, dwords / ms_line / 1e6, dwords / ms_quad / 1e6);
    (void)hipFree(table);
    return 0;
}

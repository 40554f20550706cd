// tools/interference/corun.hip -- co-runner kernels to find out WHAT slows the FPS workgroups down when the grouping
// kernel shares their CUs.  Each fits beside an FPS workgroup (<= 32 VGPRs, <= 8 KiB LDS, 256 threads).
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/interference/corun.hip -o tools/interference/libcorun.so
#include <hip/hip_runtime.h>
extern "C" {

// pure VALU: no memory traffic at all
__global__ __launch_bounds__(256) void k_valu(float *out, int iters) {
    float a = threadIdx.x, b = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) a = __builtin_fmaf(a, b, c);
    }
    if (a == 12345.678f) out[0] = a;
}
// LDS traffic only
__global__ __launch_bounds__(256) void k_lds(float *out, int iters) {
    __shared__ float s[2048];
    const int t = threadIdx.x;
    for (int i = t; i < 2048; i += 256) s[i] = i;
    __syncthreads();
    float a = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) a += s[(t + u * 67 + i) & 2047];
    }
    if (a == 12345.678f) out[0] = a;
}
// streaming stores (HBM write), almost no VALU
__global__ __launch_bounds__(256) void k_store(float *out, size_t n, int rounds) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (int r = 0; r < rounds; ++r)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = (float)r;
}
// L2-resident loads (gather-like), no HBM
__global__ __launch_bounds__(256) void k_l2load(const float *src, float *out, unsigned mask, int iters) {
    unsigned p = (blockIdx.x * 256 + threadIdx.x) & mask;
    float a = 0;
    for (int i = 0; i < iters; ++i) {
        a += src[p];
        p = (p + 64u * 977u) & mask;
    }
    if (a == 12345.678f) out[0] = a;
}

int corun_valu(float *out, int blocks, int iters, void *st) { hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, (hipStream_t)st, out, iters); return (int)hipGetLastError(); }
int corun_lds(float *out, int blocks, int iters, void *st) { hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 0, (hipStream_t)st, out, iters); return (int)hipGetLastError(); }
int corun_store(float *out, size_t n, int blocks, int rounds, void *st) { hipLaunchKernelGGL(k_store, dim3(blocks), dim3(256), 0, (hipStream_t)st, out, n, rounds); return (int)hipGetLastError(); }
int corun_l2load(const float *src, float *out, unsigned mask, int blocks, int iters, void *st) { hipLaunchKernelGGL(k_l2load, dim3(blocks), dim3(256), 0, (hipStream_t)st, src, out, mask, iters); return (int)hipGetLastError(); }
}

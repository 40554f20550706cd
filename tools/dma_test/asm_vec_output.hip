// hipcc 7.2 (ROCm 7.2.0) miscompile probe: an inline-asm output of an UNSIGNED ext_vector_type(4) whose elements are bit_cast to float one by one
// is read as element 0 four times (k3: `ds_write2_b32 v6, v2, v2`); the same with a FLOAT vector is right (k4, k5).
//   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o - tools/dma_test/asm_vec_output.hip | grep -E "^_Z|ds_write"
#include <hip/hip_runtime.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define DESC u32x4 desc = {(unsigned)(unsigned long long)p, (unsigned)((unsigned long long)p >> 32), (unsigned)n, 0x00020000u}; \
    desc[0] = __builtin_amdgcn_readfirstlane(desc[0]); desc[1] = __builtin_amdgcn_readfirstlane(desc[1]); desc[2] = __builtin_amdgcn_readfirstlane(desc[2]); \
    unsigned off = threadIdx.x * 24u; __shared__ float sm[1024];
__global__ void k3(const float* p, float* out, int n) {   // u32x4 + bit_cast, no empty asm
    DESC
    u32x4 r;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r) : "v"(off), "s"(desc) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = 0; i < 4; ++i) sm[threadIdx.x * 9 + i] = __builtin_bit_cast(float, r[i]);
    __syncthreads();
    out[threadIdx.x] = sm[(threadIdx.x * 7) % 576];
}
__global__ void k4(const float* p, float* out, int n) {   // f32x4, separate wait asm, no tie
    DESC
    f32x4 r;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r) : "v"(off), "s"(desc) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = 0; i < 4; ++i) sm[threadIdx.x * 9 + i] = r[i];
    __syncthreads();
    out[threadIdx.x] = sm[(threadIdx.x * 7) % 576];
}
struct S { unsigned v; f32x4 a; };
__global__ void k5(const float* p, float* out, int n) {   // f32x4 in a struct
    DESC
    S s; s.v = off;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(s.a) : "v"(s.v), "s"(desc) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = 0; i < 4; ++i) sm[threadIdx.x * 9 + i] = s.a[i];
    __syncthreads();
    out[threadIdx.x] = sm[(threadIdx.x * 7) % 576];
}

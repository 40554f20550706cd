// Does `buffer_load_dwordx4 ... lds` (gfx950) accept LDS destinations and global sources that are only 4-byte aligned?
// Each case: 64 lanes x 16 B from src + src_off (bytes) into LDS at dst_off (bytes); the LDS block is then dumped.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
__global__ void k(const float *src, float *out, int src_off, int dst_off, int nlanes) {
    __shared__ __attribute__((aligned(16))) float buf[512];
    for (int i = threadIdx.x; i < 512; i += 64) buf[i] = -1.0f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 4096 * 4, 0x00020000);
    if ((int)threadIdx.x < nlanes)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)((char *)buf + dst_off), 16,
                                                 threadIdx.x * 16u + (unsigned)src_off, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = buf[i];
}
int main() {
    float *hs = (float *)malloc(4096 * 4), *ho = (float *)malloc(512 * 4), *ds, *dout;
    for (int i = 0; i < 4096; ++i) hs[i] = (float)i;
    hipMalloc(&ds, 4096 * 4); hipMalloc(&dout, 512 * 4);
    hipMemcpy(ds, hs, 4096 * 4, hipMemcpyHostToDevice);
    int cases[][3] = {{0, 0, 64}, {0, 4, 64}, {0, 8, 64}, {0, 12, 64}, {4, 0, 64}, {12, 4, 64}, {20, 12, 32}, {4, 20, 32}};
    for (auto &c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, ds, dout, c[0], c[1], c[2]);
        hipMemcpy(ho, dout, 512 * 4, hipMemcpyDeviceToHost);
        int bad = 0, first_bad = -1;
        for (int i = 0; i < 512; ++i) {
            const int e = i - c[1] / 4;   // element of the copied block
            const float want = (e >= 0 && e < c[2] * 4) ? (float)(e + c[0] / 4) : -1.0f;
            if (ho[i] != want) { if (first_bad < 0) first_bad = i; ++bad; }
        }
        printf("src_off=%2d dst_off=%2d lanes=%2d: %s (mismatches %d, first at %d: got %g)\n", c[0], c[1], c[2], bad ? "WRONG" : "ok", bad,
               first_bad, first_bad >= 0 ? ho[first_bad] : 0.0f);
    }
    return 0;
}

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, const int* addr_of_lane)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    // builtin form
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + addr_of_lane[lane]));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)v[j];
}
int main()
{
    int h_addr[64];
    unsigned short h_out[256];
    int* d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int pat = 0; pat < 3; ++pat) {
        for (int l = 0; l < 64; ++l) {
            if (pat == 0) h_addr[l] = l * 4;                       // lane l -> elements 4l..4l+3 (contiguous)
            else if (pat == 1) h_addr[l] = (l & 15) / 4 * 64 + (l & 3) * 4 + (l >> 4) * 16;   // rows of 64 elements: lane i in a group: row i/4, col quad i%4; group g -> cols 16g
            else h_addr[l] = l * 100;                              // arbitrary: elements 100l..100l+3
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, d_addr);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) printf("lane %2d addr %4d -> %4d %4d %4d %4d\n", l, h_addr[l], h_out[4*l], h_out[4*l+1], h_out[4*l+2], h_out[4*l+3]);
    }
    return 0;
}

// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 as a function of the number of independent accumulators a wavefront rotates through
// and of the wavefronts per SIMD.  hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip && ./mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void chain(float* out, int iters)
{
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    s16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (short)(0x3f80 + threadIdx.x % 3); y[e] = (short)(0x3f80 + threadIdx.x % 5); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 24 / NACC; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC> void run(int wg_per_cu, int cus, float* d)
{
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    chain<NACC><<<wg_per_cu * cus, 256>>>(d, 10);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        chain<NACC><<<wg_per_cu * cus, 256>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double mfma_per_simd = (double)iters * 24 * wg_per_cu;            // 4 waves per WG = one per SIMD
    const double flops = mfma_per_simd * 4 * cus * 2.0 * 32 * 32 * 16;
    printf("accumulators %d, waves/SIMD %d: %.3f ms, %.0f TFLOP/s, %.1f ns per MFMA per SIMD (= %.1f cycles at 2.4 GHz)\n", NACC, wg_per_cu, best,
           flops / best / 1e9, best * 1e6 / mfma_per_simd, best * 1e6 / mfma_per_simd * 2.4);
}

int main()
{
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* d; hipMalloc(&d, 4 * 256 * 8 * cus * 4);
    for (int w = 1; w <= 4; w *= 2) { run<1>(w, cus, d); run<2>(w, cus, d); run<4>(w, cus, d); run<8>(w, cus, d); }
    return 0;
}

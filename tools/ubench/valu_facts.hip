// Micro-benchmark + exhaustive checks behind the round-6 16-bit neuron kernels (profiles/r06/valu_facts.log):
//  (1) issue cost (cycles per wave64 instruction per SIMD) of v_fma_f32, v_pk_fma_f32, v_rcp_f32, v_cvt_f32_f16, v_cndmask, v_mov_dpp as a
//      function of the wavefronts per SIMD — the 16-bit fused LIF kernels issue 33 (forward) / 57 (backward) VALU instructions per update and
//      this tells whether they are VALU- or HBM-bound;
//  (2) exhaustive: one Newton step on v_rcp_f32 (r1 = fma(fma(-d, r0, 1), r0, r0)) against the correctly rounded 1.0f / d for EVERY d in [1, inf):
//      which d disagree (the surrogate's 1 / (1 + u^2) has d >= 1);
//  (3) exhaustive: v_cvt_pk_bf16_f32 against the oracle's integer round-to-nearest-even narrowing for every fp32 bit pattern.
// hipcc --offload-arch=gfx950 -O3 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off -o valu_facts valu_facts.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float seed)
{
    float a[8];
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = (f2){a[i], a[i] + 0.5f}; }
    const float m = 1.0000001f, c = 1e-7f;
    int sel = threadIdx.x & 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                else if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"((f2){m, m}), "v"((f2){c, c}));
                else if (OP == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                else if (OP == 3) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a[i]));
                else if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : );
                else if (OP == 5) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                else if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"((f2){m, m}));
                else if (OP == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                else if (OP == 8) asm volatile("v_cmp_le_f32 vcc, 0, %0\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc" : "+v"(a[i]), "+v"(sel) : : "vcc");
            }
        }
    }
    float s = (float)sel;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP> int run_rate(const char* name, float* d, int cus, double mhz)
{
    const int iters = 4000;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int wpc : {1, 2, 4, 8}) {      // workgroups of 4 wavefronts per CU = wavefronts per SIMD
        rate_kernel<OP><<<wpc * cus, 256>>>(d, 10, 1.f);
        CHK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(e0));
            rate_kernel<OP><<<wpc * cus, 256>>>(d, iters, 1.f);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double instr_per_simd = (double)iters * 64 * wpc * (OP == 8 ? 2 : 1);
        printf("%-14s waves/SIMD %d: %8.3f ms  %6.2f cycles per instruction per SIMD (at %.0f MHz)\n", name, wpc, best, best * 1e-3 * mhz * 1e6 / instr_per_simd, mhz);
    }
    return 0;
}

// (2) Newton reciprocal vs correctly rounded division, d in [1, +inf): bit patterns 0x3f800000 .. 0x7f800000
__global__ __launch_bounds__(256) void rcp_check_kernel(unsigned long long* n_bad, unsigned* bad_list, unsigned long long* n_bad_non_ones)
{
    const unsigned long long total = 0x7f800000ull - 0x3f800000ull + 1ull;
    for (unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (unsigned long long)gridDim.x * 256) {
        const unsigned bits = 0x3f800000u + (unsigned)q;
        const float d = __uint_as_float(bits);
        const float ref = 1.f / d;                                   // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
        float r0;
        asm volatile("v_rcp_f32 %0, %1" : "=v"(r0) : "v"(d));
        const float e = __builtin_fmaf(-d, r0, 1.f);
        const float r1 = __builtin_fmaf(e, r0, r0);
        if (__float_as_uint(r1) != __float_as_uint(ref)) {
            const unsigned long long k = atomicAdd(n_bad, 1ull);
            if (k < 4096) bad_list[k] = bits;
            if ((bits & 0x7fffffu) != 0x7fffffu && bits < 0x7e800000u) atomicAdd(n_bad_non_ones, 1ull);
        }
    }
}

// (3) hardware bf16 narrowing vs the integer definition (oracle/np_x16.py narrow): every fp32 bit pattern
__device__ __forceinline__ unsigned short narrow_bf16_int(float f)
{
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__global__ __launch_bounds__(256) void bf16_check_kernel(unsigned long long* n_bad, unsigned long long* n_bad_non_nan, unsigned* bad_list)
{
    for (unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x; q < (1ull << 32); q += (unsigned long long)gridDim.x * 256) {
        const unsigned bits = (unsigned)q;
        const float f = __uint_as_float(bits);
        unsigned pk;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(pk) : "v"(f));
        const unsigned short hw = (unsigned short)(pk & 0xffffu), hw_hi = (unsigned short)(pk >> 16);
        const unsigned short sw = narrow_bf16_int(f);
        if (hw != sw || hw_hi != sw) {
            const unsigned long long k = atomicAdd(n_bad, 1ull);
            const bool is_nan = (bits & 0x7fffffffu) > 0x7f800000u;
            if (!is_nan) { const unsigned long long k2 = atomicAdd(n_bad_non_nan, 1ull); if (k2 < 64) bad_list[k2] = bits; }
            (void)k;
        }
    }
}

int main()
{
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount; const double mhz = prop.clockRate / 1e3;
    printf("device %s, %d CUs, %.0f MHz\n", prop.name, cus, mhz);
    float* d; CHK(hipMalloc(&d, (size_t)8 * cus * 256 * sizeof(float)));
    if (run_rate<0>("v_fma_f32", d, cus, mhz)) return 1;
    if (run_rate<7>("v_add_f32", d, cus, mhz)) return 1;
    if (run_rate<1>("v_pk_fma_f32", d, cus, mhz)) return 1;
    if (run_rate<6>("v_pk_mul_f32", d, cus, mhz)) return 1;
    if (run_rate<2>("v_rcp_f32", d, cus, mhz)) return 1;
    if (run_rate<3>("v_cvt_f32_f16", d, cus, mhz)) return 1;
    if (run_rate<4>("v_cndmask_b32", d, cus, mhz)) return 1;
    if (run_rate<5>("v_mov_b32_dpp", d, cus, mhz)) return 1;
    if (run_rate<8>("v_cmp+v_addc", d, cus, mhz)) return 1;

    unsigned long long* cnt; unsigned* lst;
    CHK(hipMalloc(&cnt, 4 * sizeof(unsigned long long))); CHK(hipMalloc(&lst, 4096 * sizeof(unsigned)));
    CHK(hipMemset(cnt, 0, 4 * sizeof(unsigned long long)));
    rcp_check_kernel<<<cus * 16, 256>>>(cnt, lst, cnt + 1);
    CHK(hipDeviceSynchronize());
    unsigned long long h[4]; static unsigned hl[4096];
    CHK(hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost)); CHK(hipMemcpy(hl, lst, sizeof(hl), hipMemcpyDeviceToHost));
    printf("rcp + one Newton step vs correctly rounded 1/d over d in [1, inf]: %llu of %llu differ; %llu of them with a mantissa that is not all ones and d < 2^126\n",
           h[0], 0x7f800000ull - 0x3f800000ull + 1ull, h[1]);
    for (unsigned long long k = 0; k < h[0] && k < 24; ++k) printf("   d bits 0x%08x (%.9g)\n", hl[k], (double)*reinterpret_cast<float*>(&hl[k]));

    CHK(hipMemset(cnt, 0, 4 * sizeof(unsigned long long)));
    bf16_check_kernel<<<cus * 16, 256>>>(cnt, cnt + 1, lst);
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost)); CHK(hipMemcpy(hl, lst, 64 * sizeof(unsigned), hipMemcpyDeviceToHost));
    printf("v_cvt_pk_bf16_f32 vs integer round-to-nearest-even over all 2^32 patterns: %llu differ, %llu of them not NaN\n", h[0], h[1]);
    for (unsigned long long k = 0; k < h[1] && k < 16; ++k) printf("   f bits 0x%08x\n", hl[k]);
    return 0;
}

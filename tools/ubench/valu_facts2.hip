// Micro-benchmark 2 (round 6): issue cost of the VALU instructions the fused neuron kernels are made of, in the forms the compiler emits them
// (compare -> VCC / SGPR pair, v_cndmask on either, carry ops, conversions, bit-field ops).  8 independent chains per wavefront, 1..8 wavefronts per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o valu_facts2 valu_facts2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define OPS(X) \
  X(0, "v_mul_f32", "v_mul_f32 %0, %0, %1", 1) \
  X(1, "v_max_f32", "v_max_f32 %0, %0, %1", 1) \
  X(2, "v_and_b32", "v_and_b32 %0, %0, %1", 1) \
  X(3, "v_lshlrev_b32", "v_lshlrev_b32 %0, 1, %0", 1) \
  X(4, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 1, %1", 1) \
  X(5, "v_bfe_u32", "v_bfe_u32 %0, %0, 2, 30", 1) \
  X(6, "v_cmp_le_f32 vcc (e32)", "v_cmp_le_f32 vcc, %1, %0\n\tv_add_f32 %0, %0, %1", 2) \
  X(7, "v_cmp_le_f32 s[20:21] (e64)", "v_cmp_le_f32 s[20:21], %1, %0\n\tv_add_f32 %0, %0, %1", 2) \
  X(8, "v_cndmask vcc after v_cmp vcc", "v_cmp_le_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc", 2) \
  X(9, "v_cndmask s[] after v_cmp s[]", "v_cmp_le_f32 s[20:21], %1, %0\n\tv_cndmask_b32 %0, %0, %1, s[20:21]", 2) \
  X(10, "v_cndmask s[22:23] const mask", "v_cndmask_b32 %0, %0, %1, s[22:23]", 1) \
  X(11, "v_cndmask vcc const mask", "v_cndmask_b32 %0, %0, %1, vcc", 1) \
  X(12, "v_addc_co_u32 vcc", "v_addc_co_u32 %0, vcc, %0, %0, vcc", 1) \
  X(13, "v_add_co_u32 vcc", "v_add_co_u32 %0, vcc, %0, %0", 1) \
  X(14, "v_add_u32", "v_add_u32 %0, %0, %0", 1) \
  X(15, "v_cvt_f32_f16 sdwa hi", "v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", 1) \
  X(16, "v_cvt_pk_f16_f32", "v_cvt_pk_f16_f32 %0, %0, %1", 1) \
  X(17, "v_cvt_pk_bf16_f32", "v_cvt_pk_bf16_f32 %0, %0, %1", 1) \
  X(18, "v_cvt_f16_f32", "v_cvt_f16_f32 %0, %0", 1) \
  X(19, "v_bcnt_u32_b32", "v_bcnt_u32_b32 %0, %0, %1", 1) \
  X(20, "v_med3_f32", "v_med3_f32 %0, %0, %1, %1", 1) \
  X(21, "v_fma_f32 2 chains interleaved w/ v_mov", "v_fma_f32 %0, %0, %1, %1\n\tv_mov_b32 %0, %0", 2) \
  X(22, "v_div_fixup_f32", "v_div_fixup_f32 %0, %0, %1, %1", 1) \
  X(23, "v_div_scale_f32", "v_div_scale_f32 %0, vcc, %0, %1, %1", 1) \
  X(24, "v_div_fmas_f32", "v_div_fmas_f32 %0, %0, %1, %1", 1) \
  X(25, "v_bfi_b32", "v_bfi_b32 %0, %0, %1, %1", 1) \
  X(26, "v_sub_f32", "v_sub_f32 %0, %0, %1", 1) \
  X(27, "v_cmp_class_f32 vcc", "v_cmp_class_f32 vcc, %0, %1\n\tv_add_f32 %0, %0, %1", 2) \
  X(28, "v_readfirstlane+v_mov s", "v_readfirstlane_b32 s24, %0\n\tv_add_f32 %0, s24, %0", 2) \
  X(29, "v_mov_b32_dpp row_shr:1", "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", 1) \
  X(30, "v_add_f32_dpp quad_perm", "v_add_f32_dpp %0, %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf", 1) \
  X(31, "v_pk_add_f32 (2 regs)", "v_pk_add_f32 %2, %2, %3", 1) \
  X(32, "v_exp_f32", "v_exp_f32 %0, %0", 1)

typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float seed)
{
    float a[8]; f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = (f2){a[i], a[i]}; }
    const float m = 1.0000001f; const f2 m2 = {m, m};
    asm volatile("s_mov_b64 s[22:23], 0x55555555\n\ts_mov_b64 vcc, 0x33333333" ::: "s22", "s23", "vcc");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#define X(ID, NAME, ASM, N) if (OP == ID) asm volatile(ASM : "+v"(a[i]) : "v"(m), "v"(p[i]), "v"(m2) : "vcc", "s20", "s21", "s24");
                OPS(X)
#undef X
                if (OP == 31) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP> int run_rate(const char* name, int n_instr, float* d, int cus, double mhz)
{
    const int iters = 2000;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    printf("%-42s", name);
    for (int wpc : {1, 2, 4, 8}) {
        rate_kernel<OP><<<wpc * cus, 256>>>(d, 10, 1.f);
        CHK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(e0));
            rate_kernel<OP><<<wpc * cus, 256>>>(d, iters, 1.f);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double groups_per_simd = (double)iters * 64 * wpc;
        printf("  w%d: %6.2f", wpc, best * 1e-3 * mhz * 1e6 / groups_per_simd);
    }
    printf("   cycles per group of %d instruction(s) per SIMD\n", n_instr);
    return 0;
}

int main()
{
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount; const double mhz = prop.clockRate / 1e3;
    printf("%d CUs, %.0f MHz\n", cus, mhz);
    float* d; CHK(hipMalloc(&d, (size_t)8 * cus * 256 * sizeof(float)));
#define X(ID, NAME, ASM, N) if (run_rate<ID>(NAME, N, d, cus, mhz)) return 1;
    OPS(X)
#undef X
    return 0;
}

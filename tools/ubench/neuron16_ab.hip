// A/B + bit-equality harness of the round-6 16-bit neuron kernels (stereospike_amd/csrc/ss_neuron16_v2.hpp) against the shipped library's entry points
// (dlopen of stereospike_amd/lib/libss_neuron.so — whatever forms that build dispatches to), at BASELINE config 5 / config 3 layer shapes.
//   neuron16_ab <T> <B> <dtype 1=f16 2=bf16> [C=32] [H=260] [W=346]
// Build: see tools/r06/build_ubench.sh (same flags as the library).
#include "../../stereospike_amd/csrc/ss_neuron16_v2.hpp"
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// second pass of the counters (the library's cnt_finish_kernel lives in ss_neuron.hip)
__global__ __launch_bounds__(SS_BLOCK) void cnt_finish2_kernel(const unsigned* __restrict__ ws, int n, unsigned long long* nnz)
{
    __shared__ unsigned long long s[2][SS_BLOCK];
    unsigned long long a0 = 0, a1 = 0;
    for (int i = threadIdx.x; i < n; i += SS_BLOCK) { a0 += ws[2 * i]; a1 += ws[2 * i + 1]; }
    s[0][threadIdx.x] = a0; s[1][threadIdx.x] = a1;
    __syncthreads();
    for (int o = SS_BLOCK / 2; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) { s[0][threadIdx.x] += s[0][threadIdx.x + o]; s[1][threadIdx.x] += s[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { if (s[0][0]) atomicAdd(&nnz[0], s[0][0]); if (s[1][0]) atomicAdd(&nnz[1], s[1][0]); }
}

__device__ __forceinline__ unsigned hash32(unsigned long long x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (unsigned)x;
}
__device__ __forceinline__ float approx_normal(unsigned long long idx, unsigned seed)
{
    const unsigned a = hash32(idx * 2 + ((unsigned long long)seed << 40)), b = hash32(idx * 2 + 1 + ((unsigned long long)seed << 40));
    const float u = ((a & 0xffff) + (a >> 16) + (b & 0xffff) + (b >> 16)) * (1.f / 65536.f) - 2.f;      // sum of 4 uniforms: std 0.577
    return u * 1.7320508f;
}
template <int DT> __global__ void fill16(unsigned short* p, long long n, float std, unsigned seed)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = narrow<DT>(approx_normal(i, seed) * std);
}
__global__ void fill32(float* p, long long n, float std, unsigned seed)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = approx_normal(i, seed) * std;
}
__global__ void fill_codes(unsigned* p, long long n, unsigned seed)       // packed skip codes 0..2
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        unsigned w = 0;
        for (int e = 0; e < 16; ++e) { const unsigned r = hash32(i * 16 + e + ((unsigned long long)seed << 40)) % 10u; w |= (r < 7 ? 0u : (r < 9 ? 1u : 2u)) << (2 * e); }
        p[i] = w;
    }
}
__global__ void diff_kernel(const unsigned* a, const unsigned* b, long long n_words, unsigned long long* cnt)
{
    unsigned long long c = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (long long)gridDim.x * 256) c += (a[i] != b[i]);
    if (c) atomicAdd(cnt, c);
}

static unsigned long long* g_cnt;
static unsigned long long diff(const void* a, const void* b, long long bytes)
{
    CHK(hipMemset(g_cnt, 0, 8));
    diff_kernel<<<4096, 256>>>((const unsigned*)a, (const unsigned*)b, bytes / 4, g_cnt);
    unsigned long long h; CHK(hipMemcpy(&h, g_cnt, 8, hipMemcpyDeviceToHost));
    return h;
}

template <typename F> static float time_us(F&& fn, int reps = 7)
{
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    fn(); CHK(hipDeviceSynchronize());
    std::vector<float> v;
    for (int r = 0; r < reps; ++r) {
        CHK(hipEventRecord(e0)); fn(); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); v.push_back(ms * 1e3f);
    }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

typedef int (*fwd_ex_t)(const ss_neuron_fwd_desc*, void*);
typedef int (*bwd_lr_t)(const void*, const float*, const float*, int, int, void*, const float*, const void*, const float*, void*, float*, float*, float*, int, long long,
                        float, int, float, const float*, float, float, int, float, int, int, void*);
typedef int (*bwd_fork_t)(const void*, const void*, void*, const float*, const void*, const float*, void*, float*, float*, float*, int, long long, float, int, float,
                          const float*, float, float, int, float, int, int, void*);
typedef int (*bwd_rc_t)(const void*, const float*, const void*, const float*, void*, float*, float*, float*, int, long long, float, int, float, const float*, float, float,
                        int, float, int, int, void*);

template <int DT, int TS> int run(int B, int C, int H, int W, void* lib)
{
    const long long N = (long long)B * C * H * W, rows = N / C, TN = (long long)TS * N;
    printf("=== T %d B %d dtype %s C %d %dx%d: N %lld, %.1f M updates\n", TS, B, DT == SS_DT_F16 ? "f16" : "bf16", C, H, W, N, TN / 1e6);
    fwd_ex_t fwd_ex = (fwd_ex_t)dlsym(lib, "ss_neuron_fwd_ex");
    bwd_lr_t bwd_lr = (bwd_lr_t)dlsym(lib, "ss_neuron_bwd_fork_lr_x16");
    bwd_fork_t bwd_fork = (bwd_fork_t)dlsym(lib, "ss_neuron_bwd_fork_x16");
    bwd_rc_t bwd_rc = (bwd_rc_t)dlsym(lib, "ss_neuron_bwd_rc_x16");
    if (!fwd_ex || !bwd_lr || !bwd_fork || !bwd_rc) { printf("symbols missing\n"); return 1; }
    unsigned short *x, *g, *g2, *gx0, *gx1, *gs0, *gs1;
    unsigned *skp, *op0, *op1, *cws; float *lrp, *lrw, *vl0, *vl1; unsigned long long* nnz;
    CHK(hipMalloc(&x, TN * 2)); CHK(hipMalloc(&g, TN * 2)); CHK(hipMalloc(&g2, TN * 2));
    CHK(hipMalloc(&gx0, TN * 2)); CHK(hipMalloc(&gx1, TN * 2)); CHK(hipMalloc(&gs0, TN * 2)); CHK(hipMalloc(&gs1, TN * 2));
    CHK(hipMalloc(&skp, TN / 4)); CHK(hipMalloc(&op0, TN / 4)); CHK(hipMalloc(&op1, TN / 4));
    CHK(hipMalloc(&lrp, (long long)TS * rows * 9 * 4)); CHK(hipMalloc(&lrw, 9 * C * 4));
    CHK(hipMalloc(&vl0, N * 4)); CHK(hipMalloc(&vl1, N * 4)); CHK(hipMalloc(&nnz, 64)); CHK(hipMalloc(&cws, (2 * (N / 256 + 2)) * 4));
    CHK(hipMalloc(&g_cnt, 8));
    unsigned* redo_flag; CHK(hipMalloc(&redo_flag, 4));
    fill16<DT><<<8192, 256>>>(x, TN, 0.06f, 1); fill16<DT><<<8192, 256>>>(g, TN, 1e-3f, 2); fill16<DT><<<8192, 256>>>(g2, TN, 1e-3f, 3);
    fill32<<<4096, 256>>>(lrp, (long long)TS * rows * 9, 1e-3f, 4); fill32<<<1, 256>>>(lrw, 9 * C, 0.3f, 5);
    fill_codes<<<4096, 256>>>(skp, TN / 16, 6);
    CHK(hipDeviceSynchronize());
    const float scale = 10.f, tau = 2.f, v_th = 1.f, v_reset = 0.f, alpha = 2.f;
    const double fwd_bytes = 2.25 * TN, fwd_bytes_skip = 2.5 * TN;

    // ---------------- forward: packed out (+ packed skip), counters through the workspace
    for (int skip = 0; skip < 2; ++skip) {
        ss_neuron_fwd_desc d; memset(&d, 0, sizeof(d));
        d.size = sizeof(d); d.act_dtype = DT; d.x_seq = x; d.skip_packed = skip ? skp : nullptr; d.out_packed = op0; d.v_last = vl0; d.nnz = nnz; d.cnt_ws = cws;
        d.T = TS; d.N = N; d.scale = scale; d.kind = SS_KIND_IF; d.tau = tau; d.v_th = v_th; d.v_reset = v_reset;
        CHK(hipMemset(nnz, 0, 16));
        if (fwd_ex(&d, nullptr) != 0) { printf("fwd_ex failed\n"); return 1; }
        unsigned long long n0[2]; CHK(hipMemcpy(n0, nnz, 16, hipMemcpyDeviceToHost));
        const float t_base = time_us([&] { fwd_ex(&d, nullptr); });
        Fwd16Args a{x, nullptr, nullptr, nullptr, nullptr, vl1, nnz, TS, N, scale, tau, v_th, v_reset, nullptr, cws, skip ? skp : nullptr, op1};
        const int grid_cap = getenv("SS_AB_FWD_GRID") ? atoi(getenv("SS_AB_FWD_GRID")) : kMaxGrid;      // A/B: a bounded grid with a grid-stride loop
        const int grid = grid_for(N / 8, grid_cap);
        auto launch_new = [&](bool with_v) {
            Fwd16Args b = a; if (!with_v) b.v_last = nullptr;
            if (skip) hipLaunchKernelGGL((neuron_fwd16_pk8_kernel<SS_KIND_IF, DT, TS, true, false>), dim3(grid), dim3(kBlock), 0, 0, b);
            else hipLaunchKernelGGL((neuron_fwd16_pk8_kernel<SS_KIND_IF, DT, TS, false, false>), dim3(grid), dim3(kBlock), 0, 0, b);
            hipLaunchKernelGGL(cnt_finish2_kernel, dim3(1), dim3(kBlock), 0, 0, cws, grid, nnz);
        };
        CHK(hipMemset(nnz, 0, 16)); CHK(hipMemset(op1, 0xff, TN / 4)); CHK(hipMemset(vl1, 0xff, N * 4));
        launch_new(true); CHK(hipDeviceSynchronize());
        unsigned long long n1[2]; CHK(hipMemcpy(n1, nnz, 16, hipMemcpyDeviceToHost));
        const unsigned long long d_out = diff(op0, op1, TN / 4), d_v = diff(vl0, vl1, N * 4);
        const float t_new = time_us([&] { launch_new(true); }), t_new_nov = time_us([&] { launch_new(false); });
        const double bytes = skip ? fwd_bytes_skip : fwd_bytes;
        printf("fwd  skip %d: base %8.1f us (%.3f of 8 TB/s)   new %8.1f us (%.3f)   new without v_last %8.1f us (%.3f)   diff words out %llu v %llu   nnz base %llu/%llu new %llu/%llu (spike rate %.3f)\n",
               skip, t_base, bytes / t_base / 8e6, t_new, bytes / t_new / 8e6, t_new_nov, bytes / t_new_nov / 8e6, d_out, d_v, n0[0], n0[1], n1[0], n1[1], (double)n0[0] / TN);
    }

    // ---------------- backward: low-rank pair + dense first gradient (the head stages), two dense gradients, one gradient
    Bwd16Args ba{g, nullptr, nullptr, nullptr, gx1, nullptr, nullptr, TS, N, scale, tau, v_th, v_reset, alpha, nullptr, 1};
    const int pair_x4 = (rows % 4 == 0) ? 1 : 0;
#define NEW_BWD(VEC, NSEG, G2, LR, GS, G2P, LRP) NEW_BWDW(VEC, NSEG, G2, LR, GS, G2P, LRP, 1)
#define NEW_BWDW(VEC, NSEG, G2, LR, GS, G2P, LRP, WV) NEW_BWDS(VEC, NSEG, G2, LR, GS, G2P, LRP, WV, false)
#define NEW_BWDS(VEC, NSEG, G2, LR, GS, G2P, LRP, WV, SUMF) do { \
        const int grid_ = grid_for(N / VEC, getenv("SS_AB_BWD_GRID") ? atoi(getenv("SS_AB_BWD_GRID")) : kMaxGridBwd); \
        (void)hipMemsetAsync(redo_flag, 0, 4, 0); \
        hipLaunchKernelGGL((neuron_bwd16_seg_kernel<SS_KIND_IF, SS_SG_ATAN, DT, TS, VEC, NSEG, G2, LR, WV, true, SUMF, 0>), dim3(grid_), dim3(kBlock), LR ? bwd16_seg_lds_bytes(TS, VEC, C) : 0, 0, ba, x, G2P, GS, LRP, lrw, C, \
                           (pair_x4 && ((64 * VEC / C) % 4 == 0)) ? 1 : 0, redo_flag); \
        hipLaunchKernelGGL((neuron_bwd16_seg_kernel<SS_KIND_IF, SS_SG_ATAN, DT, TS, VEC, NSEG, G2, LR, 2, true, SUMF, 1>), dim3(grid_ < 2048 ? grid_ : 2048), dim3(kBlock), LR ? bwd16_seg_lds_bytes(TS, VEC, C) : 0, 0, ba, x, G2P, GS, LRP, lrw, C, \
                           (pair_x4 && ((64 * VEC / C) % 4 == 0)) ? 1 : 0, redo_flag); } while (0)
#define AB(NAME, BASECALL, BYTES, HAS_SUM, ...) do { \
        CHK(hipMemset(gx0, 0xff, TN * 2)); CHK(hipMemset(gs0, 0xff, TN * 2)); \
        if ((BASECALL) != 0) { printf("base call failed\n"); return 1; } \
        const float tb_ = time_us([&] { BASECALL; }); \
        CHK(hipMemset(gx1, 0xee, TN * 2)); CHK(hipMemset(gs1, 0xee, TN * 2)); \
        __VA_ARGS__; CHK(hipDeviceSynchronize()); { hipError_t le_ = hipGetLastError(); if (le_ != hipSuccess) { printf("%s launch error %s\n", NAME, hipGetErrorString(le_)); return 1; } } \
        const unsigned long long dx_ = diff(gx0, gx1, TN * 2), ds_ = HAS_SUM ? diff(gs0, gs1, TN * 2) : 0; \
        const float tn_ = time_us([&] { __VA_ARGS__; }); \
        printf("bwd %-34s base %8.1f us (%.3f)   new %8.1f us (%.3f of 8 TB/s)   diff words g_x %llu g_sum %llu\n", NAME, tb_, (BYTES) / tb_ / 8e6, tn_, (BYTES) / tn_ / 8e6, dx_, ds_); } while (0)
    const double b_lr = (6.0 + 36.0 / C) * TN, b_lr_sum = (8.0 + 36.0 / C) * TN, b_fork = 8.0 * TN, b_fork_sum = 10.0 * TN, b_rc = 6.0 * TN;
#define BASE_LR(SUM) bwd_lr(g, lrp, lrw, 9, C, SUM, nullptr, x, nullptr, gx0, nullptr, nullptr, nullptr, TS, N, scale, SS_KIND_IF, tau, nullptr, v_th, v_reset, SS_SG_ATAN, alpha, 1, DT, nullptr)
#define BASE_FORK(SUM) bwd_fork(g, g2, SUM, nullptr, x, nullptr, gx0, nullptr, nullptr, nullptr, TS, N, scale, SS_KIND_IF, tau, nullptr, v_th, v_reset, SS_SG_ATAN, alpha, 1, DT, nullptr)
#define BASE_RC() bwd_rc(g, nullptr, x, nullptr, gx0, nullptr, nullptr, nullptr, TS, N, scale, SS_KIND_IF, tau, nullptr, v_th, v_reset, SS_SG_ATAN, alpha, 1, DT, nullptr)
    if constexpr (TS > 5) {
        AB("lr  V4 S2 w2", BASE_LR(nullptr), b_lr, false, NEW_BWDW(4, 2, true, true, nullptr, nullptr, lrp, 2));
        AB("lr  V4 S2 w3", BASE_LR(nullptr), b_lr, false, NEW_BWDW(4, 2, true, true, nullptr, nullptr, lrp, 3));
        AB("lr  V4 S3 w3", BASE_LR(nullptr), b_lr, false, NEW_BWDW(4, 3, true, true, nullptr, nullptr, lrp, 3));
        AB("lr  V4 S3 w4", BASE_LR(nullptr), b_lr, false, NEW_BWDW(4, 3, true, true, nullptr, nullptr, lrp, 4));
        AB("lr  V4 S2 w4", BASE_LR(nullptr), b_lr, false, NEW_BWDW(4, 2, true, true, nullptr, nullptr, lrp, 4));
        AB("lr  V2 S1 w1", BASE_LR(nullptr), b_lr, false, NEW_BWDW(2, 1, true, true, nullptr, nullptr, lrp, 1));
        AB("lr  V2 S1 w4", BASE_LR(nullptr), b_lr, false, NEW_BWDW(2, 1, true, true, nullptr, nullptr, lrp, 4));
        AB("lr  V2 S1 w5", BASE_LR(nullptr), b_lr, false, NEW_BWDW(2, 1, true, true, nullptr, nullptr, lrp, 5));
        AB("lr  V2 S2 w4", BASE_LR(nullptr), b_lr, false, NEW_BWDW(2, 2, true, true, nullptr, nullptr, lrp, 4));
        AB("lr  V2 S2 w6", BASE_LR(nullptr), b_lr, false, NEW_BWDW(2, 2, true, true, nullptr, nullptr, lrp, 6));
        AB("lr+sum V4 S2 w3", BASE_LR(gs0), b_lr_sum, true, NEW_BWDS(4, 2, true, true, gs1, nullptr, lrp, 3, true));
        AB("fork V4 S2 w1", BASE_FORK(nullptr), b_fork, false, NEW_BWDW(4, 2, true, false, nullptr, g2, nullptr, 1));
        AB("fork V4 S2 w4", BASE_FORK(nullptr), b_fork, false, NEW_BWDW(4, 2, true, false, nullptr, g2, nullptr, 4));
        AB("fork V2 S1 w1", BASE_FORK(nullptr), b_fork, false, NEW_BWDW(2, 1, true, false, nullptr, g2, nullptr, 1));
        AB("fork V2 S1 w5", BASE_FORK(nullptr), b_fork, false, NEW_BWDW(2, 1, true, false, nullptr, g2, nullptr, 5));
        AB("fork+sum V4 S2 w4", BASE_FORK(gs0), b_fork_sum, true, NEW_BWDS(4, 2, true, false, gs1, g2, nullptr, 4, true));
        AB("rc  V4 S2 w1", BASE_RC(), b_rc, false, NEW_BWDW(4, 2, false, false, nullptr, nullptr, nullptr, 1));
        AB("rc  V4 S2 w4", BASE_RC(), b_rc, false, NEW_BWDW(4, 2, false, false, nullptr, nullptr, nullptr, 4));
        AB("rc  V2 S1 w1", BASE_RC(), b_rc, false, NEW_BWDW(2, 1, false, false, nullptr, nullptr, nullptr, 1));
    } else {
        AB("lr  V8 S2 w3", BASE_LR(nullptr), b_lr, false, NEW_BWDW(8, 2, true, true, nullptr, nullptr, lrp, 3));
        AB("lr  V8 S1 w2", BASE_LR(nullptr), b_lr, false, NEW_BWDW(8, 1, true, true, nullptr, nullptr, lrp, 2));
        AB("lr  V4 S1 w1", BASE_LR(nullptr), b_lr, false, NEW_BWDW(4, 1, true, true, nullptr, nullptr, lrp, 1));
        AB("lr  V4 S1 w3", BASE_LR(nullptr), b_lr, false, NEW_BWDW(4, 1, true, true, nullptr, nullptr, lrp, 3));
        AB("lr  V4 S1 w4", BASE_LR(nullptr), b_lr, false, NEW_BWDW(4, 1, true, true, nullptr, nullptr, lrp, 4));
        AB("lr  V4 S2 w4", BASE_LR(nullptr), b_lr, false, NEW_BWDW(4, 2, true, true, nullptr, nullptr, lrp, 4));
        AB("lr+sum V4 S1 w3", BASE_LR(gs0), b_lr_sum, true, NEW_BWDS(4, 1, true, true, gs1, nullptr, lrp, 3, true));
        AB("lr+sum V8 S2 w3", BASE_LR(gs0), b_lr_sum, true, NEW_BWDS(8, 2, true, true, gs1, nullptr, lrp, 3, true));
        AB("fork V8 S1 w3", BASE_FORK(nullptr), b_fork, false, NEW_BWDW(8, 1, true, false, nullptr, g2, nullptr, 3));
        AB("fork V8 S1 w4", BASE_FORK(nullptr), b_fork, false, NEW_BWDW(8, 1, true, false, nullptr, g2, nullptr, 4));
        AB("fork V4 S1 w1", BASE_FORK(nullptr), b_fork, false, NEW_BWDW(4, 1, true, false, nullptr, g2, nullptr, 1));
        AB("fork+sum V8 S1 w3", BASE_FORK(gs0), b_fork_sum, true, NEW_BWDS(8, 1, true, false, gs1, g2, nullptr, 3, true));
        AB("rc  V8 S1 w4", BASE_RC(), b_rc, false, NEW_BWDW(8, 1, false, false, nullptr, nullptr, nullptr, 4));
        AB("rc  V4 S1 w1", BASE_RC(), b_rc, false, NEW_BWDW(4, 1, false, false, nullptr, nullptr, nullptr, 1));
    }
    return 0;
}

int main(int argc, char** argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 10, B = argc > 2 ? atoi(argv[2]) : 32, dt = argc > 3 ? atoi(argv[3]) : 1;
    const int C = argc > 4 ? atoi(argv[4]) : 32, H = argc > 5 ? atoi(argv[5]) : 260, W = argc > 6 ? atoi(argv[6]) : 346;
    const char* libpath = getenv("SS_LIB") ? getenv("SS_LIB") : "stereospike_amd/lib/libss_neuron.so";
    void* lib = dlopen(libpath, RTLD_NOW);
    if (!lib) { printf("dlopen %s: %s\n", libpath, dlerror()); return 1; }
    if (T == 10 && dt == SS_DT_F16) return run<SS_DT_F16, 10>(B, C, H, W, lib);
    if (T == 10 && dt == SS_DT_BF16) return run<SS_DT_BF16, 10>(B, C, H, W, lib);
    if (T == 5 && dt == SS_DT_F16) return run<SS_DT_F16, 5>(B, C, H, W, lib);
    if (T == 5 && dt == SS_DT_BF16) return run<SS_DT_BF16, 5>(B, C, H, W, lib);
    printf("unsupported T / dtype\n");
    return 1;
}

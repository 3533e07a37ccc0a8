// ss_neuron_bwd16.hip — surrogate-gradient backward of the fused neuron layers on 16-bit activations (fp16 / bf16 in HBM, fp32 arithmetic) + C-ABI.
// Design notes: see ss_neuron.hip.
#include "ss_common.hpp"
#include "ss_neuron16_v2.hpp"

namespace {

template <int KIND, int SG, int DT, int VEC>
__global__ __launch_bounds__(kBlock) void neuron_bwd16_kernel(Bwd16Args a)
{
    const long long NV = a.N / VEC;
    const float k = (KIND == SS_KIND_PLIF) ? *a.k : 0.f;
    const float scale = a.scale, tau = a.tau, v_th = a.v_th, v_reset = a.v_reset, alpha = a.alpha;
    const float c_atan = (float)(M_PI / 2.0 * (double)alpha);
    const float half_alpha = (float)((double)alpha / 2.0);
    const bool detach = a.detach_reset != 0;
    const bool want_gk = (KIND == SS_KIND_PLIF) && a.g_k_partials != nullptr;
    double acc_k = 0.0;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < NV; i += (long long)gridDim.x * kBlock) {
        float gv[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) gv[e] = a.g_v_last ? a.g_v_last[i * VEC + e] : 0.f;
        for (int t = a.T - 1; t >= 0; --t) {
            const long long base = ((long long)t * NV + i) * VEC;
            unsigned short gb[VEC], xb[VEC];
            float h[VEC], hp[VEC];
            if constexpr (VEC == 8) {
                const u16x8 g8 = *reinterpret_cast<const u16x8*>(a.g_out_seq + base);
                const f4 h0 = *reinterpret_cast<const f4*>(a.h_seq + base), h1 = *reinterpret_cast<const f4*>(a.h_seq + base + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) { gb[e] = g8[e]; h[e] = e < 4 ? h0[e & 3] : h1[e & 3]; }
            } else { gb[0] = a.g_out_seq[base]; h[0] = a.h_seq[base]; }
            if (want_gk) {
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    hp[e] = t > 0 ? a.h_seq[base - a.N + e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xh = h[e] - v_th;
                const float z = heaviside(xh);
                float g_s = widen<DT>(gb[e]);
                if (!detach) g_s = g_s + (gv[e] * v_reset - gv[e] * h[e]);
                const float g_h = surrogate_grad<SG>(xh, alpha, c_atan, half_alpha, g_s) + gv[e] * (1.f - z);
                float g_x;
                if (KIND == SS_KIND_IF) { g_x = g_h; gv[e] = g_h; }
                else if (KIND == SS_KIND_LIF) { g_x = g_h / tau; gv[e] = g_h - g_x; }
                else {
                    g_x = g_h * k; gv[e] = g_h - g_x;
                    if (want_gk) {
                        float v_prev;
                        if (t == 0) v_prev = a.v_init ? a.v_init[i * VEC + e] : v_reset;
                        else { const float zp = heaviside(hp[e] - v_th); v_prev = (1.f - zp) * hp[e] + zp * v_reset; }
                        acc_k += (double)g_h * (double)((h[e] - v_prev) / k);
                    }
                }
                xb[e] = narrow<DT>(g_x * scale);
            }
            if constexpr (VEC == 8) {
                u16x8 xv;
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[e] = xb[e];
                *reinterpret_cast<u16x8*>(a.g_x_seq + base) = xv;
            } else a.g_x_seq[base] = xb[0];
        }
        if (a.g_v_init) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) a.g_v_init[i * VEC + e] = gv[e];
        }
    }
    if (want_gk) gk_epilogue(acc_k, a.g_k_partials);   // wave-uniform
}

// Backward with h recomputed from the 16-bit layer input (ss_neuron_bwd_rc_x16): compile-time T, a lane owns VEC consecutive
// neurons (8 = one 16-B load per step for T <= 5; 4 for longer sequences to keep h_0..h_{T-1} (fp32) in registers at full occupancy).
#ifndef SS_RC16_V5
#define SS_RC16_V5 4
#endif
#ifndef SS_RC16_V10
#define SS_RC16_V10 2
#endif

// G2: a second consumer's 16-bit gradient is added on load (fp32 sum of the two widened values — not rounded to 16 bits in between,
// unlike autograd's accumulation); g_sum_seq (nullable) receives that sum narrowed once: dL/dskip of a stage that has both.
// LR (round 5; with G2): the second gradient arrives as the prediction head's rank-9 pair, exactly as in ss_neuron_bwd_fork_lr_f32 — g2[t][n] =
// sum_j lr_p[(t * N / C + n / C) * 9 + j] * lr_w[j * C + n % C], both fp32 (taps ascending, multiply and add rounded separately) — and is added to the
// widened first gradient in fp32; the first gradient may be absent (a.g_out_seq == NULL: the full-resolution stage has no other consumer).
constexpr int kLr16MaxC = 512, kLr16Rank = 9;
template <int KIND, int SG, int DT, int TS, int VEC, bool G2 = false, bool LR = false>
__global__ __launch_bounds__(kBlock) void neuron_bwd16_rc_kernel(Bwd16Args a, const unsigned short* __restrict__ x_seq,
                                                                 const unsigned short* __restrict__ g_out2_seq, unsigned short* __restrict__ g_sum_seq,
                                                                 const float* __restrict__ lr_p = nullptr, const float* __restrict__ lr_w = nullptr, int lr_C = 0)
{
    static_assert(!LR || (G2 && VEC > 1), "low-rank second gradient: vector lanes of the forked form");
    typedef typename U16Vec<VEC>::type uvec_t;
    __shared__ __attribute__((aligned(16))) float lr_ws[LR ? kLr16Rank * kLr16MaxC : 4];
    int lr_c0 = 0;
    if constexpr (LR) {
        for (int q = threadIdx.x; q < kLr16Rank * lr_C; q += kBlock) lr_ws[q] = lr_w[q];
        lr_c0 = (int)((threadIdx.x * (unsigned)VEC) % (unsigned)lr_C);        // the lane's channels are the same in every trip (kBlock * VEC is a multiple of C: host)
        __syncthreads();
    }
    const long long NV = a.N / VEC;
    const float k = (KIND == SS_KIND_PLIF) ? *a.k : 0.f;
    const float scale = a.scale, tau = a.tau, v_th = a.v_th, v_reset = a.v_reset, alpha = a.alpha;
    const float c_atan = (float)(M_PI / 2.0 * (double)alpha);
    const float half_alpha = (float)((double)alpha / 2.0);
    const bool detach = a.detach_reset != 0;
    const bool want_gk = (KIND == SS_KIND_PLIF) && a.g_k_partials != nullptr;
    double acc_k = 0.0;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < NV; i += (long long)gridDim.x * kBlock) {
        uvec_t xs[TS], gs[TS];
#pragma unroll
        for (int t = 0; t < TS; ++t) xs[t] = load_stream(reinterpret_cast<const uvec_t*>(x_seq + ((long long)t * NV + i) * VEC));
        const bool has_g1 = !LR || a.g_out_seq != nullptr;    // wave-uniform
        if (has_g1) {
#pragma unroll
            for (int t = TS - 1; t >= 0; --t) gs[t] = load_stream(reinterpret_cast<const uvec_t*>(a.g_out_seq + ((long long)t * NV + i) * VEC));
        }
        uvec_t g2[(G2 && !LR) ? TS : 1];
        if constexpr (G2 && !LR) {
#pragma unroll
            for (int t = TS - 1; t >= 0; --t) g2[t] = load_stream(reinterpret_cast<const uvec_t*>(g_out2_seq + ((long long)t * NV + i) * VEC));
        }
        float lracc[LR ? TS : 1][LR ? VEC : 1];
        if constexpr (LR) {
            const long long rows = a.N / lr_C;
            const float* pp = lr_p + ((i * VEC) / lr_C) * kLr16Rank;
            int c0v = lr_c0;
            asm volatile("" : "+v"(c0v));                  // keep the LDS reads inside the loop
            if constexpr (TS <= 5) {                       // all T x 9 pair values up front (45 registers), the nine weight slices read once
                float pj[TS][kLr16Rank];
#pragma unroll
                for (int t = TS - 1; t >= 0; --t)
#pragma unroll
                    for (int j = 0; j < kLr16Rank; ++j) pj[t][j] = pp[(long long)t * rows * kLr16Rank + j];
#pragma unroll
                for (int j = 0; j < kLr16Rank; ++j) {
                    float wj[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) wj[e] = lr_ws[j * lr_C + c0v + e];
#pragma unroll
                    for (int t = TS - 1; t >= 0; --t)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) lracc[t][e] = (j == 0) ? pj[t][0] * wj[e] : lracc[t][e] + pj[t][j] * wj[e];
                }
            } else {                                       // longer sequences: the lane's 9 x VEC weights in registers, the pair walked step by step (same op order)
                float wj[kLr16Rank][VEC];
#pragma unroll
                for (int j = 0; j < kLr16Rank; ++j)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) wj[j][e] = lr_ws[j * lr_C + c0v + e];
#pragma unroll
                for (int t = TS - 1; t >= 0; --t) {
                    float pt[kLr16Rank];
#pragma unroll
                    for (int j = 0; j < kLr16Rank; ++j) pt[j] = pp[(long long)t * rows * kLr16Rank + j];
#pragma unroll
                    for (int j = 0; j < kLr16Rank; ++j)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) lracc[t][e] = (j == 0) ? pt[0] * wj[0][e] : lracc[t][e] + pt[j] * wj[j][e];
                }
            }
        }
        float v0[VEC], gv[VEC], h[TS][VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            v0[e] = a.v_init ? a.v_init[i * VEC + e] : v_reset;
            gv[e] = a.g_v_last ? a.g_v_last[i * VEC + e] : 0.f;
        }
        {
            float v[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = v0[e];
#pragma unroll
            for (int t = 0; t < TS; ++t)
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    unsigned short xb;
                    if constexpr (VEC == 1) xb = xs[t]; else xb = xs[t][e];
                    const float hh = charge<KIND>(v[e], widen<DT>(xb) * scale, tau, k, v_reset);
                    const float z = heaviside(hh - v_th);
                    v[e] = (1.f - z) * hh + z * v_reset;
                    h[t][e] = hh;
                }
        }
#pragma unroll
        for (int t = TS - 1; t >= 0; --t) {
            uvec_t xv, sumv;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                unsigned short gb = 0;
                if (has_g1) { if constexpr (VEC == 1) gb = gs[t]; else gb = gs[t][e]; }
                const float he = h[t][e];
                const float xh = he - v_th;
                const float z = heaviside(xh);
                float g_s = has_g1 ? widen<DT>(gb) : 0.f;
                if constexpr (LR) {
                    g_s = has_g1 ? g_s + lracc[t][e] : lracc[t][e];
                    sumv[e] = narrow<DT>(g_s);
                } else if constexpr (G2) {
                    unsigned short gb2;
                    if constexpr (VEC == 1) gb2 = g2[t]; else gb2 = g2[t][e];
                    g_s = g_s + widen<DT>(gb2);
                    const unsigned short sb = narrow<DT>(g_s);
                    if constexpr (VEC == 1) sumv = sb; else sumv[e] = sb;
                }
                if (!detach) g_s = g_s + (gv[e] * v_reset - gv[e] * he);
                const float g_h = surrogate_grad<SG>(xh, alpha, c_atan, half_alpha, g_s) + gv[e] * (1.f - z);
                float g_x;
                if (KIND == SS_KIND_IF) { g_x = g_h; gv[e] = g_h; }
                else if (KIND == SS_KIND_LIF) { g_x = g_h / tau; gv[e] = g_h - g_x; }
                else {
                    g_x = g_h * k; gv[e] = g_h - g_x;
                    if (want_gk) {
                        float v_prev;
                        if (t == 0) v_prev = v0[e];
                        else { const float hp = h[t > 0 ? t - 1 : 0][e]; const float zp = heaviside(hp - v_th); v_prev = (1.f - zp) * hp + zp * v_reset; }
                        acc_k += (double)g_h * (double)((he - v_prev) / k);
                    }
                }
                const unsigned short ob = narrow<DT>(g_x * scale);
                if constexpr (VEC == 1) xv = ob; else xv[e] = ob;
            }
            store_out(reinterpret_cast<uvec_t*>(a.g_x_seq + ((long long)t * NV + i) * VEC), xv);
            if constexpr (G2) { if (g_sum_seq) *reinterpret_cast<uvec_t*>(g_sum_seq + ((long long)t * NV + i) * VEC) = sumv; }
        }
        if (a.g_v_init) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) a.g_v_init[i * VEC + e] = gv[e];
        }
    }
    if (want_gk) gk_epilogue(acc_k, a.g_k_partials);   // wave-uniform
}

template <int KIND, int SG, int DT>
int dispatch_bwd16(const Bwd16Args& a, hipStream_t s, int* grid_out)
{
    const bool vec = (a.N % 8 == 0) && aligned16(a.g_out_seq) && aligned16(a.h_seq) && aligned16(a.g_x_seq) &&
                     (!a.g_v_last || aligned16(a.g_v_last)) && (!a.g_v_init || aligned16(a.g_v_init)) &&
                     (!a.v_init || aligned16(a.v_init));
    int grid = vec ? grid_for(a.N / 8, kMaxGridBwd) : grid_for(a.N, kMaxGridBwd);
    if (a.g_k_partials && grid > kMaxGridGk) grid = kMaxGridGk;
    *grid_out = grid;
    if (vec) hipLaunchKernelGGL((neuron_bwd16_kernel<KIND, SG, DT, 8>), dim3(grid), dim3(kBlock), 0, s, a);
    else     hipLaunchKernelGGL((neuron_bwd16_kernel<KIND, SG, DT, 1>), dim3(grid), dim3(kBlock), 0, s, a);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

template <int KIND, int SG, int DT, int TS>
int launch_bwd16_rc(const Bwd16Args& a, const unsigned short* x_seq, const unsigned short* g2, unsigned short* g_sum, hipStream_t s, int* grid_out,
                    const float* lr_p = nullptr, const float* lr_w = nullptr, int lr_C = 0)
{
    constexpr int V = (TS <= 5) ? SS_RC16_V5 : SS_RC16_V10;    // measured on the MI355X: tools/bench_rc16.py
    if (lr_p) {                                               // low-rank second gradient: vector lanes only
        const bool ok = (a.N % V == 0) && aligned16(a.g_out_seq) && aligned16(x_seq) && aligned16(a.g_x_seq) && aligned16(g_sum) && aligned16(lr_w) &&
                        (!a.g_v_last || aligned16(a.g_v_last)) && (!a.g_v_init || aligned16(a.g_v_init)) && (!a.v_init || aligned16(a.v_init));
        if (!ok || (kBlock * V) % lr_C != 0 || lr_C % V != 0) return SS_EINVAL;
        int grid = grid_for(a.N / V, kMaxGridBwd);
        if (a.g_k_partials && grid > kMaxGridGk) grid = kMaxGridGk;
        *grid_out = grid;
        hipLaunchKernelGGL((neuron_bwd16_rc_kernel<KIND, SG, DT, TS, V, true, true>), dim3(grid), dim3(kBlock), 0, s, a, x_seq, g2, g_sum, lr_p, lr_w, lr_C);
        return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
    }
    const bool vec = (a.N % V == 0) && aligned16(a.g_out_seq) && aligned16(x_seq) && aligned16(a.g_x_seq) && aligned16(g2) && aligned16(g_sum) &&
                     (!a.g_v_last || aligned16(a.g_v_last)) && (!a.g_v_init || aligned16(a.g_v_init)) &&
                     (!a.v_init || aligned16(a.v_init));
    int grid = vec ? grid_for(a.N / V, kMaxGridBwd) : grid_for(a.N, kMaxGridBwd);
    if (a.g_k_partials && grid > kMaxGridGk) grid = kMaxGridGk;
    *grid_out = grid;
    if (g2) {
        if (vec) hipLaunchKernelGGL((neuron_bwd16_rc_kernel<KIND, SG, DT, TS, V, true>), dim3(grid), dim3(kBlock), 0, s, a, x_seq, g2, g_sum);
        else     hipLaunchKernelGGL((neuron_bwd16_rc_kernel<KIND, SG, DT, TS, 1, true>), dim3(grid), dim3(kBlock), 0, s, a, x_seq, g2, g_sum);
    } else {
        if (vec) hipLaunchKernelGGL((neuron_bwd16_rc_kernel<KIND, SG, DT, TS, V>), dim3(grid), dim3(kBlock), 0, s, a, x_seq, g2, g_sum);
        else     hipLaunchKernelGGL((neuron_bwd16_rc_kernel<KIND, SG, DT, TS, 1>), dim3(grid), dim3(kBlock), 0, s, a, x_seq, g2, g_sum);
    }
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

template <int KIND, int SG, int DT>
int dispatch_bwd16_rc(const Bwd16Args& a, const unsigned short* x_seq, const unsigned short* g2, unsigned short* g_sum, hipStream_t s, int* grid_out,
                      const float* lr_p = nullptr, const float* lr_w = nullptr, int lr_C = 0)
{
    switch (a.T) {
        case 1: return launch_bwd16_rc<KIND, SG, DT, 1>(a, x_seq, g2, g_sum, s, grid_out, lr_p, lr_w, lr_C);
        case 2: return launch_bwd16_rc<KIND, SG, DT, 2>(a, x_seq, g2, g_sum, s, grid_out, lr_p, lr_w, lr_C);
        case 4: return launch_bwd16_rc<KIND, SG, DT, 4>(a, x_seq, g2, g_sum, s, grid_out, lr_p, lr_w, lr_C);
        case 5: return launch_bwd16_rc<KIND, SG, DT, 5>(a, x_seq, g2, g_sum, s, grid_out, lr_p, lr_w, lr_C);
        case 8: return launch_bwd16_rc<KIND, SG, DT, 8>(a, x_seq, g2, g_sum, s, grid_out, lr_p, lr_w, lr_C);
        case 10: return launch_bwd16_rc<KIND, SG, DT, 10>(a, x_seq, g2, g_sum, s, grid_out, lr_p, lr_w, lr_C);
        default: return SS_EINVAL;
    }
}

}  // namespace

extern "C" {

// the segmented low-rank form (ss_neuron_bwd16_lr.hip)
int ss_i_bwd16_lr_seg_supported(int T, long long N, int C);
int ss_i_bwd16_lr_seg(const void* g_out_seq, const float* g_v_last, const float* v_init, void* g_x_seq, float* g_v_init, double* g_k_partials,
                      const void* x_seq, void* g_sum_seq, const float* lr_p, const float* lr_w, int C,
                      int T, long long N, float scale, int kind, float tau, const float* k, float v_th, float v_reset,
                      int surrogate, float alpha, int detach_reset, int dtype, void* stream, int* grid_out);

static int neuron_bwd_x16_impl(const void* g_out_seq, const void* g_out2_seq, void* g_sum_seq, const float* g_v_last, const float* h_seq, const void* x_seq, const float* v_init,
                               void* g_x_seq, float* g_v_init, float* g_k, float* g_k_ws,
                               int T, long long N, float scale, int kind, float tau, const float* k,
                               float v_th, float v_reset, int surrogate, float alpha, int detach_reset, int dtype, void* stream,
                               const float* lr_p = nullptr, const float* lr_w = nullptr, int lr_C = 0)
{
    if ((!g_out_seq && !lr_p) || (!h_seq && !x_seq) || !g_x_seq || T <= 0 || N < 0) return SS_EINVAL;
    if (kind < SS_KIND_IF || kind > SS_KIND_PLIF || (dtype != SS_DT_F16 && dtype != SS_DT_BF16)) return SS_EINVAL;
    if (surrogate != SS_SG_ATAN && surrogate != SS_SG_SIGMOID) return SS_EINVAL;
    if (kind == SS_KIND_PLIF && !k) return SS_EINVAL;
    if (x_seq && g_x_seq == x_seq) return SS_EINVAL;
    const bool want_gk = (kind == SS_KIND_PLIF) && g_k != nullptr;
    if (want_gk && (!g_k_ws || (reinterpret_cast<uintptr_t>(g_k_ws) & 7u))) return SS_EINVAL;   // fp64 partials
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (N == 0) {
        if (want_gk && hipMemsetAsync(g_k, 0, sizeof(float), s) != hipSuccess) return SS_ELAUNCH;
        return SS_OK;
    }
    Bwd16Args a{static_cast<const unsigned short*>(g_out_seq), g_v_last, h_seq, v_init, static_cast<unsigned short*>(g_x_seq),
                g_v_init, want_gk ? reinterpret_cast<double*>(g_k_ws) : nullptr, T, N, scale, tau, v_th, v_reset, alpha, k, detach_reset};
    const unsigned short* xq = static_cast<const unsigned short*>(x_seq);
    int grid = 0, rc;
    const unsigned short* g2q = static_cast<const unsigned short*>(g_out2_seq);
    unsigned short* gsq = (g2q || lr_p) ? static_cast<unsigned short*>(g_sum_seq) : nullptr;
    if (lr_p && ss_i_bwd16_lr_seg_supported(T, N, lr_C)) {       // round 6: the segmented form (LDS-staged pair, Newton reciprocal) where it applies
        rc = ss_i_bwd16_lr_seg(g_out_seq, g_v_last, v_init, g_x_seq, g_v_init, a.g_k_partials, x_seq, gsq, lr_p, lr_w, lr_C, T, N, scale, kind, tau, k, v_th, v_reset,
                               surrogate, alpha, detach_reset, dtype, stream, &grid);
        if (rc != SS_OK) return rc;
        if (want_gk) {
            hipLaunchKernelGGL(gk_finish_kernel, dim3(1), dim3(kBlock), 0, s, reinterpret_cast<const double*>(g_k_ws), grid, g_k);
            if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
        }
        return SS_OK;
    }
#define SS_B16D(KK, SGG, DTT) (xq ? dispatch_bwd16_rc<KK, SGG, DTT>(a, xq, g2q, gsq, s, &grid, lr_p, lr_w, lr_C) : dispatch_bwd16<KK, SGG, DTT>(a, s, &grid))
#define SS_B16(KK, SGG) (dtype == SS_DT_F16 ? SS_B16D(KK, SGG, SS_DT_F16) : SS_B16D(KK, SGG, SS_DT_BF16))
#define SS_B16S(KK) (surrogate == SS_SG_ATAN ? SS_B16(KK, SS_SG_ATAN) : SS_B16(KK, SS_SG_SIGMOID))
    switch (kind) {
        case SS_KIND_IF: rc = SS_B16S(SS_KIND_IF); break;
        case SS_KIND_LIF: rc = SS_B16S(SS_KIND_LIF); break;
        default: rc = SS_B16S(SS_KIND_PLIF); break;
    }
#undef SS_B16S
#undef SS_B16
#undef SS_B16D
    if (rc != SS_OK) return rc;
    if (want_gk) {
        hipLaunchKernelGGL(gk_finish_kernel, dim3(1), dim3(kBlock), 0, s, reinterpret_cast<const double*>(g_k_ws), grid, g_k);
        if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    }
    return SS_OK;
}

int ss_neuron_bwd_x16(const void* g_out_seq, const float* g_v_last, const float* h_seq, const float* v_init,
                      void* g_x_seq, float* g_v_init, float* g_k, float* g_k_ws,
                      int T, long long N, float scale, int kind, float tau, const float* k,
                      float v_th, float v_reset, int surrogate, float alpha, int detach_reset, int dtype, void* stream)
{
    if (!h_seq) return SS_EINVAL;
    return neuron_bwd_x16_impl(g_out_seq, nullptr, nullptr, g_v_last, h_seq, nullptr, v_init, g_x_seq, g_v_init, g_k, g_k_ws, T, N, scale, kind, tau, k,
                               v_th, v_reset, surrogate, alpha, detach_reset, dtype, stream);
}

int ss_neuron_bwd_rc_x16(const void* g_out_seq, const float* g_v_last, const void* x_seq, const float* v_init,
                         void* g_x_seq, float* g_v_init, float* g_k, float* g_k_ws,
                         int T, long long N, float scale, int kind, float tau, const float* k,
                         float v_th, float v_reset, int surrogate, float alpha, int detach_reset, int dtype, void* stream)
{
    if (!x_seq || !ss_neuron_bwd_rc_supported(T)) return SS_EINVAL;
    return neuron_bwd_x16_impl(g_out_seq, nullptr, nullptr, g_v_last, nullptr, x_seq, v_init, g_x_seq, g_v_init, g_k, g_k_ws, T, N, scale, kind, tau, k,
                               v_th, v_reset, surrogate, alpha, detach_reset, dtype, stream);
}

int ss_neuron_bwd_fork_x16(const void* g_out_seq, const void* g_out2_seq, void* g_sum_seq, const float* g_v_last, const void* x_seq,
                           const float* v_init, void* g_x_seq, float* g_v_init, float* g_k, float* g_k_ws,
                           int T, long long N, float scale, int kind, float tau, const float* k,
                           float v_th, float v_reset, int surrogate, float alpha, int detach_reset, int dtype, void* stream)
{
    if (!x_seq || !g_out2_seq || !ss_neuron_bwd_rc_supported(T)) return SS_EINVAL;
    if (g_out2_seq == g_x_seq || (g_sum_seq && g_sum_seq == g_x_seq)) return SS_EINVAL;
    return neuron_bwd_x16_impl(g_out_seq, g_out2_seq, g_sum_seq, g_v_last, nullptr, x_seq, v_init, g_x_seq, g_v_init, g_k, g_k_ws, T, N, scale, kind,
                               tau, k, v_th, v_reset, surrogate, alpha, detach_reset, dtype, stream);
}

/* ss_neuron_bwd_fork_lr_f32 on 16-bit activations (ABI 9): g_out_seq (nullable) / g_sum_seq (nullable) / x_seq / g_x_seq in `dtype`, the rank-9 pair fp32 */
int ss_neuron_bwd_fork_lr_x16_supported(int T, long long N, int C, int lr_rank)
{
    const int V = T <= 5 ? SS_RC16_V5 : SS_RC16_V10;
    if (lr_rank == kLr16Rank && ss_i_bwd16_lr_seg_supported(T, N, C)) return 1;
    return ss_neuron_bwd_rc_supported(T) && lr_rank == kLr16Rank && C >= 4 && C <= kLr16MaxC && C % V == 0 && C % 4 == 0 && (kBlock * V) % C == 0 && N > 0 && N % C == 0;
}

int ss_neuron_bwd_fork_lr_x16(const void* g_out_seq, const float* lr_p, const float* lr_w, int lr_rank, int C, void* g_sum_seq,
                              const float* g_v_last, const void* x_seq, const float* v_init, void* g_x_seq, float* g_v_init,
                              float* g_k, float* g_k_ws, int T, long long N, float scale, int kind, float tau, const float* k,
                              float v_th, float v_reset, int surrogate, float alpha, int detach_reset, int dtype, void* stream)
{
    if (!lr_p || !lr_w || !x_seq) return SS_EINVAL;
    if (!ss_neuron_bwd_fork_lr_x16_supported(T, N, C, lr_rank)) return SS_EINVAL;
    if (g_sum_seq && (!g_out_seq || g_sum_seq == g_x_seq)) return SS_EINVAL;   // without a dense first gradient the "sum" IS the low-rank pair
    return neuron_bwd_x16_impl(g_out_seq, nullptr, g_sum_seq, g_v_last, nullptr, x_seq, v_init, g_x_seq, g_v_init, g_k, g_k_ws, T, N, scale, kind,
                               tau, k, v_th, v_reset, surrogate, alpha, detach_reset, dtype, stream, lr_p, lr_w, C);
}

}  // extern "C"

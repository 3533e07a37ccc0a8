// ss_neuron_bwd.hip — surrogate-gradient backward of the fused neuron layers, fp32 activations (saved-h, recompute, forked and low-rank forms) + C-ABI.
// Design notes: see ss_neuron.hip.
#include "ss_common.hpp"

namespace {

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
struct BwdArgs {
    const float* g_out_seq; const float* g_v_last; const float* h_seq; const float* v_init;
    float* g_x_seq; float* g_v_init; double* g_k_partials;
    int T; long long N;
    float scale, tau, v_th, v_reset, alpha; const float* k; int detach_reset;
    const float* x_seq;   // non-null (templated T only): h_seq is not read, h is recomputed from the layer input (ss_neuron_bwd_rc_f32)
    const float* g_out2_seq;  // nullable: gradient from a second consumer of out_seq, added on load (ss_neuron_bwd_fork_f32)
    float* g_sum_seq;         // nullable (with g_out2_seq): g_out + g_out2 written out = dL/dskip_seq of a stage that has both
    // low-rank second gradient (ss_neuron_bwd_fork_lr_f32): g2[t][n] = sum_j lr_p[(t * N / lr_C + n / lr_C) * kLrRank + j] * lr_w[j * lr_C + n % lr_C]
    const float* lr_p; const float* lr_w; int lr_C;
};
constexpr int kLrMaxC = 512;  // widest layer that feeds a head (deconv4's output has 256 channels)
constexpr int kLrRank = 9;    // the prediction heads: 3 x 3 taps, one output channel (SNN_models.py:150-163 of the reference)

#ifndef SS_LR_WAVES
#define SS_LR_WAVES 4    // LR form, T = 5: 132 registers as compiled freely — 4 over the 4-waves-per-SIMD budget; asking for it spills 4 and
#endif                   // measures 486 vs 494 - 498 us on the 32 x 260 x 346 layer for IF / LIF; PLIF (division-heavy dL/dk term) is faster
                         // left alone: 544 vs 564 us (profiles/r02/bench_lr_variants.log)
template <int KIND, int SG, int TS, int VEC, bool RC = false, bool G2 = false, bool LR = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu((LR && TS == 5 && KIND != SS_KIND_PLIF) ? SS_LR_WAVES : 1)))
void neuron_bwd_kernel(BwdArgs a)
{
    static_assert(!LR || (RC && G2 && VEC == 4 && TS > 0), "low-rank second gradient: recompute form, float4 lanes");
    typedef typename std::conditional<VEC == 4, f4, float>::type vec_t;
    const int T = (TS > 0) ? TS : a.T;
    const long long NV = a.N / VEC;
    const float k = (KIND == SS_KIND_PLIF) ? *a.k : 0.f;
    const float scale = a.scale, tau = a.tau, v_th = a.v_th, v_reset = a.v_reset, alpha = a.alpha;
    const float c_atan = (float)(M_PI / 2.0 * (double)alpha);
    const float half_alpha = (float)((double)alpha / 2.0);
    const bool detach = a.detach_reset != 0;
    const bool want_gk = (KIND == SS_KIND_PLIF) && a.g_k_partials != nullptr;
    double acc_k = 0.0;
    // LR: the head's 9 x C weight matrix sits in LDS (<= 18 KB); the lane's 4 channels are the same in every trip of the grid-stride loop
    // (kBlock * 4 is a multiple of C: checked by the host), so a lane always reads the same nine 16-B slices
    __shared__ __attribute__((aligned(16))) float lr_ws[LR ? kLrRank * kLrMaxC : 4];
    int lr_c0 = 0;
    if constexpr (LR) {
        for (int q = threadIdx.x; q < kLrRank * a.lr_C; q += kBlock) lr_ws[q] = a.lr_w[q];
        lr_c0 = (int)((threadIdx.x * 4u) % (unsigned)a.lr_C);
        __syncthreads();
    }

    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < NV; i += (long long)gridDim.x * kBlock) {
        const vec_t* gp = reinterpret_cast<const vec_t*>(a.g_out_seq) + i;
        const vec_t* hp = reinterpret_cast<const vec_t*>(a.h_seq) + i;
        vec_t* xp = reinterpret_cast<vec_t*>(a.g_x_seq) + i;

        vec_t gv;
        if (a.g_v_last) gv = reinterpret_cast<const vec_t*>(a.g_v_last)[i];
        else { if constexpr (VEC == 4) gv = (f4){0.f, 0.f, 0.f, 0.f}; else gv = 0.f; }
        vec_t v0;   // membrane before step 0 (PLIF dL/dk only)
        if constexpr (VEC == 4) v0 = (f4){0.f, 0.f, 0.f, 0.f}; else v0 = 0.f;
        if (want_gk || RC) {
            if (a.v_init) v0 = reinterpret_cast<const vec_t*>(a.v_init)[i];
            else { if constexpr (VEC == 4) v0 = (f4){v_reset, v_reset, v_reset, v_reset}; else v0 = v_reset; }
        }

        auto step = [&](vec_t g, vec_t h, vec_t hprev, bool first) -> vec_t {
            vec_t gx;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float ge, he, hpe, gve, v0e;
                if constexpr (VEC == 4) { ge = g[e]; he = h[e]; hpe = hprev[e]; gve = gv[e]; v0e = v0[e]; }
                else { ge = g; he = h; hpe = hprev; gve = gv; v0e = v0; }
                float xh = he - v_th;
                float z = heaviside(xh);
                float g_s = ge;
                if (!detach) g_s = g_s + (gve * v_reset - gve * he);
                float g_h = surrogate_grad<SG>(xh, alpha, c_atan, half_alpha, g_s) + gve * (1.f - z);
                float g_x;
                if (KIND == SS_KIND_IF) { g_x = g_h; gve = g_h; }
                else if (KIND == SS_KIND_LIF) { g_x = g_h / tau; gve = g_h - g_x; }
                else {
                    g_x = g_h * k; gve = g_h - g_x;
                    if (want_gk) {
                        float v_prev;
                        if (first) v_prev = v0e;
                        else { float zp = heaviside(hpe - v_th); v_prev = (1.f - zp) * hpe + zp * v_reset; }
                        acc_k += (double)g_h * (double)((he - v_prev) / k);
                    }
                }
                if constexpr (VEC == 4) { gx[e] = g_x * scale; gv[e] = gve; } else { gx = g_x * scale; gv = gve; }
            }
            return gx;
        };

        if constexpr (TS > 0) {
            vec_t gs[TS], hs[TS];
            if constexpr (RC) {   // recompute h_t from the layer input with the forward kernel's exact arithmetic
                const vec_t* xq = reinterpret_cast<const vec_t*>(a.x_seq) + i;
                // issue order = consumption order: x ascending (forward recurrence), then g descending (reverse loop)
#pragma unroll
                for (int t = 0; t < TS; ++t) hs[t] = load_stream(xq + (long long)t * NV);
                if constexpr (LR) {   // second consumer's gradient in low-rank form: kLrRank floats per pixel and step instead of C
                    const long long rows = a.N / a.lr_C;
                    const float* pp = a.lr_p + ((i * 4) / a.lr_C) * kLrRank;
                    float pj[TS][kLrRank];
#pragma unroll
                    for (int t = TS - 1; t >= 0; --t)
#pragma unroll
                        for (int j = 0; j < kLrRank; ++j) pj[t][j] = pp[(long long)t * rows * kLrRank + j];
                    const bool has_g1 = a.g_out_seq != nullptr;   // wave-uniform
                    if (has_g1) {
#pragma unroll
                        for (int t = TS - 1; t >= 0; --t) gs[t] = load_stream(gp + (long long)t * NV);
                    }
                    f4 acc[TS];                                    // per step: taps in ascending order, multiply and add rounded separately
                    int c0v = lr_c0;
                    asm volatile("" : "+v"(c0v));                  // keep the nine LDS reads inside the loop (hoisted they would pin 36 registers)
#pragma unroll
                    for (int j = 0; j < kLrRank; ++j) {
                        const f4 wj = *reinterpret_cast<const f4*>(&lr_ws[j * a.lr_C + c0v]);
#pragma unroll
                        for (int t = TS - 1; t >= 0; --t) acc[t] = (j == 0) ? pj[t][0] * wj : acc[t] + pj[t][j] * wj;
                    }
#pragma unroll
                    for (int t = TS - 1; t >= 0; --t) gs[t] = has_g1 ? gs[t] + acc[t] : acc[t];
                    if (a.g_sum_seq) {   // wave-uniform
                        vec_t* sp = reinterpret_cast<vec_t*>(a.g_sum_seq) + i;
#pragma unroll
                        for (int t = TS - 1; t >= 0; --t) sp[(long long)t * NV] = gs[t];
                    }
                } else {
#pragma unroll
                for (int t = TS - 1; t >= 0; --t) gs[t] = load_stream(gp + (long long)t * NV);
                }
                if constexpr (G2 && !LR) {   // second consumer's gradient, added on load
                    const vec_t* gp2 = reinterpret_cast<const vec_t*>(a.g_out2_seq) + i;
                    vec_t g2[TS];
#pragma unroll
                    for (int t = TS - 1; t >= 0; --t) g2[t] = load_stream(gp2 + (long long)t * NV);
#pragma unroll
                    for (int t = TS - 1; t >= 0; --t) gs[t] += g2[t];
                    if (a.g_sum_seq) {   // wave-uniform
                        vec_t* sp = reinterpret_cast<vec_t*>(a.g_sum_seq) + i;
#pragma unroll
                        for (int t = TS - 1; t >= 0; --t) sp[(long long)t * NV] = gs[t];
                    }
                }
                vec_t vv = v0;
#pragma unroll
                for (int t = 0; t < TS; ++t) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        float x, v;
                        if constexpr (VEC == 4) { x = hs[t][e]; v = vv[e]; } else { x = hs[t]; v = vv; }
                        float h = charge<KIND>(v, x * scale, tau, k, v_reset);
                        float z = heaviside(h - v_th);
                        v = (1.f - z) * h + z * v_reset;
                        if constexpr (VEC == 4) { hs[t][e] = h; vv[e] = v; } else { hs[t] = h; vv = v; }
                    }
                }
            } else {
#pragma unroll
                for (int t = TS - 1; t >= 0; --t) { gs[t] = load_stream(gp + (long long)t * NV); hs[t] = load_stream(hp + (long long)t * NV); }
            }
#pragma unroll
            for (int t = TS - 1; t >= 0; --t)
                store_out(xp + (long long)t * NV, step(gs[t], hs[t], hs[t > 0 ? t - 1 : 0], t == 0));
        } else {
            vec_t gn = gp[(long long)(T - 1) * NV], hn = hp[(long long)(T - 1) * NV];
            for (int t = T - 1; t >= 0; --t) {
                vec_t gc = gn, hc = hn;
                if (t > 0) { gn = gp[(long long)(t - 1) * NV]; hn = hp[(long long)(t - 1) * NV]; }
                xp[(long long)t * NV] = step(gc, hc, hn, t == 0);
            }
        }
        if (a.g_v_init) reinterpret_cast<vec_t*>(a.g_v_init)[i] = gv;
    }

    if (want_gk) gk_epilogue(acc_k, a.g_k_partials);   // wave-uniform
}

// second pass of the dL/dk reduction: fixed order -> bit-reproducible.  tail = scalar-tail kernel's partial.
template <int KIND, int SG, int TS>
int launch_bwd(const BwdArgs& a, hipStream_t s, int* grid_out)
{
    if (a.x_seq) {                                        // recompute needs h_0..h_{T-1} in registers: templated T only
        if constexpr (TS == 0) return SS_EINVAL;
        else {
            const bool vec = (a.N % 4 == 0) && aligned16(a.g_out_seq) && aligned16(a.g_out2_seq) && aligned16(a.g_sum_seq) && aligned16(a.x_seq) && aligned16(a.g_x_seq) &&
                             (!a.g_v_last || aligned16(a.g_v_last)) && (!a.g_v_init || aligned16(a.g_v_init)) &&
                             (!a.v_init || aligned16(a.v_init));
            int grid = vec ? grid_for(a.N / 4, kMaxGridBwd) : grid_for(a.N, kMaxGridBwd);
            if (a.g_k_partials && grid > kMaxGridGk) grid = kMaxGridGk;
            *grid_out = grid;
            if (a.lr_p) {
                if (!vec || !aligned16(a.lr_w)) return SS_EINVAL;
                hipLaunchKernelGGL((neuron_bwd_kernel<KIND, SG, TS, 4, true, true, true>), dim3(grid), dim3(kBlock), 0, s, a);
            } else if (a.g_out2_seq) {
                if (vec) hipLaunchKernelGGL((neuron_bwd_kernel<KIND, SG, TS, 4, true, true>), dim3(grid), dim3(kBlock), 0, s, a);
                else     hipLaunchKernelGGL((neuron_bwd_kernel<KIND, SG, TS, 1, true, true>), dim3(grid), dim3(kBlock), 0, s, a);
            } else {
                if (vec) hipLaunchKernelGGL((neuron_bwd_kernel<KIND, SG, TS, 4, true>), dim3(grid), dim3(kBlock), 0, s, a);
                else     hipLaunchKernelGGL((neuron_bwd_kernel<KIND, SG, TS, 1, true>), dim3(grid), dim3(kBlock), 0, s, a);
            }
            return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
        }
    }
    const bool vec = (a.N % 4 == 0) && aligned16(a.g_out_seq) && aligned16(a.g_out2_seq) && aligned16(a.h_seq) && aligned16(a.x_seq) && aligned16(a.g_x_seq) &&
                     (!a.g_v_last || aligned16(a.g_v_last)) && (!a.g_v_init || aligned16(a.g_v_init)) &&
                     (!a.v_init || aligned16(a.v_init));
    int grid = vec ? grid_for(a.N / 4, kMaxGridBwd) : grid_for(a.N, kMaxGridBwd);
    if (a.g_k_partials && grid > kMaxGridGk) grid = kMaxGridGk;
    *grid_out = grid;
    if (vec) hipLaunchKernelGGL((neuron_bwd_kernel<KIND, SG, TS, 4>), dim3(grid), dim3(kBlock), 0, s, a);
    else     hipLaunchKernelGGL((neuron_bwd_kernel<KIND, SG, TS, 1>), dim3(grid), dim3(kBlock), 0, s, a);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

template <int KIND, int SG>
int dispatch_bwd_T(const BwdArgs& a, hipStream_t s, int* grid_out)
{
    switch (a.T) {
        case 1: return launch_bwd<KIND, SG, 1>(a, s, grid_out);
        case 2: return launch_bwd<KIND, SG, 2>(a, s, grid_out);
        case 4: return launch_bwd<KIND, SG, 4>(a, s, grid_out);
        case 5: return launch_bwd<KIND, SG, 5>(a, s, grid_out);
        case 8: return launch_bwd<KIND, SG, 8>(a, s, grid_out);
        case 10: return launch_bwd<KIND, SG, 10>(a, s, grid_out);
        default: return launch_bwd<KIND, SG, 0>(a, s, grid_out);
    }
}

template <int KIND>
int dispatch_bwd_sg(const BwdArgs& a, int surrogate, hipStream_t s, int* grid_out)
{
    return surrogate == SS_SG_ATAN ? dispatch_bwd_T<KIND, SS_SG_ATAN>(a, s, grid_out)
                                   : dispatch_bwd_T<KIND, SS_SG_SIGMOID>(a, s, grid_out);
}

}  // namespace

extern "C" {

long long ss_neuron_gk_ws_floats(void) { return kGkWsFloats; }

static int neuron_bwd_f32_impl(const float* g_out_seq, const float* g_out2_seq, float* g_sum_seq, const float* g_v_last, const float* h_seq, const float* x_seq,
                               const float* v_init, float* g_x_seq, float* g_v_init, float* g_k, float* g_k_ws,
                               int T, long long N, float scale, int kind, float tau, const float* k,
                               float v_th, float v_reset, int surrogate, float alpha, int detach_reset, void* stream,
                               const float* lr_p = nullptr, const float* lr_w = nullptr, int lr_C = 0)
{
    if ((!g_out_seq && !lr_p) || (!h_seq && !x_seq) || !g_x_seq || T <= 0 || N < 0) return SS_EINVAL;
    if (kind < SS_KIND_IF || kind > SS_KIND_PLIF) return SS_EINVAL;
    if (surrogate != SS_SG_ATAN && surrogate != SS_SG_SIGMOID) return SS_EINVAL;
    if (kind == SS_KIND_PLIF && !k) return SS_EINVAL;
    if (x_seq && g_x_seq == x_seq) return SS_EINVAL;      // a lane reads all of x before writing g_x, but keep the input intact
    const bool want_gk = (kind == SS_KIND_PLIF) && g_k != nullptr;
    if (want_gk && (!g_k_ws || (reinterpret_cast<uintptr_t>(g_k_ws) & 7u))) return SS_EINVAL;   // fp64 partials
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (N == 0) {
        if (want_gk && hipMemsetAsync(g_k, 0, sizeof(float), s) != hipSuccess) return SS_ELAUNCH;
        return SS_OK;
    }
    BwdArgs a{g_out_seq, g_v_last, h_seq, v_init, g_x_seq, g_v_init, want_gk ? reinterpret_cast<double*>(g_k_ws) : nullptr,
              T, N, scale, tau, v_th, v_reset, alpha, k, detach_reset, x_seq, g_out2_seq, (g_out2_seq || lr_p) ? g_sum_seq : nullptr,
              lr_p, lr_w, lr_C};
    int grid = 0, rc;
    switch (kind) {
        case SS_KIND_IF: rc = dispatch_bwd_sg<SS_KIND_IF>(a, surrogate, s, &grid); break;
        case SS_KIND_LIF: rc = dispatch_bwd_sg<SS_KIND_LIF>(a, surrogate, s, &grid); break;
        default: rc = dispatch_bwd_sg<SS_KIND_PLIF>(a, surrogate, s, &grid); break;
    }
    if (rc != SS_OK) return rc;
    if (want_gk) {
        hipLaunchKernelGGL(gk_finish_kernel, dim3(1), dim3(kBlock), 0, s, reinterpret_cast<const double*>(g_k_ws), grid, g_k);
        if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    }
    return SS_OK;
}

int ss_neuron_bwd_f32(const float* g_out_seq, const float* g_v_last, const float* h_seq, const float* v_init,
                      float* g_x_seq, float* g_v_init, float* g_k, float* g_k_ws,
                      int T, long long N, float scale, int kind, float tau, const float* k,
                      float v_th, float v_reset, int surrogate, float alpha, int detach_reset, void* stream)
{
    if (!h_seq) return SS_EINVAL;
    return neuron_bwd_f32_impl(g_out_seq, nullptr, nullptr, g_v_last, h_seq, nullptr, v_init, g_x_seq, g_v_init, g_k, g_k_ws, T, N, scale, kind, tau, k,
                               v_th, v_reset, surrogate, alpha, detach_reset, stream);
}

int ss_neuron_bwd_rc_supported(int T)
{
    return T == 1 || T == 2 || T == 4 || T == 5 || T == 8 || T == 10;
}

int ss_neuron_bwd_rc_f32(const float* g_out_seq, const float* g_v_last, const float* x_seq, const float* v_init,
                         float* g_x_seq, float* g_v_init, float* g_k, float* g_k_ws,
                         int T, long long N, float scale, int kind, float tau, const float* k,
                         float v_th, float v_reset, int surrogate, float alpha, int detach_reset, void* stream)
{
    if (!x_seq || !ss_neuron_bwd_rc_supported(T)) return SS_EINVAL;
    return neuron_bwd_f32_impl(g_out_seq, nullptr, nullptr, g_v_last, nullptr, x_seq, v_init, g_x_seq, g_v_init, g_k, g_k_ws, T, N, scale, kind, tau, k,
                               v_th, v_reset, surrogate, alpha, detach_reset, stream);
}

int ss_neuron_bwd_fork_f32(const float* g_out_seq, const float* g_out2_seq, float* g_sum_seq, const float* g_v_last, const float* h_seq, const float* x_seq,
                           const float* v_init, float* g_x_seq, float* g_v_init, float* g_k, float* g_k_ws,
                           int T, long long N, float scale, int kind, float tau, const float* k,
                           float v_th, float v_reset, int surrogate, float alpha, int detach_reset, void* stream)
{
    if ((h_seq != nullptr) == (x_seq != nullptr)) return SS_EINVAL;          // exactly one of saved h / layer input
    if (x_seq && !ss_neuron_bwd_rc_supported(T)) return SS_EINVAL;
    if (g_out2_seq && !x_seq) return SS_EINVAL;                              // the fused second gradient exists in the recompute form only
    if (g_out2_seq && (g_out2_seq == g_x_seq)) return SS_EINVAL;
    if (g_sum_seq && (!g_out2_seq || g_sum_seq == g_x_seq)) return SS_EINVAL;
    return neuron_bwd_f32_impl(g_out_seq, g_out2_seq, g_sum_seq, g_v_last, h_seq, x_seq, v_init, g_x_seq, g_v_init, g_k, g_k_ws, T, N, scale, kind,
                               tau, k, v_th, v_reset, surrogate, alpha, detach_reset, stream);
}

int ss_neuron_bwd_fork_lr_supported(int T, long long N, int C, int lr_rank)
{
    return ss_neuron_bwd_rc_supported(T) && lr_rank == kLrRank && C >= 4 && C <= kLrMaxC && C % 4 == 0 && (kBlock * 4) % C == 0 && N > 0 && N % C == 0;
}

int ss_neuron_bwd_fork_lr_f32(const float* g_out_seq, const float* lr_p, const float* lr_w, int lr_rank, int C, float* g_sum_seq,
                              const float* g_v_last, const float* x_seq, const float* v_init, float* g_x_seq, float* g_v_init,
                              float* g_k, float* g_k_ws, int T, long long N, float scale, int kind, float tau, const float* k,
                              float v_th, float v_reset, int surrogate, float alpha, int detach_reset, void* stream)
{
    if (!lr_p || !lr_w || !x_seq) return SS_EINVAL;
    if (!ss_neuron_bwd_fork_lr_supported(T, N, C, lr_rank)) return SS_EINVAL;
    if (g_sum_seq && (!g_out_seq || g_sum_seq == g_x_seq)) return SS_EINVAL;   // without a dense first gradient the "sum" IS the low-rank pair
    return neuron_bwd_f32_impl(g_out_seq, nullptr, g_sum_seq, g_v_last, nullptr, x_seq, v_init, g_x_seq, g_v_init, g_k, g_k_ws, T, N, scale, kind,
                               tau, k, v_th, v_reset, surrogate, alpha, detach_reset, stream, lr_p, lr_w, C);
}

}  // extern "C"

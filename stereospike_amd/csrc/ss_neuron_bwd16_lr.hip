// ss_neuron_bwd16_lr.hip — the low-rank-pair form of the 16-bit surrogate backward (ss_neuron_bwd_fork_lr_x16) on the segmented kernel of
// ss_neuron16_v2.hpp: its own translation unit (144 two-pass instantiations compile beside the other units).  Internal entry point, called from
// ss_neuron_bwd16.hip; shapes it does not take stay on neuron_bwd16_rc_kernel there.
#include "ss_common.hpp"
#include "ss_neuron16_v2.hpp"
#include <atomic>

namespace {

// The exact pass's request flags (ss_neuron16_v2.hpp, PASS): one word per launch in flight, taken round-robin.  The only state of the library that outlives a call
// — 4 KB of device memory: a launch clears its word, the fast pass may set it, the exact pass behind it on the same stream reads it.  1024 launches would have to be
// in flight at once for two of them to share a word, and then the worst case is an exact pass that was not needed.
constexpr int kLrGridCap = 8192;
constexpr int kRedoSlots = 1024;
__device__ unsigned ss_lr_redo_flags[kRedoSlots];

inline unsigned* redo_slot()
{
    static unsigned* base = [] { void* p = nullptr; return hipGetSymbolAddress(&p, HIP_SYMBOL(ss_lr_redo_flags)) == hipSuccess ? static_cast<unsigned*>(p) : nullptr; }();
    static std::atomic<unsigned> next{0};
    return base ? base + (next.fetch_add(1, std::memory_order_relaxed) % kRedoSlots) : nullptr;
}

// lane width / segments / wavefronts per SIMD the compiler is held to, measured on the MI355X at BASELINE config 5's and config 3's layer shapes
// (tools/ubench/neuron16_ab.hip, profiles/r06/neuron16_ab_*.log): T <= 5: 4 neurons per lane, one segment, 4 waves (128 registers) — 0.63 - 0.67 of 8 TB/s
// (round 5: 0.45 - 0.47); T > 5: 4 per lane, two segments, 3 waves (168 registers) — 0.51 - 0.55 (round 5: 0.39 - 0.41).  8 per lane (16-byte accesses) spills.
template <int KIND, int SG, int DT, int TS, bool HAS_G1, bool SUM>
int launch_lr_form(const Bwd16Args& a, const unsigned short* x_seq, unsigned short* g_sum, const float* lr_p, const float* lr_w, int C, hipStream_t s, int grid, int pair_x4)
{
    constexpr int V = 4, NSEG = TS > 5 ? 2 : 1, W = TS > 5 ? 3 : 4;
    const size_t lds = bwd16_seg_lds_bytes(TS, V, C);
    if constexpr (SG == SS_SG_ATAN) {
        unsigned* const flag = redo_slot();
        if (!flag || hipMemsetAsync(flag, 0, sizeof(unsigned), s) != hipSuccess) return SS_ELAUNCH;
        hipLaunchKernelGGL((neuron_bwd16_seg_kernel<KIND, SG, DT, TS, V, NSEG, true, true, W, HAS_G1, SUM, 0>), dim3(grid), dim3(kBlock), lds, s, a, x_seq, nullptr, g_sum, lr_p, lr_w, C, pair_x4, flag);
        if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
        // the exact pass: returns at once unless the fast pass asked for it.  Its grid is bounded (it loops) — except with dL/dk partials, which are per workgroup
        const int grid1 = a.g_k_partials ? grid : (grid < 2048 ? grid : 2048);
        hipLaunchKernelGGL((neuron_bwd16_seg_kernel<KIND, SG, DT, TS, V, NSEG, true, true, 2, HAS_G1, SUM, 1>), dim3(grid1), dim3(kBlock), lds, s, a, x_seq, nullptr, g_sum, lr_p, lr_w, C, pair_x4, flag);
    } else {
        hipLaunchKernelGGL((neuron_bwd16_seg_kernel<KIND, SG, DT, TS, V, NSEG, true, true, W, HAS_G1, SUM, 2>), dim3(grid), dim3(kBlock), lds, s, a, x_seq, nullptr, g_sum, lr_p, lr_w, C, pair_x4, nullptr);
    }
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

template <int KIND, int SG, int DT, int TS>
int launch_lr(const Bwd16Args& a, const unsigned short* x_seq, unsigned short* g_sum, const float* lr_p, const float* lr_w, int C, hipStream_t s, int* grid_out)
{
    constexpr int V = 4;
    int grid = grid_for(a.N / V, kMaxGridBwd);
    if (a.g_k_partials && grid > kMaxGridGk) grid = kMaxGridGk;
    *grid_out = grid;
    const int pair_x4 = ((a.N / C) % 4 == 0) && (((64 * V) / C) % 4 == 0) && aligned16(lr_p);
    if (g_sum && a.g_out_seq) return launch_lr_form<KIND, SG, DT, TS, true, true>(a, x_seq, g_sum, lr_p, lr_w, C, s, grid, pair_x4);
    // without the g_sum store: a bounded grid with the grid-stride loop (the wavefronts' start-up amortised over several vectors): 0.565 -> 0.593 of 8 TB/s at
    // T = 10, 0.59 -> 0.66 at T = 5; the form that also writes g_sum (and the forms without a pair) lose with it (profiles/r06/neuron16_bwd_grid_ab.log)
    if (grid > kLrGridCap) { grid = kLrGridCap; *grid_out = grid; }
    if (!a.g_out_seq) return launch_lr_form<KIND, SG, DT, TS, false, false>(a, x_seq, nullptr, lr_p, lr_w, C, s, grid, pair_x4);
    return launch_lr_form<KIND, SG, DT, TS, true, false>(a, x_seq, nullptr, lr_p, lr_w, C, s, grid, pair_x4);
}

template <int KIND, int SG, int DT>
int dispatch_lr_T(const Bwd16Args& a, const unsigned short* x_seq, unsigned short* g_sum, const float* lr_p, const float* lr_w, int C, hipStream_t s, int* grid_out)
{
    switch (a.T) {
        case 4: return launch_lr<KIND, SG, DT, 4>(a, x_seq, g_sum, lr_p, lr_w, C, s, grid_out);
        case 5: return launch_lr<KIND, SG, DT, 5>(a, x_seq, g_sum, lr_p, lr_w, C, s, grid_out);
        case 8: return launch_lr<KIND, SG, DT, 8>(a, x_seq, g_sum, lr_p, lr_w, C, s, grid_out);
        case 10: return launch_lr<KIND, SG, DT, 10>(a, x_seq, g_sum, lr_p, lr_w, C, s, grid_out);
        default: return SS_EINVAL;
    }
}

}  // namespace

extern "C" {

// shapes of the segmented form: compile-time T of the longer sequences, C a divisor of the 256 neurons of a wavefront (whole pixels per wavefront), 4 | C
__attribute__((visibility("hidden"))) int ss_i_bwd16_lr_seg_supported(int T, long long N, int C)
{
    return (T == 4 || T == 5 || T == 8 || T == 10) && C >= 4 && C <= 256 && 256 % C == 0 && N > 0 && N % C == 0 &&
           bwd16_seg_lds_bytes(T, 4, C) <= 65536;         // (narrow C: many pixels per wavefront; beyond the default LDS limit the round-5 kernel takes the shape)
}

__attribute__((visibility("hidden"))) int ss_i_bwd16_lr_seg(const void* g_out_seq, const float* g_v_last, const float* v_init, void* g_x_seq, float* g_v_init, double* g_k_partials,
                                                            const void* x_seq, void* g_sum_seq, const float* lr_p, const float* lr_w, int C,
                                                            int T, long long N, float scale, int kind, float tau, const float* k, float v_th, float v_reset,
                                                            int surrogate, float alpha, int detach_reset, int dtype, void* stream, int* grid_out)
{
    if (!ss_i_bwd16_lr_seg_supported(T, N, C)) return SS_EINVAL;
    const auto a8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; };
    if (!a8(g_out_seq) || !a8(x_seq) || !a8(g_x_seq) || !a8(g_sum_seq) || !a8(g_v_last) || !a8(v_init) || !a8(g_v_init) || !aligned16(lr_w)) return SS_EINVAL;
    Bwd16Args a{static_cast<const unsigned short*>(g_out_seq), g_v_last, nullptr, v_init, static_cast<unsigned short*>(g_x_seq), g_v_init, g_k_partials,
                T, N, scale, tau, v_th, v_reset, alpha, k, detach_reset};
    const unsigned short* xq = static_cast<const unsigned short*>(x_seq);
    unsigned short* gs = static_cast<unsigned short*>(g_sum_seq);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define SS_LRD(KK, SGG, DTT) dispatch_lr_T<KK, SGG, DTT>(a, xq, gs, lr_p, lr_w, C, s, grid_out)
#define SS_LR(KK, SGG) (dtype == SS_DT_F16 ? SS_LRD(KK, SGG, SS_DT_F16) : SS_LRD(KK, SGG, SS_DT_BF16))
#define SS_LRS(KK) (surrogate == SS_SG_ATAN ? SS_LR(KK, SS_SG_ATAN) : SS_LR(KK, SS_SG_SIGMOID))
    switch (kind) {
        case SS_KIND_IF: return SS_LRS(SS_KIND_IF);
        case SS_KIND_LIF: return SS_LRS(SS_KIND_LIF);
        default: return SS_LRS(SS_KIND_PLIF);
    }
#undef SS_LRS
#undef SS_LR
#undef SS_LRD
}

}  // extern "C"

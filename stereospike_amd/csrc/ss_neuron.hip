// ss_neuron.hip — hand-written CDNA4 (gfx950 / MI355X) kernels + the C-ABI of include/ss_neuron.h.
//
// The hot path of the reference: after every conv the
// reference runs ~10 eager point-wise kernels per time step (MultiplyBy, charge, sub, >=, cast, 1-z, mul,
// mul, add, skip add — /root/reference/network/blocks.py:106-107,161-171, SNN_models.py:152-192 plus the
// un-vendored spikingjelly single-step nodes) and autograd mirrors them in backward.  Here the whole chain,
// over all T steps, is ONE streaming kernel per layer in each direction.
//
// Design for MI355X (all HBM-bound, ~12 flops per 12 bytes — no MFMA, no GEMM reshaping):
//  * flat 1-D over N = B*C*H*W neurons; one lane owns 4 consecutive neurons (16-B dwordx4 accesses, a
//    wavefront covers 1 KiB contiguous per instruction => perfectly coalesced);
//  * the membrane v of a lane's 4 neurons lives in VGPRs across the t-loop: the state is private to the
//    lane, so staging it in LDS would only add a ds_write/ds_read round trip per step;
//    LDS is used where lanes actually exchange data — the workgroup stage of the firing-rate / dL/dk reductions;
//  * T is a template parameter for the common values so the T independent dwordx4 loads of a lane are all
//    issued before the first dependent VALU op (T x 16 B in flight per lane); a runtime-T fallback exists;
//  * grid: one f4 per lane for the backward, <= 32768 workgroups + a short grid-stride loop for the forward (A/B-measured,
//    see kMaxGrid below), 256-thread workgroups (4 waves);
//  * firing-rate counters: per-lane integer count -> wavefront butterfly (__shfl_xor over 64 lanes) ->
//    4 partials in LDS -> one 64-bit integer atomic per counter per workgroup (deterministic);
//  * dL/dk (PLIF): per-lane fp32 -> wavefront butterfly -> LDS -> one partial per workgroup in the caller's
//    workspace -> second single-workgroup pass in fixed order (deterministic, no float atomics).
//
// Rounding discipline: compiled with -ffp-contract=off; every fp32 op below is written in the order eager
// PyTorch evaluates the reference (x*scale first, true division by tau, (1-z)*h + z*v_reset literally).
#include "ss_common.hpp"
#include "ss_neuron16_v2.hpp"

namespace {

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
struct FwdArgs {
    const float* x_seq; const float* v_init; const float* skip_seq;
    float* out_seq; float* h_seq; float* v_last; unsigned long long* nnz;
    int T; long long N;
    float scale, tau, v_th, v_reset; const float* k;
    // ss_neuron_fwd_ex only (PK instantiations): 2-bit packed spike I/O, 16 neurons per 32-bit word, [T][N/16] words
    const unsigned* skip_packed;   // nullable: the skip operand read from a packed spike tensor instead of skip_seq
    unsigned* out_packed;          // nullable: out (z + skip, values 0..3) written packed; out_seq may then be NULL (4.25 B/update forward)
    unsigned* cnt_ws;              // nullable (with nnz): per-workgroup counter partials, summed by cnt_finish_kernel in a fixed order
};

// out values 0..3 of VEC consecutive neurons of one lane -> 2*VEC bits; 16 / VEC neighbouring lanes share one word
template <int VEC> __device__ __forceinline__ void store_packed(unsigned* words, long long i, unsigned bits)
{
    constexpr int LPW = 16 / VEC;                     // lanes per word: 4 (VEC = 4), 2 (VEC = 8)
    unsigned w = bits;
#pragma unroll
    for (int s = 1; s < LPW; ++s) w |= (unsigned)__shfl_down((int)bits, s, 64) << (2 * VEC * s);
    if ((threadIdx.x & (LPW - 1)) == 0) words[i / LPW] = w;
}
template <int VEC> __device__ __forceinline__ unsigned load_packed(const unsigned* words, long long i)
{
    constexpr int LPW = 16 / VEC;
    return (words[i / LPW] >> (2 * VEC * (int)(threadIdx.x & (LPW - 1)))) & ((1u << (2 * VEC)) - 1u);
}

// counter epilogue shared by the forward kernels: lane counts -> wavefront butterfly -> LDS -> per-workgroup partial (cnt_ws) or atomics
__device__ __forceinline__ void count_epilogue(unsigned c_spk, unsigned c_out, unsigned long long* nnz, unsigned* cnt_ws)
{
    __shared__ unsigned s_cnt[2][SS_BLOCK / 64];
    unsigned ws = wave_sum_u32(c_spk), wo = wave_sum_u32(c_out);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_cnt[0][wave] = ws; s_cnt[1][wave] = wo; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long ts = 0, to = 0;
#pragma unroll
        for (int w = 0; w < SS_BLOCK / 64; ++w) { ts += s_cnt[0][w]; to += s_cnt[1][w]; }
        if (cnt_ws) { cnt_ws[2 * blockIdx.x] = (unsigned)ts; cnt_ws[2 * blockIdx.x + 1] = (unsigned)to; }
        else { if (ts) atomicAdd(&nnz[0], ts); if (to) atomicAdd(&nnz[1], to); }
    }
}

// second pass of the counters: one workgroup sums the per-workgroup partials (integers: any order gives the same result) and adds
// them to nnz[0..1].  Replaces one same-address 64-bit atomic per workgroup (~12 ns each, serialised: 1 ms at 45 000 workgroups).
__global__ __launch_bounds__(SS_BLOCK) void cnt_finish_kernel(const unsigned* __restrict__ ws, int n, unsigned long long* nnz)
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

// VEC = 4: lane owns one f4 per time step; VEC = 1: scalar tail / unaligned fallback.
// PK: the ss_neuron_fwd_ex form with 2-bit packed spike output and / or packed skip input (VEC = 4, compile-time T only)
template <int KIND, int TS, bool SKIP, bool SAVE_H, int VEC, bool PK = false>
__global__ __launch_bounds__(kBlock) void neuron_fwd_kernel(FwdArgs a)
{
    typedef typename std::conditional<VEC == 4, f4, float>::type vec_t;
    const int T = (TS > 0) ? TS : a.T;
    const long long NV = a.N / VEC;                       // vectors per time step
    const float k = (KIND == SS_KIND_PLIF) ? *a.k : 0.f;
    const float scale = a.scale, tau = a.tau, v_th = a.v_th, v_reset = a.v_reset;
    unsigned c_spk = 0, c_out = 0;

    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < NV; i += (long long)gridDim.x * kBlock) {
        const vec_t* xp = reinterpret_cast<const vec_t*>(a.x_seq) + i;
        const vec_t* sp = SKIP ? reinterpret_cast<const vec_t*>(a.skip_seq) + i : nullptr;
        vec_t* op = reinterpret_cast<vec_t*>(a.out_seq) + i;
        vec_t* hp = SAVE_H ? reinterpret_cast<vec_t*>(a.h_seq) + i : nullptr;

        vec_t vv;
        if (a.v_init) vv = reinterpret_cast<const vec_t*>(a.v_init)[i];
        else { if constexpr (VEC == 4) vv = (f4){v_reset, v_reset, v_reset, v_reset}; else vv = v_reset; }

        if constexpr (PK) {
            static_assert(!PK || (TS > 0 && VEC == 4 && !SAVE_H), "packed I/O: compile-time T, 16-B lanes, no saved h");
            const long long NW = a.N / 16;                 // packed words per time step
            f4 xs[TS > 0 ? TS : 1];
            unsigned sb[TS > 0 ? TS : 1];
            const bool skip_pk = SKIP && a.skip_packed != nullptr;
#pragma unroll
            for (int t = 0; t < TS; ++t) xs[t] = load_stream(reinterpret_cast<const f4*>(xp) + (long long)t * NV);
            if (SKIP) {
                if (skip_pk) {
#pragma unroll
                    for (int t = 0; t < TS; ++t) sb[t] = load_packed<4>(a.skip_packed + (long long)t * NW, i);
                } else {
#pragma unroll
                    for (int t = 0; t < TS; ++t) {   // dense skip: small integers 0..3, exact in the 2-bit code
                        const f4 sv = reinterpret_cast<const f4*>(sp)[(long long)t * NV];
                        sb[t] = ((unsigned)sv[0] & 3u) | (((unsigned)sv[1] & 3u) << 2) | (((unsigned)sv[2] & 3u) << 4) | (((unsigned)sv[3] & 3u) << 6);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < TS; ++t) {
                f4 ov;
                unsigned bits = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float h = charge<KIND>(vv[e], xs[t][e] * scale, tau, k, v_reset);
                    const float z = heaviside(h - v_th);
                    vv[e] = (1.f - z) * h + z * v_reset;
                    unsigned code = (unsigned)(z != 0.f) + (SKIP ? ((sb[t] >> (2 * e)) & 3u) : 0u);
                    code = code > 3u ? 3u : code;       // contract (ss_neuron.h): z + skip <= 3; a 2-bit field never spills into its neighbour
                    c_spk += (z != 0.f); c_out += (code != 0u);
                    bits |= code << (2 * e);
                    ov[e] = (float)code;
                }
                if (a.out_seq) store_out(reinterpret_cast<f4*>(op) + (long long)t * NV, ov);
                if (a.out_packed) store_packed<4>(a.out_packed + (long long)t * NW, i, bits);
            }
        } else if constexpr (TS > 0) {
            // all T loads of this lane are independent of the recurrence: issue them up front
            vec_t xs[TS];
            vec_t ss[SKIP ? TS : 1];
#pragma unroll
            for (int t = 0; t < TS; ++t) {
                xs[t] = load_stream(xp + (long long)t * NV);
                if (SKIP) ss[t] = sp[(long long)t * NV];
            }
#pragma unroll
            for (int t = 0; t < TS; ++t) {
                vec_t hv, ov;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float x, v, s = 0.f;
                    if constexpr (VEC == 4) { x = xs[t][e]; v = vv[e]; if (SKIP) s = ss[t][e]; }
                    else { x = xs[t]; v = vv; if (SKIP) s = ss[t]; }
                    float h = charge<KIND>(v, x * scale, tau, k, v_reset);
                    float z = heaviside(h - v_th);
                    v = (1.f - z) * h + z * v_reset;
                    float o = SKIP ? z + s : z;
                    c_spk += (z != 0.f); c_out += (o != 0.f);
                    if constexpr (VEC == 4) { hv[e] = h; ov[e] = o; vv[e] = v; } else { hv = h; ov = o; vv = v; }
                }
                if (SAVE_H) store_h(hp + (long long)t * NV, hv);
                store_out(op + (long long)t * NV, ov);
            }
        } else {
            // runtime T: one step of look-ahead, latency otherwise hidden by occupancy
            vec_t xn = xp[0];
            vec_t sn; if (SKIP) sn = sp[0];
            for (int t = 0; t < T; ++t) {
                vec_t xc = xn, sc; if (SKIP) sc = sn;
                if (t + 1 < T) { xn = xp[(long long)(t + 1) * NV]; if (SKIP) sn = sp[(long long)(t + 1) * NV]; }
                vec_t hv, ov;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float x, v, s = 0.f;
                    if constexpr (VEC == 4) { x = xc[e]; v = vv[e]; if (SKIP) s = sc[e]; }
                    else { x = xc; v = vv; if (SKIP) s = sc; }
                    float h = charge<KIND>(v, x * scale, tau, k, v_reset);
                    float z = heaviside(h - v_th);
                    v = (1.f - z) * h + z * v_reset;
                    float o = SKIP ? z + s : z;
                    c_spk += (z != 0.f); c_out += (o != 0.f);
                    if constexpr (VEC == 4) { hv[e] = h; ov[e] = o; vv[e] = v; } else { hv = h; ov = o; vv = v; }
                }
                if (SAVE_H) hp[(long long)t * NV] = hv;
                op[(long long)t * NV] = ov;
            }
        }
        if (!PK || a.v_last) reinterpret_cast<vec_t*>(a.v_last)[i] = vv;      // (the packed training forms may leave the membrane unwritten: ss_neuron_fwd_ex)
    }

    if (a.nnz) count_epilogue(c_spk, c_out, a.nnz, a.cnt_ws);   // wave-uniform
}

// ---------------------------------------------------------------------------------------------------
// 16-bit activation I/O variants (fp16 / bf16 in HBM, fp32 arithmetic and membrane): a lane owns 8 consecutive
// neurons = one 16-B load of x, one 16-B store of out and two 16-B stores of h per time step.
// ---------------------------------------------------------------------------------------------------
// TS > 0: compile-time T, all T (independent) loads of a lane issued before the recurrence starts, like the fp32 kernel; TS = 0: run-time T.
// (the ss_neuron_fwd_ex form with 2-bit packed spike output and / or packed skip input is neuron_fwd16_pk8_kernel, ss_neuron16_v2.hpp)
template <int KIND, int DT, int TS, bool SKIP, bool SAVE_H, int VEC>
__global__ __launch_bounds__(kBlock) void neuron_fwd16_kernel(Fwd16Args a)
{
    typedef typename U16Vec<VEC>::type uvec_t;
    const int T = (TS > 0) ? TS : a.T;
    const long long NV = a.N / VEC;
    const float k = (KIND == SS_KIND_PLIF) ? *a.k : 0.f;
    const float scale = a.scale, tau = a.tau, v_th = a.v_th, v_reset = a.v_reset;
    unsigned c_spk = 0, c_out = 0;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < NV; i += (long long)gridDim.x * kBlock) {
        float v[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = a.v_init ? a.v_init[i * VEC + e] : v_reset;

        auto step = [&](uvec_t xv, uvec_t sv, long long base) {
            uvec_t ov;
            float h[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                unsigned short xb, sb = 0;
                if constexpr (VEC == 1) { xb = xv; if (SKIP) sb = sv; } else { xb = xv[e]; if (SKIP) sb = sv[e]; }
                const float hh = charge<KIND>(v[e], widen<DT>(xb) * scale, tau, k, v_reset);
                const float z = heaviside(hh - v_th);
                v[e] = (1.f - z) * hh + z * v_reset;
                const float o = SKIP ? z + widen<DT>(sb) : z;
                c_spk += (z != 0.f); c_out += (o != 0.f);
                h[e] = hh;
                if constexpr (VEC == 1) ov = narrow<DT>(o); else ov[e] = narrow<DT>(o);
            }
            store_out(reinterpret_cast<uvec_t*>(a.out_seq + base), ov);
            if (SAVE_H) {
                if constexpr (VEC % 4 == 0) {
#pragma unroll
                    for (int q = 0; q < VEC / 4; ++q)
                        *reinterpret_cast<f4*>(a.h_seq + base + 4 * q) = (f4){h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]};
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) a.h_seq[base + e] = h[e];
                }
            }
        };

        if constexpr (TS > 0) {
            uvec_t xs[TS], ss[SKIP ? TS : 1];
#pragma unroll
            for (int t = 0; t < TS; ++t) {
                const long long base = ((long long)t * NV + i) * VEC;
                xs[t] = load_stream(reinterpret_cast<const uvec_t*>(a.x_seq + base));
                if (SKIP) ss[t] = *reinterpret_cast<const uvec_t*>(a.skip_seq + base);
            }
#pragma unroll
            for (int t = 0; t < TS; ++t) step(xs[t], ss[SKIP ? t : 0], ((long long)t * NV + i) * VEC);
        } else {
            for (int t = 0; t < T; ++t) {
                const long long base = ((long long)t * NV + i) * VEC;
                uvec_t xv = load_stream(reinterpret_cast<const uvec_t*>(a.x_seq + base)), sv = xv;
                if (SKIP) sv = *reinterpret_cast<const uvec_t*>(a.skip_seq + base);
                step(xv, sv, base);
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) a.v_last[i * VEC + e] = v[e];
    }
    if (a.nnz) count_epilogue(c_spk, c_out, a.nnz, a.cnt_ws);
}

// ---------------------------------------------------------------------------------------------------
// I-neuron read-out pool
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void ipool_fwd_kernel(const float* pd, long long st, long long sk, const float* v_init,
                                                           float* depth, int T, int K, long long M, float scale, float v_reset)
{
    for (long long m = (long long)blockIdx.x * kBlock + threadIdx.x; m < M; m += (long long)gridDim.x * kBlock) {
        float v = v_init ? v_init[m] : v_reset;
        for (int t = 0; t < T; ++t)
            for (int k = 0; k < K; ++k) {
                float h = v + pd[(long long)t * st + (long long)k * sk + m] * scale;
                v = (1.f - 0.f) * h + 0.f * v_reset;
                depth[((long long)t * K + k) * M + m] = v;
            }
    }
}

__global__ __launch_bounds__(kBlock) void ipool_bwd_kernel(const float* g_depth, const float* g_v_last, float* g_pd,
                                                           long long st, long long sk, float* g_v_init,
                                                           int T, int K, long long M, float scale)
{
    for (long long m = (long long)blockIdx.x * kBlock + threadIdx.x; m < M; m += (long long)gridDim.x * kBlock) {
        float g_v = g_v_last ? g_v_last[m] : 0.f;
        for (int t = T - 1; t >= 0; --t)
            for (int k = K - 1; k >= 0; --k) {
                g_v = g_depth[((long long)t * K + k) * M + m] + g_v;
                g_pd[(long long)t * st + (long long)k * sk + m] = g_v * scale;
            }
        if (g_v_init) g_v_init[m] = g_v;
    }
}

inline int finish_counts(unsigned long long* nnz, unsigned* cnt_ws, int grid, hipStream_t s)
{
    if (!nnz || !cnt_ws) return SS_OK;
    hipLaunchKernelGGL(cnt_finish_kernel, dim3(1), dim3(kBlock), 0, s, cnt_ws, grid, nnz);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

template <int KIND, int TS, bool SKIP, bool SAVE_H>
int launch_fwd(const FwdArgs& a, hipStream_t s)
{
    const bool vec = (a.N % 4 == 0) && aligned16(a.x_seq) && aligned16(a.out_seq) && aligned16(a.v_last) &&
                     (!a.v_init || aligned16(a.v_init)) && (!SKIP || aligned16(a.skip_seq)) &&
                     (!SAVE_H || aligned16(a.h_seq));
    if (a.N == 0) return SS_OK;
    // firing-rate counters: with a partials workspace (ss_neuron_fwd_ex) the launch keeps its full grid; without one every workgroup
    // issues same-address 64-bit atomics, so their number is bounded (measured 3.3 vs 5.6 TB/s at 45 000 workgroups)
    const int cap = (a.nnz && !a.cnt_ws) ? kMaxGridGk : kMaxGrid;
    const int grid = vec ? grid_for(a.N / 4, cap) : grid_for(a.N, cap);
    if (vec) hipLaunchKernelGGL((neuron_fwd_kernel<KIND, TS, SKIP, SAVE_H, 4>), dim3(grid), dim3(kBlock), 0, s, a);
    else     hipLaunchKernelGGL((neuron_fwd_kernel<KIND, TS, SKIP, SAVE_H, 1>), dim3(grid), dim3(kBlock), 0, s, a);
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    return finish_counts(a.nnz, a.cnt_ws, grid, s);
}

// packed spike output / packed skip input (ss_neuron_fwd_ex): 16-B lanes, compile-time T, no saved h
template <int KIND, int TS, bool SKIP>
int launch_fwd_pk(const FwdArgs& a, hipStream_t s)
{
    if constexpr (TS == 0) return SS_EINVAL;
    else {
        if (a.N == 0) return SS_OK;
        const bool ok = (a.N % 16 == 0) && aligned16(a.x_seq) && aligned16(a.out_seq) && (!a.v_last || aligned16(a.v_last)) &&
                        (!a.v_init || aligned16(a.v_init)) && (!a.skip_seq || aligned16(a.skip_seq)) && !a.h_seq;
        if (!ok) return SS_EINVAL;
        const int cap = (a.nnz && !a.cnt_ws) ? kMaxGridGk : kMaxGrid;
        const int grid = grid_for(a.N / 4, cap);
        hipLaunchKernelGGL((neuron_fwd_kernel<KIND, TS, SKIP, false, 4, true>), dim3(grid), dim3(kBlock), 0, s, a);
        if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
        return finish_counts(a.nnz, a.cnt_ws, grid, s);
    }
}

template <int KIND, int TS>
int dispatch_fwd_flags(const FwdArgs& a, hipStream_t s)
{
    const bool skip = a.skip_seq != nullptr || a.skip_packed != nullptr, save_h = a.h_seq != nullptr;
    if (a.out_packed || a.skip_packed) return skip ? launch_fwd_pk<KIND, TS, true>(a, s) : launch_fwd_pk<KIND, TS, false>(a, s);
    if (skip) return save_h ? launch_fwd<KIND, TS, true, true>(a, s) : launch_fwd<KIND, TS, true, false>(a, s);
    return save_h ? launch_fwd<KIND, TS, false, true>(a, s) : launch_fwd<KIND, TS, false, false>(a, s);
}

template <int KIND>
int dispatch_fwd_T(const FwdArgs& a, hipStream_t s)
{
    switch (a.T) {
        case 1: return dispatch_fwd_flags<KIND, 1>(a, s);
        case 2: return dispatch_fwd_flags<KIND, 2>(a, s);
        case 4: return dispatch_fwd_flags<KIND, 4>(a, s);
        case 5: return dispatch_fwd_flags<KIND, 5>(a, s);
        case 8: return dispatch_fwd_flags<KIND, 8>(a, s);
        case 10: return dispatch_fwd_flags<KIND, 10>(a, s);
        default: return dispatch_fwd_flags<KIND, 0>(a, s);
    }
}

constexpr int kFwd16Pk8Grid = 4096;
#ifndef SS_F16_V5
#define SS_F16_V5 8
#endif
#ifndef SS_F16_V10
#define SS_F16_V10 4
#endif
template <int KIND, int DT, int TS>
int launch_fwd16(const Fwd16Args& a, hipStream_t s)
{
    // lane width: 8 neurons (one 16-B load per step) while all T loads fit comfortably in registers, 4 for the longer sequences
    constexpr int V = (TS == 0 || TS <= 5) ? SS_F16_V5 : SS_F16_V10;
    const bool skip = a.skip_seq != nullptr || a.skip_packed != nullptr, save_h = a.h_seq != nullptr;
    const bool vec = (a.N % V == 0) && aligned16(a.x_seq) && aligned16(a.out_seq) && aligned16(a.v_last) &&
                     (!a.v_init || aligned16(a.v_init)) && (!a.skip_seq || aligned16(a.skip_seq)) && (!save_h || aligned16(a.h_seq));
    const int cap = (a.nnz && !a.cnt_ws) ? kMaxGridGk : kMaxGrid;
    const int grid = vec ? grid_for(a.N / V, cap) : grid_for(a.N, cap);
    if (!(a.out_packed || a.skip_packed) && !a.v_last) return SS_EINVAL;      // v_last is optional in the packed forms only
    if (a.out_packed || a.skip_packed) {                  // packed spike I/O: 16-B lanes, compile-time T, no saved h
        if constexpr (TS == 0) return SS_EINVAL;
        else {
            // round 6: neuron_fwd16_pk8_kernel (ss_neuron16_v2.hpp) — 8 neurons per lane at every T, 10 - 12 VALU instructions per update instead of 33
            const bool vec8 = (a.N % 16 == 0) && aligned16(a.x_seq) && (!a.out_seq || aligned16(a.out_seq)) && (!a.v_last || aligned16(a.v_last)) &&
                              (!a.v_init || aligned16(a.v_init)) && (!a.skip_seq || aligned16(a.skip_seq));
            if (!vec8 || save_h) return SS_EINVAL;
            // a bounded grid with a grid-stride loop: at 10 instructions per update a wavefront's start-up is no longer hidden — 4096 workgroups (2 - 3 rounds of
            // resident ones) instead of one vector per lane: 0.68 -> 0.74 of 8 TB/s at T = 10, 0.54 -> 0.65 at T = 5 (profiles/r06/neuron16_fwd_grid_ab.log)
            const int grid8 = grid_for(a.N / 8, cap < kFwd16Pk8Grid ? cap : kFwd16Pk8Grid);
            if (a.out_seq) {
                if (skip) hipLaunchKernelGGL((neuron_fwd16_pk8_kernel<KIND, DT, TS, true, true>), dim3(grid8), dim3(kBlock), 0, s, a);
                else hipLaunchKernelGGL((neuron_fwd16_pk8_kernel<KIND, DT, TS, false, true>), dim3(grid8), dim3(kBlock), 0, s, a);
            } else {
                if (skip) hipLaunchKernelGGL((neuron_fwd16_pk8_kernel<KIND, DT, TS, true, false>), dim3(grid8), dim3(kBlock), 0, s, a);
                else hipLaunchKernelGGL((neuron_fwd16_pk8_kernel<KIND, DT, TS, false, false>), dim3(grid8), dim3(kBlock), 0, s, a);
            }
            if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
            return finish_counts(a.nnz, a.cnt_ws, grid8, s);
        }
    }
#define SS_L16(SK, SH) do { if (vec) hipLaunchKernelGGL((neuron_fwd16_kernel<KIND, DT, TS, SK, SH, V>), dim3(grid), dim3(kBlock), 0, s, a); \
                            else hipLaunchKernelGGL((neuron_fwd16_kernel<KIND, DT, TS, SK, SH, 1>), dim3(grid), dim3(kBlock), 0, s, a); } while (0)
    if (skip) { if (save_h) SS_L16(true, true); else SS_L16(true, false); }
    else      { if (save_h) SS_L16(false, true); else SS_L16(false, false); }
#undef SS_L16
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    return finish_counts(a.nnz, a.cnt_ws, grid, s);
}

template <int KIND, int DT>
int dispatch_fwd16(const Fwd16Args& a, hipStream_t s)
{
    switch (a.T) {
        case 1: return launch_fwd16<KIND, DT, 1>(a, s);
        case 2: return launch_fwd16<KIND, DT, 2>(a, s);
        case 4: return launch_fwd16<KIND, DT, 4>(a, s);
        case 5: return launch_fwd16<KIND, DT, 5>(a, s);
        case 8: return launch_fwd16<KIND, DT, 8>(a, s);
        case 10: return launch_fwd16<KIND, DT, 10>(a, s);
        default: return launch_fwd16<KIND, DT, 0>(a, s);
    }
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

int ss_abi_version(void) { return SS_ABI_VERSION; }

int ss_neuron_fwd_f32(const float* x_seq, const float* v_init, const float* skip_seq,
                      float* out_seq, float* h_seq, float* v_last, unsigned long long* nnz,
                      int T, long long N, float scale, int kind, float tau, const float* k,
                      float v_th, float v_reset, void* stream)
{
    if (!x_seq || !out_seq || !v_last || T <= 0 || N < 0) return SS_EINVAL;
    if (kind < SS_KIND_IF || kind > SS_KIND_PLIF) return SS_EINVAL;
    if (kind == SS_KIND_PLIF && !k) return SS_EINVAL;
    if (out_seq == x_seq) return SS_EINVAL;
    FwdArgs a{x_seq, v_init, skip_seq, out_seq, h_seq, v_last, nnz, T, N, scale, tau, v_th, v_reset, k, nullptr, nullptr, nullptr};
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (kind) {
        case SS_KIND_IF: return dispatch_fwd_T<SS_KIND_IF>(a, s);
        case SS_KIND_LIF: return dispatch_fwd_T<SS_KIND_LIF>(a, s);
        default: return dispatch_fwd_T<SS_KIND_PLIF>(a, s);
    }
}

int ss_neuron_fwd_x16(const void* x_seq, const float* v_init, const void* skip_seq,
                      void* out_seq, float* h_seq, float* v_last, unsigned long long* nnz,
                      int T, long long N, float scale, int kind, float tau, const float* k,
                      float v_th, float v_reset, int dtype, void* stream)
{
    if (!x_seq || !out_seq || !v_last || T <= 0 || N < 0) return SS_EINVAL;
    if (kind < SS_KIND_IF || kind > SS_KIND_PLIF || (dtype != SS_DT_F16 && dtype != SS_DT_BF16)) return SS_EINVAL;
    if (kind == SS_KIND_PLIF && !k) return SS_EINVAL;
    if (out_seq == x_seq) return SS_EINVAL;
    if (N == 0) return SS_OK;
    Fwd16Args a{static_cast<const unsigned short*>(x_seq), v_init, static_cast<const unsigned short*>(skip_seq),
                static_cast<unsigned short*>(out_seq), h_seq, v_last, nnz, T, N, scale, tau, v_th, v_reset, k, nullptr, nullptr, nullptr};
    hipStream_t s = static_cast<hipStream_t>(stream);
#define SS_D16(KK) (dtype == SS_DT_F16 ? dispatch_fwd16<KK, SS_DT_F16>(a, s) : dispatch_fwd16<KK, SS_DT_BF16>(a, s))
    switch (kind) {
        case SS_KIND_IF: return SS_D16(SS_KIND_IF);
        case SS_KIND_LIF: return SS_D16(SS_KIND_LIF);
        default: return SS_D16(SS_KIND_PLIF);
    }
#undef SS_D16
}

long long ss_neuron_cnt_ws_words(long long N)
{
    if (N < 0) return 0;
    return 2 * ((N + kBlock - 1) / kBlock) + 2;          // covers the scalar launch (one neuron per lane); the 16-B form uses a quarter
}

int ss_neuron_fwd_ex(const ss_neuron_fwd_desc* d, void* stream)
{
    if (!d || d->size != sizeof(ss_neuron_fwd_desc)) return SS_EINVAL;
    if (!d->x_seq || d->T <= 0 || d->N < 0 || (!d->out_seq && !d->out_packed)) return SS_EINVAL;
    if (!d->v_last && !(d->out_packed || d->skip_packed)) return SS_EINVAL;      // v_last == NULL (ABI 10): the packed forms only — the membrane after step T is not written
    if (d->kind < SS_KIND_IF || d->kind > SS_KIND_PLIF || (d->kind == SS_KIND_PLIF && !d->k)) return SS_EINVAL;
    if (d->out_seq == d->x_seq || (d->skip_seq && d->skip_packed)) return SS_EINVAL;
    if (d->cnt_ws && !d->nnz) return SS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (d->act_dtype == 0) {
        if ((d->out_packed || d->skip_packed) && (d->h_seq || d->N % 16 != 0 || !ss_neuron_bwd_rc_supported(d->T))) return SS_EINVAL;
        FwdArgs a{static_cast<const float*>(d->x_seq), d->v_init, static_cast<const float*>(d->skip_seq), static_cast<float*>(d->out_seq),
                  d->h_seq, d->v_last, d->nnz, d->T, d->N, d->scale, d->tau, d->v_th, d->v_reset, d->k, d->skip_packed, d->out_packed, d->cnt_ws};
        switch (d->kind) {
            case SS_KIND_IF: return dispatch_fwd_T<SS_KIND_IF>(a, s);
            case SS_KIND_LIF: return dispatch_fwd_T<SS_KIND_LIF>(a, s);
            default: return dispatch_fwd_T<SS_KIND_PLIF>(a, s);
        }
    }
    if (d->act_dtype != SS_DT_F16 && d->act_dtype != SS_DT_BF16) return SS_EINVAL;
    // packed I/O on 16-bit activations (ABI 9): the same contract as the fp32 form — whole words, compile-time T, no saved h
    if ((d->out_packed || d->skip_packed) && (d->h_seq || d->N % 16 != 0 || !ss_neuron_bwd_rc_supported(d->T))) return SS_EINVAL;
    if (d->N == 0) return SS_OK;
    Fwd16Args a{static_cast<const unsigned short*>(d->x_seq), d->v_init, static_cast<const unsigned short*>(d->skip_seq),
                static_cast<unsigned short*>(d->out_seq), d->h_seq, d->v_last, d->nnz, d->T, d->N, d->scale, d->tau, d->v_th, d->v_reset, d->k, d->cnt_ws,
                d->skip_packed, d->out_packed};
#define SS_D16(KK) (d->act_dtype == SS_DT_F16 ? dispatch_fwd16<KK, SS_DT_F16>(a, s) : dispatch_fwd16<KK, SS_DT_BF16>(a, s))
    switch (d->kind) {
        case SS_KIND_IF: return SS_D16(SS_KIND_IF);
        case SS_KIND_LIF: return SS_D16(SS_KIND_LIF);
        default: return SS_D16(SS_KIND_PLIF);
    }
#undef SS_D16
}

int ss_ipool_fwd_f32(const float* pd_seq, long long stride_t, long long stride_k, const float* v_init,
                     float* depth_seq, int T, int K, long long M, float scale, float v_reset, void* stream)
{
    if (!pd_seq || !depth_seq || T <= 0 || K <= 0 || M < 0) return SS_EINVAL;
    if (M == 0) return SS_OK;
    hipLaunchKernelGGL(ipool_fwd_kernel, dim3(grid_for(M)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       pd_seq, stride_t, stride_k, v_init, depth_seq, T, K, M, scale, v_reset);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_ipool_bwd_f32(const float* g_depth_seq, const float* g_v_last, float* g_pd_seq,
                     long long stride_t, long long stride_k, float* g_v_init,
                     int T, int K, long long M, float scale, void* stream)
{
    if (!g_depth_seq || !g_pd_seq || T <= 0 || K <= 0 || M < 0) return SS_EINVAL;
    if (M == 0) return SS_OK;
    hipLaunchKernelGGL(ipool_bwd_kernel, dim3(grid_for(M)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       g_depth_seq, g_v_last, g_pd_seq, stride_t, stride_k, g_v_init, T, K, M, scale);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

}  // extern "C"

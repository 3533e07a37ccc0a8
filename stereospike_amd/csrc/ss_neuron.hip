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
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <math.h>
#include <type_traits>

#include "ss_neuron.h"

namespace {

#ifndef SS_BLOCK
#define SS_BLOCK 256
#endif
#ifndef SS_MAX_GRID
#define SS_MAX_GRID 1048576
#endif
#ifndef SS_MAX_GRID_BWD
#define SS_MAX_GRID_BWD 1048576
#endif
constexpr int kBlock = SS_BLOCK;         // 4 wavefronts of 64
// Grid caps, A/B-measured in one process with interleaved rounds (profiles/r01/neuron_grid_variants*.log, neuron_variants_v2/v3.log):
// both kernels are best with one vector per lane and no grid-stride loop (forward 8 B/update form: +4 % over a 32768-workgroup cap;
// backward +9 % over 2048).
constexpr int kMaxGrid = SS_MAX_GRID;
constexpr int kMaxGridBwd = SS_MAX_GRID_BWD;
constexpr int kMaxGridGk = 2048;         // PLIF dL/dk: bounded number of workgroup partials (caller workspace, fixed-order 2nd pass)
constexpr long long kGkWsFloats = kMaxGridGk;

typedef float f4 __attribute__((ext_vector_type(4)));

// Streaming-access policy of the neuron kernels (A/B-measured with tools/bench_kernels.py, see profiles/):
//   SS_NT_H  : h_seq is written once and read only by the backward pass, much later => non-temporal store
//   SS_NT_X  : x_seq (conv output) / g_out are read exactly once                   => non-temporal load
//   SS_NT_OUT: out_seq / g_x_seq are written once and read by the NEXT kernel (a conv / GEMM), far larger than the caches at the
//              layers that matter                                                  => non-temporal store
// Default on for X and OUT since the round-1 A/B of the shipped forms (forward 8 B/update, forked recompute backward, 2.3e8 updates;
// profiles/r01/neuron_variants_v3.log): forward 5.48 -> 6.55 TB/s, backward 5.18 -> 5.50 TB/s; inside bench.py the 260 neuron launches
// of a step 4.52 -> 4.24 ms.
#ifndef SS_NT_H
#define SS_NT_H 1
#endif
#ifndef SS_NT_X
#define SS_NT_X 1
#endif
#ifndef SS_NT_OUT
#define SS_NT_OUT 1
#endif
template <typename V> __device__ __forceinline__ void store_out(V* p, V v)
{
#if SS_NT_OUT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
template <typename V> __device__ __forceinline__ V load_stream(const V* p)
{
#if SS_NT_X
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <typename V> __device__ __forceinline__ void store_h(V* p, V v)
{
#if SS_NT_H
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// ---------------------------------------------------------------------------------------------------
// element-wise pieces (scalar; applied to each of a lane's 4 neurons)
// ---------------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ float charge(float v, float xs, float tau, float k, float v_reset)
{
    if (KIND == SS_KIND_IF) return v + xs;
    float d = xs - (v - v_reset);                 // (v - 0.f) == v bit-for-bit: covers both upstream branches
    if (KIND == SS_KIND_LIF) return v + d / tau;  // IEEE-correct division (true division in the CPU reference)
    return v + d * k;
}

__device__ __forceinline__ float heaviside(float xh) { return (xh >= 0.f) ? 1.f : 0.f; }

template <int SG>
__device__ __forceinline__ float surrogate_grad(float xh, float alpha, float c_atan, float half_alpha, float g)
{
    if (SG == SS_SG_ATAN) {
        float u = xh * c_atan;
        float p = u * u;
        float r = 1.f / (p + 1.f);
        return (r * half_alpha) * g;
    }
    float s = 1.f / (1.f + expf(-(xh * alpha)));
    return ((g * (1.f - s)) * s) * alpha;
}

__device__ __forceinline__ unsigned wave_sum_u32(unsigned v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_sum_f32(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

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
                        sb[t] = (unsigned)sv[0] | ((unsigned)sv[1] << 2) | ((unsigned)sv[2] << 4) | ((unsigned)sv[3] << 6);
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
                    const unsigned code = (unsigned)(z != 0.f) + (SKIP ? ((sb[t] >> (2 * e)) & 3u) : 0u);
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
        reinterpret_cast<vec_t*>(a.v_last)[i] = vv;
    }

    if (a.nnz) count_epilogue(c_spk, c_out, a.nnz, a.cnt_ws);   // wave-uniform
}

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
struct BwdArgs {
    const float* g_out_seq; const float* g_v_last; const float* h_seq; const float* v_init;
    float* g_x_seq; float* g_v_init; float* g_k_partials;
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
    float acc_k = 0.f;
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
                        acc_k += g_h * ((he - v_prev) / k);
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

    if (want_gk) {   // wave-uniform
        __shared__ float s_k[kBlock / 64];
        float w = wave_sum_f32(acc_k);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) s_k[wave] = w;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < kBlock / 64; ++q) s += s_k[q];
            a.g_k_partials[blockIdx.x] = s;
        }
    }
}

// second pass of the dL/dk reduction: fixed order -> bit-reproducible.  tail = scalar-tail kernel's partial.
__global__ __launch_bounds__(kBlock) void gk_finish_kernel(const float* partials, int n, float* g_k)
{
    __shared__ float s[kBlock];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += kBlock) acc += partials[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = kBlock / 2; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *g_k = s[0];
}

// ---------------------------------------------------------------------------------------------------
// 16-bit activation I/O variants (fp16 / bf16 in HBM, fp32 arithmetic and membrane): a lane owns 8 consecutive
// neurons = one 16-B load of x, one 16-B store of out and two 16-B stores of h per time step.
// ---------------------------------------------------------------------------------------------------
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
template <int VEC> struct U16Vec;
template <> struct U16Vec<8> { typedef u16x8 type; };
template <> struct U16Vec<4> { typedef unsigned short type __attribute__((ext_vector_type(4))); };
template <> struct U16Vec<2> { typedef unsigned short type __attribute__((ext_vector_type(2))); };
template <> struct U16Vec<1> { typedef unsigned short type; };

template <int DT> __device__ __forceinline__ float widen(unsigned short b)
{
    if (DT == SS_DT_F16) return __half2float(__ushort_as_half(b));
    return __uint_as_float((unsigned)b << 16);
}
template <int DT> __device__ __forceinline__ unsigned short narrow(float f)
{
    // The value to store is an fp32 result (rounded once already); keep hipcc from folding the producing multiply into
    // v_fma_mixlo_f16, which would round the exact product straight to fp16 (single rounding) and break bit-parity with
    // the "fp32 arithmetic, nearest-even narrowing on store" definition of oracle/np_x16.py (seen on the MI355X: ~1e-6
    // of the g_x values differed by one fp16 ulp).
    asm volatile("" : "+v"(f));
    if (DT == SS_DT_F16) return __half_as_ushort(__float2half_rn(f));
    unsigned u = __float_as_uint(f);                       // round to nearest even (NaN kept quiet)
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

struct Fwd16Args {
    const unsigned short* x_seq; const float* v_init; const unsigned short* skip_seq;
    unsigned short* out_seq; float* h_seq; float* v_last; unsigned long long* nnz;
    int T; long long N;
    float scale, tau, v_th, v_reset; const float* k;
    unsigned* cnt_ws;              // nullable (with nnz): per-workgroup counter partials (ss_neuron_fwd_ex)
};

// TS > 0: compile-time T, all T (independent) loads of a lane issued before the recurrence starts, like the fp32 kernel; TS = 0: run-time T.
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

struct Bwd16Args {
    const unsigned short* g_out_seq; const float* g_v_last; const float* h_seq; const float* v_init;
    unsigned short* g_x_seq; float* g_v_init; float* g_k_partials;
    int T; long long N;
    float scale, tau, v_th, v_reset, alpha; const float* k; int detach_reset;
};

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
    float acc_k = 0.f;
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
                        acc_k += g_h * ((h[e] - v_prev) / k);
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
    if (want_gk) {
        __shared__ float s_k[kBlock / 64];
        float w = wave_sum_f32(acc_k);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) s_k[wave] = w;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < kBlock / 64; ++q) s += s_k[q];
            a.g_k_partials[blockIdx.x] = s;
        }
    }
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
template <int KIND, int SG, int DT, int TS, int VEC, bool G2 = false>
__global__ __launch_bounds__(kBlock) void neuron_bwd16_rc_kernel(Bwd16Args a, const unsigned short* __restrict__ x_seq,
                                                                 const unsigned short* __restrict__ g_out2_seq, unsigned short* __restrict__ g_sum_seq)
{
    typedef typename U16Vec<VEC>::type uvec_t;
    const long long NV = a.N / VEC;
    const float k = (KIND == SS_KIND_PLIF) ? *a.k : 0.f;
    const float scale = a.scale, tau = a.tau, v_th = a.v_th, v_reset = a.v_reset, alpha = a.alpha;
    const float c_atan = (float)(M_PI / 2.0 * (double)alpha);
    const float half_alpha = (float)((double)alpha / 2.0);
    const bool detach = a.detach_reset != 0;
    const bool want_gk = (KIND == SS_KIND_PLIF) && a.g_k_partials != nullptr;
    float acc_k = 0.f;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < NV; i += (long long)gridDim.x * kBlock) {
        uvec_t xs[TS], gs[TS];
#pragma unroll
        for (int t = 0; t < TS; ++t) xs[t] = load_stream(reinterpret_cast<const uvec_t*>(x_seq + ((long long)t * NV + i) * VEC));
#pragma unroll
        for (int t = TS - 1; t >= 0; --t) gs[t] = load_stream(reinterpret_cast<const uvec_t*>(a.g_out_seq + ((long long)t * NV + i) * VEC));
        uvec_t g2[G2 ? TS : 1];
        if constexpr (G2) {
#pragma unroll
            for (int t = TS - 1; t >= 0; --t) g2[t] = load_stream(reinterpret_cast<const uvec_t*>(g_out2_seq + ((long long)t * NV + i) * VEC));
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
                unsigned short gb;
                if constexpr (VEC == 1) gb = gs[t]; else gb = gs[t][e];
                const float he = h[t][e];
                const float xh = he - v_th;
                const float z = heaviside(xh);
                float g_s = widen<DT>(gb);
                if constexpr (G2) {
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
                        acc_k += g_h * ((he - v_prev) / k);
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
    if (want_gk) {
        __shared__ float s_k[kBlock / 64];
        float w = wave_sum_f32(acc_k);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) s_k[wave] = w;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < kBlock / 64; ++q) s += s_k[q];
            a.g_k_partials[blockIdx.x] = s;
        }
    }
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

// ---------------------------------------------------------------------------------------------------
// predict_depth head: nearest-upsample + valid kxk conv to one channel, as a gather over per-tap projections
// ---------------------------------------------------------------------------------------------------
// Index decoding uses 32-bit arithmetic only (one image = blockIdx.y, at most 2^31 elements per image): 64-bit
// div/mod per element made the first version of these kernels ALU-bound (profiles/r01/kernel_stats_v1.csv).
template <int K>
__global__ __launch_bounds__(kBlock) void upconv1_fwd_kernel(const float* __restrict__ P, const int* __restrict__ src_y,
                                                             const int* __restrict__ src_x, const float* __restrict__ bias,
                                                             float* __restrict__ out, int NB, int h, int w, int H, int W)
{
    const unsigned pix = blockIdx.x * kBlock + threadIdx.x;
    if (pix >= (unsigned)(H * W)) return;
    const unsigned y = pix / (unsigned)W, x = pix - y * (unsigned)W;
    const float b = bias ? *bias : 0.f;
    int sx[K], sy[K];
#pragma unroll
    for (int q = 0; q < K; ++q) { sx[q] = src_x[x + q]; sy[q] = src_y[y + q] * w; }
    const unsigned hw = (unsigned)(h * w);
    for (int img = blockIdx.y; img < NB; img += gridDim.y) {
        const float* Pn = P + (long long)img * (K * K) * hw;
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
                acc += Pn[(unsigned)(ky * K + kx) * hw + (unsigned)(sy[ky] + sx[kx])];
        out[(long long)img * (H * W) + pix] = acc + b;
    }
}

// Adjoint of the gather.  One workgroup owns a 4 x 64 tile of SOURCE pixels of one image: the g_out window that maps to
// it ((4*rep + K-1) rows x (64*rep + K-1) columns, rep ~ 2 for the decoder stages) is staged once in LDS with coalesced
// row reads; every lane then forms its K*K rectangle sums from LDS and writes K*K coalesced planes of g_P.
// Each rectangle sum is evaluated row-sums-first: C[y] = sum_x g[y][x] (x ascending), then sum_y C[y] (y ascending) —
// the order oracle/ss_neuron_ref.c restates.  Fast path (block-uniform): when every source pixel of the tile is
// replicated at most 3 times per axis, a lane reads its (K+2) x (K+2) window once with static indexing and shares the
// row sums between the K*K taps (49 LDS reads instead of ~106 for K = 5).  Windows that do not fit the LDS tile
// (large up-sampling ratios, e.g. predict_depth4's 7.9x) read g_out directly.
constexpr int kBwdTileY = 4, kBwdTileX = 64, kBwdLds = 6144;      // 24 KiB of LDS

template <int K>
__global__ __launch_bounds__(kBlock) void upconv1_bwd_kernel(const float* __restrict__ g_out, const int* __restrict__ y_lo,
                                                             const int* __restrict__ y_hi, const int* __restrict__ x_lo,
                                                             const int* __restrict__ x_hi, float* __restrict__ g_P,
                                                             int NB, int h, int w, int H, int W)
{
    __shared__ float tile[kBwdLds];
    __shared__ int s_small;
    const int tx = threadIdx.x & (kBwdTileX - 1), ty = threadIdx.x >> 6;
    const int ix0 = blockIdx.x * kBwdTileX, iy0 = blockIdx.y * kBwdTileY;
    const int ix = ix0 + tx, iy = iy0 + ty;
    const bool valid = ix < w && iy < h;
    // window of g_out covered by this source tile (block-uniform)
    const int iyl = min(iy0 + kBwdTileY, h) - 1, ixl = min(ix0 + kBwdTileX, w) - 1;
    const int r0 = max(y_lo[iy0] - (K - 1), 0), r1 = min(y_hi[iyl], H);
    const int c0 = max(x_lo[ix0] - (K - 1), 0), c1 = min(x_hi[ixl], W);
    const int rh = max(r1 - r0, 0), rw = max(c1 - c0, 0);
    const bool fits = rh * rw <= kBwdLds;
    int ylo = 0, yhi = 0, xlo = 0, xhi = 0;
    if (valid) { ylo = y_lo[iy]; yhi = y_hi[iy]; xlo = x_lo[ix]; xhi = x_hi[ix]; }
    const int ry = yhi - ylo, rx = xhi - xlo;                         // replication counts of this source pixel
    if (threadIdx.x == 0) s_small = 1;
    __syncthreads();
    if (ry > 3 || rx > 3) s_small = 0;                                // benign race: all writers store 0
    __syncthreads();
    const bool small = fits && s_small != 0;
    const unsigned hw = (unsigned)(h * w);
    for (int img = blockIdx.z; img < NB; img += gridDim.z) {
        const float* g = g_out + (long long)img * (H * W);
        if (fits) {
            for (int r = ty; r < rh; r += kBlock / kBwdTileX)
                for (int c = tx; c < rw; c += kBwdTileX) tile[r * rw + c] = g[(r0 + r) * W + (c0 + c)];
            __syncthreads();
        }
        if (valid) {
            float* gp = g_P + (long long)img * (K * K) * hw + (unsigned)(iy * w + ix);
            if (small) {
                // window rows ylo-(K-1) .. ylo+2, cols xlo-(K-1) .. xlo+2 (zero outside the image / beyond the replication)
                float acc[K][K];
#pragma unroll
                for (int r = 0; r < K + 2; ++r) {
                    const int y = ylo - (K - 1) + r;
                    const bool yok = y >= 0 && y < H && (y - r0) < rh;
                    float row[K + 2];
#pragma unroll
                    for (int c = 0; c < K + 2; ++c) {
                        const int x = xlo - (K - 1) + c;
                        const bool ok = yok && x >= 0 && x < W && (x - c0) < rw;
                        row[c] = ok ? tile[(y - r0) * rw + (x - c0)] : 0.f;
                    }
                    float C[K];                                        // row sums for the K horizontal taps
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        const int b0 = K - 1 - kx;                     // first column of tap kx inside the window
                        float cs = 0.f;
                        if (rx > 0) cs += row[b0];
                        if (rx > 1) cs += row[b0 + 1];
                        if (rx > 2) cs += row[b0 + 2];
                        C[kx] = cs;
                    }
#pragma unroll
                    for (int ky = 0; ky < K; ++ky) {
                        const int a = r - (K - 1) + ky;                // which row of tap ky's rectangle this window row is
                        if (a == 0) {
#pragma unroll
                            for (int kx = 0; kx < K; ++kx) { acc[ky][kx] = 0.f; if (ry > 0) acc[ky][kx] += C[kx]; }
                        } else if (a == 1 || a == 2) {
                            if (ry > a) {
#pragma unroll
                                for (int kx = 0; kx < K; ++kx) acc[ky][kx] += C[kx];
                            }
                        }
                    }
                }
#pragma unroll
                for (int ky = 0; ky < K; ++ky)
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) gp[(unsigned)(ky * K + kx) * hw] = acc[ky][kx];
            } else {
#pragma unroll
                for (int ky = 0; ky < K; ++ky) {
                    const int y0 = max(ylo - ky, 0), y1 = min(yhi - ky, H);
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        const int x0 = max(xlo - kx, 0), x1 = min(xhi - kx, W);
                        float acc = 0.f;
                        for (int y = y0; y < y1; ++y) {
                            float cs = 0.f;
                            if (fits) { for (int x = x0; x < x1; ++x) cs += tile[(y - r0) * rw + (x - c0)]; }
                            else      { for (int x = x0; x < x1; ++x) cs += g[y * W + x]; }
                            acc += cs;
                        }
                        gp[(unsigned)(ky * K + kx) * hw] = acc;
                    }
                }
            }
        }
        if (fits) __syncthreads();
    }
}

// XCD-aware workgroup remap (guide T1): the dispatcher places workgroup b on XCD b % 8, each XCD with a private L2.  The gather kernels'
// neighbouring workgroups (adjacent pixels of a row, and the rows above / below) read the same P / g_out lines, so each XCD is given a
// CONTIGUOUS chunk of the linear workgroup range instead of every 8th one.  Bijective for any workgroup count.  Measured
// (profiles/r01/bench_gather_xcd.log): forward gather 3.24 -> 2.76 ms per step over the four decoder stages; the (write-bound) adjoint +1 %.
#ifndef SS_CL_BWD_ROWSCAN
#define SS_CL_BWD_ROWSCAN 1
#endif
// gather outputs (out / g_P): non-temporal stores make the kernels faster in isolation (adjoint 363 -> 271 us at deconv3) but the STEP
// slower (55.3 -> 56.4 ms): their consumer runs right after and finds part of the tensor in the 256 MiB Infinity Cache.  Off.
// (profiles/r01/nt_gather_ab.log)
#ifndef SS_NT_GATHER
#define SS_NT_GATHER 0
#endif
template <typename V> __device__ __forceinline__ void store_gather(V* p, V v)
{
#if SS_NT_GATHER
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
#ifndef SS_XCD_REMAP
#define SS_XCD_REMAP 1
#endif
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg)
{
#if SS_XCD_REMAP
    const unsigned xcd = bid & 7u, q = nwg >> 3, r = nwg & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
#else
    return bid;
#endif
}

// ---------------------------------------------------------------------------------------------------
// channels-last gather kernels (decoder kept in NHWC: P is the row-major output of ONE GEMM, no transposes)
// ---------------------------------------------------------------------------------------------------
// ACT = 0: fp32 activations; SS_DT_F16 / SS_DT_BF16: the stage OUTPUT (forward) / its gradient (adjoint) live in HBM as 16-bit values
// (16-bit activation modes: the following neuron layer then runs its x16 kernels); P, g_P and all sums stay fp32.
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
template <int ACT, int VEC> __device__ __forceinline__ typename std::conditional<VEC == 4, f4, float>::type
load_act(const void* base, long long e)
{
    if constexpr (ACT == 0) {
        if constexpr (VEC == 4) return *reinterpret_cast<const f4*>(static_cast<const float*>(base) + e);
        else return static_cast<const float*>(base)[e];
    } else {
        const unsigned short* p = static_cast<const unsigned short*>(base) + e;
        if constexpr (VEC == 4) {
            const u16x4 v = *reinterpret_cast<const u16x4*>(p);
            return (f4){widen<ACT>(v[0]), widen<ACT>(v[1]), widen<ACT>(v[2]), widen<ACT>(v[3])};
        } else return widen<ACT>(*p);
    }
}
template <int ACT, int VEC> __device__ __forceinline__ void
store_act(void* base, long long e, typename std::conditional<VEC == 4, f4, float>::type v)
{
    if constexpr (ACT == 0) {
        if constexpr (VEC == 4) store_gather(reinterpret_cast<f4*>(static_cast<float*>(base) + e), v);
        else static_cast<float*>(base)[e] = v;
    } else {
        unsigned short* p = static_cast<unsigned short*>(base) + e;
        if constexpr (VEC == 4) {
            u16x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = narrow<ACT>(v[i]);
            *reinterpret_cast<u16x4*>(p) = o;
        } else *p = narrow<ACT>(v);
    }
}

template <int K, int VEC, int ACT = 0>
__global__ __launch_bounds__(kBlock) void upconv_cl_fwd_kernel(const float* __restrict__ P, const int* __restrict__ src_y,
                                                               const int* __restrict__ src_x, const float* __restrict__ bias,
                                                               void* __restrict__ out, int NB, int C, int h, int w, int H, int W)
{
    typedef typename std::conditional<VEC == 4, f4, float>::type vec_t;
    const unsigned CV = (unsigned)C / VEC;                            // channel vectors per pixel
    const unsigned idx = xcd_remap(blockIdx.x, gridDim.x) * kBlock + threadIdx.x;   // (pixel, channel vector) of one image
    if (idx >= (unsigned)(H * W) * CV) return;
    const unsigned pix = idx / CV, cv = idx - pix * CV;
    const unsigned y = pix / (unsigned)W, x = pix - y * (unsigned)W;
    const unsigned KKC = (unsigned)(K * K * C);
    unsigned off[K][K];                                               // source pixel offsets (in floats) per tap
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const unsigned sy = (unsigned)src_y[y + ky] * (unsigned)w;
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
            off[ky][kx] = (sy + (unsigned)src_x[x + kx]) * KKC + (unsigned)((ky * K + kx) * C) + cv * VEC;
    }
    vec_t b;
    if constexpr (VEC == 4) b = bias ? *reinterpret_cast<const f4*>(bias + cv * 4) : (f4){0.f, 0.f, 0.f, 0.f};
    else b = bias ? bias[cv] : 0.f;
    for (int img = blockIdx.y; img < NB; img += gridDim.y) {
        const float* Pn = P + (long long)img * h * w * KKC;
        vec_t acc;
        if constexpr (VEC == 4) acc = (f4){0.f, 0.f, 0.f, 0.f}; else acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) acc += *reinterpret_cast<const vec_t*>(Pn + off[ky][kx]);
        store_act<ACT, VEC>(out, ((long long)img * (H * W) + pix) * C + cv * VEC, (vec_t)(acc + b));
    }
}

// GP = 0: g_P written as fp32; SS_DT_BF16: g_P written as bf16 (the 16-bit modes' backward GEMMs take bf16 operands: no separate cast pass)
template <int K, int VEC, int ACT = 0, int GP = 0>
__global__ __launch_bounds__(kBlock) void upconv_cl_bwd_kernel(const void* __restrict__ g_out, const int* __restrict__ y_lo,
                                                               const int* __restrict__ y_hi, const int* __restrict__ x_lo,
                                                               const int* __restrict__ x_hi, void* __restrict__ g_P,
                                                               int NB, int C, int h, int w, int H, int W)
{
    typedef typename std::conditional<VEC == 4, f4, float>::type vec_t;
    const unsigned CV = (unsigned)C / VEC;
    const unsigned idx = xcd_remap(blockIdx.x, gridDim.x) * kBlock + threadIdx.x;   // (source pixel, channel vector) of one image
    if (idx >= (unsigned)(h * w) * CV) return;
    const unsigned sp = idx / CV, cv = idx - sp * CV;
    const unsigned iy = sp / (unsigned)w, ix = sp - iy * (unsigned)w;
    const int ylo = y_lo[iy], yhi = y_hi[iy], xlo = x_lo[ix], xhi = x_hi[ix];
    const int ry = yhi - ylo, rx = xhi - xlo;
    const unsigned KKC = (unsigned)(K * K * C);
    vec_t zero;
    if constexpr (VEC == 4) zero = (f4){0.f, 0.f, 0.f, 0.f}; else zero = 0.f;
    for (int img = blockIdx.y; img < NB; img += gridDim.y) {
        const long long gbase = (long long)img * (H * W) * C + cv * VEC;     // element offset of this lane's channels in g_out
        const long long gpo = ((long long)img * (h * w) + sp) * KKC + cv * VEC;   // element offset of this lane's channels in g_P
#if SS_CL_BWD_ROWSCAN
        if (ry >= 1 && ry <= 3 && rx <= 3) {
            // one pass over the ry + K - 1 window rows, top to bottom: each row is loaded ONCE ((K+2) vectors), reduced to its K
            // horizontal-tap sums cs[kx], and combined with the sums of the previous one / two rows (kept in registers) into the
            // vertical tap that this row completes — (ry+4)(rx+4) = 36..49 loads per lane instead of 3*7*K = 105, same summation
            // order (rows oldest first, columns left to right inside a row) => bit-identical to the per-tap form.
            vec_t p1[K], p2[K];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) { p1[kx] = zero; p2[kx] = zero; }
#pragma unroll 1
            for (int j = 0; j < ry + K - 1; ++j) {
                const int y = ylo - (K - 1) + j;
                vec_t cs[K];
#pragma unroll
                for (int kx = 0; kx < K; ++kx) cs[kx] = zero;
                if (y >= 0 && y < H) {
                    vec_t row[K + 2];
#pragma unroll
                    for (int c = 0; c < K + 2; ++c) {
                        const int x = xlo - (K - 1) + c;
                        const bool ok = x >= 0 && x < W && c < K - 1 + rx;
                        row[c] = ok ? load_act<ACT, VEC>(g_out, gbase + (long long)(y * W + x) * C) : zero;
                    }
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        const int b0 = K - 1 - kx;
                        if (rx > 0) cs[kx] += row[b0];
                        if (rx > 1) cs[kx] += row[b0 + 1];
                        if (rx > 2) cs[kx] += row[b0 + 2];
                    }
                }
                const int ky = (K - 2) + ry - j;                      // the vertical tap whose last row this is
                if (j >= ry - 1 && ky >= 0) {
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        vec_t acc = zero;
                        if (ry > 2) acc += p2[kx];
                        if (ry > 1) acc += p1[kx];
                        acc += cs[kx];
                        store_act<GP, VEC>(g_P, gpo + (ky * K + kx) * C, acc);
                    }
                }
#pragma unroll
                for (int kx = 0; kx < K; ++kx) { p2[kx] = p1[kx]; p1[kx] = cs[kx]; }
            }
#else
        if (ry <= 3 && rx <= 3) {
            // per vertical tap: up to 3 rows of a (K+2)-wide window, row sums shared between the K horizontal taps
            // (rows re-read per ky hit L1; the ky loop is kept rolled so the kernel stays below ~100 VGPRs instead of 255)
#pragma unroll 1
            for (int ky = 0; ky < K; ++ky) {
                vec_t acc[K];
#pragma unroll
                for (int kx = 0; kx < K; ++kx) acc[kx] = zero;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const int y = ylo - ky + a;
                    if (a < ry && y >= 0 && y < H) {
                        vec_t row[K + 2];
#pragma unroll
                        for (int c = 0; c < K + 2; ++c) {
                            const int x = xlo - (K - 1) + c;
                            const bool ok = x >= 0 && x < W && c < K - 1 + rx;
                            row[c] = ok ? load_act<ACT, VEC>(g_out, gbase + (long long)(y * W + x) * C) : zero;
                        }
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) {
                            const int b0 = K - 1 - kx;
                            vec_t cs = zero;
                            if (rx > 0) cs += row[b0];
                            if (rx > 1) cs += row[b0 + 1];
                            if (rx > 2) cs += row[b0 + 2];
                            acc[kx] += cs;
                        }
                    }
                }
#pragma unroll
                for (int kx = 0; kx < K; ++kx) store_act<GP, VEC>(g_P, gpo + (ky * K + kx) * C, acc[kx]);
            }
#endif
        } else {
            for (int ky = 0; ky < K; ++ky) {
                const int y0 = max(ylo - ky, 0), y1 = min(yhi - ky, H);
                for (int kx = 0; kx < K; ++kx) {
                    const int x0 = max(xlo - kx, 0), x1 = min(xhi - kx, W);
                    vec_t acc = zero;
                    for (int y = y0; y < y1; ++y) {
                        vec_t cs = zero;
                        for (int x = x0; x < x1; ++x) cs += load_act<ACT, VEC>(g_out, gbase + (long long)(y * W + x) * C);
                        acc += cs;
                    }
                    store_act<GP, VEC>(g_P, gpo + (ky * K + kx) * C, acc);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// operand preparation for the exact bf16x3 GEMM form of the encoder / bottleneck convs on spike inputs (fused.py::_SpikeConvCL)
// ---------------------------------------------------------------------------------------------------
// im2col of an NHWC fp32 array into a bf16 patch matrix A[(nb, oy, ox)][(ky, kx, c)], zero padding, stride s.  A lane converts 8
// consecutive channels of one (row, tap): two 16-B loads, one 16-B store; the k*k-fold re-read of x is served by L2.
__global__ __launch_bounds__(kBlock) void im2col_cl_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ A,
                                                                int h, int w, int C, int k, int stride, int pad, int ho, int wo)
{
    // blockIdx.x = patch row (nb, oy, ox): its decomposition is wave-uniform (scalar ALU); blockIdx.y * 256 + lane = (tap, 8-channel group)
    const unsigned row = blockIdx.x;
    const unsigned C8 = (unsigned)C / 8;
    const unsigned r = blockIdx.y * kBlock + threadIdx.x;
    if (r >= (unsigned)(k * k) * C8) return;
    const unsigned ox = row % (unsigned)wo, t = row / (unsigned)wo;
    const unsigned oy = t % (unsigned)ho, nb = t / (unsigned)ho;
    const unsigned tap = r / C8, c8 = r - tap * C8;
    const unsigned ky = tap / (unsigned)k, kx = tap - ky * (unsigned)k;
    const int iy = (int)(oy * stride + ky) - pad, ix = (int)(ox * stride + kx) - pad;
    u16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
    if (iy >= 0 && iy < h && ix >= 0 && ix < w) {
        const float* src = x + (((long long)nb * h + iy) * w + ix) * C + c8 * 8;
        const f4 a = *reinterpret_cast<const f4*>(src), b = *reinterpret_cast<const f4*>(src + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = narrow<SS_DT_BF16>(a[e]); o[4 + e] = narrow<SS_DT_BF16>(b[e]); }
    }
    *reinterpret_cast<u16x8*>(A + ((long long)row * (k * k) + tap) * C + c8 * 8) = o;
}

// g fp32 [M][N] -> g3 bf16 [M][3N] = [hi | mid | lo] with hi = bf16(g), mid = bf16(g - hi), lo = bf16(g - hi - mid): the three terms
// sum to g exactly unless g needs more than 24 significant bits below its leading one (never for fp32).
__global__ __launch_bounds__(kBlock) void split3_bf16_kernel(const float* __restrict__ g, unsigned short* __restrict__ g3,
                                                             long long M, int N)
{
    const int N4 = N / 4;
    const long long total = M * N4;
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const long long m = i / N4;
        const int n4 = (int)(i - m * N4);
        const f4 v = *reinterpret_cast<const f4*>(g + m * N + n4 * 4);
        u16x4 hi, mid, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned short bh = narrow<SS_DT_BF16>(v[e]);
            const float r1 = v[e] - widen<SS_DT_BF16>(bh);
            const unsigned short bm = narrow<SS_DT_BF16>(r1);
            const float r2 = r1 - widen<SS_DT_BF16>(bm);
            hi[e] = bh; mid[e] = bm; lo[e] = narrow<SS_DT_BF16>(r2);
        }
        unsigned short* dst = g3 + m * 3 * N + n4 * 4;
        *reinterpret_cast<u16x4*>(dst) = hi;
        *reinterpret_cast<u16x4*>(dst + N) = mid;
        *reinterpret_cast<u16x4*>(dst + 2 * N) = lo;
    }
}

// code (0..3) of a 2-bit packed spike -> bf16 bit pattern of the same small integer: 0x0000, 0x3F80, 0x4000, 0x4040
#define SS_CODE_LUT 0x404040003F800000ull
__device__ __forceinline__ unsigned short code_to_bf16(unsigned c) { return (unsigned short)((SS_CODE_LUT >> (16 * c)) & 0xFFFFu); }

// ---------------------------------------------------------------------------------------------------
// Fused projection + gather of NNConvUpsampling (forward) on the bf16 matrix cores — the per-tap projection tensor P never reaches HBM.
// ---------------------------------------------------------------------------------------------------
// Reference: /root/reference/network/blocks.py:110-132 (UpsamplingNearest2d(size = up + k - 1) -> Conv2d(k = 5, stride 1, pad 0)),
// call sites SNN_models.py:110-129 (deconv4..1).  Math as in ss_upconv_cl_fwd_f32: out[y][x][co] = sum_{ky,kx} P[src_y[y+ky]][src_x[x+kx]][ky,kx][co],
// P[s][tap][co] = sum_ci x[s][ci] W[co][ci][tap].  One workgroup (4 wavefronts) owns a 16 x 16 tile of OUTPUT pixels of one frame:
//   1. its source window (<= 128 low-resolution pixels, all C_in channels) is loaded straight into MFMA A fragments (spikes are exact in bf16);
//   2. per pass over 8 output channels: P_tile[128 sources][25 taps x 8 channels] = A (bf16) x W (fp32 split EXACTLY into 3 bf16 terms,
//      fragment-ordered by ss_upconv_fused_prep_w, streamed from L2) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: every product is
//      exact, so P has fp32-GEMM accuracy (sum of exact products, fp32 accumulate) at the bf16 MFMA rate;
//   3. the accumulators go to LDS (100 KiB), every lane gathers its pixel's 25 taps x 8 channels from LDS in the tap order of the unfused
//      kernel and stores 32 B of the NHWC output.
// HBM traffic: x once (+ halo), weights from L2, out once — instead of writing and re-reading P (25 x C_out floats per source pixel:
// 5.76 GB per step for deconv1 at config 3).  MFMA work: ~1.6x the minimal projection (source-window halo + tile padding), still
// ~2.5x less than the direct 25-tap convolution.
constexpr int kFusT = 16;                     // output tile edge
constexpr int kFusS = 128;                    // max source pixels per tile (4 M-tiles of 32)
constexpr int kFusCC = 8;                     // output channels per pass
constexpr int kFusNP = 25 * kFusCC;           // P columns per pass (200)
constexpr int kFusNT = (kFusNP + 31) / 32;    // N tiles of 32 (7; the last one is a quarter full)
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// W [C_out][C_in][5][5] fp32 -> MFMA B fragments: Wf[pass][kstep][ntile][lane][8] bf16, pass = 8 output channels, kstep over K = 3 * C_in
// (split-major: all of hi, then mid, then lo), element e of lane l = B[k = 16 kstep + 8 (l >> 5) + e][n = 32 ntile + (l & 31)], n = tap * 8 + c.
__global__ __launch_bounds__(kBlock) void upconv_fused_prep_w_kernel(const float* __restrict__ Wt, unsigned short* __restrict__ Wf, int Cin, int Cout)
{
    const int ksteps = 3 * Cin / 16, passes = Cout / kFusCC;
    const long long total = (long long)passes * ksteps * kFusNT * 64;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int lane = (int)(i & 63);
        long long r = i >> 6;
        const int nt = (int)(r % kFusNT); r /= kFusNT;
        const int ks = (int)(r % ksteps); const int pass = (int)(r / ksteps);
        const int n = 32 * nt + (lane & 31);
        u16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
        if (n < kFusNP) {
            const int tap = n / kFusCC, co = pass * kFusCC + (n % kFusCC);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 16 * ks + 8 * (lane >> 5) + e;
                const int split = k / Cin, ci = k - split * Cin;
                const float wv = Wt[((long long)co * Cin + ci) * 25 + tap];
                const unsigned short bh = narrow<SS_DT_BF16>(wv);
                const float r1 = wv - widen<SS_DT_BF16>(bh);
                const unsigned short bm = narrow<SS_DT_BF16>(r1);
                const float r2 = r1 - widen<SS_DT_BF16>(bm);
                o[e] = split == 0 ? bh : (split == 1 ? bm : narrow<SS_DT_BF16>(r2));
            }
        }
        *reinterpret_cast<u16x8*>(Wf + i * 8) = o;
    }
}

constexpr int kFusPS = 204;                   // LDS row stride of the P tile in floats: conflict-free 16-B writes (8-lane groups) and reads

// Work split inside the workgroup: wavefront w owns the P COLUMN tiles {w, w + 4} (weights stationary in its registers for a
// (pass, split) chunk, prefetched one chunk ahead) and runs them against all four 32-pixel source tiles; the product is taken as
// P^T = W^T x^T (weights as the MFMA A operand) so that a lane ends up with 4 CONSECUTIVE columns of one source pixel -> 16-B LDS stores.
template <int CIN, int COUT, bool PACKED>
__global__ __launch_bounds__(kBlock) void upconv_fused_fwd_kernel(const void* __restrict__ xin, const unsigned short* __restrict__ Wf,
                                                                  const int* __restrict__ src_y, const int* __restrict__ src_x,
                                                                  float* __restrict__ out, int h, int w, int H, int W, int tiles_x, int tiles_y)
{
    constexpr int KC = CIN / 16;                      // ci chunks of 16 = MFMA k-steps per split
    constexpr int KSTEPS = 3 * KC;
    constexpr int PASSES = COUT / kFusCC;
    constexpr int MT = kFusS / 32;                    // 4 source tiles
    __shared__ float Pt[kFusS * kFusPS];             // 104 448 B
    const int tile = blockIdx.x % (tiles_x * tiles_y), img = blockIdx.x / (tiles_x * tiles_y);
    const int y0 = (tile / tiles_x) * kFusT, x0 = (tile % tiles_x) * kFusT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sy0 = src_y[y0], sx0 = src_x[x0];
    const int SH = src_y[min(y0 + kFusT - 1, H - 1) + 4] - sy0 + 1, SW = src_x[min(x0 + kFusT - 1, W - 1) + 4] - sx0 + 1;
    // ---- 1. x fragments (MFMA B operand: column = source pixel 32 m + (lane & 31), k = channels 16 j + 8 (lane >> 5) .. + 7), all 4 source tiles
    s16x8 xf[MT][KC];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int srow = 32 * m + (lane & 31);
        const bool valid = srow < SH * SW;
        const int ly = valid ? srow / SW : 0, lx = valid ? srow - ly * SW : 0;
        const long long pix = ((long long)img * h + (sy0 + ly)) * w + (sx0 + lx);
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            s16x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
            if (valid) {
                const long long e = pix * CIN + 16 * j + 8 * (lane >> 5);
                if constexpr (PACKED) {
                    const unsigned bits = (static_cast<const unsigned*>(xin)[e >> 4] >> (2 * (int)(e & 15))) & 0xFFFFu;
#pragma unroll
                    for (int q = 0; q < 8; ++q) a[q] = (short)code_to_bf16((bits >> (2 * q)) & 3u);
                } else {
                    const float* xp = static_cast<const float*>(xin) + e;
                    const f4 lo = *reinterpret_cast<const f4*>(xp), hi = *reinterpret_cast<const f4*>(xp + 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { a[q] = (short)(__float_as_uint(lo[q]) >> 16); a[4 + q] = (short)(__float_as_uint(hi[q]) >> 16); }   // spikes: exact
                }
            }
            xf[m][j] = a;
        }
    }
    const int py = threadIdx.x >> 4, px = threadIdx.x & 15;
    const int y = y0 + py, x = x0 + px;
    const bool inside = y < H && x < W;
    int soff[25];                                                       // LDS offset of every tap's (source pixel, first channel of the tap)
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
        const int ly = inside ? src_y[y + ky] - sy0 : 0;
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) soff[ky * 5 + kx] = (ly * SW + (inside ? src_x[x + kx] - sx0 : 0)) * kFusPS + (ky * 5 + kx) * kFusCC;
    }
    const bool two = wave + 4 < kFusNT;                                  // wave 3 owns one column tile only (7 tiles)
    const s16x8* wbase = reinterpret_cast<const s16x8*>(Wf) + lane;
    // weight fragments of one (pass, split) chunk: [column tile 0 / 1][k-step]
    auto load_chunk = [&](s16x8 (&dst)[2][KC], int pass, int split) {
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const long long ks = (long long)pass * KSTEPS + split * KC + j;
            dst[0][j] = wbase[(ks * kFusNT + wave) * 64];
            dst[1][j] = two ? wbase[(ks * kFusNT + wave + 4) * 64] : (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    s16x8 wA[2][KC], wB[2][KC];
    load_chunk(wA, 0, 0);
    for (int pass = 0; pass < PASSES; ++pass) {
        // ---- 2. P^T tiles: rows = columns n of P (this wave's tiles), columns = source pixels
        f32x16 acc[2][MT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][m][r] = 0.f;
        auto mma = [&](const s16x8 (&wf)[2][KC]) {
#pragma unroll
            for (int j = 0; j < KC; ++j)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][j], xf[m][j], acc[0][m], 0, 0, 0);
                    if (two) acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][j], xf[m][j], acc[1][m], 0, 0, 0);
                }
        };
        load_chunk(wB, pass, 1);
        mma(wA);                                                         // split hi
        load_chunk(wA, pass, 2);
        mma(wB);                                                         // split mid
        if (pass + 1 < PASSES) load_chunk(wB, pass + 1, 0);              // lands during the LDS phases below
        mma(wA);                                                         // split lo
        if (pass) __syncthreads();                                       // the previous pass's gather is done with Pt
        // ---- 3a. accumulators -> LDS.  C layout of P^T: column (source) = lane & 31, row (n) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5):
        //          registers 4 q .. 4 q + 3 are 4 consecutive n of one source pixel
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t == 0 || two) {
                const int nt = wave + 4 * t;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = 32 * nt + 8 * q + 4 * (lane >> 5);
                        if (n < kFusNP)
                            *reinterpret_cast<f4*>(&Pt[(32 * m + (lane & 31)) * kFusPS + n]) =
                                (f4){acc[t][m][4 * q], acc[t][m][4 * q + 1], acc[t][m][4 * q + 2], acc[t][m][4 * q + 3]};
                    }
            }
        }
        __syncthreads();
        // ---- 3b. gather: taps in (ky, kx) order — the unfused kernel's summation order
        if (inside) {
            f4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tp = 0; tp < 25; ++tp) {
                s0 += *reinterpret_cast<const f4*>(&Pt[soff[tp]]);
                s1 += *reinterpret_cast<const f4*>(&Pt[soff[tp] + 4]);
            }
            float* op = out + (((long long)img * H + y) * W + x) * COUT + pass * kFusCC;
            *reinterpret_cast<f4*>(op) = s0;
            *reinterpret_cast<f4*>(op + 4) = s1;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < KC; ++j) wA[t][j] = wB[t][j];
    }
}

// ---------------------------------------------------------------------------------------------------
// Fused projection + gather, second form: wavefront-specialised persistent workgroups (same value as upconv_fused_fwd_kernel, bit for bit)
// ---------------------------------------------------------------------------------------------------
// What bounded the first form (profiles/r02/fused_upconv_ab.log: 22 us per tile against 5 us of MFMA time): (1) every wavefront
// fetched the whole source window itself, 16 B out of each 128-B line (4x redundant, ~8k L1 line look-ups per tile); (2) MFMA, LDS store
// and LDS gather phases ran one after the other behind barriers with ONE workgroup per CU (104 KiB LDS, 342 - 508 registers);
// (3) one workgroup per tile: dispatch, table look-ups and the first loads were exposed 30 000 times per launch.
// This form:
//   * persistent workgroups of 8 wavefronts, two per SIMD: 4 PRODUCERS (MFMA) and 4 CONSUMERS (gather); a workgroup walks a contiguous
//     band of tiles of one XCD (neighbouring tiles share their window halo through that XCD's L2);
//   * the source window goes ONCE, coalesced, from HBM to LDS as bf16 (rows padded by 16 B: conflict-free fragment reads);
//   * P is produced in PASSES of four 32-column tiles (column = tap * C_out + co: no padding columns), one tile per producer wavefront,
//     weights streamed from L2 straight into that wavefront's registers (each weight fragment is fetched once per tile per CU, three
//     (pass, split) chunks in flight), product taken transposed (weights as the A operand) so a lane holds 4 consecutive P columns of
//     one source pixel -> 16-B LDS stores into a double-buffered, swizzled pass buffer;
//   * while the producers compute pass p + 1 the consumers gather pass p from the other buffer into per-pixel accumulators held in
//     registers (taps in (ky, kx) order — the summation order of the unfused gather kernel and of the first form), write the tile's
//     output once, and fetch the NEXT tile's window (loads issued before the last gather, committed to LDS after it).
// LDS: 2 x 122 x 528 B pass buffers + 122 x (2 C_in + 16) B window (+ 9 KiB output transposition scratch for C_in 64) = 152.0 / 158.2 KiB.
#ifndef SS_F2_PRIO
#define SS_F2_PRIO 2                          // wave priority: 0 none, 1 producers 3, 2 consumers 1, 3 consumers 3
#endif
#ifndef SS_F2_ABLATE
#define SS_F2_ABLATE 0                        // development aid (make variant DEFS=-DSS_F2_ABLATE=mask; tools/bench_fused_upconv.py SS_LIB=...): skip
#endif                                        // 1 gather, 2 pass-buffer stores, 4 weight stream, 8 MFMAs, 16 next-window fetch — wrong results, timing only
constexpr int kF2Threads = 512;
#ifndef SS_F2_TRACE
#define SS_F2_TRACE 0                         // development aid: workgroup 0 records s_memtime stamps of its first steps (ss_debug_f2_trace)
#endif
#if SS_F2_TRACE
__device__ unsigned long long f2_trace[2][64][4];
#define F2_STAMP(role, slot) do { if (blockIdx.x == 0 && lane == 0 && cw == 0 && tstep < 64) f2_trace[role][tstep][slot] = clock64(); } while (0)
#else
#define F2_STAMP(role, slot) do { } while (0)
#endif
#ifndef SS_F2_BAR
#define SS_F2_BAR 1
#endif
// Workgroup barrier of the wavefront-specialised kernel: orders LDS traffic only (lgkmcnt), so the weight / window loads a wavefront has
// in flight (vmcnt) stay in flight across it — __syncthreads() would drain them at every pass
__device__ __forceinline__ void f2_barrier()
{
#if SS_F2_BAR
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}
constexpr int kF2Rows = 122;                  // source pixels of a tile held on chip (max_window <= 122, else the first form)
constexpr int kF2PRowB = 528;                 // bytes of one source pixel's row in a pass buffer: 4 column tiles x 32 channels fp32 + 16 B of
                                              // padding (consecutive rows start 4 banks apart: conflict-free 16-B stores and gathers, immediate offsets)

// W [C_out][C_in][5][5] fp32 -> Wf2[column tile nt][split][k-step j][lane][8] bf16; column n = 32 nt + (lane & 31) = tap * C_out + co,
// element e = split term of W[co][ci = 16 j + 8 (lane >> 5) + e][tap]
__global__ __launch_bounds__(kBlock) void upconv_fused2_prep_w_kernel(const float* __restrict__ Wt, unsigned short* __restrict__ Wf, int Cin, int Cout,
                                                                      int nsplit)
{
    const int KC = Cin / 16, NTL = 25 * Cout / 32;
    const long long total = (long long)NTL * nsplit * KC * 64;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int lane = (int)(i & 63);
        long long r = i >> 6;
        const int j = (int)(r % KC); r /= KC;
        const int split = (int)(r % nsplit); const int nt = (int)(r / nsplit);
        const int n = 32 * nt + (lane & 31);
        const int tap = n / Cout, co = n - tap * Cout;
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = 16 * j + 8 * (lane >> 5) + e;
            const float wv = Wt[((long long)co * Cin + ci) * 25 + tap];
            const unsigned short bh = narrow<SS_DT_BF16>(wv);
            const float r1 = wv - widen<SS_DT_BF16>(bh);
            const unsigned short bm = narrow<SS_DT_BF16>(r1);
            const float r2 = r1 - widen<SS_DT_BF16>(bm);
            o[e] = split == 0 ? bh : (split == 1 ? bm : narrow<SS_DT_BF16>(r2));
        }
        *reinterpret_cast<u16x8*>(Wf + i * 8) = o;
    }
}

// XIN: input spikes as 0 fp32, kF2Packed 2-bit packed, SS_DT_F16 / SS_DT_BF16 16-bit activations (all exact in bf16);  ODT: output fp32 (0) or
// 16-bit activations (the fp32 sums narrowed on store: decoder stages under 16-bit autocast);  NSPLIT: 3 = exact fp32 weights (hi, mid, lo),
// 1 = weights rounded once to bf16 (what bf16 autocast does to every synapse; Wf then holds the hi terms only)
constexpr int kF2Packed = 3;
template <int CIN, int COUT, int XIN, int ODT, int NSPLIT>
__global__ __launch_bounds__(kF2Threads) void upconv_fused2_fwd_kernel(const void* __restrict__ xin, const unsigned short* __restrict__ Wf,
                                                                      const int* __restrict__ src_y, const int* __restrict__ src_x,
                                                                      void* __restrict__ outv, int h, int w, int H, int W,
                                                                      int tiles_x, int tiles_y, int n_tiles)
{
    constexpr bool PACKED = XIN == kF2Packed;
    constexpr bool X16 = XIN == SS_DT_F16 || XIN == SS_DT_BF16;
    static_assert(NSPLIT == 3 || NSPLIT == 1, "three exact terms or one rounded term");
    constexpr int KC = CIN / 16;                      // MFMA k-steps per split
    constexpr int NPT = COUT / 32;                    // column tiles per tap
    constexpr int NTL = 25 * NPT;                     // column tiles in all (25 / 50: no padding)
    constexpr int NPASS = (NTL + 3) / 4;              // 7 / 13
    constexpr int TPP = 4 / NPT;                      // taps per pass (4 / 2)
    constexpr int XROWB = CIN * 2 + 16;               // bytes of one source pixel in the bf16 window + 16 B of padding (rows start 4 banks
                                                      // apart modulo 64: conflict-free fragment reads at immediate offsets)
    constexpr int XCH = CIN / 8;                      // 16-B chunks per source pixel
    constexpr int XU = (kF2Rows * XCH + 255) / 256;   // window chunks per consumer lane (4 / 8)
    constexpr bool XREG = CIN <= 64;                  // producers keep the window fragments in registers for the whole tile
    // CONT: the window buffer is free as soon as the producers hold their fragments, so the next tile's window is committed in the
    // middle of this tile and the pipeline runs through tile boundaries (pass buffer = global step parity).  Otherwise the producers
    // read the window in every pass and idle for one step per tile while the consumers commit the next one.
    constexpr bool CONT = XREG;
    constexpr int WC = NPASS - 2 < 3 ? NPASS - 2 : 3; // CONT: pass in which the next window is committed (issued in pass 0)
    const int cw_ = (threadIdx.x >> 6) & 3;
    // output transposition scratch: 16 pixels x (C_out floats + 16 B) per consumer wavefront.  CONT: its own region; otherwise the window
    // region, which is free between the producers' last pass of a tile and the commit of the next window
    constexpr int OROWB = COUT * 4 + 16;              // (fp32 in the scratch; narrowed when it leaves)
    constexpr int OSCR = 16 * OROWB;
    static_assert(CONT || 4 * OSCR <= kF2Rows * XROWB, "transposition scratch must fit the window region");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * kF2Rows * kF2PRowB + kF2Rows * XROWB + (CONT ? 4 * OSCR : 0)];
    unsigned char* const Xs = smem + 2 * kF2Rows * kF2PRowB;
    unsigned char* const Os = (CONT ? Xs + kF2Rows * XROWB : Xs) + cw_ * OSCR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool producer = wave < 4;
    const int cw = wave & 3;
    const int tiles_img = tiles_x * tiles_y;
    // XCD-aware walk: workgroup b runs on XCD b % 8; XCD k owns the contiguous band of tiles [k * per_xcd, (k + 1) * per_xcd)
    const int per_xcd = (n_tiles + 7) / 8;
    const int band0 = (int)(blockIdx.x & 7) * per_xcd, slot0 = (int)(blockIdx.x >> 3), slots = (int)(gridDim.x >> 3);
    const int band_end = min(band0 + per_xcd, n_tiles);
    [[maybe_unused]] int tstep = 0;


    if (producer) {
        if (SS_F2_PRIO == 1) __builtin_amdgcn_s_setprio(3);
        // ---- weight stream: chunk (pass, split) of this wavefront's column tile nt = 4 pass + cw sits in buffer `split`; NSPLIT in flight
        //      (NSPLIT == 1: the next pass's chunk is loaded into wnx during this pass and copied over)
        const s16x8* const wbase = reinterpret_cast<const s16x8*>(Wf) + lane;
        s16x8 wq[3][KC];
        [[maybe_unused]] s16x8 wnx[KC];
        auto load_chunk = [&](s16x8 (&dst)[KC], int pass, int split) {
            const long long c0 = ((long long)(4 * pass + cw) * NSPLIT + split) * KC;
#pragma unroll
            for (int j = 0; j < KC; ++j) dst[j] = wbase[(c0 + j) * 64];
        };
        if constexpr (NSPLIT == 3) { load_chunk(wq[0], 0, 0); load_chunk(wq[1], 0, 1); load_chunk(wq[2], 0, 2); }
        else load_chunk(wq[2], 0, 0);                                     // the single term plays the role of the last split
        f2_barrier();                                                     // first window is in LDS
        int gstep = 0;
        for (int tile = band0 + slot0; tile < band_end; tile += slots) {
            s16x8 xf[XREG ? 4 : 1][XREG ? KC : 1];
            auto xfrag = [&](int m, int j) -> s16x8 {
                const int row = min(32 * m + (lane & 31), kF2Rows - 1);
                return *reinterpret_cast<const s16x8*>(Xs + row * XROWB + ((2 * j + (lane >> 5)) << 4));
            };
            if constexpr (XREG) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int j = 0; j < KC; ++j) xf[m][j] = xfrag(m, j);
            }
#pragma unroll 1
            for (int pass = 0; pass < NPASS; ++pass, ++gstep) {
                F2_STAMP(0, 0);
                if (4 * pass + cw < NTL) {
                    const int npass = (4 * (pass + 1) + cw < NTL) ? pass + 1 : 0;     // this wavefront's next pass (wraps into the next tile)
                    unsigned char* const Pp = smem + ((CONT ? gstep : pass) & 1) * (kF2Rows * kF2PRowB);
                    f32x16 acc[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
                    // The MFMA builtin has no side effects, so hipcc re-orders these freely (it had moved every pass-buffer store behind the
                    // last MFMA and the weight loads of all three splits to the end of the pass); sched_barrier(0) pins the phases:
                    //   hi | load hi' | mid | load mid' | lo of source tiles 0, 1 | lo of tiles 2, 3 interleaved with the stores of tiles 0, 1 |
                    //   stores of tiles 2, 3 | load lo'            (x' = the same split of this wavefront's next pass)
                    // Per accumulator the order stays hi, mid, lo with k ascending (bit-identical to the first kernel form).
                    auto xop = [&](int m, int jj) -> s16x8 { if constexpr (XREG) return xf[m][jj]; else return xfrag(m, jj); };
                    auto mma = [&](f32x16& a, const s16x8& wv, const s16x8& xv, int m, int jj) {
                        if (!(SS_F2_ABLATE & 8)) a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, xv, a, 0, 0, 0);
                        else a[jj] += (float)wv[m] + (float)xv[0];
                    };
                    // splits hi, mid: k-step outer, the four source tiles inner (4 independent accumulators)
#pragma unroll
                    for (int s = 0; s < NSPLIT - 1; ++s) {
                        if constexpr (XREG) {
#pragma unroll
                            for (int jj = 0; jj < KC; ++jj)
#pragma unroll
                                for (int m = 0; m < 4; ++m) mma(acc[m], wq[s][jj], xf[m][jj], m, jj);
                        } else {
                            // the window fragments are re-read from LDS for every split (ds_read_b128, two k-steps ahead of their MFMAs); the
                            // clobber keeps the compiler from merging the three reads into 128 live registers
                            asm volatile("" ::: "memory");
                            s16x8 xa[4], xb[4], xc[4];
#pragma unroll
                            for (int m = 0; m < 4; ++m) { xa[m] = xfrag(m, 0); xb[m] = xfrag(m, 1); }
#pragma unroll
                            for (int jj = 0; jj < KC; ++jj) {
                                if (jj + 2 < KC) {
#pragma unroll
                                    for (int m = 0; m < 4; ++m) xc[m] = xfrag(m, jj + 2);
                                }
#pragma unroll
                                for (int m = 0; m < 4; ++m) mma(acc[m], wq[s][jj], xa[m], m, jj);
#pragma unroll
                                for (int m = 0; m < 4; ++m) { xa[m] = xb[m]; xb[m] = xc[m]; }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (!(SS_F2_ABLATE & 4)) load_chunk(wq[s], npass, s);     // lands two chunks of MFMA work later
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // C layout of P^T: column (source) = lane & 31, row (n) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
                    if constexpr (!XREG) asm volatile("" ::: "memory");
                    auto store_q = [&](int m, int q) {
                        const int row = 32 * m + (lane & 31);
                        if (row < kF2Rows && (!(SS_F2_ABLATE & 2) || acc[m][0] == 12345.f))
                            *reinterpret_cast<f4*>(Pp + row * kF2PRowB + ((cw * 8 + 2 * q + (lane >> 5)) << 4)) =
                                (f4){acc[m][4 * q], acc[m][4 * q + 1], acc[m][4 * q + 2], acc[m][4 * q + 3]};
                    };
                    if constexpr (NSPLIT == 1) { if (!(SS_F2_ABLATE & 4)) load_chunk(wnx, npass, 0); }
                    // split lo, source tiles 0 and 1 (two interleaved accumulator chains)
#pragma unroll
                    for (int jj = 0; jj < KC; ++jj) { mma(acc[0], wq[2][jj], xop(0, jj), 0, jj); mma(acc[1], wq[2][jj], xop(1, jj), 1, jj); }
                    __builtin_amdgcn_sched_barrier(0);
                    // split lo, tiles 2 and 3; the finished tiles 0 and 1 go to the pass buffer under these MFMAs
#pragma unroll
                    for (int jj = 0; jj < KC; ++jj) {
                        mma(acc[2], wq[2][jj], xop(2, jj), 2, jj);
                        if (2 * jj < 8) store_q(jj * 2 / 4, (jj * 2) % 4);
                        if (2 * jj + 1 < 8) store_q((jj * 2 + 1) / 4, (jj * 2 + 1) % 4);
                        mma(acc[3], wq[2][jj], xop(3, jj), 3, jj);
                    }
                    static_assert(KC >= 4, "the store interleave above expects >= 4 k-steps");
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (NSPLIT == 3) { if (!(SS_F2_ABLATE & 4)) load_chunk(wq[2], npass, 2); }
                    else {
#pragma unroll
                        for (int jj = 0; jj < KC; ++jj) wq[2][jj] = wnx[jj];
                    }
#if SS_F2_TRACE
                    if (acc[0][0] == 12345.f && acc[1][1] == 1.f && acc[2][2] == 2.f && acc[3][3] == 3.f) f2_trace[0][63][3] = 1;   // waits for the MFMAs
                    F2_STAMP(0, 1);
#endif
#pragma unroll
                    for (int q = 0; q < 4; ++q) { store_q(2, q); store_q(3, q); }
                }
#if SS_F2_TRACE
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                F2_STAMP(0, 2);
#endif
                f2_barrier();
                F2_STAMP(0, 3);
                ++tstep;
            }
            if constexpr (!CONT) { f2_barrier(); f2_barrier(); ++tstep; }  // consumers: last gather + output | next window
        }
        return;
    }

    // ------------------------------------------------------------- consumers
    if (SS_F2_PRIO == 2) __builtin_amdgcn_s_setprio(1);
    if (SS_F2_PRIO == 3) __builtin_amdgcn_s_setprio(3);
    const int ct = cw * 64 + lane;                                        // 0..255: pixel (ct >> 4, ct & 15) of the tile
    const int py = ct >> 4, px = ct & 15;
    f4 xr[XU][(PACKED || X16) ? 1 : 2];
    unsigned xrp[XU];
    struct Geo { int img, y0, x0, sy0, sx0, SH, SW; };
    auto geom = [&](int tile) {
        Geo g;
        const int t2 = tile % tiles_img;
        g.img = tile / tiles_img;
        g.y0 = (t2 / tiles_x) * kFusT; g.x0 = (t2 % tiles_x) * kFusT;
        g.sy0 = src_y[g.y0]; g.sx0 = src_x[g.x0];
        g.SH = src_y[min(g.y0 + kFusT - 1, H - 1) + 4] - g.sy0 + 1;
        g.SW = src_x[min(g.x0 + kFusT - 1, W - 1) + 4] - g.sx0 + 1;
        return g;
    };
    auto window_issue = [&](const Geo& g) {                                // HBM -> registers
        const int total = g.SH * g.SW * XCH;
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int c = ct + 256 * u;
            if (c < total) {
                const int row = c / XCH, ch = c - row * XCH;
                const int ly = row / g.SW, lx = row - ly * g.SW;
                const long long e = ((((long long)g.img * h + (g.sy0 + ly)) * w) + (g.sx0 + lx)) * CIN + 8 * ch;
                if constexpr (PACKED) {
                    xrp[u] = (static_cast<const unsigned*>(xin)[e >> 4] >> (2 * (int)(e & 15))) & 0xFFFFu;
                } else if constexpr (X16) {
                    xr[u][0] = *reinterpret_cast<const f4*>(static_cast<const unsigned short*>(xin) + e);     // 8 channels x 16 bit
                } else {
                    const float* xp = static_cast<const float*>(xin) + e;
                    xr[u][0] = *reinterpret_cast<const f4*>(xp);
                    xr[u][1] = *reinterpret_cast<const f4*>(xp + 4);
                }
            }
        }
    };
    auto window_commit = [&](const Geo& g) {                               // registers -> bf16 window in LDS
        const int total = g.SH * g.SW * XCH;
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int c = ct + 256 * u;
            if (c < total) {
                const int row = c / XCH, ch = c - row * XCH;
                s16x8 a;
                if constexpr (PACKED) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) a[q] = (short)code_to_bf16((xrp[u] >> (2 * q)) & 3u);
                } else if constexpr (XIN == SS_DT_BF16) {
                    a = __builtin_bit_cast(s16x8, xr[u][0]);
                } else if constexpr (XIN == SS_DT_F16) {
                    const u16x8 hv = __builtin_bit_cast(u16x8, xr[u][0]);
#pragma unroll
                    for (int q = 0; q < 8; ++q) a[q] = (short)(__float_as_uint(widen<SS_DT_F16>(hv[q])) >> 16);   // small integers: exact
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {                          // spikes: exact in bf16
                        a[q] = (short)(__float_as_uint(xr[u][0][q]) >> 16);
                        a[4 + q] = (short)(__float_as_uint(xr[u][1][q]) >> 16);
                    }
                }
                *reinterpret_cast<s16x8*>(Xs + row * XROWB + (ch << 4)) = a;
            }
        }
    };
    // per-pixel table: source row (ly * SW + lx) of every tap, one byte each, tap 0 in the low byte of tab[0]; consumed from the bottom,
    // TPP bytes per pass
    struct Pix { int y, x; bool inside; int ly[5], lx[5]; };
    auto pix_issue = [&](const Geo& g) {
        Pix p;
        p.y = g.y0 + py; p.x = g.x0 + px;
        p.inside = p.y < H && p.x < W;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            p.ly[k] = p.inside ? src_y[p.y + k] - g.sy0 : 0;
            p.lx[k] = p.inside ? src_x[p.x + k] - g.sx0 : 0;
        }
        return p;
    };
    auto build_tab = [&](const Pix& p, const Geo& g, unsigned (&tab)[7]) {
#pragma unroll
        for (int k = 0; k < 7; ++k) tab[k] = 0;
#pragma unroll
        for (int t = 0; t < 25; ++t) tab[t >> 2] |= (unsigned)(p.ly[t / 5] * g.SW + p.lx[t % 5]) << (8 * (t & 3));
    };
    f4 acc[COUT / 4];
    auto gather = [&](const unsigned char* Pp, int pass, unsigned (&tab)[7]) {
        constexpr int NB_ = COUT <= 32 ? 2 : 1;                            // column tiles per batch of reads issued before their adds (16 / 8 reads)
#pragma unroll
        for (int b = 0; b < 4 / NB_; ++b) {
            f4 v[NB_][8];
#pragma unroll
            for (int ii = 0; ii < NB_; ++ii) {
                const int i = NB_ * b + ii;
                if (4 * pass + i < NTL && !(SS_F2_ABLATE & 1)) {
                    const unsigned row = (tab[0] >> (8 * (i / NPT))) & 0xFFu;
                    const unsigned char* const base = Pp + row * kF2PRowB + i * 128;
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc) v[ii][cc] = *reinterpret_cast<const f4*>(base + 16 * cc);
                }
            }
#pragma unroll
            for (int ii = 0; ii < NB_; ++ii) {
                const int i = NB_ * b + ii;
                if (4 * pass + i < NTL && !(SS_F2_ABLATE & 1)) {
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc) acc[(i % NPT) * 8 + cc] += v[ii][cc];
                }
            }
        }
        // consume TPP bytes of the table
        if constexpr (TPP == 4) {
#pragma unroll
            for (int k = 0; k < 6; ++k) tab[k] = tab[k + 1];
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) tab[k] = (tab[k] >> 16) | (tab[k + 1] << 16);
            tab[6] >>= 16;
        }
    };
    // tile output: a lane holds all C_out channels of ONE pixel; written directly that is 64 partial lines per store instruction.  Instead
    // the wavefront's four pixel rows go through its LDS scratch one after the other and leave as 1-KiB contiguous stores
    // (lane -> 16-B chunk t = lane + 64 u of the row's 16 pixels x C_out floats).  LDS executes a wavefront's operations in order.
    auto store_out = [&](const Geo& g) {
        constexpr int NCH = COUT / 4;                                      // 16-B chunks (4 channels) per pixel in the scratch
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if ((lane >> 4) == r) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) *reinterpret_cast<f4*>(Os + (lane & 15) * OROWB + 16 * c) = acc[c];
            }
            asm volatile("" ::: "memory");        // hipcc 7.2 otherwise sinks the first scratch read INTO the lane-masked store region above
            const int yy = g.y0 + 4 * cw + r;
            const long long orow = (((long long)g.img * H + yy) * W + g.x0) * COUT;
#pragma unroll
            for (int u = 0; u < NCH / 4; ++u) {
                const int t = lane + 64 * u;
                const int pxl = t / NCH, c = t - pxl * NCH;
                const f4 v = *reinterpret_cast<const f4*>(Os + pxl * OROWB + 16 * c);
                if (yy < H && g.x0 + pxl < W) {
                    if constexpr (ODT == 0) {
                        *reinterpret_cast<f4*>(static_cast<float*>(outv) + orow + (long long)pxl * COUT + 4 * c) = v;
                    } else {
                        u16x4 o;
#pragma unroll
                        for (int q = 0; q < 4; ++q) o[q] = narrow<ODT>(v[q]);
                        *reinterpret_cast<u16x4*>(static_cast<unsigned short*>(outv) + orow + (long long)pxl * COUT + 4 * c) = o;
                    }
                }
            }
            asm volatile("" ::: "memory");
        }
    };

    int tile = band0 + slot0;
    Geo g = geom(min(tile, n_tiles - 1));
    if (tile < band_end) { window_issue(g); window_commit(g); }
    f2_barrier();
    if (tile >= band_end) return;

    if constexpr (CONT) {
        // ---- producers compute step 0: tables of the first tile
        Pix p = pix_issue(g);
        unsigned tab[7];
        build_tab(p, g, tab);
        f2_barrier();
        ++tstep;
        int gstep = 0;
        for (; tile < band_end; tile += slots) {
            const int ntile = tile + slots;
            const bool has_next = ntile < band_end && !(SS_F2_ABLATE & 16);
            Geo gn = g;
            if (has_next) gn = geom(ntile);
            Pix pn = p;
#pragma unroll
            for (int c = 0; c < COUT / 4; ++c) acc[c] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int pass = 0; pass < NPASS; ++pass, ++gstep) {            // gather step g while the producers compute step g + 1
                F2_STAMP(1, 0);
                if (pass == 0 && has_next) window_issue(gn);               // next window: HBM -> registers, in flight over WC steps
                if (pass == 1 && has_next) pn = pix_issue(gn);             // next tile's table look-ups, used after the last pass
                gather(smem + (gstep & 1) * (kF2Rows * kF2PRowB), pass, tab);
#if SS_F2_TRACE
                if (acc[0][0] == 12345.f && acc[7][1] == 1.f) f2_trace[1][63][3] = 1;               // waits for the gather
                F2_STAMP(1, 1);
#endif
                if (pass == WC && has_next) window_commit(gn);             // the producers hold this tile's fragments in registers
                if (pass == NPASS - 1) {
                    store_out(g);
                    if (has_next) build_tab(pn, gn, tab);
                }
                F2_STAMP(1, 2);
                if (pass < NPASS - 1 || has_next) f2_barrier();
                F2_STAMP(1, 3);
                ++tstep;
            }
            g = gn; p = pn;
        }
    } else {
        for (; tile < band_end; tile += slots) {
            // ---- step 0 (producers compute pass 0): per-pixel tables
            Pix p = pix_issue(g);
            unsigned tab[7];
            build_tab(p, g, tab);
#pragma unroll
            for (int c = 0; c < COUT / 4; ++c) acc[c] = (f4){0.f, 0.f, 0.f, 0.f};
            const int ntile = tile + slots;
            const bool has_next = ntile < band_end && !(SS_F2_ABLATE & 16);
            Geo gn = g;
            if (has_next) gn = geom(ntile);                                // scalar table look-ups of the next tile: under the producers' pass 0
            f2_barrier();
            ++tstep;
#pragma unroll 1
            for (int pass = 0; pass < NPASS; ++pass) {                     // gather pass `pass` while the producers compute pass + 1
                const bool last = pass == NPASS - 1;
                F2_STAMP(1, 0);
                if (pass == (NPASS > 3 ? NPASS - 3 : 0) && has_next) window_issue(gn);   // next window: in flight over the last gathers
                gather(smem + (pass & 1) * (kF2Rows * kF2PRowB), pass, tab);
#if SS_F2_TRACE
                if (acc[0][0] == 12345.f && acc[7][1] == 1.f) f2_trace[1][63][3] = 1;               // waits for the gather
                F2_STAMP(1, 1);
#endif
                if (last) {
                    store_out(g);                                          // through the (idle) window region
                    f2_barrier();
                    if (has_next) window_commit(gn);                       // the producers are done with this tile's window
                }
                F2_STAMP(1, 2);
                f2_barrier();
                F2_STAMP(1, 3);
                ++tstep;
            }
            g = gn;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 2-bit packed spike tensors: readers for the consumers of a packed neuron output
// ---------------------------------------------------------------------------------------------------
// packed [n_words] -> dense values.  OUT: 0 = fp32, SS_DT_F16, SS_DT_BF16.  A lane expands one byte (4 neurons); `copies` > 1 writes the
// same 4 values `copies` times with stride `copy_stride` elements: the [X X X] operand of the K-concatenated exact bf16x3 GEMM, where
// a row of C values is followed by its two repetitions (row length C, copies = 3, copy_stride = C, rows become 3*C long).
template <int OUT>
__global__ __launch_bounds__(kBlock) void unpack_spikes_kernel(const unsigned* __restrict__ packed, void* __restrict__ out, long long n4,
                                                               int C, int copies)
{
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long long)gridDim.x * kBlock) {
        const unsigned b = (packed[i >> 2] >> (8 * (int)(i & 3))) & 0xFFu;
        const unsigned c0 = b & 3u, c1 = (b >> 2) & 3u, c2 = (b >> 4) & 3u, c3 = b >> 6;
        long long e = i * 4;
        if (copies > 1) { const long long row = e / C; e = row * (long long)C * copies + (e - row * C); }
        for (int q = 0; q < copies; ++q, e += C) {
            if constexpr (OUT == 0) {
                *reinterpret_cast<f4*>(static_cast<float*>(out) + e) = (f4){(float)c0, (float)c1, (float)c2, (float)c3};
            } else {
                u16x4 o;
                if constexpr (OUT == SS_DT_BF16) { o[0] = code_to_bf16(c0); o[1] = code_to_bf16(c1); o[2] = code_to_bf16(c2); o[3] = code_to_bf16(c3); }
                else { o[0] = narrow<SS_DT_F16>((float)c0); o[1] = narrow<SS_DT_F16>((float)c1); o[2] = narrow<SS_DT_F16>((float)c2); o[3] = narrow<SS_DT_F16>((float)c3); }
                *reinterpret_cast<u16x4*>(static_cast<unsigned short*>(out) + e) = o;
            }
        }
    }
}

// im2col_cl_bf16_kernel reading its NHWC input from a packed spike tensor: 8 channels = 16 bits of one word (C % 8 == 0)
__global__ __launch_bounds__(kBlock) void im2col_cl_bf16_packed_kernel(const unsigned* __restrict__ xp, unsigned short* __restrict__ A,
                                                                       int h, int w, int C, int k, int stride, int pad, int ho, int wo)
{
    const unsigned row = blockIdx.x;
    const unsigned C8 = (unsigned)C / 8;
    const unsigned r = blockIdx.y * kBlock + threadIdx.x;
    if (r >= (unsigned)(k * k) * C8) return;
    const unsigned ox = row % (unsigned)wo, t = row / (unsigned)wo;
    const unsigned oy = t % (unsigned)ho, nb = t / (unsigned)ho;
    const unsigned tap = r / C8, c8 = r - tap * C8;
    const unsigned ky = tap / (unsigned)k, kx = tap - ky * (unsigned)k;
    const int iy = (int)(oy * stride + ky) - pad, ix = (int)(ox * stride + kx) - pad;
    u16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
    if (iy >= 0 && iy < h && ix >= 0 && ix < w) {
        const long long e = (((long long)nb * h + iy) * w + ix) * C + c8 * 8;
        const unsigned bits = (xp[e >> 4] >> (2 * (int)(e & 15))) & 0xFFFFu;
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = code_to_bf16((bits >> (2 * q)) & 3u);
    }
    *reinterpret_cast<u16x8*>(A + ((long long)row * (k * k) + tap) * C + c8 * 8) = o;
}

// ---------------------------------------------------------------------------------------------------
// Weight gradient of a synapse on spike inputs as an EXACT bf16x3 MFMA contraction over the rows:  G_W[ci][n] = sum_r x[r][ci] * g[r][n]
// ---------------------------------------------------------------------------------------------------
// The decoder's weight gradient (fused.py::_UpConvProjectedCL.backward: g_Wt = x^T @ g_P; /root/reference/network/blocks.py:110-132 under
// autograd): x [R][C_in] is a spike tensor (values 0..3: exact in bf16), g [R][N] dense fp32 (N = 25 * C_out columns, row-major).  The
// library's fp32 GEMM runs it at the fp32-MFMA rate (1.4 ms for deconv1 / deconv2 at config 3: compute-bound); here g is split EXACTLY
// into three bf16 terms in registers (truncation split: each residual is exactly representable), every product x * g_s is exact, the
// accumulation is fp32 on v_mfma_f32_32x32x16_bf16 — fp32-GEMM accuracy at the bf16 rate, bound by reading g once from HBM.
//   * the contraction index is the ROW, and an MFMA operand wants 8 consecutive k per lane: a lane loads g[r0 + 8 (lane >> 5) + e][n0 + (lane & 31)],
//     e = 0..7, as 8 dwords (a wavefront instruction covers two full 128-B lines) — the registers ARE the fragment, no LDS, no
//     transposition; the small spike operand is transposed once into fragment order by spike_wgrad_xprep_kernel (2 B/element);
//   * the N / 32 column tiles are dealt to Q workgroup kinds x 8 wavefronts (<= NTW tiles per wavefront), accumulators
//     [NTW][C_in / 32] x 16 registers stay resident while the workgroup walks its slice of the rows (split-K over gridDim.x / Q slices);
//   * partial sums go to ws[slice][n][ci] (coalesced), spike_wgrad_reduce_kernel adds the slices in a fixed order and transposes into
//     G_W[ci][n]: deterministic, no atomics.
constexpr int kSwThreads = 512;
// x [R][C_in] fp32 spike counts -> xT[k-step][ci][16 rows] bf16 (exact): the MFMA operand of a k-step is then ONE 16-B load per lane and
// C_in tile (lane -> 8 consecutive rows of one ci) instead of 8 dword loads
__global__ __launch_bounds__(kBlock) void spike_wgrad_xprep_kernel(const float* __restrict__ x, unsigned short* __restrict__ xT, long long R, int CIN)
{
    const long long KS = (R + 15) / 16, total = KS * CIN;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const long long ks = i / CIN;
        const int ci = (int)(i - ks * CIN);
        u16x8 a, b;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const long long r = 16 * ks + rr;
            const unsigned short v = r < R ? (unsigned short)(__float_as_uint(x[r * CIN + ci]) >> 16) : (unsigned short)0;
            if (rr < 8) a[rr] = v; else b[rr - 8] = v;
        }
        *reinterpret_cast<u16x8*>(xT + i * 16) = a;
        *reinterpret_cast<u16x8*>(xT + i * 16 + 8) = b;
    }
}

template <int CIT, int NTW, int PF>
__global__ __launch_bounds__(kSwThreads) void spike_wgrad_kernel(const float* __restrict__ g, const unsigned short* __restrict__ xT,
                                                                float* __restrict__ ws, long long R, int N, int Q, int CIN)
{
    // CIN: all input channels (xT / ws strides); this workgroup handles the 32 CIT channels starting at ci0 (channel groups are a second
    // kind dimension: blockIdx.x = (slice * CG + channel group) * Q + column kind)
    const int CG = CIN / (32 * CIT);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);           // wave-uniform: keeps the operand base addresses in SGPRs
    const int q = (int)(blockIdx.x % Q), cgi = (int)((blockIdx.x / Q) % CG), slice = (int)(blockIdx.x / (Q * CG)), slices = (int)(gridDim.x / (Q * CG));
    const int ci0 = 32 * CIT * cgi;
    const int NT = N / 32;
    // kind q owns the CONTIGUOUS column tiles [q tpk, (q + 1) tpk): a workgroup then reads one contiguous piece of every row of g
    const int tpk = (NT + Q - 1) / Q, kt = min(tpk, NT - q * tpk);
    const int tile0 = q * tpk + wave;                                            // this wavefront's column tiles: tile0 + 8 j, j < NTW
    if (wave >= kt) return;
    bool own[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) own[j] = wave + 8 * j < kt;
    f32x16 acc[NTW][CIT];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int t = 0; t < CIT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;
    const long long KS = (R + 15) / 16, KSF = R / 16;                            // k-steps in all / complete ones
    const long long per = (KS + slices - 1) / slices;
    const long long ks0 = slice * per, ks1 = min(ks0 + per, KS);
    // Addressing: one 32-bit lane offset for all loads; the row e of the k-step and the column tile j are folded into wave-uniform bases
    const unsigned goff = (unsigned)((lane >> 5) * 8) * (unsigned)N + (unsigned)(lane & 31);          // elements
    const unsigned xoff = (unsigned)(lane & 31) * 16u + (unsigned)(lane >> 5) * 8u;                  // bf16 elements
    float gv[PF][NTW][8];
    s16x8 xn[PF][CIT];
    auto load_step = [&](float (&gd)[NTW][8], s16x8 (&xd)[CIT], long long ks) {
        const unsigned short* xb = xT + (ks * CIN + ci0) * 16;
#pragma unroll
        for (int t = 0; t < CIT; ++t) xd[t] = *reinterpret_cast<const s16x8*>(xb + 32 * 16 * t + xoff);
        if (ks < KSF) {                                                          // all 16 rows exist: uniform bases
            const float* gb = g + ks * 16 * N + 32 * tile0;
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int j = 0; j < NTW; ++j) gd[j][e] = own[j] ? load_stream(gb + (long long)e * N + 32 * 8 * j + goff) : 0.f;
        } else {                                                                 // the ragged last k-step: rows beyond R read row R - 1
#pragma unroll                                                                   // (finite) and meet the zero rows of xT
            for (int e = 0; e < 8; ++e) {
                const long long r = min(16 * ks + 8 * (lane >> 5) + e, R - 1);
#pragma unroll
                for (int j = 0; j < NTW; ++j) gd[j][e] = own[j] ? g[r * N + 32 * (tile0 + 8 * j) + (lane & 31)] : 0.f;
            }
        }
    };
#pragma unroll
    for (int u = 0; u < PF - 1; ++u)
        if (ks0 + u < ks1) load_step(gv[u], xn[u], ks0 + u);
#pragma unroll 1
    for (long long ks = ks0; ks < ks1; ks += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (ks + u < ks1) {
                if (ks + u + PF - 1 < ks1) load_step(gv[(u + PF - 1) % PF], xn[(u + PF - 1) % PF], ks + u + PF - 1);
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    if (own[j]) {
                        s16x8 gs[3];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float v = gv[u][j][e];
                            const unsigned uh = __float_as_uint(v) & 0xFFFF0000u;
                            const float r1 = v - __uint_as_float(uh);                                          // exact
                            const unsigned um = __float_as_uint(r1) & 0xFFFF0000u;
                            const float r2 = r1 - __uint_as_float(um);                                         // exact, <= 8 significant bits
                            gs[0][e] = (short)(uh >> 16); gs[1][e] = (short)(um >> 16); gs[2][e] = (short)(__float_as_uint(r2) >> 16);
                        }
#pragma unroll
                        for (int sp = 0; sp < 3; ++sp)
#pragma unroll
                            for (int t = 0; t < CIT; ++t)
                                acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gs[sp], xn[u][t], acc[j][t], 0, 0, 0);
                    }
                }
            }
        }
    }
    // D[n][ci]: column (ci) = lane & 31, row (n) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float* const wsl = ws + (long long)slice * N * CIN;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        if (own[j]) {
#pragma unroll
            for (int t = 0; t < CIT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = 32 * (tile0 + 8 * j) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    wsl[(long long)n * CIN + ci0 + 32 * t + (lane & 31)] = acc[j][t][r];
                }
        }
    }
}

// G_W[ci][n] (+)= sum over slices of ws[slice][n][ci], slices in ascending order
__global__ __launch_bounds__(kBlock) void spike_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ gw, int slices, int N, int CIN,
                                                                   int accumulate)
{
    const long long total = (long long)N * CIN;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int n = (int)(i / CIN), ci = (int)(i - (long long)n * CIN);
        float a = 0.f;
        for (int sIdx = 0; sIdx < slices; ++sIdx) a += ws[(long long)sIdx * total + i];
        float* o = gw + (long long)ci * N + n;
        *o = accumulate ? *o + a : a;
    }
}

// ---------------------------------------------------------------------------------------------------
// Weight gradient of a 5x5 / stride 2 / pad 2 convolution on SPIKE inputs as an exact bf16x3 MFMA contraction over the output pixels
// ---------------------------------------------------------------------------------------------------
// Reference: autograd of conv1 / conv2 (nn.Conv2d(32, 64, 5, 2, 2) / (64, 128, 5, 2, 2), /root/reference/network/SNN_models.py:80-90) w.r.t.
// their weight:  g_w[co][ci][ky][kx] = sum_{nb, oy, ox} g[nb][oy][ox][co] * x[nb][2 oy + ky - 2][2 ox + kx - 2][ci],  x a spike tensor.
// Same scheme as spike_wgrad_kernel — the contraction index (16 consecutive ox of one output row = one k-step) is what a lane holds 8
// consecutive values of; g is split exactly into three bf16 terms in registers; products exact, fp32 accumulation — with the spike operand
// of tap (ky, kx) read from five column-decimated bf16 copies of x (one per kx: xK[kx][nb][iy + 2][ox / 8][ci][ox % 8] =
// x[nb][iy][2 ox + kx - 2][ci], zero padded), so that the fragment of 8 consecutive ox is ONE aligned 16-B load, coalesced over ci.  The 25 C_in / 32 "virtual
// channel" tiles (tap, ci tile) are dealt to workgroup kinds x wavefronts; every wavefront keeps NVC x (C_out / 32) accumulator tiles.
template <int CIT, int COT, int NVC>
__global__ __launch_bounds__(kSwThreads) void spike_conv_wgrad_kernel(const unsigned short* __restrict__ gT, const unsigned short* __restrict__ xK,
                                                                     float* __restrict__ ws, int NB, int h, int ho, int wo, int Q)
{
    // wavefront = (C_out tile, group of virtual-channel tiles): a wavefront loads the three pre-split g fragments of ITS C_out tile (the first
    // version split g in every wavefront: 56x redundant VALU work) and NVC spike fragments per k-step, and issues 3 NVC MFMAs
    constexpr int CIN = 32 * CIT, COUT = 32 * COT, NV = 25 * CIT, NG = 8 / COT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cot = wave % COT, grp = wave / COT;
    const int q = (int)(blockIdx.x % Q), slice = (int)(blockIdx.x / Q), slices = (int)(gridDim.x / Q);
    const int vpk = (NV + Q - 1) / Q, kv = min(vpk, NV - q * vpk);
    const int v0 = q * vpk + grp;                                          // this wavefront's virtual-channel tiles: v0 + NG j, j < NVC
    if (grp >= kv) return;
    bool own[NVC];
    long long xbase[NVC];
    const int KSR = (wo + 15) / 16, OX8 = 2 * KSR, HP = h + 4;
#pragma unroll
    for (int j = 0; j < NVC; ++j) {
        own[j] = grp + NG * j < kv;
        const int v = own[j] ? v0 + NG * j : v0;
        const int tap = v / CIT, cit = v - tap * CIT, ky = tap / 5, kx = tap - 5 * ky;
        xbase[j] = (((((long long)kx * NB) * HP + ky) * OX8 + (lane >> 5)) * CIN + 32 * cit + (lane & 31)) * 8;
    }
    f32x16 acc[NVC];
#pragma unroll
    for (int j = 0; j < NVC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const long long KS = (long long)NB * ho * KSR;
    const long long per = (KS + slices - 1) / slices;
    const long long ks0 = slice * per, ks1 = min(ks0 + per, KS);
    constexpr int PF = 2;
    s16x8 gs[PF][3], xn[PF][NVC];
    auto load_step = [&](s16x8 (&gd)[3], s16x8 (&xd)[NVC], long long ks) {
        const int c = (int)(ks % KSR);
        const long long ro = ks / KSR;                                     // nb * ho + oy
        const int oy = (int)(ro % ho);
        const long long nb = ro / ho;
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) gd[sp] = *reinterpret_cast<const s16x8*>(gT + (((ks * 3 + sp) * COT + cot) * 64 + lane) * 8);
#pragma unroll
        for (int j = 0; j < NVC; ++j)
            xd[j] = *reinterpret_cast<const s16x8*>(xK + xbase[j] + (((nb * HP + 2 * oy) * OX8 + 2 * c) * CIN) * 8LL);
    };
    if (ks0 < ks1) load_step(gs[0], xn[0], ks0);
#pragma unroll 1
    for (long long ks = ks0; ks < ks1; ks += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (ks + u < ks1) {
                if (ks + u + 1 < ks1) load_step(gs[(u + 1) % PF], xn[(u + 1) % PF], ks + u + 1);
#pragma unroll
                for (int sp = 0; sp < 3; ++sp)
#pragma unroll
                    for (int j = 0; j < NVC; ++j)
                        if (own[j]) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gs[u][sp], xn[u][j], acc[j], 0, 0, 0);
            }
        }
    }
    // D[co][ci]: column (ci) = lane & 31, row (co) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5);  ws[slice][virtual channel v * 32 + ci][co]
    float* const wsl = ws + (long long)slice * NV * 32 * COUT;
#pragma unroll
    for (int j = 0; j < NVC; ++j) {
        if (own[j]) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * cot + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                wsl[((long long)(v0 + NG * j) * 32 + (lane & 31)) * COUT + co] = acc[j][r];
            }
        }
    }
}

// g [NB * ho][wo][C_out] fp32 -> gT[k-step][split][C_out tile][lane][8] bf16: the exact three-term split of g in MFMA fragment order (lane ->
// co = 32 tile + (lane & 31), the 8 consecutive ox of its half of the k-step; zero beyond wo)
__global__ __launch_bounds__(kBlock) void spike_conv_gprep_kernel(const float* __restrict__ g, unsigned short* __restrict__ gT, long long rows, int wo,
                                                                  int COUT)
{
    const int KSR = (wo + 15) / 16, COT = COUT / 32;
    const long long total = rows * KSR * COT * 64;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int lane = (int)(i & 63);
        long long r = i >> 6;
        const int t = (int)(r % COT); r /= COT;
        const int c = (int)(r % KSR); const long long ro = r / KSR;
        const long long ks = ro * KSR + c;
        u16x8 o[3];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ox = 16 * c + 8 * (lane >> 5) + e;
            const float v = ox < wo ? g[(ro * wo + ox) * COUT + 32 * t + (lane & 31)] : 0.f;
            const unsigned uh = __float_as_uint(v) & 0xFFFF0000u;
            const float r1 = v - __uint_as_float(uh);
            const unsigned um = __float_as_uint(r1) & 0xFFFF0000u;
            const float r2 = r1 - __uint_as_float(um);
            o[0][e] = (unsigned short)(uh >> 16); o[1][e] = (unsigned short)(um >> 16); o[2][e] = (unsigned short)(__float_as_uint(r2) >> 16);
        }
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) *reinterpret_cast<u16x8*>(gT + (((ks * 3 + sp) * COT + t) * 64 + lane) * 8) = o[sp];
    }
}

// g_w[co][ci][ky][kx] (+)= sum over slices of ws[slice][(tap * CIT + ci / 32) * 32 + ci % 32][co]
__global__ __launch_bounds__(kBlock) void spike_conv_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ gw, int slices, int CIN, int COUT,
                                                                        int accumulate)
{
    const int CIT = CIN / 32;
    const long long per = 25LL * CIN * COUT;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < per; i += (long long)gridDim.x * kBlock) {
        const int co = (int)(i % COUT);
        const long long vc = i / COUT;                                     // (tap * CIT + cit) * 32 + cil
        const int cil = (int)(vc & 31), v = (int)(vc >> 5);
        const int tap = v / CIT, ci = 32 * (v - tap * CIT) + cil;
        float a = 0.f;
        for (int sIdx = 0; sIdx < slices; ++sIdx) a += ws[(long long)sIdx * per + i];
        float* o = gw + ((long long)co * CIN + ci) * 25 + tap;
        *o = accumulate ? *o + a : a;
    }
}

// x [NB][h][w][C] fp32 spike counts -> xK[kx][nb][iy + 2][ox / 8][ci][ox % 8] bf16 = x[nb][iy][2 ox + kx - 2][ci] (zero outside), ox < 16 ceil(wo / 16).
// A lane owns (nb, padded row, 8-ox chunk, ci): 19 input columns -> the five kx fragments; reads and 16-B writes coalesced over ci.
template <bool PACKED>
__global__ __launch_bounds__(kBlock) void spike_conv_xprep_kernel(const void* __restrict__ xv, unsigned short* __restrict__ xK, int NB, int h, int w, int C,
                                                                  int wo)
{
    const float* x = static_cast<const float*>(xv);
    const unsigned* xp = static_cast<const unsigned*>(xv);                  // PACKED: the 2-bit packed spike tensor (16 neurons per word)
    const int OX8 = 2 * ((wo + 15) / 16), HP = h + 4;
    const long long total = (long long)NB * HP * OX8 * C;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int ci = (int)(i % C);
        long long r = i / C;
        const int o8 = (int)(r % OX8); r /= OX8;
        const int iyp = (int)(r % HP); const int nb = (int)(r / HP);
        const int iy = iyp - 2;
        unsigned short v[19];
#pragma unroll
        for (int t = 0; t < 19; ++t) {
            const int ix = 16 * o8 - 2 + t;
            const bool ok = iy >= 0 && iy < h && ix >= 0 && ix < w;
            const long long el = (((long long)nb * h + iy) * w + ix) * C + ci;
            if constexpr (PACKED) v[t] = ok ? code_to_bf16((xp[el >> 4] >> (2 * (int)(el & 15))) & 3u) : (unsigned short)0;
            else v[t] = ok ? (unsigned short)(__float_as_uint(x[el]) >> 16) : (unsigned short)0;
        }
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
            u16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = v[2 * e + kx];
            *reinterpret_cast<u16x8*>(xK + ((((((long long)kx * NB + nb) * HP + iyp) * OX8 + o8) * C + ci) * 8)) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Decoder backward, fused: adjoint gather (g_P) + exact bf16x3 MFMA weight gradient in ONE pass — g_P is written once for the data-gradient
// GEMM and never read back for the weight gradient
// ---------------------------------------------------------------------------------------------------
// Reference: autograd through NNConvUpsampling (/root/reference/network/blocks.py:110-132; decoder stages SNN_models.py:110-129).
//   g_P[src][tap][co] = sum over the output pixels whose tap lands on src of g_y[pix][co]        (ss_upconv_cl_bwd_f32: same value, same
//                                                                                                 summation order — rows top to bottom, columns
//                                                                                                 left to right inside a row)
//   g_W[ci][tap, co]  = sum_src x[src][ci] * g_P[src][tap][co]                                   (ss_spike_wgrad_f32: x spikes, exact products)
// A lane computes g_P for 8 CONSECUTIVE SOURCE PIXELS of one column n = (tap, co) — exactly the MFMA fragment of the contraction over
// sources (k = 8 (lane >> 5) + e, row n = lane & 31) — from the g_y window of the tile held in LDS ([row][col][co] fp32, lanes = consecutive
// co: conflict-free), stores the 8 values to g_P (128-B segments), splits them exactly into three bf16 terms and multiplies them with the
// spike fragments (pre-transposed, one source row = ceil(w / 16) k-steps, zero padded).  Work split as in spike_wgrad_kernel: Q workgroup
// kinds own contiguous ranges of the 25 C_out / 32 column tiles, a wavefront keeps its <= NTW tiles' accumulators for the whole launch,
// slices of the (frame, TR source rows, 16 source columns) tiles are walked persistently; partials -> ws -> spike_wgrad_reduce_kernel.
template <int CIN, int COUT, int TR, int NTW, int WRM, int WCM>
__global__ __launch_bounds__(kSwThreads, CIN == 64 ? 4 : 2) void upconv_bwd_fused_kernel(const float* __restrict__ gy, const unsigned short* __restrict__ xT,
                                                                     const int* __restrict__ y_lo, const int* __restrict__ y_hi,
                                                                     const int* __restrict__ x_lo, const int* __restrict__ x_hi,
                                                                     float* __restrict__ gP, float* __restrict__ ws,
                                                                     int NB, int h, int w, int H, int W, int Q)
{
    constexpr int CIT = CIN / 32, N = 25 * COUT, NT = N / 32, NPT = COUT / 32, CB = COUT * 4, ROWB = WCM * CB, C4 = COUT / 4;
    __shared__ __attribute__((aligned(16))) unsigned char wnd[WRM * ROWB + 2 * CB];     // + slack: the unconditional third column read
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = (int)(blockIdx.x % Q), slice = (int)(blockIdx.x / Q), slices = (int)(gridDim.x / Q);
    const int tpk = (NT + Q - 1) / Q, kt = min(tpk, NT - q * tpk);
    const int tile0 = q * tpk + wave;
    bool own[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) own[j] = wave + 8 * j < kt;
    f32x16 acc[NTW][CIT];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int t = 0; t < CIT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;
    const int KSR = (w + 15) / 16, RG = (h + TR - 1) / TR;
    const int n_tiles = NB * RG * KSR;
    const unsigned xoffT = (unsigned)(lane & 31) * 16u + (unsigned)(lane >> 5) * 8u;
#pragma unroll 1
    for (int t = slice; t < n_tiles; t += slices) {
        const int c = t % KSR, rg = (t / KSR) % RG, nb = t / (KSR * RG);
        const int sy0 = rg * TR, sx0 = 16 * c, nrow = min(TR, h - sy0);
        const int wy0 = y_lo[sy0] - 4, WRt = y_hi[sy0 + nrow - 1] - wy0;
        const int wx0 = x_lo[sx0] - 4, WCt = x_hi[min(sx0 + 15, w - 1)] - wx0;
        __syncthreads();                                                  // the previous tile's reads of the window are done
        // ---- g_y window -> LDS (zero outside the image)
        const int rowf4 = min(WCt + 2, WCM) * C4;                          // + 2: the multiplied-by-zero third column must be finite
        for (int i = threadIdx.x; i < WRt * rowf4; i += kSwThreads) {
            const int wy = i / rowf4, rem = i - wy * rowf4;
            const int y = wy0 + wy, x = wx0 + rem / C4;
            f4 v = {0.f, 0.f, 0.f, 0.f};
            if (y >= 0 && y < H && x >= 0 && x < W)
                v = load_stream(reinterpret_cast<const f4*>(gy + (((long long)nb * H + y) * W + x) * COUT) + (rem % C4));
            *reinterpret_cast<f4*>(wnd + wy * ROWB + rem * 16) = v;
        }
        // ---- this lane's 8 source columns: window byte offset of their first contributing column at kx = 0, and 0 / 1 multipliers of the
        //      second / third column (fma(r, 1, cs) == cs + r and fma(r, 0, cs) == cs exactly: the adjoint kernel's sums, no selects)
        int xo[8];
        float m1[8], m2[8];
        unsigned valid = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int sx = sx0 + 8 * (lane >> 5) + e;
            const int sxc = min(sx, w - 1);
            const int lo = x_lo[sxc], hi = x_hi[sxc];
            xo[e] = (lo - wx0) * CB;
            m1[e] = hi - lo > 1 ? 1.f : 0.f;
            m2[e] = hi - lo > 2 ? 1.f : 0.f;
            if (sx < w) valid |= 1u << e;
        }
        __syncthreads();
        if (own[0]) {
#pragma unroll 1
            for (int tr = 0; tr < nrow; ++tr) {
                const int sy = sy0 + tr;
                const int ys = y_lo[sy] - wy0, ry = y_hi[sy] - y_lo[sy];
                const long long srow = (long long)nb * h + sy;                // source row index
                s16x8 xn[CIT];
                {
                    const unsigned short* xb = xT + (srow * KSR + c) * (CIN * 16);
#pragma unroll
                    for (int tt = 0; tt < CIT; ++tt) xn[tt] = *reinterpret_cast<const s16x8*>(xb + 32 * 16 * tt + xoffT);
                }
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    if (own[j]) {
                        const int tile = tile0 + 8 * j;
                        const int tap = tile / NPT, half = tile - tap * NPT;
                        const int ky = tap / 5, kx = tap - 5 * ky;
                        const unsigned char* const base = wnd + (ys - ky) * ROWB - kx * CB + (half * 32 + (lane & 31)) * 4;
                        float gv[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const unsigned char* const p = base + xo[e];      // rows / columns at immediate offsets
                            float a = 0.f;
#pragma unroll
                            for (int rr = 0; rr < 3; ++rr) {
                                if (rr < ry) {                                // wave-uniform
                                    float cs = *reinterpret_cast<const float*>(p + rr * ROWB);
                                    cs = __builtin_fmaf(*reinterpret_cast<const float*>(p + rr * ROWB + CB), m1[e], cs);
                                    cs = __builtin_fmaf(*reinterpret_cast<const float*>(p + rr * ROWB + 2 * CB), m2[e], cs);
                                    a += cs;
                                }
                            }
                            gv[e] = a;
                        }
                        // g_P[(srow * w + sx)][n], n = 32 tile + (lane & 31)
                        float* const gp = gP + (srow * w + sx0 + 8 * (lane >> 5)) * N + 32 * tile + (lane & 31);
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (valid & (1u << e)) store_out(gp + (long long)e * N, gv[e]);
                        s16x8 gs[3];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float v = gv[e];
                            const unsigned uh = __float_as_uint(v) & 0xFFFF0000u;
                            const float r1 = v - __uint_as_float(uh);
                            const unsigned um = __float_as_uint(r1) & 0xFFFF0000u;
                            const float r2 = r1 - __uint_as_float(um);
                            gs[0][e] = (short)(uh >> 16); gs[1][e] = (short)(um >> 16); gs[2][e] = (short)(__float_as_uint(r2) >> 16);
                        }
#pragma unroll
                        for (int sp = 0; sp < 3; ++sp)
#pragma unroll
                            for (int tt = 0; tt < CIT; ++tt)
                                acc[j][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gs[sp], xn[tt], acc[j][tt], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (!own[0]) return;
    float* const wsl = ws + (long long)slice * N * CIN;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        if (own[j]) {
#pragma unroll
            for (int tt = 0; tt < CIT; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = 32 * (tile0 + 8 * j) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    wsl[(long long)n * CIN + 32 * tt + (lane & 31)] = acc[j][tt][r];
                }
        }
    }
}

// x [NB * h][w][C_in] fp32 spike counts -> xT[(source row) * KSR + k-step][ci][16 sources] bf16, KSR = ceil(w / 16), zero padded
__global__ __launch_bounds__(kBlock) void upconv_bwd_xprep_kernel(const float* __restrict__ x, unsigned short* __restrict__ xT, long long rows, int w,
                                                                  int CIN)
{
    const int KSR = (w + 15) / 16;
    const long long total = rows * KSR * CIN;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int ci = (int)(i % CIN);
        const long long k = i / CIN;
        const int c = (int)(k % KSR);
        const long long row = k / KSR;
        u16x8 a, b;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int sx = 16 * c + rr;
            const unsigned short v = sx < w ? (unsigned short)(__float_as_uint(x[(row * w + sx) * CIN + ci]) >> 16) : (unsigned short)0;
            if (rr < 8) a[rr] = v; else b[rr - 8] = v;
        }
        *reinterpret_cast<u16x8*>(xT + i * 16) = a;
        *reinterpret_cast<u16x8*>(xT + i * 16 + 8) = b;
    }
}

// ---------------------------------------------------------------------------------------------------
// Dense x dense fp32 GEMM on the bf16 matrix cores with SIX cross terms:  C[R][N] = A[R][K] @ B[K][N]   (decoder data gradient g_x = g_P @ W2)
// ---------------------------------------------------------------------------------------------------
// Both operands are dense fp32 (no spike operand), so the exact 3-term split of ONE operand is not enough.  a = ah + am + al and
// b = bh + bm + bl exactly (round-to-nearest splits, |am| <= 2^-8 |a|, |al| <= 2^-16 |a|); of the nine products the six
//   ah bh + ah bm + am bh + ah bl + am bm + al bh
// are kept (each exact in fp32), the three dropped ones are <= (2 * 2^-24 + 2^-32) |a b|: the rounding of ONE fp32 product.  Accumulation is
// fp32 in the MFMA.  Error bound asserted in tests/: |C - C_float64| <= 2^-21 sum_k |a||b| (measured worst element: 1.05 x 2^-22).  6 bf16 MFMAs per fp32-MFMA-equivalent at 16x the
// rate: the library's fp32 GEMM is compute-bound at 110 - 133 TFLOP/s on these shapes (K = 800 .. 6400, N = 64 .. 512).
//   * workgroup = 8 wavefronts x 32 rows; all N <= 256 columns per workgroup (N = 512: two column halves, A read twice);
//   * A: a lane loads its row's 8 consecutive k (32 B) per k-step and splits them in registers — the registers are the fragments; a ring of
//     KB k-steps keeps one LDS stage of loads in flight;
//   * B: split once into fragment order by gemm6_prep_b_kernel, streamed through a double-buffered LDS stage of KB k-steps (all 8
//     wavefronts read the same fragments), one barrier per stage.
constexpr int kG6Threads = 512;
template <int CIT, int KB>
__global__ __launch_bounds__(kG6Threads) void gemm6_kernel(const float* __restrict__ A, const unsigned short* __restrict__ Bf, float* __restrict__ C,
                                                          long long R, int K, int N, int col_kinds)
{
    constexpr int STG = KB * 3 * CIT * 1024;                               // bytes of one B stage (KB k-steps x 3 splits x CIT tiles x 1 KiB)
    constexpr int LPT = STG / 16 / kG6Threads;                             // 16-B pieces per thread per stage
    static_assert(STG % (16 * kG6Threads) == 0, "stage must divide among the threads");
    __shared__ __attribute__((aligned(16))) unsigned char bs[2 * STG];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kind = (int)(blockIdx.x % col_kinds);
    const long long m0 = (long long)(blockIdx.x / col_kinds) * 256 + 32 * wave;
    A += (long long)blockIdx.y * R * K; C += (long long)blockIdx.y * R * N; Bf += (long long)blockIdx.y * K * N * 3;    // batch (blockIdx.y)
    const int KS = K / 16, NST = (KS + KB - 1) / KB;
    const int NTall = N / 32;                                              // column tiles of B in all; this workgroup: [kind * CIT, kind * CIT + CIT)
    const long long row = min(m0 + (lane & 31), R - 1);
    const float* const arow = A + row * K + 8 * (lane >> 5);
    f32x16 acc[CIT];
#pragma unroll
    for (int t = 0; t < CIT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // B stage addressing: the stage's pieces are [k-step][split][tile of this kind][lane][16 B]; global Bf is [k-step][split][all tiles][lane][8 bf16]
    f4 st[LPT];
    auto stage_issue = [&](int stg) {
#pragma unroll
        for (int u = 0; u < LPT; ++u) {
            const int pidx = threadIdx.x + kG6Threads * u;                  // 16-B piece within the stage
            const int ln = pidx & 63, tt = (pidx >> 6) % CIT, sp = ((pidx >> 6) / CIT) % 3, kk = (pidx >> 6) / (3 * CIT);
            const int ks = min(stg * KB + kk, KS - 1);
            st[u] = *reinterpret_cast<const f4*>(Bf + ((((long long)ks * 3 + sp) * NTall + kind * CIT + tt) * 64 + ln) * 8);
        }
    };
    auto stage_commit = [&](int buf) {
#pragma unroll
        for (int u = 0; u < LPT; ++u) *reinterpret_cast<f4*>(bs + buf * STG + (threadIdx.x + kG6Threads * u) * 16) = st[u];
    };
    f4 av[KB][2];
    auto a_load = [&](f4 (&d)[2], int ks) {
        const float* p = arow + 16 * min(ks, KS - 1);
        d[0] = load_stream(reinterpret_cast<const f4*>(p));
        d[1] = load_stream(reinterpret_cast<const f4*>(p + 4));
    };
    stage_issue(0);
#pragma unroll
    for (int j = 0; j < KB; ++j) a_load(av[j], j);
    stage_commit(0);
    __syncthreads();
    // The bf16 MFMA's fp32 accumulation is not exactly round-to-nearest: measured against float64 every accumulator drifts DOWN by ~2^-28 of the
    // magnitude sum (tools/diag_gemm6_bias.py; the fp32 MFMA of the library shows 1e-11).  Harmless per element, but coherent over all elements:
    // a cancelling reduction of the result (a PLIF node's scalar dL/dw) lost two digits.  So the sign of the running sum alternates every kFlip
    // stages — acc = -acc and A enters negated — which turns the drift of the negative phases upward and cancels it in expectation.
    constexpr int kFlip = 4;
    bool neg = false;
#pragma unroll 1
    for (int stg = 0; stg < NST; ++stg) {
        const bool more = stg + 1 < NST;
        if (more) stage_issue(stg + 1);
        if (((stg / kFlip) & 1) != (int)neg) {
            neg = !neg;
#pragma unroll
            for (int t = 0; t < CIT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = -acc[t][r];
        }
        const float sgn = neg ? -1.f : 1.f;
        const unsigned char* const bb = bs + (stg & 1) * STG + lane * 16;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            if (stg * KB + j < KS) {
                // split this k-step's 8 values of A (round to nearest: residuals <= 2^-8, 2^-16)
                s16x8 ah, am, al;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = av[j][e >> 2][e & 3] * sgn;             // (__bf16) casts: v_cvt_pk_bf16_f32 on gfx950 (round to nearest even)
                    const __bf16 h1 = (__bf16)v;
                    const float r1 = v - (float)h1;
                    const __bf16 h2 = (__bf16)r1;
                    const float r2 = r1 - (float)h2;
                    const __bf16 h3 = (__bf16)r2;
                    ah[e] = __builtin_bit_cast(short, h1); am[e] = __builtin_bit_cast(short, h2); al[e] = __builtin_bit_cast(short, h3);
                }
                if (more) a_load(av[j], (stg + 1) * KB + j);                // this slot's next occupant: one stage ahead
                const unsigned char* const bk = bb + j * (3 * CIT * 1024);
                // column tiles two at a time, term-major (consecutive MFMAs alternate between two accumulators); the fragments of the NEXT
                // pair are read from LDS before this pair's 12 MFMAs are issued (pinned: hipcc otherwise places each read right before its use)
                s16x8 bq[2][6];
                auto b_read = [&](s16x8 (&d)[6], int t) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        d[0 + u] = *reinterpret_cast<const s16x8*>(bk + (0 * CIT + t + u) * 1024);
                        d[2 + u] = *reinterpret_cast<const s16x8*>(bk + (1 * CIT + t + u) * 1024);
                        d[4 + u] = *reinterpret_cast<const s16x8*>(bk + (2 * CIT + t + u) * 1024);
                    }
                };
                b_read(bq[0], 0);
#pragma unroll
                for (int t = 0; t < CIT; t += 2) {
                    const int cur = (t >> 1) & 1;
                    if (t + 2 < CIT) b_read(bq[cur ^ 1], t + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    const s16x8 (&b)[6] = bq[cur];                            // [0,1] hi, [2,3] mid, [4,5] lo of tiles t, t + 1
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[t + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[4 + u], acc[t + u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[t + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b[2 + u], acc[t + u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[t + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b[0 + u], acc[t + u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[t + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[2 + u], acc[t + u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[t + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b[0 + u], acc[t + u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[t + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[0 + u], acc[t + u], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (more) stage_commit((stg + 1) & 1);
        __syncthreads();
    }
    // D[row][col]: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const float fin = neg ? -1.f : 1.f;
#pragma unroll
    for (int t = 0; t < CIT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long rr = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (rr < R) store_out(C + rr * N + 32 * (kind * CIT + t) + (lane & 31), acc[t][r] * fin);
        }
}

// B [K][N] fp32 -> Bf[k-step][split][column tile][lane][8] bf16, element e = split term of B[16 ks + 8 (lane >> 5) + e][32 tile + (lane & 31)]
__global__ __launch_bounds__(kBlock) void gemm6_prep_b_kernel(const float* __restrict__ B, unsigned short* __restrict__ Bf, int K, int N)
{
    const int KS = K / 16, NT = N / 32;
    const long long total = (long long)KS * 3 * NT * 64;
    B += (long long)blockIdx.y * K * N; Bf += (long long)blockIdx.y * K * N * 3;                                           // batch (blockIdx.y)
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int lane = (int)(i & 63);
        long long r = i >> 6;
        const int tile = (int)(r % NT); r /= NT;
        const int sp = (int)(r % 3); const int ks = (int)(r / 3);
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = B[(long long)(16 * ks + 8 * (lane >> 5) + e) * N + 32 * tile + (lane & 31)];
            const unsigned short h1 = narrow<SS_DT_BF16>(v);
            const float r1 = v - widen<SS_DT_BF16>(h1);
            const unsigned short h2 = narrow<SS_DT_BF16>(r1);
            const float r2 = r1 - widen<SS_DT_BF16>(h2);
            o[e] = sp == 0 ? h1 : (sp == 1 ? h2 : narrow<SS_DT_BF16>(r2));
        }
        *reinterpret_cast<u16x8*>(Bf + i * 8) = o;
    }
}

// ---------------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) DATA GRADIENT of a 3x3 / stride 1 / pad 1 convolution, NHWC (the four bottleneck convs)
// ---------------------------------------------------------------------------------------------------
// Reference: the autograd backward of SEWResBlock's conv1 / conv2 (/root/reference/network/blocks.py:146-159) w.r.t. their input:
//   g_in[nb][y][x][ci] = sum_{co, a, b} g[nb][y + a - 1][x + b - 1][co] * Wf[a][b][co][ci],      Wf[a][b][co][ci] = W[co][ci][2 - a][2 - b]
// — a dense x dense contraction (no spike operand, so no exact bf16 split).  As 2 x 2 output tiles on 4 x 4 input tiles (Lavin & Gray):
//   V = B^T d B (input transform), U = G Wf G^T (weights), M_k = V_k U_k for the 16 transform positions k (ONE batched fp32 GEMM
//   [16][tiles x C_out] @ [16][C_out x C_in] on the library), Y = A^T M A (output transform): 2.25x fewer multiplications than the
//   direct form.  fp32 throughout; the transforms only add / subtract (weights: x 0.5), every op rounds once (-ffp-contract=off), so the
//   three kernels are bit-exact against oracle/np_winograd.py.
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
__global__ __launch_bounds__(kBlock) void wino_dgrad_weights_kernel(const float* __restrict__ Wt, float* __restrict__ U, int Co, int Ci)
{
    const long long n = (long long)Co * Ci;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        const int co = (int)(i / Ci), ci = (int)(i - (long long)co * Ci);
        const float* wp = Wt + i * 9;                                   // W[co][ci][ky][kx]
        float f[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) f[a][b] = wp[(2 - a) * 3 + (2 - b)];
        float t[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            t[0][b] = f[0][b];
            t[1][b] = 0.5f * ((f[0][b] + f[1][b]) + f[2][b]);
            t[2][b] = 0.5f * ((f[0][b] - f[1][b]) + f[2][b]);
            t[3][b] = f[2][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float u0 = t[a][0], u1 = 0.5f * ((t[a][0] + t[a][1]) + t[a][2]), u2 = 0.5f * ((t[a][0] - t[a][1]) + t[a][2]), u3 = t[a][2];
            U[((long long)(4 * a + 0) * Co + co) * Ci + ci] = u0;
            U[((long long)(4 * a + 1) * Co + co) * Ci + ci] = u1;
            U[((long long)(4 * a + 2) * Co + co) * Ci + ci] = u2;
            U[((long long)(4 * a + 3) * Co + co) * Ci + ci] = u3;
        }
    }
}

// g [NB][H][W][C] -> V[16][T][C], T = NB * th * tw tiles (th = ceil(H / 2), tw = ceil(W / 2)); a lane owns 4 consecutive channels of a tile
__global__ __launch_bounds__(kBlock) void wino_dgrad_input_kernel(const float* __restrict__ g, float* __restrict__ V, long long T, int H, int W,
                                                                  int C, int th, int tw)
{
    const int C4 = C / 4;
    const long long n = T * C4;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        const long long tile = i / C4;
        const int c = (int)(i - tile * C4) * 4;
        const int tx = (int)(tile % tw);
        const long long r = tile / tw;
        const int ty = (int)(r % th);
        const long long nb = r / th;
        f4 d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int y = 2 * ty - 1 + a;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int x = 2 * tx - 1 + b;
                d[a][b] = (y >= 0 && y < H && x >= 0 && x < W) ? *reinterpret_cast<const f4*>(g + ((nb * H + y) * W + x) * C + c)
                                                               : (f4){0.f, 0.f, 0.f, 0.f};
            }
        }
        f4 t[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            t[0][b] = d[0][b] - d[2][b];
            t[1][b] = d[1][b] + d[2][b];
            t[2][b] = d[2][b] - d[1][b];
            t[3][b] = d[1][b] - d[3][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float* vp = V + ((long long)(4 * a) * T + tile) * C + c;
            *reinterpret_cast<f4*>(vp) = t[a][0] - t[a][2];
            *reinterpret_cast<f4*>(vp + T * C) = t[a][1] + t[a][2];
            *reinterpret_cast<f4*>(vp + 2 * T * C) = t[a][2] - t[a][1];
            *reinterpret_cast<f4*>(vp + 3 * T * C) = t[a][1] - t[a][3];
        }
    }
}

// M[16][T][C] -> g_in [NB][H][W][C]
__global__ __launch_bounds__(kBlock) void wino_dgrad_output_kernel(const float* __restrict__ M, float* __restrict__ gin, long long T, int H, int W,
                                                                   int C, int th, int tw)
{
    const int C4 = C / 4;
    const long long n = T * C4;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        const long long tile = i / C4;
        const int c = (int)(i - tile * C4) * 4;
        const int tx = (int)(tile % tw);
        const long long r = tile / tw;
        const int ty = (int)(r % th);
        const long long nb = r / th;
        f4 m[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) m[a][b] = *reinterpret_cast<const f4*>(M + ((long long)(4 * a + b) * T + tile) * C + c);
        f4 t[2][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            t[0][b] = (m[0][b] + m[1][b]) + m[2][b];
            t[1][b] = (m[1][b] - m[2][b]) - m[3][b];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int y = 2 * ty + a;
            if (y >= H) continue;
            float* op = gin + ((nb * H + y) * W + 2 * tx) * C + c;
            *reinterpret_cast<f4*>(op) = (t[a][0] + t[a][1]) + t[a][2];
            if (2 * tx + 1 < W) *reinterpret_cast<f4*>(op + C) = (t[a][1] - t[a][2]) - t[a][3];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// voxeliser: events -> two-polarity count frames (datasets/MVSEC/utils.py:215-281)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void voxelize_kernel(const double* __restrict__ ev, long long E, const double* __restrict__ start,
                                                          const double* __restrict__ end, int G, unsigned* __restrict__ counts, int H, int W)
{
    const double t0 = ev[2];
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < E; i += (long long)gridDim.x * kBlock) {
        const double* e = ev + i * 4;
        const double t = e[2] - t0;
        const long long x = (long long)e[0], y = (long long)e[1];          // int(): truncation toward zero
        if (x < 0 || x >= W || y < 0 || y >= H) continue;
        const int ch = (e[3] == 1.0) ? 0 : 1;
        int lo = 0, hi = G;                                                // first g with start[g] >= t
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (start[mid] < t) lo = mid + 1; else hi = mid; }
        for (int g = lo - 1; g >= 0 && g >= lo - 2; --g)                   // the last two frames that started before t
            if (start[g] < t && t < end[g])
                atomicAdd(&counts[(((long long)g * 2 + ch) * H + y) * W + x], 1u);
    }
}

// ---------------------------------------------------------------------------------------------------
// fused loss statistics / gradient (network/loss.py:7-24,44-75; network/metrics.py:83-95)
// ---------------------------------------------------------------------------------------------------
constexpr int kLossT = 16;                        // 16 x 16 output pixels per workgroup (256 lanes)
constexpr int kLossMaxGrid = 65535;

__device__ __forceinline__ float residual_at(const float* __restrict__ pred, const float* __restrict__ gt, int y, int x, int H, int W)
{
    if (y < 0 || y >= H || x < 0 || x >= W) return 0.f;               // zero padding of F.conv2d(..., padding=1)
    const float g = gt[y * W + x];
    return (g != g) ? 0.f : pred[y * W + x] - g;                        // NaN ground truth = invalid pixel -> residual 0
}

__global__ __launch_bounds__(256) void loss_stats_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                         double* __restrict__ partials, int H, int W, int tiles_x, int tiles_per_img)
{
    __shared__ float r[kLossT + 2][kLossT + 2];
    __shared__ float red[5][4];
    const int tile = blockIdx.x % tiles_per_img, img = blockIdx.x / tiles_per_img;
    const int y0 = (tile / tiles_x) * kLossT, x0 = (tile % tiles_x) * kLossT;
    const float* p = pred + (long long)img * H * W;
    const float* g = gt + (long long)img * H * W;
    for (int i = threadIdx.x; i < (kLossT + 2) * (kLossT + 2); i += 256) {
        const int ty = i / (kLossT + 2), tx = i % (kLossT + 2);
        r[ty][tx] = residual_at(p, g, y0 + ty - 1, x0 + tx - 1, H, W);
    }
    __syncthreads();
    const int ty = threadIdx.x / kLossT, tx = threadIdx.x % kLossT;
    const int y = y0 + ty, x = x0 + tx;
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (y < H && x < W) {
        const float gv = g[y * W + x];
        if (gv == gv) {
            const float rc = r[ty + 1][tx + 1];
            // sobelX = [[1,0,-1],[2,0,-2],[1,0,-1]], sobelY = [[1,2,1],[0,0,0],[-1,-2,-1]]  (cross-correlation)
            const float gx = (r[ty][tx] - r[ty][tx + 2]) + 2.f * (r[ty + 1][tx] - r[ty + 1][tx + 2]) + (r[ty + 2][tx] - r[ty + 2][tx + 2]);
            const float gy = (r[ty][tx] + 2.f * r[ty][tx + 1] + r[ty][tx + 2]) - (r[ty + 2][tx] + 2.f * r[ty + 2][tx + 1] + r[ty + 2][tx + 2]);
            v[0] = 1.f; v[1] = rc; v[2] = rc * rc; v[3] = fabsf(gx) + fabsf(gy); v[4] = fabsf(rc);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 5; ++q) { const float w = wave_sum_f32(v[q]); if (lane == 0) red[q][wave] = w; }
    __syncthreads();
    if (threadIdx.x < 5) {
        const int q = threadIdx.x;
        partials[(long long)blockIdx.x * 5 + q] = (double)red[q][0] + (double)red[q][1] + (double)red[q][2] + (double)red[q][3];
    }
}

__global__ __launch_bounds__(256) void loss_finish_kernel(const double* __restrict__ partials, long long n, double* __restrict__ sums)
{
    __shared__ double s[5][256];
    double acc[5] = {0, 0, 0, 0, 0};
    for (long long i = threadIdx.x; i < n; i += 256)
#pragma unroll
        for (int q = 0; q < 5; ++q) acc[q] += partials[i * 5 + q];
#pragma unroll
    for (int q = 0; q < 5; ++q) s[q][threadIdx.x] = acc[q];
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o)
#pragma unroll
            for (int q = 0; q < 5; ++q) s[q][threadIdx.x] += s[q][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 5) sums[threadIdx.x] = s[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                        const double* __restrict__ sums, const float* __restrict__ coef,
                                                        float* __restrict__ g_pred, int H, int W, int tiles_x, int tiles_per_img)
{
    __shared__ float r[kLossT + 4][kLossT + 4];          // residual, halo 2
    __shared__ float sx[kLossT + 2][kLossT + 2];         // sgn(gx) * mask, halo 1
    __shared__ float sy[kLossT + 2][kLossT + 2];
    const int tile = blockIdx.x % tiles_per_img, img = blockIdx.x / tiles_per_img;
    const int y0 = (tile / tiles_x) * kLossT, x0 = (tile % tiles_x) * kLossT;
    const float* p = pred + (long long)img * H * W;
    const float* g = gt + (long long)img * H * W;
    for (int i = threadIdx.x; i < (kLossT + 4) * (kLossT + 4); i += 256) {
        const int ty = i / (kLossT + 4), tx = i % (kLossT + 4);
        r[ty][tx] = residual_at(p, g, y0 + ty - 2, x0 + tx - 2, H, W);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (kLossT + 2) * (kLossT + 2); i += 256) {
        const int ty = i / (kLossT + 2), tx = i % (kLossT + 2);
        const int y = y0 + ty - 1, x = x0 + tx - 1;
        float vx = 0.f, vy = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const float gv = g[y * W + x];
            if (gv == gv) {
                const int a = ty, b = tx;                // r index of (y-1, x-1) is [ty][tx]
                const float gx = (r[a][b] - r[a][b + 2]) + 2.f * (r[a + 1][b] - r[a + 1][b + 2]) + (r[a + 2][b] - r[a + 2][b + 2]);
                const float gy = (r[a][b] + 2.f * r[a][b + 1] + r[a][b + 2]) - (r[a + 2][b] + 2.f * r[a + 2][b + 1] + r[a + 2][b + 2]);
                vx = (gx > 0.f) ? 1.f : (gx < 0.f ? -1.f : 0.f);
                vy = (gy > 0.f) ? 1.f : (gy < 0.f ? -1.f : 0.f);
            }
        }
        sx[ty][tx] = vx; sy[ty][tx] = vy;
    }
    __syncthreads();
    const int ty = threadIdx.x / kLossT, tx = threadIdx.x % kLossT;
    const int y = y0 + ty, x = x0 + tx;
    if (y < H && x < W) {
        const float gv = g[y * W + x];
        float out = 0.f;
        if (gv == gv) {
            const float n = (float)sums[0], s1 = (float)sums[1], c_si = coef[0], c_gm = coef[1];
            const float rc = r[ty + 2][tx + 2];
            // adjoint of the cross-correlation: T(p) = sum_{p'} s(p') K[p - p'],  p' = p - d  =>  K[d] with d in [-1,1]^2
            // sobelX[dy+1][dx+1] = {1,2,1}[dy+1] * {1,0,-1}[dx+1];  sobelY[dy+1][dx+1] = {1,0,-1}[dy+1] * {1,2,1}[dx+1]
            const int a = ty + 1, b = tx + 1;            // s index of p
            float T = 0.f;
            // p' = p - d: d = (dy,dx);  sx[a - dy][b - dx] * sobelX[dy+1][dx+1]
            T += sx[a + 1][b + 1] * 1.f + sx[a + 1][b - 1] * -1.f;      // dy = -1: row weight 1, dx = -1 -> +1, dx = +1 -> -1
            T += sx[a][b + 1] * 2.f + sx[a][b - 1] * -2.f;              // dy = 0
            T += sx[a - 1][b + 1] * 1.f + sx[a - 1][b - 1] * -1.f;      // dy = +1
            T += sy[a + 1][b + 1] * 1.f + sy[a + 1][b] * 2.f + sy[a + 1][b - 1] * 1.f;      // dy = -1: +{1,2,1}
            T += -(sy[a - 1][b + 1] * 1.f + sy[a - 1][b] * 2.f + sy[a - 1][b - 1] * 1.f);   // dy = +1: -{1,2,1}
            out = c_si * (2.f * rc / n - 2.f * s1 / (n * n)) + (c_gm / n) * T;
        }
        g_pred[(long long)img * H * W + y * W + x] = out;
    }
}

// ---------------------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------------------
inline int grid_for(long long work_items, int cap = kMaxGrid)
{
    long long g = (work_items + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

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
        const bool ok = (a.N % 16 == 0) && aligned16(a.x_seq) && aligned16(a.out_seq) && aligned16(a.v_last) &&
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

template <int K, int ACT = 0>
int launch_cl_fwd(const float* P, const int* sy, const int* sx, const float* bias, void* out, int NB, int C, int h, int w,
                         int H, int W, hipStream_t s)
{
    const bool vec = (C % 4 == 0) && aligned16(P) && aligned16(out) && (!bias || aligned16(bias));
    const long long per_img = (long long)H * W * (vec ? C / 4 : C);
    const dim3 grid((unsigned)((per_img + kBlock - 1) / kBlock), (unsigned)(NB < 65535 ? NB : 65535));
    if (vec) hipLaunchKernelGGL((upconv_cl_fwd_kernel<K, 4, ACT>), grid, dim3(kBlock), 0, s, P, sy, sx, bias, out, NB, C, h, w, H, W);
    else     hipLaunchKernelGGL((upconv_cl_fwd_kernel<K, 1, ACT>), grid, dim3(kBlock), 0, s, P, sy, sx, bias, out, NB, C, h, w, H, W);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

template <int K, int ACT = 0, int GP = 0>
int launch_cl_bwd(const void* g_out, const int* y_lo, const int* y_hi, const int* x_lo, const int* x_hi, void* g_P,
                         int NB, int C, int h, int w, int H, int W, hipStream_t s)
{
    const bool vec = (C % 4 == 0) && aligned16(g_out) && aligned16(g_P);
    const long long per_img = (long long)h * w * (vec ? C / 4 : C);
    const dim3 grid((unsigned)((per_img + kBlock - 1) / kBlock), (unsigned)(NB < 65535 ? NB : 65535));
    if (vec) hipLaunchKernelGGL((upconv_cl_bwd_kernel<K, 4, ACT, GP>), grid, dim3(kBlock), 0, s, g_out, y_lo, y_hi, x_lo, x_hi, g_P, NB, C, h, w, H, W);
    else     hipLaunchKernelGGL((upconv_cl_bwd_kernel<K, 1, ACT, GP>), grid, dim3(kBlock), 0, s, g_out, y_lo, y_hi, x_lo, x_hi, g_P, NB, C, h, w, H, W);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

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
    const bool skip = a.skip_seq != nullptr, save_h = a.h_seq != nullptr;
    const bool vec = (a.N % V == 0) && aligned16(a.x_seq) && aligned16(a.out_seq) && aligned16(a.v_last) &&
                     (!a.v_init || aligned16(a.v_init)) && (!skip || aligned16(a.skip_seq)) && (!save_h || aligned16(a.h_seq));
    const int cap = (a.nnz && !a.cnt_ws) ? kMaxGridGk : kMaxGrid;
    const int grid = vec ? grid_for(a.N / V, cap) : grid_for(a.N, cap);
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
int launch_bwd16_rc(const Bwd16Args& a, const unsigned short* x_seq, const unsigned short* g2, unsigned short* g_sum, hipStream_t s, int* grid_out)
{
    constexpr int V = (TS <= 5) ? SS_RC16_V5 : SS_RC16_V10;    // measured on the MI355X: tools/bench_rc16.py
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
int dispatch_bwd16_rc(const Bwd16Args& a, const unsigned short* x_seq, const unsigned short* g2, unsigned short* g_sum, hipStream_t s, int* grid_out)
{
    switch (a.T) {
        case 1: return launch_bwd16_rc<KIND, SG, DT, 1>(a, x_seq, g2, g_sum, s, grid_out);
        case 2: return launch_bwd16_rc<KIND, SG, DT, 2>(a, x_seq, g2, g_sum, s, grid_out);
        case 4: return launch_bwd16_rc<KIND, SG, DT, 4>(a, x_seq, g2, g_sum, s, grid_out);
        case 5: return launch_bwd16_rc<KIND, SG, DT, 5>(a, x_seq, g2, g_sum, s, grid_out);
        case 8: return launch_bwd16_rc<KIND, SG, DT, 8>(a, x_seq, g2, g_sum, s, grid_out);
        case 10: return launch_bwd16_rc<KIND, SG, DT, 10>(a, x_seq, g2, g_sum, s, grid_out);
        default: return SS_EINVAL;
    }
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

int ss_abi_version(void) { return SS_ABI_VERSION; }

long long ss_neuron_gk_ws_floats(void) { return kGkWsFloats; }

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
    if (want_gk && !g_k_ws) return SS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (N == 0) {
        if (want_gk && hipMemsetAsync(g_k, 0, sizeof(float), s) != hipSuccess) return SS_ELAUNCH;
        return SS_OK;
    }
    BwdArgs a{g_out_seq, g_v_last, h_seq, v_init, g_x_seq, g_v_init, want_gk ? g_k_ws : nullptr,
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
        hipLaunchKernelGGL(gk_finish_kernel, dim3(1), dim3(kBlock), 0, s, g_k_ws, grid, g_k);
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
                static_cast<unsigned short*>(out_seq), h_seq, v_last, nnz, T, N, scale, tau, v_th, v_reset, k, nullptr};
    hipStream_t s = static_cast<hipStream_t>(stream);
#define SS_D16(KK) (dtype == SS_DT_F16 ? dispatch_fwd16<KK, SS_DT_F16>(a, s) : dispatch_fwd16<KK, SS_DT_BF16>(a, s))
    switch (kind) {
        case SS_KIND_IF: return SS_D16(SS_KIND_IF);
        case SS_KIND_LIF: return SS_D16(SS_KIND_LIF);
        default: return SS_D16(SS_KIND_PLIF);
    }
#undef SS_D16
}

static int neuron_bwd_x16_impl(const void* g_out_seq, const void* g_out2_seq, void* g_sum_seq, const float* g_v_last, const float* h_seq, const void* x_seq, const float* v_init,
                               void* g_x_seq, float* g_v_init, float* g_k, float* g_k_ws,
                               int T, long long N, float scale, int kind, float tau, const float* k,
                               float v_th, float v_reset, int surrogate, float alpha, int detach_reset, int dtype, void* stream)
{
    if (!g_out_seq || (!h_seq && !x_seq) || !g_x_seq || T <= 0 || N < 0) return SS_EINVAL;
    if (kind < SS_KIND_IF || kind > SS_KIND_PLIF || (dtype != SS_DT_F16 && dtype != SS_DT_BF16)) return SS_EINVAL;
    if (surrogate != SS_SG_ATAN && surrogate != SS_SG_SIGMOID) return SS_EINVAL;
    if (kind == SS_KIND_PLIF && !k) return SS_EINVAL;
    if (x_seq && g_x_seq == x_seq) return SS_EINVAL;
    const bool want_gk = (kind == SS_KIND_PLIF) && g_k != nullptr;
    if (want_gk && !g_k_ws) return SS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (N == 0) {
        if (want_gk && hipMemsetAsync(g_k, 0, sizeof(float), s) != hipSuccess) return SS_ELAUNCH;
        return SS_OK;
    }
    Bwd16Args a{static_cast<const unsigned short*>(g_out_seq), g_v_last, h_seq, v_init, static_cast<unsigned short*>(g_x_seq),
                g_v_init, want_gk ? g_k_ws : nullptr, T, N, scale, tau, v_th, v_reset, alpha, k, detach_reset};
    const unsigned short* xq = static_cast<const unsigned short*>(x_seq);
    int grid = 0, rc;
    const unsigned short* g2q = static_cast<const unsigned short*>(g_out2_seq);
    unsigned short* gsq = g2q ? static_cast<unsigned short*>(g_sum_seq) : nullptr;
#define SS_B16D(KK, SGG, DTT) (xq ? dispatch_bwd16_rc<KK, SGG, DTT>(a, xq, g2q, gsq, s, &grid) : dispatch_bwd16<KK, SGG, DTT>(a, s, &grid))
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
        hipLaunchKernelGGL(gk_finish_kernel, dim3(1), dim3(kBlock), 0, s, g_k_ws, grid, g_k);
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

long long ss_neuron_cnt_ws_words(long long N)
{
    if (N < 0) return 0;
    return 2 * ((N + kBlock - 1) / kBlock) + 2;          // covers the scalar launch (one neuron per lane); the 16-B form uses a quarter
}

int ss_neuron_fwd_ex(const ss_neuron_fwd_desc* d, void* stream)
{
    if (!d || d->size != sizeof(ss_neuron_fwd_desc)) return SS_EINVAL;
    if (!d->x_seq || !d->v_last || d->T <= 0 || d->N < 0 || (!d->out_seq && !d->out_packed)) return SS_EINVAL;
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
    if (d->out_packed || d->skip_packed || !d->out_seq) return SS_EINVAL;          // packed I/O: fp32 activations only
    if (d->N == 0) return SS_OK;
    Fwd16Args a{static_cast<const unsigned short*>(d->x_seq), d->v_init, static_cast<const unsigned short*>(d->skip_seq),
                static_cast<unsigned short*>(d->out_seq), d->h_seq, d->v_last, d->nnz, d->T, d->N, d->scale, d->tau, d->v_th, d->v_reset, d->k, d->cnt_ws};
#define SS_D16(KK) (d->act_dtype == SS_DT_F16 ? dispatch_fwd16<KK, SS_DT_F16>(a, s) : dispatch_fwd16<KK, SS_DT_BF16>(a, s))
    switch (d->kind) {
        case SS_KIND_IF: return SS_D16(SS_KIND_IF);
        case SS_KIND_LIF: return SS_D16(SS_KIND_LIF);
        default: return SS_D16(SS_KIND_PLIF);
    }
#undef SS_D16
}

int ss_unpack_spikes(const unsigned int* packed, void* out, long long n, int out_dtype, int row_len, int copies, void* stream)
{
    if (!packed || !out || n < 0 || n % 16 != 0 || copies < 1 || !aligned16(out)) return SS_EINVAL;
    if (out_dtype != 0 && out_dtype != SS_DT_F16 && out_dtype != SS_DT_BF16) return SS_EINVAL;
    if (copies > 1 && (row_len <= 0 || row_len % 8 != 0 || n % row_len != 0)) return SS_EINVAL;
    if (n == 0) return SS_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long n4 = n / 4;
    const int grid = grid_for(n4, kMaxGridBwd), C = copies > 1 ? row_len : 4;
    if (out_dtype == 0) hipLaunchKernelGGL(unpack_spikes_kernel<0>, dim3(grid), dim3(kBlock), 0, s, packed, out, n4, C, copies);
    else if (out_dtype == SS_DT_F16) hipLaunchKernelGGL(unpack_spikes_kernel<SS_DT_F16>, dim3(grid), dim3(kBlock), 0, s, packed, out, n4, C, copies);
    else hipLaunchKernelGGL(unpack_spikes_kernel<SS_DT_BF16>, dim3(grid), dim3(kBlock), 0, s, packed, out, n4, C, copies);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_im2col_cl_bf16_packed(const unsigned int* x_packed, void* A, long long NB, int h, int w, int C, int k, int stride, int pad, int ho, int wo, void* stream)
{
    if (!x_packed || !A || NB <= 0 || h <= 0 || w <= 0 || C <= 0 || C % 8 != 0 || k <= 0 || stride <= 0 || pad < 0 || ho <= 0 || wo <= 0) return SS_EINVAL;
    if ((ho - 1) * stride - pad + k - 1 >= h + pad || (wo - 1) * stride - pad + k - 1 >= w + pad) return SS_EINVAL;
    if (!aligned16(A) || (NB * h * w * C) % 16 != 0) return SS_EINVAL;
    const long long rows = NB * ho * wo;
    const long long per_row = (long long)k * k * (C / 8);
    if (rows > 0x7fffffffLL || per_row > 65535LL * kBlock) return SS_EINVAL;
    hipLaunchKernelGGL(im2col_cl_bf16_packed_kernel, dim3((unsigned)rows, (unsigned)((per_row + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), x_packed, static_cast<unsigned short*>(A), h, w, C, k, stride, pad, ho, wo);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_upconv_fused_supported(int Cin, int Cout, int k)
{
    return k == 5 && ((Cin == 64 && Cout == 32) || (Cin == 128 && Cout == 64));
}

long long ss_upconv_fused_wf_elems(int Cin, int Cout)
{
    if (Cin <= 0 || Cout <= 0 || Cin % 16 || Cout % kFusCC) return 0;
    return (long long)(Cout / kFusCC) * (3 * Cin / 16) * kFusNT * 64 * 8;
}

int ss_upconv_fused_prep_w(const float* W, void* Wf, int Cin, int Cout, void* stream)
{
    if (!W || !Wf || !ss_upconv_fused_wf_elems(Cin, Cout) || !aligned16(Wf)) return SS_EINVAL;
    hipLaunchKernelGGL(upconv_fused_prep_w_kernel, dim3(grid_for(ss_upconv_fused_wf_elems(Cin, Cout) / 8, 4096)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), W, static_cast<unsigned short*>(Wf), Cin, Cout);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_upconv_fused_fwd_f32(const float* x, const unsigned int* x_packed, const void* Wf, const int* src_y, const int* src_x, float* out,
                            long long NB, int Cin, int Cout, int h, int w, int H, int W, int max_window, void* stream)
{
    if ((!x && !x_packed) || !Wf || !src_y || !src_x || !out || NB < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SS_EINVAL;
    if (!ss_upconv_fused_supported(Cin, Cout, 5) || max_window <= 0 || max_window > kFusS) return SS_EINVAL;
    if (!aligned16(Wf) || !aligned16(out) || (x && !aligned16(x))) return SS_EINVAL;
    if (x_packed && (NB * h * w * Cin) % 16 != 0) return SS_EINVAL;
    if (NB == 0) return SS_OK;
    const int tx = (W + kFusT - 1) / kFusT, ty = (H + kFusT - 1) / kFusT;
    const long long blocks = NB * tx * ty;
    if (blocks > 0x7fffffffLL || NB * H * W * (long long)Cout > 0x7fffffffffLL) return SS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned short* wf = static_cast<const unsigned short*>(Wf);
#define SS_FUS(CI, CO) do { if (x_packed) hipLaunchKernelGGL((upconv_fused_fwd_kernel<CI, CO, true>), dim3((unsigned)blocks), dim3(kBlock), 0, s, \
                                                             static_cast<const void*>(x_packed), wf, src_y, src_x, out, h, w, H, W, tx, ty); \
                            else hipLaunchKernelGGL((upconv_fused_fwd_kernel<CI, CO, false>), dim3((unsigned)blocks), dim3(kBlock), 0, s, \
                                                    static_cast<const void*>(x), wf, src_y, src_x, out, h, w, H, W, tx, ty); } while (0)
    if (Cin == 64) SS_FUS(64, 32); else SS_FUS(128, 64);
#undef SS_FUS
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

#if SS_F2_TRACE
int ss_debug_f2_trace(unsigned long long* host_dst)
{
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(f2_trace), sizeof(unsigned long long) * 2 * 64 * 4) == hipSuccess ? SS_OK : SS_ELAUNCH;
}
#endif

int ss_upconv_fused2_supported(int Cin, int Cout, int k, int max_window)
{
    return ss_upconv_fused_supported(Cin, Cout, k) && max_window > 0 && max_window <= kF2Rows;
}

long long ss_upconv_fused2_wf_elems(int Cin, int Cout)
{
    if (Cin <= 0 || Cout <= 0 || Cin % 16 || Cout % 32) return 0;
    return 25LL * Cout * 3 * Cin;
}

static int fused2_prep(const float* W, void* Wf, int Cin, int Cout, int nsplit, void* stream)
{
    if (!W || !Wf || !ss_upconv_fused2_wf_elems(Cin, Cout) || !aligned16(Wf) || (nsplit != 1 && nsplit != 3)) return SS_EINVAL;
    hipLaunchKernelGGL(upconv_fused2_prep_w_kernel, dim3(grid_for(ss_upconv_fused2_wf_elems(Cin, Cout) / 24 * nsplit, 4096)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), W, static_cast<unsigned short*>(Wf), Cin, Cout, nsplit);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_upconv_fused2_prep_w(const float* W, void* Wf, int Cin, int Cout, void* stream) { return fused2_prep(W, Wf, Cin, Cout, 3, stream); }
int ss_upconv_fused2_prep_w_x16(const float* W, void* Wf, int Cin, int Cout, int nsplit, void* stream) { return fused2_prep(W, Wf, Cin, Cout, nsplit, stream); }

// xin_kind: 0 fp32, kF2Packed packed, SS_DT_F16 / SS_DT_BF16;  out_dt: 0 fp32, SS_DT_F16 / SS_DT_BF16;  nsplit 3 / 1
static int fused2_launch(const void* xin, int xin_kind, const void* Wf, const int* src_y, const int* src_x, void* out, int out_dt, int nsplit,
                         long long NB, int Cin, int Cout, int h, int w, int H, int W, int max_window, void* stream)
{
    if (!xin || !Wf || !src_y || !src_x || !out || NB < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SS_EINVAL;
    if (!ss_upconv_fused2_supported(Cin, Cout, 5, max_window)) return SS_EINVAL;
    if (!aligned16(Wf) || !aligned16(out) || (xin_kind != kF2Packed && !aligned16(xin))) return SS_EINVAL;
    if (xin_kind == kF2Packed && (NB * h * w * Cin) % 16 != 0) return SS_EINVAL;
    if (NB == 0) return SS_OK;
    const int tx = (W + kFusT - 1) / kFusT, ty = (H + kFusT - 1) / kFusT;
    const long long tiles = NB * tx * ty;
    if (tiles > 0x7fffffffLL || NB * H * W * (long long)Cout > 0x7fffffffffLL) return SS_EINVAL;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
        return SS_ELAUNCH;
    // persistent workgroups: one per CU (the LDS footprint admits no more), a multiple of 8 so that workgroup b stays on XCD b % 8
    const long long per_xcd = (tiles + 7) / 8;
    const unsigned grid = 8u * (unsigned)(per_xcd < cus / 8 ? per_xcd : cus / 8);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned short* wf = static_cast<const unsigned short*>(Wf);
#define SS_FUS2(CI, CO, XI, OD, NS) hipLaunchKernelGGL((upconv_fused2_fwd_kernel<CI, CO, XI, OD, NS>), dim3(grid), dim3(kF2Threads), 0, s, \
                                                       xin, wf, src_y, src_x, out, h, w, H, W, tx, ty, (int)tiles)
#define SS_FUS2_SHAPES(XI, OD, NS) do { if (Cin == 64) SS_FUS2(64, 32, XI, OD, NS); else SS_FUS2(128, 64, XI, OD, NS); } while (0)
    if (xin_kind == 0 && out_dt == 0 && nsplit == 3) SS_FUS2_SHAPES(0, 0, 3);
    else if (xin_kind == kF2Packed && out_dt == 0 && nsplit == 3) SS_FUS2_SHAPES(kF2Packed, 0, 3);
    else if (xin_kind == SS_DT_F16 && out_dt == SS_DT_F16 && nsplit == 3) SS_FUS2_SHAPES(SS_DT_F16, SS_DT_F16, 3);
    else if (xin_kind == SS_DT_BF16 && out_dt == SS_DT_BF16 && nsplit == 1) SS_FUS2_SHAPES(SS_DT_BF16, SS_DT_BF16, 1);
    else return SS_EINVAL;
#undef SS_FUS2_SHAPES
#undef SS_FUS2
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_upconv_fused2_fwd_f32(const float* x, const unsigned int* x_packed, const void* Wf, const int* src_y, const int* src_x, float* out,
                             long long NB, int Cin, int Cout, int h, int w, int H, int W, int max_window, void* stream)
{
    if (!x && !x_packed) return SS_EINVAL;
    return fused2_launch(x_packed ? static_cast<const void*>(x_packed) : static_cast<const void*>(x), x_packed ? kF2Packed : 0, Wf, src_y, src_x, out, 0, 3,
                         NB, Cin, Cout, h, w, H, W, max_window, stream);
}

int ss_upconv_fused2_fwd_x16(const void* x, int dtype, const void* Wf, int nsplit, const int* src_y, const int* src_x, void* out,
                             long long NB, int Cin, int Cout, int h, int w, int H, int W, int max_window, void* stream)
{
    if (dtype != SS_DT_F16 && dtype != SS_DT_BF16) return SS_EINVAL;
    return fused2_launch(x, dtype, Wf, src_y, src_x, out, dtype, nsplit, NB, Cin, Cout, h, w, H, W, max_window, stream);
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

int ss_upconv1_fwd_f32(const float* P, const int* src_y, const int* src_x, const float* bias, float* out,
                       long long NB, int k, int h, int w, int H, int W, void* stream)
{
    if (!P || !src_y || !src_x || !out || NB < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SS_EINVAL;
    if (k != 1 && k != 3 && k != 5) return SS_EINVAL;
    if (NB == 0) return SS_OK;
    if (NB > 0x7fffffffLL || (long long)H * W > 0x7fffffffLL || (long long)k * k * h * w > 0x7fffffffLL) return SS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((H * W + kBlock - 1) / kBlock), (unsigned)(NB < 65535 ? NB : 65535));
    const int nb = (int)NB;
    if (k == 1) hipLaunchKernelGGL(upconv1_fwd_kernel<1>, grid, dim3(kBlock), 0, s, P, src_y, src_x, bias, out, nb, h, w, H, W);
    else if (k == 3) hipLaunchKernelGGL(upconv1_fwd_kernel<3>, grid, dim3(kBlock), 0, s, P, src_y, src_x, bias, out, nb, h, w, H, W);
    else hipLaunchKernelGGL(upconv1_fwd_kernel<5>, grid, dim3(kBlock), 0, s, P, src_y, src_x, bias, out, nb, h, w, H, W);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_upconv1_bwd_f32(const float* g_out, const int* y_lo, const int* y_hi, const int* x_lo, const int* x_hi,
                       float* g_P, long long NB, int k, int h, int w, int H, int W, void* stream)
{
    if (!g_out || !y_lo || !y_hi || !x_lo || !x_hi || !g_P || NB < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SS_EINVAL;
    if (k != 1 && k != 3 && k != 5) return SS_EINVAL;
    if (NB == 0) return SS_OK;
    if (NB > 0x7fffffffLL || (long long)H * W > 0x7fffffffLL || (long long)k * k * h * w > 0x7fffffffLL) return SS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((w + kBwdTileX - 1) / kBwdTileX), (unsigned)((h + kBwdTileY - 1) / kBwdTileY),
                    (unsigned)(NB < 65535 ? NB : 65535));
    if (grid.y > 65535u) return SS_EINVAL;
    const int nb = (int)NB;
    if (k == 1) hipLaunchKernelGGL(upconv1_bwd_kernel<1>, grid, dim3(kBlock), 0, s, g_out, y_lo, y_hi, x_lo, x_hi, g_P, nb, h, w, H, W);
    else if (k == 3) hipLaunchKernelGGL(upconv1_bwd_kernel<3>, grid, dim3(kBlock), 0, s, g_out, y_lo, y_hi, x_lo, x_hi, g_P, nb, h, w, H, W);
    else hipLaunchKernelGGL(upconv1_bwd_kernel<5>, grid, dim3(kBlock), 0, s, g_out, y_lo, y_hi, x_lo, x_hi, g_P, nb, h, w, H, W);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_upconv_cl_fwd_f32(const float* P, const int* src_y, const int* src_x, const float* bias, float* out,
                         long long NB, int k, int C, int h, int w, int H, int W, void* stream)
{
    if (!P || !src_y || !src_x || !out || NB < 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SS_EINVAL;
    if (k != 1 && k != 3 && k != 5) return SS_EINVAL;
    if (NB > 0x7fffffffLL || (long long)H * W * C > 0x7fffffffLL || (long long)k * k * C * h * w > 0x7fffffffLL) return SS_EINVAL;
    if (NB == 0) return SS_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (k == 1) return launch_cl_fwd<1>(P, src_y, src_x, bias, out, (int)NB, C, h, w, H, W, s);
    if (k == 3) return launch_cl_fwd<3>(P, src_y, src_x, bias, out, (int)NB, C, h, w, H, W, s);
    return launch_cl_fwd<5>(P, src_y, src_x, bias, out, (int)NB, C, h, w, H, W, s);
}

int ss_upconv_cl_bwd_f32(const float* g_out, const int* y_lo, const int* y_hi, const int* x_lo, const int* x_hi,
                         float* g_P, long long NB, int k, int C, int h, int w, int H, int W, void* stream)
{
    if (!g_out || !y_lo || !y_hi || !x_lo || !x_hi || !g_P || NB < 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SS_EINVAL;
    if (k != 1 && k != 3 && k != 5) return SS_EINVAL;
    if (NB > 0x7fffffffLL || (long long)H * W * C > 0x7fffffffLL || (long long)k * k * C * h * w > 0x7fffffffLL) return SS_EINVAL;
    if (NB == 0) return SS_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (k == 1) return launch_cl_bwd<1>(g_out, y_lo, y_hi, x_lo, x_hi, g_P, (int)NB, C, h, w, H, W, s);
    if (k == 3) return launch_cl_bwd<3>(g_out, y_lo, y_hi, x_lo, x_hi, g_P, (int)NB, C, h, w, H, W, s);
    return launch_cl_bwd<5>(g_out, y_lo, y_hi, x_lo, x_hi, g_P, (int)NB, C, h, w, H, W, s);
}

int ss_upconv_cl_fwd_x16(const float* P, const int* src_y, const int* src_x, const float* bias, void* out,
                         long long NB, int k, int C, int h, int w, int H, int W, int dtype, void* stream)
{
    if (!P || !src_y || !src_x || !out || NB < 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SS_EINVAL;
    if (k != 5 || (dtype != SS_DT_F16 && dtype != SS_DT_BF16)) return SS_EINVAL;            // decoder stages only (k = 5)
    if (NB > 0x7fffffffLL || (long long)H * W * C > 0x7fffffffLL || (long long)k * k * C * h * w > 0x7fffffffLL) return SS_EINVAL;
    if (NB == 0) return SS_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dtype == SS_DT_F16 ? launch_cl_fwd<5, SS_DT_F16>(P, src_y, src_x, bias, out, (int)NB, C, h, w, H, W, s)
                              : launch_cl_fwd<5, SS_DT_BF16>(P, src_y, src_x, bias, out, (int)NB, C, h, w, H, W, s);
}

int ss_upconv_cl_bwd_x16(const void* g_out, const int* y_lo, const int* y_hi, const int* x_lo, const int* x_hi,
                         float* g_P, long long NB, int k, int C, int h, int w, int H, int W, int dtype, void* stream)
{
    if (!g_out || !y_lo || !y_hi || !x_lo || !x_hi || !g_P || NB < 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SS_EINVAL;
    if (k != 5 || (dtype != SS_DT_F16 && dtype != SS_DT_BF16)) return SS_EINVAL;
    if (NB > 0x7fffffffLL || (long long)H * W * C > 0x7fffffffLL || (long long)k * k * C * h * w > 0x7fffffffLL) return SS_EINVAL;
    if (NB == 0) return SS_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dtype == SS_DT_F16 ? launch_cl_bwd<5, SS_DT_F16>(g_out, y_lo, y_hi, x_lo, x_hi, g_P, (int)NB, C, h, w, H, W, s)
                              : launch_cl_bwd<5, SS_DT_BF16>(g_out, y_lo, y_hi, x_lo, x_hi, g_P, (int)NB, C, h, w, H, W, s);
}

int ss_upconv_cl_bwd_lowp(const void* g_out, int g_dtype, const int* y_lo, const int* y_hi, const int* x_lo, const int* x_hi,
                          void* g_P_bf16, long long NB, int k, int C, int h, int w, int H, int W, void* stream)
{
    if (!g_out || !y_lo || !y_hi || !x_lo || !x_hi || !g_P_bf16 || NB < 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SS_EINVAL;
    if (k != 5 || (g_dtype != 0 && g_dtype != SS_DT_F16 && g_dtype != SS_DT_BF16)) return SS_EINVAL;
    if (NB > 0x7fffffffLL || (long long)H * W * C > 0x7fffffffLL || (long long)k * k * C * h * w > 0x7fffffffLL) return SS_EINVAL;
    if (NB == 0) return SS_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (g_dtype == 0) return launch_cl_bwd<5, 0, SS_DT_BF16>(g_out, y_lo, y_hi, x_lo, x_hi, g_P_bf16, (int)NB, C, h, w, H, W, s);
    if (g_dtype == SS_DT_F16) return launch_cl_bwd<5, SS_DT_F16, SS_DT_BF16>(g_out, y_lo, y_hi, x_lo, x_hi, g_P_bf16, (int)NB, C, h, w, H, W, s);
    return launch_cl_bwd<5, SS_DT_BF16, SS_DT_BF16>(g_out, y_lo, y_hi, x_lo, x_hi, g_P_bf16, (int)NB, C, h, w, H, W, s);
}

int ss_voxelize_f64(const double* events, long long E, const double* start, const double* end, int G,
                    unsigned int* counts, int H, int W, void* stream)
{
    if (!start || !end || !counts || E < 0 || G <= 0 || H <= 0 || W <= 0 || (E > 0 && !events)) return SS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(counts, 0, sizeof(unsigned) * (size_t)G * 2 * H * W, s) != hipSuccess) return SS_ELAUNCH;
    if (E == 0) return SS_OK;
    hipLaunchKernelGGL(voxelize_kernel, dim3(grid_for(E)), dim3(kBlock), 0, s, events, E, start, end, G, counts, H, W);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_im2col_cl_bf16(const float* x, void* A, long long NB, int h, int w, int C, int k, int stride, int pad, int ho, int wo, void* stream)
{
    if (!x || !A || NB <= 0 || h <= 0 || w <= 0 || C <= 0 || C % 8 != 0 || k <= 0 || stride <= 0 || pad < 0 || ho <= 0 || wo <= 0) return SS_EINVAL;
    if ((ho - 1) * stride - pad + k - 1 >= h + pad || (wo - 1) * stride - pad + k - 1 >= w + pad) return SS_EINVAL;
    if (!aligned16(x) || !aligned16(A)) return SS_EINVAL;
    const long long rows = NB * ho * wo;
    const long long per_row = (long long)k * k * (C / 8);
    if (rows > 0x7fffffffLL || per_row > 65535LL * kBlock) return SS_EINVAL;
    hipLaunchKernelGGL(im2col_cl_bf16_kernel, dim3((unsigned)rows, (unsigned)((per_row + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), x, static_cast<unsigned short*>(A), h, w, C, k, stride, pad, ho, wo);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_split3_bf16(const float* g, void* g3, long long M, int N, void* stream)
{
    if (!g || !g3 || M <= 0 || N <= 0 || N % 4 != 0 || !aligned16(g) || !aligned16(g3)) return SS_EINVAL;
    hipLaunchKernelGGL(split3_bf16_kernel, dim3(grid_for(M * (N / 4), kMaxGridBwd)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       g, static_cast<unsigned short*>(g3), M, N);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_spike_wgrad_supported(int Cin, int N)
{
    return (Cin == 64 || Cin == 128 || Cin == 256 || Cin == 512) && N > 0 && N % 32 == 0 && N / 32 <= 256;
}

static int spike_wgrad_plan(int Cin, int N, int* Q, int* slices)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return 0;
    // accumulators per wavefront: NTW x CIT x 16 registers: C_in 64 -> 2 column tiles x 2 channel tiles, 128 -> 1 x 4, 256 -> 1 x 8,
    // 512 -> 1 x 8 in two channel groups
    const int ntw = Cin == 64 ? 2 : 1;
    const int cg = Cin == 512 ? 2 : 1;
    const int nt = N / 32;
    *Q = (nt + 8 * ntw - 1) / (8 * ntw);
    *slices = cus / (*Q * cg) > 0 ? cus / (*Q * cg) : 1;
    return 1;
}

long long ss_spike_wgrad_ws_floats(int Cin, int N, long long R)
{
    int Q = 0, slices = 0;
    if (R <= 0 || !ss_spike_wgrad_supported(Cin, N) || !spike_wgrad_plan(Cin, N, &Q, &slices)) return 0;
    return (long long)slices * N * Cin + ((R + 15) / 16) * Cin * 8;           // split-K partials + the bf16 fragment-order copy of x
}

int ss_spike_wgrad_f32(const float* g, const float* x, float* g_w, float* ws, long long R, int Cin, int N, int accumulate, void* stream)
{
    if (!g || !x || !g_w || !ws || R <= 0 || !ss_spike_wgrad_supported(Cin, N) || !aligned16(ws)) return SS_EINVAL;
    int Q = 0, slices = 0;
    if (!spike_wgrad_plan(Cin, N, &Q, &slices)) return SS_ELAUNCH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned short* xT = reinterpret_cast<unsigned short*>(ws + (long long)slices * N * Cin);
    hipLaunchKernelGGL(spike_wgrad_xprep_kernel, dim3(grid_for(((R + 15) / 16) * Cin, kMaxGridBwd)), dim3(kBlock), 0, s, x, xT, R, Cin);
    const unsigned grid = (unsigned)(Q * slices * (Cin == 512 ? 2 : 1));
    if (Cin == 64) hipLaunchKernelGGL((spike_wgrad_kernel<2, 2, 3>), dim3(grid), dim3(kSwThreads), 0, s, g, xT, ws, R, N, Q, Cin);
    else if (Cin == 128) hipLaunchKernelGGL((spike_wgrad_kernel<4, 1, 3>), dim3(grid), dim3(kSwThreads), 0, s, g, xT, ws, R, N, Q, Cin);
    else hipLaunchKernelGGL((spike_wgrad_kernel<8, 1, 2>), dim3(grid), dim3(kSwThreads), 0, s, g, xT, ws, R, N, Q, Cin);
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    hipLaunchKernelGGL(spike_wgrad_reduce_kernel, dim3(grid_for((long long)N * Cin, 1024)), dim3(kBlock), 0, s, ws, g_w, slices, N, Cin, accumulate);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

// TR source rows per tile, window capacity (rows x columns of output pixels) per compiled shape
static void upconv_bwd_fused_shape(int Cin, int* TR, int* WRM, int* WCM, int* ntw)
{
    if (Cin == 64) { *TR = 4; *WRM = 14; *WCM = 40; *ntw = 2; } else { *TR = 2; *WRM = 10; *WCM = 40; *ntw = 2; }
}

int ss_upconv_bwd_fused_supported(int Cin, int Cout, int k, int max_rows4, int max_rows2, int max_cols16, int max_span)
{
    // max_rows4 / max_rows2: largest output-row span (incl. the k - 1 taps) of 4 / 2 consecutive source rows; max_cols16: the same for 16
    // consecutive source columns; max_span: most output rows / columns one source pixel collects per tap — computed by the caller from the tables
    if (!ss_upconv_fused_supported(Cin, Cout, k) || max_span < 1 || max_span > 3) return 0;
    int TR, WRM, WCM, ntw;
    upconv_bwd_fused_shape(Cin, &TR, &WRM, &WCM, &ntw);
    return (TR == 4 ? max_rows4 : max_rows2) <= WRM && max_cols16 <= WCM && max_rows2 > 0 && max_rows4 > 0 && max_cols16 > 0;
}

static int upconv_bwd_fused_plan(int Cin, int Cout, int* Q, int* slices)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return 0;
    int TR, WRM, WCM, ntw;
    upconv_bwd_fused_shape(Cin, &TR, &WRM, &WCM, &ntw);
    const int nt = 25 * Cout / 32;
    *Q = (nt + 8 * ntw - 1) / (8 * ntw);
    const int wgs = Cin == 64 ? 2 * cus : cus;            // C_in 64: 118 registers, 67 KiB LDS -> two workgroups per CU (window loads overlap compute)
    *slices = wgs / *Q > 0 ? wgs / *Q : 1;
    return 1;
}

long long ss_upconv_bwd_fused_ws_floats(int Cin, int Cout, long long NB, int h, int w)
{
    int Q = 0, slices = 0;
    if (!ss_upconv_fused_supported(Cin, Cout, 5) || NB <= 0 || h <= 0 || w <= 0 || !upconv_bwd_fused_plan(Cin, Cout, &Q, &slices)) return 0;
    return (long long)slices * 25 * Cout * Cin + NB * h * ((w + 15) / 16) * Cin * 8;
}

int ss_upconv_bwd_fused_f32(const float* g_out, const float* x, const int* y_lo, const int* y_hi, const int* x_lo, const int* x_hi,
                            float* g_P, float* g_w, float* ws, long long NB, int Cin, int Cout, int h, int w, int H, int W,
                            int accumulate, void* stream)
{
    if (!g_out || !x || !y_lo || !y_hi || !x_lo || !x_hi || !g_P || !g_w || !ws || NB <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SS_EINVAL;
    if (!ss_upconv_fused_supported(Cin, Cout, 5) || !aligned16(g_out) || !aligned16(ws) || NB * h * (long long)w > 0x7fffffffLL) return SS_EINVAL;
    int Q = 0, slices = 0;
    if (!upconv_bwd_fused_plan(Cin, Cout, &Q, &slices)) return SS_ELAUNCH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int N = 25 * Cout;
    unsigned short* xT = reinterpret_cast<unsigned short*>(ws + (long long)slices * N * Cin);
    hipLaunchKernelGGL(upconv_bwd_xprep_kernel, dim3(grid_for(NB * h * ((w + 15) / 16) * Cin, kMaxGridBwd)), dim3(kBlock), 0, s, x, xT, NB * h, w, Cin);
    const unsigned grid = (unsigned)(Q * slices);
    if (Cin == 64) hipLaunchKernelGGL((upconv_bwd_fused_kernel<64, 32, 4, 2, 14, 40>), dim3(grid), dim3(kSwThreads), 0, s, g_out, xT, y_lo, y_hi, x_lo, x_hi,
                                      g_P, ws, (int)NB, h, w, H, W, Q);
    else hipLaunchKernelGGL((upconv_bwd_fused_kernel<128, 64, 2, 2, 10, 40>), dim3(grid), dim3(kSwThreads), 0, s, g_out, xT, y_lo, y_hi, x_lo, x_hi,
                            g_P, ws, (int)NB, h, w, H, W, Q);
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    hipLaunchKernelGGL(spike_wgrad_reduce_kernel, dim3(grid_for((long long)N * Cin, 1024)), dim3(kBlock), 0, s, ws, g_w, slices, N, Cin, accumulate);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_gemm6_supported(int K, int N)
{
    return K > 0 && K % 16 == 0 && (N == 64 || N == 128 || N == 256 || N == 512);
}

long long ss_gemm6_ws_floats(int K, int N)
{
    return ss_gemm6_supported(K, N) ? (long long)K * N * 3 / 2 : 0;       // the 3 bf16 terms of B in fragment order (per batch entry)
}

int ss_gemm6_batched_f32(const float* A, const float* B, float* C, float* ws, int batch, long long R, int K, int N, void* stream)
{
    if (!A || !B || !C || !ws || R <= 0 || batch <= 0 || batch > 65535 || !ss_gemm6_supported(K, N) || !aligned16(A) || !aligned16(ws) || (K % 4) != 0 ||
        ((R * K) % 4) != 0) return SS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned short* Bf = reinterpret_cast<unsigned short*>(ws);
    hipLaunchKernelGGL(gemm6_prep_b_kernel, dim3(grid_for((long long)K / 16 * 3 * (N / 32) * 64, 4096), batch), dim3(kBlock), 0, s, B, Bf, K, N);
    const long long mt = (R + 255) / 256;
    const int kinds = N == 512 ? 2 : 1;
    if (mt * kinds > 0x7fffffffLL) return SS_EINVAL;
    const dim3 grid((unsigned)(mt * kinds), batch);
    if (N == 64) hipLaunchKernelGGL((gemm6_kernel<2, 8>), grid, dim3(kG6Threads), 0, s, A, Bf, C, R, K, N, kinds);
    else if (N == 128) hipLaunchKernelGGL((gemm6_kernel<4, 4>), grid, dim3(kG6Threads), 0, s, A, Bf, C, R, K, N, kinds);
    else hipLaunchKernelGGL((gemm6_kernel<8, 2>), grid, dim3(kG6Threads), 0, s, A, Bf, C, R, K, N, kinds);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_gemm6_f32(const float* A, const float* B, float* C, float* ws, long long R, int K, int N, void* stream)
{
    return ss_gemm6_batched_f32(A, B, C, ws, 1, R, K, N, stream);
}

int ss_spike_conv_wgrad_supported(int Cin, int Cout, int k, int stride, int pad)
{
    return k == 5 && stride == 2 && pad == 2 && ((Cin == 32 && Cout == 64) || (Cin == 64 && Cout == 128));
}

static int spike_conv_wgrad_plan(int Cin, int* Q, int* slices)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return 0;
    // a workgroup covers (8 / C_out tiles) groups x 7 virtual-channel tiles: C_in 32 (25 tiles, 2 C_out tiles): 28 -> one kind; C_in 64 (50, 4): 14 -> 4 kinds
    const int nv = 25 * Cin / 32, cov = Cin == 32 ? 28 : 14;
    *Q = (nv + cov - 1) / cov;
    *slices = cus / *Q > 0 ? cus / *Q : 1;
    return 1;
}

long long ss_spike_conv_wgrad_ws_floats(int Cin, int Cout, long long NB, int h, int w)
{
    int Q = 0, slices = 0;
    if (!ss_spike_conv_wgrad_supported(Cin, Cout, 5, 2, 2) || NB <= 0 || h <= 0 || w <= 0 || !spike_conv_wgrad_plan(Cin, &Q, &slices)) return 0;
    const int ho = (h + 4 - 5) / 2 + 1, wo = (w + 4 - 5) / 2 + 1;
    const long long ksr = (wo + 15) / 16, oxp = ksr * 16;
    return (long long)slices * 25 * Cin * Cout + (5LL * NB * (h + 4) * Cin * oxp + 1) / 2 + NB * ho * ksr * (Cout / 32) * 768 + 8;
}

int ss_spike_conv_wgrad_f32(const float* g, const float* x, const unsigned int* x_packed, float* g_w, float* ws, long long NB, int Cin, int Cout, int h,
                            int w, int accumulate, void* stream)
{
    if (x_packed && (NB * h * w * Cin) % 16 != 0) return SS_EINVAL;
    if (!g || (!x && !x_packed) || !g_w || !ws || NB <= 0 || NB > 0x7fffffff || h <= 0 || w <= 0 || !ss_spike_conv_wgrad_supported(Cin, Cout, 5, 2, 2) || !aligned16(ws))
        return SS_EINVAL;
    int Q = 0, slices = 0;
    if (!spike_conv_wgrad_plan(Cin, &Q, &slices)) return SS_ELAUNCH;
    const int ho = (h + 4 - 5) / 2 + 1, wo = (w + 4 - 5) / 2 + 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long ksr = (wo + 15) / 16, oxp = ksr * 16;
    const long long part = (long long)slices * 25 * Cin * Cout, xk = ((5LL * NB * (h + 4) * Cin * oxp + 1) / 2 + 3) & ~3LL;
    unsigned short* xK = reinterpret_cast<unsigned short*>(ws + part);
    unsigned short* gT = reinterpret_cast<unsigned short*>(ws + part + xk);
    if (x_packed) hipLaunchKernelGGL(spike_conv_xprep_kernel<true>, dim3(grid_for(NB * (h + 4) * (oxp / 8) * Cin, kMaxGridBwd)), dim3(kBlock), 0, s,
                                     static_cast<const void*>(x_packed), xK, (int)NB, h, w, Cin, wo);
    else hipLaunchKernelGGL(spike_conv_xprep_kernel<false>, dim3(grid_for(NB * (h + 4) * (oxp / 8) * Cin, kMaxGridBwd)), dim3(kBlock), 0, s,
                            static_cast<const void*>(x), xK, (int)NB, h, w, Cin, wo);
    hipLaunchKernelGGL(spike_conv_gprep_kernel, dim3(grid_for(NB * ho * ksr * (Cout / 32) * 64, kMaxGridBwd)), dim3(kBlock), 0, s, g, gT, NB * ho, wo, Cout);
    const unsigned grid = (unsigned)(Q * slices);
    if (Cin == 32) hipLaunchKernelGGL((spike_conv_wgrad_kernel<1, 2, 7>), dim3(grid), dim3(kSwThreads), 0, s, gT, xK, ws, (int)NB, h, ho, wo, Q);
    else hipLaunchKernelGGL((spike_conv_wgrad_kernel<2, 4, 7>), dim3(grid), dim3(kSwThreads), 0, s, gT, xK, ws, (int)NB, h, ho, wo, Q);
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    hipLaunchKernelGGL(spike_conv_wgrad_reduce_kernel, dim3(grid_for(25LL * Cin * Cout, 1024)), dim3(kBlock), 0, s, ws, g_w, slices, Cin, Cout, accumulate);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_wino_dgrad_weights_f32(const float* W, float* U, int Cout, int Cin, void* stream)
{
    if (!W || !U || Cout <= 0 || Cin <= 0) return SS_EINVAL;
    hipLaunchKernelGGL(wino_dgrad_weights_kernel, dim3(grid_for((long long)Cout * Cin, 4096)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       W, U, Cout, Cin);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_wino_dgrad_input_f32(const float* g, float* V, long long NB, int H, int W, int C, void* stream)
{
    if (!g || !V || NB < 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0 || !aligned16(g) || !aligned16(V)) return SS_EINVAL;
    if (NB == 0) return SS_OK;
    const int th = (H + 1) / 2, tw = (W + 1) / 2;
    const long long T = NB * th * tw;
    hipLaunchKernelGGL(wino_dgrad_input_kernel, dim3(grid_for(T * (C / 4), kMaxGridBwd)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       g, V, T, H, W, C, th, tw);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_wino_dgrad_output_f32(const float* M, float* g_in, long long NB, int H, int W, int C, void* stream)
{
    if (!M || !g_in || NB < 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0 || !aligned16(M) || !aligned16(g_in)) return SS_EINVAL;
    if (NB == 0) return SS_OK;
    const int th = (H + 1) / 2, tw = (W + 1) / 2;
    const long long T = NB * th * tw;
    hipLaunchKernelGGL(wino_dgrad_output_kernel, dim3(grid_for(T * (C / 4), kMaxGridBwd)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       M, g_in, T, H, W, C, th, tw);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

long long ss_loss_ws_doubles(void) { return (long long)kLossMaxGrid * 5; }

int ss_loss_stats_f32(const float* pred, const float* gt, double* sums, double* ws, long long B, int H, int W, void* stream)
{
    if (!pred || !gt || !sums || !ws || B <= 0 || H <= 0 || W <= 0) return SS_EINVAL;
    const int tx = (W + kLossT - 1) / kLossT, ty = (H + kLossT - 1) / kLossT;
    const long long blocks = B * tx * ty;
    if (blocks > kLossMaxGrid || (long long)H * W > 0x7fffffffLL) return SS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(loss_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pred, gt, ws, H, W, tx, tx * ty);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, s, ws, blocks, sums);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_loss_grad_f32(const float* pred, const float* gt, const double* sums, const float* coef, float* g_pred,
                     long long B, int H, int W, void* stream)
{
    if (!pred || !gt || !sums || !coef || !g_pred || B <= 0 || H <= 0 || W <= 0) return SS_EINVAL;
    const int tx = (W + kLossT - 1) / kLossT, ty = (H + kLossT - 1) / kLossT;
    const long long blocks = B * tx * ty;
    if (blocks > 0x7fffffffLL || (long long)H * W > 0x7fffffffLL) return SS_EINVAL;
    hipLaunchKernelGGL(loss_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), pred, gt, sums,
                       coef, g_pred, H, W, tx, tx * ty);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

}  // extern "C"

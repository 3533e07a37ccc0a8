// ss_neuron16_v2.hpp — round-6 forms of the fused neuron kernels on 16-bit activations (fp16 / bf16 in HBM, fp32 arithmetic and membrane).
//
// Why they exist (profiles/r06/valu_facts*.log, tools/ubench/valu_facts*.hip): on gfx950 a wavefront's VALU instruction costs ~4.2 cycles of its SIMD
// (v_fma / mul / add / sub / and / add_u32: 2.3 - 3.1; v_rcp / v_exp: 8; v_pk_*_f32: 4.3 for two values), and the round-5 kernels issued 33 (forward)
// / 57 (backward) VALU instructions per neuron update: at half the bytes per update of the fp32 kernels that is 80 % / 67 % VALU-busy — they were
// bound by the vector ALU, not by HBM or by the width of their accesses.  These forms cut the instruction stream:
//   * forward: the spike bits of a lane's neurons are gathered with ONE carry instruction per neuron (v_cmp -> VCC, v_addc shifts the bit in) and
//     everything after that — the skip add, the 2-bit codes, both firing-rate counts — is bit-parallel arithmetic per time step, not per neuron;
//     the membrane reset is ONE select between two exactly equivalent sums ((1 - z) * h + z * v_reset  ==  z ? fma(0, h, v_reset) : h + 0 * v_reset,
//     bit for bit incl. NaN / inf / signed zeros, because z is 0 or 1 and both products are exact);
//   * backward: 8 | 4 neurons per lane with the T steps split into segments (checkpoint the membrane at the segment starts, re-sweep a segment to get
//     its h just before its backward steps): h of ONE segment lives in registers instead of all T steps — 16-byte lanes at T = 5, full occupancy at T = 10;
//   * low-rank second gradient (prediction head): the rank-9 pair of a wavefront's pixels is staged ONCE per step through wavefront-private LDS
//     (coalesced, no block barrier) instead of 9 per-lane loads per step held in 90 registers, and the product runs tap-outer with the weights read
//     from LDS once per segment;
//   * bf16 narrowing on v_cvt_pk_bf16_f32 (equal to the integer round-to-nearest-even definition on all 2^32 patterns, NaNs included).
// Arithmetic, operation order and every rounding are those of the round-5 kernels and of oracle/np_x16.py: the tests compare bit for bit.
#pragma once
#include "ss_common.hpp"

namespace {

struct Fwd16Args {
    const unsigned short* x_seq; const float* v_init; const unsigned short* skip_seq;
    unsigned short* out_seq; float* h_seq; float* v_last; unsigned long long* nnz;
    int T; long long N;
    float scale, tau, v_th, v_reset; const float* k;
    unsigned* cnt_ws;              // nullable (with nnz): per-workgroup counter partials (ss_neuron_fwd_ex)
    // ss_neuron_fwd_ex only (PK instantiations, round 5): 2-bit packed spike I/O exactly as in the fp32 kernel — the packed format knows no activation dtype
    const unsigned* skip_packed;   // nullable: the skip operand read from a packed spike tensor instead of skip_seq
    unsigned* out_packed;          // nullable: out (z + skip, values 0..3) written packed; out_seq may then be NULL (2.25 B/update forward)
};

struct Bwd16Args {
    const unsigned short* g_out_seq; const float* g_v_last; const float* h_seq; const float* v_init;
    unsigned short* g_x_seq; float* g_v_init; double* g_k_partials;
    int T; long long N;
    float scale, tau, v_th, v_reset, alpha; const float* k; int detach_reset;
};

typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int VEC> struct U32Words;                       // VEC 16-bit values as 32-bit words
template <> struct U32Words<8> { typedef u32x4 type; };
template <> struct U32Words<4> { typedef u32x2 type; };
typedef unsigned u32x1 __attribute__((ext_vector_type(1)));
template <> struct U32Words<2> { typedef u32x1 type; };

// two adjacent 16-bit values of one register -> fp32 pair (element 0 = low half)
template <int DT> __device__ __forceinline__ f2v widen2(unsigned w)
{
    if constexpr (DT == SS_DT_F16) return (f2v){__half2float(__ushort_as_half((unsigned short)(w & 0xffffu))), __half2float(__ushort_as_half((unsigned short)(w >> 16)))};
    else return (f2v){__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
}
// fp32 pair -> two 16-bit values, round to nearest even (the values are fp32 RESULTS: see narrow<DT> about v_fma_mixlo)
template <int DT> __device__ __forceinline__ unsigned narrow2(f2v f)
{
    if constexpr (DT == SS_DT_F16) return (unsigned)narrow<DT>(f[0]) | ((unsigned)narrow<DT>(f[1]) << 16);
    else {
        unsigned r; float a = f[0], b = f[1];
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
}

// {p.lo * w.lo, p.lo * w.hi} / {p.hi * w.lo, p.hi * w.hi}: one packed multiply, the tap taken from either half of its register pair (op_sel)
__device__ __forceinline__ f2v pk_mul_lo(f2v p, f2v w) { f2v r; asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(p), "v"(w)); return r; }
__device__ __forceinline__ f2v pk_mul_hi(f2v p, f2v w) { f2v r; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(p), "v"(w)); return r; }

// 1 / d for the ATan surrogate's d = 1 + u^2 >= 1: v_rcp_f32 and ONE Newton step equal the correctly rounded quotient for EVERY d in [1, 2^126)
// (exhaustive: tools/ubench/valu_facts.hip, profiles/r06/valu_facts.log — the 2^24 values that differ all have d >= 2^126, where 1 / d is denormal);
// 4 instructions instead of the 10 + hazard no-ops of the IEEE division sequence.  The kernel tracks the largest d it has seen (as an integer: NaN > inf >
// finite) and redoes the whole iteration with the true division when one reaches 2^126 (|h - v_th| > 1e18: only inf / NaN / overflow get there).
__device__ __forceinline__ float rcp_newton(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.f);
    return __builtin_fmaf(e, r0, r0);
}
constexpr unsigned kRcpNewtonLimit = 0x7e800000u;      // 2^126

template <int KIND> __device__ __forceinline__ f2v charge2(f2v v, f2v xs, float tau, float k, float v_reset)
{
    if constexpr (KIND == SS_KIND_IF) return v + xs;
    else if constexpr (KIND == SS_KIND_PLIF) { const f2v d = xs - (v - v_reset); return v + d * k; }
    else return (f2v){charge<KIND>(v[0], xs[0], tau, k, v_reset), charge<KIND>(v[1], xs[1], tau, k, v_reset)};
}
// membrane after the hard reset, exactly (1 - z) * h + z * v_reset for z in {0, 1} (see the header comment); c0 = 0 * v_reset
__device__ __forceinline__ f2v reset_if_fired(f2v h, f2v xh, float v_reset, float c0)
{
    const f2v v1 = (f2v){__builtin_fmaf(0.f, h[0], v_reset), __builtin_fmaf(0.f, h[1], v_reset)};
    const f2v v0 = h + c0;
    return (f2v){xh[0] >= 0.f ? v1[0] : v0[0], xh[1] >= 0.f ? v1[1] : v0[1]};
}

// one forward step of a lane's VEC neurons without outputs (the backward kernel's membrane sweep); h of the step is returned through hh
template <int KIND, int DT, int VEC>
__device__ __forceinline__ void sweep_step(f2v (&v)[VEC / 2], const typename U32Words<VEC>::type xw, f2v (&hh)[VEC / 2], float scale, float tau, float k,
                                           float v_th, float v_reset, float c0)
{
#pragma unroll
    for (int q = 0; q < VEC / 2; ++q) {
        const f2v xs = widen2<DT>(xw[q]) * scale;
        const f2v h = charge2<KIND>(v[q], xs, tau, k, v_reset);
        v[q] = reset_if_fired(h, h - v_th, v_reset, c0);
        hh[q] = h;
    }
}

// ---------------------------------------------------------------------------------------------------
// backward with h recomputed from the 16-bit layer input — segmented form
//   VEC  neurons per lane (8: 16-byte accesses, 4: 8-byte), NSEG segments of the T steps
//   G2   a second consumer's gradient is added on load (dense 16-bit g_out2_seq, or with LR the prediction head's rank-9 pair)
// ---------------------------------------------------------------------------------------------------
constexpr int kLr2Rank = 9;
#ifndef SS_LR_PIPE
#define SS_LR_PIPE 0
#endif
// dynamic LDS bytes of the low-rank form: 9 x C weights + per wavefront T x 9 x (64 * VEC / C) pair values
inline size_t bwd16_seg_lds_bytes(int T, int VEC, int C) { return sizeof(float) * ((size_t)kLr2Rank * C + (size_t)(kBlock / 64) * T * kLr2Rank * ((64 * VEC) / C)); }

// PASS (ATan surrogate): 0 = the fast pass — rcp_newton reciprocals; a wavefront that saw a denominator >= 2^126 raises *redo_flag; 1 = the exact pass, a second
// launch behind the first that returns at once unless *redo_flag is set and otherwise redoes the whole layer with the IEEE division (same loads, same stores);
// 2 = one launch with the IEEE division (Sigmoid surrogate; A/B).  Two launches instead of a redo branch inside one kernel: the branch cost 24 registers of the
// fast pass (shared allocation) — spills to scratch at 3 wavefronts per SIMD, 1.19 x the algorithmic HBM bytes instead of 1.07 x (profiles/r06/).
template <int KIND, int SG, int DT, int TS, int VEC, int NSEG, bool G2, bool LR, int WAVES = 1, bool HAS_G1 = true, bool SUM = false, int PASS = 2>
__global__ __launch_bounds__(kBlock, WAVES) void neuron_bwd16_seg_kernel(Bwd16Args a, const unsigned short* __restrict__ x_seq,
                                                                  const unsigned short* __restrict__ g_out2_seq, unsigned short* __restrict__ g_sum_seq,
                                                                  const float* __restrict__ lr_p, const float* __restrict__ lr_w, int lr_C, int pair_x4,
                                                                  unsigned* __restrict__ redo_flag = nullptr)
{
    constexpr bool EXACT = !(SG == SS_SG_ATAN && PASS == 0);
    if constexpr (PASS == 1) { if (*redo_flag == 0u) return; }              // (uniform; written by the previous launch on this stream)
    static_assert(VEC == 8 || VEC == 4 || VEC == 2, "vector lanes");
    static_assert(NSEG >= 1 && NSEG <= TS, "segments");
    static_assert(!LR || G2, "the low-rank pair is a second gradient");
    static_assert(LR || HAS_G1, "only the low-rank form can run without a dense first gradient");
    static_assert(!SUM || G2, "g_sum_seq is the sum of two gradients");     // (a compile-time flag: a branch per step would split the step loop into blocks and the compiler sinks the tap sums across them)
    typedef typename U32Words<VEC>::type wvec_t;
    constexpr int NP = VEC / 2;
    constexpr int SEGMAX = (TS + NSEG - 1) / NSEG;
    // dynamic LDS (LR only; bwd16_seg_lds_bytes): the 9 x C weights of the pair, then per wavefront the pair values of its pixels for the T steps.
    // Sized by the launch, not for the largest C: at C = 32 a workgroup takes 7 - 13 KB instead of 42 KB, which capped the CU at 3 workgroups.
    extern __shared__ __attribute__((aligned(16))) float lr_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // scalar: every address below is (scalar base) + (32-bit lane offset)
    const int lr_per = LR ? kLr2Rank * ((64 * VEC) / lr_C) : 0;                               // pair floats of a wavefront and step (host: C divides 64 * VEC)
    float* const lr_ws = lr_lds;
    float* const lr_pw = lr_lds + kLr2Rank * lr_C + wave * (TS * lr_per);                      // [TS][lr_per], this wavefront's
    if constexpr (LR) {
        for (int q = threadIdx.x; q < kLr2Rank * lr_C; q += kBlock) lr_ws[q] = lr_w[q];
        __syncthreads();
    }
    const long long N = a.N, NV = N / VEC;
    const float k = (KIND == SS_KIND_PLIF) ? *a.k : 0.f;
    const float scale = a.scale, tau = a.tau, v_th = a.v_th, v_reset = a.v_reset, alpha = a.alpha;
    const float c0 = 0.f * v_reset;
    const float c_atan = (float)(M_PI / 2.0 * (double)alpha);
    const float half_alpha = (float)((double)alpha / 2.0);
    const bool detach = a.detach_reset != 0;
    const bool want_gk = (KIND == SS_KIND_PLIF) && a.g_k_partials != nullptr;
    constexpr bool has_g1 = HAS_G1;
    double acc_k = 0.0;
    [[maybe_unused]] unsigned dmax = 0u;                          // largest ATan denominator seen (as an integer: NaN > inf > finite)
    const int lr_log2C = LR ? __builtin_ctz((unsigned)lr_C) : 0;                              // C divides 64 * VEC: a power of two (host)
    // wave-uniform trip count: the wavefront's lanes stage the low-rank pair together
    for (long long i0 = (long long)blockIdx.x * kBlock + wave * 64; i0 < NV; i0 += (long long)gridDim.x * kBlock) {
        const bool active = i0 + lane < NV;
        // idle lanes of the last wavefront repeat the last vector — inputs, pixel, channels — and so store the same values to the same address
        const unsigned lo = (unsigned)(active ? lane : (int)(NV - 1 - i0)) * VEC;
        // the lane's channels / pixel inside the wavefront's 64 * VEC neurons
        const int lr_c0 = LR ? (int)(lo & (unsigned)(lr_C - 1)) : 0;
        const int lr_pl = LR ? (int)(lo >> lr_log2C) : 0;
        const unsigned short* xb = x_seq + i0 * VEC;
        const unsigned short* gb = has_g1 ? a.g_out_seq + i0 * VEC : xb;
        const unsigned short* g2b = (G2 && !LR) ? g_out2_seq + i0 * VEC : xb;
        unsigned short* gxb = a.g_x_seq + i0 * VEC;

        // ---- the pair of this wavefront's pixels, all T steps: HBM -> registers -> wavefront-private LDS (issued first: needed first)
        if constexpr (LR) {
            const long long rows = N / lr_C, lim = rows * kLr2Rank;
            const long long p0 = ((i0 * VEC) / lr_C) * kLr2Rank;                               // scalar: first pair float of the wavefront's pixels within a step
            const float* pb = lr_p + p0;
            // LDS-DMA (global_load_lds): no staging registers; the wavefront's s_waitcnt vmcnt(0) below orders it before the LDS reads
            if (pair_x4) {                                      // rows % 4 == 0 and 4 | pixels per wavefront (host): one aligned float4 per lane and step
                if (lane * 4 < lr_per && p0 + lane * 4 + 3 < lim) {
#pragma unroll
                    for (int t = 0; t < TS; ++t)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb + (long long)t * lim + lane * 4),
                                                         (__attribute__((address_space(3))) void*)(lr_pw + t * lr_per), 16, 0, 0);
                }
            } else {
                for (int r = 0; r * 64 < lr_per; ++r) {
                    const int q = r * 64 + lane;
                    if (q < lr_per && p0 + q < lim) {
#pragma unroll
                        for (int t = 0; t < TS; ++t)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb + (long long)t * lim + q),
                                                             (__attribute__((address_space(3))) void*)(lr_pw + t * lr_per + r * 64), 4, 0, 0);
                    }
                }
            }
        }

        {
            // ---- the lane's x of all steps (read twice: membrane sweep, then segment by segment)
            wvec_t xs[TS];
#pragma unroll
            for (int t = 0; t < TS; ++t) xs[t] = load_stream(reinterpret_cast<const wvec_t*>(xb + (long long)t * N + lo));
            f2v v0[NP], gv[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                if (a.v_init) v0[q] = *reinterpret_cast<const f2v*>(a.v_init + i0 * VEC + lo + 2 * q); else v0[q] = (f2v){v_reset, v_reset};
                if (a.g_v_last) gv[q] = *reinterpret_cast<const f2v*>(a.g_v_last + i0 * VEC + lo + 2 * q); else gv[q] = (f2v){0.f, 0.f};
            }

            // ---- membrane at the segment starts
            f2v vchk[NSEG][NP];
            {
                f2v v[NP], hh[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) { v[q] = v0[q]; vchk[0][q] = v0[q]; }
#pragma unroll
                for (int s = 0; s + 1 < NSEG; ++s) {
#pragma unroll
                    for (int t = s * TS / NSEG; t < (s + 1) * TS / NSEG; ++t) sweep_step<KIND, DT, VEC>(v, xs[t], hh, scale, tau, k, v_th, v_reset, c0);
#pragma unroll
                    for (int q = 0; q < NP; ++q) vchk[s + 1][q] = v[q];
                }
            }

            // ---- segments, last first
#pragma unroll
            for (int s = NSEG - 1; s >= 0; --s) {
                const int t0 = s * TS / NSEG, t1 = (s + 1) * TS / NSEG;
                // gradients of the segment's steps
                wvec_t gs[SEGMAX], g2[(G2 && !LR) ? SEGMAX : 1];
                if (has_g1) {
#pragma unroll
                    for (int t = t1 - 1; t >= t0; --t) gs[t - t0] = load_stream(reinterpret_cast<const wvec_t*>(gb + (long long)t * N + lo));
                }
                if constexpr (G2 && !LR) {
#pragma unroll
                    for (int t = t1 - 1; t >= t0; --t) g2[t - t0] = load_stream(reinterpret_cast<const wvec_t*>(g2b + (long long)t * N + lo));
                }
                // h of the segment's steps
                f2v h[SEGMAX][NP];
                {
                    f2v v[NP];
#pragma unroll
                    for (int q = 0; q < NP; ++q) v[q] = vchk[s][q];
#pragma unroll
                    for (int t = t0; t < t1; ++t) sweep_step<KIND, DT, VEC>(v, xs[t], h[t - t0], scale, tau, k, v_th, v_reset, c0);
                }
                // second gradient of the segment's steps from the pair: taps ascending, multiply and add rounded separately (ss_neuron_bwd_fork_lr_f32's order).
                // The taps are read from LDS two at a time; a packed multiply takes either half of that register pair for both of its products (op_sel).
                f2v lracc[LR ? SEGMAX : 1][LR ? NP : 1];
                if constexpr (LR) {
                    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): the pair has landed in LDS (and x / g are needed next anyway)
                    __builtin_amdgcn_wave_barrier();
                    const float* pl = lr_pw + lr_pl * kLr2Rank;
                    const float* wl = &lr_ws[lr_c0];
                    // software pipeline over the tap pairs: the LDS reads of pair jj + 1 (its weights, its T-segment of pair values) are issued before the
                    // multiply-adds of pair jj, into the other half of a double buffer (SS_LR_PIPE 2; 1: all reads of a pair up front, one wait; 0: read at use)
                    f2v wa[2][NP], wb[2][NP], pq[2][SEGMAX];
                    auto lr_load = [&](int j, int buf) {
#pragma unroll
                        for (int q = 0; q < NP; ++q) {
                            wa[buf][q] = *reinterpret_cast<const f2v*>(wl + j * lr_C + 2 * q);
                            if (j + 1 < kLr2Rank) wb[buf][q] = *reinterpret_cast<const f2v*>(wl + (j + 1) * lr_C + 2 * q);
                        }
#pragma unroll
                        for (int t = t1 - 1; t >= t0; --t) {
                            pq[buf][t - t0][0] = pl[t * lr_per + j];
                            pq[buf][t - t0][1] = (j + 1 < kLr2Rank) ? pl[t * lr_per + j + 1] : 0.f;
                        }
                    };
                    if constexpr (SS_LR_PIPE == 2) lr_load(0, 0);
#pragma unroll
                    for (int jj = 0; jj < (kLr2Rank + 1) / 2; ++jj) {
                        const int j = 2 * jj;
                        const int buf = SS_LR_PIPE == 2 ? (jj & 1) : 0;
                        __builtin_amdgcn_sched_barrier(0);           // (hoisted to the top, all reads together would cost 2 * 9 * NP + 10 * SEG registers)
                        if constexpr (SS_LR_PIPE == 2) { if (j + 2 < kLr2Rank) lr_load(j + 2, (jj + 1) & 1); }
                        else lr_load(j, 0);
                        if constexpr (SS_LR_PIPE >= 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int t = t1 - 1; t >= t0; --t) {
                            const f2v pp = pq[buf][t - t0];
#pragma unroll
                            for (int q = 0; q < NP; ++q) {
                                const f2v m0 = pk_mul_lo(pp, wa[buf][q]);
                                lracc[t - t0][q] = (j == 0) ? m0 : lracc[t - t0][q] + m0;
                                if (j + 1 < kLr2Rank) { const f2v m1 = pk_mul_hi(pp, wb[buf][q]); lracc[t - t0][q] = lracc[t - t0][q] + m1; }
                            }
                        }
                    }
                }
                // backward steps of the segment
#pragma unroll
                for (int t = t1 - 1; t >= t0; --t) {
                    wvec_t xv, sumv;
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const f2v he = h[t - t0][q];
                        const f2v xh = he - v_th;
                        f2v g_s = has_g1 ? widen2<DT>(gs[t - t0][q]) : (f2v){0.f, 0.f};
                        if constexpr (LR) {
                            g_s = has_g1 ? g_s + lracc[t - t0][q] : lracc[t - t0][q];
                            if constexpr (SUM) sumv[q] = narrow2<DT>(g_s);
                        } else if constexpr (G2) {
                            g_s = g_s + widen2<DT>(g2[t - t0][q]);
                            if constexpr (SUM) sumv[q] = narrow2<DT>(g_s);
                        }
                        if (!detach) g_s = g_s + (gv[q] * v_reset - gv[q] * he);
                        f2v sgv;
                        if constexpr (SG == SS_SG_ATAN && !EXACT) {       // surrogate_grad<ATan>'s operations, the reciprocal by rcp_newton
                            const f2v u = xh * c_atan;
                            const f2v d = u * u + 1.f;
                            dmax = max(dmax, max(__float_as_uint(d[0]), __float_as_uint(d[1])));
                            const f2v r = (f2v){rcp_newton(d[0]), rcp_newton(d[1])};
                            sgv = (r * half_alpha) * g_s;
                        } else sgv = (f2v){surrogate_grad<SG>(xh[0], alpha, c_atan, half_alpha, g_s[0]), surrogate_grad<SG>(xh[1], alpha, c_atan, half_alpha, g_s[1])};
                        const f2v omz = (f2v){xh[0] >= 0.f ? 0.f : 1.f, xh[1] >= 0.f ? 0.f : 1.f};      // 1 - z
                        const f2v g_h = sgv + gv[q] * omz;
                        f2v g_x;
                        if constexpr (KIND == SS_KIND_IF) { g_x = g_h; gv[q] = g_h; }
                        else if constexpr (KIND == SS_KIND_LIF) { g_x = (f2v){g_h[0] / tau, g_h[1] / tau}; gv[q] = g_h - g_x; }
                        else {
                            g_x = g_h * k; gv[q] = g_h - g_x;
                            if (want_gk) {
                                f2v v_prev;
                                if (t == t0) v_prev = vchk[s][q];
                                else { const f2v hp = h[t > t0 ? t - t0 - 1 : 0][q]; v_prev = reset_if_fired(hp, hp - v_th, v_reset, c0); }
                                // element order of the round-5 kernel: ascending within the lane; idle lanes add nothing
                                acc_k += active ? (double)g_h[0] * (double)((he[0] - v_prev[0]) / k) : 0.0;
                                acc_k += active ? (double)g_h[1] * (double)((he[1] - v_prev[1]) / k) : 0.0;
                            }
                        }
                        xv[q] = narrow2<DT>(g_x * scale);
                    }
                    // no branch on `active`: an idle lane of the last wavefront holds the last vector's inputs, so it stores the same values to the same address
                    store_out(reinterpret_cast<wvec_t*>(gxb + (long long)t * N + lo), xv);
                    if constexpr (SUM) *reinterpret_cast<wvec_t*>(g_sum_seq + i0 * VEC + (long long)t * N + lo) = sumv;
                }
            }
            if (a.g_v_init) {
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<f2v*>(a.g_v_init + i0 * VEC + lo + 2 * q) = gv[q];
            }
        }
    }
    if constexpr (!EXACT) {                             // a denominator left rcp_newton's range somewhere in this wavefront: ask for the exact pass
        if (__any((int)(dmax >= kRcpNewtonLimit)) && lane == 0) atomicOr(redo_flag, 1u);
    }
    if (want_gk) gk_epilogue(acc_k, a.g_k_partials);   // wave-uniform
}

// counter epilogue of the forward kernel: lane counts -> wavefront butterfly -> LDS -> per-workgroup partial (cnt_ws) or atomics
__device__ __forceinline__ void count_epilogue16(unsigned c_spk, unsigned c_out, unsigned long long* nnz, unsigned* cnt_ws)
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

// ---------------------------------------------------------------------------------------------------
// forward, 2-bit packed spike output and / or packed skip input (the ss_neuron_fwd_ex training form): 8 neurons per lane, compile-time T
//   DENSE: the output is ALSO written as 16-bit values (a consumer that cannot read packed spikes)
// ---------------------------------------------------------------------------------------------------
// the 8 spike bits of a lane -> bit 2e of a 16-bit field (the low bit of neuron e's 2-bit code)
__device__ __forceinline__ unsigned spread8(unsigned z)
{
    z = (z | (z << 4)) & 0x0F0Fu;
    z = (z | (z << 2)) & 0x3333u;
    return (z | (z << 1)) & 0x5555u;
}

template <int KIND, int DT, int TS, bool SKIP, bool DENSE>
__global__ __launch_bounds__(kBlock) void neuron_fwd16_pk8_kernel(Fwd16Args a)
{
    static_assert(TS > 0, "compile-time T");
    constexpr int NP = 4;
    const long long NV = a.N / 8, NW = a.N / 16;                 // vectors / packed words per time step (N % 16 == 0: a word's two lanes are both in range or both out)
    const float k = (KIND == SS_KIND_PLIF) ? *a.k : 0.f;
    const float scale = a.scale, tau = a.tau, v_th = a.v_th, v_reset = a.v_reset;
    const float c0 = 0.f * v_reset;
    unsigned c_spk = 0, c_out = 0;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < NV; i += (long long)gridDim.x * kBlock) {
        u32x4 xs[TS];
#pragma unroll
        for (int t = 0; t < TS; ++t) xs[t] = load_stream(reinterpret_cast<const u32x4*>(a.x_seq + ((long long)t * NV + i) * 8));
        unsigned sb[SKIP ? TS : 1];
        if constexpr (SKIP) {
            if (a.skip_packed) {                                 // wave-uniform
#pragma unroll
                for (int t = 0; t < TS; ++t) sb[t] = (a.skip_packed[(long long)t * NW + (i >> 1)] >> (16 * (int)(i & 1))) & 0xffffu;
            } else {
#pragma unroll
                for (int t = 0; t < TS; ++t) {                   // dense 16-bit skip: small integers 0..3, exact in the 2-bit code (read modulo 4)
                    const u32x4 sv = *reinterpret_cast<const u32x4*>(a.skip_seq + ((long long)t * NV + i) * 8);
                    unsigned f = 0u;
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const f2v s2 = widen2<DT>(sv[q]);
                        f |= (((unsigned)s2[0] & 3u) << (4 * q)) | (((unsigned)s2[1] & 3u) << (4 * q + 2));
                    }
                    sb[t] = f;
                }
            }
        }
        f2v v[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) { if (a.v_init) v[q] = *reinterpret_cast<const f2v*>(a.v_init + i * 8 + 2 * q); else v[q] = (f2v){v_reset, v_reset}; }
#pragma unroll
        for (int t = 0; t < TS; ++t) {
            unsigned zm = 0u;                                     // after the loop: bit e = spike of neuron e
#pragma unroll
            for (int q = NP - 1; q >= 0; --q) {
                const f2v xsc = widen2<DT>(xs[t][q]) * scale;
                const f2v h = charge2<KIND>(v[q], xsc, tau, k, v_reset);
                const f2v xh = h - v_th;
                const f2v v1 = (f2v){__builtin_fmaf(0.f, h[0], v_reset), __builtin_fmaf(0.f, h[1], v_reset)};
                const f2v v0 = h + c0;
                float n0, n1;
                // z = (xh >= 0) -> VCC; v = z ? v1 : v0; zm = 2 * zm + z (the carry-in shifts the spike bit in): three instructions per neuron
                asm("v_cmp_le_f32 vcc, 0, %2\n\tv_cndmask_b32 %0, %3, %4, vcc\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc"
                    : "=v"(n1), "+v"(zm) : "v"(xh[1]), "v"(v0[1]), "v"(v1[1]) : "vcc");
                asm("v_cmp_le_f32 vcc, 0, %2\n\tv_cndmask_b32 %0, %3, %4, vcc\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc"
                    : "=v"(n0), "+v"(zm) : "v"(xh[0]), "v"(v0[0]), "v"(v1[0]) : "vcc");
                v[q] = (f2v){n0, n1};
            }
            const unsigned zf = spread8(zm);
            unsigned cf = zf;
            if constexpr (SKIP) {                                 // 2-bit fields: skip + z, saturating at 3 (contract of ss_neuron.h: z + skip <= 3)
                const unsigned full = sb[t] & (sb[t] >> 1) & zf;  // fields that hold 3 and would receive a spike
                cf = sb[t] + (zf ^ full);
            }
            c_spk += __builtin_popcount(zm);
            c_out += __builtin_popcount((cf | (cf >> 1)) & 0x5555u);
            if constexpr (DENSE) {
                if (a.out_seq) {
                    u32x4 ov;
#pragma unroll
                    for (int q = 0; q < NP; ++q)
                        ov[q] = (unsigned)code_to_op<DT>((cf >> (4 * q)) & 3u) | ((unsigned)code_to_op<DT>((cf >> (4 * q + 2)) & 3u) << 16);
                    store_out(reinterpret_cast<u32x4*>(a.out_seq + ((long long)t * NV + i) * 8), ov);
                }
            }
            if (a.out_packed) {
                const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)cf, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]: the odd neighbour's codes, on the VALU (a shuffle would go through the LDS pipe)
                if ((threadIdx.x & 1) == 0) a.out_packed[(long long)t * NW + (i >> 1)] = cf | (hi << 16);
            }
        }
        if (a.v_last) {
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<f2v*>(a.v_last + i * 8 + 2 * q) = v[q];
        }
    }
    if (a.nnz) count_epilogue16(c_spk, c_out, a.nnz, a.cnt_ws);   // wave-uniform
}

}  // namespace

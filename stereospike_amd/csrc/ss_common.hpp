// ss_common.hpp — shared device helpers of the ss_*.hip translation units (internal linkage: every unit gets its own copy).
#pragma once
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
constexpr long long kGkWsFloats = 2 * kMaxGridGk;   // the partials are doubles

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

// PLIF dL/dk = sum over every neuron and time step of a layer of g_h * (h - v_prev) / k: ONE heavily cancelling sum (condition number ~1e3
// on the bottleneck layers), so it is carried in fp64 from the lane's accumulator to the last addition — the kernels are HBM-bound, the
// fp64 adds are free — and rounded to fp32 once, when the scalar is written.  Order: lane (grid-stride, steps descending) -> wavefront
// butterfly -> the workgroup's wavefronts ascending -> one partial per workgroup in the caller's workspace -> gk_finish_kernel (fixed order).
__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void gk_epilogue(double acc_k, double* __restrict__ partials)
{
    __shared__ double s_k[SS_BLOCK / 64];
    const double w = wave_sum_f64(acc_k);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_k[wave] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < SS_BLOCK / 64; ++q) s += s_k[q];
        partials[blockIdx.x] = s;
    }
}

// second pass of the dL/dk reduction: one workgroup, fixed order -> bit-reproducible
__global__ __launch_bounds__(SS_BLOCK) void gk_finish_kernel(const double* __restrict__ partials, int n, float* __restrict__ g_k)
{
    __shared__ double s[SS_BLOCK];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += SS_BLOCK) acc += partials[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = SS_BLOCK / 2; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *g_k = (float)s[0];
}

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

// Workgroup barrier that orders LDS traffic only: __syncthreads() carries an s_waitcnt vmcnt(0) whenever vector-memory loads are in flight, i.e. it DRAINS every
// prefetch a kernel has issued across it (guide: cdna_hip_programming.md "Pipelining across barriers") — a window fetched three stages ahead is waited for at
// the very next stage barrier.  This one waits for the wavefront's own LDS operations (lgkmcnt) and leaves vmcnt alone.  Loads whose registers are consumed
// before the barrier are still waited for where they are used (the compiler's own partial vmcnt(N)).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// code (0..3) of a 2-bit packed spike -> bf16 bit pattern of the same small integer: 0x0000, 0x3F80, 0x4000, 0x4040
#define SS_CODE_LUT 0x404040003F800000ull
__device__ __forceinline__ unsigned short code_to_bf16(unsigned c) { return (unsigned short)((SS_CODE_LUT >> (16 * c)) & 0xFFFFu); }
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------
// 16-bit activation modes (round 5): the SAME kernels on 16-bit I/O with SINGLE-term operands on the native matrix-core type.
// DT = 0: fp32 I/O, operands as exact multi-term bf16 splits (3 weight terms, 3 terms of a dense fp32 operand: six cross terms kept);
// DT = SS_DT_BF16 / SS_DT_F16: activations / activation gradients are stored in that format — they ARE the operand (no split, no conversion); spikes are
// exact in either format; the fp32 master weight is rounded ONCE to the format (what autocast does to a synapse); products are then exact in fp32
// (8 x 8 / 11 x 11 significand bits), accumulation is fp32 in the MFMA, the result is narrowed once on store.  Weight gradients stay fp32.
// ---------------------------------------------------------------------------------------------------
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
template <int DT> __device__ __forceinline__ f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c)
{
    if constexpr (DT == SS_DT_F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// code (0..3) of a 2-bit packed spike -> the 16-bit pattern of the same small integer in the operand format (DT = 0: bf16)
template <int DT> __device__ __forceinline__ unsigned short code_to_op(unsigned c)
{
    if constexpr (DT == SS_DT_F16) return (unsigned short)((0x420040003C000000ull >> (16 * c)) & 0xFFFFu);    // 0x0000, 0x3C00, 0x4000, 0x4200
    else return code_to_bf16(c);
}
// fp32 -> the operand format, round to nearest even (weights: once per step in the prep kernels; fp32 inputs of the first layer)
template <int DT> __device__ __forceinline__ unsigned short round_op(float v) { return narrow<(DT == SS_DT_F16 ? SS_DT_F16 : SS_DT_BF16)>(v); }
template <int DT> __device__ __forceinline__ float widen_op(unsigned short b) { return widen<(DT == SS_DT_F16 ? SS_DT_F16 : SS_DT_BF16)>(b); }
// output element type of a kernel's activation result: float (DT = 0) or the 16-bit pattern
template <int DT> struct ActT { typedef unsigned short type; };
template <> struct ActT<0> { typedef float type; };
// store 4 consecutive activation values (16 B as fp32, 8 B narrowed)
template <int DT> __device__ __forceinline__ void store_act4(typename ActT<DT>::type* p, float a, float b, float c, float d)
{
    if constexpr (DT == 0) *reinterpret_cast<f4*>(p) = (f4){a, b, c, d};
    else { u16x4 o; o[0] = narrow<DT>(a); o[1] = narrow<DT>(b); o[2] = narrow<DT>(c); o[3] = narrow<DT>(d); *reinterpret_cast<u16x4*>(p) = o; }
}

// x [NB * h][w][C_in] fp32 spike counts (or, PACKED, the 2-bit packed spike tensor) -> xT[(source row) * KSR + k-step][ci][16 sources] bf16,
// KSR = ceil(w / 16), zero padded
template <bool PACKED, int DT = 0>             // DT != 0: the operand format is DT; a dense input is then the 16-bit spike tensor itself
__global__ __launch_bounds__(kBlock) void upconv_bwd_xprep_kernel(const void* __restrict__ xv, unsigned short* __restrict__ xT, long long rows, int w,
                                                                  int CIN)
{
    const float* x = static_cast<const float*>(xv);
    const unsigned short* x16 = static_cast<const unsigned short*>(xv);
    const unsigned* xp = static_cast<const unsigned*>(xv);
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
            const long long el = (row * w + sx) * CIN + ci;
            unsigned short v = 0;
            if (sx < w) {
                if constexpr (PACKED) v = code_to_op<DT>((xp[el >> 4] >> (2 * (int)(el & 15))) & 3u);
                else if constexpr (DT != 0) v = x16[el];
                else v = (unsigned short)(__float_as_uint(x[el]) >> 16);
            }
            if (rr < 8) a[rr] = v; else b[rr - 8] = v;
        }
        *reinterpret_cast<u16x8*>(xT + i * 16) = a;
        *reinterpret_cast<u16x8*>(xT + i * 16 + 8) = b;
    }
}

inline int grid_for(long long work_items, int cap = kMaxGrid)
{
    long long g = (work_items + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

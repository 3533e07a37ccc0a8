// ss_head.hip — the full-resolution prediction head on 2-BIT PACKED spikes (include/ss_neuron.h: ss_head_proj_packed_f32, ss_head_wgrad_packed_f32).
//
// Reference: predict_depth1 = NNConvUpsampling(32, 1, kernel_size=3, up_size=(260, 346)) on the output of deconv1 + skip
// (/root/reference/network/SNN_models.py:133-148 built, :186-188 called; blocks.py:110-132).  In the projected form (ss_upconv.hip) the head is
//     P[pixel][tap] = sum_c x[pixel][c] * W[0][c][tap]            (9 taps; was: a library fp32 GEMM with N = 9 over the dense 0.92 GB tensor)
//     depth map     = gather of P (ss_upconv_cl_fwd_f32, unchanged)
// and its weight gradient  g_W[c][tap] = sum_pixel x[pixel][c] * g_P[pixel][tap]  (was: a library split-K GEMM with K = 7.2 M over the same
// dense tensor).  x is the stage's spike output (+ skip add: values 0 .. 3), its ONLY other use being these two contractions — so when both
// read the 2-bit packed form (1/16 of the bytes) the neuron layer of the largest decoder stage writes no dense output at all
// (4.5 instead of 8.25 B/update).
//
// Forward: v_mfma_f32_32x32x16_bf16 with M = 32 pixels, K = channels, N = 9 taps (padded to 32): the spike codes expand to bf16 in registers
// (exact), the weight is split exactly into three bf16 terms held as fragments in registers — every product exact, fp32 accumulation.
// Weight gradient: v_mfma_f32_16x16x32_bf16 with M = taps, N = channels, K = rows; g_P split into three bf16 terms in registers (exact products),
// operands transposed through a per-wavefront LDS slice; per-wavefront partials -> fixed-order fp64 second pass: deterministic.  (A lane-per-
// channel VALU form with broadcast or scalar loads of g_P ran 1.4 - 1.5 ms at config 3 — latency / constant-cache bound; profiles/r03/bench_head.log.)
#include "ss_common.hpp"

namespace {

constexpr int kHdTaps = 9;

template <int C>
__global__ __launch_bounds__(kBlock) void head_proj_packed_kernel(const unsigned* __restrict__ xp, const float* __restrict__ Wt, float* __restrict__ P,
                                                                  long long rows)
{
    constexpr int KS = C / 16, WPP = C / 16;                                    // k-steps; packed words per pixel
    const int lane = threadIdx.x & 63, n = lane & 31, half = lane >> 5;
    // B fragments: element e of k-step s = split term of W[c = 16 s + 8 half + e][tap n] (0 for the padding columns n >= 9)
    s16x8 b[KS][3];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = n < kHdTaps ? Wt[(16 * s + 8 * half + e) * kHdTaps + n] : 0.f;
            const __bf16 h1 = (__bf16)v;
            const float r1 = v - (float)h1;
            const __bf16 h2 = (__bf16)r1;
            const __bf16 h3 = (__bf16)(r1 - (float)h2);
            b[s][0][e] = __builtin_bit_cast(short, h1); b[s][1][e] = __builtin_bit_cast(short, h2); b[s][2][e] = __builtin_bit_cast(short, h3);
        }
    const long long n_blocks = (rows + 31) / 32;
    const long long wave = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), n_waves = (long long)gridDim.x * (kBlock / 64);
#pragma unroll 1
    for (long long blk = wave; blk < n_blocks; blk += n_waves) {
        const long long p0 = blk * 32;
        const long long pix = min(p0 + n, rows - 1);                            // A: row m = lane & 31 = pixel of the block
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const unsigned wv = xp[pix * WPP + s] >> (16 * half);               // channels 16 s + 8 half .. + 7: 16 bits of the word
            s16x8 a;
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = (short)code_to_bf16((wv >> (2 * e)) & 3u);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[s][2], acc, 0, 0, 0);     // smallest terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[s][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[s][0], acc, 0, 0, 0);
        }
        // D[row = (r & 3) + 8 (r >> 2) + 4 half][col = n] -> P[p0 + row][n]
        if (n < kHdTaps) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long p = p0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (p < rows) P[p * kHdTaps + n] = acc[r];
            }
        }
    }
}

// g_W partials: one [C][9] block per wavefront, on v_mfma_f32_16x16x32_bf16: D[m = tap (9 of 16)][n = channel (16 per tile)] += sum over 32 rows of
// A[m][row] * B[row][n].  A wavefront stages 64 rows at a time — g_P (64 x 9 floats) and the packed words, coalesced loads — TRANSPOSED into its
// slice of LDS, so that a lane's 8 consecutive rows of one tap / one packed word are two 16-B reads; g_P is split into three bf16 terms in
// registers (exact sum), the spike codes expand to bf16 exactly: three MFMAs per channel tile and 32 rows, every product exact, fp32 accumulation.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kHwRows = 64;                                                     // rows per staging block
constexpr int kHwGS = 68;                                                       // row stride (floats) of the transposed g_P slice: 16-B aligned, banks spread

template <int C>
__global__ __launch_bounds__(kBlock) void head_wgrad_packed_kernel(const unsigned* __restrict__ xp, const float* __restrict__ gP, float* __restrict__ part,
                                                                   long long rows, long long rows_per_wave)
{
    constexpr int WPP = C / 16, NT = C / 16;                                    // packed words per pixel = channel tiles of 16
    __shared__ __attribute__((aligned(16))) float gT[kBlock / 64][kHdTaps * kHwGS];
    __shared__ __attribute__((aligned(16))) unsigned xT[kBlock / 64][WPP * kHwRows];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long wave = (long long)blockIdx.x * (kBlock / 64) + wv;
    const long long r0 = wave * rows_per_wave, r1 = min(rows, r0 + rows_per_wave);
    const int m = min(lane & 15, kHdTaps - 1), kq = lane >> 4;                   // A: tap row (rows m >= 9 of D are never stored), k group of 8 rows
    const int cn = lane & 15;                                                   // B: channel within the tile
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float* const gt = gT[wv];
    unsigned* const xt = xT[wv];
#pragma unroll 1
    for (long long b0 = r0; b0 < r1; b0 += kHwRows) {
        const int nr = (int)min((long long)kHwRows, r1 - b0);
        // ---- stage: g_P[b0 .. b0 + nr)[9] -> gt[tap][row], words -> xt[word][row]; rows beyond nr are zero (they contribute nothing)
        float gv[kHdTaps];
        unsigned wvv[WPP];
#pragma unroll
        for (int q = 0; q < kHdTaps; ++q) {
            const int idx = lane + 64 * q;
            gv[q] = idx < nr * kHdTaps ? gP[b0 * kHdTaps + idx] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < WPP; ++q) {
            const int idx = lane + 64 * q;
            wvv[q] = idx < nr * WPP ? xp[b0 * WPP + idx] : 0u;
        }
#pragma unroll
        for (int q = 0; q < kHdTaps; ++q) {
            const int idx = lane + 64 * q, row = idx / kHdTaps, j = idx - row * kHdTaps;
            gt[j * kHwGS + row] = gv[q];
        }
#pragma unroll
        for (int q = 0; q < WPP; ++q) {
            const int idx = lane + 64 * q, row = idx / WPP, i = idx - row * WPP;
            xt[i * kHwRows + row] = wvv[q];
        }
        // (one wavefront writes and reads its own slice: program order + lgkmcnt, no barrier)
#pragma unroll
        for (int ks = 0; ks < kHwRows / 32; ++ks) {
            if (32 * ks >= nr) break;                                           // wave-uniform
            const int rb = 32 * ks + 8 * kq;
            const f4 ga = *reinterpret_cast<const f4*>(gt + m * kHwGS + rb), gb = *reinterpret_cast<const f4*>(gt + m * kHwGS + rb + 4);
            s16x8 ah, am, al;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = e < 4 ? ga[e] : gb[e - 4];
                const __bf16 h1 = (__bf16)v;
                const float q1 = v - (float)h1;
                const __bf16 h2 = (__bf16)q1;
                const __bf16 h3 = (__bf16)(q1 - (float)h2);
                ah[e] = __builtin_bit_cast(short, h1); am[e] = __builtin_bit_cast(short, h2); al[e] = __builtin_bit_cast(short, h3);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                const u4 wa = *reinterpret_cast<const u4*>(xt + t * kHwRows + rb), wb = *reinterpret_cast<const u4*>(xt + t * kHwRows + rb + 4);
                s16x8 bx;
#pragma unroll
                for (int e = 0; e < 8; ++e) bx[e] = (short)code_to_bf16(((e < 4 ? wa[e] : wb[e - 4]) >> (2 * cn)) & 3u);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bx, acc[t], 0, 0, 0);     // smallest terms first
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bx, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bx, acc[t], 0, 0, 0);
            }
        }
    }
    // D[row = 4 (lane >> 4) + r][col = lane & 15] -> part[wave][c = 16 t + col][tap = row]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tap = 4 * kq + r;
            if (tap < kHdTaps) part[(wave * C + 16 * t + cn) * kHdTaps + tap] = acc[t][r];
        }
}

// second pass: g_W[c][j] (+)= sum over the wavefront partials in fp64 — one workgroup per element, strided partial sums + a fixed-order tree
__global__ __launch_bounds__(kBlock) void head_wgrad_finish_kernel(const float* __restrict__ part, float* __restrict__ gW, long long n_waves, int CJ, int accumulate)
{
    __shared__ double sh[kBlock];
    const int i = blockIdx.x;
    double s = 0.0;
    for (long long w = threadIdx.x; w < n_waves; w += kBlock) s += (double)part[w * CJ + i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = kBlock / 2; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) gW[i] = accumulate ? (float)((double)gW[i] + sh[0]) : (float)sh[0];
}

constexpr int kHdWaves = 4096;                                                  // wavefront partials of the weight gradient (1024 workgroups)

}  // namespace

extern "C" {

int ss_head_packed_supported(int Cin, int Cout, int k)
{
    return k == 3 && Cout == 1 && (Cin == 32 || Cin == 64);
}

long long ss_head_wgrad_packed_ws_floats(int Cin)
{
    return (Cin == 32 || Cin == 64) ? (long long)kHdWaves * Cin * kHdTaps : 0;
}

int ss_head_proj_packed_f32(const unsigned int* x_packed, const float* Wt, float* P, long long rows, int Cin, void* stream)
{
    if (!x_packed || !Wt || !P || rows <= 0 || !ss_head_packed_supported(Cin, 1, 3)) return SS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long blocks = (rows + 31) / 32;
    const unsigned grid = (unsigned)((blocks + 3) / 4 < 8192 ? (blocks + 3) / 4 : 8192);
    if (Cin == 32) hipLaunchKernelGGL((head_proj_packed_kernel<32>), dim3(grid), dim3(kBlock), 0, s, x_packed, Wt, P, rows);
    else hipLaunchKernelGGL((head_proj_packed_kernel<64>), dim3(grid), dim3(kBlock), 0, s, x_packed, Wt, P, rows);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_head_wgrad_packed_f32(const unsigned int* x_packed, const float* g_P, float* g_Wt, float* ws, long long rows, int Cin, int accumulate, void* stream)
{
    if (!x_packed || !g_P || !g_Wt || !ws || rows <= 0 || !ss_head_packed_supported(Cin, 1, 3)) return SS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long rpw = (rows + kHdWaves - 1) / kHdWaves;
    const long long n_waves = (rows + rpw - 1) / rpw;
    const unsigned grid = (unsigned)((n_waves + 3) / 4);
    // wavefronts beyond n_waves (the last workgroup) see an empty row range and write zero partials inside the kHdWaves-sized workspace
    if (Cin == 32) hipLaunchKernelGGL((head_wgrad_packed_kernel<32>), dim3(grid), dim3(kBlock), 0, s, x_packed, g_P, ws, rows, rpw);
    else hipLaunchKernelGGL((head_wgrad_packed_kernel<64>), dim3(grid), dim3(kBlock), 0, s, x_packed, g_P, ws, rows, rpw);
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    hipLaunchKernelGGL(head_wgrad_finish_kernel, dim3(Cin * kHdTaps), dim3(kBlock), 0, s, ws, g_Wt, (long long)grid * 4, Cin * kHdTaps, accumulate);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

}  // extern "C"

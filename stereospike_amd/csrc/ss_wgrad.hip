// ss_wgrad.hip — exact bf16x3 MFMA contractions on spike operands (im2col / split operand preparation, weight gradients, fused decoder backward), the six-term dense GEMM, packed-spike readers + their C-ABI entry points (include/ss_neuron.h).
#include "ss_common.hpp"
#include <stdlib.h>

namespace {

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

// Epilogue of the exact bf16x3 weight-gradient GEMM of an encoder / bottleneck convolution (fused._SpikeConvCL.backward): the library GEMM leaves
// parts[S][K = (ky, kx, ci)][3 terms][C_out] (split-K slices x the three bf16 terms of the output gradient); this kernel sums slices and terms in a fixed
// order (slice-major, terms hi, mid, lo inside) and writes the Conv2d layout g_W[C_out][C_in][k][k] — one launch instead of two torch reductions and a
// permuting copy.  A workgroup owns 32 output channels x 8 input channels x all taps: reads are 128-byte rows along C_out, the LDS tile turns them into
// the 4 kk * 8-byte contiguous runs of the output.
constexpr int kWr3Co = 32, kWr3Ci = 8;
__global__ __launch_bounds__(kBlock) void wgrad_reduce3_kernel(const float* __restrict__ parts, float* __restrict__ gw, int S, int kk, int Cin, int Cout)
{
    extern __shared__ float wr3_tile[];                                       // [co 32][ci 8 * kk + 1]
    const int cib = Cin / kWr3Ci;
    const int co0 = (int)(blockIdx.x / cib) * kWr3Co, ci0 = (int)(blockIdx.x % cib) * kWr3Ci;
    const int row = kWr3Ci * kk + 1;
    const long long slice = (long long)kk * Cin * 3 * Cout;
    for (int it = threadIdx.x; it < kk * kWr3Ci * kWr3Co; it += kBlock) {
        const int co = it % kWr3Co, r = it / kWr3Co, ci = r % kWr3Ci, tap = r / kWr3Ci;
        const float* src = parts + ((long long)(tap * Cin + ci0 + ci) * 3) * Cout + co0 + co;
        float a = 0.f;
        for (int sl = 0; sl < S; ++sl) {
            const float* q = src + sl * slice;
            a += (q[0] + q[Cout]) + q[2 * Cout];
        }
        wr3_tile[co * row + ci * kk + tap] = a;
    }
    __syncthreads();
    const int run = kWr3Ci * kk;                                             // contiguous floats per output channel
    for (int it = threadIdx.x; it < kWr3Co * run; it += kBlock) {
        const int co = it / run, e = it - co * run;
        gw[((long long)(co0 + co) * Cin + ci0) * kk + e] = wr3_tile[co * row + e];
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

// im2col_cl_bf16_kernel reading its NHWC input from a packed spike tensor (8 channels = 16 bits of one word, C % 8 == 0), or from a DENSE 16-bit NHWC array
// (any values: an activation gradient — a plain gather of 16-byte granules; the data gradient of a 3 x 3 / stride 1 / padding 1 convolution is the same
// convolution of g with the flipped, channel-transposed kernel: the caller flips the weight matrix, the patch matrix is the plain one).
// DT: the operand format of the patch matrix built from spikes (0 / SS_DT_BF16: bf16; SS_DT_F16: fp16 — the 16-bit activation modes' single-term GEMMs).
// Round 6: a workgroup owns kI2cRows consecutive patch rows and its threads walk the row's granules — 24 .. 32 granules per thread instead of ONE (the
// one-granule form launched 3.6e5 workgroups of four short-lived wavefronts for a bottleneck patch matrix at config 5's share and wrote at 2.6 TB/s); the
// rows' (frame, oy, ox) are wave-uniform scalars, a granule's (tap, channel group) is computed once per thread and granule column; four loads in flight.
constexpr int kI2cRows = 8;
template <int DT, bool PACKED>
__global__ __launch_bounds__(kBlock) void im2col_rows_kernel(const void* __restrict__ xin, unsigned short* __restrict__ A,
                                                             int h, int w, int C, int k, int stride, int pad, int ho, int wo, unsigned rows)
{
    const unsigned C8 = (unsigned)C / 8, per_row = (unsigned)(k * k) * C8;
    const unsigned row0 = blockIdx.x * kI2cRows;
    int nbq[kI2cRows], iy0[kI2cRows], ix0[kI2cRows];                              // wave-uniform: frame, top-left input pixel of the row's patch
#pragma unroll
    for (int j = 0; j < kI2cRows; ++j) {
        const unsigned row = min(row0 + j, rows - 1);
        const unsigned ox = row % (unsigned)wo, t = row / (unsigned)wo;
        nbq[j] = (int)(t / (unsigned)ho);
        iy0[j] = (int)((t % (unsigned)ho) * stride) - pad;
        ix0[j] = (int)(ox * stride) - pad;
    }
    for (unsigned r = threadIdx.x; r < per_row; r += kBlock) {
        const unsigned tap = r / C8, c8 = r - tap * C8;
        const int ky = (int)(tap / (unsigned)k), kx = (int)tap - ky * k;
        unsigned short* const dst = A + ((long long)row0 * (k * k) + tap) * C + c8 * 8;
#pragma unroll
        for (int j0 = 0; j0 < kI2cRows; j0 += 4) {
            u16x8 o[4];
            [[maybe_unused]] unsigned bits[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iy = iy0[j0 + j] + ky, ix = ix0[j0 + j] + kx;
                const bool ok = iy >= 0 && iy < h && ix >= 0 && ix < w;
                const long long e = (((long long)nbq[j0 + j] * h + (ok ? iy : 0)) * w + (ok ? ix : 0)) * C + c8 * 8;
                if constexpr (PACKED) {
                    bits[j] = ok ? (static_cast<const unsigned*>(xin)[e >> 4] >> (2 * (int)(e & 15))) & 0xFFFFu : 0u;       // code 0 -> 0.0
                } else {
                    o[j] = (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
                    if (ok) o[j] = *reinterpret_cast<const u16x8*>(static_cast<const unsigned short*>(xin) + e);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (PACKED) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) o[j][q] = code_to_op<DT>((bits[j] >> (2 * q)) & 3u);
                }
                if (row0 + j0 + j < rows) *reinterpret_cast<u16x8*>(dst + (long long)(j0 + j) * (k * k) * C) = o[j];
            }
        }
    }
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
// DT != 0 (16-bit activation modes): g is ONE term (it is stored in the operand format), one MFMA per (k-step, virtual-channel tile)
template <int CIT, int COT, int NVC, int DT = 0>
__global__ __launch_bounds__(kSwThreads) void spike_conv_wgrad_kernel(const unsigned short* __restrict__ gT, const unsigned short* __restrict__ xK,
                                                                     float* __restrict__ ws, int NB, int h, int ho, int wo, int Q)
{
    constexpr int NSP = DT ? 1 : 3;
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
    s16x8 gs[PF][NSP], xn[PF][NVC];
    auto load_step = [&](s16x8 (&gd)[NSP], s16x8 (&xd)[NVC], long long ks) {
        const int c = (int)(ks % KSR);
        const long long ro = ks / KSR;                                     // nb * ho + oy
        const int oy = (int)(ro % ho);
        const long long nb = ro / ho;
#pragma unroll
        for (int sp = 0; sp < NSP; ++sp) gd[sp] = *reinterpret_cast<const s16x8*>(gT + (((ks * NSP + sp) * COT + cot) * 64 + lane) * 8);
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
                for (int sp = 0; sp < NSP; ++sp)
#pragma unroll
                    for (int j = 0; j < NVC; ++j)
                        if (own[j]) acc[j] = mfma32<DT>(gs[u][sp], xn[u][j], acc[j]);
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
// DT != 0: g is a 16-bit tensor already in the operand format — ONE term, a plain re-layout
template <int DT = 0>
__global__ __launch_bounds__(kBlock) void spike_conv_gprep_kernel(const typename ActT<DT>::type* __restrict__ g, unsigned short* __restrict__ gT, long long rows, int wo,
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
        if constexpr (DT != 0) {
            u16x8 o1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ox = 16 * c + 8 * (lane >> 5) + e;
                o1[e] = ox < wo ? g[(ro * wo + ox) * COUT + 32 * t + (lane & 31)] : (unsigned short)0;
            }
            *reinterpret_cast<u16x8*>(gT + ((ks * COT + t) * 64 + lane) * 8) = o1;
            continue;
        }
        u16x8 o[3];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ox = 16 * c + 8 * (lane >> 5) + e;
            const float v = ox < wo ? (float)g[(ro * wo + ox) * COUT + 32 * t + (lane & 31)] : 0.f;
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
template <bool PACKED, int DT = 0>             // DT != 0: the operand format is DT; a dense input is then the 16-bit spike tensor itself
__global__ __launch_bounds__(kBlock) void spike_conv_xprep_kernel(const void* __restrict__ xv, unsigned short* __restrict__ xK, int NB, int h, int w, int C,
                                                                  int wo)
{
    const float* x = static_cast<const float*>(xv);
    const unsigned short* x16 = static_cast<const unsigned short*>(xv);
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
            if constexpr (PACKED) v[t] = ok ? code_to_op<DT>((xp[el >> 4] >> (2 * (int)(el & 15))) & 3u) : (unsigned short)0;
            else if constexpr (DT != 0) v[t] = ok ? x16[el] : (unsigned short)0;
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
// The same weight gradient straight from the 2-BIT PACKED spike tensor, second form (round 5): no operand-preparation passes at all
// ---------------------------------------------------------------------------------------------------
// spike_conv_wgrad_kernel reads every fragment from global memory: five column-decimated bf16 copies of x (spike_conv_xprep_kernel writes 1.15 GB for conv1 at
// BASELINE config 3: 0.28 ms, HBM-bound) and a fragment-ordered copy of g (spike_conv_gprep_kernel), and it is bound by the vector-memory path (a 1 KB fragment
// per MFMA in the 16-bit modes: the L1 rate).  Here a workgroup stages what a tile of 2 output rows x 32 output columns touches — the 7 x 67-pixel window of x
// (32 input channels, expanded from the packed words to the operand format) and the 64 x 64 tile of g (16-bit modes: as stored; fp32 mode: split exactly into
// three bf16 planes while staged) — ONCE in LDS, both pixel-major as they lie in HBM, and every MFMA fragment — 8 consecutive output pixels of one channel,
// i.e. the arrays read against their grain, x with a pixel stride of 2 — is two ds_read_b64_tr_b16: the transposed LDS read hands lane i of a 16-lane group
// column i of the 4 x 16 block whose rows the group's lanes address individually (out[i][r] = in[lane 4 r + i / 4][element i % 4], profiles/r04/tr16.log).
//   * kind = (32 input channels, 64 output channels); wavefront w of 8 = output-channel tile w & 1 and the taps (w >> 1) + 4 j, j < 7: 7 | 6 accumulator tiles
//     (D[co][ci], the first form's orientation and partial-sum layout: spike_conv_wgrad_reduce_kernel finishes) that stay in registers over the slice's tiles;
//   * a k-step = 16 consecutive ox of one output row: 2 NSP transposed reads for g, 2 per tap for x, 7 NSP MFMAs;
//   * LDS is double-buffered (one barrier per tile): the next tile's packed words and g granules are fetched into registers before this tile's MFMAs and
//     expanded / stored after them;
//   * bank conflicts of the transposed reads (4 pixels x 64 B of one pixel row per 32-lane pass): x pixels sit 128 B apart (stride 2), g pixels 128 B — pixel
//     pairs would meet in the same banks; x swaps the two pixels of a column pair where bit 2 of the column is set, g swaps the 64-byte halves of pixels with
//     bit 1 set: the four pixels of a pass cover the 64 banks once.
constexpr int kTwThreads = 512;
constexpr int kTwTR = 2, kTwTC = 32;                                             // output rows x columns of a tile
constexpr int kTwWR = 2 * kTwTR + 3, kTwWC = 2 * kTwTC + 3;                      // 7 x 67 window pixels
constexpr int kTwXRow = 68 * 64;                                                // bytes of a window row: 68 pixel slots x 32 channels x 2 B
constexpr int kTwXBytes = kTwWR * kTwXRow;                                       // 30 464
constexpr int kTwGPlane = kTwTR * kTwTC * 128;                                   // 8 192: 64 pixels x 64 output channels x 2 B
constexpr int kTwXItems = kTwWR * kTwWC * 2, kTwXIter = (kTwXItems + kTwThreads - 1) / kTwThreads;      // (pixel, 16-code word) items per tile, per thread
static_assert(kTwTR * kTwTC * 8 == kTwThreads, "one 8-channel granule of the g tile per thread");

template <int DT>
__global__ __launch_bounds__(kTwThreads, 1) void spike_conv_wgrad_tr_kernel(const typename ActT<DT>::type* __restrict__ g, const unsigned* __restrict__ xp,
                                                                            float* __restrict__ ws, int NB, int h, int w, int ho, int wo, int CIN, int COUT, int slices)
{
    constexpr int NSP = DT ? 1 : 3;
    __shared__ __attribute__((aligned(16))) unsigned char xw[2][kTwXBytes];
    __shared__ __attribute__((aligned(16))) unsigned char gw[2][NSP * kTwGPlane];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int CIT = CIN / 32;
    const int kind = (int)(blockIdx.x / slices), slice = (int)(blockIdx.x % slices);        // kinds of one slice: blockIdx apart by `slices` — the same XCD when 8 | slices
    const int cit = kind % CIT, cp = kind / CIT;                                             // 32-input-channel tile, 64-output-channel pair
    const int cot_l = wave & 1, tq = wave >> 1;
    const int RG = (ho + kTwTR - 1) / kTwTR, CG = (wo + kTwTC - 1) / kTwTC;
    const long long n_tiles = (long long)NB * RG * CG;
    const long long t_begin = n_tiles * slice / slices, t_end = n_tiles * (slice + 1) / slices;
    // this lane as a SOURCE lane of the transposed reads: pixel r_s of a 4-pixel block, channels 4 j_s .. + 3 of its 16-lane group's 16-channel half
    const int r_s = (lane & 15) >> 2, j_s = lane & 3, hsel = (lane >> 4) & 1, khalf = lane >> 5;
    int xoff[7];
    bool own[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int tap = tq + 4 * j;
        own[j] = tap < 25;                                                                   // wave-uniform
        const int ky = own[j] ? tap / 5 : 0, kx = own[j] ? tap - 5 * (tap / 5) : 0;
        const int cl = 2 * r_s + kx;                                                         // low part of the window column; + 16 khalf + 8 q + 32 half (no effect on bits 0..2)
        xoff[j] = ky * kTwXRow + ((cl ^ ((cl >> 2) & 1)) + 16 * khalf) * 64 + (16 * hsel + 4 * j_s) * 2;
    }
    const int gbase = (8 * khalf + r_s) * 128 + (((32 * cot_l + 16 * hsel + 4 * j_s) * 2) ^ ((r_s >> 1) << 6));
    f32x16 acc[7];
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // ---- staging: registers of the NEXT tile
    unsigned xr[kTwXIter];
    u16x8 gr16;
    f4 gra, grb;
    auto fetch = [&](long long tl) {
        const int cg = (int)(tl % CG);
        const long long q = tl / CG;
        const int rg = (int)(q % RG);
        const long long nb = q / RG;
        const int oy0 = kTwTR * rg, ox0 = kTwTC * cg;
        const int iy0 = 2 * oy0 - 2, ix0 = 2 * ox0 - 2;
#pragma unroll
        for (int u = 0; u < kTwXIter; ++u) {
            const int i = threadIdx.x + kTwThreads * u;
            const int pix = i >> 1, jw = i & 1;
            const int wr = pix / kTwWC, wc = pix - wr * kTwWC;
            const int iy = iy0 + wr, ix = ix0 + wc;
            xr[u] = 0u;
            if (i < kTwXItems && iy >= 0 && iy < h && ix >= 0 && ix < w) xr[u] = xp[(((nb * h + iy) * w + ix) * CIN + 32 * cit) / 16 + jw];
        }
        const int p = threadIdx.x >> 3, gran = threadIdx.x & 7;
        const int oy = oy0 + (p >> 5), ox = ox0 + (p & 31);
        const bool ok = oy < ho && ox < wo;
        const long long el = ((nb * ho + oy) * wo + ox) * COUT + 64 * cp + 8 * gran;
        if constexpr (DT != 0) {
            gr16 = (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (ok) gr16 = *reinterpret_cast<const u16x8*>(g + el);
        } else {
            gra = (f4){0.f, 0.f, 0.f, 0.f}; grb = gra;
            if (ok) { gra = *reinterpret_cast<const f4*>(g + el); grb = *reinterpret_cast<const f4*>(g + el + 4); }
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int u = 0; u < kTwXIter; ++u) {
            const int i = threadIdx.x + kTwThreads * u;
            const int pix = i >> 1, jw = i & 1;
            const int wr = pix / kTwWC, wc = pix - wr * kTwWC;
            if (i < kTwXItems) {
                unsigned char* const pp = xw[buf] + wr * kTwXRow + (wc ^ ((wc >> 2) & 1)) * 64 + 32 * jw;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    u16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = code_to_op<DT>((xr[u] >> (2 * (8 * q + e))) & 3u);
                    *reinterpret_cast<u16x8*>(pp + 16 * q) = o;
                }
            }
        }
        const int p = threadIdx.x >> 3, gran = threadIdx.x & 7;
        unsigned char* const gp = gw[buf] + p * 128 + ((gran * 16) ^ (((p >> 1) & 1) << 6));
        if constexpr (DT != 0) {
            *reinterpret_cast<u16x8*>(gp) = gr16;
        } else {
            u16x8 o[3];
#pragma unroll
            for (int e = 0; e < 8; ++e) {                                                    // truncation split: every residual is exactly representable, g = hi + mid + lo
                const float v = e < 4 ? gra[e] : grb[e - 4];
                const unsigned uh = __float_as_uint(v) & 0xFFFF0000u;
                const float r1 = v - __uint_as_float(uh);
                const unsigned um = __float_as_uint(r1) & 0xFFFF0000u;
                const float r2 = r1 - __uint_as_float(um);
                o[0][e] = (unsigned short)(uh >> 16); o[1][e] = (unsigned short)(um >> 16); o[2][e] = (unsigned short)(__float_as_uint(r2) >> 16);
            }
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) *reinterpret_cast<u16x8*>(gp + sp * kTwGPlane) = o[sp];
        }
    };
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef s16x4 __attribute__((address_space(3))) * lds4_t;
    int cur = 0;
    if (t_begin < t_end) { fetch(t_begin); commit(0); }
#pragma unroll 1
    for (long long tl = t_begin; tl < t_end; ++tl) {
        __syncthreads();                                                                     // buffer `cur` is complete; every reader of the other one (the previous tile) is done
        const bool has_next = tl + 1 < t_end;
        if (has_next) fetch(tl + 1);
        const unsigned char* const xb = xw[cur];
        const unsigned char* const gb = gw[cur] + gbase;
#pragma unroll
        for (int ks = 0; ks < 2 * kTwTR; ++ks) {
            const int rr = ks >> 1, hh = ks & 1;
            s16x8 ga[NSP];
#pragma unroll
            for (int sp = 0; sp < NSP; ++sp) {
                const unsigned char* const a = gb + sp * kTwGPlane + (32 * rr + 16 * hh) * 128;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(a));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(a + 4 * 128));
                ga[sp] = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                if (own[j]) {
                    const unsigned char* const b = xb + xoff[j] + 2 * rr * kTwXRow + 32 * hh * 64;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(b));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(b + 8 * 64));
                    const s16x8 xf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                    for (int sp = NSP - 1; sp >= 0; --sp) acc[j] = mfma32<DT>(ga[sp], xf, acc[j]);         // smallest term first
                }
            }
        }
        if (has_next) commit(cur ^ 1);
        cur ^= 1;
    }
    // D[co][ci]: column (ci) = lane & 31, row (co) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5);  ws[slice][virtual channel v * 32 + ci][co], v = tap * CIT + cit
    float* const wsl = ws + (long long)slice * 25 * CIN * COUT;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        if (own[j]) {
            const int v = (tq + 4 * j) * CIT + cit;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 64 * cp + 32 * cot_l + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                wsl[((long long)v * 32 + (lane & 31)) * COUT + co] = acc[j][r];
            }
        }
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


}  // namespace

extern "C" {

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
    hipLaunchKernelGGL((im2col_rows_kernel<0, true>), dim3((unsigned)((rows + kI2cRows - 1) / kI2cRows)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), static_cast<const void*>(x_packed), static_cast<unsigned short*>(A), h, w, C, k, stride, pad, ho, wo, (unsigned)rows);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

/* ss_im2col_cl_bf16_packed with the patch matrix in `dtype` (ABI 9: SS_DT_F16 for the fp16 activation mode's single-term GEMMs; SS_DT_BF16 == the bf16 form) */
int ss_im2col_cl_packed_x16(const unsigned int* x_packed, void* A, long long NB, int h, int w, int C, int k, int stride, int pad, int ho, int wo, int dtype, void* stream)
{
    if (dtype == SS_DT_BF16) return ss_im2col_cl_bf16_packed(x_packed, A, NB, h, w, C, k, stride, pad, ho, wo, stream);
    if (dtype != SS_DT_F16) return SS_EINVAL;
    if (!x_packed || !A || NB <= 0 || h <= 0 || w <= 0 || C <= 0 || C % 8 != 0 || k <= 0 || stride <= 0 || pad < 0 || ho <= 0 || wo <= 0) return SS_EINVAL;
    if ((ho - 1) * stride - pad + k - 1 >= h + pad || (wo - 1) * stride - pad + k - 1 >= w + pad) return SS_EINVAL;
    if (!aligned16(A) || (NB * h * w * C) % 16 != 0) return SS_EINVAL;
    const long long rows = NB * ho * wo;
    const long long per_row = (long long)k * k * (C / 8);
    if (rows > 0x7fffffffLL || per_row > 65535LL * kBlock) return SS_EINVAL;
    hipLaunchKernelGGL((im2col_rows_kernel<SS_DT_F16, true>), dim3((unsigned)((rows + kI2cRows - 1) / kI2cRows)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), static_cast<const void*>(x_packed), static_cast<unsigned short*>(A), h, w, C, k, stride, pad, ho, wo, (unsigned)rows);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

/* im2col of a dense 16-bit NHWC array (fp16 or bf16: the element type is not interpreted) into a patch matrix A[(nb, oy, ox)][(ky, kx, c)] of the same type (ABI 9) */
int ss_im2col_cl_x16(const void* x, void* A, long long NB, int h, int w, int C, int k, int stride, int pad, int ho, int wo, void* stream)
{
    if (!x || !A || NB <= 0 || h <= 0 || w <= 0 || C <= 0 || C % 8 != 0 || k <= 0 || stride <= 0 || pad < 0 || ho <= 0 || wo <= 0) return SS_EINVAL;
    if ((ho - 1) * stride - pad + k - 1 >= h + pad || (wo - 1) * stride - pad + k - 1 >= w + pad) return SS_EINVAL;
    if (!aligned16(x) || !aligned16(A)) return SS_EINVAL;
    const long long rows = NB * ho * wo;
    const long long per_row = (long long)k * k * (C / 8);
    if (rows > 0x7fffffffLL || per_row > 65535LL * kBlock) return SS_EINVAL;
    hipLaunchKernelGGL((im2col_rows_kernel<0, false>), dim3((unsigned)((rows + kI2cRows - 1) / kI2cRows)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), x, static_cast<unsigned short*>(A), h, w, C, k, stride, pad, ho, wo, (unsigned)rows);
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

int ss_wgrad_reduce3_f32(const float* parts, float* g_w, int S, int k, int Cin, int Cout, void* stream)
{
    if (!parts || !g_w || S < 1 || k < 1 || k > 7 || Cin <= 0 || Cout <= 0 || Cin % kWr3Ci != 0 || Cout % kWr3Co != 0) return SS_EINVAL;
    const int kk = k * k;
    const size_t lds = (size_t)kWr3Co * (kWr3Ci * kk + 1) * sizeof(float);
    hipLaunchKernelGGL(wgrad_reduce3_kernel, dim3((unsigned)((Cout / kWr3Co) * (Cin / kWr3Ci))), dim3(kBlock), lds, static_cast<hipStream_t>(stream),
                       parts, g_w, S, kk, Cin, Cout);
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

// A/B switch (tools/): SS_SPIKE_WGRAD_TR=0 keeps the first form (xprep + gprep + global-memory fragments) on packed input too — read once
static bool spike_conv_wgrad_tr_on()
{
    static const char* const e = getenv("SS_SPIKE_WGRAD_TR");
    return !(e && e[0] == '0');
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

/* workspace of the packed-input (window / transposed-read) form alone: its per-slice partial sums — 0 when that form is off (SS_SPIKE_WGRAD_TR=0) or the shape is
   not supported; then ss_spike_conv_wgrad_ws_floats applies.  (ABI 10; ADVICE r05: the full figure also covers the first form's five decimated copies of x and
   the fragment-ordered g — 1.9 GB at conv1 / config 3 — which the packed form never touches.) */
long long ss_spike_conv_wgrad_tr_ws_floats(int Cin, int Cout)
{
    int Q = 0, slices = 0;
    if (!spike_conv_wgrad_tr_on() || !ss_spike_conv_wgrad_supported(Cin, Cout, 5, 2, 2) || !spike_conv_wgrad_plan(Cin, &Q, &slices)) return 0;
    const int kinds = (Cin / 32) * (Cout / 64), sl = slices * Q / kinds > 0 ? slices * Q / kinds : 1;
    return (long long)sl * 25 * Cin * Cout + 8;
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
    if (x_packed && spike_conv_wgrad_tr_on()) {                                  // packed input: the window / transposed-read form, no preparation passes
        const int kinds = (Cin / 32) * (Cout / 64), sl = slices * Q / kinds > 0 ? slices * Q / kinds : 1;
        if ((long long)sl * 25 * Cin * Cout > ss_spike_conv_wgrad_ws_floats(Cin, Cout, NB, h, w)) return SS_EINVAL;
        hipLaunchKernelGGL(spike_conv_wgrad_tr_kernel<0>, dim3(kinds * sl), dim3(kTwThreads), 0, s, g, x_packed, ws, (int)NB, h, w, ho, wo, Cin, Cout, sl);
        if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
        hipLaunchKernelGGL(spike_conv_wgrad_reduce_kernel, dim3(grid_for(25LL * Cin * Cout, 1024)), dim3(kBlock), 0, s, ws, g_w, sl, Cin, Cout, accumulate);
        return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
    }
    if (x_packed) hipLaunchKernelGGL(spike_conv_xprep_kernel<true>, dim3(grid_for(NB * (h + 4) * (oxp / 8) * Cin, kMaxGridBwd)), dim3(kBlock), 0, s,
                                     static_cast<const void*>(x_packed), xK, (int)NB, h, w, Cin, wo);
    else hipLaunchKernelGGL(spike_conv_xprep_kernel<false>, dim3(grid_for(NB * (h + 4) * (oxp / 8) * Cin, kMaxGridBwd)), dim3(kBlock), 0, s,
                            static_cast<const void*>(x), xK, (int)NB, h, w, Cin, wo);
    hipLaunchKernelGGL(spike_conv_gprep_kernel<0>, dim3(grid_for(NB * ho * ksr * (Cout / 32) * 64, kMaxGridBwd)), dim3(kBlock), 0, s, g, gT, NB * ho, wo, Cout);
    const unsigned grid = (unsigned)(Q * slices);
    if (Cin == 32) hipLaunchKernelGGL((spike_conv_wgrad_kernel<1, 2, 7>), dim3(grid), dim3(kSwThreads), 0, s, gT, xK, ws, (int)NB, h, ho, wo, Q);
    else hipLaunchKernelGGL((spike_conv_wgrad_kernel<2, 4, 7>), dim3(grid), dim3(kSwThreads), 0, s, gT, xK, ws, (int)NB, h, ho, wo, Q);
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    hipLaunchKernelGGL(spike_conv_wgrad_reduce_kernel, dim3(grid_for(25LL * Cin * Cout, 1024)), dim3(kBlock), 0, s, ws, g_w, slices, Cin, Cout, accumulate);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

/* ss_spike_conv_wgrad_f32 on a 16-bit output gradient (ABI 9): g in `dtype` (the operand itself: one term), x the dense 16-bit spike tensor of the same
   dtype or the 2-bit packed one, g_w fp32 (exact products, fp32 accumulation, fixed-order reduction); ws as for the fp32 form */
int ss_spike_conv_wgrad_x16(const void* g, const void* x, const unsigned int* x_packed, float* g_w, float* ws, long long NB, int Cin, int Cout, int h,
                            int w, int accumulate, int dtype, void* stream)
{
    if (x_packed && (NB * h * w * Cin) % 16 != 0) return SS_EINVAL;
    if (!g || (!x && !x_packed) || !g_w || !ws || NB <= 0 || NB > 0x7fffffff || h <= 0 || w <= 0 || !ss_spike_conv_wgrad_supported(Cin, Cout, 5, 2, 2) || !aligned16(ws))
        return SS_EINVAL;
    if (dtype != SS_DT_F16 && dtype != SS_DT_BF16) return SS_EINVAL;
    int Q = 0, slices = 0;
    if (!spike_conv_wgrad_plan(Cin, &Q, &slices)) return SS_ELAUNCH;
    const int ho = (h + 4 - 5) / 2 + 1, wo = (w + 4 - 5) / 2 + 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long ksr = (wo + 15) / 16, oxp = ksr * 16;
    const long long part = (long long)slices * 25 * Cin * Cout, xk = ((5LL * NB * (h + 4) * Cin * oxp + 1) / 2 + 3) & ~3LL;
    unsigned short* xK = reinterpret_cast<unsigned short*>(ws + part);
    unsigned short* gT = reinterpret_cast<unsigned short*>(ws + part + xk);
    if (x_packed && spike_conv_wgrad_tr_on()) {
        const int kinds = (Cin / 32) * (Cout / 64), sl = slices * Q / kinds > 0 ? slices * Q / kinds : 1;
        if ((long long)sl * 25 * Cin * Cout > ss_spike_conv_wgrad_ws_floats(Cin, Cout, NB, h, w)) return SS_EINVAL;
        const unsigned short* gg16 = static_cast<const unsigned short*>(g);
        if (dtype == SS_DT_F16) hipLaunchKernelGGL(spike_conv_wgrad_tr_kernel<SS_DT_F16>, dim3(kinds * sl), dim3(kTwThreads), 0, s, gg16, x_packed, ws, (int)NB, h, w, ho, wo, Cin, Cout, sl);
        else hipLaunchKernelGGL(spike_conv_wgrad_tr_kernel<SS_DT_BF16>, dim3(kinds * sl), dim3(kTwThreads), 0, s, gg16, x_packed, ws, (int)NB, h, w, ho, wo, Cin, Cout, sl);
        if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
        hipLaunchKernelGGL(spike_conv_wgrad_reduce_kernel, dim3(grid_for(25LL * Cin * Cout, 1024)), dim3(kBlock), 0, s, ws, g_w, sl, Cin, Cout, accumulate);
        return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
    }
    const int xg = grid_for(NB * (h + 4) * (oxp / 8) * Cin, kMaxGridBwd), gg = grid_for(NB * ho * ksr * (Cout / 32) * 64, kMaxGridBwd);
    const unsigned grid = (unsigned)(Q * slices);
    const unsigned short* g16 = static_cast<const unsigned short*>(g);
#define SS_SCW16(DTT) do { \
        if (x_packed) hipLaunchKernelGGL((spike_conv_xprep_kernel<true, DTT>), dim3(xg), dim3(kBlock), 0, s, static_cast<const void*>(x_packed), xK, (int)NB, h, w, Cin, wo); \
        else hipLaunchKernelGGL((spike_conv_xprep_kernel<false, DTT>), dim3(xg), dim3(kBlock), 0, s, x, xK, (int)NB, h, w, Cin, wo); \
        hipLaunchKernelGGL(spike_conv_gprep_kernel<DTT>, dim3(gg), dim3(kBlock), 0, s, g16, gT, NB * ho, wo, Cout); \
        if (Cin == 32) hipLaunchKernelGGL((spike_conv_wgrad_kernel<1, 2, 7, DTT>), dim3(grid), dim3(kSwThreads), 0, s, gT, xK, ws, (int)NB, h, ho, wo, Q); \
        else hipLaunchKernelGGL((spike_conv_wgrad_kernel<2, 4, 7, DTT>), dim3(grid), dim3(kSwThreads), 0, s, gT, xK, ws, (int)NB, h, ho, wo, Q); } while (0)
    if (dtype == SS_DT_F16) SS_SCW16(SS_DT_F16); else SS_SCW16(SS_DT_BF16);
#undef SS_SCW16
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    hipLaunchKernelGGL(spike_conv_wgrad_reduce_kernel, dim3(grid_for(25LL * Cin * Cout, 1024)), dim3(kBlock), 0, s, ws, g_w, slices, Cin, Cout, accumulate);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

}  // extern "C"

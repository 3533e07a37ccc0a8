// ss_conv_dgrad.hip — DATA gradient of the stride-2 5x5 encoder convolutions as a six-term bf16 implicit GEMM on the matrix cores
// (include/ss_neuron.h: ss_conv_s2_dgrad_f32).
//
// Reference: autograd's backward of conv1 .. conv4 of the encoder w.r.t. their INPUT — nn.Conv2d(C, 2C, kernel_size=5, stride=2, padding=2,
// bias=False) (/root/reference/network/SNN_models.py:80-101, 268-289, 455-476; torch's convolution_backward in the reference).  Both operands
// are dense fp32 (the neuron backward's g_x and the weight), so until now this was MIOpen's fp32 implicit GEMM (igemm_bwd_gtcx35_nhwc_fp32):
// 1.5 - 2.0 ms per layer at BASELINE config 3, AT the fp32-MFMA rate (92 - 119 of 157 TFLOP/s), 7 ms of a 42 ms step.  Here both operands are
// split into three bf16 terms and the SIX cross terms of ss_gemm6_f32 are kept (hh, hm, mh, hl, lh, mm: relative error 2^-24 per product —
// fp32-product accuracy, fp32 accumulation) on v_mfma_f32_32x32x16_bf16: 6 MFMAs per fp32 product at 16x the rate.
//
//   g_x[nb][iy][ix][ci] = sum_{ky,kx,co} g[nb][(iy + 2 - ky) / 2][(ix + 2 - kx) / 2][co] * W[co][ci][ky][kx]     over the taps with
//                         iy + 2 - ky and ix + 2 - kx even and inside the output map
//
// A stride-2 transposed convolution decomposes by the PARITY of the input pixel: pixel (2 j + py, 2 i + px) only sees the taps with
// ky = py, kx = px (mod 2) — 9 / 6 / 6 / 4 taps for the four classes — at the output pixels (j + dy, i + dx), dy = (py + 2 - ky) / 2 in
// {-1, 0, 1}: four stride-1 convolutions over g that share one window.  GEMM view per class: M = the class's pixels, N = C_in,
// K = (tap of the class, co).
//   * Rows are counted in a PADDED row space: every frame owns ho + 1 rows, the last one a zero row — the vertical padding of this frame and
//     of the next — so tiles run across frame boundaries (no ragged row tiles: the small maps of conv3 / conv4 would waste a third of the
//     matrix cores on them).
//   * A workgroup (4 wavefronts, two per CU) owns 4 RB padded rows x CB columns of (j, i) (RB * CB = 32 = the MFMA's M); wavefront = RB rows; it
//     carries the accumulators of all four classes (4 x NT tiles of 32 input channels).
//   * The window of g (4 RB + 2 rows x CB + 2 columns, 32 output channels at a time) is split ONCE per element while it is staged: three bf16
//     planes in LDS, 64-B pixels with swizzled 16-B granules.  An A fragment is then three 16-B LDS reads per lane — no VALU work on the
//     operands in the main loop.
//   * The weight (split once, sign-alternated, in fragment order: conv_s2_dgrad_prep_kernel) streams L2 -> LDS double-buffered, one tap (two
//     k-steps) per stage and barrier.
//   * A tap's 12 NT MFMAs accumulate on a scratch accumulator that starts at zero; the class's running sum takes one fp32 addition per tap.
//     The bf16 MFMA's fp32 accumulation drifts down by ~2^-28 of the magnitude sum per instruction: the weight fragments of
//     every second tap of a class are negated and its scratch sum is subtracted instead of added, which cancels the drift.
// HBM traffic: g once per (C_in / 32 NT) workgroup kinds (neighbours: L2), the weights from L2, g_x once.
#include "ss_common.hpp"
#include <stdlib.h>

namespace {

constexpr int kDgThreads = 256;
#ifndef SS_DG_ABL
#define SS_DG_ABL 0              // timing experiments (tools/_abl_dgrad.sh; results are WRONG with any bit set): 1 no per-tap barrier, 2 no sign flip,
#endif                           // 4 no B-fragment LDS reads, 8 no A-fragment LDS reads, 16 no weight stream (global loads + LDS stores), 32 no window staging, 64 no g_x stores

__host__ __device__ constexpr int dg_cnt(int cls) { return cls == 0 ? 9 : (cls == 3 ? 4 : 6); }     // taps of class (py, px) = (cls >> 1, cls & 1)

// weight [C_out][C_in][5][5] fp32 -> Bf[chunk c of 32 co][tap][g][split][ci tile t][lane][8] bf16: element e of a lane = split term of
// s * W[co = 32 c + 16 g + 8 (lane >> 5) + e][ci = 32 t + (lane & 31)][ky][kx], s = (-1)^n for the n-th tap of its class over all chunks
// DT != 0 (16-bit activation modes): ONE term, the weight rounded once to the operand format, no sign alternation
template <int DT = 0>
__global__ __launch_bounds__(kBlock) void conv_s2_dgrad_prep_kernel(const float* __restrict__ W, unsigned short* __restrict__ Bf, unsigned* __restrict__ counters,
                                                                    int Cin, int Cout)
{
    constexpr int NSP = DT ? 1 : 3;
    if (blockIdx.x == 0 && threadIdx.x < 8) counters[threadIdx.x] = 0u;         // the per-XCD item counters of the main kernel
    const int NTALL = Cin / 32;
    const long long total = (long long)(Cout / 32) * 25 * 2 * NSP * NTALL * 64;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int lane = (int)(i & 63);
        long long r = i >> 6;
        const int t = (int)(r % NTALL); r /= NTALL;
        const int sp = (int)(r % NSP); r /= NSP;
        const int g = (int)(r & 1); r >>= 1;
        const int tap = (int)(r % 25); const int c = (int)(r / 25);
        const int ky = tap / 5, kx = tap - 5 * ky;
        const int cls = (ky & 1) * 2 + (kx & 1);
        const int n = c * dg_cnt(cls) + (ky >> 1) * ((kx & 1) ? 2 : 3) + (kx >> 1);
        const int ci = 32 * t + (lane & 31);
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int co = 32 * c + 16 * g + 8 * (lane >> 5) + e;
            float v = W[(((long long)co * Cin + ci) * 5 + ky) * 5 + kx];
            if (DT == 0 && (n & 1)) v = -v;               // (16-bit modes accumulate on the class sums directly: no scratch sum, no sign)
            if constexpr (DT != 0) { o[e] = round_op<DT>(v); continue; }
            const unsigned short h1 = narrow<SS_DT_BF16>(v);
            const float r1 = v - widen<SS_DT_BF16>(h1);
            const unsigned short h2 = narrow<SS_DT_BF16>(r1);
            const float r2 = r1 - widen<SS_DT_BF16>(h2);
            o[e] = sp == 0 ? h1 : (sp == 1 ? h2 : narrow<SS_DT_BF16>(r2));
        }
        *reinterpret_cast<u16x8*>(Bf + i * 8) = o;
    }
}

// DT != 0 (16-bit activation modes, round 5): g IS the operand (one 16-byte copy per granule into ONE plane), one weight term, one MFMA per (k-step, M block,
// N tile) on the native matrix-core type, g_x narrowed once on store
template <int CO, int NTALL, int NT, int CB, int MB = 1, int DT = 0>
__global__ __launch_bounds__(kDgThreads, 2) void conv_s2_dgrad_kernel(const typename ActT<DT>::type* __restrict__ G, const unsigned short* __restrict__ Bf,
                                                                      typename ActT<DT>::type* __restrict__ gx, unsigned* __restrict__ counters, int NB, int h, int w, int ho, int wo)
{
    constexpr int NSP = DT ? 1 : 3;
    constexpr int CI = 32 * NTALL, KINDS = NTALL / NT, NCH = CO / 32;
    constexpr int RB = 32 / CB, TJR = 4 * RB * MB, WR = TJR + 2, WC = CB + 2, ROWB = WC * 64, PLANE = WR * ROWB;      // MB: M blocks (RB rows each) per wavefront
    static_assert(MB == 1 || (2 * RB) % 4 == 0, "the granule swizzle of a wavefront's M blocks must agree");
    constexpr int TAPB = 2 * NSP * NT * 1024;                                   // bytes of one tap's fragments: 2 k-steps x 3 splits x NT tiles
    constexpr int TPS = DT ? 5 : 1;                                             // taps per weight stage and barrier: one tap with three splits; a whole kernel row (ky) in the 16-bit
    constexpr int STG = TPS * TAPB;                                             // modes, whose taps are 2 NT MFMAs per wavefront — too little work to pay a barrier each
    constexpr int LPT = STG / 16 / kDgThreads, REM = STG / 16 - LPT * kDgThreads;
    constexpr int kItems = WR * WC * 4, kIter = (kItems + kDgThreads - 1) / kDgThreads;     // (window pixel, 8-channel granule)
    __shared__ __attribute__((aligned(16))) unsigned char wnd[NSP * PLANE];
    __shared__ __attribute__((aligned(16))) unsigned char bst[2 * STG];
    const int lane = threadIdx.x & 63;
    const int mb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);            // wavefront = RB padded rows of the tile
    const int tx = lane & 31, half = lane >> 5;
    const int rr = tx / CB, cc = tx - rr * CB;                                  // this lane's pixel of the M block
    const int HP = ho + 1;                                                      // padded rows per frame
    const long long RT = (long long)NB * HP;
    const int RG = (int)((RT + TJR - 1) / TJR), CG = (wo + CB - 1) / CB;
    const long long n_items = (long long)RG * CG * KINDS;
    // Work distribution: the items are cut into 8 contiguous ranges, one per XCD (the dispatcher places workgroup b on XCD b % 8: neighbouring
    // tiles — shared window halos, the same weights — meet in one L2); inside its range a workgroup DRAWS the next item from the range's counter
    // (zeroed by the prep kernel).  A static split leaves 2 or 3 items per workgroup at conv4 (1 080 items over 512 workgroups) and the launch ends
    // with the few CUs whose two workgroups both got 3; drawn items end within one item of each other.  Items are independent: same result bits.
    __shared__ unsigned next_item;
    const unsigned xcd = blockIdx.x & 7u;
    const long long t_begin = n_items * xcd / 8, t_end = n_items * (xcd + 1) / 8;
    // LDS byte offsets of this lane's A fragments (plane 0, dy = 0, first k-step) for dx = -1, 0, 1; the second k-step is ^ 32
    int abase[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int row = RB * MB * mb + rr + 1, col = cc + d;
        abase[d] = row * ROWB + col * 64 + ((half ^ (((col >> 2) + 2 * row) & 3)) << 4);
    }
    f4 st[LPT + 1];
    auto stage_issue = [&](long long stage, int kind) {                         // stage = c * 25 + first tap of the stage
#pragma unroll
        for (int u = 0; u <= LPT; ++u) {
            if (u == LPT && (REM == 0 || (int)threadIdx.x >= REM)) break;
            const int pidx = threadIdx.x + kDgThreads * u;                      // 16-B piece of the stage: [g][split][tile of this kind][lane]
            const int ln = pidx & 63, tt = (pidx >> 6) % NT, gs = (pidx >> 6) / NT;
            st[u] = *reinterpret_cast<const f4*>(Bf + (((stage * (2 * NSP) + gs) * NTALL + kind * NT + tt) * 64 + ln) * 8);
        }
    };
    auto stage_commit = [&](unsigned char* dst) {
#pragma unroll
        for (int u = 0; u <= LPT; ++u) {
            if (u == LPT && (REM == 0 || (int)threadIdx.x >= REM)) break;
            *reinterpret_cast<f4*>(dst + (threadIdx.x + kDgThreads * u) * 16) = st[u];
        }
    };
#pragma unroll 1
    for (;;) {
        __syncthreads();                                                        // everyone has read the previous draw
        if (threadIdx.x == 0) next_item = atomicAdd(counters + xcd, 1u);
        __syncthreads();
        const long long it = t_begin + next_item;
        if (it >= t_end) break;
        const int kind = (int)(it % KINDS);
        const long long tl = it / KINDS;
        const int cg = (int)(tl % CG);
        const long long R0 = (tl / CG) * TJR;                                   // first padded row of the tile
        const int i0 = cg * CB;
        const bool active = R0 + RB * MB * mb < RT;                             // wave-uniform
        // ---- global offsets of this thread's window items (channel 0 of the chunk), -1 outside the map / in a padding row
        int goff[kIter];                                                        // element offsets (the host checks NB ho wo C_out < 2^31)
#pragma unroll
        for (int u = 0; u < kIter; ++u) {
            const int i = threadIdx.x + kDgThreads * u;
            const int pix = i >> 2, q = i & 3;
            const int wy = pix / WC, wc = pix - wy * WC;
            const long long Rp = R0 - 1 + wy;
            const int ox = i0 - 1 + wc;
            goff[u] = -1;
            if (i < kItems && Rp >= 0 && Rp < RT && ox >= 0 && ox < wo) {
                const long long nb = Rp / HP;
                const int j = (int)(Rp - nb * HP);
                if (j < ho) goff[u] = (int)(((nb * ho + j) * wo + ox) * CO + 8 * q);
            }
        }
        f32x16 acc[MB][4][NT];
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][k][t][r] = 0.f;
#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {
            __syncthreads();                                                    // the previous chunk's / tile's readers of the window are done
            // ---- window of g (32 channels of this chunk) -> three bf16 planes in LDS (g = gh + gm + gl exactly), zero outside
            if constexpr (DT != 0) {
                u16x8 vv[kIter];
#pragma unroll
                for (int u = 0; u < kIter; ++u) {
                    vv[u] = (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
                    if (goff[u] >= 0) vv[u] = *reinterpret_cast<const u16x8*>(G + goff[u] + 32 * c);
                }
#pragma unroll
                for (int u = 0; u < kIter; ++u) {
                    const int i = threadIdx.x + kDgThreads * u;
                    const int pix = i >> 2, q = i & 3;
                    const int wy = pix / WC, wc = pix - wy * WC;
                    if (i < kItems) *reinterpret_cast<u16x8*>(wnd + wy * ROWB + wc * 64 + ((q ^ (((wc >> 2) + 2 * wy) & 3)) << 4)) = vv[u];
                }
            } else if (!(SS_DG_ABL & 32)) {
                f4 va[kIter], vb[kIter];
#pragma unroll
                for (int u = 0; u < kIter; ++u) {
                    va[u] = (f4){0.f, 0.f, 0.f, 0.f}; vb[u] = va[u];
                    if (goff[u] >= 0) {
                        const float* p = G + goff[u] + 32 * c;
                        va[u] = *reinterpret_cast<const f4*>(p); vb[u] = *reinterpret_cast<const f4*>(p + 4);
                    }
                }
#pragma unroll
                for (int u = 0; u < kIter; ++u) {
                    const int i = threadIdx.x + kDgThreads * u;
                    const int pix = i >> 2, q = i & 3;
                    const int wy = pix / WC, wc = pix - wy * WC;
                    if (i < kItems) {
                        u16x8 o1, o2, o3;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float v = e < 4 ? va[u][e] : vb[u][e - 4];
                            const __bf16 h1 = (__bf16)v;
                            const float r1 = v - (float)h1;
                            const __bf16 h2 = (__bf16)r1;
                            const __bf16 h3 = (__bf16)(r1 - (float)h2);
                            o1[e] = __builtin_bit_cast(unsigned short, h1); o2[e] = __builtin_bit_cast(unsigned short, h2);
                            o3[e] = __builtin_bit_cast(unsigned short, h3);
                        }
                        unsigned char* const pp = wnd + wy * ROWB + wc * 64 + ((q ^ (((wc >> 2) + 2 * wy) & 3)) << 4);
                        *reinterpret_cast<u16x8*>(pp) = o1;
                        *reinterpret_cast<u16x8*>(pp + PLANE) = o2;
                        *reinterpret_cast<u16x8*>(pp + 2 * PLANE) = o3;
                    }
                }
            }
            // ---- weight stage 0 of this chunk
            stage_issue((long long)c * 25, kind);
            stage_commit(bst);
            __syncthreads();
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
                    const int tap = ky * 5 + kx;
                    const int py = ky & 1, px = kx & 1, cls = py * 2 + px;
                    const int dy = (py + 2 - ky) / 2, dx = (px + 2 - kx) / 2;   // (py + 2 - ky) is even: exact, also for -2
                    const int idx = (ky >> 1) * (px ? 2 : 3) + (kx >> 1);       // index of the tap within its class
                    const int sidx = tap / TPS, sub = tap - sidx * TPS;         // weight stage of this tap, its place in it
                    const bool more = (sidx + 1) * TPS < 25;
                    if (sub == 0 && more && !(SS_DG_ABL & 16)) stage_issue((long long)c * 25 + (sidx + 1) * TPS, kind);
                    if (active) {
                        // The tap's 12 NT MFMAs run on a scratch accumulator that starts at zero; the class's running sum takes ONE fp32 addition
                        // per tap (25 C_out / 32 roundings of the large sum per element instead of 12 x as many: 3.5x closer to float64 at
                        // K = 4608, tests).  The tap's products carry the sign (-1)^n of its weight fragments; fma(+-1, tmp, acc) undoes it.
                        const unsigned char* const bk = bst + (sidx & 1) * STG + sub * TAPB + lane * 16;
                        f32x16 tmp[MB][NT];
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            // the granule swizzle depends on the window row: one row up / down moves the slot by 2 (^ 32 bytes), like the second k-step
                            const unsigned char* const ap = wnd + dy * ROWB + (abase[dx + 1] ^ ((g ^ (dy & 1)) << 5));
                            s16x8 a[MB][NSP], b[NSP][NT];
#pragma unroll
                            for (int i = 0; i < MB; ++i)
#pragma unroll
                                for (int sp = 0; sp < NSP; ++sp)
                                    a[i][sp] = *reinterpret_cast<const s16x8*>((SS_DG_ABL & 8) ? wnd + lane * 16 + sp * 1024 : ap + i * RB * ROWB + sp * PLANE);
#pragma unroll
                            for (int sp = 0; sp < NSP; ++sp)
#pragma unroll
                                for (int t = 0; t < NT; ++t) b[sp][t] = *reinterpret_cast<const s16x8*>((SS_DG_ABL & 4) ? bst + lane * 16 + (sp * NT + t) * 1024 : bk + ((g * NSP + sp) * NT + t) * 1024);
                            // six cross terms, smallest first: al bh, am bm, ah bl, am bh, ah bm, ah bh — term-major over the wavefront's M blocks and N tiles
                            // (consecutive MFMAs run on different accumulators)
                            constexpr int kNQ = DT ? 1 : 6;
                            constexpr int kTa[6] = {DT ? 0 : 2, 1, 0, 1, 0, 0}, kTb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                            for (int q = 0; q < kNQ; ++q)
#pragma unroll
                                for (int i = 0; i < MB; ++i)
#pragma unroll
                                    for (int t = 0; t < NT; ++t) {
                                        if constexpr (DT != 0) {                // 16-bit modes: straight onto the class sum (the drift is far below the store's rounding)
                                            acc[i][cls][t] = mfma32<DT>(b[0][t], a[i][0], acc[i][cls][t]);
                                        } else if (g == 0 && q == 0) {
                                            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                                            tmp[i][t] = mfma32<DT>(b[kTb[q]][t], a[i][kTa[q]], zero);      // D^T: rows = input channels, columns = pixels
                                        } else {
                                            tmp[i][t] = mfma32<DT>(b[kTb[q]][t], a[i][kTa[q]], tmp[i][t]);
                                        }
                                    }
                        }
                        if constexpr (DT == 0) {
                            const float sgn = (((c * dg_cnt(cls) + idx) & 1) && !(SS_DG_ABL & 2)) ? -1.f : 1.f;
#pragma unroll
                            for (int i = 0; i < MB; ++i)
#pragma unroll
                                for (int t = 0; t < NT; ++t)
#pragma unroll
                                    for (int r = 0; r < 16; ++r) acc[i][cls][t][r] = __builtin_fmaf(sgn, tmp[i][t][r], acc[i][cls][t][r]);
                        }
                    }
                    if (sub == TPS - 1) {
                        if (more && !(SS_DG_ABL & 16)) stage_commit(bst + ((sidx + 1) & 1) * STG);
                        if (!(SS_DG_ABL & 1)) __syncthreads();
                    }
                }
            }
        }
        // ---- the product is taken TRANSPOSED (weights as the A operand): D[ci = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][pixel m = lane & 31], so a lane holds 4
        //      consecutive input channels of ONE pixel per register quad -> g_x[nb][2 j + py][2 (i0 + cc_m) + px][32 (kind NT + t) + ci] as 16-byte stores
        //      (4 per class and channel tile instead of 16 4-byte ones: what the 4-byte form cost, profiles/r04/sub_trace_v2.log)
        if (active) {
            const int qm = tx / CB, ccm = tx - qm * CB;                         // this lane's pixel: padded row qm, column ccm of the M block
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const long long Rp = R0 + RB * (MB * mb + i) + qm;
                const long long nb = Rp / HP;
                const int j = (int)(Rp - nb * HP);
                const bool row_ok = Rp < RT && j < ho;                          // (j == ho: the frame's padding row)
#pragma unroll
                for (int cls = 0; cls < 4; ++cls) {
                    const int py = cls >> 1, px = cls & 1;
                    const int iy = 2 * j + py, ix = 2 * (i0 + ccm) + px;
                    if (row_ok && iy < h && ix < w && (!(SS_DG_ABL & 64) || acc[i][cls][0][0] == 12345.678f)) {
                        typename ActT<DT>::type* const op = gx + ((nb * h + iy) * w + ix) * CI + 32 * (kind * NT) + 4 * half;
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int q4 = 0; q4 < 4; ++q4)
                                store_act4<DT>(op + 32 * t + 8 * q4, acc[i][cls][t][4 * q4], acc[i][cls][t][4 * q4 + 1], acc[i][cls][t][4 * q4 + 2], acc[i][cls][t][4 * q4 + 3]);
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" {

int ss_conv_s2_dgrad_supported(int Cin, int Cout, int k, int stride, int pad)
{
    return k == 5 && stride == 2 && pad == 2 && Cout == 2 * Cin && (Cin == 32 || Cin == 64 || Cin == 128 || Cin == 256);
}

long long ss_conv_s2_dgrad_ws_floats(int Cin, int Cout)
{
    if (Cin <= 0 || Cout <= 0 || Cin % 32 != 0 || Cout % 32 != 0) return 0;
    return (long long)25 * Cin * Cout * 3 / 2 + 8;                              // the weight as three bf16 terms in fragment order + 8 item counters
}

int ss_conv_s2_dgrad_f32(const float* g, const float* weight, float* g_x, float* ws, long long NB, int Cin, int Cout, int h, int w, void* stream)
{
    if (!g || !weight || !g_x || !ws || NB <= 0 || h <= 0 || w <= 0) return SS_EINVAL;
    if (!ss_conv_s2_dgrad_supported(Cin, Cout, 5, 2, 2) || !aligned16(g) || !aligned16(ws) || !aligned16(g_x)) return SS_EINVAL;
    const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
    if (NB * h * (long long)w * Cin > 0x7fffffffffLL || NB * ho * (long long)wo * Cout > 0x7fffffffLL) return SS_EINVAL;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return SS_ELAUNCH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned short* Bf = reinterpret_cast<unsigned short*>(ws);
    unsigned* counters = reinterpret_cast<unsigned*>(ws + (long long)25 * Cin * Cout * 3 / 2);
    hipLaunchKernelGGL(conv_s2_dgrad_prep_kernel<0>, dim3(grid_for((long long)25 * Cin * Cout * 3 / 8, 4096)), dim3(kBlock), 0, s, weight, Bf, counters, Cin, Cout);
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    // column blocking of the (j, i) grid: 8 columns x 4 rows per wavefront wastes <= 1 / 8 of a block on any width; 32 x 1 only when it fits as well
    bool wide = ((wo + 31) / 32) * 32 <= ((wo + 7) / 8) * 8;
    static const char* const cb_env = getenv("SS_DGRAD_CB");                    // tuning knob (tools/bench_conv_dgrad.py): 32 | 8 — read once, not per call
    if (cb_env) wide = cb_env[0] == '3';
    const int MB = (Cin == 32 && !wide) ? 2 : 1;                                 // C_in 32 (one N tile): two M blocks per wavefront share the weight fragments

    // two workgroups per CU, persistent: each draws items from the range of `blockIdx.x & 7` (its XCD as dispatched — for speed only).  The 8 ranges
    // are drawn by 8 distinct residues, so the grid is never smaller than 8: on a device (or partition) with fewer than 4 CUs every range still
    // has a workgroup and no part of g_x stays unwritten
    const unsigned grid = (unsigned)(2 * cus < 8 ? 8 : 2 * cus);
#define SS_DG(CO_, NTALL_, NT_) do { \
        if (wide) hipLaunchKernelGGL((conv_s2_dgrad_kernel<CO_, NTALL_, NT_, 32>), dim3(grid), dim3(kDgThreads), 0, s, g, Bf, g_x, counters, (int)NB, h, w, ho, wo); \
        else hipLaunchKernelGGL((conv_s2_dgrad_kernel<CO_, NTALL_, NT_, 8>), dim3(grid), dim3(kDgThreads), 0, s, g, Bf, g_x, counters, (int)NB, h, w, ho, wo); } while (0)
    if (Cin == 32 && MB == 2) hipLaunchKernelGGL((conv_s2_dgrad_kernel<64, 1, 1, 8, 2>), dim3(grid), dim3(kDgThreads), 0, s, g, Bf, g_x, counters, (int)NB, h, w, ho, wo);
    else if (Cin == 32) SS_DG(64, 1, 1);
    else if (Cin == 64) SS_DG(128, 2, 2);
    else if (Cin == 128) SS_DG(256, 4, 2);
    else SS_DG(512, 8, 2);
#undef SS_DG
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

/* ss_conv_s2_dgrad_f32 on 16-bit activation gradients (ABI 9): g and g_x in `dtype`, weight fp32 (rounded once to `dtype`), fp32 accumulation; ws as for
   the fp32 form. */
int ss_conv_s2_dgrad_x16(const void* g, const float* weight, void* g_x, float* ws, long long NB, int Cin, int Cout, int h, int w, int dtype, void* stream)
{
    if (!g || !weight || !g_x || !ws || NB <= 0 || h <= 0 || w <= 0 || (dtype != SS_DT_F16 && dtype != SS_DT_BF16)) return SS_EINVAL;
    if (!ss_conv_s2_dgrad_supported(Cin, Cout, 5, 2, 2) || !aligned16(g) || !aligned16(ws) || !aligned16(g_x)) return SS_EINVAL;
    const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
    if (NB * h * (long long)w * Cin > 0x7fffffffffLL || NB * ho * (long long)wo * Cout > 0x7fffffffLL) return SS_EINVAL;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return SS_ELAUNCH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned short* Bf = reinterpret_cast<unsigned short*>(ws);
    unsigned* counters = reinterpret_cast<unsigned*>(ws + (long long)25 * Cin * Cout * 3 / 2);
    const int pg = grid_for((long long)25 * Cin * Cout / 8, 4096);
    if (dtype == SS_DT_F16) hipLaunchKernelGGL(conv_s2_dgrad_prep_kernel<SS_DT_F16>, dim3(pg), dim3(kBlock), 0, s, weight, Bf, counters, Cin, Cout);
    else hipLaunchKernelGGL(conv_s2_dgrad_prep_kernel<SS_DT_BF16>, dim3(pg), dim3(kBlock), 0, s, weight, Bf, counters, Cin, Cout);
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    const bool wide = ((wo + 31) / 32) * 32 <= ((wo + 7) / 8) * 8;
    const unsigned grid = (unsigned)(2 * cus < 8 ? 8 : 2 * cus);
    const unsigned short* g16 = static_cast<const unsigned short*>(g);
    unsigned short* gx16 = static_cast<unsigned short*>(g_x);
#define SS_DG16D(CO_, NTALL_, NT_, DTT) do { \
        if (wide) hipLaunchKernelGGL((conv_s2_dgrad_kernel<CO_, NTALL_, NT_, 32, 1, DTT>), dim3(grid), dim3(kDgThreads), 0, s, g16, Bf, gx16, counters, (int)NB, h, w, ho, wo); \
        else hipLaunchKernelGGL((conv_s2_dgrad_kernel<CO_, NTALL_, NT_, 8, 1, DTT>), dim3(grid), dim3(kDgThreads), 0, s, g16, Bf, gx16, counters, (int)NB, h, w, ho, wo); } while (0)
#define SS_DG16(DTT) do { \
        if (Cin == 32 && !wide) hipLaunchKernelGGL((conv_s2_dgrad_kernel<64, 1, 1, 8, 2, DTT>), dim3(grid), dim3(kDgThreads), 0, s, g16, Bf, gx16, counters, (int)NB, h, w, ho, wo); \
        else if (Cin == 32) SS_DG16D(64, 1, 1, DTT); else if (Cin == 64) SS_DG16D(128, 2, 2, DTT); else if (Cin == 128) SS_DG16D(256, 4, 2, DTT); else SS_DG16D(512, 8, 2, DTT); } while (0)
    if (dtype == SS_DT_F16) SS_DG16(SS_DT_F16); else SS_DG16(SS_DT_BF16);
#undef SS_DG16
#undef SS_DG16D
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

}  // extern "C"


// ---------------------------------------------------------------------------------------------------
// WEIGHT gradient of the first encoder layer, nn.Conv2d(C_in = 4 | 2, 32, kernel_size=5, stride=1, padding=2, bias=False) on the event-voxel
// input (/root/reference/network/SNN_models.py:75-79, 263-267, 450-454; autograd in the reference; include/ss_neuron.h: ss_dense_conv_s1_wgrad_f32):
//     g_W[co][ci][ky][kx] = sum_{nb, y, x} g[nb][y][x][co] * x[nb][y + ky - 2][x + kx - 2][ci]
// — the last MIOpen call of the default training step (igemm_wrw, 0.68 ms at config 3 for 23 GMAC: the contraction runs over 7.2 M pixels into a
// 32 x 100 result).  Here: v_mfma_f32_32x32x16_bf16 with M = co, N = (tap, ci) in tiles of 32, K = 16 pixels along a row; BOTH operands split
// into three bf16 terms, six cross terms kept (any input values; exact for the voxeliser's integer counts), fp32 accumulation.
//   * A workgroup owns tiles of 4 rows x 32 columns of pixels, loaded one tile ahead into registers.  g is staged TRANSPOSED in LDS (gT[co][pixel], fp32), so that a lane's 8 consecutive
//     pixels of one output channel are two 16-B reads, split in registers once per k-step and used for all N tiles; the input window
//     (8 x 36 pixels) is split once while it is staged: three bf16 planes xT[plane][ci][row][col].  A B fragment — 8 consecutive columns
//     starting at an arbitrary column (the tap's kx) — is five aligned 4-B reads funnel-shifted by the column parity (v_alignbyte).
//   * A wavefront handles every fourth k-step of the tile for ALL N tiles (the A fragment is built once); a tile's MFMAs accumulate on a scratch
//     accumulator that starts at zero, the running sum takes one fp32 addition per tile (sign alternating per tile: the MFMA drift of DESIGN.md
//     3.8); the wavefronts' sums meet in LDS, the workgroup partials in a fixed-order fp64 second pass: deterministic.
// ---------------------------------------------------------------------------------------------------
namespace {

constexpr int kW1Threads = 256;
constexpr int kW1TR = 4, kW1TC = 32;                   // pixel rows x columns of a tile: 8 k-steps of 16 pixels (two per wavefront)
constexpr int kW1WR = kW1TR + 4, kW1WC = 40;           // window rows; window columns padded from 36 to 40 (80-B rows: 4-B aligned reads)
constexpr int kW1GS = kW1TR * kW1TC + 4;               // pixel stride (floats) of a channel's row in gT: 16-B aligned, banks spread
constexpr int kW1Groups = 512;                         // workgroups (two per CU) = partial blocks of the second pass

// DT != 0 (16-bit activation modes): g arrives in the 16-bit format (widened exactly while it is staged — it IS the operand), the fp32 input is rounded once
// to the format as in the forward; one MFMA per (k-step, N tile); the weight gradient stays fp32
template <int CI, int DT = 0>
__global__ __launch_bounds__(kW1Threads, 2) void dense_conv_s1_wgrad_kernel(const typename ActT<DT>::type* __restrict__ G, const float* __restrict__ X, float* __restrict__ part,
                                                                            int NB, int h, int w)
{
    constexpr int NSP = DT ? 1 : 3;
    constexpr int NV = 25 * CI, NT = (NV + 31) / 32;                            // valid (tap, ci) columns; N tiles
    constexpr int CS = kW1WR * kW1WC + 2;                                       // elements per input channel of a plane: + one dword, so that the CI lanes of a tap
                                                                                // (same window position, consecutive channels) read CI different banks
    constexpr int PL = CI * CS;                                                 // elements of one bf16 plane of the window
    __shared__ __attribute__((aligned(16))) float gT[32 * kW1GS];
    __shared__ __attribute__((aligned(16))) unsigned short xT[NSP * PL + 8];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mn = lane & 31, kq = lane >> 5;                                   // A: m = co; B: n = column of the N tile; k group of 8 pixels
    const int RG = (h + kW1TR - 1) / kW1TR, CG = (w + kW1TC - 1) / kW1TC;
    const long long n_tiles = (long long)NB * RG * CG;
    const long long t_begin = n_tiles * blockIdx.x / gridDim.x, t_end = n_tiles * (blockIdx.x + 1) / gridDim.x;
    // this lane's B columns: (tap, ci) of N tile t -> element offset of (ci, ky, kx) inside a plane, for the pixel at tile row 0, column 0
    int boff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int nv = min(32 * t + mn, NV - 1), tap = nv / CI, ci = nv - tap * CI, ky = tap / 5, kx = tap - 5 * ky;
        boff[t] = ci * CS + ky * kW1WC + kx;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bool neg = false;
    // a tile's g values and window pixels are loaded into registers one tile AHEAD (the loads fly during the previous tile's MFMAs) and written
    // to LDS — g transposed, the window split into its three bf16 planes — once that tile's readers are done
    constexpr int GU = (kW1TR * kW1TC * 8) / kW1Threads, XU = (kW1WR * kW1WC + kW1Threads - 1) / kW1Threads;
    f4 gv[GU];
    float xv[XU][CI];
    auto load_tile = [&](long long tl) {
        const int cg = (int)(tl % CG);
        const long long rr = tl / CG;
        const int rg = (int)(rr % RG), nb = (int)(rr / RG);
        const int y0 = kW1TR * rg, x0 = kW1TC * cg;
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = threadIdx.x + kW1Threads * u;
            const int pix = i >> 3, q = i & 7, py = pix / kW1TC, px = pix - py * kW1TC;
            gv[u] = (f4){0.f, 0.f, 0.f, 0.f};
            if (y0 + py < h && x0 + px < w) {
                if constexpr (DT == 0) gv[u] = load_stream(reinterpret_cast<const f4*>(G + (((long long)nb * h + y0 + py) * w + x0 + px) * 32) + q);
                else {
                    const u16x4 g4 = load_stream(reinterpret_cast<const u16x4*>(G + (((long long)nb * h + y0 + py) * w + x0 + px) * 32) + q);
                    gv[u] = (f4){widen<DT>(g4[0]), widen<DT>(g4[1]), widen<DT>(g4[2]), widen<DT>(g4[3])};
                }
            }
        }
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = threadIdx.x + kW1Threads * u;
            const int wy = i / kW1WC, wx = i - wy * kW1WC;
            const int iy = y0 - 2 + wy, ix = x0 - 2 + wx;
#pragma unroll
            for (int c = 0; c < CI; ++c) xv[u][c] = 0.f;
            if (i < kW1WR * kW1WC && wx < kW1TC + 4 && iy >= 0 && iy < h && ix >= 0 && ix < w) {
                const float* p = X + (((long long)nb * h + iy) * w + ix) * CI;
#pragma unroll
                for (int c = 0; c < CI; ++c) xv[u][c] = p[c];
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = threadIdx.x + kW1Threads * u;
            const int pix = i >> 3, q = i & 7;
#pragma unroll
            for (int e = 0; e < 4; ++e) gT[(4 * q + e) * kW1GS + pix] = gv[u][e];
        }
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = threadIdx.x + kW1Threads * u;
            const int wy = i / kW1WC, wx = i - wy * kW1WC;
            if (i < kW1WR * kW1WC) {
#pragma unroll
                for (int c = 0; c < CI; ++c) {
                    const float v = xv[u][c];
                    unsigned short* const q = xT + c * CS + wy * kW1WC + wx;
                    if constexpr (DT != 0) { q[0] = round_op<DT>(v); continue; }
                    const __bf16 h1 = (__bf16)v;
                    const float r1 = v - (float)h1;
                    const __bf16 h2 = (__bf16)r1;
                    const __bf16 h3 = (__bf16)(r1 - (float)h2);
                    if constexpr (DT == 0) {
                        q[0] = __builtin_bit_cast(unsigned short, h1);
                        q[PL] = __builtin_bit_cast(unsigned short, h2);
                        q[2 * PL] = __builtin_bit_cast(unsigned short, h3);
                    }
                }
            }
        }
    };
    if (t_begin < t_end) load_tile(t_begin);
#pragma unroll 1
    for (long long tl = t_begin; tl < t_end; ++tl) {
        __syncthreads();                                                        // the previous tile's readers are done
        store_tile();
        __syncthreads();
        if (tl + 1 < t_end) load_tile(tl + 1);
        const float sgn = neg ? -1.f : 1.f;
        f32x16 tmp[NT];
#pragma unroll
        for (int j = 0; j < kW1TR / 2; ++j) {                                   // this wavefront's k-steps: ks = wv + 4 j -> tile row, column half
            const int ks = wv + 4 * j, py = ks >> 1, pxb = 16 * (ks & 1) + 8 * kq;
            const float* const gp = gT + mn * kW1GS + py * kW1TC + pxb;
            const f4 ga = *reinterpret_cast<const f4*>(gp), gb = *reinterpret_cast<const f4*>(gp + 4);
            s16x8 ah, am, al;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = (e < 4 ? ga[e] : gb[e - 4]) * sgn;
                if constexpr (DT != 0) { ah[e] = (short)round_op<DT>(v); continue; }     // (exact: v is a value of the format)
                const __bf16 h1 = (__bf16)v;
                const float r1 = v - (float)h1;
                const __bf16 h2 = (__bf16)r1;
                const __bf16 h3 = (__bf16)(r1 - (float)h2);
                ah[e] = __builtin_bit_cast(short, h1); am[e] = __builtin_bit_cast(short, h2); al[e] = __builtin_bit_cast(short, h3);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                // 8 consecutive window columns starting at element `el` of a plane: five aligned dwords, shifted by the start's parity
                const int el = boff[t] + py * kW1WC + pxb;
                const unsigned sh = (el & 1) * 2;
                s16x8 b[NSP];
#pragma unroll
                for (int sp = 0; sp < NSP; ++sp) {
                    const unsigned* const d = reinterpret_cast<const unsigned*>(xT + sp * PL + (el & ~1));
                    unsigned dw[5];
#pragma unroll
                    for (int q = 0; q < 5; ++q) dw[q] = d[q];
                    typedef unsigned u4 __attribute__((ext_vector_type(4)));
                    u4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = __builtin_amdgcn_alignbyte(dw[q + 1], dw[q], sh);
                    b[sp] = __builtin_bit_cast(s16x8, o);
                }
                if constexpr (DT != 0) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    tmp[t] = mfma32<DT>(ah, b[0], j == 0 ? zero : tmp[t]);
                } else {
                // six cross terms, smallest first: al bh, am bm, ah bl, am bh, ah bm, ah bh
                if (j == 0) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    tmp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b[0], zero, 0, 0, 0);
                } else {
                    tmp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b[0], tmp[t], 0, 0, 0);
                }
                tmp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b[NSP > 1 ? 1 : 0], tmp[t], 0, 0, 0);
                tmp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[NSP - 1], tmp[t], 0, 0, 0);
                tmp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b[0], tmp[t], 0, 0, 0);
                tmp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[NSP > 1 ? 1 : 0], tmp[t], 0, 0, 0);
                tmp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[0], tmp[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = __builtin_fmaf(sgn, tmp[t][r], acc[t][r]);
        neg = !neg;
    }
    // ---- the four wavefronts' sums meet in LDS, wavefront 0 first (fixed order): D[row = co = (r & 3) + 8 (r >> 2) + 4 kq][col = n] -> red[co][32 t + n],
    //      then one coalesced block part[workgroup][co][32 t + n]
    float* const red = gT;                                                      // 32 x NT x 32 floats <= the gT slice
    static_assert(32 * NT * 32 <= 32 * kW1GS, "the reduction block fits the g slice");
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        __syncthreads();
        if (wv == q) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* const d = red + ((r & 3) + 8 * (r >> 2) + 4 * kq) * (NT * 32) + 32 * t + mn;
                    *d = q == 0 ? acc[t][r] : *d + acc[t][r];
                }
        }
    }
    __syncthreads();
    float* const pp = part + (long long)blockIdx.x * 32 * NT * 32;
    for (int i = threadIdx.x; i < 32 * NT * 32; i += kW1Threads) pp[i] = red[i];
}

// second pass, stage 1: out[range][i] = sum over the partial blocks of the range (in order, fp64) — coalesced over i
__global__ __launch_bounds__(kBlock) void dense_conv_s1_wgrad_reduce_kernel(const float* __restrict__ part, double* __restrict__ out, int n_part, int per_range, int len)
{
    const int i = blockIdx.x * kBlock + threadIdx.x, rg = blockIdx.y;
    if (i >= len) return;
    const int p0 = rg * per_range, p1 = min(n_part, p0 + per_range);
    double s = 0.0;
    for (int p = p0; p < p1; ++p) s += (double)part[(long long)p * len + i];
    out[(long long)rg * len + i] = s;
}

// second pass, stage 2: g_W[co][ci][ky][kx] (+)= sum over the ranges (in order) of D[co][tap * CI + ci]
__global__ __launch_bounds__(kBlock) void dense_conv_s1_wgrad_finish_kernel(const double* __restrict__ mid, float* __restrict__ gW, int n_ranges, int CI, int NTW,
                                                                            int accumulate)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;                            // flat index of g_W [32][CI][25]
    if (i >= 32 * CI * 25) return;
    const int co = i / (25 * CI), rem = i - co * 25 * CI, ci = rem / 25, tap = rem - ci * 25;
    const long long src = (long long)co * NTW + tap * CI + ci;
    double s = 0.0;
    for (int r = 0; r < n_ranges; ++r) s += mid[(long long)r * 32 * NTW + src];
    gW[i] = accumulate ? (float)((double)gW[i] + s) : (float)s;
}

// ---------------------------------------------------------------------------------------------------
// The same weight gradient for C_in = 4, second form (round 5): both tiles pixel-major in LDS, every fragment two transposed LDS reads
// ---------------------------------------------------------------------------------------------------
// dense_conv_s1_wgrad_kernel takes the same time in the fp32 mode (six MFMAs per product) and the 16-bit modes (one): it is bound by building its fragments — g
// transposed into LDS with 4-byte stores, a B fragment = five dword reads + four funnel shifts per split plane.  Here (the scheme of spike_conv_wgrad_tr_kernel,
// ss_wgrad.hip) the g tile (8 rows x 32 pixels x 32 channels) and the input window (12 x 36 pixels x 4 channels = 8 bytes per pixel) are staged as they lie in
// HBM — split into three bf16 planes in the fp32 mode, the operand format as stored / rounded once in the 16-bit modes — and ds_read_b64_tr_b16 does both
// transpositions: a source lane points at one pixel of ITS column's tap (an 8-byte window pixel is exactly the four channels of a tap), so the 25 taps x 4
// channels of an N tile cost two reads per plane; taps beyond 24 read a zero pad.  Same tile order, sign alternation, partial-block layout and second pass as the
// first form (which stays for C_in = 2).
constexpr int kW2TR = 8, kW2TC = 32;                                             // pixel rows x columns of a tile: 16 k-steps, four per wavefront
constexpr int kW2WR = kW2TR + 4, kW2WC = kW2TC + 4;                              // 12 x 36 window pixels
constexpr int kW2XWin = kW2WR * kW2WC * 8;                                      // 3 456 bytes of window per plane
constexpr int kW2XPlane = kW2XWin + 2304;                                       // + the zero pad a phantom tap's lanes read (the largest k-step offset is 2 176)
constexpr int kW2GPlane = kW2TR * kW2TC * 64;                                   // 16 384: 256 pixels x 32 channels x 2 B

template <int DT>
__global__ __launch_bounds__(kW1Threads, 2) void dense_conv_s1_wgrad_tr_kernel(const typename ActT<DT>::type* __restrict__ G, const float* __restrict__ X,
                                                                               float* __restrict__ part, int NB, int h, int w)
{
    constexpr int NSP = DT ? 1 : 3, NT = 4, CI = 4;
    __shared__ __attribute__((aligned(16))) unsigned char gl[NSP * kW2GPlane];
    __shared__ __attribute__((aligned(16))) unsigned char xl[NSP * kW2XPlane];
    static_assert(32 * NT * 32 * 4 <= kW2GPlane, "the reduction block fits one g plane");
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int RG = (h + kW2TR - 1) / kW2TR, CG = (w + kW2TC - 1) / kW2TC;
    const long long n_tiles = (long long)NB * RG * CG;
    const long long t_begin = n_tiles * blockIdx.x / gridDim.x, t_end = n_tiles * (blockIdx.x + 1) / gridDim.x;
    // this lane as a SOURCE lane of the transposed reads: pixel r_s of a 4-pixel block; channels 4 j_s .. + 3 of its 16-lane group's half of g / the tap
    // 8 t + 4 hsel + j_s of N tile t
    const int r_s = (lane & 15) >> 2, j_s = lane & 3, hsel = (lane >> 4) & 1, khalf = lane >> 5;
    const int gbase = (8 * khalf + r_s) * 64 + (16 * hsel + 4 * j_s) * 2;
    int boff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tap = 8 * t + 4 * hsel + j_s, ky = tap / 5, kx = tap - 5 * ky;
        boff[t] = tap < 25 ? (ky * kW2WC + kx + 8 * khalf + r_s) * 8 : kW2XWin;
    }
    for (int i = threadIdx.x; i < NSP * 2304 / 16; i += kW1Threads)               // the zero pads (never written again)
        *reinterpret_cast<f4*>(xl + (i / 144) * kW2XPlane + kW2XWin + (i % 144) * 16) = (f4){0.f, 0.f, 0.f, 0.f};
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bool neg = false;
    constexpr int GU = kW2TR * kW2TC * 4 / kW1Threads;                            // (pixel, 8-channel granule) items of g per thread: 4
    constexpr int XU = (kW2WR * kW2WC + kW1Threads - 1) / kW1Threads;              // window pixels per thread: 2
    f4 ga[DT ? 1 : GU], gb[DT ? 1 : GU], xv[XU];
    u16x8 g16[DT ? GU : 1];
    auto load_tile = [&](long long tl) {
        const int cg = (int)(tl % CG);
        const long long rr = tl / CG;
        const int rg = (int)(rr % RG), nb = (int)(rr / RG);
        const int y0 = kW2TR * rg, x0 = kW2TC * cg;
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = threadIdx.x + kW1Threads * u;
            const int pix = i >> 2, q = i & 3, py = pix >> 5, px = pix & 31;
            const bool ok = y0 + py < h && x0 + px < w;
            const long long el = (((long long)nb * h + y0 + py) * w + x0 + px) * 32 + 8 * q;
            if constexpr (DT != 0) {
                g16[u] = (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) g16[u] = load_stream(reinterpret_cast<const u16x8*>(G + el));
            } else {
                ga[u] = (f4){0.f, 0.f, 0.f, 0.f}; gb[u] = ga[u];
                if (ok) { ga[u] = load_stream(reinterpret_cast<const f4*>(G + el)); gb[u] = load_stream(reinterpret_cast<const f4*>(G + el + 4)); }
            }
        }
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = threadIdx.x + kW1Threads * u;
            const int wy = i / kW2WC, wx = i - wy * kW2WC;
            const int iy = y0 - 2 + wy, ix = x0 - 2 + wx;
            xv[u] = (f4){0.f, 0.f, 0.f, 0.f};
            if (i < kW2WR * kW2WC && iy >= 0 && iy < h && ix >= 0 && ix < w) xv[u] = *reinterpret_cast<const f4*>(X + (((long long)nb * h + iy) * w + ix) * CI);
        }
    };
    auto store_tile = [&](bool negate) {
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = threadIdx.x + kW1Threads * u;
            unsigned char* const d = gl + (i >> 2) * 64 + (i & 3) * 16;
            if constexpr (DT != 0) {
                u16x8 o = g16[u];
                if (negate) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (unsigned short)(o[e] ^ 0x8000u);      // (-0 for a zero: harmless)
                }
                *reinterpret_cast<u16x8*>(d) = o;
            } else {
                u16x8 o[3];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v0 = e < 4 ? ga[u][e] : gb[u][e - 4];
                    const float v = negate ? -v0 : v0;
                    const __bf16 h1 = (__bf16)v;
                    const float r1 = v - (float)h1;
                    const __bf16 h2 = (__bf16)r1;
                    const __bf16 h3 = (__bf16)(r1 - (float)h2);
                    o[0][e] = __builtin_bit_cast(unsigned short, h1); o[1][e] = __builtin_bit_cast(unsigned short, h2); o[2][e] = __builtin_bit_cast(unsigned short, h3);
                }
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) *reinterpret_cast<u16x8*>(d + sp * kW2GPlane) = o[sp];
            }
        }
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = threadIdx.x + kW1Threads * u;
            if (i < kW2WR * kW2WC) {
                u16x4 o[NSP];
#pragma unroll
                for (int c = 0; c < CI; ++c) {
                    const float v = xv[u][c];
                    if constexpr (DT != 0) { o[0][c] = round_op<DT>(v); continue; }
                    const __bf16 h1 = (__bf16)v;
                    const float r1 = v - (float)h1;
                    const __bf16 h2 = (__bf16)r1;
                    const __bf16 h3 = (__bf16)(r1 - (float)h2);
                    o[0][c] = __builtin_bit_cast(unsigned short, h1);
                    if constexpr (NSP == 3) { o[1][c] = __builtin_bit_cast(unsigned short, h2); o[2][c] = __builtin_bit_cast(unsigned short, h3); }
                }
#pragma unroll
                for (int sp = 0; sp < NSP; ++sp) *reinterpret_cast<u16x4*>(xl + sp * kW2XPlane + i * 8) = o[sp];
            }
        }
    };
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef s16x4 __attribute__((address_space(3))) * lds4_t;
    auto frag = [&](const unsigned char* p0, int step) {                            // 8 consecutive pixels of one channel: two transposed reads, `step` bytes = 4 pixels
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0 + step));
        return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    if (t_begin < t_end) load_tile(t_begin);
#pragma unroll 1
    for (long long tl = t_begin; tl < t_end; ++tl) {
        __syncthreads();                                                        // the previous tile's readers are done
        store_tile(neg);
        __syncthreads();
        if (tl + 1 < t_end) load_tile(tl + 1);
        const float sgn = neg ? -1.f : 1.f;
        f32x16 tmp[NT];
#pragma unroll
        for (int j = 0; j < kW2TR / 2; ++j) {                                   // this wavefront's k-steps: ks = wv + 4 j -> tile row ks >> 1, column half ks & 1
            const int ks = wv + 4 * j, py = ks >> 1, hh = ks & 1;
            const int goff = (py * 32 + 16 * hh) * 64, xoff = (py * kW2WC + 16 * hh) * 8;
            s16x8 a[NSP];
#pragma unroll
            for (int sp = 0; sp < NSP; ++sp) a[sp] = frag(gl + sp * kW2GPlane + gbase + goff, 4 * 64);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                s16x8 b[NSP];
#pragma unroll
                for (int sp = 0; sp < NSP; ++sp) b[sp] = frag(xl + sp * kW2XPlane + boff[t] + xoff, 4 * 8);
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if constexpr (DT != 0) {
                    tmp[t] = mfma32<DT>(a[0], b[0], j == 0 ? zero : tmp[t]);
                } else {                                                        // six cross terms, smallest first: al bh, am bm, ah bl, am bh, ah bm, ah bh
                    tmp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], j == 0 ? zero : tmp[t], 0, 0, 0);
                    tmp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], tmp[t], 0, 0, 0);
                    tmp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], tmp[t], 0, 0, 0);
                    tmp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], tmp[t], 0, 0, 0);
                    tmp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], tmp[t], 0, 0, 0);
                    tmp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], tmp[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = __builtin_fmaf(sgn, tmp[t][r], acc[t][r]);
        neg = !neg;
    }
    // ---- the four wavefronts' sums meet in LDS, wavefront 0 first (fixed order): D[row = co = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][col = n] -> red[co][32 t + n]
    float* const red = reinterpret_cast<float*>(gl);
    const int mn = lane & 31, kq = lane >> 5;
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        __syncthreads();
        if (wv == q) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* const d = red + ((r & 3) + 8 * (r >> 2) + 4 * kq) * (NT * 32) + 32 * t + mn;
                    *d = q == 0 ? acc[t][r] : *d + acc[t][r];
                }
        }
    }
    __syncthreads();
    float* const pp = part + (long long)blockIdx.x * 32 * NT * 32;
    for (int i = threadIdx.x; i < 32 * NT * 32; i += kW1Threads) pp[i] = red[i];
}

constexpr int kW1Ranges = 16;

}  // namespace

// A/B switch (tools/): SS_S1_WGRAD_TR=0 keeps the first form for C_in = 4 too — read once
static bool s1_wgrad_tr_on()
{
    static const char* const e = getenv("SS_S1_WGRAD_TR");
    return !(e && e[0] == '0');
}

extern "C" {

int ss_dense_conv_s1_wgrad_supported(int Cin, int Cout, int k, int stride, int pad)
{
    return k == 5 && stride == 1 && pad == 2 && Cout == 32 && (Cin == 4 || Cin == 2);
}

long long ss_dense_conv_s1_wgrad_ws_floats(int Cin)
{
    if (Cin != 4 && Cin != 2) return 0;
    return (long long)(kW1Groups + 2 * kW1Ranges) * 32 * ((25 * Cin + 31) / 32) * 32;      // workgroup partials + the fp64 range sums
}

int ss_dense_conv_s1_wgrad_f32(const float* g, const float* x, float* g_w, float* ws, long long NB, int Cin, int Cout, int h, int w, int accumulate, void* stream)
{
    if (!g || !x || !g_w || !ws || NB <= 0 || NB > 0x7fffffff || h <= 0 || w <= 0 || !ss_dense_conv_s1_wgrad_supported(Cin, Cout, 5, 1, 2)) return SS_EINVAL;
    if (!aligned16(g) || !aligned16(ws)) return SS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long n_tiles = NB * ((h + kW1TR - 1) / kW1TR) * ((w + kW1TC - 1) / kW1TC);
    const unsigned grid = (unsigned)(n_tiles < kW1Groups ? n_tiles : kW1Groups);
    const int NTW = ((25 * Cin + 31) / 32) * 32;
    if (Cin == 4 && s1_wgrad_tr_on()) hipLaunchKernelGGL((dense_conv_s1_wgrad_tr_kernel<0>), dim3(grid), dim3(kW1Threads), 0, s, g, x, ws, (int)NB, h, w);
    else if (Cin == 4) hipLaunchKernelGGL((dense_conv_s1_wgrad_kernel<4>), dim3(grid), dim3(kW1Threads), 0, s, g, x, ws, (int)NB, h, w);
    else hipLaunchKernelGGL((dense_conv_s1_wgrad_kernel<2>), dim3(grid), dim3(kW1Threads), 0, s, g, x, ws, (int)NB, h, w);
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    const int len = 32 * NTW, per_range = ((int)grid + kW1Ranges - 1) / kW1Ranges;
    double* mid = reinterpret_cast<double*>(ws + (long long)kW1Groups * len);
    hipLaunchKernelGGL(dense_conv_s1_wgrad_reduce_kernel, dim3((len + kBlock - 1) / kBlock, kW1Ranges), dim3(kBlock), 0, s, ws, mid, (int)grid, per_range, len);
    hipLaunchKernelGGL(dense_conv_s1_wgrad_finish_kernel, dim3((32 * Cin * 25 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, mid, g_w, kW1Ranges, Cin, NTW, accumulate);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

/* ss_dense_conv_s1_wgrad_f32 on a 16-bit output gradient (ABI 9): g in `dtype`, x fp32 (rounded once to `dtype`: the forward's operand), g_w fp32 */
int ss_dense_conv_s1_wgrad_x16(const void* g, const float* x, float* g_w, float* ws, long long NB, int Cin, int Cout, int h, int w, int accumulate, int dtype, void* stream)
{
    if (!g || !x || !g_w || !ws || NB <= 0 || NB > 0x7fffffff || h <= 0 || w <= 0 || !ss_dense_conv_s1_wgrad_supported(Cin, Cout, 5, 1, 2)) return SS_EINVAL;
    if (!aligned16(g) || !aligned16(ws) || (dtype != SS_DT_F16 && dtype != SS_DT_BF16)) return SS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long n_tiles = NB * ((h + kW1TR - 1) / kW1TR) * ((w + kW1TC - 1) / kW1TC);
    const unsigned grid = (unsigned)(n_tiles < kW1Groups ? n_tiles : kW1Groups);
    const int NTW = ((25 * Cin + 31) / 32) * 32;
    const unsigned short* g16 = static_cast<const unsigned short*>(g);
#define SS_W116(CI_) do { if (dtype == SS_DT_F16) hipLaunchKernelGGL((dense_conv_s1_wgrad_kernel<CI_, SS_DT_F16>), dim3(grid), dim3(kW1Threads), 0, s, g16, x, ws, (int)NB, h, w); \
                          else hipLaunchKernelGGL((dense_conv_s1_wgrad_kernel<CI_, SS_DT_BF16>), dim3(grid), dim3(kW1Threads), 0, s, g16, x, ws, (int)NB, h, w); } while (0)
    if (Cin == 4 && s1_wgrad_tr_on()) {
        if (dtype == SS_DT_F16) hipLaunchKernelGGL((dense_conv_s1_wgrad_tr_kernel<SS_DT_F16>), dim3(grid), dim3(kW1Threads), 0, s, g16, x, ws, (int)NB, h, w);
        else hipLaunchKernelGGL((dense_conv_s1_wgrad_tr_kernel<SS_DT_BF16>), dim3(grid), dim3(kW1Threads), 0, s, g16, x, ws, (int)NB, h, w);
    } else if (Cin == 4) SS_W116(4); else SS_W116(2);
#undef SS_W116
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    const int len = 32 * NTW, per_range = ((int)grid + kW1Ranges - 1) / kW1Ranges;
    double* mid = reinterpret_cast<double*>(ws + (long long)kW1Groups * len);
    hipLaunchKernelGGL(dense_conv_s1_wgrad_reduce_kernel, dim3((len + kBlock - 1) / kBlock, kW1Ranges), dim3(kBlock), 0, s, ws, mid, (int)grid, per_range, len);
    hipLaunchKernelGGL(dense_conv_s1_wgrad_finish_kernel, dim3((32 * Cin * 25 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, mid, g_w, kW1Ranges, Cin, NTW, accumulate);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

}  // extern "C"

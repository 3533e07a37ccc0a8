// ss_upconv_bwd.hip — decoder backward with the per-tap gradient tensor g_P held ON CHIP only: adjoint gather + data-gradient contraction in one
// kernel (include/ss_neuron.h: ss_upconv_bwd_dgrad_f32).
//
// Reference: autograd through NNConvUpsampling (/root/reference/network/blocks.py:110-132: UpsamplingNearest2d(size = up + k - 1) -> Conv2d(k = 5)),
// decoder call sites /root/reference/network/SNN_models.py:110-129, w.r.t. the stage INPUT.  In the projected form (ss_upconv.hip)
//     g_P[src][tap][co] = sum over the output pixels whose tap lands on src of g_y[pix][co]      (adjoint of the gather: ss_upconv_cl_bwd_f32)
//     g_x[src][ci]      = sum_{tap, co} g_P[src][tap][co] * W[co][ci][tap]                       (was: fp32 library GEMM / ss_gemm6_f32 on g_P in HBM)
// g_P is 25 * C_out floats per source pixel — 5.76 GB per step for deconv1 at BASELINE config 3, written by the adjoint kernel and read
// back by the GEMM (0.06 of the HBM roofline on the stage's own I/O).  Here a workgroup owns a tile of 4 x 32 source pixels:
//   1. the g_y window of the tile (<= 13 rows x 72 columns, 32 output channels at a time) is staged in LDS, zero outside the image;
//   2. a wavefront owns ONE source row x 32 source columns (= the M dimension of v_mfma_f32_32x32x16_bf16) and one half of the 32 channels;
//      a lane forms g_P[src][tap][8 consecutive co] — exactly the A fragment of a k-step, k = (tap, co) — from the window with the adjoint
//      kernel's own summation order (row sums left to right, rows top to bottom; the row sums of the previous window row are carried in
//      registers, so a window pixel is read from LDS once per channel octet for all 25 taps) — bit-identical to ss_upconv_cl_bwd_f32's g_P;
//   3. the 8 values are split into three bf16 terms in registers and multiplied with the weight fragments (split once, fragment-ordered by
//      upconv_bwd_dgrad_prep_kernel, streamed L2 -> LDS, double-buffered, one k-step per stage) with the SIX cross terms of ss_gemm6_f32:
//      fp32-product accuracy, |g_x - float64| <= 2^-21 sum |g_P| |W| (asserted in tests/), on the bf16 matrix cores;
//   4. the two channel-half partial sums of a source row are added through LDS (fixed order) and g_x is written once.
// HBM / L2 traffic per stage: g_y once (+ window halo), the weights from L2, g_x once.  The weight gradient of the stage is the existing
// fused adjoint + exact MFMA contraction (ss_upconv_bwd_fused_f32, now without its g_P store) — g_P reaches HBM in neither.
// C_in > 64: the ci range is cut into blocks of 64 handled by different workgroups (the A fragments are rebuilt per block: the VALU work
// of step 2 is duplicated, the MFMA work is not).
#include "ss_common.hpp"

namespace {

#ifndef SS_DG_ABLATE
#define SS_DG_ABLATE 0                         // development aid (make variant DEFS=-DSS_DG_ABLATE=mask; tools/bench_upconv_bwd.py SS_LIB=...): 1 weight
#endif                                       // stream, 2 row sums, 4 MFMAs, 8 bf16 split, 16 per-k-step barrier — wrong results, timing only
#ifndef SS_DG_PIPE
#define SS_DG_PIPE 1                           // the next k-step's fragment is split before (1) / after (0) this k-step's MFMAs in program order
#endif
constexpr int kDgThreads = 256;              // 4 wavefronts = the 4 source rows of a tile; TWO workgroups per CU (74 KB of LDS each): one loads its
                                             // window while the other computes (the first version — one 8-wavefront workgroup per CU, 155 KB — spent
                                             // 0.9 of its 2.1 ms at deconv1 in unoverlapped per-tile latency: profiles/r03/upconv_bwd_dgrad_ablations.log)
constexpr int kDgTR = 4;                     // source rows per tile (one per wavefront)
constexpr int kDgTC = 32;                    // source columns per tile = MFMA M
constexpr int kDgWR = 13;                    // window rows: 4 source rows x <= 3 replicas + 4 tap rows (measured max 13 at every pyramid level)
constexpr int kDgWCmax = 72;                 // window columns of real data: 32 source columns (measured max 72: 40 -> 84)
constexpr int kDgWC = kDgWCmax + 2;          // + 2 slack columns (the unconditional 6th / 7th column reads of a 2-replica pixel: multiplied by 0)
constexpr int kDgCh = 16;                    // output channels per chunk = the k extent of one MFMA k-step
constexpr int kDgPix = kDgCh * 4;            // bytes per window pixel: 4 granules of 16 B, granule g of window column c stored at slot g ^ ((c >> 1) & 3):
                                             // lanes = consecutive source columns step 2 window columns, so 8 lanes x 16 B hit 8 distinct bank groups
constexpr int kDgRowB = kDgWC * kDgPix;
constexpr int kDgStage = 3 * 2 * 1024;       // one k-step of weights: [split term][ci tile of 32][lane][8 bf16]

// weight [C_out][C_in][5][5] fp32 -> Bf[ci block of 64][chunk of 16 co][s][split][tile][lane][8] bf16, s = (4 - ky) * 5 + kx (the order in
// which the row scan completes the taps), element e of a lane = split term of W[co = 16 chunk + 8 (lane >> 5) + e][ci = 64 blk + 32 tile
// + (lane & 31)][ky][kx]  (round-to-nearest split, as gemm6_prep_b_kernel)
__global__ __launch_bounds__(kBlock) void upconv_bwd_dgrad_prep_kernel(const float* __restrict__ W, unsigned short* __restrict__ Bf, int Cin, int Cout)
{
    const int NCH = Cout / kDgCh;
    const long long total = (long long)(Cin / 64) * NCH * 25 * 3 * 2 * 64;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int lane = (int)(i & 63);
        long long r = i >> 6;
        const int t = (int)(r % 2); r /= 2;
        const int sp = (int)(r % 3); r /= 3;
        const int s = (int)(r % 25); r /= 25;
        const int c = (int)(r % NCH); const int blk = (int)(r / NCH);
        const int ky = 4 - s / 5, kx = s % 5;
        const int ci = 64 * blk + 32 * t + (lane & 31);
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int co = kDgCh * c + 8 * (lane >> 5) + e;
            const float v = W[(((long long)co * Cin + ci) * 5 + ky) * 5 + kx];
            const unsigned short h1 = narrow<SS_DT_BF16>(v);
            const float r1 = v - widen<SS_DT_BF16>(h1);
            const unsigned short h2 = narrow<SS_DT_BF16>(r1);
            const float r2 = r1 - widen<SS_DT_BF16>(h2);
            o[e] = sp == 0 ? h1 : (sp == 1 ? h2 : narrow<SS_DT_BF16>(r2));
        }
        *reinterpret_cast<u16x8*>(Bf + i * 8) = o;
    }
}

// 8 fp32 values -> three bf16 terms (round to nearest: residuals <= 2^-8, 2^-16), as ss_gemm6_f32 splits its A operand
__device__ __forceinline__ void dg_split(const float (&v)[8], float sgn, s16x8& ah, s16x8& am, s16x8& al)
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = v[e] * sgn;
#if SS_DG_ABLATE & 8
        ah[e] = (short)(__float_as_uint(x) >> 16); am[e] = (short)__float_as_uint(x); al[e] = am[e];
#else
        const __bf16 h1 = (__bf16)x;                                            // v_cvt_pk_bf16_f32 on gfx950 (round to nearest even)
        const float r1 = x - (float)h1;
        const __bf16 h2 = (__bf16)r1;
        const float r2 = r1 - (float)h2;
        const __bf16 h3 = (__bf16)r2;
        ah[e] = __builtin_bit_cast(short, h1); am[e] = __builtin_bit_cast(short, h2); al[e] = __builtin_bit_cast(short, h3);
#endif
    }
}

template <int COUT>
__global__ __launch_bounds__(kDgThreads, 2) void upconv_bwd_dgrad_kernel(const float* __restrict__ gy, const unsigned short* __restrict__ Bf,
                                                                         const int* __restrict__ y_lo, const int* __restrict__ y_hi,
                                                                         const int* __restrict__ x_lo, const int* __restrict__ x_hi,
                                                                         float* __restrict__ gx, int NB, int h, int w, int H, int W, int CIN)
{
    constexpr int NCH = COUT / kDgCh;
    __shared__ __attribute__((aligned(16))) unsigned char wnd[kDgWR * kDgRowB];
    __shared__ __attribute__((aligned(16))) unsigned char bst[2 * kDgStage];
    const int lane = threadIdx.x & 63;
    const int mb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);            // wavefront = source row of the tile
    const int NBLK = CIN / 64, RG = (h + kDgTR - 1) / kDgTR, CG = (w + kDgTC - 1) / kDgTC;
    const long long n_tiles = (long long)NB * RG * CG * NBLK;
    // a contiguous range of tiles per workgroup, neighbouring ranges on the same XCD (window halos and the weights are shared through its L2)
    const unsigned g = xcd_remap(blockIdx.x, gridDim.x);
    const long long t_begin = n_tiles * g / gridDim.x, t_end = n_tiles * (g + 1) / gridDim.x;
#pragma unroll 1
    for (long long tl = t_begin; tl < t_end; ++tl) {
        const int blk = (int)(tl % NBLK);
        long long rr = tl / NBLK;
        const int cg = (int)(rr % CG); rr /= CG;
        const int rg = (int)(rr % RG);
        const int nb = (int)(rr / RG);
        const int sy0 = kDgTR * rg, sx0 = kDgTC * cg, nrow = min(kDgTR, h - sy0);
        const int wy0 = y_lo[sy0] - 4, WRt = y_hi[sy0 + nrow - 1] - wy0;
        const int wx0 = x_lo[sx0] - 4, WCt = x_hi[min(sx0 + kDgTC - 1, w - 1)] - wx0;
        const bool active = mb < nrow;                                          // wave-uniform
        const int sy = min(sy0 + mb, h - 1);
        const int ylo = __builtin_amdgcn_readfirstlane(y_lo[sy]);
        const int ry = __builtin_amdgcn_readfirstlane(y_hi[sy]) - ylo;          // replicas of this wavefront's source row: wave-uniform
        const int sxc = min(sx0 + (lane & 31), w - 1);
        const int xlo = x_lo[sxc], rx = x_hi[sxc] - xlo;
        const float m1 = rx > 1 ? 1.f : 0.f, m2 = rx > 2 ? 1.f : 0.f;           // fma(r, 1, cs) == cs + r, fma(r, 0, cs) == cs exactly: no selects
        const int wc0 = xlo - 4 - wx0;                                          // window column of this lane's first tap column
        const unsigned char* const row_base = wnd + (ylo - 4 - wy0) * kDgRowB;
        const int g0 = 2 * (lane >> 5);                                         // this lane's channel octet = granules g0, g0 + 1 of a pixel
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        bool neg = false;
        // row sums of one window row for the 5 horizontal taps: cs[kx] = row[4 - kx] (+ row[5 - kx] (+ row[6 - kx])), the adjoint kernel's order
        auto rowsums = [&](int j, float (&cs)[5][8]) {
            const unsigned char* const rp = row_base + j * kDgRowB;
#pragma unroll
            for (int c = 0; c < 7; ++c) {                                       // columns stream through 8 registers: column c is tap kx's first
                const int wc = wc0 + c;                                         // column for kx = 4 - c, its second for 5 - c, its third for 6 - c
                const unsigned char* const pp = rp + wc * kDgPix + ((g0 ^ ((wc >> 1) & 3)) << 4);
                const f4 a = *reinterpret_cast<const f4*>(pp);
                const f4 b = *reinterpret_cast<const f4*>(reinterpret_cast<const unsigned char*>((uintptr_t)pp ^ 16u));   // granule g0 + 1: slot ^ 1
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float r = e < 4 ? a[e & 3] : b[e & 3];
                        if (c == 4 - kx) cs[kx][e] = r;
                        else if (c == 5 - kx) cs[kx][e] = __builtin_fmaf(r, m1, cs[kx][e]);
                        else if (c == 6 - kx) cs[kx][e] = __builtin_fmaf(r, m2, cs[kx][e]);
                    }
                }
            }
        };
#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {
            __syncthreads();                                                    // the previous chunk's / tile's readers of the window are done
            // ---- g_y window (16 channels of this chunk) -> LDS, zero outside the image and in the two slack columns.  All loads of a thread
            //      are issued before its first store.
            {
                constexpr int kPerRow = kDgWC * 4, kIter = (kDgWR * kPerRow + kDgThreads - 1) / kDgThreads, kBatch = 8;
#pragma unroll 1
                for (int u0 = 0; u0 < kIter; u0 += kBatch) {
                    f4 buf[kBatch];
#pragma unroll
                    for (int v = 0; v < kBatch; ++v) {
                        const int i = threadIdx.x + kDgThreads * (u0 + v);
                        const int wy = i / kPerRow, rem = i - wy * kPerRow;
                        const int col = rem >> 2, q = rem & 3;
                        const int y = wy0 + wy, x = wx0 + col;
                        buf[v] = (f4){0.f, 0.f, 0.f, 0.f};
                        if (wy < WRt && col < WCt && y >= 0 && y < H && x >= 0 && x < W)
                            buf[v] = load_stream(reinterpret_cast<const f4*>(gy + (((long long)nb * H + y) * W + x) * COUT + kDgCh * c) + q);
                    }
#pragma unroll
                    for (int v = 0; v < kBatch; ++v) {
                        const int i = threadIdx.x + kDgThreads * (u0 + v);
                        const int wy = i / kPerRow, rem = i - wy * kPerRow;
                        const int col = rem >> 2, q = rem & 3;
                        if (wy < WRt && col < WCt + 2) *reinterpret_cast<f4*>(wnd + wy * kDgRowB + col * kDgPix + ((q ^ ((col >> 1) & 3)) << 4)) = buf[v];
                    }
                }
            }
            // ---- weight stage 0 of this (ci block, chunk)
            const unsigned char* const bsrc = reinterpret_cast<const unsigned char*>(Bf) + ((long long)blk * NCH + c) * 25 * kDgStage;
            f4 st0, st1 = {0.f, 0.f, 0.f, 0.f};
            st0 = *reinterpret_cast<const f4*>(bsrc + threadIdx.x * 16);
            if (threadIdx.x < 128) st1 = *reinterpret_cast<const f4*>(bsrc + (256 + threadIdx.x) * 16);
            *reinterpret_cast<f4*>(bst + threadIdx.x * 16) = st0;
            if (threadIdx.x < 128) *reinterpret_cast<f4*>(bst + (256 + threadIdx.x) * 16) = st1;
            __syncthreads();
            float p1[5][8];                                                     // row sums of the previous window row
            if (active && ry > 1) rowsums(ry - 2, p1);
#pragma unroll 1
            for (int grp = 0; grp < 5; ++grp) {                                 // vertical tap ky = 4 - grp is completed by window row ry - 1 + grp
                float cs[5][8];
                if (active && !((SS_DG_ABLATE & 2) && grp > 0)) {
                    if (ry > 2) {                                               // the third replica row (1 source row in ~30): re-summed, not carried,
                        float p2[5][8];                                         // and folded into p1 — (p2 + p1) + cs is the adjoint kernel's order
                        rowsums(ry - 3 + grp, p2);
#pragma unroll
                        for (int kx = 0; kx < 5; ++kx)
#pragma unroll
                            for (int e = 0; e < 8; ++e) p1[kx][e] = p2[kx][e] + p1[kx][e];
                    }
                    rowsums(ry - 1 + grp, cs);
                }
                // The bf16 MFMA's fp32 accumulation drifts down by ~2^-28 of the magnitude sum (ss_gemm6_f32): the sign of the running sum
                // alternates per vertical tap, which cancels the drift in expectation.
                if ((((c * 5 + grp) & 1) != 0) != neg) {
                    neg = !neg;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[t][r] = -acc[t][r];
                }
                const float sgn = neg ? -1.f : 1.f;
                // g_P[src][tap (ky, kx)][co]: rows oldest first, as ss_upconv_cl_bwd_f32
                auto tap_value = [&](int kx, float (&v)[8]) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[e] = ry > 1 ? p1[kx][e] + cs[kx][e] : cs[kx][e];
                    }
                };
                s16x8 ah, am, al;
                if (active) { float v[8]; tap_value(0, v); dg_split(v, sgn, ah, am, al); }
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
                    const int s = grp * 5 + kx;
                    const bool more = s + 1 < 25 && !(SS_DG_ABLATE & 1);
                    if (more) {
                        st0 = *reinterpret_cast<const f4*>(bsrc + (long long)(s + 1) * kDgStage + threadIdx.x * 16);
                        if (threadIdx.x < 128) st1 = *reinterpret_cast<const f4*>(bsrc + (long long)(s + 1) * kDgStage + (256 + threadIdx.x) * 16);
                    }
                    if (active) {
                        const unsigned char* const bk = bst + (s & 1) * kDgStage + lane * 16;
                        s16x8 b[6];                                             // [0,1] hi, [2,3] mid, [4,5] lo of ci tiles 0, 1
#pragma unroll
                        for (int u = 0; u < 6; ++u) b[u] = *reinterpret_cast<const s16x8*>(bk + u * 1024);
                        // the NEXT k-step's fragment is split (VALU) in the shadow of this k-step's twelve MFMAs
                        s16x8 nh = ah, nm = am, nl = al;
#if SS_DG_PIPE
                        if (kx < 4) { float v[8]; tap_value(kx + 1, v); dg_split(v, sgn, nh, nm, nl); }
#endif
#if SS_DG_ABLATE & 4
                        acc[0][0] += (float)(ah[0] + am[1] + al[2] + b[0][0] + b[3][1] + b[5][2]);
#else
#pragma unroll
                        for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[4 + u], acc[u], 0, 0, 0);
#pragma unroll
                        for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b[2 + u], acc[u], 0, 0, 0);
#pragma unroll
                        for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b[0 + u], acc[u], 0, 0, 0);
#pragma unroll
                        for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[2 + u], acc[u], 0, 0, 0);
#pragma unroll
                        for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b[0 + u], acc[u], 0, 0, 0);
#pragma unroll
                        for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[0 + u], acc[u], 0, 0, 0);
#endif
                        ah = nh; am = nm; al = nl;
#if !SS_DG_PIPE
                        if (kx < 4) { float v[8]; tap_value(kx + 1, v); dg_split(v, sgn, ah, am, al); }
#endif
                    }
                    if (more) {
                        unsigned char* const dst = bst + ((s + 1) & 1) * kDgStage;
                        *reinterpret_cast<f4*>(dst + threadIdx.x * 16) = st0;
                        if (threadIdx.x < 128) *reinterpret_cast<f4*>(dst + (256 + threadIdx.x) * 16) = st1;
                    }
                    if (!(SS_DG_ABLATE & 16)) __syncthreads();
                    __builtin_amdgcn_sched_barrier(0);                        // keep the later k-steps' VALU work below (register pressure)
                }
                if (active) {
#pragma unroll
                    for (int kx = 0; kx < 5; ++kx)
#pragma unroll
                        for (int e = 0; e < 8; ++e) p1[kx][e] = cs[kx][e];
                }
            }
        }
        // ---- tile epilogue: D[row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][col = lane & 31]
        if (active) {
            const float fin = neg ? -1.f : 1.f;
            const long long rowbase = ((long long)nb * h + (sy0 + mb)) * w;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int sxm = sx0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (sxm < w) store_out(gx + (rowbase + sxm) * CIN + 64 * blk + 32 * t + (lane & 31), acc[t][r] * fin);
                }
        }
    }
}

}  // namespace

extern "C" {

int ss_upconv_bwd_dgrad_supported(int Cin, int Cout, int k, int max_rows4, int max_cols32, int max_span)
{
    // max_rows4 / max_cols32: largest output-row / -column span (incl. the k - 1 taps) of 4 consecutive source rows / 32 consecutive source
    // columns; max_span: most output rows / columns one source pixel collects per tap — computed by the caller from the resize tables
    if (k != 5 || Cin < 64 || Cin % 64 != 0 || (Cout != 32 && Cout != 64 && Cout != 128 && Cout != 256)) return 0;      // C_out % 16 == 0
    return max_span >= 1 && max_span <= 3 && max_rows4 > 0 && max_rows4 <= kDgWR && max_cols32 > 0 && max_cols32 <= kDgWCmax;
}

long long ss_upconv_bwd_dgrad_ws_floats(int Cin, int Cout)
{
    if (Cin <= 0 || Cout <= 0 || Cin % 64 != 0 || Cout % 32 != 0) return 0;
    return (long long)25 * Cout * Cin * 3 / 2;                                  // the weight as three bf16 terms in fragment order
}

int ss_upconv_bwd_dgrad_f32(const float* g_out, const float* weight, const int* y_lo, const int* y_hi, const int* x_lo, const int* x_hi,
                            float* g_x, float* ws, long long NB, int Cin, int Cout, int h, int w, int H, int W, void* stream)
{
    if (!g_out || !weight || !y_lo || !y_hi || !x_lo || !x_hi || !g_x || !ws || NB <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SS_EINVAL;
    if (!ss_upconv_bwd_dgrad_supported(Cin, Cout, 5, kDgWR, kDgWCmax, 1)) return SS_EINVAL;     // shape only: the caller checked the extents
    if (!aligned16(g_out) || !aligned16(ws) || !aligned16(g_x) || NB * h * (long long)w > 0x7fffffffLL) return SS_EINVAL;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return SS_ELAUNCH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned short* Bf = reinterpret_cast<unsigned short*>(ws);
    hipLaunchKernelGGL(upconv_bwd_dgrad_prep_kernel, dim3(grid_for((long long)25 * Cout * Cin * 3 / 8, 4096)), dim3(kBlock), 0, s, weight, Bf, Cin, Cout);
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    const long long n_tiles = NB * ((h + kDgTR - 1) / kDgTR) * ((w + kDgTC - 1) / kDgTC) * (Cin / 64);
    const unsigned grid = (unsigned)(n_tiles < 2 * cus ? n_tiles : 2 * cus);     // two workgroups per CU (74 KB of LDS each), persistent over their tile ranges
#define SS_DG(CO) hipLaunchKernelGGL((upconv_bwd_dgrad_kernel<CO>), dim3(grid), dim3(kDgThreads), 0, s, g_out, Bf, y_lo, y_hi, x_lo, x_hi, g_x, \
                                     (int)NB, h, w, H, W, Cin)
    switch (Cout) {
        case 32: SS_DG(32); break;
        case 64: SS_DG(64); break;
        case 128: SS_DG(128); break;
        default: SS_DG(256); break;
    }
#undef SS_DG
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

}  // extern "C"

// ss_upconv.hip — NNConvUpsampling kernels (gather forms, fused projection + gather on the bf16 matrix cores) + their C-ABI entry points (include/ss_neuron.h).
#include "ss_common.hpp"

namespace {

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

}  // namespace

extern "C" {

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

}  // extern "C"

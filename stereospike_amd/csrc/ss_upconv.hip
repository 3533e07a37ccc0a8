// ss_upconv.hip — NNConvUpsampling kernels (gather forms, fused projection + gather on the bf16 matrix cores) + their C-ABI entry points (include/ss_neuron.h).
#include "ss_common.hpp"
#include <stdlib.h>

namespace {

#ifndef SS_CL_BWD_PREFETCH
#define SS_CL_BWD_PREFETCH 1                  // A/B: make variant VARIANT=nopf DEFS=-DSS_CL_BWD_PREFETCH=0 (profiles/r06/adjoint_prefetch_ab.log)
#endif

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

// a window row of the adjoint as it sits in HBM (16-bit activations: the 8-byte pattern, widened at use — half the registers of the widened form while a
// prefetched row waits for its turn)
template <int ACT, int VEC> struct ClRaw { typedef typename std::conditional<ACT == 0, typename std::conditional<VEC == 4, f4, float>::type, typename U16Vec<VEC>::type>::type type; };
template <int ACT, int VEC> __device__ __forceinline__ typename ClRaw<ACT, VEC>::type load_raw(const void* base, long long e)
{
    if constexpr (ACT == 0) return load_act<0, VEC>(base, e);
    else return *reinterpret_cast<const typename U16Vec<VEC>::type*>(static_cast<const unsigned short*>(base) + e);
}
template <int ACT, int VEC> __device__ __forceinline__ typename std::conditional<VEC == 4, f4, float>::type widen_raw(typename ClRaw<ACT, VEC>::type v)
{
    if constexpr (ACT == 0) return v;
    else if constexpr (VEC == 4) return (f4){widen<ACT>(v[0]), widen<ACT>(v[1]), widen<ACT>(v[2]), widen<ACT>(v[3])};
    else return widen<ACT>(v);
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

// GP = 0: g_P written as fp32; SS_DT_BF16 / SS_DT_F16: g_P written in that format (the 16-bit modes' backward GEMMs take 16-bit operands: no separate cast pass)
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
            if constexpr (SS_CL_BWD_PREFETCH && ACT != 0) {
            // the NEXT row's loads are issued before this row is reduced and stored (round 6: with one row in flight per lane every row paid a full memory
            // latency — the 16-bit form wrote g_P at 2 - 2.6 TB/s: 1.52 -> 1.25 ms for deconv3 at config 5's share.
            // fp32 input: the 28 extra registers cost a wavefront per SIMD and the kernel LOSES 10 % — the one-row form stays there; profiles/r06/adjoint_prefetch_ab.log)
            typedef typename ClRaw<ACT, VEC>::type raw_t;
            raw_t nxt[K + 2];
            auto fetch_row = [&](int j) {
                const int y = ylo - (K - 1) + j;
                const bool yok = y >= 0 && y < H;
#pragma unroll
                for (int c = 0; c < K + 2; ++c) {
                    const int x = xlo - (K - 1) + c;
                    const bool ok = yok && x >= 0 && x < W && c < K - 1 + rx;
                    nxt[c] = raw_t{};
                    if (ok) nxt[c] = load_raw<ACT, VEC>(g_out, gbase + (long long)(y * W + x) * C);
                }
            };
            fetch_row(0);
#pragma unroll 1
            for (int j = 0; j < ry + K - 1; ++j) {
                vec_t row[K + 2];
#pragma unroll
                for (int c = 0; c < K + 2; ++c) row[c] = widen_raw<ACT, VEC>(nxt[c]);          // (a zero pattern widens to +0: rows / columns outside the image)
                if (j + 1 < ry + K - 1) fetch_row(j + 1);
                vec_t cs[K];
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int b0 = K - 1 - kx;
                    cs[kx] = zero;
                    if (rx > 0) cs[kx] += row[b0];
                    if (rx > 1) cs[kx] += row[b0 + 1];
                    if (rx > 2) cs[kx] += row[b0 + 2];
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
            } else {
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
        } else if constexpr (K == 3 && VEC == 1) {
            // Prediction heads at large up-sampling ratios (x4 .. x15: a source pixel collects up to 16 x 16 output pixels per tap).  The nine rectangles are ONE
            // rectangle shifted by (ky, kx): a single scan of their union — rows ylo-2 .. yhi-1, columns xlo-2 .. xhi-1 — feeds all nine sums, every element loaded
            // once instead of up to nine times (round 6: 2 304 -> 324 loads per lane for predict_depth4; the per-tap loops below took 0.72 ms there at config 5's
            // share).  Same summation order as the per-tap form — within a row x ascending from zero, rows ascending — hence the same bits.
            vec_t acc[K][K];
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) acc[ky][kx] = zero;
            const int ya = max(ylo - (K - 1), 0), yb = min(yhi, H), xa = max(xlo - (K - 1), 0), xb = min(xhi, W);
            for (int y = ya; y < yb; ++y) {
                vec_t cs[K];
#pragma unroll
                for (int kx = 0; kx < K; ++kx) cs[kx] = zero;
                for (int x = xa; x < xb; ++x) {
                    const vec_t v = load_act<ACT, VEC>(g_out, gbase + (long long)(y * W + x) * C);
#pragma unroll
                    for (int kx = 0; kx < K; ++kx)
                        if (x >= xlo - kx && x < xhi - kx) cs[kx] += v;
                }
#pragma unroll
                for (int ky = 0; ky < K; ++ky)
                    if (y >= ylo - ky && y < yhi - ky) {
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) acc[ky][kx] += cs[kx];
                    }
            }
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) store_act<GP, VEC>(g_P, gpo + (ky * K + kx) * C, acc[ky][kx]);
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


template <int K, int ACT = 0>
int launch_cl_fwd(const float* P, const int* sy, const int* sx, const float* bias, void* out, int NB, int C, int h, int w,
                         int H, int W, hipStream_t s)
{
    const bool vec = (C % 4 == 0) && aligned16(P) && aligned16(out) && (!bias || aligned16(bias));
    const long long per_img = (long long)H * W * (vec ? C / 4 : C);
    dim3 grid((unsigned)((per_img + kBlock - 1) / kBlock), (unsigned)(NB < 65535 ? NB : 65535));
    if (C == 1) {
        // prediction heads (one channel): a thread's work per frame is nine loads and a store — let it walk several frames, its index arithmetic and table
        // look-ups paid once (round 6; A/B profiles/r06/heads_ab.log)
        static const int per_frame = getenv("SS_HEAD_GATHER_OLD") ? atoi(getenv("SS_HEAD_GATHER_OLD")) : 0;
        const long long gy = 16384 / (long long)grid.x;
        if (!per_frame) grid.y = (unsigned)(gy < 1 ? 1 : (gy > NB ? NB : gy));
    }
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

int ss_upconv_cl_bwd_lowp_dt(const void* g_out, int g_dtype, const int* y_lo, const int* y_hi, const int* x_lo, const int* x_hi,
                             void* g_P, int gp_dtype, long long NB, int k, int C, int h, int w, int H, int W, void* stream)
{
    if (!g_out || !y_lo || !y_hi || !x_lo || !x_hi || !g_P || NB < 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SS_EINVAL;
    if (k != 5 || (g_dtype != 0 && g_dtype != SS_DT_F16 && g_dtype != SS_DT_BF16)) return SS_EINVAL;
    if (gp_dtype != SS_DT_BF16 && !(gp_dtype == SS_DT_F16 && g_dtype == SS_DT_F16)) return SS_EINVAL;     // fp16 g_P: the fp16 mode's own gradients only
    if (NB > 0x7fffffffLL || (long long)H * W * C > 0x7fffffffLL || (long long)k * k * C * h * w > 0x7fffffffLL) return SS_EINVAL;
    if (NB == 0) return SS_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (gp_dtype == SS_DT_F16) return launch_cl_bwd<5, SS_DT_F16, SS_DT_F16>(g_out, y_lo, y_hi, x_lo, x_hi, g_P, (int)NB, C, h, w, H, W, s);
    if (g_dtype == 0) return launch_cl_bwd<5, 0, SS_DT_BF16>(g_out, y_lo, y_hi, x_lo, x_hi, g_P, (int)NB, C, h, w, H, W, s);
    if (g_dtype == SS_DT_F16) return launch_cl_bwd<5, SS_DT_F16, SS_DT_BF16>(g_out, y_lo, y_hi, x_lo, x_hi, g_P, (int)NB, C, h, w, H, W, s);
    return launch_cl_bwd<5, SS_DT_BF16, SS_DT_BF16>(g_out, y_lo, y_hi, x_lo, x_hi, g_P, (int)NB, C, h, w, H, W, s);
}

int ss_upconv_cl_bwd_lowp(const void* g_out, int g_dtype, const int* y_lo, const int* y_hi, const int* x_lo, const int* x_hi,
                          void* g_P_bf16, long long NB, int k, int C, int h, int w, int H, int W, void* stream)
{
    return ss_upconv_cl_bwd_lowp_dt(g_out, g_dtype, y_lo, y_hi, x_lo, x_hi, g_P_bf16, SS_DT_BF16, NB, k, C, h, w, H, W, stream);
}

}  // extern "C"

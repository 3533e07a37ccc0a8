// ss_upconv_box.hip — the decoder's backward on the BOX-SUM image (round 4; include/ss_neuron.h: ss_upconv_boxsum_f32, ss_upconv_box_dgrad_f32,
// ss_upconv_box_wgrad_f32).
//
// Reference: autograd through NNConvUpsampling (/root/reference/network/blocks.py:110-132: UpsamplingNearest2d(size = up + k - 1) -> Conv2d(k = 5)),
// decoder call sites /root/reference/network/SNN_models.py:110-129.  In the projected form of ss_upconv.hip the stage's backward went through the per-tap
// tensor g_P[src][tap][co] = sum of g_y over the output pixels whose tap lands on src — 25 C_out floats per SOURCE pixel, 6.25 x the stage's own output
// gradient (5.76 / 2.9 / 1.5 / 0.8 GB at BASELINE config 3), written once and read once or twice, or re-formed in registers per MFMA fragment by the fused
// kernels of round 3 (70 VALU instructions per k-step: 0.12 - 0.25 of the MFMA peak).  But g_P only ever holds RECTANGLE sums of g_y, and a nearest resize
// by ~2 has few distinct rectangles: with VR / HR the lists of distinct vertical / horizontal output ranges a (source index, tap) pair collects (id 0 = the
// empty range; host side: fused.box_tables, restated in oracle/np_upconv_box.py) and vmap[iy][ky], hmap[ix][kx] the maps into them,
//
//     B[nb][j][i][co]            = sum_{Y in VR[j]} ( sum_{X in HR[i]} g_y[nb][Y][X][co] )          — the box-sum image, ~ (H + 18)(W + 18) C_out values
//     g_P[nb][iy][ix][tap][co]   = B[nb][vmap[iy][ky]][hmap[ix][kx]][co]                            — bit for bit (same summation order as ss_upconv_cl_bwd_f32)
//     g_x[nb][iy][ix][ci]        = sum_{tap, co} B[...][co] * W[co][ci][tap]                        — a strided 5 x 5 convolution over B
//     g_W[co][ci][tap]           = sum_{nb, iy, ix} x[nb][iy][ix][ci] * B[...][co]                  — its weight gradient
//
// so the adjoint gather becomes ONE small HBM-bound kernel (boxsum: reads g_y once, writes B once, already split into the three bf16 planes every MFMA
// contraction of this engine wants — h + m + l == B exactly), and both contractions are implicit GEMMs whose operand fragments are plain 16-byte LDS reads
// through two small index maps: no VALU work on operands in the main loops, no g_P anywhere.
//
// HBM layout of B: [nb][C_out / 8][plane 3][NVR][NHR][8] bf16 — a window row of one 8-channel chunk and plane is contiguous.
#include "ss_common.hpp"

namespace {

constexpr int kBxCo = 8;                       // output channels per chunk of the plane layout (= half a k-step: a k-step is 2 taps x 8 channels)

__device__ __forceinline__ void bx_split3(const float (&v)[8], u16x8& h, u16x8& m, u16x8& l)
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const unsigned short h1 = narrow<SS_DT_BF16>(v[e]);
        const float r1 = v[e] - widen<SS_DT_BF16>(h1);
        const unsigned short h2 = narrow<SS_DT_BF16>(r1);
        const float r2 = r1 - widen<SS_DT_BF16>(h2);
        h[e] = h1; m[e] = h2; l[e] = narrow<SS_DT_BF16>(r2);
    }
}

// ---------------------------------------------------------------------------------------------------
// K1: g_y [NB][H][W][C_out] fp32 -> B planes.  A lane owns one horizontal range i and one 8-channel chunk and walks JS consecutive vertical ranges,
// carrying the horizontal sum of the last output row (consecutive ranges overlap in one row): ~3 pixel reads per B element instead of 4 - 9.
// Lanes: 4 chunks x 16 ranges per wavefront — a pixel's 4 x 32 B are one 128-B line, a chunk's 16 granules one 256-B store segment.
// ---------------------------------------------------------------------------------------------------
constexpr int kBxJS = 8;

template <int CG>                              // chunks per lane group: min(C_out / 8, 4)
__global__ __launch_bounds__(kBlock) void upconv_boxsum_kernel(const float* __restrict__ gy, const int* __restrict__ vr, const int* __restrict__ hr,
                                                               unsigned short* __restrict__ Bp, int NB, int H, int W, int COUT, int NVR, int NHR)
{
    const int NCH = COUT / kBxCo, NCG = NCH / CG;
    const int IPB = kBlock / CG;                                              // horizontal ranges per workgroup
    const int ib = (NHR + IPB - 1) / IPB, jb = (NVR + kBxJS - 1) / kBxJS;
    long long b = blockIdx.x;
    const int bi = (int)(b % ib); b /= ib;
    const int bj = (int)(b % jb); b /= jb;
    const int cg = (int)(b % NCG);
    const int nb = (int)(b / NCG);
    const int c = cg * CG + (int)(threadIdx.x % CG);
    const int i = bi * IPB + (int)(threadIdx.x / CG);
    if (i >= NHR) return;
    const int x0 = hr[2 * i], nx = hr[2 * i + 1];
    const float* const g0 = gy + ((long long)nb * H * W + x0) * COUT + kBxCo * c;
    int cy = -1;                                                               // output row whose horizontal sum is cached
    float cv[8];
    auto hsum = [&](int y, float (&o)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
        const float* p = g0 + (long long)y * W * COUT;
        for (int x = 0; x < nx; ++x) {
            const f4 a = *reinterpret_cast<const f4*>(p + (long long)x * COUT), bb = *reinterpret_cast<const f4*>(p + (long long)x * COUT + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] += a[e]; o[4 + e] += bb[e]; }
        }
    };
    const long long plane = (long long)NVR * NHR * kBxCo;
    unsigned short* const out0 = Bp + (((long long)nb * NCH + c) * 3) * plane + (long long)i * kBxCo;
    for (int jj = 0; jj < kBxJS; ++jj) {
        const int j = bj * kBxJS + jj;
        if (j >= NVR) break;
        const int y0 = vr[2 * j], ny = vr[2 * j + 1];
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (nx > 0) {
            for (int y = y0; y < y0 + ny; ++y) {
                float hs[8];
                if (y == cy) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) hs[e] = cv[e];
                } else {
                    hsum(y, hs);
                }
                if (y == y0 + ny - 1) {                                        // the next range most likely starts with this row
                    cy = y;
#pragma unroll
                    for (int e = 0; e < 8; ++e) cv[e] = hs[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += hs[e];
            }
        }
        u16x8 ph, pm, pl;
        bx_split3(acc, ph, pm, pl);
        unsigned short* const o = out0 + (long long)j * NHR * kBxCo;
        *reinterpret_cast<u16x8*>(o) = ph;
        *reinterpret_cast<u16x8*>(o + plane) = pm;
        *reinterpret_cast<u16x8*>(o + 2 * plane) = pl;
    }
}

// ---------------------------------------------------------------------------------------------------
// K2: data gradient  g_x[nb][iy][ix][ci] = sum_{tap, co} B[nb][vmap[iy][ky]][hmap[ix][kx]][co] * W[co][ci][tap]   (six bf16 cross terms, fp32-product accuracy)
// ---------------------------------------------------------------------------------------------------
// A workgroup (4 wavefronts; two workgroups per CU) owns a tile of 4 source rows x 32 source columns and 64 input channels; a wavefront owns a block of
// 4 rows x 8 columns of it (= the M dimension of v_mfma_f32_32x32x16_bf16: <= 1 / 8 of a block wasted on any width).  Per 8-channel chunk of C_out the tile's
// window of the three B planes is staged in LDS (16-byte pixels; row 0 and columns 0, 1 are zeros = the empty range); a k-step is 2 taps x 8 channels,
// the lane's A fragment of a plane ONE 16-byte LDS read at a per-lane address computed once per tile from the two maps; the weight — split once into three
// bf16 terms in fragment order, the sign of odd chunks flipped (upconv_box_dgrad_prep_kernel) — streams L2 -> LDS double-buffered, one k-step per stage
// and barrier.  Six cross terms hh, hm, mh, hl, lh, mm in ss_gemm6_f32's order; the running sum's sign alternates per chunk (the bf16 MFMA's fp32
// accumulation drifts down by ~2^-28 of the magnitude sum: ss_wgrad.hip).  |g_x - float64| <= 2^-21 sum |B| |W| element-wise.
constexpr int kB2Threads = 256;
constexpr int kB2TR = 4, kB2TC = 32;
constexpr int kB2WR = 18, kB2WC = 76;          // window rows (1 zero + <= 17 ranges) x columns (2 zero + <= 74 ranges)
constexpr int kB2Plane = kB2WR * kB2WC * 16;   // bytes of one plane of the window
constexpr int kB2KS = 13;                      // k-steps per chunk: 25 taps in pairs (the 26th is a zero-weight phantom)
constexpr int kB2Stage = 3 * 2 * 1024;         // one k-step of weights: [plane][ci tile of 32][lane][8 bf16]

// weight [C_out][C_in][5][5] fp32 -> Wf[ci block of 64][chunk][s][plane][tile][lane][8] bf16: element e of a lane = split term of
// (+ / -) W[co = 8 chunk + e][ci = 64 blk + 32 tile + (lane & 31)][tap = 2 s + (lane >> 5)]  (tap 25: zero; odd chunks negated)
__global__ __launch_bounds__(kBlock) void upconv_box_dgrad_prep_kernel(const float* __restrict__ W, unsigned short* __restrict__ Wf, int Cin, int Cout)
{
    const int NCH = Cout / kBxCo;
    const long long total = (long long)(Cin / 64) * NCH * kB2KS * 3 * 2 * 64;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int lane = (int)(i & 63);
        long long r = i >> 6;
        const int t = (int)(r % 2); r /= 2;
        const int sp = (int)(r % 3); r /= 3;
        const int s = (int)(r % kB2KS); r /= kB2KS;
        const int c = (int)(r % NCH); const int blk = (int)(r / NCH);
        const int tap = 2 * s + (lane >> 5);
        const int ci = 64 * blk + 32 * t + (lane & 31);
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int co = kBxCo * c + e;
            float v = tap < 25 ? W[((long long)co * Cin + ci) * 25 + tap] : 0.f;
            if (c & 1) v = -v;
            const unsigned short h1 = narrow<SS_DT_BF16>(v);
            const float r1 = v - widen<SS_DT_BF16>(h1);
            const unsigned short h2 = narrow<SS_DT_BF16>(r1);
            const float r2 = r1 - widen<SS_DT_BF16>(h2);
            o[e] = sp == 0 ? h1 : (sp == 1 ? h2 : narrow<SS_DT_BF16>(r2));
        }
        *reinterpret_cast<u16x8*>(Wf + i * 8) = o;
    }
}

template <int COUT>
__global__ __launch_bounds__(kB2Threads, 2) void upconv_box_dgrad_kernel(const unsigned short* __restrict__ Bp, const unsigned short* __restrict__ Wf,
                                                                         const int* __restrict__ vmap, const int* __restrict__ hmap,
                                                                         const int* __restrict__ tj, const int* __restrict__ ti,
                                                                         float* __restrict__ gx, int NB, int h, int w, int NVR, int NHR, int CIN)
{
    constexpr int NCH = COUT / kBxCo;
    __shared__ __attribute__((aligned(16))) unsigned char wnd[3 * kB2Plane];
    __shared__ __attribute__((aligned(16))) unsigned char bst[2 * kB2Stage];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NBLK = CIN / 64, RG = (h + kB2TR - 1) / kB2TR, CG = (w + kB2TC - 1) / kB2TC;
    const long long n_tiles = (long long)NB * RG * CG * NBLK;
    // zero row / zero columns of the window: written once (the staging below never touches them)
    for (int i = threadIdx.x; i < 3 * kB2WR * kB2WC; i += kB2Threads) {
        const int r = (i / kB2WC) % kB2WR, cc = i % kB2WC;
        if (r == 0 || cc < 2) *reinterpret_cast<f4*>(wnd + i * 16) = (f4){0.f, 0.f, 0.f, 0.f};
    }
    const unsigned g = xcd_remap(blockIdx.x, gridDim.x);
    const long long t_begin = n_tiles * g / gridDim.x, t_end = n_tiles * (g + 1) / gridDim.x;
    const long long plane_g = (long long)NVR * NHR * kBxCo;                    // elements of one plane of one chunk in HBM
#pragma unroll 1
    for (long long tl = t_begin; tl < t_end; ++tl) {
        const int blk = (int)(tl % NBLK);
        long long rr = tl / NBLK;
        const int cg = (int)(rr % CG); rr /= CG;
        const int rg = (int)(rr % RG);
        const int nb = (int)(rr / RG);
        const int sy0 = kB2TR * rg, sx0 = kB2TC * cg;
        const int j0 = tj[2 * rg], nj = tj[2 * rg + 1], i0 = ti[2 * cg], ni = ti[2 * cg + 1];
        // ---- this lane's pixel and its 13 A-fragment addresses (bytes inside a plane of the window)
        const int m = lane & 31;
        const int sy = min(sy0 + (m >> 3), h - 1), sx = min(sx0 + 8 * wv + (m & 7), w - 1);
        int rofs[5], cofs[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int jv = vmap[sy * 5 + q], iv = hmap[sx * 5 + q];
            rofs[q] = jv ? jv - j0 + 1 : 0;
            cofs[q] = iv ? iv - i0 + 2 : 0;
        }
        int addr[kB2KS];
#pragma unroll
        for (int s = 0; s < kB2KS; ++s) {
            // tap = 2 s + (lane >> 5): both candidates are compile-time, the lane half selects
            const int t0 = 2 * s, t1 = 2 * s + 1;
            const int r0 = rofs[t0 / 5], c0 = cofs[t0 % 5];
            const int r1 = t1 < 25 ? rofs[t1 / 5] : 0, c1 = t1 < 25 ? cofs[t1 % 5] : 0;
            const int rw = (lane >> 5) ? r1 : r0, cw = (lane >> 5) ? c1 : c0;
            addr[s] = (rw * kB2WC + (cw ^ ((rw >> 1) & 1))) * 16;
        }
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {
            __syncthreads();                                                    // the previous chunk's / tile's readers of the window and of the stages are done
            // ---- window of chunk c: 3 planes x nj rows x ni granules, HBM rows contiguous; loads of a thread issued in batches before its stores
            {
                const unsigned short* const src = Bp + (((long long)nb * NCH + c) * 3) * plane_g + ((long long)j0 * NHR + i0) * kBxCo;
                const int rows3 = 3 * nj;
                constexpr int kPerRow = 80, kBatch = 8;                         // 80 >= 74: idx / 80 by multiply-shift
                const int total = rows3 * kPerRow;
#pragma unroll 1
                for (int u0 = 0; u0 * kB2Threads < total; u0 += kBatch) {
                    f4 buf[kBatch];
#pragma unroll
                    for (int v = 0; v < kBatch; ++v) {
                        const int idx = threadIdx.x + kB2Threads * (u0 + v);
                        const int rp = idx / kPerRow, cc = idx - rp * kPerRow;
                        const int p = rp / nj, r = rp - p * nj;
                        buf[v] = (f4){0.f, 0.f, 0.f, 0.f};
                        if (rp < rows3 && cc < ni)
                            buf[v] = *reinterpret_cast<const f4*>(src + (long long)p * plane_g + ((long long)r * NHR + cc) * kBxCo);
                    }
#pragma unroll
                    for (int v = 0; v < kBatch; ++v) {
                        const int idx = threadIdx.x + kB2Threads * (u0 + v);
                        const int rp = idx / kPerRow, cc = idx - rp * kPerRow;
                        const int p = rp / nj, r = rp - p * nj;
                        if (rp < rows3 && cc < ni) {
                            const int rw = r + 1, cw = cc + 2;
                            *reinterpret_cast<f4*>(wnd + p * kB2Plane + (rw * kB2WC + (cw ^ ((rw >> 1) & 1))) * 16) = buf[v];
                        }
                    }
                }
            }
            // ---- weight stage 0 of this (ci block, chunk)
            const unsigned char* const bsrc = reinterpret_cast<const unsigned char*>(Wf) + ((long long)blk * NCH + c) * kB2KS * kB2Stage;
            f4 st0, st1 = {0.f, 0.f, 0.f, 0.f};
            st0 = *reinterpret_cast<const f4*>(bsrc + threadIdx.x * 16);
            if (threadIdx.x < 128) st1 = *reinterpret_cast<const f4*>(bsrc + (256 + threadIdx.x) * 16);
            *reinterpret_cast<f4*>(bst + threadIdx.x * 16) = st0;
            if (threadIdx.x < 128) *reinterpret_cast<f4*>(bst + (256 + threadIdx.x) * 16) = st1;
            __syncthreads();
            if (c > 0) {                                                        // sign of the running sum alternates per chunk (odd chunks' weights are negated)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] = -acc[t][r];
            }
#pragma unroll
            for (int s = 0; s < kB2KS; ++s) {
                const bool more = s + 1 < kB2KS;
                if (more) {
                    st0 = *reinterpret_cast<const f4*>(bsrc + (long long)(s + 1) * kB2Stage + threadIdx.x * 16);
                    if (threadIdx.x < 128) st1 = *reinterpret_cast<const f4*>(bsrc + (long long)(s + 1) * kB2Stage + (256 + threadIdx.x) * 16);
                }
                const unsigned char* const ap = wnd + addr[s];
                const s16x8 ah = *reinterpret_cast<const s16x8*>(ap), am = *reinterpret_cast<const s16x8*>(ap + kB2Plane),
                            al = *reinterpret_cast<const s16x8*>(ap + 2 * kB2Plane);
                const unsigned char* const bk = bst + (s & 1) * kB2Stage + lane * 16;
                s16x8 b[6];                                                     // [0,1] hi, [2,3] mid, [4,5] lo of ci tiles 0, 1
#pragma unroll
                for (int u = 0; u < 6; ++u) b[u] = *reinterpret_cast<const s16x8*>(bk + u * 1024);
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
                if (more) {
                    unsigned char* const dst = bst + ((s + 1) & 1) * kB2Stage;
                    *reinterpret_cast<f4*>(dst + threadIdx.x * 16) = st0;
                    if (threadIdx.x < 128) *reinterpret_cast<f4*>(dst + (256 + threadIdx.x) * 16) = st1;
                    __syncthreads();
                }
            }
        }
        // ---- tile epilogue: D[pixel = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][ci = lane & 31]; the sum carries the sign of the last chunk
        const float fin = ((NCH - 1) & 1) ? -1.f : 1.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pm = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int py = sy0 + (pm >> 3), px = sx0 + 8 * wv + (pm & 7);
                if (py < h && px < w) store_out(gx + (((long long)nb * h + py) * w + px) * CIN + 64 * blk + 32 * t + (lane & 31), acc[t][r] * fin);
            }
    }
}


// ---------------------------------------------------------------------------------------------------
// K3: weight gradient  g_W[co][ci][tap] = sum_{nb, iy, ix} x[nb][iy][ix][ci] * B[nb][vmap[iy][ky]][hmap[ix][kx]][co]   (x spikes: EXACT products, three bf16 terms of B)
// ---------------------------------------------------------------------------------------------------
// v_mfma_f32_32x32x16_bf16 with M = 4 taps x 8 channels of one chunk ("tap quad": 7 quads cover the 25 taps, 3 phantom rows), N = 32 input channels,
// K = 16 consecutive source pixels of a row.  A workgroup (4 wavefronts; two per CU) is one KIND — an 8-channel chunk of C_out and a block of 32 NT input
// channels — whose accumulators (<= 7 tiles of 32 x 32 per wavefront) stay in registers while it walks its slice of the 4 x 32-pixel tiles: the window of
// the chunk's three B planes is staged exactly as in the data-gradient kernel, the spike operand comes pre-transposed from upconv_bwd_xprep_kernel (one
// 16-byte load per lane and k-step).  The A fragment — 8 consecutive PIXELS of one (tap, channel) row, i.e. the window read against its grain — is two
// ds_read_b64_tr_b16 per plane: the LDS transpose read hands lane i of a 16-lane group column i of the 4 x 16 block whose rows the group's lanes address
// INDIVIDUALLY (measured: out[i][r] = in[lane 4 r + i / 4][element i % 4], profiles/r04/tr16.log), so each source lane points at "its" pixel through the
// two index maps and the gather along k costs nothing.  Partials -> ws[slice][co][tap][ci] -> upconv_box_wgrad_reduce_kernel (fixed order): deterministic.
template <int NT>                              // input-channel tiles per kind (1 | 2); a wavefront owns ci tile wv % NT and the tap quads q = wv / NT (mod 4 / NT)
__global__ __launch_bounds__(kB2Threads, 2) void upconv_box_wgrad_kernel(const unsigned short* __restrict__ Bp, const unsigned short* __restrict__ xT,
                                                                         const int* __restrict__ vmap, const int* __restrict__ hmap,
                                                                         const int* __restrict__ tj, const int* __restrict__ ti,
                                                                         float* __restrict__ ws, int NB, int h, int w, int NVR, int NHR, int CIN, int COUT, int KINDS)
{
    constexpr int QS = 4 / NT;                                                  // stride of a wavefront's tap quads
    constexpr int NQ = (7 + QS - 1) / QS;                                       // most quads a wavefront owns (7 | 4 | 2)
    __shared__ __attribute__((aligned(16))) unsigned char wnd[3 * kB2Plane];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NCH = COUT / kBxCo, CIB = CIN / (32 * NT);
    const int kind = (int)(blockIdx.x % KINDS), slice = (int)(blockIdx.x / KINDS), slices = (int)(gridDim.x / KINDS);
    const int c = kind / CIB, cib = kind - c * CIB;                              // chunk of C_out, block of input channels
    const int cit = cib * NT + wv % NT;                                          // this wavefront's ci tile (of CIN / 32)
    const int q0 = wv / NT;                                                      // first tap quad; then q0 + QS, ...
    const int RG = (h + kB2TR - 1) / kB2TR, CG = (w + kB2TC - 1) / kB2TC, KSR = (w + 15) / 16;
    const long long n_tiles = (long long)NB * RG * CG;
    const long long t_begin = n_tiles * slice / slices, t_end = n_tiles * (slice + 1) / slices;
    const long long plane_g = (long long)NVR * NHR * kBxCo;
    for (int i = threadIdx.x; i < 3 * kB2WR * kB2WC; i += kB2Threads) {
        const int r = (i / kB2WC) % kB2WR, cc = i % kB2WC;
        if (r == 0 || cc < 2) *reinterpret_cast<f4*>(wnd + i * 16) = (f4){0.f, 0.f, 0.f, 0.f};
    }
    // this lane as a SOURCE lane of the transpose reads: pixel L >> 2 of a 4-pixel sub-block, columns 4 (L & 3) .. + 3 of the 16-row half g of the M tile
    const int L = lane & 15, g = (lane >> 4) & 1, oct = lane >> 5;
    const int tq4 = 2 * g + ((L & 3) >> 1), coq = (L & 3) & 1;
    int kyq[NQ], kxq[NQ];
    bool real[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int t = 4 * (q0 + QS * u) + tq4;
        real[u] = (q0 + QS * u) < 7 && t < 25;
        kyq[u] = real[u] ? t / 5 : 0;
        kxq[u] = real[u] ? t - 5 * (t / 5) : 0;
    }
    f32x16 acc[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    const unsigned xoffT = (unsigned)(lane & 31) * 16u + (unsigned)(lane >> 5) * 8u;
#pragma unroll 1
    for (long long tl = t_begin; tl < t_end; ++tl) {
        const int cg = (int)(tl % CG);
        const long long rr = tl / CG;
        const int rg = (int)(rr % RG), nb = (int)(rr / RG);
        const int sy0 = kB2TR * rg, sx0 = kB2TC * cg;
        const int j0 = tj[2 * rg], nj = tj[2 * rg + 1], i0 = ti[2 * cg], ni = ti[2 * cg + 1];
        // ---- address parts of this lane: rowenc[row][quad] = r_w * WC * 16 | swizzle bit << 4;  cpart[half * 2 + rd][quad] = c_w * 16 (+ 8 for the odd channel quad)
        int rowenc[kB2TR][NQ], cpart[4][NQ];
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
#pragma unroll
            for (int r = 0; r < kB2TR; ++r) {
                const int sy = min(sy0 + r, h - 1);
                const int jv = real[u] ? vmap[sy * 5 + kyq[u]] : 0;
                const int rw = jv ? jv - j0 + 1 : 0;
                rowenc[r][u] = rw * (kB2WC * 16) | (((rw >> 1) & 1) << 4);
            }
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                const int sx = min(sx0 + 16 * (jx >> 1) + 8 * oct + 4 * (jx & 1) + (L >> 2), w - 1);
                const int iv = real[u] ? hmap[sx * 5 + kxq[u]] : 0;
                const int cw = iv ? iv - i0 + 2 : 0;
                cpart[jx][u] = cw * 16 + 8 * coq;
            }
        }
        __syncthreads();                                                        // the previous tile's readers of the window are done
        {
            const unsigned short* const src = Bp + (((long long)nb * NCH + c) * 3) * plane_g + ((long long)j0 * NHR + i0) * kBxCo;
            const int rows3 = 3 * nj;
            constexpr int kPerRow = 80, kBatch = 8;
            const int total = rows3 * kPerRow;
#pragma unroll 1
            for (int u0 = 0; u0 * kB2Threads < total; u0 += kBatch) {
                f4 buf[kBatch];
#pragma unroll
                for (int v = 0; v < kBatch; ++v) {
                    const int idx = threadIdx.x + kB2Threads * (u0 + v);
                    const int rp = idx / kPerRow, cc = idx - rp * kPerRow;
                    const int p = rp / nj, r = rp - p * nj;
                    buf[v] = (f4){0.f, 0.f, 0.f, 0.f};
                    if (rp < rows3 && cc < ni)
                        buf[v] = *reinterpret_cast<const f4*>(src + (long long)p * plane_g + ((long long)r * NHR + cc) * kBxCo);
                }
#pragma unroll
                for (int v = 0; v < kBatch; ++v) {
                    const int idx = threadIdx.x + kB2Threads * (u0 + v);
                    const int rp = idx / kPerRow, cc = idx - rp * kPerRow;
                    const int p = rp / nj, r = rp - p * nj;
                    if (rp < rows3 && cc < ni) {
                        const int rw = r + 1, cw = cc + 2;
                        *reinterpret_cast<f4*>(wnd + p * kB2Plane + (rw * kB2WC + (cw ^ ((rw >> 1) & 1))) * 16) = buf[v];
                    }
                }
            }
        }
        __syncthreads();
        const int nrow = min(kB2TR, h - sy0);
#pragma unroll
        for (int r = 0; r < kB2TR; ++r) {
            if (r < nrow) {                                                     // wave-uniform
                const long long srow = (long long)nb * h + sy0 + r;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    if (sx0 + 16 * half < w) {                                  // wave-uniform: the k-step has at least one real pixel (xT is zero padded)
                        const int ks = (sx0 >> 4) + half;
                        const s16x8 xf = *reinterpret_cast<const s16x8*>(xT + (srow * KSR + ks) * ((long long)CIN * 16) + (long long)cit * (32 * 16) + xoffT);
#pragma unroll
                        for (int u = 0; u < NQ; ++u) {
                            if (q0 + QS * u < 7) {                              // wave-uniform
                                const int rp = rowenc[r][u] & ~16, sb = rowenc[r][u] & 16;
                                const int a0 = rp + (cpart[2 * half][u] ^ sb), a1 = rp + (cpart[2 * half + 1][u] ^ sb);
#pragma unroll
                                for (int p = 0; p < 3; ++p) {
                                    typedef short s16x4 __attribute__((ext_vector_type(4)));
                                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                        (s16x4 __attribute__((address_space(3)))*)(wnd + p * kB2Plane + a0));
                                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                        (s16x4 __attribute__((address_space(3)))*)(wnd + p * kB2Plane + a1));
                                    const s16x8 af = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, xf, acc[u], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    // ---- partials: D[m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][ci = lane & 31], m = 8 (tap - 4 quad) + channel  ->  ws[slice][co][tap][ci]
    float* const wsl = ws + (long long)slice * COUT * 25 * CIN;
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        if (q0 + QS * u < 7) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int tap = 4 * (q0 + QS * u) + (mm >> 3), co = kBxCo * c + (mm & 7);
                if (tap < 25) wsl[((long long)co * 25 + tap) * CIN + 32 * cit + (lane & 31)] = acc[u][r];
            }
        }
    }
}

// g_W [C_out][C_in][25] (+)= sum over slices of ws[slice][co][tap][ci], slices in ascending order
__global__ __launch_bounds__(kBlock) void upconv_box_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ gw, int slices, int Cout, int Cin, int accumulate)
{
    const long long total = (long long)Cout * 25 * Cin;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int ci = (int)(i % Cin);
        const long long rr = i / Cin;
        const int tap = (int)(rr % 25), co = (int)(rr / 25);
        float a = 0.f;
        for (int sIdx = 0; sIdx < slices; ++sIdx) a += ws[(long long)sIdx * total + i];
        float* o = gw + ((long long)co * Cin + ci) * 25 + tap;
        *o = accumulate ? *o + a : a;
    }
}

}  // namespace

extern "C" {

long long ss_upconv_box_elems(long long NB, int Cout, int NVR, int NHR)
{
    if (NB <= 0 || Cout <= 0 || Cout % kBxCo != 0 || NVR <= 0 || NHR <= 0) return 0;
    return NB * (long long)Cout * 3 * NVR * NHR;                                // bf16 elements of the three planes
}

int ss_upconv_boxsum_f32(const float* g_out, const int* vr, const int* hr, void* box, long long NB, int Cout, int H, int W, int NVR, int NHR, void* stream)
{
    if (!g_out || !vr || !hr || !box || NB <= 0 || H <= 0 || W <= 0 || NVR <= 0 || NHR <= 0 || Cout <= 0 || Cout % kBxCo != 0) return SS_EINVAL;
    if (!aligned16(g_out) || !aligned16(box)) return SS_EINVAL;
    const int NCH = Cout / kBxCo;
    const int CG = NCH % 4 == 0 ? 4 : (NCH % 2 == 0 ? 2 : 1);
    const long long ib = (NHR + kBlock / CG - 1) / (kBlock / CG), jb = (NVR + kBxJS - 1) / kBxJS;
    const long long grid = NB * (NCH / CG) * jb * ib;
    if (grid > 0x7fffffffLL) return SS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned short* Bp = static_cast<unsigned short*>(box);
    if (CG == 4) hipLaunchKernelGGL((upconv_boxsum_kernel<4>), dim3((unsigned)grid), dim3(kBlock), 0, s, g_out, vr, hr, Bp, (int)NB, H, W, Cout, NVR, NHR);
    else if (CG == 2) hipLaunchKernelGGL((upconv_boxsum_kernel<2>), dim3((unsigned)grid), dim3(kBlock), 0, s, g_out, vr, hr, Bp, (int)NB, H, W, Cout, NVR, NHR);
    else hipLaunchKernelGGL((upconv_boxsum_kernel<1>), dim3((unsigned)grid), dim3(kBlock), 0, s, g_out, vr, hr, Bp, (int)NB, H, W, Cout, NVR, NHR);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_upconv_box_dgrad_supported(int Cin, int Cout, int k, int max_rows4, int max_cols32)
{
    // max_rows4 / max_cols32: most distinct-range ids (span incl. gaps) the maps of 4 consecutive source rows / 32 consecutive source columns reach — computed by
    // the caller from its tables (fused.box_tables)
    if (k != 5 || Cin < 64 || Cin % 64 != 0 || (Cout != 32 && Cout != 64 && Cout != 128 && Cout != 256)) return 0;
    return max_rows4 > 0 && max_rows4 <= kB2WR - 1 && max_cols32 > 0 && max_cols32 <= kB2WC - 2;
}

long long ss_upconv_box_dgrad_ws_floats(int Cin, int Cout)
{
    if (Cin <= 0 || Cout <= 0 || Cin % 64 != 0 || Cout % kBxCo != 0) return 0;
    return (long long)(Cin / 64) * (Cout / kBxCo) * kB2KS * kB2Stage / 4;       // the weight as three bf16 terms in fragment order
}

int ss_upconv_box_dgrad_f32(const void* box, const float* weight, const int* vmap, const int* hmap, const int* tile_rows, const int* tile_cols,
                            float* g_x, float* ws, long long NB, int Cin, int Cout, int h, int w, int NVR, int NHR, void* stream)
{
    if (!box || !weight || !vmap || !hmap || !tile_rows || !tile_cols || !g_x || !ws || NB <= 0 || h <= 0 || w <= 0 || NVR <= 0 || NHR <= 0) return SS_EINVAL;
    if (!ss_upconv_box_dgrad_supported(Cin, Cout, 5, 1, 1)) return SS_EINVAL;                    // shape only: the caller checked the extents
    if (!aligned16(box) || !aligned16(ws) || !aligned16(g_x) || NB * h * (long long)w > 0x7fffffffLL) return SS_EINVAL;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return SS_ELAUNCH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned short* Wf = reinterpret_cast<unsigned short*>(ws);
    const long long frag16 = (long long)(Cin / 64) * (Cout / kBxCo) * kB2KS * 3 * 2 * 64;
    hipLaunchKernelGGL(upconv_box_dgrad_prep_kernel, dim3(grid_for(frag16, 4096)), dim3(kBlock), 0, s, weight, Wf, Cin, Cout);
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    const long long n_tiles = NB * ((h + kB2TR - 1) / kB2TR) * ((w + kB2TC - 1) / kB2TC) * (Cin / 64);
    const unsigned grid = (unsigned)(n_tiles < 2 * cus ? n_tiles : 2 * cus);     // two workgroups per CU, persistent over their tile ranges
    const unsigned short* Bp = static_cast<const unsigned short*>(box);
#define SS_BD(CO) hipLaunchKernelGGL((upconv_box_dgrad_kernel<CO>), dim3(grid), dim3(kB2Threads), 0, s, Bp, Wf, vmap, hmap, tile_rows, tile_cols, g_x, \
                                     (int)NB, h, w, NVR, NHR, Cin)
    switch (Cout) {
        case 32: SS_BD(32); break;
        case 64: SS_BD(64); break;
        case 128: SS_BD(128); break;
        default: SS_BD(256); break;
    }
#undef SS_BD
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

static int box_wgrad_plan(int Cin, int Cout, int* NT, int* kinds, int* slices)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return 0;
    *NT = Cin % 64 == 0 ? 2 : 1;                                               // (a kind of 4 ci tiles would need 7 tap quads per wavefront: 256 registers + spills)
    *kinds = (Cout / kBxCo) * (Cin / (32 * *NT));
    int sl = (2 * cus) / *kinds;                                                // two workgroups per CU
    if (sl < 1) sl = 1;
    *slices = sl;
    return 1;
}

int ss_upconv_box_wgrad_supported(int Cin, int Cout, int k, int max_rows4, int max_cols32)
{
    if (k != 5 || Cin < 32 || Cin % 32 != 0 || Cout < kBxCo || Cout % kBxCo != 0) return 0;
    return max_rows4 > 0 && max_rows4 <= kB2WR - 1 && max_cols32 > 0 && max_cols32 <= kB2WC - 2;
}

long long ss_upconv_box_wgrad_ws_floats(int Cin, int Cout, long long NB, int h, int w)
{
    int NT = 0, kinds = 0, slices = 0;
    if (!ss_upconv_box_wgrad_supported(Cin, Cout, 5, 1, 1) || NB <= 0 || h <= 0 || w <= 0 || !box_wgrad_plan(Cin, Cout, &NT, &kinds, &slices)) return 0;
    return (long long)slices * Cout * 25 * Cin + (NB * h * ((w + 15) / 16) * Cin * 16 + 1) / 2 + 8;      // slice partials + the spike operand in fragment order (bf16)
}

int ss_upconv_box_wgrad_f32(const void* box, const float* x, const unsigned int* x_packed, const int* vmap, const int* hmap, const int* tile_rows,
                            const int* tile_cols, float* g_w, float* ws, long long NB, int Cin, int Cout, int h, int w, int NVR, int NHR, int accumulate,
                            void* stream)
{
    if (x_packed && (NB * h * (long long)w * Cin) % 16 != 0) return SS_EINVAL;
    if (!box || (!x && !x_packed) || !vmap || !hmap || !tile_rows || !tile_cols || !g_w || !ws || NB <= 0 || h <= 0 || w <= 0 || NVR <= 0 || NHR <= 0) return SS_EINVAL;
    if (!ss_upconv_box_wgrad_supported(Cin, Cout, 5, 1, 1) || !aligned16(box) || !aligned16(ws) || NB * h * (long long)w > 0x7fffffffLL) return SS_EINVAL;
    int NT = 0, kinds = 0, slices = 0;
    if (!box_wgrad_plan(Cin, Cout, &NT, &kinds, &slices)) return SS_ELAUNCH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    long long part = (long long)slices * Cout * 25 * Cin;
    part = (part + 3) / 4 * 4;                                                  // xT 16-byte aligned
    unsigned short* xT = reinterpret_cast<unsigned short*>(ws + part);
    if (x_packed) hipLaunchKernelGGL(upconv_bwd_xprep_kernel<true>, dim3(grid_for(NB * h * ((w + 15) / 16) * Cin, kMaxGridBwd)), dim3(kBlock), 0, s,
                                     static_cast<const void*>(x_packed), xT, NB * h, w, Cin);
    else hipLaunchKernelGGL(upconv_bwd_xprep_kernel<false>, dim3(grid_for(NB * h * ((w + 15) / 16) * Cin, kMaxGridBwd)), dim3(kBlock), 0, s,
                            static_cast<const void*>(x), xT, NB * h, w, Cin);
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    const unsigned grid = (unsigned)(kinds * slices);
    const unsigned short* Bp = static_cast<const unsigned short*>(box);
#define SS_BW(NT_) hipLaunchKernelGGL((upconv_box_wgrad_kernel<NT_>), dim3(grid), dim3(kB2Threads), 0, s, Bp, xT, vmap, hmap, tile_rows, tile_cols, ws, \
                                      (int)NB, h, w, NVR, NHR, Cin, Cout, kinds)
    if (NT == 2) SS_BW(2); else SS_BW(1);
#undef SS_BW
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    hipLaunchKernelGGL(upconv_box_wgrad_reduce_kernel, dim3(grid_for((long long)Cout * 25 * Cin, 1024)), dim3(kBlock), 0, s, ws, g_w, slices, Cout, Cin, accumulate);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

}  // extern "C"

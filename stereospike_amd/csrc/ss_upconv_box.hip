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
#include <stdlib.h>

namespace {

#ifndef SS_BX_ABLATE
#define SS_BX_ABLATE 0                        // development aid (make variant DEFS=-DSS_BX_ABLATE=mask): data-gradient kernel without 1 window traffic,
#endif                                       // 2 weight stream, 4 MFMAs, 8 per-stage barriers — wrong results, timing only (profiles/r04/box_dgrad_ablations.log)
#ifndef SS_BX_TRACE
#define SS_BX_TRACE 0                         // development aid: thread 0 of workgroup 0 stamps s_memtime over its first items of the data-gradient kernel (ss_debug_box_trace)
#endif
#if SS_BX_TRACE
__device__ unsigned long long bx_trace[64][16];
#define BX_STAMP(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0 && titem < 64) bx_trace[titem][slot] = clock64(); } while (0)
#else
#define BX_STAMP(slot) do { } while (0)
#endif
constexpr int kBxCo = 8;                       // output channels per chunk of the plane layout (= half a k-step: a k-step is 2 taps x 8 channels)

// 16-bit activation modes (DT != 0, round 5): g_y arrives in the 16-bit format and the box sum — fp32 additions of <= 9 widened values, the fp32 kernel's order —
// is stored as ONE plane of the format: one more rounding of the class the mode already has at every layer crossing, and exactly what the modes' g_P path does
// (ss_upconv_cl_bwd_lowp writes every g_P element — the same rectangle sum — as bf16).  Measured end to end (tests/test_gpu_04_x16_parity.py): weight tensors
// 7.4e-3 -> see profiles/r05 of their 1e-2 bar in bf16.  (The plane count stays a compile-time trait: two planes — hi + lo, 16 significand bits in bf16 — were the
// first build; 2 MFMAs per k-step and twice the plane traffic for a rounding the g_P path does not spare either.)
// The weight is ONE term (rounded once to the format), so a k-step costs NP MFMAs instead of six (data gradient) / three (weight gradient).
template <int DT> struct BxT { static constexpr int NP = DT == 0 ? 3 : 1, NW = DT == 0 ? 3 : 1; };
// Round 6: with ONE plane and ONE weight term an item of the contraction kernels (a tile x an 8-channel chunk) holds a third / a sixth of the fp32 mode's MFMAs
// behind the same staging, barriers and address work — the 16-bit kernels ran at 0.12 - 0.2 of the MFMA peak on overhead.  The chunks of a frame are adjacent
// in HBM exactly as the planes of a chunk are ([nb][chunk][plane][NVR][NHR][8] with one plane), so in these modes the window's planes are NG CONSECUTIVE CHUNKS
// (template argument NG: 2 where C_out % 16 == 0, else 1): the data gradient adds their products into the same accumulators (a k-step = NG x NT MFMAs), the weight
// gradient keeps one accumulator set per chunk and shares the spike fragments and the addresses between them.

template <int DT> __device__ __forceinline__ void bx_split(const float (&v)[8], u16x8 (&pl)[BxT<DT>::NP])
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const unsigned short h1 = round_op<DT>(v[e]);
        pl[0][e] = h1;
        if constexpr (BxT<DT>::NP > 1) {
            const float r1 = v[e] - widen_op<DT>(h1);
            const unsigned short h2 = round_op<DT>(r1);
            pl[1][e] = h2;
            if constexpr (BxT<DT>::NP > 2) pl[2][e] = round_op<DT>(r1 - widen_op<DT>(h2));
        }
    }
}

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
#ifndef SS_BX_HSUM_UNROLL
#define SS_BX_HSUM_UNROLL 1                    // the <= 3 pixel loads of a range row issued together, additions in x order (same bits): 0.65 -> 0.61 ms (boxsum_variants.log)
#endif
#ifndef SS_BX_JS
#define SS_BX_JS 4                             // vertical ranges a lane walks (A/B: profiles/r04/boxsum_variants.log)
#endif
constexpr int kBxJS = SS_BX_JS;

template <int CG, int DT = 0>                  // chunks per lane group: min(C_out / 8, 4)
__global__ __launch_bounds__(kBlock) void upconv_boxsum_kernel(const typename ActT<DT>::type* __restrict__ gy, const int* __restrict__ vr, const int* __restrict__ hr,
                                                               unsigned short* __restrict__ Bp, int NB, int H, int W, int COUT, int NVR, int NHR)
{
    constexpr int NP = BxT<DT>::NP;
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
    const typename ActT<DT>::type* const g0 = gy + ((long long)nb * H * W + x0) * COUT + kBxCo * c;
    int cy = -1;                                                               // output row whose horizontal sum is cached
    float cv[8];
    auto hsum = [&](int y, float (&o)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
        const typename ActT<DT>::type* p = g0 + (long long)y * W * COUT;
        if constexpr (DT != 0) {                                                // 8 channels = ONE 16-byte load per pixel, widened exactly; same order of additions
            u16x8 a16[3];
#pragma unroll
            for (int x = 0; x < 3; ++x)
                if (x < nx) a16[x] = *reinterpret_cast<const u16x8*>(p + (long long)x * COUT);
#pragma unroll
            for (int x = 0; x < 3; ++x)
                if (x < nx) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += widen<DT>(a16[x][e]);
                }
            for (int x = 3; x < nx; ++x) {
                const u16x8 a2 = *reinterpret_cast<const u16x8*>(p + (long long)x * COUT);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += widen<DT>(a2[e]);
            }
        } else {
#if SS_BX_HSUM_UNROLL
        // ranges are <= 3 pixels wide (a nearest resize by ~2): the loads of a row go out together, the additions stay in x order (same bits)
        f4 a[3], bb[3];
#pragma unroll
        for (int x = 0; x < 3; ++x)
            if (x < nx) { a[x] = *reinterpret_cast<const f4*>(p + (long long)x * COUT); bb[x] = *reinterpret_cast<const f4*>(p + (long long)x * COUT + 4); }
#pragma unroll
        for (int x = 0; x < 3; ++x)
            if (x < nx) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] += a[x][e]; o[4 + e] += bb[x][e]; }
            }
        for (int x = 3; x < nx; ++x) {
            const f4 a2 = *reinterpret_cast<const f4*>(p + (long long)x * COUT), b2 = *reinterpret_cast<const f4*>(p + (long long)x * COUT + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] += a2[e]; o[4 + e] += b2[e]; }
        }
#else
        for (int x = 0; x < nx; ++x) {
            const f4 a = *reinterpret_cast<const f4*>(p + (long long)x * COUT), bb = *reinterpret_cast<const f4*>(p + (long long)x * COUT + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] += a[e]; o[4 + e] += bb[e]; }
        }
#endif
        }
    };
    const long long plane = (long long)NVR * NHR * kBxCo;
    unsigned short* const out0 = Bp + (((long long)nb * NCH + c) * NP) * plane + (long long)i * kBxCo;
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
        unsigned short* const o = out0 + (long long)j * NHR * kBxCo;
        if constexpr (DT == 0) {
            u16x8 ph, pm, pl;
            bx_split3(acc, ph, pm, pl);
            *reinterpret_cast<u16x8*>(o) = ph;
            *reinterpret_cast<u16x8*>(o + plane) = pm;
            *reinterpret_cast<u16x8*>(o + 2 * plane) = pl;
        } else {
            u16x8 pq[NP];
            bx_split<DT>(acc, pq);
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<u16x8*>(o + q * plane) = pq[q];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Window of the box planes in LDS, shared by the two contraction kernels
// ---------------------------------------------------------------------------------------------------
// A tile is <= 4 source rows (cut shorter by the host where a triple-replicated row would need more than 13 distinct vertical ranges) x 32 source columns.
// Per 8-channel chunk of C_out its window of the three planes sits in LDS as 16-byte pixels: [plane][WR rows][WC columns] + ONE zero pixel per plane behind
// them (= the empty range, and what a phantom tap reads: a lane's address is either a window slot or the zero pixel).  Column slot = c ^ ((r >> 1) & 1): a
// wavefront's lanes are 4 rows x 8 columns of source pixels = every second window row / column, and with this swizzle the 16 lanes of each ds_read_b128
// group hit 16 distinct 16-byte bank groups (brute force over all offsets: conflict-free for WC = 76 and 80).
constexpr int kB2Threads = 256;
constexpr int kB2TR = 4, kB2TC = 32;
constexpr int kB2WR = 15, kB2WC = 76;          // window rows x columns = the most distinct vertical / horizontal ranges a tile may reach
constexpr int kB2Zero = kB2WR * kB2WC * 16;    // byte offset of the zero pixel inside a plane
constexpr int kB2Plane = kB2Zero + 16;         // bytes of one plane of the window (18256)
constexpr int kB2KS = 13;                      // k-steps per chunk: 25 taps in pairs (the 26th is a zero-weight phantom)

__device__ __forceinline__ int bx_slot(int rw, int cw) { return (rw * kB2WC + (cw ^ ((rw >> 1) & 1))) * 16; }      // rw, cw: window row / column (0-based)
// byte offset (inside a plane) of the pixel of range ids (jv, iv); either id 0 = the empty range -> the zero pixel
__device__ __forceinline__ int bx_addr(int jv, int iv, int j0, int i0) { return (jv && iv) ? bx_slot(jv - j0, iv - i0) : kB2Zero; }

__device__ __forceinline__ void bx_zero_borders(unsigned char* wnd, int np = 3)
{
    if ((int)threadIdx.x < np) *reinterpret_cast<f4*>(wnd + threadIdx.x * kB2Plane + kB2Zero) = (f4){0.f, 0.f, 0.f, 0.f};
}

// The window of (frame nb, chunk c) — 3 planes x nj rows x ni granules, HBM rows contiguous — travels HBM -> registers -> LDS in two halves so that the
// NEXT window's loads are in flight while the current one is multiplied.  Thread t < 228 owns plane t / 76 and window column t % 76 and walks the 15 rows:
// the per-row address arithmetic is scalar (uniform row base + one per-thread 32-bit offset; LDS: one per-thread base + an immediate per row), so a
// staging costs ~15 loads + 15 stores per thread and next to no VALU work (the first form spent ~600 VALU instructions per window on index math —
// 6 per MFMA of the item: profiles/r04/pmc_box_v3).
constexpr int kWinRegs = kB2WR;

struct BxLane { unsigned voff_plane; int cc; int lds0, lds1; bool act; };

__device__ __forceinline__ BxLane bx_lane(long long plane_g, int plane_bytes = kB2Plane, int np = 3)
{
    BxLane L;
    const int t = threadIdx.x, p = t / kB2WC;
    L.cc = t - p * kB2WC;
    L.act = p < np;
    L.voff_plane = (unsigned)((L.act ? p : 0) * plane_g + L.cc * kBxCo);       // elements from the window's first granule (row 0)
    L.lds0 = (L.act ? p : 0) * plane_bytes + L.cc * 16;                            // rows with swizzle bit 0
    L.lds1 = (L.act ? p : 0) * plane_bytes + (L.cc ^ 1) * 16;                      // rows with swizzle bit 1 ((r >> 1) & 1)
    return L;
}

__device__ __forceinline__ void bx_win_load(f4 (&buf)[kWinRegs], const BxLane& L, const unsigned short* __restrict__ src, int NHR, int nj, int ni)
{
    // (an execution-mask-free form — clamped column, scalar row branch only — measured SLOWER: 1.27 -> 1.53 ms on deconv1, profiles/r04/bench_box_bwd_v6.log)
    const bool on = L.act && L.cc < ni;
#pragma unroll
    for (int r = 0; r < kB2WR; ++r) {
        buf[r] = (f4){0.f, 0.f, 0.f, 0.f};
        if (on && r < nj) buf[r] = *reinterpret_cast<const f4*>(src + (long long)r * NHR * kBxCo + L.voff_plane);
    }
}

// rows [R0, R1) only: the contraction kernels issue a window a few rows per weight stage, BEHIND that stage's weight loads — the vector-memory counter
// retires in issue order, so a window fetched in one piece ahead of the weight stream makes the first weight wait of the item wait for all of HBM's latency
// (ablations, profiles/r04/box_dgrad_ablations.log: window traffic 0.28 ms of deconv1's 1.29 ms although every load was "in flight" for a whole item)
__device__ __forceinline__ void bx_win_load_rows(f4 (&buf)[kWinRegs], const BxLane& L, const unsigned short* __restrict__ src, int NHR, int nj, int ni,
                                                 int R0, int R1)            // R0, R1: compile-time after the callers' loops are unrolled
{
    const bool on = L.act && L.cc < ni;
#pragma unroll
    for (int r = R0; r < R1; ++r) {
        if (r < kB2WR) {
            buf[r] = (f4){0.f, 0.f, 0.f, 0.f};
            if (on && r < nj) buf[r] = *reinterpret_cast<const f4*>(src + (long long)r * NHR * kBxCo + L.voff_plane);
        }
    }
}

__device__ __forceinline__ void bx_win_store(unsigned char* wnd, const f4 (&buf)[kWinRegs], const BxLane& L, int nj, int ni)
{
    const bool on = L.act && L.cc < ni;
#pragma unroll
    for (int r = 0; r < kB2WR; ++r)
        if (on && r < nj) *reinterpret_cast<f4*>(wnd + (((r >> 1) & 1) ? L.lds1 : L.lds0) + r * (kB2WC * 16)) = buf[r];
}

// ---------------------------------------------------------------------------------------------------
// K2: data gradient  g_x[nb][iy][ix][ci] = sum_{tap, co} B[nb][vmap[iy][ky]][hmap[ix][kx]][co] * W[co][ci][tap]   (six bf16 cross terms, fp32-product accuracy)
// ---------------------------------------------------------------------------------------------------
// A workgroup (4 wavefronts; two workgroups per CU) owns one tile and 32 NT input channels; a wavefront owns a block of 4 rows x 8 columns of it (= the M
// dimension of v_mfma_f32_32x32x16_bf16: <= 1 / 8 of a block wasted on any width).  A k-step is 2 taps x 8 channels; the lane's A fragment of a plane is ONE
// 16-byte LDS read at a per-lane address computed once per tile from the two maps.  The weight — split once into three bf16 terms in fragment order, the
// sign of odd chunks flipped (upconv_box_dgrad_prep_kernel) — streams L2 -> registers -> LDS double-buffered in STAGES of KPS k-steps: 24 MFMAs between two
// barriers (NT = 2: two k-steps per stage; NT = 4: one), the next stage's loads in flight for a whole stage.  Six cross terms hh, hm, mh, hl, lh, mm in
// ss_gemm6_f32's order; the running sum's sign alternates per chunk (the bf16 MFMA's fp32 accumulation drifts down by ~2^-28 of the magnitude sum).
// weight [C_out][C_in][5][5] fp32 -> Wf[ci block of 32 NT][chunk][k-step 13][plane 3][tile NT][lane 64][8] bf16: element e of a lane = split term of
// (+ / -) W[co = 8 chunk + e][ci = 32 NT blk + 32 tile + (lane & 31)][tap = 2 s + (lane >> 5)]  (tap 25: zero; odd chunks negated)
template <int DT = 0, int NG = 1>           // NG > 1 (16-bit modes): "plane" sp of chunk GROUP c is chunk NG c + sp; the sign alternates per group
__global__ __launch_bounds__(kBlock) void upconv_box_dgrad_prep_kernel(const float* __restrict__ W, unsigned short* __restrict__ Wf, int Cin, int Cout, int NT)
{
    static_assert(DT != 0 || NG == 1, "chunk groups: 16-bit modes only");
    constexpr int NW = DT == 0 ? BxT<DT>::NW : NG;
    const int NCH = Cout / (kBxCo * NG);
    const long long total = (long long)(Cin / (32 * NT)) * NCH * kB2KS * NW * NT * 64;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int lane = (int)(i & 63);
        long long r = i >> 6;
        const int t = (int)(r % NT); r /= NT;
        const int sp = (int)(r % NW); r /= NW;
        const int s = (int)(r % kB2KS); r /= kB2KS;
        const int c = (int)(r % NCH); const int blk = (int)(r / NCH);
        const int tap = 2 * s + (lane >> 5);
        const int ci = 32 * NT * blk + 32 * t + (lane & 31);
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int co = DT == 0 ? kBxCo * c + e : kBxCo * (NG * c + sp) + e;
            float v = tap < 25 ? W[((long long)co * Cin + ci) * 25 + tap] : 0.f;
            if (c & 1) v = -v;
            if constexpr (DT != 0) { o[e] = round_op<DT>(v); continue; }
            const unsigned short h1 = narrow<SS_DT_BF16>(v);
            const float r1 = v - widen<SS_DT_BF16>(h1);
            const unsigned short h2 = narrow<SS_DT_BF16>(r1);
            const float r2 = r1 - widen<SS_DT_BF16>(h2);
            o[e] = sp == 0 ? h1 : (sp == 1 ? h2 : narrow<SS_DT_BF16>(r2));
        }
        *reinterpret_cast<u16x8*>(Wf + i * 8) = o;
    }
}

template <int NT, int KPS, int DT = 0, int NG = 1>      // NCH: chunk GROUPS (C_out / (8 NG))
__global__ __launch_bounds__(kB2Threads, 2) void upconv_box_dgrad_kernel(const unsigned short* __restrict__ Bp, const unsigned short* __restrict__ Wf,
                                                                         const int* __restrict__ vmap, const int* __restrict__ hmap,
                                                                         const int* __restrict__ tr, const int* __restrict__ tc,
                                                                         typename ActT<DT>::type* __restrict__ gx, int NB, int h, int w, int NVR, int NHR, int CIN, int NCH, int RG, int CG)
{
    static_assert(DT != 0 || NG == 1, "chunk groups: 16-bit modes only");
    constexpr int NP = DT == 0 ? BxT<DT>::NP : NG, NW = DT == 0 ? BxT<DT>::NW : NG;     // window planes / weight planes of a k-step (16-bit modes: the NG chunks of the group)
    constexpr int kKB = NW * NT * 1024;                                         // bytes of one k-step of weights
    constexpr int kStage = KPS * kKB;                                           // 12 KB
    constexpr int kNS = (kB2KS + KPS - 1) / KPS;                                // stages per chunk (7 | 13)
    constexpr int kF4 = kStage / (16 * kB2Threads);                             // 16-byte pieces per thread and stage (3)
    static_assert(kStage % (16 * kB2Threads) == 0, "stage size");
    __shared__ __attribute__((aligned(16))) unsigned char wnd[NP * kB2Plane];
    __shared__ __attribute__((aligned(16))) unsigned char bst[2 * kStage];
    __shared__ int tiles[4 * 64 + 2 * 16];                                       // the row / column tile tables (<= 64 row tiles, <= 16 column tiles: checked by the host)
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NBLK = CIN / (32 * NT);
    const long long n_tiles = (long long)NB * RG * CG * NBLK;
    bx_zero_borders(wnd, NP);
    for (int i = threadIdx.x; i < 4 * RG; i += kB2Threads) tiles[i] = tr[i];
    for (int i = threadIdx.x; i < 2 * CG; i += kB2Threads) tiles[256 + i] = tc[i];
    __syncthreads();
    const unsigned g = xcd_remap(blockIdx.x, gridDim.x);
    const long long t_begin = n_tiles * g / gridDim.x, t_end = n_tiles * (g + 1) / gridDim.x;
    const long long plane_g = (long long)NVR * NHR * kBxCo;                    // elements of one plane of one chunk in HBM
    // work items = (tile, chunk) pairs in tile-major order; item -> (frame, row tile, column tile, ci block, chunk) and its window in HBM
    struct Item { int nb, rg, cg, blk, c, j0, nj, i0, ni; };
    auto extents = [&](Item& d) {
        d.j0 = tiles[4 * d.rg + 2]; d.nj = tiles[4 * d.rg + 3];
        d.i0 = tiles[256 + 2 * d.cg]; d.ni = tiles[256 + 2 * d.cg + 1];
    };
    auto decode = [&](long long item) {                                         // (64-bit divisions: once per workgroup; the loop advances incrementally)
        Item d;
        d.c = (int)(item % NCH);
        long long tl = item / NCH;
        d.blk = (int)(tl % NBLK); tl /= NBLK;
        d.cg = (int)(tl % CG); tl /= CG;
        d.rg = (int)(tl % RG);
        d.nb = (int)(tl / RG);
        extents(d);
        return d;
    };
    auto advance = [&](Item& d) {
        if (++d.c == NCH) { d.c = 0; if (++d.blk == NBLK) { d.blk = 0; if (++d.cg == CG) { d.cg = 0; if (++d.rg == RG) { d.rg = 0; ++d.nb; } } } }
        extents(d);
    };
    const BxLane bl = bx_lane(plane_g, kB2Plane, NP);
    auto win_src = [&](const Item& d) { return Bp + (((long long)d.nb * NCH + d.c) * NP) * plane_g + ((long long)d.j0 * NHR + d.i0) * kBxCo; };
    const long long it_begin = t_begin * NCH, it_end = t_end * NCH;
    if (it_begin >= it_end) return;
    f4 wbuf[kWinRegs];
    Item cur = decode(it_begin);
    bx_win_load(wbuf, bl, win_src(cur), NHR, cur.nj, cur.ni);                   // the first window: the only one whose latency is exposed
    bx_win_store(wnd, wbuf, bl, cur.nj, cur.ni);
    int addr[kB2KS];
    // two accumulator sets (alternating k-steps of a stage) when a stage holds more than one k-step: with NT = 2 that makes 4 independent MFMA chains per wavefront
    // instead of 2; added once per tile.  16-bit modes (one term: a k-step is NT MFMAs): stages of 6 | 3 k-steps — the fp32 mode's stage BYTES, a third of its barriers
    constexpr int NACC = (KPS > 1 && NT <= 2) ? 2 : 1;                          // (k-step ks of a stage accumulates on set ks % NACC; NT = 4 has four chains already and no registers for eight)
    f32x16 acc[NACC][NT];
    static_assert(kNS & 1, "the stage double buffer's parity flips per item");
    constexpr int kWR = (kB2WR + kNS - 3) / (kNS - 2);                          // window rows fetched per stage (3 | 2): done two stages before the item ends
    auto wsrc = [&](const Item& d) { return reinterpret_cast<const unsigned char*>(Wf) + ((long long)d.blk * NCH + d.c) * kB2KS * kKB; };
    f4 st[kF4];
    {   // weight stage 0 of the first item (every later one arrives during its predecessor's last stage)
        const unsigned char* const b0 = wsrc(cur);
#pragma unroll
        for (int u = 0; u < kF4; ++u) st[u] = *reinterpret_cast<const f4*>(b0 + (threadIdx.x + kB2Threads * u) * 16);
#pragma unroll
        for (int u = 0; u < kF4; ++u) *reinterpret_cast<f4*>(bst + (threadIdx.x + kB2Threads * u) * 16) = st[u];
    }
    int par = 0;                                                                // half of the stage buffer that holds stage 0 of the current item
    [[maybe_unused]] int titem = 0;
#pragma unroll 1
    for (long long item = it_begin; item < it_end; ++item) {
        BX_STAMP(0);
        const bool has_next = item + 1 < it_end;
        Item nxt = cur;
        if (has_next) advance(nxt);
        const unsigned short* const wn = win_src(nxt);
        const int sy0 = tiles[4 * cur.rg], nrow = tiles[4 * cur.rg + 1], sx0 = kB2TC * cur.cg;
        if (cur.c == 0) {
            // ---- a new tile: this lane's pixel and its 13 A-fragment addresses (bytes inside a plane of the window)
            const int m = lane & 31;
            const int sy = min(sy0 + min(m >> 3, nrow - 1), h - 1), sx = min(sx0 + 8 * wv + (m & 7), w - 1);
            int jvq[5], ivq[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) { jvq[q] = vmap[sy * 5 + q]; ivq[q] = hmap[sx * 5 + q]; }
#pragma unroll
            for (int s2 = 0; s2 < kB2KS; ++s2) {
                const int t0 = 2 * s2, t1 = 2 * s2 + 1;                         // tap = 2 s + (lane >> 5): both candidates are compile-time, the lane half selects
                const int a0 = bx_addr(jvq[t0 / 5], ivq[t0 % 5], cur.j0, cur.i0);
                const int a1 = t1 < 25 ? bx_addr(jvq[t1 / 5], ivq[t1 % 5], cur.j0, cur.i0) : kB2Zero;
                addr[s2] = (lane >> 5) ? a1 : a0;
            }
#pragma unroll
            for (int a2 = 0; a2 < NACC; ++a2)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a2][t][r] = 0.f;
        } else {                                                                // sign of the running sum alternates per chunk (odd chunks' weights are negated)
#pragma unroll
            for (int a2 = 0; a2 < NACC; ++a2)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a2][t][r] = -acc[a2][t][r];
        }
        const unsigned char* const bsrc = wsrc(cur);
        const unsigned char* const bnxt = wsrc(nxt);
        BX_STAMP(1);
        lds_barrier();                                                          // stage 0 and this item's window (stored at the end of the previous item) are visible
        BX_STAMP(2);
                                                                                // (LDS-only barriers in the item loop: __syncthreads() would drain the prefetches, ss_common.hpp)
#pragma unroll
        for (int sg = 0; sg < kNS; ++sg) {
            const bool more = sg + 1 < kNS;
            // the last stage of a chunk may hold fewer k-steps (13 = 6 x 2 + 1): only what exists is fetched
            constexpr int kLastBytes = (kB2KS - (kNS - 1) * KPS) * kKB;
            if (!(SS_BX_ABLATE & 2)) {
                if (more) {
#pragma unroll
                    for (int u = 0; u < kF4; ++u) {
                        const int off = (threadIdx.x + kB2Threads * u) * 16;
                        if (sg + 2 < kNS || off < kLastBytes) st[u] = *reinterpret_cast<const f4*>(bsrc + (long long)(sg + 1) * kStage + off);
                    }
                } else if (has_next) {                                          // stage 0 of the NEXT item, into the half this item's last stage does not read
#pragma unroll
                    for (int u = 0; u < kF4; ++u) st[u] = *reinterpret_cast<const f4*>(bnxt + (threadIdx.x + kB2Threads * u) * 16);
                }
            }
            if (has_next && !(SS_BX_ABLATE & 1)) {                              // the next window, kWR rows per stage, issued behind the weight loads
                bx_win_load_rows(wbuf, bl, wn, NHR, nxt.nj, nxt.ni, sg * kWR, (sg + 1) * kWR);
            }
            // hipcc sinks these loads down to their first use — the LDS stores at the END of the stage — and then waits for them at once: the whole L2
            // latency exposed per stage.  The scheduling barrier keeps them up here.
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KPS; ++ks) {
                const int s2 = sg * KPS + ks;
                if (s2 < kB2KS) {
                    const unsigned char* const ap = wnd + addr[s2];
                    const unsigned char* const bk = bst + ((sg & 1) ^ par) * kStage + ks * kKB + lane * 16;
                    s16x8 fa[NP], fb[NW][NT];
#pragma unroll
                    for (int p2 = 0; p2 < NP; ++p2) fa[p2] = *reinterpret_cast<const s16x8*>(ap + p2 * kB2Plane);
#pragma unroll
                    for (int p2 = 0; p2 < NW; ++p2)
#pragma unroll
                        for (int u = 0; u < NT; ++u) fb[p2][u] = *reinterpret_cast<const s16x8*>(bk + (p2 * NT + u) * 1024);
                    // fp32 mode: the six cross terms, smallest first: (h, l) (m, m) (l, h) (h, m) (m, h) (h, h); 16-bit modes: the planes of B against the ONE weight term, lo first
                    // (round 6: the planes are the group's chunks, each against ITS weight plane)
                    constexpr int kNQ = DT ? NP : 6;
                    constexpr int pa[6] = {0, 1, 2, 0, 1, 0}, pb[6] = {DT ? 0 : 2, 1, DT ? 2 : 0, DT ? 0 : 1, DT ? 1 : 0, 0};
#pragma unroll
                    for (int q = 0; q < kNQ; ++q)
#pragma unroll
                        for (int u = 0; u < NT; ++u) {
                            if (SS_BX_ABLATE & 4) acc[ks % NACC][u][q] += (float)(fa[pa[q]][0] + fb[pb[q]][u][1]);
                            else acc[ks % NACC][u] = mfma32<DT>(fb[pb[q]][u], fa[pa[q]], acc[ks % NACC][u]);      // D^T: rows = input channels, columns = pixels
                        }
                }
            }
#if SS_BX_TRACE
            if (sg == 0 || sg == kNS - 1) { if (acc[0][0][0] == 12345.f && acc[NACC - 1][NT - 1][3] == 3.f) bx_trace[63][15] = 1; BX_STAMP(sg == 0 ? 3 : 5); }
#endif
            if (more || has_next) {
                unsigned char* const dst = bst + (((sg + 1) & 1) ^ par) * kStage;
#pragma unroll
                for (int u = 0; u < kF4; ++u) {
                    const int off = (threadIdx.x + kB2Threads * u) * 16;
                    if (!(SS_BX_ABLATE & 2) && (!more || sg + 2 < kNS || off < kLastBytes)) *reinterpret_cast<f4*>(dst + off) = st[u];
                }
                if (more && !(SS_BX_ABLATE & 8)) lds_barrier();
            }
#if SS_BX_TRACE
            if (sg == 0) BX_STAMP(4);
#endif
        }
        par ^= 1;
        BX_STAMP(6);
        lds_barrier();                                                          // every reader of this item's window and stages is done
        BX_STAMP(7);
        if (has_next && !(SS_BX_ABLATE & 1)) bx_win_store(wnd, wbuf, bl, nxt.nj, nxt.ni);
        BX_STAMP(8);
        if (cur.c == NCH - 1) {
            // ---- tile epilogue.  The product is taken TRANSPOSED (weights as the A operand): D[ci = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][pixel = lane & 31] —
            //      a lane holds 4 consecutive input channels of ONE pixel per register quad: 4 NT 16-byte stores instead of 16 NT 4-byte ones (what the
            //      4-byte form cost the sub-pixel forward: profiles/r04/sub_trace_v2.log).  The sum carries the sign of the last chunk.
            const float fin = ((NCH - 1) & 1) ? -1.f : 1.f;
            const int pm = lane & 31;
            const int py = sy0 + (pm >> 3), px = sx0 + 8 * wv + (pm & 7);
            if ((pm >> 3) < nrow && px < w) {
                typename ActT<DT>::type* const op = gx + (((long long)cur.nb * h + py) * w + px) * CIN + 32 * NT * cur.blk + 4 * (lane >> 5);
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x = acc[0][t][4 * q + e];
                            if constexpr (NACC > 1) x += acc[1][t][4 * q + e];
                            v[e] = x * fin;
                        }
                        store_act4<DT>(op + 32 * t + 8 * q, v[0], v[1], v[2], v[3]);
                    }
            }
        }
#if SS_BX_TRACE
        if (blockIdx.x == 0 && threadIdx.x == 0 && titem < 64) bx_trace[titem][9] = clock64(), bx_trace[titem][10] = (unsigned long long)cur.c;
        ++titem;
#endif
        cur = nxt;
    }
}

// The weight-gradient kernel keeps its window as a RING of 16 rows (row slot = vertical range id & 15) and walks the tiles of a column strip top to bottom:
// consecutive row tiles share 7 of their <= 15 vertical ranges, so only the new rows are fetched and stored (the kernel is bound by staging 54 KB of window
// per 96 - 192 MFMAs of a wavefront; deconv2 0.87 -> 0.81 ms, deconv1 unchanged by this alone: profiles/r04/bench_box_bwd_v11.log .. v13.log).
constexpr int kB3WR = 16;
constexpr int kB3Zero = kB3WR * kB2WC * 16;
constexpr int kB3Plane = kB3Zero + 16;
__device__ __forceinline__ void bx3_store_rows(unsigned char* wnd, const f4 (&buf)[kWinRegs], const BxLane& L, int jf, int cnt, int ni)
{
    const bool on = L.act && L.cc < ni;
#pragma unroll
    for (int r = 0; r < kB2WR; ++r) {
        const int sl = (jf + r) & (kB3WR - 1);
        if (on && r < cnt) *reinterpret_cast<f4*>(wnd + (((sl >> 1) & 1) ? L.lds1 : L.lds0) + sl * (kB2WC * 16)) = buf[r];
    }
}

// ---------------------------------------------------------------------------------------------------
// K3: weight gradient  g_W[co][ci][tap] = sum_{nb, iy, ix} x[nb][iy][ix][ci] * B[nb][vmap[iy][ky]][hmap[ix][kx]][co]   (x spikes: EXACT products, three bf16 terms of B)
// ---------------------------------------------------------------------------------------------------
// v_mfma_f32_32x32x16_bf16 with M = 4 taps x 8 channels of one chunk ("tap quad": 7 quads cover the 25 taps, 3 phantom rows), N = 32 input channels,
// K = 16 consecutive source pixels of a row.  A workgroup (4 wavefronts; two per CU) is one KIND — an 8-channel chunk of C_out and a block of 32 NT input
// channels — whose accumulators (NQ tiles of 32 x 32 per wavefront) stay in registers while it walks its slice of the tiles: the window of the chunk's three
// B planes is staged exactly as in the data-gradient kernel, the spike operand comes pre-transposed from upconv_bwd_xprep_kernel (one 16-byte load per lane
// and k-step, all eight of a tile issued before the window is staged).  The A fragment — 8 consecutive PIXELS of one (tap, channel) row, i.e. the window
// read against its grain — is two ds_read_b64_tr_b16 per plane: the LDS transpose read hands lane i of a 16-lane group column i of the 4 x 16 block whose rows
// the group's lanes address INDIVIDUALLY (measured: out[i][r] = in[lane 4 r + i / 4][element i % 4], profiles/r04/tr16.log), so each source lane points at
// "its" pixel through the two index maps and the gather along k costs nothing.  The main loop is branch-free: rows / columns beyond the tile meet a zeroed
// spike fragment, phantom taps and quads read the window's zero row.  Partials -> ws[slice][co][tap][ci] -> upconv_box_wgrad_reduce_kernel (fixed order).
template <int NT, bool PF, int DT = 0, int NG = 1>      // NT: input-channel tiles per kind = per WAVEFRONT (1 | 2 | 4): a wavefront owns the tap quads wv and wv + 4 for all of
                                               // them, so every transposed A fragment feeds NT MFMAs.  PF: the next tile's window is fetched into registers while this
                                               // tile multiplies (14 x 16 B per thread: NT <= 2 only — NT = 4 holds 128 accumulator registers)
                                               // NG (16-bit modes, BxT comment): the window's planes are NG consecutive chunks, one accumulator set each — a transposed
                                               // A fragment still feeds NT MFMAs, a spike fragment NG x 2 of them; NT x NG <= 4 (128 accumulator registers)
__global__ __launch_bounds__(kB2Threads, 2) void upconv_box_wgrad_kernel(const unsigned short* __restrict__ Bp, const unsigned short* __restrict__ xT,
                                                                         const int* __restrict__ vmap, const int* __restrict__ hmap,
                                                                         const int* __restrict__ tr, const int* __restrict__ tc,
                                                                         float* __restrict__ ws, int NB, int h, int w, int NVR, int NHR, int CIN, int COUT, int KINDS,
                                                                         int RG, int CG)
{
    static_assert(DT != 0 || NG == 1, "chunk groups: 16-bit modes only");
    constexpr int NQ = 2;                                                       // tap quads per wavefront: wv, wv + 4 (the 8th is a phantom: zero row)
    constexpr int NP = DT == 0 ? BxT<DT>::NP : NG;                              // window planes: the three terms of a chunk | the NG chunks of a group
    constexpr int NA = DT == 0 ? 1 : NG;                                        // accumulator sets
    __shared__ __attribute__((aligned(16))) unsigned char wnd[NP * kB3Plane];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NCH = COUT / (kBxCo * NG), CIB = CIN / (32 * NT);                 // NCH: chunk groups
    const int kind = (int)(blockIdx.x % KINDS), slice = (int)(blockIdx.x / KINDS), slices = (int)(gridDim.x / KINDS);
    const int c = kind / CIB, cib = kind - c * CIB;                              // chunk (group) of C_out, block of input channels
    const int KSR = (w + 15) / 16;
    const long long n_tiles = (long long)NB * RG * CG;
    const long long t_begin = n_tiles * slice / slices, t_end = n_tiles * (slice + 1) / slices;
    const long long plane_g = (long long)NVR * NHR * kBxCo;
    if ((int)threadIdx.x < NP) *reinterpret_cast<f4*>(wnd + threadIdx.x * kB3Plane + kB3Zero) = (f4){0.f, 0.f, 0.f, 0.f};        // the zero pixel of each plane
    // this lane as a SOURCE lane of the transpose reads: pixel L >> 2 of a 4-pixel sub-block, columns 4 (L & 3) .. + 3 of the 16-row half g of the M tile
    const int L = lane & 15, g = (lane >> 4) & 1, oct = lane >> 5;
    const int tq4 = 2 * g + ((L & 3) >> 1), coq = (L & 3) & 1;
    int kyq[NQ], kxq[NQ];
    bool real[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int t = 4 * (wv + 4 * u) + tq4;
        real[u] = t < 25;
        kyq[u] = real[u] ? t / 5 : 0;
        kxq[u] = real[u] ? t - 5 * (t / 5) : 0;
    }
    f32x16 acc[NA][NQ][NT];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int u = 0; u < NQ; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][u][t][r] = 0.f;
    const unsigned xoffT = (unsigned)(lane & 31) * 16u + (unsigned)(lane >> 5) * 8u;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    const unsigned short* const xbase = xT + (long long)(cib * NT) * (32 * 16) + xoffT;
    // the spike fragments of k-step sidx = (row sidx >> 1, half sidx & 1) of the tile (NT ci tiles); rows / halves outside the map are zeroed
    auto load_x = [&](s16x8 (&xf)[NT], int nb, int sy0, int nrow, int sx0, int sidx) {
        const int r = sidx >> 1, half = sidx & 1;
        const bool ok = r < nrow && sx0 + 16 * half < w;
        const long long srow = (long long)nb * h + min(sy0 + r, h - 1);
        const int ks = min((sx0 >> 4) + half, KSR - 1);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            xf[t] = *reinterpret_cast<const s16x8*>(xbase + (srow * KSR + ks) * ((long long)CIN * 16) + t * (32 * 16));
            if (!ok) xf[t] = (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    __shared__ int tiles[4 * 64 + 2 * 16];                                       // the row / column tile tables (<= 64 row tiles, <= 16 column tiles: checked by the host)
    for (int i = threadIdx.x; i < 4 * RG; i += kB2Threads) tiles[i] = tr[i];
    for (int i = threadIdx.x; i < 2 * CG; i += kB2Threads) tiles[256 + i] = tc[i];
    // the two index maps in LDS (when they fit): every tile starts by looking 16 ids per lane up, and from global memory those dependent loads sat between the
    // tile's first barrier and its first MFMA
    constexpr int kMapH = 160, kMapW = 192;
    __shared__ int vm_s[5 * kMapH], hm_s[5 * kMapW];
    const bool maps_in_lds = h <= kMapH && w <= kMapW;
    if (maps_in_lds) {
        for (int i = threadIdx.x; i < 5 * h; i += kB2Threads) vm_s[i] = vmap[i];
        for (int i = threadIdx.x; i < 5 * w; i += kB2Threads) hm_s[i] = hmap[i];
    }
    __syncthreads();
    const BxLane bl = bx_lane(plane_g, kB3Plane, NP);
    // rows from vertical range id jf on, columns from the column tile's first horizontal id
    auto win_src = [&](int nb, int jf, int cg) { return Bp + (((long long)nb * NCH + c) * NP) * plane_g + ((long long)jf * NHR + tiles[256 + 2 * cg]) * kBxCo; };
    // tile -> (frame, column tile, row tile), ROW tile fastest (a column strip top to bottom): one 64-bit division per workgroup, then incremental
    int rg = (int)(t_begin % RG), cg = (int)((t_begin / RG) % CG), nb = (int)(t_begin / ((long long)CG * RG));
    f4 wbuf[kWinRegs];                                                          // (a slice without tiles still writes its — zero — partials below)
    [[maybe_unused]] int pj1 = 0;
    if constexpr (PF) {
        if (t_begin < t_end) {
            bx_win_load(wbuf, bl, win_src(nb, tiles[4 * rg + 2], cg), NHR, tiles[4 * rg + 3], tiles[256 + 2 * cg + 1]);     // the first window: the only latency exposed
            bx3_store_rows(wnd, wbuf, bl, tiles[4 * rg + 2], tiles[4 * rg + 3], tiles[256 + 2 * cg + 1]);
        }
    }
#pragma unroll 1
    for (long long tl = t_begin; tl < t_end; ++tl) {
        const int sy0 = tiles[4 * rg], nrow = tiles[4 * rg + 1], j0 = tiles[4 * rg + 2], nj = tiles[4 * rg + 3];
        const int sx0 = kB2TC * cg, i0 = tiles[256 + 2 * cg], ni = tiles[256 + 2 * cg + 1];
        int rgn = rg + 1, cgn = cg, nbn = nb;                                   // the next tile: one row tile down, or the top of the next strip
        if (rgn == RG) { rgn = 0; if (++cgn == CG) { cgn = 0; ++nbn; } }
        const int j0n = tiles[4 * rgn + 2], nin = tiles[256 + 2 * cgn + 1];
        // rows of the next window the ring does not hold yet: all of them at the top of a strip, else those below this tile's last row
        const int jfn = rgn ? max(j0n, j0 + nj) : j0n;
        const int njn = j0n + tiles[4 * rgn + 3] - jfn;                          // rows to fetch (<= 15)
        // PF: the next tile's window is fetched two rows per k-step, each pair issued BEHIND that k-step's spike loads (in-order vector-memory counter: see
        // bx_win_load_rows)
        const bool pf_next = PF && tl + 1 < t_end;
        const unsigned short* const wn = win_src(nbn, jfn, cgn);
        // the spike fragments come straight from L2, TWO k-steps ahead of their MFMAs (three rotating buffers): one k-step = 12 NT MFMAs = 400 - 800 cycles
        // is shorter than the L2 latency, at a distance of one every k-step waited for its load
        // (NT = 4: the third buffer spills, distance 1 measured faster — 0.81 vs 1.00 ms on deconv2, profiles/r04/bench_box_bwd_v12.log)
        constexpr int PD = NT <= 2 ? 2 : 1, XB = PD + 1;
        s16x8 xq[XB][NT];
        load_x(xq[0], nb, sy0, nrow, sx0, 0);                                   // in flight while the window is staged
        if constexpr (PD > 1) load_x(xq[1], nb, sy0, nrow, sx0, 1);
        if (pf_next) bx_win_load_rows(wbuf, bl, wn, NHR, njn, nin, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        // ---- this lane's read addresses in two halves: rowp[row][quad] = byte offset of the window row | swizzle bit in bit 4, colp[half * 2 + rd][quad] = byte
        //      offset of "its" pixel's column (+ 8 for the odd channel quad); negative = the empty range (-> the zero pixel)
        int rowp[kB2TR][NQ], colp[4][NQ];
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
#pragma unroll
            for (int r = 0; r < kB2TR; ++r) {
                const int vi = min(sy0 + min(r, nrow - 1), h - 1) * 5 + kyq[u];
                const int jv = real[u] ? (maps_in_lds ? vm_s[vi] : vmap[vi]) : 0;
                const int rw = jv & (kB3WR - 1);                                // ring slot
                rowp[r][u] = jv ? (rw * (kB2WC * 16) | (((rw >> 1) & 1) << 4)) : -1;
            }
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                const int hi2 = min(sx0 + 16 * (jx >> 1) + 8 * oct + 4 * (jx & 1) + (L >> 2), w - 1) * 5 + kxq[u];
                const int iv = real[u] ? (maps_in_lds ? hm_s[hi2] : hmap[hi2]) : 0;
                colp[jx][u] = iv ? (iv - i0) * 16 : -1;
            }
        }
        auto addr_of = [&](int r, int jx, int u) {
            const int rp = rowp[r][u], cp = colp[jx][u];
            const int a = (rp & ~16) + (cp ^ (rp & 16));
            return ((rp | cp) < 0 ? kB3Zero : a) + 8 * coq;
        };
        if constexpr (!PF) {
            __syncthreads();                                                    // the previous tile's readers of the window are done
            const int jf = (rg && tl > t_begin) ? max(j0, pj1) : j0;             // (pj1: one past the previous tile's last row — same strip when rg > 0)
            bx_win_load(wbuf, bl, win_src(nb, jf, cg), NHR, j0 + nj - jf, ni);
            bx3_store_rows(wnd, wbuf, bl, jf, j0 + nj - jf, ni);
        }
        if constexpr (PF) lds_barrier(); else __syncthreads();                  // this tile's window (PF: stored at the end of the previous tile) is visible
        // k-step s = (row s >> 1, half s & 1): its A fragments (2 quads x 3 planes x 2 transpose reads) are fetched one k-step AHEAD of its MFMAs
        auto read_a = [&](int sidx, s16x4 (&lo)[NQ][NP], s16x4 (&hi)[NQ][NP]) {
            const int r = sidx >> 1, half = sidx & 1;
            int a0[NQ], a1[NQ];
#pragma unroll
            for (int u = 0; u < NQ; ++u) { a0[u] = addr_of(r, 2 * half, u); a1[u] = addr_of(r, 2 * half + 1, u); }
#pragma unroll
            for (int u = 0; u < NQ; ++u)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    lo[u][p] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(wnd + p * kB3Plane + a0[u]));
                    hi[u][p] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(wnd + p * kB3Plane + a1[u]));
                }
        };
        auto mma = [&](s16x4 (&lo)[NQ][NP], s16x4 (&hi)[NQ][NP], s16x8 (&xf)[NT]) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int u = 0; u < NQ; ++u) {
                    const s16x8 af = {lo[u][p][0], lo[u][p][1], lo[u][p][2], lo[u][p][3], hi[u][p][0], hi[u][p][1], hi[u][p][2], hi[u][p][3]};
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[DT ? p : 0][u][t] = mfma32<DT>(af, xf[t], acc[DT ? p : 0][u][t]);
                }
        };
        s16x4 la[NQ][NP], ha[NQ][NP], lb[NQ][NP], hb[NQ][NP];
        read_a(0, la, ha);
        if constexpr (PD > 1) {
#pragma unroll
            for (int sidx = 0; sidx < 2 * kB2TR; ++sidx) {                      // (the buffers rotate statically)
                if (sidx + 1 < 2 * kB2TR) { if (sidx & 1) read_a(sidx + 1, la, ha); else read_a(sidx + 1, lb, hb); }
                if (sidx + PD < 2 * kB2TR) load_x(xq[(sidx + PD) % XB], nb, sy0, nrow, sx0, sidx + PD);
                if (pf_next) bx_win_load_rows(wbuf, bl, wn, NHR, njn, nin, 2 * (sidx + 1), 2 * (sidx + 1) + 2);
                __builtin_amdgcn_sched_barrier(0);
                if (sidx & 1) mma(lb, hb, xq[sidx % XB]); else mma(la, ha, xq[sidx % XB]);
            }
        } else {
#pragma unroll
            for (int sidx = 0; sidx < 2 * kB2TR; sidx += 2) {                   // two k-steps per trip: the two buffers alternate statically
                read_a(sidx + 1, lb, hb);
                load_x(xq[1], nb, sy0, nrow, sx0, sidx + 1);
                if (pf_next) bx_win_load_rows(wbuf, bl, wn, NHR, njn, nin, 2 * (sidx + 1), 2 * (sidx + 1) + 2);
                __builtin_amdgcn_sched_barrier(0);
                mma(la, ha, xq[0]);
                if (sidx + 2 < 2 * kB2TR) {
                    read_a(sidx + 2, la, ha);
                    load_x(xq[0], nb, sy0, nrow, sx0, sidx + 2);
                    if (pf_next) bx_win_load_rows(wbuf, bl, wn, NHR, njn, nin, 2 * (sidx + 2), 2 * (sidx + 2) + 2);
                }
                __builtin_amdgcn_sched_barrier(0);
                mma(lb, hb, xq[1]);
            }
        }
        if constexpr (PF) {
            lds_barrier();                                                      // every reader of this tile's window is done
            if (tl + 1 < t_end) bx3_store_rows(wnd, wbuf, bl, jfn, njn, nin);
        }
        pj1 = j0 + nj;
        cg = cgn; rg = rgn; nb = nbn;
    }
    // ---- partials: D[m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][ci = lane & 31], m = 8 (tap - 4 quad) + channel  ->  ws[slice][co][tap][ci]
    float* const wsl = ws + (long long)slice * COUT * 25 * CIN;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int u = 0; u < NQ; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mm = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int tap = 4 * (wv + 4 * u) + (mm >> 3), co = kBxCo * (NA * c + a) + (mm & 7);
                    if (tap < 25) wsl[((long long)co * 25 + tap) * CIN + 32 * (cib * NT + t) + (lane & 31)] = acc[a][u][t][r];
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

#if SS_BX_TRACE
int ss_debug_box_trace(unsigned long long* host_dst)
{
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(bx_trace), sizeof(unsigned long long) * 64 * 16) == hipSuccess ? SS_OK : SS_ELAUNCH;
}
#endif

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

int ss_upconv_box_window(int* max_rows, int* max_cols)
{
    if (max_rows) *max_rows = kB2WR;                                            // distinct vertical range ids (span) a tile of <= 4 source rows may reach
    if (max_cols) *max_cols = kB2WC;                                        // ... and horizontal ones 32 source columns may reach
    return kB2TR;
}

int ss_upconv_box_dgrad_supported(int Cin, int Cout, int k, int max_tile_rows, int max_cols32)
{
    // max_tile_rows / max_cols32: largest id span of the caller's row tiles (fused.box_tables cuts them so that they fit) / of 32 consecutive source columns
    if (k != 5 || Cin < 64 || Cin % 64 != 0 || Cout < kBxCo || Cout % kBxCo != 0) return 0;
    return max_tile_rows > 0 && max_tile_rows <= kB2WR && max_cols32 > 0 && max_cols32 <= kB2WC;
}

int ss_upconv_box_tiles_supported(int n_row_tiles, int w, long long pixels)
{
    return n_row_tiles > 0 && n_row_tiles <= 64 && w > 0 && (w + kB2TC - 1) / kB2TC <= 16 && pixels > 0 && pixels <= 0x7fffffffLL;
}

long long ss_upconv_box_dgrad_ws_floats(int Cin, int Cout)
{
    if (Cin <= 0 || Cout <= 0 || Cin % 64 != 0 || Cout % kBxCo != 0) return 0;
    return (long long)(Cin / 32) * (Cout / kBxCo) * kB2KS * 3 * 1024 / 4;       // the weight as three bf16 terms in fragment order
}

int ss_upconv_box_dgrad_f32(const void* box, const float* weight, const int* vmap, const int* hmap, const int* tile_rows, int n_row_tiles, const int* tile_cols,
                            float* g_x, float* ws, long long NB, int Cin, int Cout, int h, int w, int NVR, int NHR, void* stream)
{
    if (!box || !weight || !vmap || !hmap || !tile_rows || !tile_cols || !g_x || !ws || NB <= 0 || h <= 0 || w <= 0 || NVR <= 0 || NHR <= 0 || n_row_tiles <= 0) return SS_EINVAL;
    if (!ss_upconv_box_dgrad_supported(Cin, Cout, 5, 1, 1)) return SS_EINVAL;                    // shape only: the caller checked the extents
    if (!ss_upconv_box_tiles_supported(n_row_tiles, w, NB * h * (long long)w)) return SS_EINVAL;    // the tile tables live in LDS (maps up to ~250 x 512 source pixels)
    if (!aligned16(box) || !aligned16(ws) || !aligned16(g_x)) return SS_EINVAL;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return SS_ELAUNCH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned short* Wf = reinterpret_cast<unsigned short*>(ws);
    static const int force_nt2 = getenv("SS_BOX_DGRAD_NT2") ? atoi(getenv("SS_BOX_DGRAD_NT2")) : 0;          // A/B only (profiles/r04/box_nt_ab.log)
    const int NT = (Cin % 128 == 0 && !force_nt2) ? 4 : 2;                      // 128 input channels per workgroup where there are that many
    const long long frag16 = (long long)(Cin / 32) * (Cout / kBxCo) * kB2KS * 3 * 64;
    hipLaunchKernelGGL(upconv_box_dgrad_prep_kernel<0>, dim3(grid_for(frag16, 4096)), dim3(kBlock), 0, s, weight, Wf, Cin, Cout, NT);
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    const int RG = n_row_tiles, CG = (w + kB2TC - 1) / kB2TC;
    const long long n_tiles = NB * RG * CG * (Cin / (32 * NT));
    const unsigned grid = (unsigned)(n_tiles < 2 * cus ? n_tiles : 2 * cus);     // two workgroups per CU, persistent over their tile ranges
    const unsigned short* Bp = static_cast<const unsigned short*>(box);
    if (NT == 4) hipLaunchKernelGGL((upconv_box_dgrad_kernel<4, 1>), dim3(grid), dim3(kB2Threads), 0, s, Bp, Wf, vmap, hmap, tile_rows, tile_cols, g_x,
                                    (int)NB, h, w, NVR, NHR, Cin, Cout / kBxCo, RG, CG);
    else hipLaunchKernelGGL((upconv_box_dgrad_kernel<2, 2>), dim3(grid), dim3(kB2Threads), 0, s, Bp, Wf, vmap, hmap, tile_rows, tile_cols, g_x,
                            (int)NB, h, w, NVR, NHR, Cin, Cout / kBxCo, RG, CG);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

// x16: the 16-bit modes' plan (chunk groups of NG = 2 where C_out allows, then NT <= 2: BxT comment); *NG is written only then
static int box_wgrad_plan(int Cin, int Cout, int* NT, int* kinds, int* slices, bool x16 = false, int* NG = nullptr)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return 0;
    static const int force_nt2 = getenv("SS_BOX_WGRAD_NT2") ? atoi(getenv("SS_BOX_WGRAD_NT2")) : 0;          // A/B only (profiles/r04/box_nt_ab.log)
    static const int force_ng1 = getenv("SS_BOX_X16_NG1") ? atoi(getenv("SS_BOX_X16_NG1")) : 0;              // A/B only (profiles/r06/box_x16_ng_ab.log)
    const int ng = (x16 && Cout % (2 * kBxCo) == 0 && !force_ng1) ? 2 : 1;
    if (NG) *NG = ng;
    *NT = (Cin % 128 == 0 && !force_nt2 && ng == 1) ? 4 : (Cin % 64 == 0 ? 2 : 1);        // input-channel tiles per kind (= per wavefront)
    *kinds = (Cout / (kBxCo * ng)) * (Cin / (32 * *NT));
    int sl = (2 * cus) / *kinds;                                                // two workgroups per CU
    if (sl < 1) sl = 1;
    *slices = sl;
    return 1;
}

int ss_upconv_box_wgrad_supported(int Cin, int Cout, int k, int max_tile_rows, int max_cols32)
{
    if (k != 5 || Cin < 32 || Cin % 32 != 0 || Cout < kBxCo || Cout % kBxCo != 0) return 0;
    return max_tile_rows > 0 && max_tile_rows <= kB2WR && max_cols32 > 0 && max_cols32 <= kB2WC;
}

long long ss_upconv_box_wgrad_ws_floats(int Cin, int Cout, long long NB, int h, int w)
{
    int NT = 0, kinds = 0, slices = 0, nt16 = 0, kinds16 = 0, slices16 = 0, ng16 = 0;
    if (!ss_upconv_box_wgrad_supported(Cin, Cout, 5, 1, 1) || NB <= 0 || h <= 0 || w <= 0 || !box_wgrad_plan(Cin, Cout, &NT, &kinds, &slices)) return 0;
    if (!box_wgrad_plan(Cin, Cout, &nt16, &kinds16, &slices16, true, &ng16)) return 0;
    if (slices16 > slices) slices = slices16;                                   // one size for both entry points (the 16-bit plan has fewer kinds, hence more slices)
    return (long long)slices * Cout * 25 * Cin + (NB * h * ((w + 15) / 16) * Cin * 16 + 1) / 2 + 8;      // slice partials + the spike operand in fragment order (bf16)
}

int ss_upconv_box_wgrad_f32(const void* box, const float* x, const unsigned int* x_packed, const int* vmap, const int* hmap, const int* tile_rows, int n_row_tiles,
                            const int* tile_cols, float* g_w, float* ws, long long NB, int Cin, int Cout, int h, int w, int NVR, int NHR, int accumulate,
                            void* stream)
{
    if (x_packed && (NB * h * (long long)w * Cin) % 16 != 0) return SS_EINVAL;
    if (!box || (!x && !x_packed) || !vmap || !hmap || !tile_rows || !tile_cols || !g_w || !ws || NB <= 0 || h <= 0 || w <= 0 || NVR <= 0 || NHR <= 0 || n_row_tiles <= 0) return SS_EINVAL;
    if (!ss_upconv_box_wgrad_supported(Cin, Cout, 5, 1, 1) || !aligned16(box) || !aligned16(ws)) return SS_EINVAL;
    if (!ss_upconv_box_tiles_supported(n_row_tiles, w, NB * h * (long long)w)) return SS_EINVAL;    // the tile tables live in LDS
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
    const int RG = n_row_tiles, CG = (w + kB2TC - 1) / kB2TC;
#define SS_BW(NT_, PF_) hipLaunchKernelGGL((upconv_box_wgrad_kernel<NT_, PF_>), dim3(grid), dim3(kB2Threads), 0, s, Bp, xT, vmap, hmap, tile_rows, tile_cols, ws, \
                                      (int)NB, h, w, NVR, NHR, Cin, Cout, kinds, RG, CG)
    if (NT == 4) SS_BW(4, false); else if (NT == 2) SS_BW(2, true); else SS_BW(1, true);
#undef SS_BW
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    hipLaunchKernelGGL(upconv_box_wgrad_reduce_kernel, dim3(grid_for((long long)Cout * 25 * Cin, 1024)), dim3(kBlock), 0, s, ws, g_w, slices, Cout, Cin, accumulate);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

/* ---- the three kernels on 16-bit activation gradients (ABI 9).  g_out / g_x in `dtype`; the box image holds ss_upconv_box_planes_x16(dtype) planes of that
   format (one); the weight is rounded once to `dtype`; x: the dense 16-bit spike tensor of the same dtype, or the 2-bit packed one; g_w fp32. */
int ss_upconv_box_planes_x16(int dtype) { return dtype == SS_DT_BF16 ? BxT<SS_DT_BF16>::NP : (dtype == SS_DT_F16 ? BxT<SS_DT_F16>::NP : 0); }

int ss_upconv_boxsum_x16(const void* g_out, const int* vr, const int* hr, void* box, long long NB, int Cout, int H, int W, int NVR, int NHR, int dtype, void* stream)
{
    if (!g_out || !vr || !hr || !box || NB <= 0 || H <= 0 || W <= 0 || NVR <= 0 || NHR <= 0 || Cout <= 0 || Cout % kBxCo != 0) return SS_EINVAL;
    if (!aligned16(g_out) || !aligned16(box) || (dtype != SS_DT_F16 && dtype != SS_DT_BF16)) return SS_EINVAL;
    const int NCH = Cout / kBxCo;
    const int CG = NCH % 4 == 0 ? 4 : (NCH % 2 == 0 ? 2 : 1);
    const long long ib = (NHR + kBlock / CG - 1) / (kBlock / CG), jb = (NVR + kBxJS - 1) / kBxJS;
    const long long grid = NB * (NCH / CG) * jb * ib;
    if (grid > 0x7fffffffLL) return SS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned short* Bp = static_cast<unsigned short*>(box);
    const unsigned short* g16 = static_cast<const unsigned short*>(g_out);
#define SS_BS16(CG_, DTT) hipLaunchKernelGGL((upconv_boxsum_kernel<CG_, DTT>), dim3((unsigned)grid), dim3(kBlock), 0, s, g16, vr, hr, Bp, (int)NB, H, W, Cout, NVR, NHR)
#define SS_BS16D(DTT) do { if (CG == 4) SS_BS16(4, DTT); else if (CG == 2) SS_BS16(2, DTT); else SS_BS16(1, DTT); } while (0)
    if (dtype == SS_DT_F16) SS_BS16D(SS_DT_F16); else SS_BS16D(SS_DT_BF16);
#undef SS_BS16D
#undef SS_BS16
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_upconv_box_dgrad_x16(const void* box, const float* weight, const int* vmap, const int* hmap, const int* tile_rows, int n_row_tiles, const int* tile_cols,
                            void* g_x, float* ws, long long NB, int Cin, int Cout, int h, int w, int NVR, int NHR, int dtype, void* stream)
{
    if (!box || !weight || !vmap || !hmap || !tile_rows || !tile_cols || !g_x || !ws || NB <= 0 || h <= 0 || w <= 0 || NVR <= 0 || NHR <= 0 || n_row_tiles <= 0) return SS_EINVAL;
    if (!ss_upconv_box_dgrad_supported(Cin, Cout, 5, 1, 1) || (dtype != SS_DT_F16 && dtype != SS_DT_BF16)) return SS_EINVAL;
    if (!ss_upconv_box_tiles_supported(n_row_tiles, w, NB * h * (long long)w)) return SS_EINVAL;
    if (!aligned16(box) || !aligned16(ws) || !aligned16(g_x)) return SS_EINVAL;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return SS_ELAUNCH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned short* Wf = reinterpret_cast<unsigned short*>(ws);
    const int NT = Cin % 128 == 0 ? 4 : 2;
    static const int force_ng1 = getenv("SS_BOX_X16_NG1") ? atoi(getenv("SS_BOX_X16_NG1")) : 0;                // A/B only (profiles/r06/box_x16_ng_ab.log)
    const int NG = (Cout % (2 * kBxCo) == 0 && !force_ng1) ? 2 : 1;           // chunks per window (BxT comment)
    const long long frag16 = (long long)(Cin / 32) * (Cout / kBxCo) * kB2KS * 64;
#define SS_BP16(DTT, NGG) hipLaunchKernelGGL((upconv_box_dgrad_prep_kernel<DTT, NGG>), dim3(grid_for(frag16, 4096)), dim3(kBlock), 0, s, weight, Wf, Cin, Cout, NT)
    if (dtype == SS_DT_F16) { if (NG == 2) SS_BP16(SS_DT_F16, 2); else SS_BP16(SS_DT_F16, 1); }
    else { if (NG == 2) SS_BP16(SS_DT_BF16, 2); else SS_BP16(SS_DT_BF16, 1); }
#undef SS_BP16
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    const int RG = n_row_tiles, CG = (w + kB2TC - 1) / kB2TC;
    const long long n_tiles = NB * RG * CG * (Cin / (32 * NT));
    const unsigned grid = (unsigned)(n_tiles < 2 * cus ? n_tiles : 2 * cus);
    const unsigned short* Bp = static_cast<const unsigned short*>(box);
    unsigned short* gx16 = static_cast<unsigned short*>(g_x);
#define SS_BD16K(NT_, KPS_, DTT, NGG) hipLaunchKernelGGL((upconv_box_dgrad_kernel<NT_, KPS_, DTT, NGG>), dim3(grid), dim3(kB2Threads), 0, s, Bp, Wf, vmap, hmap, tile_rows, \
                                                        tile_cols, gx16, (int)NB, h, w, NVR, NHR, Cin, Cout / (kBxCo * NGG), RG, CG)
    // stage sizes: the fp32 mode's 12 KB where NT = 2 (NG x 3 | 6 k-steps), 16 | 12 KB at NT = 4
#define SS_BD16(DTT) do { \
        if (NG == 2) { if (NT == 4) SS_BD16K(4, 2, DTT, 2); else SS_BD16K(2, 3, DTT, 2); } \
        else { if (NT == 4) SS_BD16K(4, 3, DTT, 1); else SS_BD16K(2, 6, DTT, 1); } } while (0)
    if (dtype == SS_DT_F16) SS_BD16(SS_DT_F16); else SS_BD16(SS_DT_BF16);
#undef SS_BD16
#undef SS_BD16K
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_upconv_box_wgrad_x16(const void* box, const void* x, const unsigned int* x_packed, const int* vmap, const int* hmap, const int* tile_rows, int n_row_tiles,
                            const int* tile_cols, float* g_w, float* ws, long long NB, int Cin, int Cout, int h, int w, int NVR, int NHR, int accumulate,
                            int dtype, void* stream)
{
    if (x_packed && (NB * h * (long long)w * Cin) % 16 != 0) return SS_EINVAL;
    if (!box || (!x && !x_packed) || !vmap || !hmap || !tile_rows || !tile_cols || !g_w || !ws || NB <= 0 || h <= 0 || w <= 0 || NVR <= 0 || NHR <= 0 || n_row_tiles <= 0) return SS_EINVAL;
    if (!ss_upconv_box_wgrad_supported(Cin, Cout, 5, 1, 1) || !aligned16(box) || !aligned16(ws) || (dtype != SS_DT_F16 && dtype != SS_DT_BF16)) return SS_EINVAL;
    if (!ss_upconv_box_tiles_supported(n_row_tiles, w, NB * h * (long long)w)) return SS_EINVAL;
    int NT = 0, kinds = 0, slices = 0, NG = 1;
    if (!box_wgrad_plan(Cin, Cout, &NT, &kinds, &slices, true, &NG)) return SS_ELAUNCH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    long long part = (long long)slices * Cout * 25 * Cin;
    part = (part + 3) / 4 * 4;
    unsigned short* xT = reinterpret_cast<unsigned short*>(ws + part);
    const int xg = grid_for(NB * h * ((w + 15) / 16) * Cin, kMaxGridBwd);
    const unsigned grid = (unsigned)(kinds * slices);
    const unsigned short* Bp = static_cast<const unsigned short*>(box);
    const int RG = n_row_tiles, CG = (w + kB2TC - 1) / kB2TC;
#define SS_BW16K(NT_, PF_, DTT, NGG) hipLaunchKernelGGL((upconv_box_wgrad_kernel<NT_, PF_, DTT, NGG>), dim3(grid), dim3(kB2Threads), 0, s, Bp, xT, vmap, hmap, tile_rows, \
                                                       tile_cols, ws, (int)NB, h, w, NVR, NHR, Cin, Cout, kinds, RG, CG)
#define SS_BW16(DTT) do { \
        if (x_packed) hipLaunchKernelGGL((upconv_bwd_xprep_kernel<true, DTT>), dim3(xg), dim3(kBlock), 0, s, static_cast<const void*>(x_packed), xT, NB * h, w, Cin); \
        else hipLaunchKernelGGL((upconv_bwd_xprep_kernel<false, DTT>), dim3(xg), dim3(kBlock), 0, s, x, xT, NB * h, w, Cin); \
        if (NG == 2) { if (NT == 2) SS_BW16K(2, false, DTT, 2); else SS_BW16K(1, true, DTT, 2); } \
        else if (NT == 4) SS_BW16K(4, false, DTT, 1); else if (NT == 2) SS_BW16K(2, true, DTT, 1); else SS_BW16K(1, true, DTT, 1); } while (0)
    if (dtype == SS_DT_F16) SS_BW16(SS_DT_F16); else SS_BW16(SS_DT_BF16);
#undef SS_BW16
#undef SS_BW16K
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    hipLaunchKernelGGL(upconv_box_wgrad_reduce_kernel, dim3(grid_for((long long)Cout * 25 * Cin, 1024)), dim3(kBlock), 0, s, ws, g_w, slices, Cout, Cin, accumulate);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

}  // extern "C"

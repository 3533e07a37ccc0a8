// ss_upconv_sub.hip — the decoder stage's FORWARD in the sub-pixel ("merged tap") form (round 4; include/ss_neuron.h: ss_upconv_sub_prep_f32,
// ss_upconv_sub_fwd_f32).
//
// Reference: NNConvUpsampling (/root/reference/network/blocks.py:110-132: UpsamplingNearest2d(size = up + k - 1) -> Conv2d(k = 5, bias=False)) at the
// decoder call sites /root/reference/network/SNN_models.py:110-129, input = a spike tensor:
//
//     y[nb][Y][X][co] = sum_{ky, kx, ci} W[co][ci][ky][kx] * x[nb][src_y[Y + ky]][src_x[X + kx]][ci]
//
// The projected form of ss_upconv.hip multiplies at SOURCE resolution (P = x W: 25 C_out products per source pixel and input channel — the minimum) and
// then gathers 25 taps of P per output pixel; fused in one kernel (upconv_fused2_fwd_kernel) P stays in LDS but every 16 x 16 output tile recomputes the
// projection of its 122-pixel source window, 1.9 x the minimum, and half the wavefronts spend their issue slots on the gather.  A nearest resize by ~2
// lets the taps be merged instead: the five rows src_y[Y .. Y + 4] take only 2 - 3 distinct values, in runs — (2, 2, 1) / (1, 2, 2) on the regular
// lattice, (3, 2) / (2, 3) / (1, 3, 1) where the resize repeats a row three times — so with the run structure of an output row (its vertical CLASS) and
// column (horizontal class)
//
//     Wm[cv][ch][co][ci][r][c] = sum_{ky in run r of cv} sum_{kx in run c of ch} W[co][ci][ky][kx]                 (ss_upconv_sub_prep_f32, every step)
//     y[nb][Y][X][co]          = sum_{r, c < 3} sum_ci Wm[class(Y)][class(X)][co][ci][r][c] * x[nb][src_y[Y + k0_r]][src_x[X + k0_c]][ci]
//
// is a plain implicit GEMM — 9 C_in multiply-adds per output element (1.44 x the minimum), no per-tap tensor, no gather, no halo recomputation, the
// output written once.  Pixels of one class pair share their weights: M = 32 of them (4 lattice rows x 8 lattice columns) per MFMA tile.  Host side
// (fused.sub_tables, restated in oracle/np_upconv_sub.py): the classes, and per class BLOCKS of <= 16 output rows / <= 32 output columns with the
// distinct source rows / columns they read (<= 20 / 36) — the few irregular rows and columns form small blocks of their own, so every output pixel belongs
// to exactly one (row block, column block) TILE and nothing is special-cased in the kernel.
//
// Numerics: spike counts are exact in bf16; Wm is added in fp32 (ky outer, kx inner, from +0) and split into three bf16 terms (exact), so every product
// is exact and y differs from the float64 value of the reference formula by the fp32 rounding of the <= 9-term weight sums and the fp32 accumulation of
// the MFMA over 9 C_in products (terms lo, mid, hi per k-step; the running sum's sign alternates per 16-channel group against the bf16 MFMA's downward
// drift).  Tolerance class of every fp32 convolution; NOT bit-identical to the projected kernels (different association).
#include "ss_common.hpp"

namespace {

#ifndef SS_SB_ABLATE
#define SS_SB_ABLATE 0                        // development aid (timing only, wrong results): 1 no window traffic after a tile's first group, 2 no weight
#endif                                       // stream, 4 no MFMAs, 8 no per-stage barriers (profiles/r04/sub_fwd_ablations.log)
#ifndef SS_SB_WGS
#define SS_SB_WGS 3                           // workgroups per CU of the packed-input kernel: 3 (168 registers, 44 B of scratch) measured 12 % faster than 2
#endif                                       // (profiles/r04/bench_sub_fwd_v4*.log); the dense-input fallback keeps 2 (its staging needs the registers)
#ifndef SS_SB_NT_STORE
#define SS_SB_NT_STORE 0                      // A/B: non-temporal 16-byte stores in the epilogue (profiles/r04/bench_sub_fwd_v9.log)
#endif
#ifndef SS_SB_TRACE
#define SS_SB_TRACE 0                         // development aid: wavefront 0 of workgroup 0 records s_memtime stamps of its first (cot, g) iterations (ss_debug_sub_trace)
#endif
#if SS_SB_TRACE
__device__ unsigned long long sb_trace[96][8];
__device__ unsigned long long sb_trace2[32][8];       // per tile: 0 loop top (after the barrier), 1 records in LDS, 2 per-lane setup done, 3 first window + stage committed, 4 last MFMA, 5 epilogue issued
#define SB_STAMP2(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0 && ttile < 32) sb_trace2[ttile][slot] = clock64(); } while (0)
#define SB_STAMP(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0 && titer < 96) sb_trace[titer][slot] = clock64(); } while (0)
#else
#define SB_STAMP(slot) do { } while (0)
#define SB_STAMP2(slot) do { } while (0)
#endif
constexpr int kSbThreads = 256;
constexpr int kSbBR = 16, kSbBC = 32;           // output rows / columns of a tile (4 M-blocks of 4 rows per wavefront; 4 wavefronts of 8 columns)
constexpr int kSbWR = 20, kSbWC = 36;           // distinct source rows / columns a block may read (the on-chip window)
constexpr int kSbVRec = 88, kSbHRec = 168;      // ints per row-block / column-block record: cls, n_out, n_src, out[BR | BC], src[WR | WC], slot[BR | BC][3]
// TALL tiles: a column block of <= 8 columns (the fragments of the irregular column classes, the last piece of a regular one) would keep ONE wavefront busy for
// the time of a whole tile; it is paired with tall row blocks of <= 64 rows instead, the four wavefronts stacked vertically (16 rows each)
constexpr int kSbTR = 64, kSbTWR = 68, kSbTRec = 328;                          // rows, distinct source rows, ints per tall row-block record (same fields)
constexpr int kSbPix = 32;                      // window bytes per source pixel: 16 input channels (one k-step) as bf16
// weight stage: the <= 3 merged taps of one run row x NTERM split terms, 1 KB fragments (32 output channels): 9 KB in the fp32 mode
// 16-bit activation modes (DT != 0, round 5): the taps are rounded ONCE to the operand format (the mode's weight), their merged sums — which need more bits than
// one 16-bit value holds — travel as TWO terms of the format (hi + lo: 16 / 22 significand bits); 2 MFMAs per merged tap instead of 3, the output narrowed on store
template <int DT> struct SbT { static constexpr int NTERM = DT ? 2 : 3, STAGE = 3 * NTERM * 1024; };
constexpr int kSbWnd = kSbWR * kSbWC * kSbPix;  // 23040 B

// W [C_out][C_in][5][5] fp32 -> Wm fragments [pair = cv * NHC + ch][C_out / 32][g = C_in / 16][r 3][c 3][term 3][lane 64][8] bf16: element e of a lane =
// split term of (-1)^g * sum of W[co = 32 cot + (lane & 31)][ci = 16 g + 8 (lane >> 5) + e] over the taps of run (r, c); zero where the class has no such run
template <int DT = 0>
__global__ __launch_bounds__(kBlock) void upconv_sub_prep_kernel(const float* __restrict__ W, const int* __restrict__ vcls, const int* __restrict__ hcls,
                                                                 unsigned short* __restrict__ Wm, int Cin, int Cout, int NVC, int NHC)
{
    constexpr int NTERM = SbT<DT>::NTERM;
    const int NCOT = Cout / 32, G = Cin / 16;
    const long long total = (long long)NVC * NHC * NCOT * G * 9 * 64;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int lane = (int)(i & 63);
        long long q = i >> 6;
        const int c = (int)(q % 3); q /= 3;
        const int r = (int)(q % 3); q /= 3;
        const int g = (int)(q % G); q /= G;
        const int cot = (int)(q % NCOT); q /= NCOT;
        const int ch = (int)(q % NHC), cv = (int)(q / NHC);
        const int ky0 = vcls[cv * 8 + 1 + r], kyn = vcls[cv * 8 + 4 + r], kx0 = hcls[ch * 8 + 1 + c], kxn = hcls[ch * 8 + 4 + c];
        const int co = 32 * cot + (lane & 31);
        u16x8 ph, pm, pl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = 16 * g + 8 * (lane >> 5) + e;
            const float* wp = W + ((long long)co * Cin + ci) * 25;
            float a = 0.f;
            for (int ky = ky0; ky < ky0 + kyn; ++ky)
                for (int kx = kx0; kx < kx0 + kxn; ++kx) a += DT ? widen_op<DT>(round_op<DT>(wp[ky * 5 + kx])) : wp[ky * 5 + kx];
            if (g & 1) a = -a;
            if constexpr (DT != 0) {
                const unsigned short t1 = round_op<DT>(a);
                ph[e] = t1; pm[e] = round_op<DT>(a - widen_op<DT>(t1));
                continue;
            }
            const unsigned short h1 = narrow<SS_DT_BF16>(a);
            const float r1 = a - widen<SS_DT_BF16>(h1);
            const unsigned short h2 = narrow<SS_DT_BF16>(r1);
            const float r2 = r1 - widen<SS_DT_BF16>(h2);
            ph[e] = h1; pm[e] = h2; pl[e] = narrow<SS_DT_BF16>(r2);
        }
        // [.. r][c][term][lane][8]: this thread's three fragments sit 1 KB apart
        unsigned short* o = Wm + ((i >> 6) * NTERM * 64 + lane) * 8;
        *reinterpret_cast<u16x8*>(o) = ph;
        *reinterpret_cast<u16x8*>(o + 64 * 8) = pm;
        if constexpr (NTERM == 3) *reinterpret_cast<u16x8*>(o + 2 * 64 * 8) = pl;
    }
}

// A workgroup (4 wavefronts; three workgroups per CU on packed input) draws tiles from a counter.  Per tile and per 32-channel slice of C_out the accumulators (4 M-blocks of
// 32 pixels per wavefront) stay in registers while the kernel walks the C_in / 16 input-channel groups: the group's window — the tile's distinct source
// rows x columns, 16 channels as bf16 — is staged HBM / L2 -> registers -> LDS (the next group's loads are issued under the current group's last weight
// stage), the merged weights stream L2 -> registers -> LDS double-buffered in stages of one run row (<= 3 taps x 3 terms = 9 KB, 36 MFMAs per wavefront
// between two barriers).  A k-step is one merged tap x 16 channels: the lane's A fragment is ONE 16-byte LDS read at (row slot, column slot) of its pixel,
// each weight fragment is read once per wavefront and feeds all 4 M-blocks: 7 KB of LDS reads per 12 MFMAs.
template <bool PACKED, int DT = 0>
__global__ __launch_bounds__(kSbThreads, PACKED ? SS_SB_WGS : 2) void upconv_sub_fwd_kernel(const void* __restrict__ xin, const unsigned short* __restrict__ Wm,
                                                                       const int* __restrict__ vblk, const int* __restrict__ hblk, const int* __restrict__ tblk,
                                                                       const int* __restrict__ order, unsigned* __restrict__ counter,
                                                                       typename ActT<DT>::type* __restrict__ out, int NB, int h, int w, int H, int W, int CIN, int COUT,
                                                                       int NVB, int NHB, int NHC, int NORD)
{
    constexpr int NTERM = SbT<DT>::NTERM, kSbStage = SbT<DT>::STAGE;         // (shadows the fp32 form's stage size)
    __shared__ __attribute__((aligned(16))) unsigned char wnd[kSbWnd];
    __shared__ __attribute__((aligned(16))) unsigned char bst[2 * kSbStage];
    __shared__ int vrec[kSbTRec], hrec[kSbHRec];
    __shared__ long long s_next;
    const int lane = threadIdx.x & 63, m = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = CIN / 16, NCOT = COUT / 32;
    const long long n_tiles = (long long)NB * NORD;                              // NORD pairs: (row block, wide column block) and (tall row block, narrow column block)
    [[maybe_unused]] int titer = 0;
    [[maybe_unused]] int ttile = 0;
    constexpr int kWItems = PACKED ? 3 : 6;                                     // window items per thread: PACKED (pixel) = one 32-bit word of 16 codes;
                                                                                // dense (pixel, half) = 8 fp32 channels
    constexpr int kSt = (kSbStage / 16 + kSbThreads - 1) / kSbThreads;          // 16-byte pieces per thread and weight stage (3; the last partial)
    // Tiles are drawn from a counter (the first one is the workgroup's index): a frame's 285 tiles cost between 1 / 16 and 1 of a full one, and with 10 - 30
    // tiles per workgroup a static round-robin left the last workgroups running alone (deconv2: 0.88 -> 0.82 ms, profiles/r04/bench_sub_fwd_v7.log).  Tile t
    // = (class-pair block order[t / NB], frame t % NB): the host sorts the blocks by cost, the expensive ones first, so the tail is made of small tiles; the
    // frames of one block run together (its weights stay in L2).
    if (threadIdx.x == 0) s_next = blockIdx.x;
    long long drawn = 0;                                                        // thread 0: the tile after this one, drawn while this one runs
#pragma unroll 1
    for (;;) {
        __syncthreads();                                                        // the previous tile's readers of the records are done; s_next is written
        const long long tile = s_next;
        if (tile >= n_tiles) break;
        SB_STAMP2(0);
        const int pair = order[tile / NB], nb = (int)(tile % NB);
        const bool tall = pair >= NVB * NHB;                                    // pairs behind the normal ones: tall row block * NHB + column block
        const int vb = (tall ? pair - NVB * NHB : pair) / NHB, hb = pair - (pair / NHB) * NHB;
        if (threadIdx.x == 0) drawn = (long long)atomicAdd(counter, 1u) + gridDim.x;
        const int* const vsrc = tall ? tblk + vb * kSbTRec : vblk + vb * kSbVRec;
        const int vlen = tall ? kSbTRec : kSbVRec, vBR = tall ? kSbTR : kSbBR, vWR = tall ? kSbTWR : kSbWR;
        for (int i = threadIdx.x; i < vlen; i += kSbThreads) vrec[i] = vsrc[i];
        for (int i = threadIdx.x; i < kSbHRec; i += kSbThreads) hrec[i] = hblk[hb * kSbHRec + i];
        __syncthreads();
        SB_STAMP2(1);
        const int cv = vrec[0], nv = vrec[1], nsv = vrec[2], ch = hrec[0], nh = hrec[1], nsh = hrec[2];
        const int ngv = vrec[vlen - 1], ngh = hrec[kSbHRec - 1];                      // runs per class (copied into the records' last word by the host)
        const int nblk = (nv + 3) >> 2;
        const int row0 = tall ? 16 * wv : 0, col0 = tall ? 0 : 8 * wv;           // this wavefront's first row / column slot of the tile
        const int pitch = tall ? nsh : kSbWC;                                   // window pixels per source row
        const bool active = tall ? row0 < nv : col0 < nh;                       // wave-uniform: this wavefront has pixels in the tile
        // this lane's pixel in M-block b: row 4 b + (m >> 3), column 8 wv + (m & 7); byte offsets of its source row / column per run
        int rs[4][3], cs[3];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int ri = min(row0 + 4 * b + (m >> 3), nv - 1);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                // the two 16-byte halves of a pixel (channels 0-7 | 8-15) swap places on odd window rows: the 16 lanes of a ds_read_b128 group are two lattice
                // rows x 8 columns — without the swap both rows hit the even 16-byte bank groups (2-way conflict on every A read: SQ_LDS_BANK_CONFLICT
                // 5.3e7 cycles against 4.4e7 active, profiles/r04/pmc_sub_v1.txt)
                const int sl = vrec[3 + vBR + vWR + 3 * ri + r];
                rs[b][r] = sl * (pitch * kSbPix) + 16 * ((sl & 1) ^ half);
            }
        }
        {
            const int cj = min(col0 + (m & 7), nh - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) cs[c] = hrec[3 + kSbBC + kSbWC + 3 * cj + c] * kSbPix;
        }
        const int npix = nsv * nsh;
        const unsigned char* const wbase = reinterpret_cast<const unsigned char*>(Wm) + (long long)(cv * NHC + ch) * NCOT * G * (3 * kSbStage);
        // ---- the window of group g: loads into registers | conversion + LDS stores
        unsigned wreg_p[PACKED ? kWItems : 1];
        // (dense fp32 input — the fallback for inputs nobody packed — is staged synchronously at the group boundary, 3 items at a time: prefetching its
        //  48 registers across the MFMA region spilled)
        unsigned xoff[kWItems];                                                 // element index of the item's first channel in group 0 (< 2^32: checked by the host)
        int loff[kWItems];                                                      // its byte offset in the window (-1: no such item)
#pragma unroll
        for (int u = 0; u < kWItems; ++u) {
            const int it = threadIdx.x + kSbThreads * u;
            const int pix = PACKED ? it : (it >> 1);
            xoff[u] = 0u; loff[u] = -1;
            if (pix < npix) {
                const int wy = pix / nsh, wx = pix - wy * nsh;
                xoff[u] = (unsigned)(((((long long)nb * h + vrec[3 + vBR + wy]) * w) + hrec[3 + kSbBC + wx]) * CIN + (PACKED ? 0 : 8 * (it & 1)));
                loff[u] = (wy * pitch + wx) * kSbPix + (PACKED ? 16 * (wy & 1) : 16 * ((it & 1) ^ (wy & 1)));      // (PACKED: where channels 0-7 go)
            }
        }
        auto win_issue = [&](int g) {
#pragma unroll
            for (int u = 0; u < kWItems; ++u) {
                if (loff[u] >= 0) {
                    if constexpr (PACKED) wreg_p[u] = static_cast<const unsigned*>(xin)[(xoff[u] >> 4) + g];
                }
            }
        };
        auto win_commit = [&]([[maybe_unused]] int g) {
            if constexpr (PACKED) {
#pragma unroll
                for (int u = 0; u < kWItems; ++u) {
                    if (loff[u] >= 0) {
                        unsigned char* const pp = wnd + loff[u];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            u16x8 o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = code_to_op<DT>((wreg_p[u] >> (2 * (8 * q + e))) & 3u);
                            *reinterpret_cast<u16x8*>(q == 0 ? pp : wnd + (loff[u] ^ 16)) = o;
                        }
                    }
                }
            } else if constexpr (DT != 0) {                                  // dense 16-bit spike tensor: already the operand
#pragma unroll
                for (int u0 = 0; u0 < kWItems; u0 += 3) {
                    u16x8 d[3];
#pragma unroll
                    for (int v = 0; v < 3; ++v)
                        if (loff[u0 + v] >= 0) d[v] = *reinterpret_cast<const u16x8*>(static_cast<const unsigned short*>(xin) + xoff[u0 + v] + 16 * g);
#pragma unroll
                    for (int v = 0; v < 3; ++v)
                        if (loff[u0 + v] >= 0) *reinterpret_cast<u16x8*>(wnd + loff[u0 + v]) = d[v];
                }
            } else {
#pragma unroll
                for (int u0 = 0; u0 < kWItems; u0 += 3) {
                    f4 d[3][2];
#pragma unroll
                    for (int v = 0; v < 3; ++v) {
                        if (loff[u0 + v] >= 0) {
                            const float* xp = static_cast<const float*>(xin) + xoff[u0 + v] + 16 * g;
                            d[v][0] = *reinterpret_cast<const f4*>(xp);
                            d[v][1] = *reinterpret_cast<const f4*>(xp + 4);
                        }
                    }
#pragma unroll
                    for (int v = 0; v < 3; ++v) {
                        if (loff[u0 + v] >= 0) {
                            u16x8 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {                       // spike counts: exact in bf16 (the high half of the fp32 pattern)
                                o[e] = (unsigned short)(__float_as_uint(d[v][0][e]) >> 16);
                                o[4 + e] = (unsigned short)(__float_as_uint(d[v][1][e]) >> 16);
                            }
                            *reinterpret_cast<u16x8*>(wnd + loff[u0 + v]) = o;
                        }
                    }
                }
            }
        };
        // ---- weight stages: stage r of (cot, g) = ngh taps x 3 terms, contiguous in Wm
        f4 st[kSt];
        const int stage_bytes = ngh * NTERM * 1024;
        auto stage_issue = [&](const unsigned char* src) {
#pragma unroll
            for (int u = 0; u < kSt; ++u) {
                const int off = (threadIdx.x + kSbThreads * u) * 16;
                if (off < stage_bytes) st[u] = *reinterpret_cast<const f4*>(src + off);
            }
        };
        auto stage_commit = [&](unsigned char* dst) {
#pragma unroll
            for (int u = 0; u < kSt; ++u) {
                const int off = (threadIdx.x + kSbThreads * u) * 16;
                if (off < stage_bytes) *reinterpret_cast<f4*>(dst + off) = st[u];
            }
        };
        auto stage_src = [&](int cot, int g, int r) { return wbase + (((long long)cot * G + g) * 3 + r) * kSbStage; };

        // The (cot, g) loop is compiled once per (runs per row class, runs per column class, all four M-blocks present): with these three as run-time
        // values hipcc guarded every MFMA with a branch and shuttled the accumulators through v_mov_b64 (10 VALU instructions per MFMA:
        // profiles/r04/pmc_sub_v1.txt); the tile picks its instance with one switch.
        auto run_tile = [&](auto ngv_c, auto ngh_c, auto full_c) {
            constexpr int NGV = decltype(ngv_c)::value, NGH = decltype(ngh_c)::value;
            constexpr bool FULL = decltype(full_c)::value;
            // the first (cot, g) of the tile: its latency is exposed
            SB_STAMP2(2);
            win_issue(0);
            stage_issue(stage_src(0, 0, 0));
            win_commit(0);
            stage_commit(bst);
            SB_STAMP2(3);
            f32x16 acc[4];
            int par = 0;                                                        // half of the stage buffer that holds the current stage
#pragma unroll 1
            for (int cot = 0; cot < NCOT; ++cot) {
#pragma unroll 1
                for (int g = 0; g < G; ++g) {
                    if (g == 0) {
#pragma unroll
                        for (int b2 = 0; b2 < 4; ++b2)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[b2][r] = 0.f;
                    } else {                                                    // the running sum's sign alternates per group (odd groups' weights are negated)
#pragma unroll
                        for (int b2 = 0; b2 < 4; ++b2)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[b2][r] = -acc[b2][r];
                    }
                    const bool last_g = g + 1 == G;
                    const bool has_next = !(last_g && cot + 1 == NCOT);
                    const int gn = last_g ? 0 : g + 1, cotn = last_g ? cot + 1 : cot;
                    SB_STAMP(0);
                    lds_barrier();                                              // this group's window and first stage are visible (LDS only: loads stay in flight)
                    SB_STAMP(1);
#pragma unroll
                    for (int r = 0; r < NGV; ++r) {
                        const bool more = r + 1 < NGV;
                        if (!(SS_SB_ABLATE & 2)) {
                            if (more) stage_issue(stage_src(cot, g, r + 1));
                            else if (has_next) stage_issue(stage_src(cotn, gn, 0));
                        }
                        // the next group's window: its (HBM) loads are issued under the FIRST stage — three stages of MFMAs to land; the weights (L2) one stage ahead
                        if (r == 0 && has_next && !(SS_SB_ABLATE & 1)) win_issue(gn);
                        __builtin_amdgcn_sched_barrier(0);
                        if (active) {
                            const unsigned char* const bk = bst + par * kSbStage + lane * 16;
#pragma unroll
                            for (int c = 0; c < NGH; ++c) {
                                s16x8 bp[NTERM];
#pragma unroll
                                for (int p = 0; p < NTERM; ++p) bp[p] = *reinterpret_cast<const s16x8*>(bk + (c * NTERM + p) * 1024);
                                s16x8 a[4];
#pragma unroll
                                for (int b2 = 0; b2 < 4; ++b2) a[b2] = *reinterpret_cast<const s16x8*>(wnd + rs[b2][r] + cs[c]);
#pragma unroll
                                for (int p = NTERM - 1; p >= 0; --p)            // smallest terms first
#pragma unroll
                                    for (int b2 = 0; b2 < 4; ++b2)
                                        if (FULL || b2 == 0) {                  // (partial tiles: the irregular row blocks hold <= 4 rows = one M-block)
                                            if (SS_SB_ABLATE & 4) acc[b2][p] += (float)(a[b2][0] + bp[p][1]);
                                            else acc[b2] = mfma32<DT>(bp[p], a[b2], acc[b2]);       // D^T: rows = channels, columns = pixels
                                        }
                            }
                        }
#if SS_SB_TRACE
                        if (acc[0][0] == 12345.f && acc[1][1] == 1.f && acc[2][2] == 2.f && acc[3][3] == 3.f) sb_trace[95][7] = 1;   // waits for the MFMAs
#endif
                        SB_STAMP(2 + 2 * r);
                        if (more) {
                            if (!(SS_SB_ABLATE & 2)) stage_commit(bst + (par ^ 1) * kSbStage);
                            if (!(SS_SB_ABLATE & 8)) lds_barrier();
                            par ^= 1;
                        }
                        SB_STAMP(3 + 2 * r);
                    }
                    if (has_next) {
                        lds_barrier();                                          // every reader of this group's window is done
                        if (!(SS_SB_ABLATE & 1)) win_commit(gn);
                        if (!(SS_SB_ABLATE & 2)) stage_commit(bst + (par ^ 1) * kSbStage);
                        par ^= 1;
                    }
#if SS_SB_TRACE
                    if (blockIdx.x == 0 && threadIdx.x == 0 && titer < 96) { sb_trace[titer][7] = ((unsigned long long)(tile & 0xffff) << 32) | (unsigned)(cot * 256 + g) | ((unsigned long long)(NGV * 16 + NGH * 4 + (FULL ? 1 : 0)) << 48); }
                    ++titer;
#endif
                }
                SB_STAMP2(4);
                // ---- the product is taken TRANSPOSED (weights as the A operand): D[co = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][pixel = lane & 31], so a lane holds
                //      4 consecutive channels of ONE pixel per register quad -> 4 16-byte stores per M-block instead of 16 4-byte ones (the epilogue of the
                //      first form took 8 - 15 k cycles per tile, a quarter of a deconv1 tile: profiles/r04/sub_trace_v2.log).  Plain stores: L2 merges the four
                //      32-byte pieces of a pixel's 128-byte line.  The sum carries the sign of the last group.
                if (active) {
                    const float fin = ((G - 1) & 1) ? -1.f : 1.f;
                    const int cj = col0 + (m & 7);
                    if (cj < nh) {
                        typename ActT<DT>::type* const ob = out + ((long long)nb * H * W + hrec[3 + cj]) * COUT + 32 * cot + 4 * half;
#pragma unroll
                        for (int b2 = 0; b2 < (FULL ? 4 : 1); ++b2) {
                            const int ri = row0 + 4 * b2 + (m >> 3);
                            if (ri < nv) {
                                typename ActT<DT>::type* const op = ob + (long long)vrec[3 + ri] * W * COUT;
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                {
                                    const f4 v = (f4){acc[b2][4 * q] * fin, acc[b2][4 * q + 1] * fin, acc[b2][4 * q + 2] * fin, acc[b2][4 * q + 3] * fin};
                                    if constexpr (DT != 0) store_act4<DT>(op + 8 * q, v[0], v[1], v[2], v[3]);
                                    else {
#if SS_SB_NT_STORE
                                    __builtin_nontemporal_store(v, reinterpret_cast<f4*>(op + 8 * q));
#else
                                    *reinterpret_cast<f4*>(op + 8 * q) = v;
#endif
                                    }
                                }
                            }
                        }
                    }
                }
            }
        };
        using std::integral_constant;
        const int sel = (ngv == 3 ? 4 : 0) + (ngh == 3 ? 2 : 0) + (nblk > 1 ? 1 : 0);        // (classes have 2 or 3 runs: checked where the host tables are built)
        switch (sel) {
        case 0: run_tile(integral_constant<int, 2>{}, integral_constant<int, 2>{}, std::false_type{}); break;
        case 1: run_tile(integral_constant<int, 2>{}, integral_constant<int, 2>{}, std::true_type{}); break;
        case 2: run_tile(integral_constant<int, 2>{}, integral_constant<int, 3>{}, std::false_type{}); break;
        case 3: run_tile(integral_constant<int, 2>{}, integral_constant<int, 3>{}, std::true_type{}); break;
        case 4: run_tile(integral_constant<int, 3>{}, integral_constant<int, 2>{}, std::false_type{}); break;
        case 5: run_tile(integral_constant<int, 3>{}, integral_constant<int, 2>{}, std::true_type{}); break;
        case 6: run_tile(integral_constant<int, 3>{}, integral_constant<int, 3>{}, std::false_type{}); break;
        case 7: run_tile(integral_constant<int, 3>{}, integral_constant<int, 3>{}, std::true_type{}); break;
        default: break;
        }
        SB_STAMP2(5);
#if SS_SB_TRACE
        ++ttile;
#endif
        __syncthreads();                                                        // every reader of s_next is past it
        if (threadIdx.x == 0) s_next = drawn;
    }
}

}  // namespace

extern "C" {

#if SS_SB_TRACE
int ss_debug_sub_trace(unsigned long long* host_dst)
{
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(sb_trace), sizeof(unsigned long long) * 96 * 8) == hipSuccess ? SS_OK : SS_ELAUNCH;
}
int ss_debug_sub_trace2(unsigned long long* host_dst)
{
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(sb_trace2), sizeof(unsigned long long) * 32 * 8) == hipSuccess ? SS_OK : SS_ELAUNCH;
}
#endif

/* tile geometry the host tables must respect: rows / columns per block, distinct source rows / columns per block, ints per block record */
int ss_upconv_sub_geometry(int* block_rows, int* block_cols, int* window_rows, int* window_cols, int* vrec_ints, int* hrec_ints)
{
    if (block_rows) *block_rows = kSbBR;
    if (block_cols) *block_cols = kSbBC;
    if (window_rows) *window_rows = kSbWR;
    if (window_cols) *window_cols = kSbWC;
    if (vrec_ints) *vrec_ints = kSbVRec;
    if (hrec_ints) *hrec_ints = kSbHRec;
    return 3;                                  /* runs per class */
}

/* the tall row blocks paired with column blocks of <= narrow_cols columns: rows, distinct source rows, ints per record; window pixels a tile may hold */
int ss_upconv_sub_tall_geometry(int* block_rows, int* window_rows, int* trec_ints, int* narrow_cols)
{
    if (block_rows) *block_rows = kSbTR;
    if (window_rows) *window_rows = kSbTWR;
    if (trec_ints) *trec_ints = kSbTRec;
    if (narrow_cols) *narrow_cols = 8;
    return kSbWR * kSbWC;
}

int ss_upconv_sub_supported(int Cin, int Cout, int k)
{
    return k == 5 && Cin > 0 && Cout > 0 && Cin % 16 == 0 && Cout % 32 == 0;
}

long long ss_upconv_sub_wm_elems(int Cin, int Cout, int NVC, int NHC)
{
    if (!ss_upconv_sub_supported(Cin, Cout, 5) || NVC < 1 || NHC < 1) return 0;
    return (long long)NVC * NHC * (Cout / 32) * (Cin / 16) * 27 * 512;      /* bf16 elements */
}

int ss_upconv_sub_prep_f32(const float* weight, const int* vcls, const int* hcls, void* wm, int Cin, int Cout, int NVC, int NHC, void* stream)
{
    if (!weight || !vcls || !hcls || !wm || !ss_upconv_sub_supported(Cin, Cout, 5) || NVC < 1 || NHC < 1 || !aligned16(wm)) return SS_EINVAL;
    const long long total = (long long)NVC * NHC * (Cout / 32) * (Cin / 16) * 9 * 64;
    hipLaunchKernelGGL(upconv_sub_prep_kernel<0>, dim3(grid_for(total, 4096)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       weight, vcls, hcls, static_cast<unsigned short*>(wm), Cin, Cout, NVC, NHC);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

/* 16-bit activation modes (ABI 9): the merged-tap weights as TWO terms of `dtype` (taps rounded once to it first); wm: ss_upconv_sub_wm_elems(...) elements
   (two thirds used) */
int ss_upconv_sub_prep_x16(const float* weight, const int* vcls, const int* hcls, void* wm, int Cin, int Cout, int NVC, int NHC, int dtype, void* stream)
{
    if (!weight || !vcls || !hcls || !wm || !ss_upconv_sub_supported(Cin, Cout, 5) || NVC < 1 || NHC < 1 || !aligned16(wm)) return SS_EINVAL;
    if (dtype != SS_DT_F16 && dtype != SS_DT_BF16) return SS_EINVAL;
    const long long total = (long long)NVC * NHC * (Cout / 32) * (Cin / 16) * 9 * 64;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == SS_DT_F16) hipLaunchKernelGGL(upconv_sub_prep_kernel<SS_DT_F16>, dim3(grid_for(total, 4096)), dim3(kBlock), 0, s, weight, vcls, hcls, static_cast<unsigned short*>(wm), Cin, Cout, NVC, NHC);
    else hipLaunchKernelGGL(upconv_sub_prep_kernel<SS_DT_BF16>, dim3(grid_for(total, 4096)), dim3(kBlock), 0, s, weight, vcls, hcls, static_cast<unsigned short*>(wm), Cin, Cout, NVC, NHC);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

/* ss_upconv_sub_fwd_f32 on 16-bit activations: x (nullable) the dense 16-bit spike tensor, x_packed (nullable) the 2-bit packed one, out in `dtype` */
int ss_upconv_sub_fwd_x16(const void* x, const unsigned int* x_packed, const void* wm, const int* vblk, const int* hblk, const int* order, unsigned int* counter,
                          void* out, long long NB, int Cin, int Cout, int h, int w, int H, int W, int NVB, int NHB, int NHC, const int* tblk, int NTB, int NORD,
                          int dtype, void* stream)
{
    if ((!x && !x_packed) || !wm || !vblk || !hblk || !order || !counter || !out || NB <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || NVB < 1 || NHB < 1 || NHC < 1
        || NORD < 1 || NTB < 0 || (NTB > 0 && !tblk) || NORD > (NVB + NTB) * NHB || (dtype != SS_DT_F16 && dtype != SS_DT_BF16))
        return SS_EINVAL;
    if (!ss_upconv_sub_supported(Cin, Cout, 5) || !aligned16(out) || !aligned16(wm) || (x && !x_packed && !aligned16(x))) return SS_EINVAL;
    if (NB * (long long)NORD > 0x7fffffffLL || NB * h * (long long)w * Cin > 0xffffffffLL) return SS_EINVAL;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return SS_ELAUNCH;
    const long long n_tiles = NB * (long long)NORD;
    const long long wgs = (long long)(x_packed ? SS_SB_WGS : 2) * cus;
    const unsigned grid = (unsigned)(n_tiles < wgs ? n_tiles : wgs);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(counter, 0, sizeof(unsigned), s) != hipSuccess) return SS_ELAUNCH;
    unsigned short* o16 = static_cast<unsigned short*>(out);
    const unsigned short* wmp = static_cast<const unsigned short*>(wm);
#define SS_SUB16(DTT) do { \
        if (x_packed) hipLaunchKernelGGL((upconv_sub_fwd_kernel<true, DTT>), dim3(grid), dim3(kSbThreads), 0, s, static_cast<const void*>(x_packed), wmp, vblk, hblk, \
                                         tblk ? tblk : vblk, order, counter, o16, (int)NB, h, w, H, W, Cin, Cout, NVB, NHB, NHC, NORD); \
        else hipLaunchKernelGGL((upconv_sub_fwd_kernel<false, DTT>), dim3(grid), dim3(kSbThreads), 0, s, x, wmp, vblk, hblk, \
                                tblk ? tblk : vblk, order, counter, o16, (int)NB, h, w, H, W, Cin, Cout, NVB, NHB, NHC, NORD); } while (0)
    if (dtype == SS_DT_F16) SS_SUB16(SS_DT_F16); else SS_SUB16(SS_DT_BF16);
#undef SS_SUB16
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_upconv_sub_fwd_f32(const float* x, const unsigned int* x_packed, const void* wm, const int* vblk, const int* hblk, const int* order, unsigned int* counter,
                          float* out, long long NB, int Cin, int Cout, int h, int w, int H, int W, int NVB, int NHB, int NHC, const int* tblk, int NTB, int NORD, void* stream)
{
    if ((!x && !x_packed) || !wm || !vblk || !hblk || !order || !counter || !out || NB <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || NVB < 1 || NHB < 1 || NHC < 1
        || NORD < 1 || NTB < 0 || (NTB > 0 && !tblk) || NORD > (NVB + NTB) * NHB)
        return SS_EINVAL;
    if (!ss_upconv_sub_supported(Cin, Cout, 5) || !aligned16(out) || !aligned16(wm) || (x && !x_packed && !aligned16(x))) return SS_EINVAL;
    if (NB * (long long)NORD > 0x7fffffffLL || NB * h * (long long)w * Cin > 0xffffffffLL) return SS_EINVAL;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return SS_ELAUNCH;
    const long long n_tiles = NB * (long long)NORD;
    const long long wgs = (long long)(x_packed ? SS_SB_WGS : 2) * cus;
    const unsigned grid = (unsigned)(n_tiles < wgs ? n_tiles : wgs);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(counter, 0, sizeof(unsigned), s) != hipSuccess) return SS_ELAUNCH;       // the tile counter of this launch
    if (x_packed) hipLaunchKernelGGL((upconv_sub_fwd_kernel<true>), dim3(grid), dim3(kSbThreads), 0, s, static_cast<const void*>(x_packed),
                                     static_cast<const unsigned short*>(wm), vblk, hblk, tblk ? tblk : vblk, order, counter, out, (int)NB, h, w, H, W, Cin, Cout, NVB, NHB, NHC, NORD);
    else hipLaunchKernelGGL((upconv_sub_fwd_kernel<false>), dim3(grid), dim3(kSbThreads), 0, s, static_cast<const void*>(x),
                            static_cast<const unsigned short*>(wm), vblk, hblk, tblk ? tblk : vblk, order, counter, out, (int)NB, h, w, H, W, Cin, Cout, NVB, NHB, NHC, NORD);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

}  // extern "C"

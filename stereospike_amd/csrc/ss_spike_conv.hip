// ss_spike_conv.hip — forward of the stride-2 5x5 encoder convolutions on SPIKE inputs as an exact bf16x3 implicit GEMM on the matrix cores
// (include/ss_neuron.h: ss_spike_conv_fwd_f32).
//
// Reference: conv1 / conv2 of the encoder, nn.Conv2d(C, 2C, kernel_size=5, stride=2, padding=2, bias=False)
// (/root/reference/network/SNN_models.py:80-90, 268-278, 455-465), whose input is the spike tensor of the previous stage (values 0 / 1: exact in
// bf16).  Until round 3 their forward was MIOpen's fp32 implicit GEMM: 1.42 ms each at BASELINE config 3, AT the fp32-MFMA rate (130 of 157
// TFLOP/s) — the contraction itself had to change to go faster.  With a spike operand the fp32 weight split EXACTLY into three bf16 terms gives
// exact products on v_mfma_f32_32x32x16_bf16 (3 MFMAs at 16x the fp32 rate), fp32 accumulation: fp32-convolution accuracy, as every other
// exact-split synapse of the build.  No im2col: its patch matrix would be 25x the input (2.9 GB at conv1; measured slower than
// MIOpen in round 1, profiles/r01/conv_as_gemm.log).
//
//   out[nb][oy][ox][co] = sum_{ky,kx,ci} x[nb][2 oy + ky - 2][2 ox + kx - 2][ci] * W[co][ci][ky][kx]        (zero padding)
//
// GEMM view: M = output pixels, N = C_out, K = (tap, ci).  A workgroup (4 wavefronts, TWO per CU) owns a tile of 4 output rows x 32 output
// columns — a wavefront = one output row = the M dimension of the MFMA:
//   * the input window of the tile (11 rows x 67 columns, 32 input channels at a time) is staged in LDS as bf16 — expanded from the 2-bit
//     packed spike tensor (16x less to read than fp32; or converted from a dense fp32 tensor) — 64-B pixels whose 16-B granules sit at slot
//     g ^ ((col >> 2) & 3): lanes = consecutive output columns = every second window column read 8 distinct bank groups per LDS cycle;
//   * an A fragment of a k-step (one tap, 16 input channels) is ONE 16-B LDS read per lane — no VALU work in the main loop at all;
//   * the weight fragments (split once, fragment-ordered by spike_conv_fwd_prep_kernel) stream L2 -> LDS, double-buffered, one k-step per
//     stage, shared by the four wavefronts; 3 x C_out / 32 MFMAs per k-step and wavefront.
// HBM traffic: the packed input (58 MB at conv1), the weights from L2, the output once.
#include "ss_common.hpp"

namespace {

constexpr int kScThreads = 256;
constexpr int kScTR = 4, kScTC = 32;                   // output rows (one per wavefront) x output columns (= MFMA M) of a tile
constexpr int kScWR = 2 * (kScTR - 1) + 5;             // 11 window rows
constexpr int kScWC = 2 * (kScTC - 1) + 5;             // 67 window columns
constexpr int kScPix = 64;                             // bytes per window pixel: 32 input channels as bf16 = 4 granules of 8 channels
constexpr int kScRowB = kScWC * kScPix;

// weight [C_out][C_in][5][5] fp32 -> Bf[chunk of 32 ci][s = tap * 2 + ci16 group][split][co tile][lane][8] bf16: element e of a lane = split term
// of W[co = 32 tile + (lane & 31)][ci = 32 chunk + 16 group + 8 (lane >> 5) + e][ky][kx]   (exact 3-way split, round to nearest)
// DT != 0 (16-bit activation modes): ONE term, the weight rounded once to the operand format
template <int DT = 0>
__global__ __launch_bounds__(kBlock) void spike_conv_fwd_prep_kernel(const float* __restrict__ W, unsigned short* __restrict__ Bf, int Cin, int Cout)
{
    constexpr int NSP = DT ? 1 : 3;
    const int NT = Cout / 32;
    const long long total = (long long)(Cin / 32) * 50 * NSP * NT * 64;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int lane = (int)(i & 63);
        long long r = i >> 6;
        const int t = (int)(r % NT); r /= NT;
        const int sp = (int)(r % NSP); r /= NSP;
        const int s = (int)(r % 50); const int c = (int)(r / 50);
        const int tap = s >> 1, g = s & 1, ky = tap / 5, kx = tap - 5 * ky;
        const int co = 32 * t + (lane & 31);
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = 32 * c + 16 * g + 8 * (lane >> 5) + e;
            const float v = W[(((long long)co * Cin + ci) * 5 + ky) * 5 + kx];
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

// NTW: 32-channel output tiles per workgroup.  NTW == C_out / 32 for the shipped shapes (conv1, conv2); the wide shapes of conv3 / conv4 (C_out 256 / 512)
// run C_out / (32 NTW) workgroup SLICES per tile (slice = blockIdx.x % slices), each re-staging the window — built for the A/B against the library path
// (profiles/r04/conv34_ab.log), not dispatched by the network
// DT != 0 (16-bit activation modes, round 5): ONE weight term on the native matrix-core type, the dense input (if any) is the 16-bit spike tensor itself,
// the output is narrowed once on store
template <int CIN, int COUT, bool PACKED, int NTW = COUT / 32, int DT = 0>
__global__ __launch_bounds__(kScThreads, 2) void spike_conv_fwd_kernel(const void* __restrict__ xin, const unsigned short* __restrict__ Bf,
                                                                       typename ActT<DT>::type* __restrict__ out, int NB, int h, int w, int ho, int wo)
{
    constexpr int NTG = COUT / 32, NT = NTW, NCH = CIN / 32, NSL = NTG / NTW, NSP = DT ? 1 : 3;
    static_assert(NTG % NTW == 0, "slices");
    // k-steps per weight stage and barrier: as many as fit twice beside the 46 KB window at two workgroups per CU (10 | KPS: a tap row is 10 k-steps) —
    // fp32 mode: a tap (2) at C_out 64, one k-step at C_out 128; 16-bit modes (one term): half a tap row (5) at C_out 64, a tap (2) at C_out 128
    constexpr int KPS = NSP * NT <= 2 ? 5 : (NSP * NT <= 6 ? 2 : 1);
    static_assert(10 % KPS == 0, "a stage never spans two tap rows");
    constexpr int KST = NSP * NT * 1024;                                        // bytes of one k-step's weight fragments (of this slice: the LDS stage)
    constexpr int KSTG = NSP * NTG * 1024;                                      // ... of all of C_out: the stride in Bf
    constexpr int STG = KPS * KST;                                              // bytes of one weight stage
    constexpr int LPT = STG / 16 / kScThreads;                                  // whole 16-B pieces per thread and stage (+ a partial round)
    constexpr int REM = STG / 16 - LPT * kScThreads;
    __shared__ __attribute__((aligned(16))) unsigned char wnd[kScWR * kScRowB];
    __shared__ __attribute__((aligned(16))) unsigned char bst[2 * STG];
    const int lane = threadIdx.x & 63;
    const int mb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);            // wavefront = output row of the tile
    const int RG = (ho + kScTR - 1) / kScTR, CG = (wo + kScTC - 1) / kScTC;
    const long long n_tiles = (long long)NB * RG * CG;
    const int slice = NSL > 1 ? (int)(blockIdx.x % NSL) : 0;
    const unsigned gdim = gridDim.x / NSL;
    const unsigned g = xcd_remap(blockIdx.x / NSL, gdim);                       // contiguous tile ranges, neighbours on one XCD (halo rows in its L2)
    const long long t_begin = n_tiles * g / gdim, t_end = n_tiles * (g + 1) / gdim;
    const int tx = lane & 31, half = lane >> 5;
    f4 st[LPT + 1];
    // byte offset inside Bf's k-steps of 16-byte piece p of a stage (slices: NTW KB of each split term of each k-step)
    auto piece = [&](int p) -> long long {
        if constexpr (NSL == 1) return (long long)p * 16;
        else {
            const int ks = p / (NSP * NT * 64), rem = p - ks * (NSP * NT * 64), sp = rem / (NT * 64), q = rem - sp * (NT * 64);
            return (long long)ks * KSTG + (sp * NTG + NT * slice) * 1024 + q * 16;
        }
    };
    auto stage_issue = [&](const unsigned char* src) {
#pragma unroll
        for (int u = 0; u < LPT; ++u) st[u] = *reinterpret_cast<const f4*>(src + piece(threadIdx.x + kScThreads * u));
        if (REM && (int)threadIdx.x < REM) st[LPT] = *reinterpret_cast<const f4*>(src + piece(threadIdx.x + kScThreads * LPT));
    };
    auto stage_commit = [&](unsigned char* dst) {
#pragma unroll
        for (int u = 0; u < LPT; ++u) *reinterpret_cast<f4*>(dst + (threadIdx.x + kScThreads * u) * 16) = st[u];
        if (REM && (int)threadIdx.x < REM) *reinterpret_cast<f4*>(dst + (threadIdx.x + kScThreads * LPT) * 16) = st[LPT];
    };
    // PACKED input: the window of an item (tile, 32-channel chunk) is kPkItems 32-bit words (pixel, 16 codes) — few enough to fetch one item AHEAD into registers,
    // so that no tile starts by waiting for HBM / L2 (the tile's MFMAs are 4 - 8 us; the exposed fetch was a quarter of that)
    constexpr int kPkItems = kScWR * kScWC * 2, kPkIter = (kPkItems + kScThreads - 1) / kScThreads;
    [[maybe_unused]] unsigned pkw[kPkIter];
    [[maybe_unused]] auto pk_fetch = [&](long long tile, int chunk) {
        const unsigned* xp = static_cast<const unsigned*>(xin);
        const int cg_ = (int)(tile % CG);
        const long long rr_ = tile / CG;
        const int rg_ = (int)(rr_ % RG), nb_ = (int)(rr_ / RG);
        const int iy0_ = 2 * kScTR * rg_ - 2, ix0_ = 2 * kScTC * cg_ - 2;
#pragma unroll
        for (int u = 0; u < kPkIter; ++u) {
            const int i = threadIdx.x + kScThreads * u;
            const int pix = i >> 1, j = i & 1;
            const int wy = pix / kScWC, col = pix - wy * kScWC;
            const int iy = iy0_ + wy, ix = ix0_ + col;
            pkw[u] = 0u;
            if (i < kPkItems && iy >= 0 && iy < h && ix >= 0 && ix < w)
                pkw[u] = xp[((((long long)nb_ * h + iy) * w + ix) * CIN + 32 * chunk) / 16 + j];
        }
    };
    if constexpr (PACKED) { if (t_begin < t_end) pk_fetch(t_begin, 0); }
    if (t_begin < t_end) stage_issue(reinterpret_cast<const unsigned char*>(Bf));
#pragma unroll 1
    for (long long tl = t_begin; tl < t_end; ++tl) {
        const int cg = (int)(tl % CG);
        long long rr = tl / CG;
        const int rg = (int)(rr % RG);
        const int nb = (int)(rr / RG);
        const int oy0 = kScTR * rg, ox0 = kScTC * cg;
        const int iy0 = 2 * oy0 - 2, ix0 = 2 * ox0 - 2;                         // input coordinates of window (0, 0)
        const bool active = oy0 + mb < ho;                                      // wave-uniform
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        bool neg = false;
#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {
            __syncthreads();                                                    // the previous chunk's / tile's readers of the window are done
            // ---- input window (32 channels of this chunk) -> LDS as bf16, zero outside the image
            if constexpr (PACKED) {
                // the words of THIS item were fetched during the previous item's MFMAs (pk_fetch below): only the expansion is left
#pragma unroll
                for (int u = 0; u < kPkIter; ++u) {
                    const int i = threadIdx.x + kScThreads * u;
                    const int pix = i >> 1, j = i & 1;
                    const int wy = pix / kScWC, col = pix - wy * kScWC;
                    if (i < kPkItems) {
                        unsigned char* const pp = wnd + wy * kScRowB + col * kScPix;
                        const int swz = (col >> 2) & 3;
#pragma unroll
                        for (int q = 0; q < 2; ++q) {                             // granule 2 j + q = channels 16 j + 8 q .. + 7
                            u16x8 o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = code_to_op<DT>((pkw[u] >> (2 * (8 * q + e))) & 3u);
                            *reinterpret_cast<u16x8*>(pp + (((2 * j + q) ^ swz) << 4)) = o;
                        }
                    }
                }
                if (c + 1 < NCH) pk_fetch(tl, c + 1);
                else if (tl + 1 < t_end) pk_fetch(tl + 1, 0);
            } else if constexpr (DT != 0) {
                const unsigned short* x = static_cast<const unsigned short*>(xin);       // the 16-bit spike tensor: already the operand
                constexpr int kItems = kScWR * kScWC * 4, kIter = (kItems + kScThreads - 1) / kScThreads;   // (pixel, 8-channel granule)
#pragma unroll 1
                for (int u0 = 0; u0 < kIter; u0 += 4) {
                    u16x8 vv[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int i = threadIdx.x + kScThreads * (u0 + v);
                        const int pix = i >> 2, q = i & 3;
                        const int wy = pix / kScWC, col = pix - wy * kScWC;
                        const int iy = iy0 + wy, ix = ix0 + col;
                        vv[v] = (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
                        if (i < kItems && iy >= 0 && iy < h && ix >= 0 && ix < w)
                            vv[v] = *reinterpret_cast<const u16x8*>(x + (((long long)nb * h + iy) * w + ix) * CIN + 32 * c + 8 * q);
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int i = threadIdx.x + kScThreads * (u0 + v);
                        const int pix = i >> 2, q = i & 3;
                        const int wy = pix / kScWC, col = pix - wy * kScWC;
                        if (i < kItems) *reinterpret_cast<u16x8*>(wnd + wy * kScRowB + col * kScPix + ((q ^ ((col >> 2) & 3)) << 4)) = vv[v];
                    }
                }
            } else {
                const float* x = static_cast<const float*>(xin);
                constexpr int kItems = kScWR * kScWC * 4, kIter = (kItems + kScThreads - 1) / kScThreads;   // (pixel, 8-channel granule)
#pragma unroll 1
                for (int u0 = 0; u0 < kIter; u0 += 4) {
                    f4 va[4], vb[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int i = threadIdx.x + kScThreads * (u0 + v);
                        const int pix = i >> 2, q = i & 3;
                        const int wy = pix / kScWC, col = pix - wy * kScWC;
                        const int iy = iy0 + wy, ix = ix0 + col;
                        va[v] = (f4){0.f, 0.f, 0.f, 0.f}; vb[v] = va[v];
                        if (i < kItems && iy >= 0 && iy < h && ix >= 0 && ix < w) {
                            const float* p = x + (((long long)nb * h + iy) * w + ix) * CIN + 32 * c + 8 * q;
                            va[v] = *reinterpret_cast<const f4*>(p); vb[v] = *reinterpret_cast<const f4*>(p + 4);
                        }
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int i = threadIdx.x + kScThreads * (u0 + v);
                        const int pix = i >> 2, q = i & 3;
                        const int wy = pix / kScWC, col = pix - wy * kScWC;
                        if (i < kItems) {
                            u16x8 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {                         // spike counts: exact in bf16 (the high half of the fp32 pattern)
                                o[e] = (unsigned short)(__float_as_uint(va[v][e]) >> 16);
                                o[4 + e] = (unsigned short)(__float_as_uint(vb[v][e]) >> 16);
                            }
                            *reinterpret_cast<u16x8*>(wnd + wy * kScRowB + col * kScPix + ((q ^ ((col >> 2) & 3)) << 4)) = o;
                        }
                    }
                }
            }
            // ---- weight stage 0 of this chunk: in registers since the previous item's last stage (the very first one: since before the loop)
            const unsigned char* const bsrc = reinterpret_cast<const unsigned char*>(Bf) + (long long)c * 50 * KSTG;
            const unsigned char* const bnext = reinterpret_cast<const unsigned char*>(Bf) + (long long)(c + 1 < NCH ? c + 1 : 0) * 50 * KSTG;
            stage_commit(bst);
            __syncthreads();
            // window byte offset of this lane's pixel at tap (0, 0): row 2 mb, column 2 tx
            const unsigned char* const lane_row = wnd + (2 * mb) * kScRowB;
#pragma unroll 1
            for (int ky = 0; ky < 5; ++ky) {
                // The bf16 MFMA's fp32 accumulation drifts DOWN by ~2^-28 of the magnitude sum per instruction — coherent over
                // the 150 / 300 MFMAs of an output element, it reached 2.6x MIOpen's worst error at K = 1600.  The sign of the running sum
                // alternates per tap row (acc = -acc, the spike fragment enters with its sign bits set: exact), which cancels the drift.
                if ((((c * 5 + ky) & 1) != 0) != neg) {
                    neg = !neg;
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[t][r] = -acc[t][r];
                }
                const short sgn = neg ? (short)0x8000 : (short)0;
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
#pragma unroll
                    for (int gi = 0; gi < 2; ++gi) {
                        const int j = kx * 2 + gi;                              // k-step 10 ky + j; stage = k-step / KPS (same MFMA order for any KPS: same result bits)
                        const int sub = j % KPS, sg = ky * (10 / KPS) + j / KPS;
                        const bool first = sub == 0, last = sub == KPS - 1, more = sg + 1 < 50 / KPS;
                        if (first) stage_issue(more ? bsrc + (long long)(sg + 1) * KPS * KSTG : bnext);      // (the last stage of an item fetches stage 0 of the next one)
                        if (active) {
                            const int col = 2 * tx + kx;
                            s16x8 a = *reinterpret_cast<const s16x8*>(lane_row + ky * kScRowB + col * kScPix + (((2 * gi + half) ^ ((col >> 2) & 3)) << 4));
#pragma unroll
                            for (int e = 0; e < 8; ++e) a[e] = (short)(a[e] ^ sgn);          // -0 for a zero spike count: harmless
                            const unsigned char* const bk = bst + (sg & 1) * STG + sub * KST + lane * 16;
                            s16x8 b[NSP * NT];                                  // [split][tile]
#pragma unroll
                            for (int u = 0; u < NSP * NT; ++u) b[u] = *reinterpret_cast<const s16x8*>(bk + u * 1024);
#pragma unroll
                            for (int sp = NSP - 1; sp >= 0; --sp)               // smallest terms first
#pragma unroll
                                for (int t = 0; t < NT; ++t) acc[t] = mfma32<DT>(b[sp * NT + t], a, acc[t]);     // D^T: rows = output channels, columns = pixels
                        }
                        if (last) {
                            if (more) stage_commit(bst + ((sg + 1) & 1) * STG);
                            __syncthreads();
                        }
                    }
                }
            }
        }
        // ---- the product is taken transposed (weights as the A operand): D[co = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][pixel = lane & 31] -> a lane holds 4
        //      consecutive output channels of ONE pixel per register quad: 4 NT 16-byte stores per tile row instead of 16 NT 4-byte ones
        if (active) {
            const int ox = ox0 + tx;
            if (ox < wo) {
                typename ActT<DT>::type* const op = out + (((long long)nb * ho + (oy0 + mb)) * wo + ox) * COUT + 32 * NT * slice + 4 * half;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = neg ? -acc[t][4 * q + e] : acc[t][4 * q + e];
                        store_act4<DT>(op + 32 * t + 8 * q, v[0], v[1], v[2], v[3]);
                    }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// First encoder layer: nn.Conv2d(C_in = 4 | 2, 32, kernel_size=5, stride=1, padding=2, bias=False) on the event-voxel input
// (/root/reference/network/SNN_models.py:75-79, 263-267, 450-454).  K = 25 C_in <= 100: the layer is bound by WRITING its 32-channel output
// (0.92 GB at config 3), MIOpen's grouped-conv kernel takes 1.6 ms for it.  Here: implicit GEMM with BOTH operands split into three bf16 terms
// and the six cross terms of ss_gemm6_f32 kept — exact for the integer event counts the voxeliser produces, fp32-product accuracy for ANY
// fp32 input (no precondition on the data).  The whole weight (<= 7 k-steps x 3 terms) lives in registers as fragments: no staging, no barrier
// in the main loop.  k = (tap, ci) flattened, zero rows beyond tap 24.
// ---------------------------------------------------------------------------------------------------
constexpr int kS1Threads = 256;
constexpr int kS1TR = 16, kS1TC = 32;                  // output rows (4 per wavefront) x output columns of a tile
constexpr int kS1WR = kS1TR + 4, kS1WC = kS1TC + 4;

// DT != 0 (16-bit activation modes): the fp32 input and the weight are rounded ONCE to the operand format (what autocast does to both operands of the
// first layer; event counts are exact), one MFMA per k-step, the output narrowed on store
template <int CI, int DT = 0>
__global__ __launch_bounds__(kS1Threads) void dense_conv_s1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W, typename ActT<DT>::type* __restrict__ out,
                                                                        int NB, int h, int w)
{
    constexpr int COUT = 32, KS = (25 * CI + 15) / 16, TPL = 8 / CI;           // k-steps; taps per lane and k-step
    constexpr int PIXB = CI * 2, PLANE = kS1WR * kS1WC * PIXB;                 // bytes per window pixel and per split plane
    constexpr int NSP = DT ? 1 : 3;
    __shared__ __attribute__((aligned(16))) unsigned char wnd[NSP * PLANE + 16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tx = lane & 31, half = lane >> 5;
    // ---- the weight as B fragments in registers: b[ks][split], element e = split of W[co = tx][ci][ky][kx], k = 16 ks + 8 half + e = tap * CI + ci
    s16x8 b[KS][NSP];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = 16 * ks + 8 * half + e, tap = kk / CI, ci = kk - tap * CI;
            const float v = tap < 25 ? W[((long long)tx * CI + ci) * 25 + tap] : 0.f;
            if constexpr (DT != 0) { b[ks][0][e] = (short)round_op<DT>(v); continue; }
            const __bf16 h1 = (__bf16)v;
            const float r1 = v - (float)h1;
            const __bf16 h2 = (__bf16)r1;
            const __bf16 h3 = (__bf16)(r1 - (float)h2);
            if constexpr (DT == 0) { b[ks][0][e] = __builtin_bit_cast(short, h1); b[ks][1][e] = __builtin_bit_cast(short, h2); b[ks][2][e] = __builtin_bit_cast(short, h3); }
        }
    // window byte offsets of this lane's taps (relative to its output pixel), per k-step: tap -> (ky, kx); padding taps read tap 24 (B is zero there)
    int toff[KS][TPL];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int q = 0; q < TPL; ++q) {
            const int tap = min((16 * ks + 8 * half) / CI + q, 24), ky = tap / 5, kx = tap - 5 * ky;
            toff[ks][q] = (ky * kS1WC + kx) * PIXB;
        }
    const int RG = (h + kS1TR - 1) / kS1TR, CG = (w + kS1TC - 1) / kS1TC;
    const long long n_tiles = (long long)NB * RG * CG;
#pragma unroll 1
    for (long long tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        const int cg = (int)(tl % CG);
        long long rr = tl / CG;
        const int rg = (int)(rr % RG);
        const int nb = (int)(rr / RG);
        const int oy0 = kS1TR * rg, ox0 = kS1TC * cg;
        __syncthreads();
        // ---- input window -> three bf16 split planes (x = xh + xm + xl: exact for integers < 2^24, fp32 accuracy otherwise), zero outside the image
        for (int i = threadIdx.x; i < kS1WR * kS1WC; i += kS1Threads) {
            const int wy = i / kS1WC, col = i - wy * kS1WC;
            const int iy = oy0 - 2 + wy, ix = ox0 - 2 + col;
            float v[CI];
#pragma unroll
            for (int c = 0; c < CI; ++c) v[c] = 0.f;
            if (iy >= 0 && iy < h && ix >= 0 && ix < w) {
                const float* p = x + (((long long)nb * h + iy) * w + ix) * CI;
#pragma unroll
                for (int c = 0; c < CI; ++c) v[c] = p[c];
            }
#pragma unroll
            for (int c = 0; c < CI; ++c) {
                unsigned short* const q = reinterpret_cast<unsigned short*>(wnd + i * PIXB) + c;
                if constexpr (DT != 0) { q[0] = round_op<DT>(v[c]); continue; }
                const __bf16 h1 = (__bf16)v[c];
                const float r1 = v[c] - (float)h1;
                const __bf16 h2 = (__bf16)r1;
                const __bf16 h3 = (__bf16)(r1 - (float)h2);
                if constexpr (DT == 0) {
                    q[0] = __builtin_bit_cast(unsigned short, h1);
                    q[PLANE / 2] = __builtin_bit_cast(unsigned short, h2);
                    q[PLANE] = __builtin_bit_cast(unsigned short, h3);
                }
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int rb = 0; rb < kS1TR / 4; ++rb) {                                // this wavefront's 4 output rows, one M block (32 columns) each
            const int ty = 4 * wave + rb;
            if (oy0 + ty >= h) break;                                           // wave-uniform
            const unsigned char* const pix = wnd + (ty * kS1WC + tx) * PIXB;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s16x8 a[NSP];
#pragma unroll
                for (int sp = 0; sp < NSP; ++sp)
#pragma unroll
                    for (int q = 0; q < TPL; ++q) {
                        const unsigned short* const src = reinterpret_cast<const unsigned short*>(pix + sp * PLANE + toff[ks][q]);
#pragma unroll
                        for (int c = 0; c < CI; ++c) a[sp][q * CI + c] = (short)src[c];
                    }
                if constexpr (DT != 0) {
                    acc = mfma32<DT>(a[0], b[ks][0], acc);
                } else {
                // six cross terms, smallest first: al bh, am bm, ah bl, am bh, ah bm, ah bh
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NSP - 1], b[ks][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NSP > 1 ? 1 : 0], b[ks][NSP > 1 ? 1 : 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[ks][NSP - 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NSP > 1 ? 1 : 0], b[ks][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[ks][NSP > 1 ? 1 : 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[ks][0], acc, 0, 0, 0);
                }
            }
            // (a transposed product with 16-byte plain stores — what pays in the MFMA-bound kernels — measured SLOWER here, 0.295 -> 0.32 ms: this kernel is
            //  bound by writing its output, and 16 non-temporal 128-byte rows per wavefront store beat 4 x 32-byte pieces per line; profiles/r04/bench_f4_epilogues_v2_*.json)
            const long long rowbase = ((long long)nb * h + (oy0 + ty)) * w;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (ox < w) {
                    if constexpr (DT == 0) store_out(out + (rowbase + ox) * COUT + tx, acc[r]);
                    else store_out(out + (rowbase + ox) * COUT + tx, narrow<DT>(acc[r]));
                }
            }
        }
    }
}

}  // namespace

extern "C" {

int ss_spike_conv_fwd_supported(int Cin, int Cout, int k, int stride, int pad)
{
    return k == 5 && stride == 2 && pad == 2 && ((Cin == 32 && Cout == 64) || (Cin == 64 && Cout == 128));
}

/* 1 for the shapes ss_spike_conv_fwd_f32 runs in output-channel SLICES (conv3: 128 -> 256, conv4: 256 -> 512 — 2 / 4 workgroup slices of 128 channels per
   tile).  Measured slower than the library path there (profiles/r04/conv34_ab.log): kept for that comparison, not dispatched by the network. */
int ss_spike_conv_fwd_wide_supported(int Cin, int Cout, int k, int stride, int pad)
{
    return k == 5 && stride == 2 && pad == 2 && ((Cin == 128 && Cout == 256) || (Cin == 256 && Cout == 512));
}

/* ss_spike_conv_fwd_f32 on 16-bit activations (ABI 9): x (nullable) = the dense 16-bit spike tensor, x_packed (nullable) the 2-bit packed one, weight fp32
   (rounded once to `dtype` in the prep kernel), out in `dtype`; ws as for the fp32 form (a third of it is used). */
int ss_spike_conv_fwd_x16(const void* x, const unsigned int* x_packed, const float* weight, void* out, float* ws,
                          long long NB, int Cin, int Cout, int h, int w, int dtype, void* stream)
{
    if ((!x && !x_packed) || !weight || !out || !ws || NB <= 0 || h <= 0 || w <= 0 || (dtype != SS_DT_F16 && dtype != SS_DT_BF16)) return SS_EINVAL;
    if (!ss_spike_conv_fwd_supported(Cin, Cout, 5, 2, 2) || !aligned16(out) || !aligned16(ws) || (x && !x_packed && !aligned16(x))) return SS_EINVAL;
    const int ho = (h + 4 - 5) / 2 + 1, wo = (w + 4 - 5) / 2 + 1;
    if (NB * h * (long long)w * Cin > 0x7fffffffffLL || (x_packed && (NB * h * (long long)w * Cin) % 16 != 0)) return SS_EINVAL;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return SS_ELAUNCH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned short* Bf = reinterpret_cast<unsigned short*>(ws);
    unsigned short* o16 = static_cast<unsigned short*>(out);
    const int pg = grid_for((long long)25 * Cin * Cout / 8, 4096);
    if (dtype == SS_DT_F16) hipLaunchKernelGGL(spike_conv_fwd_prep_kernel<SS_DT_F16>, dim3(pg), dim3(kBlock), 0, s, weight, Bf, Cin, Cout);
    else hipLaunchKernelGGL(spike_conv_fwd_prep_kernel<SS_DT_BF16>, dim3(pg), dim3(kBlock), 0, s, weight, Bf, Cin, Cout);
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    const long long n_tiles = NB * ((ho + kScTR - 1) / kScTR) * ((wo + kScTC - 1) / kScTC);
    const unsigned grid = (unsigned)(n_tiles < 2 * cus ? n_tiles : 2 * cus);
#define SS_SC16(CI, CO, DTT) do { if (x_packed) hipLaunchKernelGGL((spike_conv_fwd_kernel<CI, CO, true, CO / 32, DTT>), dim3(grid), dim3(kScThreads), 0, s, \
                                      static_cast<const void*>(x_packed), Bf, o16, (int)NB, h, w, ho, wo); \
                                  else hipLaunchKernelGGL((spike_conv_fwd_kernel<CI, CO, false, CO / 32, DTT>), dim3(grid), dim3(kScThreads), 0, s, \
                                      x, Bf, o16, (int)NB, h, w, ho, wo); } while (0)
    if (dtype == SS_DT_F16) { if (Cin == 32) SS_SC16(32, 64, SS_DT_F16); else SS_SC16(64, 128, SS_DT_F16); }
    else { if (Cin == 32) SS_SC16(32, 64, SS_DT_BF16); else SS_SC16(64, 128, SS_DT_BF16); }
#undef SS_SC16
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

long long ss_spike_conv_fwd_ws_floats(int Cin, int Cout)
{
    if (Cin <= 0 || Cout <= 0 || Cin % 32 != 0 || Cout % 32 != 0) return 0;
    return (long long)25 * Cin * Cout * 3 / 2;                                  // the weight as three bf16 terms in fragment order
}

int ss_spike_conv_fwd_f32(const float* x, const unsigned int* x_packed, const float* weight, float* out, float* ws,
                          long long NB, int Cin, int Cout, int h, int w, void* stream)
{
    if ((!x && !x_packed) || !weight || !out || !ws || NB <= 0 || h <= 0 || w <= 0) return SS_EINVAL;
    const bool wide = ss_spike_conv_fwd_wide_supported(Cin, Cout, 5, 2, 2) != 0;
    if ((!wide && !ss_spike_conv_fwd_supported(Cin, Cout, 5, 2, 2)) || !aligned16(out) || !aligned16(ws) || (x && !x_packed && !aligned16(x))) return SS_EINVAL;
    const int ho = (h + 4 - 5) / 2 + 1, wo = (w + 4 - 5) / 2 + 1;
    if (NB * h * (long long)w * Cin > 0x7fffffffffLL || (x_packed && (NB * h * (long long)w * Cin) % 16 != 0)) return SS_EINVAL;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return SS_ELAUNCH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned short* Bf = reinterpret_cast<unsigned short*>(ws);
    hipLaunchKernelGGL(spike_conv_fwd_prep_kernel<0>, dim3(grid_for((long long)25 * Cin * Cout * 3 / 8, 4096)), dim3(kBlock), 0, s, weight, Bf, Cin, Cout);
    if (hipGetLastError() != hipSuccess) return SS_ELAUNCH;
    const long long n_tiles = NB * ((ho + kScTR - 1) / kScTR) * ((wo + kScTC - 1) / kScTC);
    unsigned grid = (unsigned)(n_tiles < 2 * cus ? n_tiles : 2 * cus);           // two workgroups per CU, persistent over their tile ranges
#define SS_SC(CI, CO) do { if (x_packed) hipLaunchKernelGGL((spike_conv_fwd_kernel<CI, CO, true>), dim3(grid), dim3(kScThreads), 0, s, \
                               static_cast<const void*>(x_packed), Bf, out, (int)NB, h, w, ho, wo); \
                           else hipLaunchKernelGGL((spike_conv_fwd_kernel<CI, CO, false>), dim3(grid), dim3(kScThreads), 0, s, \
                               static_cast<const void*>(x), Bf, out, (int)NB, h, w, ho, wo); } while (0)
#define SS_SCW(CI, CO) do { const unsigned tg = (unsigned)(n_tiles * (CO / 128) < 2 * cus ? n_tiles : 2 * cus / (CO / 128)); grid = tg * (CO / 128); \
                           if (x_packed) hipLaunchKernelGGL((spike_conv_fwd_kernel<CI, CO, true, 4>), dim3(grid), dim3(kScThreads), 0, s, \
                               static_cast<const void*>(x_packed), Bf, out, (int)NB, h, w, ho, wo); \
                           else hipLaunchKernelGGL((spike_conv_fwd_kernel<CI, CO, false, 4>), dim3(grid), dim3(kScThreads), 0, s, \
                               static_cast<const void*>(x), Bf, out, (int)NB, h, w, ho, wo); } while (0)
    if (Cin == 32) SS_SC(32, 64); else if (Cin == 64) SS_SC(64, 128); else if (Cin == 128) SS_SCW(128, 256); else SS_SCW(256, 512);
#undef SS_SCW
#undef SS_SC
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_dense_conv_s1_fwd_supported(int Cin, int Cout, int k, int stride, int pad)
{
    return k == 5 && stride == 1 && pad == 2 && Cout == 32 && (Cin == 4 || Cin == 2);
}

int ss_dense_conv_s1_fwd_f32(const float* x, const float* weight, float* out, long long NB, int Cin, int Cout, int h, int w, void* stream)
{
    if (!x || !weight || !out || NB <= 0 || h <= 0 || w <= 0 || !ss_dense_conv_s1_fwd_supported(Cin, Cout, 5, 1, 2)) return SS_EINVAL;
    if (NB * h * (long long)w > 0x7fffffffLL) return SS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long n_tiles = NB * ((h + kS1TR - 1) / kS1TR) * ((w + kS1TC - 1) / kS1TC);
    const unsigned grid = (unsigned)(n_tiles < 4096 ? n_tiles : 4096);
    if (Cin == 4) hipLaunchKernelGGL((dense_conv_s1_fwd_kernel<4>), dim3(grid), dim3(kS1Threads), 0, s, x, weight, out, (int)NB, h, w);
    else hipLaunchKernelGGL((dense_conv_s1_fwd_kernel<2>), dim3(grid), dim3(kS1Threads), 0, s, x, weight, out, (int)NB, h, w);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

/* ss_dense_conv_s1_fwd_f32 with a 16-bit output (ABI 9): x fp32 (the event-voxel input), x and weight rounded once to `dtype`, out in `dtype` */
int ss_dense_conv_s1_fwd_x16(const float* x, const float* weight, void* out, long long NB, int Cin, int Cout, int h, int w, int dtype, void* stream)
{
    if (!x || !weight || !out || NB <= 0 || h <= 0 || w <= 0 || !ss_dense_conv_s1_fwd_supported(Cin, Cout, 5, 1, 2) || (dtype != SS_DT_F16 && dtype != SS_DT_BF16)) return SS_EINVAL;
    if (NB * h * (long long)w > 0x7fffffffLL) return SS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long n_tiles = NB * ((h + kS1TR - 1) / kS1TR) * ((w + kS1TC - 1) / kS1TC);
    const unsigned grid = (unsigned)(n_tiles < 4096 ? n_tiles : 4096);
    unsigned short* o16 = static_cast<unsigned short*>(out);
#define SS_S116(CI_) do { if (dtype == SS_DT_F16) hipLaunchKernelGGL((dense_conv_s1_fwd_kernel<CI_, SS_DT_F16>), dim3(grid), dim3(kS1Threads), 0, s, x, weight, o16, (int)NB, h, w); \
                          else hipLaunchKernelGGL((dense_conv_s1_fwd_kernel<CI_, SS_DT_BF16>), dim3(grid), dim3(kS1Threads), 0, s, x, weight, o16, (int)NB, h, w); } while (0)
    if (Cin == 4) SS_S116(4); else SS_S116(2);
#undef SS_S116
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

}  // extern "C"

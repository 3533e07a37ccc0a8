// ss_misc.hip — Winograd data-gradient transforms, voxeliser, fused loss statistics / gradient + their C-ABI entry points (include/ss_neuron.h).
#include "ss_common.hpp"

namespace {

// ---------------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) DATA GRADIENT of a 3x3 / stride 1 / pad 1 convolution, NHWC (the four bottleneck convs)
// ---------------------------------------------------------------------------------------------------
// Reference: the autograd backward of SEWResBlock's conv1 / conv2 (/root/reference/network/blocks.py:146-159) w.r.t. their input:
//   g_in[nb][y][x][ci] = sum_{co, a, b} g[nb][y + a - 1][x + b - 1][co] * Wf[a][b][co][ci],      Wf[a][b][co][ci] = W[co][ci][2 - a][2 - b]
// — a dense x dense contraction (no spike operand, so no exact bf16 split).  As 2 x 2 output tiles on 4 x 4 input tiles (Lavin & Gray):
//   V = B^T d B (input transform), U = G Wf G^T (weights), M_k = V_k U_k for the 16 transform positions k (ONE batched fp32 GEMM
//   [16][tiles x C_out] @ [16][C_out x C_in] on the library), Y = A^T M A (output transform): 2.25x fewer multiplications than the
//   direct form.  fp32 throughout; the transforms only add / subtract (weights: x 0.5), every op rounds once (-ffp-contract=off), so the
//   three kernels are bit-exact against oracle/np_winograd.py.
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
__global__ __launch_bounds__(kBlock) void wino_dgrad_weights_kernel(const float* __restrict__ Wt, float* __restrict__ U, int Co, int Ci)
{
    const long long n = (long long)Co * Ci;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        const int co = (int)(i / Ci), ci = (int)(i - (long long)co * Ci);
        const float* wp = Wt + i * 9;                                   // W[co][ci][ky][kx]
        float f[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) f[a][b] = wp[(2 - a) * 3 + (2 - b)];
        float t[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            t[0][b] = f[0][b];
            t[1][b] = 0.5f * ((f[0][b] + f[1][b]) + f[2][b]);
            t[2][b] = 0.5f * ((f[0][b] - f[1][b]) + f[2][b]);
            t[3][b] = f[2][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float u0 = t[a][0], u1 = 0.5f * ((t[a][0] + t[a][1]) + t[a][2]), u2 = 0.5f * ((t[a][0] - t[a][1]) + t[a][2]), u3 = t[a][2];
            U[((long long)(4 * a + 0) * Co + co) * Ci + ci] = u0;
            U[((long long)(4 * a + 1) * Co + co) * Ci + ci] = u1;
            U[((long long)(4 * a + 2) * Co + co) * Ci + ci] = u2;
            U[((long long)(4 * a + 3) * Co + co) * Ci + ci] = u3;
        }
    }
}

// g [NB][H][W][C] -> V[16][T][C], T = NB * th * tw tiles (th = ceil(H / 2), tw = ceil(W / 2)); a lane owns 4 consecutive channels of a tile
__global__ __launch_bounds__(kBlock) void wino_dgrad_input_kernel(const float* __restrict__ g, float* __restrict__ V, long long T, int H, int W,
                                                                  int C, int th, int tw)
{
    const int C4 = C / 4;
    const long long n = T * C4;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        const long long tile = i / C4;
        const int c = (int)(i - tile * C4) * 4;
        const int tx = (int)(tile % tw);
        const long long r = tile / tw;
        const int ty = (int)(r % th);
        const long long nb = r / th;
        f4 d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int y = 2 * ty - 1 + a;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int x = 2 * tx - 1 + b;
                d[a][b] = (y >= 0 && y < H && x >= 0 && x < W) ? *reinterpret_cast<const f4*>(g + ((nb * H + y) * W + x) * C + c)
                                                               : (f4){0.f, 0.f, 0.f, 0.f};
            }
        }
        f4 t[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            t[0][b] = d[0][b] - d[2][b];
            t[1][b] = d[1][b] + d[2][b];
            t[2][b] = d[2][b] - d[1][b];
            t[3][b] = d[1][b] - d[3][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float* vp = V + ((long long)(4 * a) * T + tile) * C + c;
            *reinterpret_cast<f4*>(vp) = t[a][0] - t[a][2];
            *reinterpret_cast<f4*>(vp + T * C) = t[a][1] + t[a][2];
            *reinterpret_cast<f4*>(vp + 2 * T * C) = t[a][2] - t[a][1];
            *reinterpret_cast<f4*>(vp + 3 * T * C) = t[a][1] - t[a][3];
        }
    }
}

// M[16][T][C] -> g_in [NB][H][W][C]
__global__ __launch_bounds__(kBlock) void wino_dgrad_output_kernel(const float* __restrict__ M, float* __restrict__ gin, long long T, int H, int W,
                                                                   int C, int th, int tw)
{
    const int C4 = C / 4;
    const long long n = T * C4;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        const long long tile = i / C4;
        const int c = (int)(i - tile * C4) * 4;
        const int tx = (int)(tile % tw);
        const long long r = tile / tw;
        const int ty = (int)(r % th);
        const long long nb = r / th;
        f4 m[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) m[a][b] = *reinterpret_cast<const f4*>(M + ((long long)(4 * a + b) * T + tile) * C + c);
        f4 t[2][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            t[0][b] = (m[0][b] + m[1][b]) + m[2][b];
            t[1][b] = (m[1][b] - m[2][b]) - m[3][b];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int y = 2 * ty + a;
            if (y >= H) continue;
            float* op = gin + ((nb * H + y) * W + 2 * tx) * C + c;
            *reinterpret_cast<f4*>(op) = (t[a][0] + t[a][1]) + t[a][2];
            if (2 * tx + 1 < W) *reinterpret_cast<f4*>(op + C) = (t[a][1] - t[a][2]) - t[a][3];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// voxeliser: events -> two-polarity count frames (datasets/MVSEC/utils.py:215-281)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void voxelize_kernel(const double* __restrict__ ev, long long E, const double* __restrict__ start,
                                                          const double* __restrict__ end, int G, unsigned* __restrict__ counts, int H, int W)
{
    const double t0 = ev[2];
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < E; i += (long long)gridDim.x * kBlock) {
        const double* e = ev + i * 4;
        const double t = e[2] - t0;
        const long long x = (long long)e[0], y = (long long)e[1];          // int(): truncation toward zero
        if (x < 0 || x >= W || y < 0 || y >= H) continue;
        const int ch = (e[3] == 1.0) ? 0 : 1;
        int lo = 0, hi = G;                                                // first g with start[g] >= t
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (start[mid] < t) lo = mid + 1; else hi = mid; }
        for (int g = lo - 1; g >= 0 && g >= lo - 2; --g)                   // the last two frames that started before t
            if (start[g] < t && t < end[g])
                atomicAdd(&counts[(((long long)g * 2 + ch) * H + y) * W + x], 1u);
    }
}

// ---------------------------------------------------------------------------------------------------
// fused loss statistics / gradient (network/loss.py:7-24,44-75; network/metrics.py:83-95)
// ---------------------------------------------------------------------------------------------------
constexpr int kLossT = 16;                        // 16 x 16 output pixels per workgroup (256 lanes)
constexpr int kLossMaxGrid = 65535;

__device__ __forceinline__ float residual_at(const float* __restrict__ pred, const float* __restrict__ gt, int y, int x, int H, int W)
{
    if (y < 0 || y >= H || x < 0 || x >= W) return 0.f;               // zero padding of F.conv2d(..., padding=1)
    const float g = gt[y * W + x];
    return (g != g) ? 0.f : pred[y * W + x] - g;                        // NaN ground truth = invalid pixel -> residual 0
}

__global__ __launch_bounds__(256) void loss_stats_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                         double* __restrict__ partials, int H, int W, int tiles_x, int tiles_per_img)
{
    __shared__ float r[kLossT + 2][kLossT + 2];
    __shared__ float red[5][4];
    const int tile = blockIdx.x % tiles_per_img, img = blockIdx.x / tiles_per_img;
    const int y0 = (tile / tiles_x) * kLossT, x0 = (tile % tiles_x) * kLossT;
    const float* p = pred + (long long)img * H * W;
    const float* g = gt + (long long)img * H * W;
    for (int i = threadIdx.x; i < (kLossT + 2) * (kLossT + 2); i += 256) {
        const int ty = i / (kLossT + 2), tx = i % (kLossT + 2);
        r[ty][tx] = residual_at(p, g, y0 + ty - 1, x0 + tx - 1, H, W);
    }
    __syncthreads();
    const int ty = threadIdx.x / kLossT, tx = threadIdx.x % kLossT;
    const int y = y0 + ty, x = x0 + tx;
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (y < H && x < W) {
        const float gv = g[y * W + x];
        if (gv == gv) {
            const float rc = r[ty + 1][tx + 1];
            // sobelX = [[1,0,-1],[2,0,-2],[1,0,-1]], sobelY = [[1,2,1],[0,0,0],[-1,-2,-1]]  (cross-correlation)
            const float gx = (r[ty][tx] - r[ty][tx + 2]) + 2.f * (r[ty + 1][tx] - r[ty + 1][tx + 2]) + (r[ty + 2][tx] - r[ty + 2][tx + 2]);
            const float gy = (r[ty][tx] + 2.f * r[ty][tx + 1] + r[ty][tx + 2]) - (r[ty + 2][tx] + 2.f * r[ty + 2][tx + 1] + r[ty + 2][tx + 2]);
            v[0] = 1.f; v[1] = rc; v[2] = rc * rc; v[3] = fabsf(gx) + fabsf(gy); v[4] = fabsf(rc);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 5; ++q) { const float w = wave_sum_f32(v[q]); if (lane == 0) red[q][wave] = w; }
    __syncthreads();
    if (threadIdx.x < 5) {
        const int q = threadIdx.x;
        partials[(long long)blockIdx.x * 5 + q] = (double)red[q][0] + (double)red[q][1] + (double)red[q][2] + (double)red[q][3];
    }
}

__global__ __launch_bounds__(256) void loss_finish_kernel(const double* __restrict__ partials, long long n, double* __restrict__ sums)
{
    __shared__ double s[5][256];
    double acc[5] = {0, 0, 0, 0, 0};
    for (long long i = threadIdx.x; i < n; i += 256)
#pragma unroll
        for (int q = 0; q < 5; ++q) acc[q] += partials[i * 5 + q];
#pragma unroll
    for (int q = 0; q < 5; ++q) s[q][threadIdx.x] = acc[q];
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o)
#pragma unroll
            for (int q = 0; q < 5; ++q) s[q][threadIdx.x] += s[q][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 5) sums[threadIdx.x] = s[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                        const double* __restrict__ sums, const float* __restrict__ coef,
                                                        float* __restrict__ g_pred, int H, int W, int tiles_x, int tiles_per_img)
{
    __shared__ float r[kLossT + 4][kLossT + 4];          // residual, halo 2
    __shared__ float sx[kLossT + 2][kLossT + 2];         // sgn(gx) * mask, halo 1
    __shared__ float sy[kLossT + 2][kLossT + 2];
    const int tile = blockIdx.x % tiles_per_img, img = blockIdx.x / tiles_per_img;
    const int y0 = (tile / tiles_x) * kLossT, x0 = (tile % tiles_x) * kLossT;
    const float* p = pred + (long long)img * H * W;
    const float* g = gt + (long long)img * H * W;
    for (int i = threadIdx.x; i < (kLossT + 4) * (kLossT + 4); i += 256) {
        const int ty = i / (kLossT + 4), tx = i % (kLossT + 4);
        r[ty][tx] = residual_at(p, g, y0 + ty - 2, x0 + tx - 2, H, W);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (kLossT + 2) * (kLossT + 2); i += 256) {
        const int ty = i / (kLossT + 2), tx = i % (kLossT + 2);
        const int y = y0 + ty - 1, x = x0 + tx - 1;
        float vx = 0.f, vy = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const float gv = g[y * W + x];
            if (gv == gv) {
                const int a = ty, b = tx;                // r index of (y-1, x-1) is [ty][tx]
                const float gx = (r[a][b] - r[a][b + 2]) + 2.f * (r[a + 1][b] - r[a + 1][b + 2]) + (r[a + 2][b] - r[a + 2][b + 2]);
                const float gy = (r[a][b] + 2.f * r[a][b + 1] + r[a][b + 2]) - (r[a + 2][b] + 2.f * r[a + 2][b + 1] + r[a + 2][b + 2]);
                vx = (gx > 0.f) ? 1.f : (gx < 0.f ? -1.f : 0.f);
                vy = (gy > 0.f) ? 1.f : (gy < 0.f ? -1.f : 0.f);
            }
        }
        sx[ty][tx] = vx; sy[ty][tx] = vy;
    }
    __syncthreads();
    const int ty = threadIdx.x / kLossT, tx = threadIdx.x % kLossT;
    const int y = y0 + ty, x = x0 + tx;
    if (y < H && x < W) {
        const float gv = g[y * W + x];
        float out = 0.f;
        if (gv == gv) {
            const float n = (float)sums[0], s1 = (float)sums[1], c_si = coef[0], c_gm = coef[1];
            const float rc = r[ty + 2][tx + 2];
            // adjoint of the cross-correlation: T(p) = sum_{p'} s(p') K[p - p'],  p' = p - d  =>  K[d] with d in [-1,1]^2
            // sobelX[dy+1][dx+1] = {1,2,1}[dy+1] * {1,0,-1}[dx+1];  sobelY[dy+1][dx+1] = {1,0,-1}[dy+1] * {1,2,1}[dx+1]
            const int a = ty + 1, b = tx + 1;            // s index of p
            float T = 0.f;
            // p' = p - d: d = (dy,dx);  sx[a - dy][b - dx] * sobelX[dy+1][dx+1]
            T += sx[a + 1][b + 1] * 1.f + sx[a + 1][b - 1] * -1.f;      // dy = -1: row weight 1, dx = -1 -> +1, dx = +1 -> -1
            T += sx[a][b + 1] * 2.f + sx[a][b - 1] * -2.f;              // dy = 0
            T += sx[a - 1][b + 1] * 1.f + sx[a - 1][b - 1] * -1.f;      // dy = +1
            T += sy[a + 1][b + 1] * 1.f + sy[a + 1][b] * 2.f + sy[a + 1][b - 1] * 1.f;      // dy = -1: +{1,2,1}
            T += -(sy[a - 1][b + 1] * 1.f + sy[a - 1][b] * 2.f + sy[a - 1][b - 1] * 1.f);   // dy = +1: -{1,2,1}
            out = c_si * (2.f * rc / n - 2.f * s1 / (n * n)) + (c_gm / n) * T;
        }
        g_pred[(long long)img * H * W + y * W + x] = out;
    }
}


}  // namespace

extern "C" {

int ss_voxelize_f64(const double* events, long long E, const double* start, const double* end, int G,
                    unsigned int* counts, int H, int W, void* stream)
{
    if (!start || !end || !counts || E < 0 || G <= 0 || H <= 0 || W <= 0 || (E > 0 && !events)) return SS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(counts, 0, sizeof(unsigned) * (size_t)G * 2 * H * W, s) != hipSuccess) return SS_ELAUNCH;
    if (E == 0) return SS_OK;
    hipLaunchKernelGGL(voxelize_kernel, dim3(grid_for(E)), dim3(kBlock), 0, s, events, E, start, end, G, counts, H, W);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_wino_dgrad_weights_f32(const float* W, float* U, int Cout, int Cin, void* stream)
{
    if (!W || !U || Cout <= 0 || Cin <= 0) return SS_EINVAL;
    hipLaunchKernelGGL(wino_dgrad_weights_kernel, dim3(grid_for((long long)Cout * Cin, 4096)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       W, U, Cout, Cin);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_wino_dgrad_input_f32(const float* g, float* V, long long NB, int H, int W, int C, void* stream)
{
    if (!g || !V || NB < 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0 || !aligned16(g) || !aligned16(V)) return SS_EINVAL;
    if (NB == 0) return SS_OK;
    const int th = (H + 1) / 2, tw = (W + 1) / 2;
    const long long T = NB * th * tw;
    hipLaunchKernelGGL(wino_dgrad_input_kernel, dim3(grid_for(T * (C / 4), kMaxGridBwd)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       g, V, T, H, W, C, th, tw);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_wino_dgrad_output_f32(const float* M, float* g_in, long long NB, int H, int W, int C, void* stream)
{
    if (!M || !g_in || NB < 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0 || !aligned16(M) || !aligned16(g_in)) return SS_EINVAL;
    if (NB == 0) return SS_OK;
    const int th = (H + 1) / 2, tw = (W + 1) / 2;
    const long long T = NB * th * tw;
    hipLaunchKernelGGL(wino_dgrad_output_kernel, dim3(grid_for(T * (C / 4), kMaxGridBwd)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       M, g_in, T, H, W, C, th, tw);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

long long ss_loss_ws_doubles(void) { return (long long)kLossMaxGrid * 5; }

int ss_loss_stats_f32(const float* pred, const float* gt, double* sums, double* ws, long long B, int H, int W, void* stream)
{
    if (!pred || !gt || !sums || !ws || B <= 0 || H <= 0 || W <= 0) return SS_EINVAL;
    const int tx = (W + kLossT - 1) / kLossT, ty = (H + kLossT - 1) / kLossT;
    const long long blocks = B * tx * ty;
    if (blocks > kLossMaxGrid || (long long)H * W > 0x7fffffffLL) return SS_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(loss_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pred, gt, ws, H, W, tx, tx * ty);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, s, ws, blocks, sums);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

int ss_loss_grad_f32(const float* pred, const float* gt, const double* sums, const float* coef, float* g_pred,
                     long long B, int H, int W, void* stream)
{
    if (!pred || !gt || !sums || !coef || !g_pred || B <= 0 || H <= 0 || W <= 0) return SS_EINVAL;
    const int tx = (W + kLossT - 1) / kLossT, ty = (H + kLossT - 1) / kLossT;
    const long long blocks = B * tx * ty;
    if (blocks > 0x7fffffffLL || (long long)H * W > 0x7fffffffLL) return SS_EINVAL;
    hipLaunchKernelGGL(loss_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), pred, gt, sums,
                       coef, g_pred, H, W, tx, tx * ty);
    return hipGetLastError() == hipSuccess ? SS_OK : SS_ELAUNCH;
}

/* sha256 (first 16 hex digits) of the sources this library was built from — every ss_*.hip in the Makefile's unit order, ss_common.hpp, include/ss_neuron.h —
   computed by the Makefile and compiled in: the measurement files under profiles/ record it, bench.py compares it with the library it has loaded */
#ifndef SS_SRC_HASH
#define SS_SRC_HASH "unknown"
#endif
const char* ss_source_hash(void) { return SS_SRC_HASH; }

}  // extern "C"

/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Plain-C, scalar, single-thread restatement of the fused neuron
 * recurrence behind include/ss_neuron.h.  Nothing under stereospike_amd/ links, loads or calls this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may (as the checker).
 *
 * PARITY UNPINNED for the neuron arithmetic: the equations live in the un-vendored, un-pinned third-party
 * package `spikingjelly` (/root/reference/requirements.txt:3), absent from this image.  They are restated
 * from its published clock_driven single-step algorithm (oracle/sj_clock_driven.py is
 * the op-by-op torch form) and anchored on the reference's call sites:
 *   charge / fire / reset : SNN_models.py:78,85,90,95,100,113,118,123,128 (IFNode), :266..316 (LIFNode /
 *                           ParametricLIFNode), blocks.py:150,157
 *   gain                  : blocks.py:106-107 (MultiplyBy) — x*scale is rounded to fp32 BEFORE the charge
 *   skip / SEW add        : SNN_models.py:171,176,181,186; blocks.py:171
 *   firing-rate counts    : SNN_models.py:219-242
 *   I-neuron pool         : SNN_models.py:150,172-188
 * tests/test_oracle.py pins this file bit-for-bit (forward) against oracle/sj_clock_driven.py run through
 * torch autograd, and tests/golden/ pins that module inside the reference's own network/ *.py graph.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: every fp32 op rounds once, like eager PyTorch).
 */
#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <stddef.h>

#define SS_KIND_IF 0
#define SS_KIND_LIF 1
#define SS_KIND_PLIF 2
#define SS_SG_ATAN 0
#define SS_SG_SIGMOID 1

static inline float charge(int kind, float v, float xs, float tau, float k, float v_reset)
{
    if (kind == SS_KIND_IF)
        return v + xs;
    /* (v - 0.f) == v bit-for-bit, so one expression covers both upstream branches (v_reset == 0 or not) */
    float d = xs - (v - v_reset);
    if (kind == SS_KIND_LIF)
        return v + d / tau;
    return v + d * k;
}

int ss_ref_neuron_fwd_f32(const float *x_seq, const float *v_init, const float *skip_seq,
                          float *out_seq, float *h_seq, float *v_last, unsigned long long *nnz,
                          int T, long long N, float scale, int kind, float tau, const float *k_ptr,
                          float v_th, float v_reset)
{
    if (!x_seq || !out_seq || !v_last || T <= 0 || N < 0 || kind < 0 || kind > 2) return -22;
    if (kind == SS_KIND_PLIF && !k_ptr) return -22;
    const float k = (kind == SS_KIND_PLIF) ? *k_ptr : 0.f;
    unsigned long long c_spk = 0, c_out = 0;
    for (long long n = 0; n < N; ++n) {
        float v = v_init ? v_init[n] : v_reset;
        for (int t = 0; t < T; ++t) {
            const size_t i = (size_t)t * (size_t)N + (size_t)n;
            float xs = x_seq[i] * scale;
            float h = charge(kind, v, xs, tau, k, v_reset);
            float z = ((h - v_th) >= 0.f) ? 1.f : 0.f;
            v = (1.f - z) * h + z * v_reset;
            float o = skip_seq ? z + skip_seq[i] : z;
            if (h_seq) h_seq[i] = h;   /* may alias x_seq: x_seq[i] was read above */
            out_seq[i] = o;
            c_spk += (z != 0.f);
            c_out += (o != 0.f);
        }
        v_last[n] = v;
    }
    if (nnz) { nnz[0] += c_spk; nnz[1] += c_out; }
    return 0;
}

static inline float surrogate_grad(int surrogate, float xh, float alpha, float g)
{
    if (surrogate == SS_SG_ATAN) {
        /* upstream: alpha / 2 / (1 + (pi / 2 * alpha * x).pow_(2)) * grad
         * torch evaluates  scalar / tensor  as  tensor.reciprocal() * scalar                        */
        float c = (float)(M_PI / 2.0 * (double)alpha);
        float u = xh * c;
        float p = u * u;
        float r = 1.f / (p + 1.f);
        return (r * (float)((double)alpha / 2.0)) * g;
    }
    /* upstream: sgax = (x * alpha).sigmoid_(); grad * (1. - sgax) * sgax * alpha */
    float s = 1.f / (1.f + expf(-(xh * alpha)));
    return ((g * (1.f - s)) * s) * alpha;
}

int ss_ref_neuron_bwd_f32(const float *g_out_seq, const float *g_v_last, const float *h_seq, const float *v_init,
                          float *g_x_seq, float *g_v_init, float *g_k,
                          int T, long long N, float scale, int kind, float tau, const float *k_ptr,
                          float v_th, float v_reset, int surrogate, float alpha, int detach_reset)
{
    if (!g_out_seq || !h_seq || !g_x_seq || T <= 0 || N < 0 || kind < 0 || kind > 2) return -22;
    if (surrogate != SS_SG_ATAN && surrogate != SS_SG_SIGMOID) return -22;
    if (kind == SS_KIND_PLIF && !k_ptr) return -22;
    const float k = (kind == SS_KIND_PLIF) ? *k_ptr : 0.f;
    double acc_k = 0.0;     /* the oracle keeps the "true" sum in double; the kernel is compared to tolerance */
    for (long long n = 0; n < N; ++n) {
        float g_v = g_v_last ? g_v_last[n] : 0.f;
        for (int t = T - 1; t >= 0; --t) {
            const size_t i = (size_t)t * (size_t)N + (size_t)n;
            float h = h_seq[i];
            float xh = h - v_th;
            float z = (xh >= 0.f) ? 1.f : 0.f;
            float g_s = g_out_seq[i];
            if (!detach_reset)
                g_s = g_s + (g_v * v_reset - g_v * h);
            float g_h = surrogate_grad(surrogate, xh, alpha, g_s) + g_v * (1.f - z);
            float g_x;
            if (kind == SS_KIND_IF) {
                g_x = g_h;
                g_v = g_h;
            } else if (kind == SS_KIND_LIF) {
                g_x = g_h / tau;
                g_v = g_h - g_x;
            } else {
                float v_prev;
                if (t > 0) {
                    float hp = h_seq[i - (size_t)N];
                    float zp = ((hp - v_th) >= 0.f) ? 1.f : 0.f;
                    v_prev = (1.f - zp) * hp + zp * v_reset;
                } else {
                    v_prev = v_init ? v_init[n] : v_reset;
                }
                g_x = g_h * k;
                g_v = g_h - g_x;
                acc_k += (double)g_h * (double)((h - v_prev) / k);   /* as the kernel: fp32 factors, fp64 product and sum */
            }
            g_x_seq[i] = g_x * scale;
        }
        if (g_v_init) g_v_init[n] = g_v;
    }
    if (g_k) *g_k = (float)acc_k;
    return 0;
}

int ss_ref_ipool_fwd_f32(const float *pd_seq, long long stride_t, long long stride_k, const float *v_init,
                         float *depth_seq, int T, int K, long long M, float scale, float v_reset)
{
    if (!pd_seq || !depth_seq || T <= 0 || K <= 0 || M < 0) return -22;
    for (long long m = 0; m < M; ++m) {
        float v = v_init ? v_init[m] : v_reset;
        for (int t = 0; t < T; ++t)
            for (int k = 0; k < K; ++k) {
                float h = v + pd_seq[(size_t)t * stride_t + (size_t)k * stride_k + m] * scale;
                /* IFNode(v_threshold=inf): spike == 0, reset evaluated literally */
                v = (1.f - 0.f) * h + 0.f * v_reset;
                depth_seq[((size_t)t * K + k) * (size_t)M + m] = v;
            }
    }
    return 0;
}

int ss_ref_ipool_bwd_f32(const float *g_depth_seq, const float *g_v_last, float *g_pd_seq,
                         long long stride_t, long long stride_k, float *g_v_init,
                         int T, int K, long long M, float scale)
{
    if (!g_depth_seq || !g_pd_seq || T <= 0 || K <= 0 || M < 0) return -22;
    for (long long m = 0; m < M; ++m) {
        float g_v = g_v_last ? g_v_last[m] : 0.f;
        for (int t = T - 1; t >= 0; --t)
            for (int k = K - 1; k >= 0; --k) {
                g_v = g_depth_seq[((size_t)t * K + k) * (size_t)M + m] + g_v;
                g_pd_seq[(size_t)t * stride_t + (size_t)k * stride_k + m] = g_v * scale;
            }
        if (g_v_init) g_v_init[m] = g_v;
    }
    return 0;
}

/* predict_depth head (SNN_models.py:133-148; blocks.py:124-128) as a gather over per-tap projections — see
 * include/ss_neuron.h ss_upconv1_*.  tests/test_oracle.py pins it against torch's UpsamplingNearest2d + Conv2d. */
int ss_ref_upconv1_fwd_f32(const float *P, const int *src_y, const int *src_x, const float *bias, float *out,
                           long long NB, int k, int h, int w, int H, int W)
{
    if (!P || !src_y || !src_x || !out || NB < 0 || k <= 0) return -22;
    const float b = bias ? *bias : 0.f;
    for (long long nb = 0; nb < NB; ++nb)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float acc = 0.f;
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx)
                        acc += P[((nb * (k * k) + ky * k + kx) * h + src_y[y + ky]) * (long long)w + src_x[x + kx]];
                out[(nb * H + y) * (long long)W + x] = acc + b;
            }
    return 0;
}

int ss_ref_upconv1_bwd_f32(const float *g_out, const int *y_lo, const int *y_hi, const int *x_lo, const int *x_hi,
                           float *g_P, long long NB, int k, int h, int w, int H, int W)
{
    if (!g_out || !y_lo || !y_hi || !x_lo || !x_hi || !g_P || NB < 0 || k <= 0) return -22;
    for (long long nb = 0; nb < NB; ++nb)
        for (int tap = 0; tap < k * k; ++tap)
            for (int iy = 0; iy < h; ++iy)
                for (int ix = 0; ix < w; ++ix) {
                    const int ky = tap / k, kx = tap % k;
                    int y0 = y_lo[iy] - ky, y1 = y_hi[iy] - ky, x0 = x_lo[ix] - kx, x1 = x_hi[ix] - kx;
                    if (y0 < 0) y0 = 0;
                    if (x0 < 0) x0 = 0;
                    if (y1 > H) y1 = H;
                    if (x1 > W) x1 = W;
                    /* rectangle sum, row sums first: C[y] = sum_x (x ascending), then sum_y C[y] (y ascending) */
                    float acc = 0.f;
                    for (int y = y0; y < y1; ++y) {
                        float cs = 0.f;
                        for (int x = x0; x < x1; ++x) cs += g_out[(nb * H + y) * (long long)W + x];
                        acc += cs;
                    }
                    g_P[((nb * (k * k) + tap) * h + iy) * (long long)w + ix] = acc;
                }
    return 0;
}

/* channels-last variants (include/ss_neuron.h ss_upconv_cl_*): P [NB][h][w][k*k*C] (channel = tap*C + c), out [NB][H][W][C] */
int ss_ref_upconv_cl_fwd_f32(const float *P, const int *src_y, const int *src_x, const float *bias, float *out,
                             long long NB, int k, int C, int h, int w, int H, int W)
{
    if (!P || !src_y || !src_x || !out || NB < 0 || k <= 0 || C <= 0) return -22;
    const long long KKC = (long long)k * k * C;
    for (long long nb = 0; nb < NB; ++nb)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                for (int c = 0; c < C; ++c) {
                    float acc = 0.f;
                    for (int ky = 0; ky < k; ++ky)
                        for (int kx = 0; kx < k; ++kx)
                            acc += P[((nb * h + src_y[y + ky]) * (long long)w + src_x[x + kx]) * KKC + (ky * k + kx) * C + c];
                    out[((nb * H + y) * (long long)W + x) * C + c] = acc + (bias ? bias[c] : 0.f);
                }
    return 0;
}

int ss_ref_upconv_cl_bwd_f32(const float *g_out, const int *y_lo, const int *y_hi, const int *x_lo, const int *x_hi,
                             float *g_P, long long NB, int k, int C, int h, int w, int H, int W)
{
    if (!g_out || !y_lo || !y_hi || !x_lo || !x_hi || !g_P || NB < 0 || k <= 0 || C <= 0) return -22;
    const long long KKC = (long long)k * k * C;
    for (long long nb = 0; nb < NB; ++nb)
        for (int iy = 0; iy < h; ++iy)
            for (int ix = 0; ix < w; ++ix)
                for (int tap = 0; tap < k * k; ++tap) {
                    const int ky = tap / k, kx = tap % k;
                    int y0 = y_lo[iy] - ky, y1 = y_hi[iy] - ky, x0 = x_lo[ix] - kx, x1 = x_hi[ix] - kx;
                    if (y0 < 0) y0 = 0;
                    if (x0 < 0) x0 = 0;
                    if (y1 > H) y1 = H;
                    if (x1 > W) x1 = W;
                    for (int c = 0; c < C; ++c) {
                        float acc = 0.f;
                        for (int y = y0; y < y1; ++y) {
                            float cs = 0.f;
                            for (int x = x0; x < x1; ++x) cs += g_out[((nb * H + y) * (long long)W + x) * C + c];
                            acc += cs;
                        }
                        g_P[((nb * h + iy) * (long long)w + ix) * KKC + tap * C + c] = acc;
                    }
                }
    return 0;
}

/* ORACLE — TEST INFRASTRUCTURE.  Drives every entry point of ss_neuron_ref.c on small ragged inputs under
 * -fsanitize=address,undefined (oracle/Makefile target _build/ss_oracle_asan_check; tests/test_oracle.py runs it). */
#include <stdio.h>
#include <stdlib.h>

int ss_ref_neuron_fwd_f32(const float *, const float *, const float *, float *, float *, float *, unsigned long long *,
                          int, long long, float, int, float, const float *, float, float);
int ss_ref_neuron_bwd_f32(const float *, const float *, const float *, const float *, float *, float *, float *,
                          int, long long, float, int, float, const float *, float, float, int, float, int);
int ss_ref_ipool_fwd_f32(const float *, long long, long long, const float *, float *, int, int, long long, float, float);
int ss_ref_ipool_bwd_f32(const float *, const float *, float *, long long, long long, float *, int, int, long long, float);
int ss_ref_upconv1_fwd_f32(const float *, const int *, const int *, const float *, float *, long long, int, int, int, int, int);
int ss_ref_upconv1_bwd_f32(const float *, const int *, const int *, const int *, const int *, float *, long long, int, int, int, int, int);
int ss_ref_upconv_cl_fwd_f32(const float *, const int *, const int *, const float *, float *, long long, int, int, int, int, int, int);
int ss_ref_upconv_cl_bwd_f32(const float *, const int *, const int *, const int *, const int *, float *, long long, int, int, int, int, int, int);

static float *rnd(size_t n) { float *p = malloc(n * sizeof(float) + 1); for (size_t i = 0; i < n; ++i) p[i] = (float)rand() / RAND_MAX - 0.4f; return p; }

int main(void)
{
    int rc = 0;
    for (int kind = 0; kind < 3; ++kind)
        for (int T = 1; T <= 3; ++T) {
            const long long N = 37;
            float *x = rnd(T * N), *sk = rnd(T * N), *vi = rnd(N), *out = rnd(T * N), *h = rnd(T * N), *vl = rnd(N), *g = rnd(T * N);
            float *gx = rnd(T * N), *gv = rnd(N), k = 0.3f, gk = 0;
            unsigned long long nnz[2] = {0, 0};
            rc |= ss_ref_neuron_fwd_f32(x, T > 1 ? vi : NULL, T == 2 ? sk : NULL, out, h, vl, nnz, T, N, 10.f, kind, 3.f, &k, 1.f, 0.1f);
            for (int sg = 0; sg < 2; ++sg)
                rc |= ss_ref_neuron_bwd_f32(g, NULL, h, T > 1 ? vi : NULL, gx, gv, kind == 2 ? &gk : NULL, T, N, 10.f, kind, 3.f, &k,
                                            1.f, 0.1f, sg, 2.f, sg);
            free(x); free(sk); free(vi); free(out); free(h); free(vl); free(g); free(gx); free(gv);
        }
    {
        const int T = 2, K = 4; const long long M = 19;
        float *pd = rnd(T * K * M), *d = rnd(T * K * M), *gp = rnd(T * K * M), *gv = rnd(M);
        rc |= ss_ref_ipool_fwd_f32(pd, K * M, M, NULL, d, T, K, M, 10.f, 0.f);
        rc |= ss_ref_ipool_bwd_f32(d, NULL, gp, K * M, M, gv, T, K, M, 10.f);
        free(pd); free(d); free(gp); free(gv);
    }
    {
        const int k = 3, h = 3, w = 4, H = 7, W = 9, C = 2; const long long NB = 2;
        int sy[9], sx[11], ylo[3], yhi[3], xlo[4], xhi[4];
        for (int i = 0; i < H + k - 1; ++i) sy[i] = i * h / (H + k - 1);
        for (int i = 0; i < W + k - 1; ++i) sx[i] = i * w / (W + k - 1);
        for (int i = 0; i < h; ++i) { ylo[i] = H + k; yhi[i] = 0; }
        for (int i = 0; i < w; ++i) { xlo[i] = W + k; xhi[i] = 0; }
        for (int i = 0; i < H + k - 1; ++i) { if (i < ylo[sy[i]]) ylo[sy[i]] = i; if (i + 1 > yhi[sy[i]]) yhi[sy[i]] = i + 1; }
        for (int i = 0; i < W + k - 1; ++i) { if (i < xlo[sx[i]]) xlo[sx[i]] = i; if (i + 1 > xhi[sx[i]]) xhi[sx[i]] = i + 1; }
        float *P = rnd(NB * k * k * C * h * w), *o = rnd(NB * C * H * W), *gP = rnd(NB * k * k * C * h * w), b[2] = {0.5f, -0.5f};
        rc |= ss_ref_upconv1_fwd_f32(P, sy, sx, b, o, NB * C, k, h, w, H, W);
        rc |= ss_ref_upconv1_bwd_f32(o, ylo, yhi, xlo, xhi, gP, NB * C, k, h, w, H, W);
        rc |= ss_ref_upconv_cl_fwd_f32(P, sy, sx, b, o, NB, k, C, h, w, H, W);
        rc |= ss_ref_upconv_cl_bwd_f32(o, ylo, yhi, xlo, xhi, gP, NB, k, C, h, w, H, W);
        free(P); free(o); free(gP);
    }
    printf("asan_check rc=%d\n", rc);
    return rc;
}

/* Minimal C caller of the C-ABI (include/ss_neuron.h): one fused IF layer, T = 5 steps, forward + recompute backward, on the
 * default stream.  Build (what tests/test_abi.py::test_c_caller_compiles_and_links does):
 *     gcc -std=c99 -Iinclude examples/c_caller.c -Lstereospike_amd/lib -lss_neuron -L/opt/rocm/lib -lamdhip64 -o c_caller
 * Run on an MI355X with LD_LIBRARY_PATH=stereospike_amd/lib:/opt/rocm/lib.  Uses the HIP runtime's C entry points directly. */
#include <stdio.h>
#include <stdlib.h>
#include "ss_neuron.h"

/* the three HIP runtime calls this example needs (declared here so the file compiles without the HIP headers) */
int hipMalloc(void** ptr, size_t size);
int hipMemcpy(void* dst, const void* src, size_t size, int kind);   /* kind: 1 = host->device, 2 = device->host */
int hipDeviceSynchronize(void);

int main(void)
{
    enum { T = 5, N = 4096 };
    const float scale = 10.f, v_th = 1.f, v_reset = 0.f;
    float *hx = (float*)malloc(sizeof(float) * T * N), *hout = (float*)malloc(sizeof(float) * T * N);
    for (int i = 0; i < T * N; ++i) hx[i] = 0.02f * (float)((i * 7919) % 13);
    float *x, *out, *v_last, *g, *gx;
    if (hipMalloc((void**)&x, sizeof(float) * T * N) || hipMalloc((void**)&out, sizeof(float) * T * N) ||
        hipMalloc((void**)&v_last, sizeof(float) * N) || hipMalloc((void**)&g, sizeof(float) * T * N) ||
        hipMalloc((void**)&gx, sizeof(float) * T * N)) { fprintf(stderr, "hipMalloc failed\n"); return 2; }
    hipMemcpy(x, hx, sizeof(float) * T * N, 1);
    hipMemcpy(g, hx, sizeof(float) * T * N, 1);
    /* forward: gain + charge + fire + reset over T steps, no saved h (the backward below recomputes it from x) */
    int rc = ss_neuron_fwd_f32(x, NULL, NULL, out, NULL, v_last, NULL, T, N, scale, SS_KIND_IF, 2.f, NULL, v_th, v_reset, NULL);
    if (rc != SS_OK) { fprintf(stderr, "ss_neuron_fwd_f32 -> %d\n", rc); return 1; }
    if (!ss_neuron_bwd_rc_supported(T)) return 3;
    rc = ss_neuron_bwd_rc_f32(g, NULL, x, NULL, gx, NULL, NULL, NULL, T, N, scale, SS_KIND_IF, 2.f, NULL, v_th, v_reset,
                              SS_SG_ATAN, 2.f, 1, NULL);
    if (rc != SS_OK) { fprintf(stderr, "ss_neuron_bwd_rc_f32 -> %d\n", rc); return 1; }
    hipDeviceSynchronize();
    hipMemcpy(hout, out, sizeof(float) * T * N, 2);
    long spikes = 0;
    for (int i = 0; i < T * N; ++i) spikes += hout[i] != 0.f;
    printf("ABI %d: %ld spikes out of %d updates\n", ss_abi_version(), spikes, T * N);
    return 0;
}

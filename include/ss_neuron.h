/*
 * ss_neuron.h — C-ABI of the MI355X-native StereoSpike neuron engine (libss_neuron.so).
 *
 * This is the drop-in boundary for the one hot path BASELINE.json names: the post-conv point-wise
 * chain of every spiking stage of the reference network, fused over the T-step loop
 *
 *      MultiplyBy            /root/reference/network/blocks.py:106-107      y = x * scale
 *      IF/LIF/PLIF charge    spikingjelly.clock_driven.neuron (un-vendored; call sites
 *                            /root/reference/network/SNN_models.py:78..128,266..316, blocks.py:150,157)
 *      Heaviside fire        surrogate.ATan / surrogate.Sigmoid forward (train.py:118, blocks.py:142)
 *      hard reset            BaseNode.neuronal_reset, detach_reset=True at every call site
 *      skip / SEW add        SNN_models.py:171,176,181,186 (out_deconvK + out_conv(K-1)); blocks.py:171 (out += identity)
 *      firing-rate counts    SNN_models.py:219-242 (count_nonzero / numel)
 *      I-neuron read-out     SNN_models.py:150,172-188 (IFNode(v_threshold=inf) used as an accumulator)
 *
 * and the surrogate-gradient backward of the same chain (what torch autograd does op by op in the
 * reference).  The reference has no FFI for this path (it is 100 % Python); the Python
 * binding a maintainer would add is shown in INTEGRATION.md and shipped in stereospike_amd/_lib.py.
 *
 * Conventions
 *  - plain pointers and sizes; no allocation, no hidden global state; the caller owns every buffer.
 *  - every pointer is DEVICE memory (HBM) visible to the HIP device that `stream` belongs to.
 *  - sequences are laid out [T][N] contiguous, N = B*C*H*W of one layer (NCHW flattened), fp32.
 *  - calls are asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null stream).
 *  - return 0 on success, a negative errno-style code otherwise (never throws across the ABI).
 *  - deterministic: integer atomics for the counters, fixed-order two-pass reduction for g_k.
 *  - thread-safe for disjoint buffers.
 *  - rounding: every binary fp32 op rounds separately (no FMA contraction), in the order the eager
 *    PyTorch reference evaluates them, so spikes / h / v are bit-identical to the CPU reference.
 */
#ifndef SS_NEURON_H
#define SS_NEURON_H

#ifdef __cplusplus
extern "C" {
#endif

#define SS_OK        0
#define SS_EINVAL  (-22)   /* bad argument (null pointer, T<=0, unknown kind, misaligned buffer ...) */
#define SS_ELAUNCH  (-5)   /* the HIP runtime refused the launch (hipGetLastError != hipSuccess) */

/* neuron kind — which charge equation */
#define SS_KIND_IF    0    /* h = v + xs                                  neuron.IFNode            */
#define SS_KIND_LIF   1    /* h = v + (xs - (v - v_reset)) / tau          neuron.LIFNode (true division) */
#define SS_KIND_PLIF  2    /* h = v + (xs - (v - v_reset)) * k, k=sigmoid(w)   neuron.ParametricLIFNode */

/* surrogate gradient used by the backward */
#define SS_SG_ATAN     0   /* g * (1 / (1 + (pi/2*alpha*x)^2)) * (alpha/2)      surrogate.ATan    */
#define SS_SG_SIGMOID  1   /* g * (1 - s) * s * alpha, s = sigmoid(alpha*x)     surrogate.Sigmoid */

/* ABI version of this header; ss_abi_version() of the loaded library must match. */
#define SS_ABI_VERSION 10
int ss_abi_version(void);
/* first 16 hex digits of the sha256 over the sources the loaded library was built from (ABI 9; profiles/ evidence records it, bench.py checks it) */
const char* ss_source_hash(void);

/* Number of floats the caller must provide as `g_k_ws` (8-byte aligned) to the ss_neuron_bwd_* entry points when g_k != NULL.
 * ABI 5: the workspace holds the per-workgroup partials of the PLIF dL/dk sum as DOUBLES — that scalar is one heavily cancelling sum over
 * every neuron and step of a layer (condition number ~1e3 on the bottleneck layers), so it is accumulated in fp64 from the lane to the
 * last addition and rounded to fp32 once (g_k). */
long long ss_neuron_gk_ws_floats(void);

/*
 * Fused forward over T steps.  For n in [0,N), with v = v_init ? v_init[n] : v_reset:
 *   for t in 0..T-1:
 *      xs  = x_seq[t][n] * scale
 *      h   = charge(kind, v, xs)                      (see SS_KIND_*)
 *      z   = ((h - v_th) >= 0) ? 1.f : 0.f            (1 at exactly 0; v_th = +inf never fires)
 *      v   = (1.f - z) * h + z * v_reset              (hard reset, evaluated literally)
 *      out_seq[t][n] = skip_seq ? z + skip_seq[t][n] : z
 *      if (h_seq) h_seq[t][n] = h
 *   v_last[n] = v
 *   if (nnz) { nnz[0] += #{(t,n): z != 0};  nnz[1] += #{(t,n): out != 0} }      (64-bit integer atomics)
 *
 * x_seq is the conv output BEFORE MultiplyBy.  h_seq == NULL selects the inference variant (8 B/update
 * instead of 12).  `k` is a DEVICE pointer to one float, sigmoid(w), read only when kind == SS_KIND_PLIF
 * (so the learnable w never has to be synchronised to the host); `tau` is read only for SS_KIND_LIF.
 * out_seq may not alias x_seq; h_seq MAY alias x_seq (in-place h over the dead conv output).
 * Replaces: the Sequential(conv, MultiplyBy, Node) tails and the adds cited at the top of this file.
 */
int ss_neuron_fwd_f32(const float* x_seq, const float* v_init, const float* skip_seq,
                      float* out_seq, float* h_seq, float* v_last, unsigned long long* nnz,
                      int T, long long N,
                      float scale, int kind, float tau, const float* k,
                      float v_th, float v_reset,
                      void* stream);

/*
 * Surrogate-gradient backward of ss_neuron_fwd_f32 (reverse-t loop; what autograd does through
 * surrogate.*.backward, neuronal_reset, neuronal_charge and MultiplyBy in the reference).
 * With g_v = g_v_last ? g_v_last[n] : 0, for t = T-1..0:
 *      xh  = h_seq[t][n] - v_th;  z = (xh >= 0)
 *      g_s = g_out_seq[t][n]                       (+ g_v*v_reset - g_v*h  when detach_reset == 0)
 *      g_h = sg'(xh) * g_s + g_v * (1.f - z)
 *      IF  : g_x = g_h;        g_v = g_h
 *      LIF : g_x = g_h / tau;  g_v = g_h - g_x
 *      PLIF: g_x = g_h * k;    g_v = g_h - g_x;   acc_k += g_h * ((h - v_prev) / k)
 *      g_x_seq[t][n] = g_x * scale
 *   g_v_init[n] = g_v   (if non-NULL)
 * dL/d skip_seq is g_out_seq itself (identity) and is therefore not written.
 * g_k (PLIF, may be NULL): *g_k = sum over all (t,n) of acc_k, reduced in a fixed order (per-lane ->
 * wavefront -> workgroup -> second pass over workgroup partials held in g_k_ws) so repeated runs are
 * bit-identical.  v_init (the same pointer given to fwd, NULL => v_reset) is read only to form v_prev at t=0.
 */
int ss_neuron_bwd_f32(const float* g_out_seq, const float* g_v_last, const float* h_seq, const float* v_init,
                      float* g_x_seq, float* g_v_init, float* g_k, float* g_k_ws,
                      int T, long long N,
                      float scale, int kind, float tau, const float* k,
                      float v_th, float v_reset, int surrogate, float alpha, int detach_reset,
                      void* stream);

/*
 * The same backward WITHOUT the saved h_seq: h_t is recomputed inside the kernel from the layer input x_seq (the pointer given to
 * ss_neuron_fwd_f32, unmodified since) and v_init with the forward kernel's exact arithmetic, all T values of a lane living in
 * registers.  The forward is then called with h_seq = NULL and writes 4 B/update less (8 instead of 12); this launch reads x where
 * the other reads h (12 B/update either way).  Results are bit-identical to ss_neuron_bwd_f32.  Only the compile-time time-step
 * counts are supported: ss_neuron_bwd_rc_supported(T) != 0 (T in {1, 2, 4, 5, 8, 10}); otherwise SS_EINVAL — use the saved-h form.
 */
int ss_neuron_bwd_rc_supported(int T);
int ss_neuron_bwd_rc_f32(const float* g_out_seq, const float* g_v_last, const float* x_seq, const float* v_init,
                         float* g_x_seq, float* g_v_init, float* g_k, float* g_k_ws,
                         int T, long long N,
                         float scale, int kind, float tau, const float* k,
                         float v_th, float v_reset, int surrogate, float alpha, int detach_reset,
                         void* stream);

/*
 * The backward for an output with TWO consumers (the fork the reference's graph makes wherever a spike tensor feeds both the next
 * synapse and a skip / SEW add / prediction head: SNN_models.py:157-186, blocks.py:161-171).  PyTorch's autograd would sum the two
 * incoming gradients in a separate pass (12 B per element); here g_out2_seq (nullable) is added on load: g = g_out + g_out2 (one fp32
 * rounding, the same value the separate add produces) for 4 B per element.  Exactly one of h_seq (saved-h form) / x_seq (recompute
 * form, supported T only) is non-NULL; g_out2_seq requires the recompute form (else SS_EINVAL: add the gradients beforehand);
 * g_sum_seq (nullable, needs g_out2_seq): the summed gradient g is also written out — it is dL/dskip_seq of a stage that has both a
 * fused skip add and a forked output (the decoder stages).  Everything else as ss_neuron_bwd_f32.
 */
int ss_neuron_bwd_fork_f32(const float* g_out_seq, const float* g_out2_seq, float* g_sum_seq, const float* g_v_last, const float* h_seq, const float* x_seq,
                           const float* v_init, float* g_x_seq, float* g_v_init, float* g_k, float* g_k_ws,
                           int T, long long N,
                           float scale, int kind, float tau, const float* k,
                           float v_th, float v_reset, int surrogate, float alpha, int detach_reset,
                           void* stream);

/*
 * ABI 4.  ss_neuron_bwd_fork_f32 (recompute form) whose SECOND gradient arrives in low-rank form.  The second consumer of a decoder stage's
 * output — and, through the fused skip add, of the full-resolution encoder layer's — is a prediction head: a 3 x 3 synapse with ONE output
 * channel on the projected form (/root/reference/network/SNN_models.py:150-163, blocks.py:110-132), so its input gradient is
 * g2[t][pixel][c] = sum_{j < 9} lr_p[t][pixel][j] * lr_w[j][c]: 9 floats per pixel instead of C.  The kernel forms g2 in registers
 * (taps ascending; multiply and add rounded separately) and adds it on load: g = g_out + g2, or g = g2 when g_out_seq is NULL — the
 * head's data-gradient GEMM and its C-channel output never exist, and this launch reads 36 / C B per update instead of 4 for it.
 *   lr_p [T, N / C, lr_rank], lr_w [lr_rank, C] (16-B aligned), lr_rank == 9, C = channels of the NHWC layer (C <= 512, (256 * 4) % C == 0, N % C == 0);
 *   g_sum_seq (nullable, needs g_out_seq): g written out, as in ss_neuron_bwd_fork_f32.
 * SS_EINVAL when ss_neuron_bwd_fork_lr_supported(T, N, C, lr_rank) is 0 or a buffer is not 16-B aligned.
 */
int ss_neuron_bwd_fork_lr_supported(int T, long long N, int C, int lr_rank);
int ss_neuron_bwd_fork_lr_f32(const float* g_out_seq, const float* lr_p, const float* lr_w, int lr_rank, int C, float* g_sum_seq,
                              const float* g_v_last, const float* x_seq, const float* v_init, float* g_x_seq, float* g_v_init,
                              float* g_k, float* g_k_ws, int T, long long N,
                              float scale, int kind, float tau, const float* k,
                              float v_th, float v_reset, int surrogate, float alpha, int detach_reset,
                              void* stream);

/*
 * 16-bit activation I/O variants (BASELINE.json configs 2 and 5: bf16 / fp16 activations, fp32 membrane state).
 * x_seq, skip_seq, out_seq (and g_out_seq, g_x_seq) hold IEEE fp16 (dtype = SS_DT_F16) or bfloat16 (SS_DT_BF16) values;
 * every input is widened to fp32 on load, ALL arithmetic and the membrane (v_init, v_last, h_seq, g_v_*) stay fp32 exactly
 * as in the f32 entry points, results are rounded to nearest-even on store (spikes 0/1/2/3 are exact in both formats).
 * h_seq stays fp32 so the backward recomputes the very spike mask the forward produced.
 * Algorithmic bytes per update: 8 B forward-train (2 + 2 + 4), 4 B inference, 8 B backward (2 + 4 + 2).
 */
#define SS_DT_F16   1
#define SS_DT_BF16  2
int ss_neuron_fwd_x16(const void* x_seq, const float* v_init, const void* skip_seq,
                      void* out_seq, float* h_seq, float* v_last, unsigned long long* nnz,
                      int T, long long N, float scale, int kind, float tau, const float* k,
                      float v_th, float v_reset, int dtype, void* stream);
int ss_neuron_bwd_x16(const void* g_out_seq, const float* g_v_last, const float* h_seq, const float* v_init,
                      void* g_x_seq, float* g_v_init, float* g_k, float* g_k_ws,
                      int T, long long N, float scale, int kind, float tau, const float* k,
                      float v_th, float v_reset, int surrogate, float alpha, int detach_reset, int dtype, void* stream);

/* ss_neuron_bwd_rc_f32 for 16-bit activations: x_seq is the fp16 / bf16 layer input given to ss_neuron_fwd_x16 (called with
 * h_seq = NULL: 4 B/update instead of 8); this launch moves 6 B/update instead of 8.  g_x / g_v_init bit-identical to
 * ss_neuron_bwd_x16; dL/dk sums the same terms in another fp32 order (a lane owns 4 or 2 neurons here, 8 there). */
int ss_neuron_bwd_rc_x16(const void* g_out_seq, const float* g_v_last, const void* x_seq, const float* v_init,
                         void* g_x_seq, float* g_v_init, float* g_k, float* g_k_ws,
                         int T, long long N,
                         float scale, int kind, float tau, const float* k,
                         float v_th, float v_reset, int surrogate, float alpha, int detach_reset,
                         int dtype, void* stream);

/*
 * ss_neuron_bwd_fork_f32 for 16-bit activations (recompute form only): g_out_seq and g_out2_seq are the 16-bit gradients of the two
 * consumers of out_seq; the kernel widens both and adds them in fp32 (no 16-bit rounding of the sum, unlike autograd's accumulation
 * kernel); g_sum_seq (nullable) receives that sum narrowed once = dL/dskip_seq of a stage with a fused skip add.
 */
int ss_neuron_bwd_fork_x16(const void* g_out_seq, const void* g_out2_seq, void* g_sum_seq, const float* g_v_last, const void* x_seq,
                           const float* v_init, void* g_x_seq, float* g_v_init, float* g_k, float* g_k_ws,
                           int T, long long N, float scale, int kind, float tau, const float* k,
                           float v_th, float v_reset, int surrogate, float alpha, int detach_reset, int dtype, void* stream);

/*
 * Extended forward (descriptor form; ABI 2).  Same arithmetic as ss_neuron_fwd_f32 / ss_neuron_fwd_x16 plus
 *
 *  - 2-BIT PACKED SPIKE TENSORS (the reference's spike tensors take the values 0..3 only:
 *    /root/reference/network/SNN_models.py:171-192, blocks.py:171).  Layout: [T][N/16] 32-bit words, neuron n of step t in bits
 *    2*(n%16)+{0,1} of word t*(N/16) + n/16, code = out value (z + skip).  out_packed != NULL writes the output packed; out_seq may then
 *    be NULL (forward 4.25 B/update instead of 8: read x 4 + write 0.25) or also be written for a consumer that needs fp32.
 *    skip_packed != NULL reads the skip operand from a packed tensor (0.25 B/update instead of 4).  fp32 activations, compile-time
 *    T (ss_neuron_bwd_rc_supported(T)), N % 16 == 0, h_seq == NULL (training recomputes h: ss_neuron_bwd_rc_f32); else SS_EINVAL.
 *    CONTRACT: z + skip <= 3 (the reference's largest value; a dense skip_seq is read modulo 4 and an out code above 3 saturates at 3 —
 *    a 2-bit field never spills into its neighbour, but such an output is not the sum any more: keep wider skips on the dense form).
 *    Readers for the consumers: ss_unpack_spikes, ss_im2col_cl_bf16_packed, and skip_packed itself.
 *  - FIRING-RATE COUNTERS WITHOUT SAME-ADDRESS ATOMICS: with cnt_ws (ss_neuron_cnt_ws_words(N) 32-bit words) every workgroup stores
 *    its two partial counts and a one-workgroup second pass adds their sum to nnz[0..1]; the launch keeps its full grid (with
 *    cnt_ws == NULL the atomics form bounds the grid to 2048 workgroups).  Integer sums: deterministic either way.
 *
 *  - v_last == NULL (ABI 10), packed forms only (out_packed or skip_packed given): the membrane after step T is not written — 4 B per neuron less, 10 - 36 % of
 *    what a packed-only forward moves.  The reference resets every membrane before the next pass (/root/reference/train.py:221); a caller that needs it
 *    later runs the same call again on the layer input it keeps for ss_neuron_bwd_rc_* (stereospike_amd.fused.membrane_after).
 *
 * act_dtype: 0 = fp32 activations (x_seq / skip_seq / out_seq are float*), SS_DT_F16 / SS_DT_BF16 = 16-bit activations (no packed I/O).
 * `size` must be sizeof(ss_neuron_fwd_desc) (lets the struct grow without breaking old callers).
 */
typedef struct ss_neuron_fwd_desc {
    unsigned int size;
    int act_dtype;
    const void* x_seq;
    const float* v_init;                 /* nullable => v_reset */
    const void* skip_seq;                /* nullable; dense skip operand (act_dtype elements) */
    const unsigned int* skip_packed;     /* nullable; packed skip operand (exclusive with skip_seq) */
    void* out_seq;                       /* nullable when out_packed is given */
    unsigned int* out_packed;            /* nullable */
    float* h_seq;                        /* nullable */
    float* v_last;                       /* nullable in the packed forms (ABI 10) */
    unsigned long long* nnz;             /* nullable, [2] */
    unsigned int* cnt_ws;                /* nullable; requires nnz */
    int T;
    long long N;
    float scale;
    int kind;
    float tau;
    const float* k;
    float v_th, v_reset;
} ss_neuron_fwd_desc;
int ss_neuron_fwd_ex(const ss_neuron_fwd_desc* desc, void* stream);
long long ss_neuron_cnt_ws_words(long long N);

/* packed [n/16 words] -> dense values 0..3 as fp32 (out_dtype 0), fp16 or bf16.  copies > 1 (K-concatenated exact bf16x3 GEMM operand
 * [X X X]): the n values are rows of row_len elements and every row is written `copies` times back to back (output rows are
 * copies*row_len long).  n % 16 == 0; with copies > 1: row_len % 8 == 0 and n % row_len == 0. */
int ss_unpack_spikes(const unsigned int* packed, void* out, long long n, int out_dtype, int row_len, int copies, void* stream);
/* ss_im2col_cl_bf16 with the NHWC input given as a packed spike tensor (C % 8 == 0, NB*h*w*C % 16 == 0). */
int ss_im2col_cl_bf16_packed(const unsigned int* x_packed, void* A, long long NB, int h, int w, int C, int k, int stride, int pad,
                             int ho, int wo, void* stream);

/* (ABI 9: the fused projection + gather kernels of rounds 2 - 3 — ss_upconv_fused_*, ss_upconv_fused2_* — and the fused adjoint kernels ss_upconv_bwd_fused_f32 /
 * ss_upconv_bwd_dgrad_f32 are GONE: the sub-pixel forward (ss_upconv_sub_*) and the box-sum backward (ss_upconv_box_*) replaced them on every stage they ran on;
 * geometries those do not take run the projection GEMM + ss_upconv_cl_* gather forms.) */

/*
 * Weight gradient of a synapse on spike inputs as an exact bf16x3 MFMA contraction over the rows (ABI 3) — the decoder's
 * g_W = x^T @ g_P (autograd of NNConvUpsampling, /root/reference/network/blocks.py:110-132; call sites SNN_models.py:110-129):
 *     g_w[ci][n] (+)= sum_r x[r][ci] * g[r][n]        x [R][C_in] fp32 spike counts (exact in bf16), g [R][N] fp32, g_w [C_in][N]
 * g is split exactly into three bf16 terms in registers, products are exact, accumulation fp32 (v_mfma_f32_32x32x16_bf16); split-K over
 * the rows with a fixed-order second pass (deterministic).  ss_spike_wgrad_supported(C_in, N): C_in in {64, 128, 256, 512}, N % 32 == 0.
 * ws: caller workspace of ss_spike_wgrad_ws_floats(C_in, N, R) floats (16-byte aligned: split-K partials + the spike operand transposed into
 * MFMA fragment order as bf16).  accumulate != 0 adds to g_w.
 */
int ss_spike_wgrad_supported(int Cin, int N);
long long ss_spike_wgrad_ws_floats(int Cin, int N, long long R);
int ss_spike_wgrad_f32(const float* g, const float* x, float* g_w, float* ws, long long R, int Cin, int N, int accumulate, void* stream);

/*
 * ABI 7 — the decoder's backward on the BOX-SUM image (ss_upconv_box.hip; round 4).  Replaces, for one decoder stage, the whole autograd backward of
 * NNConvUpsampling (/root/reference/network/blocks.py:110-132; call sites SNN_models.py:110-129) — the adjoint gather ss_upconv_cl_bwd_f32 and both
 * contractions on its per-tap tensor g_P (ss_gemm6_f32 / ss_spike_wgrad_f32):
 *
 *   ss_upconv_boxsum_f32    B[nb][j][i][co] = sum_{Y in VR[j]} ( sum_{X in HR[i]} g_out[nb][Y][X][co] ), rows top to bottom, columns left to right, every
 *                           sum started from +0 (ss_upconv_cl_bwd_f32's order: g_P[nb][iy][ix][ky,kx][co] == B[nb][vmap[iy][ky]][hmap[ix][kx]][co] bit for bit),
 *                           stored as three bf16 planes h + m + l == B (round-to-nearest split): box [NB][C_out / 8][3][NVR][NHR][8] bf16,
 *                           ss_upconv_box_elems(NB, C_out, NVR, NHR) elements.  vr [NVR][2], hr [NHR][2]: (start, length) of the distinct vertical /
 *                           horizontal output ranges, id 0 = the empty range.
 *   ss_upconv_box_dgrad_f32 g_x[nb][iy][ix][ci] = sum_{ky,kx,co} B[nb][vmap[iy][ky]][hmap[ix][kx]][co] * weight[co][ci][ky][kx]: six bf16 cross terms on the
 *                           matrix cores, |g_x - float64| <= 2^-20 sum |B| |W| element-wise (2^-21 typical: ss_gemm6_f32's accuracy).  vmap [h][5], hmap [w][5]: range ids;
 *                           tile_rows [n_row_tiles][4]: (first source row, rows <= 4, first id, id count <= 15) — the caller cuts the source rows into tiles whose
 *                           non-empty vertical ranges fit the on-chip window (a triple-replicated row shortens its tile); tile_cols [ceil(w / 32)][2]: (first id,
 *                           id count <= 76) of 32 consecutive source columns; ws: ss_upconv_box_dgrad_ws_floats(C_in, C_out) floats.
 *   ss_upconv_box_wgrad_f32 g_w[co][ci][ky][kx] (+)= sum_{nb,iy,ix} x[nb][iy][ix][ci] * B[nb][vmap[iy][ky]][hmap[ix][kx]][co]: x a spike tensor (values exact
 *                           in bf16; fp32 NHWC or the 2-bit packed form), every product exact, fp32 accumulation, fixed-order reduction (deterministic):
 *                           |g_w - float64| <= 2^-20 sum |x| |B| (fp32 accumulation over up to 1.8 M source pixels; measured <= 2^-21).  ws: ss_upconv_box_wgrad_ws_floats(C_in, C_out, NB, h, w) floats.
 * g_out [NB][H][W][C_out], x / g_x [NB][h][w][C_in] NHWC fp32; weight / g_w [C_out][C_in][5][5] (the Conv2d parameter's own layout).
 * *_supported: k = 5; dgrad C_in % 64 == 0, C_out % 8 == 0; wgrad C_in % 32 == 0, C_out % 8 == 0; largest id span of tile_rows <= 15 and of tile_cols <= 76
 * (ss_upconv_box_window).  No g_P anywhere; HBM traffic of the three launches: g_out once, the box image (1.5 x g_out's bytes) written
 * once and read ~twice (window halos), x once per 8 output channels in its 2-byte fragment form, g_x once.
 */
long long ss_upconv_box_elems(long long NB, int Cout, int NVR, int NHR);
int ss_upconv_boxsum_f32(const float* g_out, const int* vr, const int* hr, void* box, long long NB, int Cout, int H, int W, int NVR, int NHR, void* stream);
int ss_upconv_box_window(int* max_tile_rows, int* max_cols32);   /* returns the most source rows of a row tile (4); the id spans a tile may reach */
int ss_upconv_box_dgrad_supported(int Cin, int Cout, int k, int max_tile_rows, int max_cols32);
/* 1 when a map of n_row_tiles row tiles x w source columns x `pixels` = NB*h*w source pixels fits the kernels' on-chip tile tables (<= 64 row tiles,
   <= 16 column tiles of 32) and 32-bit pixel indices — the launch entry points refuse anything else with SS_EINVAL (ABI 9) */
int ss_upconv_box_tiles_supported(int n_row_tiles, int w, long long pixels);
long long ss_upconv_box_dgrad_ws_floats(int Cin, int Cout);
int ss_upconv_box_dgrad_f32(const void* box, const float* weight, const int* vmap, const int* hmap, const int* tile_rows, int n_row_tiles, const int* tile_cols,
                            float* g_x, float* ws, long long NB, int Cin, int Cout, int h, int w, int NVR, int NHR, void* stream);
int ss_upconv_box_wgrad_supported(int Cin, int Cout, int k, int max_tile_rows, int max_cols32);
long long ss_upconv_box_wgrad_ws_floats(int Cin, int Cout, long long NB, int h, int w);
int ss_upconv_box_wgrad_f32(const void* box, const float* x, const unsigned int* x_packed, const int* vmap, const int* hmap, const int* tile_rows, int n_row_tiles,
                            const int* tile_cols, float* g_w, float* ws, long long NB, int Cin, int Cout, int h, int w, int NVR, int NHR, int accumulate,
                            void* stream);

/*
 * Dense x dense fp32 GEMM on the bf16 matrix cores with six cross terms (ABI 3) — the decoder's data gradient g_x = g_P @ W2 (autograd of
 * NNConvUpsampling, /root/reference/network/blocks.py:110-132): C [R][N] = A [R][K] @ B [K][N], all fp32 row-major.  Both operands are split
 * exactly into three bf16 terms (round to nearest); the six products ah bh, ah bm, am bh, ah bl, am bm, al bh are kept (each exact), fp32
 * accumulation: |C - exact| <= 2^-21 sum_k |a||b| (the dropped terms amount to one fp32 product rounding).  K % 16 == 0, N in {64, 128,
 * 256, 512}; ws: ss_gemm6_ws_floats(K, N) floats (B's terms in fragment order), 16-byte aligned.
 */
int ss_gemm6_supported(int K, int N);
long long ss_gemm6_ws_floats(int K, int N);
int ss_gemm6_f32(const float* A, const float* B, float* C, float* ws, long long R, int K, int N, void* stream);
/* batch independent products C[b] = A[b] @ B[b] (contiguous [batch][R][K], [batch][K][N], [batch][R][N]; ws: batch * ss_gemm6_ws_floats) — the 16
 * transform-domain products of the Winograd data gradient. */
int ss_gemm6_batched_f32(const float* A, const float* B, float* C, float* ws, int batch, long long R, int K, int N, void* stream);

/*
 * Weight gradient of a 5x5 / stride 2 / pad 2 convolution on spike inputs (ABI 3) — autograd of the reference's conv1 / conv2
 * (nn.Conv2d(32, 64, 5, 2, 2), nn.Conv2d(64, 128, 5, 2, 2): /root/reference/network/SNN_models.py:80-90) w.r.t. their weight:
 *     g_w[co][ci][ky][kx] (+)= sum_{nb, oy, ox} g[nb][oy][ox][co] * x[nb][2 oy + ky - 2][2 ox + kx - 2][ci]
 * g [NB][ho][wo][C_out] fp32 (NHWC), x [NB][h][w][C_in] fp32 spike counts (exact in bf16), g_w [C_out][C_in][5][5].  g is split exactly into
 * three bf16 terms in registers; products exact; fp32 accumulation on v_mfma_f32_32x32x16_bf16; split-K with a fixed-order second pass.
 * ws: ss_spike_conv_wgrad_ws_floats floats (partials + five column-decimated bf16 copies of x), 16-byte aligned.
 */
int ss_spike_conv_wgrad_supported(int Cin, int Cout, int k, int stride, int pad);
long long ss_spike_conv_wgrad_ws_floats(int Cin, int Cout, long long NB, int h, int w);
/* ABI 10: the workspace the packed-input form (x_packed given) needs — its partial sums only; 0 when that form does not apply (then the figure above) */
long long ss_spike_conv_wgrad_tr_ws_floats(int Cin, int Cout);
int ss_spike_conv_wgrad_f32(const float* g, const float* x, const unsigned int* x_packed, float* g_w, float* ws, long long NB, int Cin, int Cout, int h,
                            int w, int accumulate, void* stream);     /* x_packed != NULL: the input as a 2-bit packed spike tensor (x may be NULL) */

/*
 * ABI 5 — FORWARD of the stride-2 5x5 encoder convolutions on spike inputs as an exact bf16x3 implicit GEMM on the matrix cores
 * (ss_spike_conv.hip).  Replaces torch.nn.functional.conv2d (MIOpen fp32) for conv1 / conv2 of the reference's encoder
 * (/root/reference/network/SNN_models.py:80-90: nn.Conv2d(C, 2C, kernel_size=5, stride=2, padding=2, bias=False) on the previous stage's spikes):
 *   out[nb][oy][ox][co] = sum_{ky,kx,ci} x[nb][2 oy + ky - 2][2 ox + kx - 2][ci] * weight[co][ci][ky][kx]       (zero padding)
 * x [NB][h][w][C_in] NHWC: a dense fp32 spike tensor (values exact in bf16) or, when x_packed != NULL, the 2-bit packed form (x may then be
 * NULL); weight [C_out][C_in][5][5] fp32 (split exactly into three bf16 terms: every product exact, fp32 accumulation — fp32-convolution
 * accuracy); out [NB][ho][wo][C_out] fp32, ho = (h - 1) / 2 + 1.  ws: ss_spike_conv_fwd_ws_floats floats.  Compiled shapes: (32 -> 64), (64 -> 128).
 */
int ss_spike_conv_fwd_supported(int Cin, int Cout, int k, int stride, int pad);
/* (ABI 8) the wide shapes of conv3 / conv4 (128 -> 256, 256 -> 512; /root/reference/network/SNN_models.py:91-101) that ss_spike_conv_fwd_f32 ALSO accepts,
 * in output-channel slices of 128 per workgroup: an A/B against the im2col + library-GEMM path, which the network keeps (profiles/r04/conv34_ab.log) */
int ss_spike_conv_fwd_wide_supported(int Cin, int Cout, int k, int stride, int pad);
long long ss_spike_conv_fwd_ws_floats(int Cin, int Cout);
int ss_spike_conv_fwd_f32(const float* x, const unsigned int* x_packed, const float* weight, float* out, float* ws,
                          long long NB, int Cin, int Cout, int h, int w, void* stream);
/* The FIRST encoder layer, nn.Conv2d(C_in = 4 | 2, 32, kernel_size=5, stride=1, padding=2, bias=False) on the event-voxel input
 * (/root/reference/network/SNN_models.py:75-79, 450-454): x [NB][h][w][C_in] dense fp32 (ANY values: both operands are split into three bf16
 * terms, six cross terms kept — exact for integer event counts, fp32-product accuracy otherwise), weight [32][C_in][5][5], out [NB][h][w][32]. */
int ss_dense_conv_s1_fwd_supported(int Cin, int Cout, int k, int stride, int pad);
int ss_dense_conv_s1_fwd_f32(const float* x, const float* weight, float* out, long long NB, int Cin, int Cout, int h, int w, void* stream);
/* ABI 6 — its WEIGHT gradient (autograd in the reference): g_w[32][C_in][5][5] (+)= sum over pixels of g[nb][y][x][co] * x[nb][y + ky - 2][x + kx - 2][ci],
 * g [NB][h][w][32], x [NB][h][w][C_in] dense fp32 NHWC (ANY values: six bf16 cross terms per product on the matrix cores, fp32 accumulation per tile,
 * fixed-order fp64 second pass — deterministic).  ws: ss_dense_conv_s1_wgrad_ws_floats floats, 16-byte aligned.  Replaces MIOpen's igemm_wrw — the
 * last MIOpen call of the default training step. */
int ss_dense_conv_s1_wgrad_supported(int Cin, int Cout, int k, int stride, int pad);
long long ss_dense_conv_s1_wgrad_ws_floats(int Cin);
int ss_dense_conv_s1_wgrad_f32(const float* g, const float* x, float* g_w, float* ws, long long NB, int Cin, int Cout, int h, int w, int accumulate, void* stream);

/*
 * ABI 6 — DATA gradient of the stride-2 5x5 encoder convolutions as a six-term bf16 implicit GEMM on the matrix cores (ss_conv_dgrad.hip).
 * Replaces torch's convolution_backward (MIOpen fp32 igemm_bwd) w.r.t. the INPUT of conv1 .. conv4 of the reference's encoder
 * (/root/reference/network/SNN_models.py:80-101: nn.Conv2d(C, 2C, kernel_size=5, stride=2, padding=2, bias=False); autograd in the reference):
 *   g_x[nb][iy][ix][ci] = sum_{ky,kx,co} g[nb][(iy + 2 - ky) / 2][(ix + 2 - kx) / 2][co] * weight[co][ci][ky][kx]
 *                         over the taps with iy + 2 - ky, ix + 2 - kx even and inside the [ho][wo] map, ho = (h - 1) / 2 + 1
 * g [NB][ho][wo][C_out] fp32 NHWC (ANY values), weight [C_out][C_in][5][5] fp32, g_x [NB][h][w][C_in] fp32 NHWC — every element written.
 * Both operands are split into three bf16 terms, six cross terms kept: |g_x - float64| <= 2^-21 sum |g||w| element-wise (fp32-product accuracy,
 * fp32 accumulation); deterministic (no atomics, fixed summation order).  ws: ss_conv_s2_dgrad_ws_floats floats, 16-byte aligned.
 * Compiled shapes: C_in in {32, 64, 128, 256}, C_out = 2 C_in.
 */
int ss_conv_s2_dgrad_supported(int Cin, int Cout, int k, int stride, int pad);
long long ss_conv_s2_dgrad_ws_floats(int Cin, int Cout);
int ss_conv_s2_dgrad_f32(const float* g, const float* weight, float* g_x, float* ws, long long NB, int Cin, int Cout, int h, int w, void* stream);

/*
 * ABI 6 — the full-resolution prediction head on 2-bit packed spikes (ss_head.hip).  predict_depth1 = NNConvUpsampling(32, 1, kernel_size=3)
 * (/root/reference/network/SNN_models.py:133-148, called at :186-188; blocks.py:110-132) in the projected form of ss_upconv_cl_fwd_f32:
 *   ss_head_proj_packed_f32 : P[row][tap] = sum_c x[row][c] * Wt[c][tap]   — replaces the library GEMM [rows, C] x [C, 9] on the DENSE spike tensor;
 *   ss_head_wgrad_packed_f32: g_Wt[c][tap] (+)= sum_row x[row][c] * g_P[row][tap] — replaces the library split-K GEMM (autograd of the head w.r.t.
 *                             its weight), deterministic (per-wavefront partials, fixed-order fp64 second pass).
 * x_packed: the head's input [rows][C] (rows = NB * h * w source pixels, NHWC) as 2-bit codes 0 .. 3, 16 per 32-bit word (the out_packed form of
 * ss_neuron_fwd_ex); Wt [C][9] fp32 (split exactly into three bf16 terms: every product exact, fp32 accumulation on the matrix cores);
 * P, g_P [rows][9] fp32.  ws: ss_head_wgrad_packed_ws_floats floats.  Compiled: C in {32, 64}, one output channel, 3 x 3 taps.
 */
int ss_head_packed_supported(int Cin, int Cout, int k);
long long ss_head_wgrad_packed_ws_floats(int Cin);
int ss_head_proj_packed_f32(const unsigned int* x_packed, const float* Wt, float* P, long long rows, int Cin, void* stream);
int ss_head_wgrad_packed_f32(const unsigned int* x_packed, const float* g_P, float* g_Wt, float* ws, long long rows, int Cin, int accumulate, void* stream);

/*
 * Winograd F(2x2, 3x3) data gradient of a 3x3 / stride 1 / pad 1 convolution in NHWC — the backward of SEWResBlock's conv1 / conv2
 * w.r.t. their input (/root/reference/network/blocks.py:146-159; autograd's conv backward in the reference), ABI 3:
 *     g_in[nb][y][x][ci] = sum_{co, a, b} g[nb][y + a - 1][x + b - 1][co] * W[co][ci][2 - a][2 - b]
 * in three hand-written transform kernels around ONE batched fp32 GEMM that the caller runs on the library:
 *   ss_wino_dgrad_weights_f32: W [C_out][C_in][3][3] -> U [16][C_out][C_in]          (U_k = G Wf G^T, redo after every weight update)
 *   ss_wino_dgrad_input_f32  : g [NB][H][W][C_out]   -> V [16][T][C_out]             (V = B^T d B; T = NB * ceil(H/2) * ceil(W/2) tiles)
 *   caller                   : M[k] = V[k] @ U[k]      [16][T][C_in]
 *   ss_wino_dgrad_output_f32 : M [16][T][C_in]       -> g_in [NB][H][W][C_in]        (Y = A^T M A)
 * 2.25x fewer multiplications than the direct contraction; fp32, every op rounds once; C % 4 == 0, 16-byte aligned buffers.
 */
int ss_wino_dgrad_weights_f32(const float* W, float* U, int Cout, int Cin, void* stream);
int ss_wino_dgrad_input_f32(const float* g, float* V, long long NB, int H, int W, int C, void* stream);
int ss_wino_dgrad_output_f32(const float* M, float* g_in, long long NB, int H, int W, int C, void* stream);

/*
 * I-neuron read-out pool (SNN_models.py:150,172-188; ANN_models.py:111,130-146): one shared non-firing
 * IF membrane that the K predict_depth heads charge in the order K-1..0 of the reference's forward
 * (predict_depth4 first), every time step.  pd_seq element (t,k,m) is at pd_seq[t*stride_t + k*stride_k + m],
 * k = 0 is the head charged FIRST.  With v = v_init ? v_init[m] : v_reset:
 *   for t: for k:  h = v + pd(t,k,m) * scale;  v = (1.f - 0.f) * h + 0.f * v_reset;  depth_seq[t][k][m] = v
 * depth_seq is [T][K][M] contiguous; depth_seq[T-1][K-1] is the pool's final membrane.
 * The additions happen in exactly this (t outer, k inner) order — it is part of fp32 parity.
 */
int ss_ipool_fwd_f32(const float* pd_seq, long long stride_t, long long stride_k, const float* v_init,
                     float* depth_seq, int T, int K, long long M,
                     float scale, float v_reset, void* stream);

/*
 * Backward of the pool: g_pd(t,k,m) = scale * sum of g_depth_seq over every (t',k') at or after (t,k)
 * in charge order (+ g_v_last), accumulated from the last charge backwards; g_v_init[m] = the full sum.
 * g_pd_seq uses the same strides as pd_seq.
 */
int ss_ipool_bwd_f32(const float* g_depth_seq, const float* g_v_last,
                     float* g_pd_seq, long long stride_t, long long stride_k, float* g_v_init,
                     int T, int K, long long M, float scale, void* stream);

/*
 * predict_depth head synapse (SNN_models.py:133-148): NNConvUpsampling(C -> 1, k, up_size=(H,W), bias=True), i.e.
 * UpsamplingNearest2d(size=(H+k-1, W+k-1)) followed by a valid k x k conv to ONE channel (blocks.py:124-128).
 * Because the output has a single channel, the channel contraction commutes with the resize: with the k*k per-tap
 * projections of the LOW-RES map  P[nb][tap][iy][ix] = sum_c w[c][tap] * in[nb][c][iy][ix]  (a 1x1 conv, done by the
 * caller), the head is the gather
 *      out[nb][y][x] = bias + sum_{ky,kx} P[nb][ky*k+kx][ src_y[y+ky] ][ src_x[x+kx] ]
 * where src_y / src_x are the nearest-neighbour source indices of the up-sampled rows / columns (int32 tables of
 * H+k-1 / W+k-1 entries, computed by the caller exactly as torch's UpsamplingNearest2d does).  The (H+k-1)x(W+k-1)xC
 * up-sampled tensor (93 MB per sample-step for predict_depth4) is never materialised.  bias: device pointer to one
 * float, or NULL.  Taps are summed in (ky, kx) order, bias last.
 */
int ss_upconv1_fwd_f32(const float* P, const int* src_y, const int* src_x, const float* bias, float* out,
                       long long NB, int k, int h, int w, int H, int W, void* stream);

/*
 * Adjoint of the gather:  g_P[nb][tap][iy][ix] = sum of g_out[nb][Y-ky][X-kx] over the up-sampled rows Y in
 * [y_lo[iy], y_hi[iy]) and columns X in [x_lo[ix], x_hi[ix]) that map to (iy, ix), restricted to 0 <= Y-ky < H,
 * 0 <= X-kx < W.  Each g_P element is produced by exactly one lane in a fixed order (no atomics => deterministic).
 * y_lo/y_hi: int32 [h]; x_lo/x_hi: int32 [w].  (dL/dbias = sum(g_out) is left to the caller.)
 */
int ss_upconv1_bwd_f32(const float* g_out, const int* y_lo, const int* y_hi, const int* x_lo, const int* x_hi,
                       float* g_P, long long NB, int k, int h, int w, int H, int W, void* stream);

/*
 * Channels-last variants of the two gather kernels, for C_out > 1 stages kept in NHWC memory end to end:
 *   P     [NB][h][w][k*k*C]   channel index = tap*C + c   (exactly the row-major result of the single GEMM
 *                              x[NB*h*w, C_in] @ W[C_in, k*k*C]: no per-image batching, no layout transposes)
 *   out   [NB][H][W][C]
 *   out[nb][y][x][c] = (bias ? bias[c] : 0) + sum_{ky,kx} P[nb][src_y[y+ky]][src_x[x+kx]][(ky*k+kx)*C + c]   (taps in (ky,kx) order,
 *   bias last).  A lane owns 4 consecutive channels of one pixel (16-B accesses, consecutive lanes = consecutive
 *   channels then consecutive pixels => coalesced reads of P and writes of out).  C % 4 != 0 falls back to one channel per lane.
 * The adjoint g_P[nb][iy][ix][tap*C + c] = rectangle sum of g_out[nb][.][.][c] (rows first, as ss_upconv1_bwd_f32).
 */
int ss_upconv_cl_fwd_f32(const float* P, const int* src_y, const int* src_x, const float* bias, float* out,
                         long long NB, int k, int C, int h, int w, int H, int W, void* stream);
int ss_upconv_cl_bwd_f32(const float* g_out, const int* y_lo, const int* y_hi, const int* x_lo, const int* x_hi,
                         float* g_P, long long NB, int k, int C, int h, int w, int H, int W, void* stream);

/* The same pair for the 16-bit activation modes (decoder stages, k = 5): the stage output `out` (forward) and its gradient `g_out`
 * (adjoint) are fp16 / bf16 in HBM (dtype = SS_DT_F16 / SS_DT_BF16) — narrowed to nearest-even on store, widened on load — so the
 * following neuron layer runs its x16 kernels; P, g_P, the bias and every sum stay fp32. */
int ss_upconv_cl_fwd_x16(const float* P, const int* src_y, const int* src_x, const float* bias, void* out,
                         long long NB, int k, int C, int h, int w, int H, int W, int dtype, void* stream);
int ss_upconv_cl_bwd_x16(const void* g_out, const int* y_lo, const int* y_hi, const int* x_lo, const int* x_hi,
                         float* g_P, long long NB, int k, int C, int h, int w, int H, int W, int dtype, void* stream);

/* Adjoint gather writing g_P directly as bf16 (nearest-even narrowing of the fp32 sums) — for the 16-bit modes, whose backward GEMMs
 * take bf16 operands: no fp32 g_P round trip and no separate cast pass.  g_dtype: 0 = fp32 g_out, SS_DT_F16 / SS_DT_BF16 = 16-bit. */
int ss_upconv_cl_bwd_lowp(const void* g_out, int g_dtype, const int* y_lo, const int* y_hi, const int* x_lo, const int* x_hi,
                          void* g_P_bf16, long long NB, int k, int C, int h, int w, int H, int W, void* stream);
/* ABI 10 (round 6): the same with g_P in `gp_dtype` — SS_DT_BF16 (== ss_upconv_cl_bwd_lowp) or, for an fp16 g_out, SS_DT_F16: the fp16 activation mode's g_P forms
 * (deconv3 / deconv4 backward) then run their two GEMMs on fp16 operands — the mode's own weight rounding, the data gradient written in fp16 by the GEMM's
 * epilogue — instead of bf16 operands and an fp32 -> fp16 conversion pass over the result. */
int ss_upconv_cl_bwd_lowp_dt(const void* g_out, int g_dtype, const int* y_lo, const int* y_hi, const int* x_lo, const int* x_hi,
                             void* g_P, int gp_dtype, long long NB, int k, int C, int h, int w, int H, int W, void* stream);

/*
 * Voxeliser — the step BEFORE the path: events -> per-pixel two-polarity count frames,
 * /root/reference/datasets/MVSEC/utils.py:215-281 (mvsecCumulateSpikesIntoFrames; a python loop per event there).
 *   events : [E][4] float64 row-major (X, Y, TIME, POLARITY), exactly the array the reference carries around
 *   start, end : float64 [G] open-interval bounds of the G = n_chunks * frames_per_depth_map frames, evaluated by the
 *                caller in the reference's own float64 expression order (utils.py:259-260) so boundary events match
 *   counts : uint32 [G][2][H][W], zeroed here; counts[g][POLARITY == 1 ? 0 : 1][(int)Y][(int)X] += 1 for every event with
 *            start[g] < TIME - events[0].TIME < end[g]   (strict on both sides; a time inside two overlapping intervals
 *            counts twice, as in the reference).  Events whose truncated coordinates fall outside H x W are skipped
 *            (the reference would raise IndexError: its rectifier keeps x == 346 / y == 260, utils.py:52-55).
 * One lane per event, binary search over the bounds, integer atomics => deterministic.
 */
int ss_voxelize_f64(const double* events, long long E, const double* start, const double* end, int G,
                    unsigned int* counts, int H, int W, void* stream);

/*
 * Operand preparation for the exact bf16x3 GEMM form of the encoder / bottleneck convolutions whose input is a spike tensor
 * (/root/reference/network/SNN_models.py:83-101: conv2..conv4, blocks.py:146-159: the SEW bottleneck convs): the convolution itself is a
 * library GEMM (PyTorch-ROCm / hipBLASLt, bf16 operands, fp32 accumulate); these two entry points build its operands.
 *   ss_im2col_cl_bf16: x fp32 NHWC [NB][h][w][C] (C % 8 == 0) -> A bf16 [NB*ho*wo][k*k*C], A[(nb,oy,ox)][(ky,kx,c)] =
 *                      x[nb][oy*stride-pad+ky][ox*stride-pad+kx][c] (0 outside), narrowed to nearest-even bf16 (exact for spike counts).
 *   ss_split3_bf16   : g fp32 [M][N] (N % 4 == 0) -> g3 bf16 [M][3N] = [hi | mid | lo], hi + mid + lo == g exactly.
 */
int ss_im2col_cl_bf16(const float* x, void* A, long long NB, int h, int w, int C, int k, int stride, int pad, int ho, int wo, void* stream);
int ss_split3_bf16(const float* g, void* g3, long long M, int N, void* stream);
/*   ss_wgrad_reduce3_f32 (ABI 8): epilogue of that weight-gradient GEMM (autograd's Conv2d weight gradient, /root/reference/network/SNN_models.py:91-101,
 *                      blocks.py:146-159): parts fp32 [S][k*k*C_in rows (ky, kx, ci)][3][C_out] — S split-K slices of A^T @ [g_hi | g_mid | g_lo] —
 *                      -> g_w fp32 [C_out][C_in][k][k] = sum over slices (ascending) of ((hi + mid) + lo).  C_in % 8 == 0, C_out % 32 == 0, k <= 7. */
int ss_wgrad_reduce3_f32(const float* parts, float* g_w, int S, int k, int Cin, int Cout, void* stream);

/*
 * Fused per-scale loss statistics — the step AFTER the path: /root/reference/network/loss.py:7-24
 * (ScaleInvariant_Loss), :44-75 (GradientMatching_Loss) and network/metrics.py:83-95 (MeanDepthError) for ONE prediction
 * map against the ground truth, batch-wide exactly as the reference (mask, n and the sums run over the whole [B,1,H,W] batch).
 * With r = (gt is NaN) ? 0 : pred - gt  and Sobel cross-correlations with zero padding 1 (loss.py:61-69):
 *   sums[0] = n  = #valid pixels         sums[1] = sum r          sums[2] = sum r^2
 *   sums[3] = sum over valid pixels of |sobelX * r| + |sobelY * r|          sums[4] = sum |r|
 * so that  ScaleInvariant = sums[2]/n - (sums[1]/n)^2,  GradientMatching = sums[3]/n,  MDE = sums[4]/n.
 * Accumulation: fp32 per lane -> wavefront -> workgroup partial in fp64 (ws, ss_loss_ws_doubles() doubles) -> fixed-order
 * fp64 second pass => deterministic.  sums are float64[5] in device memory.
 */
long long ss_loss_ws_doubles(void);
int ss_loss_stats_f32(const float* pred, const float* gt, double* sums, double* ws, long long B, int H, int W, void* stream);

/*
 * Gradient of   c_si * ScaleInvariant + c_gm * GradientMatching   w.r.t. pred, with (c_si, c_gm) = coef[0], coef[1] read from
 * DEVICE memory (they are the upstream gradients of the two terms — scale weight, alpha and d loss already folded in — and live on
 * the device, so no host synchronisation is needed between loss.backward() and this launch):  g_pred = valid ? c_si*(2 r/n - 2 sums[1]/n^2) + (c_gm/n) * (sobelX^T[sgn(gx) m] + sobelY^T[sgn(gy) m]) : 0
 * where m is the valid mask and sgn(0) = 0 (torch.abs backward).  sums are the forward's.
 */
int ss_loss_grad_f32(const float* pred, const float* gt, const double* sums, const float* coef, float* g_pred,
                     long long B, int H, int W, void* stream);

/*
 * ABI 8 — decoder stage FORWARD in the sub-pixel ("merged tap") form (stereospike_amd/csrc/ss_upconv_sub.hip).  Replaces, for spike inputs,
 * NNConvUpsampling.forward (/root/reference/network/blocks.py:110-132: UpsamplingNearest2d(size = up + 4) -> Conv2d(k = 5, bias=False)) at the decoder
 * call sites /root/reference/network/SNN_models.py:110-129:
 *     y[nb][Y][X][co] = sum_{ky, kx, ci} W[co][ci][ky][kx] * x[nb][src_y[Y + ky]][src_x[X + kx]][ci]
 * evaluated as  sum_{r, c < 3, ci} Wm[class(Y)][class(X)][co][ci][r][c] * x[nb][src_y[Y + k0_r]][src_x[X + k0_c]][ci]  with the taps that read one source
 * pixel added first (9 instead of 25 multiply-adds per output element and input channel; no per-tap tensor, no gather).  Host tables (fused.sub_tables;
 * restated in oracle/np_upconv_sub.py), all int32:
 *   vcls [NVC][8] / hcls [NHC][8]   : per class of output rows / columns: number of runs, first tap of run 0..2, tap count of run 0..2, 0
 *   vblk [NVB][vrec_ints]            : per block of <= block_rows output rows of ONE class: class, n_out, n_src, out[block_rows] (output rows),
 *                                      src[window_rows] (the distinct source rows they read), slot[block_rows][3] (index into src of run r's source row),
 *                                      last word of the record: the class's number of runs
 *   hblk [NHB][hrec_ints]            : the same for columns (block_cols, window_cols)
 * Every output row belongs to exactly one row block, every column to one column block; a (row block, column block) pair is one tile.
 *   tblk [NTB][trec_ints]            : TALL row blocks (<= 64 rows, <= 68 distinct source rows; same fields) — paired with the column blocks of <= 8 columns,
 *                                      the four wavefronts of a tile stacked vertically (ss_upconv_sub_tall_geometry; NTB may be 0)
 *   order [NORD]                     : the tiles of a frame, most expensive first: pair = row block * NHB + column block for a normal tile,
 *                                      (NVB + tall row block) * NHB + column block for a tall one; every output pixel in exactly one tile.  The kernel's
 *                                      workgroups draw tiles (pair order[t / NB], frame t % NB) from `counter`, one device word the entry point zeroes on the
 *                                      launch stream
 *   ss_upconv_sub_geometry : block_rows 16, block_cols 32, window_rows 20, window_cols 36, record sizes 88 / 168; returns the runs per class (3).
 *   ss_upconv_sub_prep_f32 : weight [C_out][C_in][5][5] fp32 -> wm (ss_upconv_sub_wm_elems bf16 elements, 16-byte aligned): the merged taps, added in fp32
 *                            (ky outer, kx inner, from +0), exactly split into three bf16 terms, in MFMA fragment order.  Once per weight update.
 *   ss_upconv_sub_fwd_f32  : x fp32 NHWC [NB][h][w][C_in] spike counts exact in bf16 (or x_packed: the 2-bit packed tensor; x may then be NULL) ->
 *                            out fp32 NHWC [NB][H][W][C_out], every element written.  C_in % 16 == 0, C_out % 32 == 0, NB h w C_in < 2^32.
 * Accuracy: exact products; |out - float64| <= 2^-21 sum |x||W| element-wise (fp32 rounding of the <= 9-term weight sums + fp32 accumulation over
 * 9 C_in products).  Deterministic; not bit-identical to the projected form (different association of the same sum).
 */
int ss_upconv_sub_geometry(int* block_rows, int* block_cols, int* window_rows, int* window_cols, int* vrec_ints, int* hrec_ints);
int ss_upconv_sub_tall_geometry(int* block_rows, int* window_rows, int* trec_ints, int* narrow_cols);   /* returns the window's capacity in pixels */
int ss_upconv_sub_supported(int Cin, int Cout, int k);
long long ss_upconv_sub_wm_elems(int Cin, int Cout, int NVC, int NHC);
int ss_upconv_sub_prep_f32(const float* weight, const int* vcls, const int* hcls, void* wm, int Cin, int Cout, int NVC, int NHC, void* stream);
int ss_upconv_sub_fwd_f32(const float* x, const unsigned int* x_packed, const void* wm, const int* vblk, const int* hblk, const int* order, unsigned int* counter,
                          float* out, long long NB, int Cin, int Cout, int h, int w, int H, int W, int NVB, int NHB, int NHC,
                          const int* tblk, int NTB, int NORD, void* stream);

/*
 * ABI 9 — the 16-bit activation modes (BASELINE.json configs 2 / 5: bf16 / fp16 activations in HBM, fp32 membranes) on the engine's OWN synapse kernels
 * (round 5).  The reference is fp32-only (/root/reference/train.py:194-197); the modes are a build-side addition whose semantics are:
 *   * activations (synapse outputs = neuron inputs) and activation gradients are STORED in `dtype` (SS_DT_BF16 | SS_DT_F16) and are, as stored, the
 *     operands of the next contraction — no conversion pass anywhere;
 *   * spike tensors travel 2-bit packed (the packed format knows no dtype) or dense in `dtype` (small integers: exact);
 *   * the fp32 master weight of a synapse is rounded ONCE to `dtype` (round to nearest even) inside the kernel's weight-preparation launch — what
 *     torch.autocast does to a convolution's weight; every product of two `dtype` values is exact in fp32 (8 x 8 / 11 x 11 significand bits), the
 *     accumulation is fp32 on v_mfma_f32_32x32x16_{bf16,f16}, the result is narrowed once on store;
 *   * weight gradients are fp32 (fp32 accumulation of exact products, fixed-order reductions) — they are NOT rounded to 16 bits as autocast's are.
 * One MFMA per k-step where the fp32 mode issues three (spike operand) or six (dense operands).  Every entry point below is the `_f32` entry point of
 * the same name with the activation pointers retyped; argument meaning, validation, workspace sizes (`*_ws_floats` of the fp32 form: upper bounds),
 * determinism and error behaviour are those of the fp32 form.  Error bounds: |result - float64(result from the SAME rounded operands)| <= 2^-21 sum |a||b|
 * before the final narrowing (fp32 accumulation only).
 *   ss_neuron_fwd_ex                act_dtype != 0 now also takes out_packed / skip_packed (out_seq may then be NULL): 2.25 B/update forward
 *   ss_neuron_bwd_fork_lr_x16       ss_neuron_bwd_fork_lr_f32 with g_out_seq (nullable) / g_sum_seq / x_seq / g_x_seq in `dtype`; the rank-9 pair stays fp32
 *   ss_dense_conv_s1_{fwd,wgrad}_x16  first encoder layer: the fp32 event-voxel input and the weight rounded once to `dtype`; out / g in `dtype`
 *   ss_spike_conv_{fwd,wgrad}_x16   conv1 / conv2: x = dense `dtype` spikes (nullable) or x_packed; out / g in `dtype`; g_w fp32
 *   ss_conv_s2_dgrad_x16            encoder data gradients: g, g_x in `dtype`
 *   ss_im2col_cl_packed_x16         ss_im2col_cl_bf16_packed with the patch matrix in `dtype`;  ss_im2col_cl_x16: patch matrix of a DENSE 16-bit array
 *   ss_upconv_sub_{prep,fwd}_x16    decoder stage forward: taps rounded once to `dtype`, their merged sums carried as TWO `dtype` terms (16 / 22 bits)
 *   ss_upconv_box{sum,_dgrad,_wgrad}_x16  decoder stage backward on the box-sum image: ss_upconv_box_planes_x16(dtype) planes of `dtype` (one: the box sum rounded once, as the g_P path rounds g_P)
 *                                         (round 6: the two contraction kernels stage TWO consecutive 8-channel chunks per window where C_out % 16 == 0 — same layout, same entry points;
 *                                         ss_upconv_box_wgrad_ws_floats returns one size that serves the f32 and the x16 entry point)
 */
int ss_neuron_bwd_fork_lr_x16_supported(int T, long long N, int C, int lr_rank);
int ss_neuron_bwd_fork_lr_x16(const void* g_out_seq, const float* lr_p, const float* lr_w, int lr_rank, int C, void* g_sum_seq,
                              const float* g_v_last, const void* x_seq, const float* v_init, void* g_x_seq, float* g_v_init,
                              float* g_k, float* g_k_ws, int T, long long N, float scale, int kind, float tau, const float* k,
                              float v_th, float v_reset, int surrogate, float alpha, int detach_reset, int dtype, void* stream);
int ss_dense_conv_s1_fwd_x16(const float* x, const float* weight, void* out, long long NB, int Cin, int Cout, int h, int w, int dtype, void* stream);
int ss_dense_conv_s1_wgrad_x16(const void* g, const float* x, float* g_w, float* ws, long long NB, int Cin, int Cout, int h, int w, int accumulate, int dtype, void* stream);
int ss_spike_conv_fwd_x16(const void* x, const unsigned int* x_packed, const float* weight, void* out, float* ws,
                          long long NB, int Cin, int Cout, int h, int w, int dtype, void* stream);
int ss_spike_conv_wgrad_x16(const void* g, const void* x, const unsigned int* x_packed, float* g_w, float* ws, long long NB, int Cin, int Cout, int h,
                            int w, int accumulate, int dtype, void* stream);
int ss_conv_s2_dgrad_x16(const void* g, const float* weight, void* g_x, float* ws, long long NB, int Cin, int Cout, int h, int w, int dtype, void* stream);
int ss_im2col_cl_packed_x16(const unsigned int* x_packed, void* A, long long NB, int h, int w, int C, int k, int stride, int pad, int ho, int wo, int dtype, void* stream);
int ss_im2col_cl_x16(const void* x, void* A, long long NB, int h, int w, int C, int k, int stride, int pad, int ho, int wo, void* stream);
int ss_upconv_sub_prep_x16(const float* weight, const int* vcls, const int* hcls, void* wm, int Cin, int Cout, int NVC, int NHC, int dtype, void* stream);
int ss_upconv_sub_fwd_x16(const void* x, const unsigned int* x_packed, const void* wm, const int* vblk, const int* hblk, const int* order, unsigned int* counter,
                          void* out, long long NB, int Cin, int Cout, int h, int w, int H, int W, int NVB, int NHB, int NHC, const int* tblk, int NTB, int NORD,
                          int dtype, void* stream);
int ss_upconv_box_planes_x16(int dtype);
int ss_upconv_boxsum_x16(const void* g_out, const int* vr, const int* hr, void* box, long long NB, int Cout, int H, int W, int NVR, int NHR, int dtype, void* stream);
int ss_upconv_box_dgrad_x16(const void* box, const float* weight, const int* vmap, const int* hmap, const int* tile_rows, int n_row_tiles, const int* tile_cols,
                            void* g_x, float* ws, long long NB, int Cin, int Cout, int h, int w, int NVR, int NHR, int dtype, void* stream);
int ss_upconv_box_wgrad_x16(const void* box, const void* x, const unsigned int* x_packed, const int* vmap, const int* hmap, const int* tile_rows, int n_row_tiles,
                            const int* tile_cols, float* g_w, float* ws, long long NB, int Cin, int Cout, int h, int w, int NVR, int NHR, int accumulate,
                            int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SS_NEURON_H */

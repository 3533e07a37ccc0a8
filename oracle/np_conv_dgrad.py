"""TEST INFRASTRUCTURE (CPU checker, never imported by the product): numpy restatement of the data gradient of the reference's stride-2 5x5 encoder
convolutions, nn.Conv2d(C, 2C, kernel_size=5, stride=2, padding=2, bias=False) (/root/reference/network/SNN_models.py:80-101; autograd's
convolution_backward w.r.t. the input), in the FORM the HIP kernel ss_conv_s2_dgrad_f32 evaluates it (stereospike_amd/csrc/ss_conv_dgrad.hip):

    g_x[nb, 2 j + py, 2 i + px, ci] = sum over the taps (ky, kx) with ky = py, kx = px (mod 2) and over co of
                                      g[nb, j + (py + 2 - ky) / 2, i + (px + 2 - kx) / 2, co] * W[co, ci, ky, kx]          (zero outside the map)

— four stride-1 convolutions over g, one per parity class of the input pixel (9 / 6 / 6 / 4 taps).  Pinned in tests/test_oracle.py against torch's own
conv2d input gradient on the CPU; the HIP kernel is held against it in tests/test_gpu_01_kernels.py."""
import numpy as np

CLASS_TAPS = {(py, px): [(ky, kx) for ky in range(5) for kx in range(5) if ky % 2 == py and kx % 2 == px] for py in (0, 1) for px in (0, 1)}


def conv_s2_dgrad(g, w, h, wd, dtype=np.float64):
    """g [NB, ho, wo, C_out] (NHWC), w [C_out, C_in, 5, 5] -> g_x [NB, h, wd, C_in]; ho = (h - 1) // 2 + 1, wo = (wd - 1) // 2 + 1."""
    g = np.asarray(g, dtype)
    w = np.asarray(w, dtype)
    NB, ho, wo, Cout = g.shape
    Cin = w.shape[1]
    assert ho == (h - 1) // 2 + 1 and wo == (wd - 1) // 2 + 1 and w.shape[0] == Cout
    gp = np.zeros((NB, ho + 2, wo + 2, Cout), dtype)                 # one zero row / column on every side: dy, dx in {-1, 0, 1}
    gp[:, 1:-1, 1:-1] = g
    gx = np.zeros((NB, h, wd, Cin), dtype)
    for (py, px), taps in CLASS_TAPS.items():
        nj, ni = (h - py + 1) // 2, (wd - px + 1) // 2               # pixels (2 j + py, 2 i + px) of this class inside the map
        if nj <= 0 or ni <= 0:
            continue
        acc = np.zeros((NB, nj, ni, Cin), dtype)
        for ky, kx in taps:
            dy, dx = (py + 2 - ky) // 2, (px + 2 - kx) // 2
            acc += gp[:, 1 + dy:1 + dy + nj, 1 + dx:1 + dx + ni] @ w[:, :, ky, kx]
        gx[:, py::2, px::2] = acc
    return gx

"""Round 6 A/B: the decoder stages' adjoint gather (ss_upconv_cl_bwd_f32 / ss_upconv_cl_bwd_lowp, k 5) with the next window row prefetched against the build
without (SS_LIB=stereospike_amd/lib/libss_neuron_nopf.so: make variant VARIANT=nopf DEFS=-DSS_CL_BWD_PREFETCH=0), deconv3 / deconv4 geometries, plus the x2
prediction head (k 3, one channel), with digests.

    [SS_LIB=...] python tools/r06/bench_adjoint.py
"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stereospike_amd import _lib                  # noqa: E402
from stereospike_amd.fused import nearest_tables  # noqa: E402

if os.environ.get('SS_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['SS_LIB'])
DEV = 'cuda:0'


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def digest(t):
    return hashlib.sha1(t.contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:12]


def main():
    print('SS_LIB =', os.environ.get('SS_LIB', 'default'))
    for dt, NB in ((torch.float16, 320), (torch.bfloat16, 80), (torch.float32, 80)):
        for name, C, k, (h, w), (H, W) in (('deconv3', 128, 5, (33, 44), (65, 87)), ('deconv4', 256, 5, (17, 22), (33, 44)), ('head1', 1, 3, (130, 173), (260, 346))):
            if C == 1 and dt != torch.float32:
                continue
            _, ylo, yhi = (t.to(DEV) for t in nearest_tables(h, H + k - 1))
            _, xlo, xhi = (t.to(DEV) for t in nearest_tables(w, W + k - 1))
            gen = torch.Generator(device=DEV).manual_seed(7)
            g = (torch.randn(NB, H, W, C, device=DEV, generator=gen) * 1e-2).to(dt)
            if dt == torch.float32:
                gP = torch.empty((NB * h * w, k * k * C), dtype=torch.float32, device=DEV)
                t = timed(lambda: _lib.upconv_cl_bwd(g, ylo, yhi, xlo, xhi, gP, NB, k, C, h, w, H, W))
            else:
                gP = torch.empty((NB * h * w, k * k * C), dtype=torch.bfloat16, device=DEV)
                t = timed(lambda: _lib.upconv_cl_bwd_lowp(g, ylo, yhi, xlo, xhi, gP, NB, k, C, h, w, H, W))
            gb = gP.numel() * gP.element_size() / 1e9
            print(f'{str(dt)[6:]:9s} {name} NB {NB}: {1e3 * t:7.1f} us, g_P {gb:5.2f} GB = {gb / t:5.2f} TB/s written   sha1 {digest(gP)}', flush=True)
            del g, gP


if __name__ == '__main__':
    main()

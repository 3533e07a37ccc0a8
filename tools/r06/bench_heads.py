"""Round 6: the prediction heads' gather (ss_upconv_cl_fwd_f32, k 3, one channel) and its adjoint (ss_upconv_cl_bwd_f32) per head geometry, at config 5's per-GPU
share (320 frames) and config 3 (80 frames); digests of the results for bit-for-bit comparison between library builds.

    python tools/r06/bench_heads.py
"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stereospike_amd import _lib                  # noqa: E402
from stereospike_amd.fused import nearest_tables  # noqa: E402

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
    return hashlib.sha1(t.contiguous().view(torch.int32).cpu().numpy().tobytes()).hexdigest()[:12]


def main():
    print('library', _lib.source_hash())
    H, W, k = 260, 346, 3
    for NB in (320, 80):
        for name, (h, w) in (('head4', (17, 22)), ('head3', (33, 44)), ('head2', (65, 87)), ('head1', (130, 173))):
            sy, ylo, yhi = (t.to(DEV) for t in nearest_tables(h, H + k - 1))
            sx, xlo, xhi = (t.to(DEV) for t in nearest_tables(w, W + k - 1))
            gen = torch.Generator(device=DEV).manual_seed(3)
            P = torch.randn(NB, h, w, 9, device=DEV, generator=gen)
            bias = torch.randn(1, device=DEV, generator=gen)
            out = torch.empty(NB, H, W, 1, device=DEV)
            g = torch.randn(NB, H, W, 1, device=DEV, generator=gen)
            gP = torch.empty_like(P)
            tf = timed(lambda: _lib.upconv_cl_fwd(P, sy, sx, bias, out, NB, k, 1, h, w, H, W))
            tb = timed(lambda: _lib.upconv_cl_bwd(g, ylo, yhi, xlo, xhi, gP, NB, k, 1, h, w, H, W))
            mb_f = (P.numel() + out.numel()) * 4 / 1e6
            print(f'NB {NB:4d} {name} ({h}x{w}): gather {1e3 * tf:7.1f} us ({mb_f / tf / 1e3:5.2f} TB/s of P + out)   adjoint {1e3 * tb:7.1f} us ({mb_f / tb / 1e3:5.2f} TB/s)   '
                  f'sha1 out {digest(out)} g_P {digest(gP)}', flush=True)


if __name__ == '__main__':
    main()

"""Round 6: the patch-matrix kernels (ss_im2col_cl_packed_x16 / ss_im2col_cl_bf16_packed / ss_im2col_cl_x16) at the shapes of conv3 / conv4 / the bottleneck,
config 5's per-GPU share (320 frames) and config 3 (80 frames): time and written bytes per second.

    python tools/r06/bench_im2col.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stereospike_amd import _lib      # noqa: E402

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


def main():
    print('library', _lib.source_hash())
    for NB, dt in ((320, torch.float16), (80, torch.bfloat16)):
        for name, C, (h, w), k, s, p in (('conv3', 128, (65, 87), 5, 2, 2), ('conv4', 256, (33, 44), 5, 2, 2), ('bottleneck', 512, (17, 22), 3, 1, 1)):
            ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
            n = NB * h * w * C
            xp = torch.randint(-2 ** 31, 2 ** 31 - 1, (n // 16,), dtype=torch.int32, device=DEV)
            xd = torch.randn(NB, h, w, C, device=DEV).to(dt)
            A = torch.empty((NB * ho * wo, k * k * C), dtype=dt, device=DEV)
            gb = A.numel() * 2 / 1e9
            t1 = timed(lambda: _lib.im2col_cl_packed_x16(xp, A, NB, h, w, C, k, s, p, ho, wo))
            t2 = timed(lambda: _lib.im2col_cl_x16(xd, A, NB, h, w, C, k, s, p, ho, wo))
            print(f'{str(dt)[6:]:9s} NB {NB:4d} {name:10s} A {gb:6.3f} GB: packed in {1e3 * t1:7.1f} us = {gb / t1:5.2f} TB/s written   dense 16-bit in {1e3 * t2:7.1f} us = {gb / t2:5.2f} TB/s',
                  flush=True)
            del A, xd, xp


if __name__ == '__main__':
    main()

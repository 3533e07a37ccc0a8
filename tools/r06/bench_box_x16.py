"""Round 6 A/B: the 16-bit box kernels with chunk GROUPS as window planes (ss_upconv_box.hip, BxT comment) against the one-chunk form (SS_BOX_X16_NG1=1).
Times ss_upconv_box_dgrad_x16 / ss_upconv_box_wgrad_x16 on the deconv1 / deconv2 geometries at config 5's per-GPU share (fp16, 320 frames)
and at config 3's shapes in bf16 (80 frames); prints a digest of the results so that the two runs can be compared bit for bit.

    python tools/r06/bench_box_x16.py            # run once per setting of SS_BOX_X16_NG1 (the switch is read once per process)
"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stereospike_amd import _lib, fused                      # noqa: E402
from stereospike_amd.network.blocks import NNConvUpsampling  # noqa: E402

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
    return hashlib.sha1(t.detach().contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:12]


def main():
    print('SS_BOX_X16_NG1 =', os.environ.get('SS_BOX_X16_NG1', '0'))
    for dt, NB in ((torch.float16, 320), (torch.bfloat16, 80)):
        for name, Cin, Cout, (h, w), (H, W) in (('deconv1', 64, 32, (130, 173), (260, 346)), ('deconv2', 128, 64, (65, 87), (130, 173))):
            up = NNConvUpsampling(Cin, Cout, 5, (H, W)).to(DEV)
            tables = up._tables(h, w, torch.device(DEV))
            bt = fused.box_tables(tables, H, W)
            gen = torch.Generator(device=DEV).manual_seed(5)
            g = (torch.randn(NB, H, W, Cout, device=DEV, generator=gen) * 1e-2).to(dt)
            wt = up.up[1].weight.detach().contiguous()
            box = _lib.upconv_boxsum_x16(g, bt, NB, Cout, H, W)
            x = (torch.rand(NB, h, w, Cin, device=DEV, generator=gen) < 0.1).float()
            xd = x.to(dt)                                       # dense spike operand (the packed one differs in upconv_bwd_xprep_kernel only)
            g_x = torch.empty((NB, h, w, Cin), device=DEV, dtype=dt)
            g_w = torch.empty((Cout, Cin, 5, 5), device=DEV)
            t_d = timed(lambda: _lib.upconv_box_dgrad_x16(box, wt, bt, g_x, NB, Cin, Cout, h, w))
            t_w = timed(lambda: _lib.upconv_box_wgrad_x16(box, xd, None, bt, g_w, NB, Cin, Cout, h, w))
            fl = 2.0 * NB * h * w * Cin * Cout * 25
            print(f'{str(dt)[6:]:9s} {name} NB {NB}: dgrad {t_d:7.3f} ms ({fl / t_d / 1e9 / 2500:.3f} of 2.5 PFLOP/s)  wgrad(+xprep+reduce) {t_w:7.3f} ms '
                  f'({fl / t_w / 1e9 / 2500:.3f})   sha1 g_x {digest(g_x)} g_w {digest(g_w)}   |g_w| {float(g_w.abs().sum()):.6e}', flush=True)
            del box, g, x, g_x


if __name__ == '__main__':
    main()

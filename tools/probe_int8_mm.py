#!/usr/bin/env python3
"""Probe: hipBLASLt int8 x int8 -> int32 GEMM (torch._int_mm) against the bf16 GEMM of the exact-split synapses at their config-3 shapes."""
import torch
dev = 'cuda:0'
shapes = [('conv3', 116160, 3200, 256), ('conv4', 29920, 6400, 512), ('bottleneck', 29920, 4608, 512), ('deconv3', 116160, 256, 3200), ('deconv4', 29920, 512, 6400)]


def t(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, M, K, N in shapes:
    a8 = torch.randint(0, 4, (M, K), dtype=torch.int8, device=dev)
    b8 = torch.randint(-127, 128, (K, 3 * N), dtype=torch.int8, device=dev)
    ab, bb = a8.to(torch.bfloat16), b8.to(torch.bfloat16)
    try:
        y = torch._int_mm(a8, b8)
        ok = torch.equal(y[:256].double(), (a8[:256].double() @ b8.double()))
        ti = t(lambda: torch._int_mm(a8, b8))
    except Exception as e:
        ok, ti = repr(e)[:200], float('nan')
    tb = t(lambda: torch.mm(ab, bb, out_dtype=torch.float32))
    fl = 2.0 * M * K * 3 * N
    print(f'{name:10s} M {M} K {K} N {3 * N}: int8 {ti:.3f} ms ({fl / ti / 1e9:.0f} TOP/s) exact {ok} | bf16 {tb:.3f} ms ({fl / tb / 1e9:.0f} TFLOP/s)')
    # K-concatenated form [A A A] is not needed for int8: digits go to the N side

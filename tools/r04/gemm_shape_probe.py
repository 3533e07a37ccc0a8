#!/usr/bin/env python3
"""The decoder's deconv3 projection GEMM (bf16 x bf16 -> fp32, M 116160, K 768 = 3 x 256, N 3200 = 25 x 128) ran at 633 us against 431 us for the other
exact-split GEMMs of the same MAC count (kernel_stats_v3.csv: MT256x192x64 vs MT256x256x64): does another way of asking for the same product do better?"""
import torch, time
dev = 'cuda:0'
M, K, N = 116160, 768, 3200
A = (torch.rand(M, K, device=dev) < 0.2).to(torch.bfloat16)
B = torch.randn(K, N, device=dev).to(torch.bfloat16)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
out = torch.empty(M, N, device=dev)
Bt = B.t().contiguous()
At = A.t().contiguous()
cases = {
    'mm(A, B, out_dtype=f32)': lambda: torch.mm(A, B, out_dtype=torch.float32),
    'mm(A, B^T.t()) (B stored [N, K])': lambda: torch.mm(A, Bt.t(), out_dtype=torch.float32),
    'two halves of N (1600 each)': lambda: (torch.mm(A, B[:, :1600], out_dtype=torch.float32), torch.mm(A, B[:, 1600:], out_dtype=torch.float32)),
    'N padded to 3328 = 13 x 256': lambda: torch.mm(A, torch.nn.functional.pad(B, (0, 128)), out_dtype=torch.float32),
    'two halves of M': lambda: (torch.mm(A[:M // 2], B, out_dtype=torch.float32), torch.mm(A[M // 2:], B, out_dtype=torch.float32)),
    '(B^T A^T) -> P^T': lambda: torch.mm(Bt, At, out_dtype=torch.float32),
}
for k, fn in cases.items():
    print(f'{k:40s} {t(fn) * 1e3:8.1f} us')
# the reference shapes that run at 431 us
A2 = (torch.rand(116160, 3200, device=dev) < 0.2).to(torch.bfloat16); B2 = torch.randn(3200, 768, device=dev).to(torch.bfloat16)
print(f'{"conv3 forward shape (M 116160, K 3200, N 768)":40s} {t(lambda: torch.mm(A2, B2, out_dtype=torch.float32)) * 1e3:8.1f} us')

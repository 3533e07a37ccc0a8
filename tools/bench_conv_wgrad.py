import sys, torch
sys.path.insert(0, '.')
from stereospike_amd import miopen_cache
miopen_cache.enable(skip_naive_solvers=True)
from stereospike_amd import _lib
torch.backends.cudnn.benchmark = True
dev = 'cuda:0'
for name, Cin, Cout, (h, w) in (('conv1', 32, 64, (260, 346)), ('conv2', 64, 128, (130, 173))):
    NB = 80
    x = (torch.rand(NB, h, w, Cin, device=dev) < 0.3).float()
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    g = torch.randn(NB, ho, wo, Cout, device=dev)
    gw = torch.empty(Cout, Cin, 5, 5, device=dev)
    wt = torch.empty(Cout, Cin, 5, 5, device=dev).contiguous(memory_format=torch.channels_last)
    def mi(): return torch.ops.aten.convolution_backward(g.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2), wt, None, [2, 2], [2, 2], [1, 1], False, [0, 0], 1, [False, True, False])
    def own(): _lib.spike_conv_wgrad(g, x, gw, NB, Cin, Cout, h, w)
    for tag, fn in (('MIOpen fp32 wgrad', mi), ('ss_spike_conv_wgrad_f32', own)):
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        print(name, tag, round(e0.elapsed_time(e1) / 5, 3), 'ms', flush=True)

from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3): own()
    torch.cuda.synchronize()
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:4]:
    print(f'{e.device_time_total / 3 / 1e3:7.3f} ms  {e.key[:90]}')

"""Trajectory-pinned end-to-end parity (test infrastructure; GPU tests only).

Why: spike masks are bit-exact at the kernel boundary (tests/test_gpu_kernels.py), but a FREE-running comparison of two whole
networks is a comparison of two chaotic trajectories: the product's synapses (MIOpen fp32 convolutions, fp32 GEMM + gather,
exact bf16x3 MFMA GEMMs) and the oracle's (oneDNN) differ by fp32 summation order, a membrane that sits within that difference of
its threshold flips, and one flipped spike moves 25 x C_out downstream membranes by O(0.5).  Round 1 bounded that with "noise
floors" (3 - 10 % spike mismatch, gradient cosine 0.6) — bars that cannot see a moderately wrong kernel.

What is done instead: the ORACLE is pinned to the product's trajectory.

  1. the product runs freely (the shipped code path, nothing patched in) while a recorder notes the pure spike tensor z of every
     neuron layer;
  2. the oracle (oracle/ref_network.py, with `float64_convs`: every synapse in float64, rounded once) runs the same input and
     weights step by step, and at every neuron layer its Heaviside output is REPLACED by the product's z (the surrogate-gradient
     path is kept: spike = s + (z - s).detach()).  So every oracle layer sees exactly the inputs the product's layer saw, computes
     its own membrane h, and disagrees with the product only where |h - v_th| is within the synapse rounding error;
  3. asserted: (a) the fraction of disagreeing neurons per layer is tiny and EVERY disagreeing neuron is near threshold (a wrong
     tap, index, gain, reset or time order produces disagreements far from threshold and fails here); (b) with the trajectories
     thus identical, depths, loss, MDE and EVERY parameter gradient of the composed forward + backward (fork handles, summed skip
     gradients, exact-split weight gradients, split-K, I-pool order, fused loss) match the oracle's autograd at fp32 tolerance.

Nothing here is chaotic: no noise floors, no seeds to be lucky with, independent of which MIOpen solver a box picks."""
import contextlib
import types

import torch

from _util import ref_network as rn, sj


@contextlib.contextmanager
def record_product_spikes(net):
    """Record, per neuron node of the product `net` (keyed by module name), the list of pure spike sequences z [T, B, C, H, W]
    (uint8, CPU, logical NCHW) its fused HIP launches produced.  Wraps BaseNode.forward_sequence in this process only."""
    from stereospike_amd.clock_driven import neuron
    names = {id(m): n for n, m in net.named_modules() if isinstance(m, neuron.BaseNode)}
    rec = {}
    orig = neuron.BaseNode.forward_sequence

    def wrapped(self, x_seq, scale=1., skip_seq=None, nnz=None, channels_last=False, fork=False, pack=0, skip_packed=None):
        from stereospike_amd import fused
        res = orig(self, x_seq, scale, skip_seq, nnz, channels_last, fork, pack, skip_packed)
        out = (res[0] if fork else res).detach()
        if pack and self.last_packed is not None:          # packed(-only) output: the data lives in the packed tensor (out may be an anchor)
            out = fused.unpack_dense(self.last_packed, x_seq.shape)
        out = out.float()
        if skip_packed is not None:
            skip = fused.unpack_dense(skip_packed, x_seq.shape)
        else:
            skip = None if skip_seq is None else skip_seq.detach().float()
        z = out if skip is None else out - skip
        if channels_last:
            z = z.permute(0, 1, 4, 2, 3)
        rec.setdefault(names[id(self)], []).append(z.to(torch.uint8).cpu())
        return res
    neuron.BaseNode.forward_sequence = wrapped
    try:
        yield rec
    finally:
        neuron.BaseNode.forward_sequence = orig


def run_oracle_pinned(orc, x, z_by_node, float64=True):
    """x [B, T, C, H, W]; z_by_node {node name: [T, B, C, H, W] uint8}.  Returns (output of the last step, stats) where
    stats[name] = dict(total, flips, max_margin) — max_margin = largest |h - v_th| among neurons whose own Heaviside disagreed with z."""
    stats, step = {}, {}
    nodes = {n: m for n, m in orc.named_modules() if isinstance(m, sj.BaseNode) and n in z_by_node}
    missing = [n for n, m in orc.named_modules() if isinstance(m, sj.BaseNode) and n not in z_by_node and m.v_threshold != float('inf')]
    assert not missing, f'product recorded no spikes for {missing}'

    def make_fire(name, node):
        def fire(self):
            xh = self.v - self.v_threshold
            s = self.surrogate_function(xh)
            z = z_by_node[name][step[name]].to(s.dtype)
            step[name] += 1
            with torch.no_grad():
                flip = s != z
                st = stats[name]
                st['total'] += z.numel()
                nf = int(flip.sum())
                if nf:
                    st['flips'] += nf
                    st['max_margin'] = max(st['max_margin'], float(xh[flip].abs().max()))
            self.spike = s + (z - s).detach()
        return types.MethodType(fire, node)

    for n, m in nodes.items():
        stats[n] = dict(total=0, flips=0, max_margin=0.0)
        step[n] = 0
        m.neuronal_fire = make_fire(n, m)
    try:
        with (rn.float64_convs(orc) if float64 else contextlib.nullcontext()):
            out = rn.run_sequence(orc, x)
    finally:
        for m in nodes.values():
            m.__dict__.pop('neuronal_fire', None)
    for n in nodes:
        assert step[n] == len(z_by_node[n]), (n, step[n], len(z_by_node[n]))
    return out, stats


def rel_l2(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-300))


def pinned_parity(orc, net, x, gt, returns_spikes=True, is_ann=False, amp_dtype=None, float64=True):
    """Free product run (forward + Total_Loss + backward) vs the trajectory-pinned oracle.  Returns a report dict."""
    from stereospike_amd.clock_driven import functional
    from stereospike_amd.network.loss import Total_Loss
    from stereospike_amd.network.metrics import MeanDepthError
    dev = next(net.parameters()).device
    net.zero_grad()
    functional.reset_net(net)
    xg, gg = x.to(dev), gt.to(dev)
    with record_product_spikes(net) as rec:
        with (torch.autocast('cuda', dtype=amp_dtype) if amp_dtype is not None else contextlib.nullcontext()):
            res = net(xg) if is_ann else net.forward_sequence(xg)
            d, s = res if returns_spikes else (res, [])
            L = Total_Loss()(d, gg, s)
        mde = MeanDepthError(d[0].detach(), gg)
        L.backward()
    if dev.type == 'cuda':
        torch.cuda.synchronize()
    # one fused launch per node covers all T steps: [T, B, C, H, W] -> T tensors [B, C, H, W]
    z_by_node = {}
    for n, lst in rec.items():
        assert len(lst) == 1, (n, len(lst))
        z_by_node[n] = list(lst[0])
    orc.zero_grad()
    (res_o, stats) = run_oracle_pinned(orc, x, z_by_node, float64=float64)
    d_o, s_o = res_o if returns_spikes else (res_o, [])
    L_o = rn.total_loss(d_o, gt, s_o)
    mde_o = rn.mean_depth_error(d_o[0].detach(), gt)
    L_o.backward()
    scale = max(float(t.detach().abs().max()) for t in d_o)
    rep = dict(
        layers={n: dict(flip_frac=st['flips'] / max(1, st['total']), flips=st['flips'], max_margin=st['max_margin']) for n, st in stats.items()},
        spike_out_mismatch=max([float((a.detach().float().cpu() != b.detach()).float().mean()) for a, b in zip(s, s_o)] or [0.0]),
        depth_max_abs_rel=max(float((a.detach().cpu() - b.detach()).abs().max()) for a, b in zip(d, d_o)) / scale,
        loss=[float(L), float(L_o)], loss_rel=abs(float(L) - float(L_o)) / abs(float(L_o)),
        mde=[float(mde), float(mde_o)], mde_rel=abs(float(mde) - float(mde_o)) / abs(float(mde_o)),
        product_spike_density=[float(t.count_nonzero()) / t.numel() for t in s],
        grad_rel_l2={k: rel_l2(p.grad, dict(orc.named_parameters())[k].grad) for k, p in net.named_parameters()})
    rep['flip_frac_max'] = max([v['flip_frac'] for v in rep['layers'].values()] or [0.0])
    rep['margin_max'] = max([v['max_margin'] for v in rep['layers'].values()] or [0.0])
    rep['grad_rel_l2_max'] = max(rep['grad_rel_l2'].values())
    return rep

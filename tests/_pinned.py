"""Trajectory-pinned end-to-end parity (test infrastructure; GPU tests only).

Why: spike masks are bit-exact at the kernel boundary (tests/test_gpu_01_kernels.py), but a FREE-running comparison of two whole
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

Nothing here is chaotic: no noise floors, no seeds to be lucky with.

Bars are per KIND of quantity:
  * weight TENSORS: relative L2 vs the oracle's autograd (a wrong surrogate constant, tap or gain moves them by >= 1e-3);
  * the 0-dim PLIF `w` of a node: dL/dw = k(1-k) * sum_{t,n} g_h (x - v) is ONE cancelling sum over a whole layer and all steps (condition
    number up to ~1e3 on the bottleneck nodes), so |product - oracle| is held against the MAGNITUDE sum k(1-k) * sum |g_h (x - v)| — both
    the signed sum and the magnitude sum are taken in float64 from hooks on the oracle's own membranes (`plif_w` in the report).  The
    synapse rounding of the incoming gradient (whatever MIOpen solver or GEMM order produced it) enters that scalar relative to the
    magnitude sum, not relative to the cancelled value."""
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


def run_oracle_pinned(orc, x, z_by_node, float64=True, stats=None, plif=None, narrow=None):
    """x [B, T, C, H, W]; z_by_node {node name: [T, B, C, H, W] uint8}.  Returns (output of the last step, stats) where
    stats[name] = dict(total, flips, max_margin) — max_margin = largest |h - v_th| among neurons whose own Heaviside disagreed with z.
    stats / plif: dicts to accumulate into over several calls (chunks of a batch).  plif[name] = dict(sum, abs): float64
    sum_{t,n} g_h (x - v) and sum |g_h (x - v)| of a ParametricLIFNode, filled when the caller runs backward (tensor hooks on h)."""
    stats = {} if stats is None else stats
    step = {}
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

    def make_charge(name, node):
        def charge(self, x):               # ParametricLIFNode.neuronal_charge of oracle/sj_clock_driven.py, op for op, + the float64 hooks
            d = x - self.v if (self.v_reset is None or self.v_reset == 0.) else x - (self.v - self.v_reset)
            h = self.v + d * self.w.sigmoid()
            if h.requires_grad:
                dd = d.detach().double()

                def hook(g):
                    t = g.double() * dd
                    plif[name]['sum'] += float(t.sum())
                    plif[name]['abs'] += float(t.abs().sum())
                h.register_hook(hook)
            self.v = h
        return types.MethodType(charge, node)

    for n, m in nodes.items():
        stats.setdefault(n, dict(total=0, flips=0, max_margin=0.0))
        step[n] = 0
        m.neuronal_fire = make_fire(n, m)
        if plif is not None and isinstance(m, sj.ParametricLIFNode):
            plif.setdefault(n, dict(sum=0.0, abs=0.0))
            m.neuronal_charge = make_charge(n, m)
    try:
        with (rn.float64_convs(orc, narrow) if float64 else contextlib.nullcontext()):
            out = rn.run_sequence(orc, x)
    finally:
        for m in nodes.values():
            m.__dict__.pop('neuronal_fire', None)
            m.__dict__.pop('neuronal_charge', None)
    for n in nodes:
        assert step[n] == len(z_by_node[n]), (n, step[n], len(z_by_node[n]))
    return out, stats


def rel_l2(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-300))


def _oracle_chunk_worker(conn, orc, x, z_by_node, float64, amp_dtype, returns_spikes, threads, own=True):
    """One chunk of samples of a large batch in its own process (oracle on CPU only; the parent holds the GPU): forward WITH the graph kept,
    depths to the parent, depth gradients back, backward, parameter gradients + flip statistics + PLIF sums to the parent."""
    torch.set_num_threads(max(1, int(threads)))
    stats, plif = {}, {}
    orc.zero_grad()
    narrow = narrowing_points(amp_dtype, own) if amp_dtype is not None else None
    out, _ = run_oracle_pinned(orc, x, z_by_node, float64=float64, stats=stats, plif=plif, narrow=narrow)
    d, s = out if returns_spikes else (out, [])
    # payloads travel as numpy arrays (pickled by value): torch's shared-memory tensor passing needs the sender to outlive the receive
    conn.send(([t.detach().numpy() for t in d], [t.detach().numpy() for t in s], stats))
    g = [torch.from_numpy(a) for a in conn.recv()]
    torch.autograd.backward(list(d), g)
    conn.send(({k: p.grad.numpy() for k, p in orc.named_parameters()}, plif))
    conn.recv()                                   # the parent's acknowledgement: only now may this process go away
    conn.close()


def _oracle_in_processes(orc, x, gt, z_by_node, chunk, returns_spikes, float64, amp_dtype, stats, plif):
    """The chunked oracle evaluation of pinned_parity with every chunk in its own process, all at once (a config-3-sized batch on the GPU
    box's 256 host cores: 16 samples in the time of two).  Same function as the sequential form (tests/test_host_wiring.py)."""
    import os
    import torch.multiprocessing as mp
    B = x.shape[0]
    chunks = [(c0, min(B, c0 + chunk)) for c0 in range(0, B, chunk)]
    ctx = mp.get_context('spawn')
    sj.reset_net(orc)                             # the module travels by pickle: no graph-attached membranes / gradients on it
    orc.zero_grad(set_to_none=True)
    threads = max(1, (os.cpu_count() or 8) // len(chunks))
    procs, keep = [], []
    for c0, c1 in chunks:
        a, b = ctx.Pipe()
        zc = {n: [z[c0:c1].clone() for z in lst] for n, lst in z_by_node.items()}
        xc = x[c0:c1].clone()
        keep.append(xc)
        pr = ctx.Process(target=_oracle_chunk_worker, args=(b, orc, xc, zc, float64, amp_dtype, returns_spikes, threads))
        pr.start()
        b.close()
        procs.append((pr, a))
        keep.append((zc,))                        # arguments travel through shared memory: alive until the workers are done
    try:
        fw = [a.recv() for _, a in procs]
        d_o = [torch.cat([torch.from_numpy(f[0][i]) for f in fw]).requires_grad_() for i in range(4)]
        s_o = [torch.cat([torch.from_numpy(f[1][i]) for f in fw]) for i in range(len(fw[0][1]))]
        for f in fw:
            for n, st in f[2].items():
                acc = stats.setdefault(n, dict(total=0, flips=0, max_margin=0.0))
                acc['total'] += st['total']
                acc['flips'] += st['flips']
                acc['max_margin'] = max(acc['max_margin'], st['max_margin'])
        L_o = rn.total_loss(d_o, gt, s_o)
        L_o.backward()
        for (_, a), (c0, c1) in zip(procs, chunks):
            a.send([g.grad[c0:c1].numpy().copy() for g in d_o])
        params = dict(orc.named_parameters())
        for _, a in procs:                        # chunks in order: a fixed summation order of the parameter gradients
            grads, pl = a.recv()
            for k, g in grads.items():
                g = torch.from_numpy(g)
                params[k].grad = g.clone() if params[k].grad is None else params[k].grad + g
            for n, v in pl.items():
                acc = plif.setdefault(n, dict(sum=0.0, abs=0.0))
                acc['sum'] += v['sum']
                acc['abs'] += v['abs']
            a.send('done')
    finally:
        for pr, _ in procs:
            pr.join(60)
            if pr.is_alive():
                pr.terminate()
    assert all(pr.exitcode == 0 for pr, _ in procs), [pr.exitcode for pr, _ in procs]
    return d_o, s_o, L_o


def launch_tags():
    """{tag: launches} of the fused launches recorded by fused.TIMER since it was last cleared."""
    from stereospike_amd import fused
    return {k: v['launches'] for k, v in fused.TIMER.summary().items()}


def narrowing_points(amp_dtype, own=True, plan=None):
    """Where the product's 16-bit activation modes store 16-bit values / use 16-bit weights (module name -> (weight dtype, output dtype)).
    own=True — the shipped path since round 5 (EngineConfig.X16_OWN_KERNELS: every synapse on the engine's own single-term kernels, ABI 9):
      encoder + bottleneck convs, decoder stages on the sub-pixel forward (deconv1 .. deconv3 at 260x346; `plan` = the product run's net.plan() says which stages
        took it on the geometry at hand — the 64x80 pyramid's deconv3 does not): weight rounded ONCE to the mode's dtype inside the kernel's weight preparation,
        output stored in the mode's dtype;
      deconv4 (projection GEMM + gather, stereospike_amd/fused.py::_UpConvProjectedCL): output in the mode's dtype; weight rounded ONCE to the mode's dtype
        (round 6: the fp16 mode too — one fp16 term, as every other synapse of the mode; it was the exact bf16x3 split);
      prediction heads (k = 3, fp32 output — they feed the fp32 I-pool): heads 1 / 2 read packed spikes with the exact fp32 weight; heads 3 / 4 as deconv4.
    own=False — the round-2 .. 4 path (X16_OWN_KERNELS off: encoder / bottleneck synapses = MIOpen convolutions under autocast): encoder + bottleneck
      weight and output in the autocast dtype; every decoder stage and head as deconv4 / heads 3, 4 above."""
    def policy(name):
        dec_w = torch.bfloat16 if amp_dtype == torch.bfloat16 else (torch.float16 if (own and amp_dtype == torch.float16) else None)
        if name.startswith('predict_depth'):
            return (None if (own and name[13] in '12') else dec_w), None
        if name.startswith('deconv'):
            stage = name.split('.')[0]
            sub = (plan[stage].get('synapse_fwd', '').startswith('upconv_sub_mfma_x16') if (plan is not None and stage in plan) else stage[6] in '123')
            return (amp_dtype if (own and sub) else dec_w), amp_dtype
        return amp_dtype, amp_dtype
    return policy


def pinned_parity(orc, net, x, gt, returns_spikes=True, is_ann=False, amp_dtype=None, float64=True, oracle_chunk=None, loss_scale=1.0,
                  oracle_procs=False, penalize_spikes=False, beta=1.0, x16_own=True):
    """Free product run (forward + Total_Loss + backward) vs the trajectory-pinned oracle.  Returns a report dict.
    amp_dtype: the product runs under torch.autocast with 16-bit activations (fp32 membranes); the oracle narrows at the same points
    (narrowing_points).  loss_scale: the product's loss is multiplied by it before backward and its gradients divided afterwards (fp16
    activation gradients underflow otherwise: engine.Trainer's GradScaler does the same).
    oracle_chunk: evaluate the oracle over chunks of that many samples (the network is per-sample independent; only the loss couples the
    batch): pass 1 forward per chunk -> depths; the loss and its depth gradients on the whole batch; pass 2 forward + backward per chunk
    with those depth gradients (parameter gradients accumulate).  Bounds the CPU memory of a config-3-sized batch.
    oracle_procs: every chunk in its own process, all at once, one forward each (the graph is kept while the parent forms the loss).
    penalize_spikes / beta: Total_Loss(penalize_spikes=True, beta=...) on both sides (/root/reference/network/loss.py:96-107,126-135): the
    loss then also reads the five RETURNED spike tensors, so a gradient enters them directly (un-chunked oracle only)."""
    from stereospike_amd import fused
    from stereospike_amd.clock_driven import functional
    from stereospike_amd.network.loss import Total_Loss
    from stereospike_amd.network.metrics import MeanDepthError
    dev = next(net.parameters()).device
    net.zero_grad()
    functional.reset_net(net)
    xg, gg = x.to(dev), gt.to(dev)
    fused.TIMER.clear()
    fused.TIMER.enabled = dev.type == 'cuda'        # HIP events: the launch tags exist on the GPU only
    try:
        with record_product_spikes(net) as rec:
            with (torch.autocast('cuda', dtype=amp_dtype) if amp_dtype is not None else contextlib.nullcontext()):
                res = net(xg) if is_ann else net.forward_sequence(xg)
                d, s = res if returns_spikes else (res, [])
                L = Total_Loss(penalize_spikes=penalize_spikes, beta=beta)(d, gg, s)
            mde = MeanDepthError(d[0].detach(), gg)
            (L * loss_scale if loss_scale != 1.0 else L).backward()
            if loss_scale != 1.0:
                for p in net.parameters():
                    p.grad.div_(loss_scale)
        if dev.type == 'cuda':
            torch.cuda.synchronize()
        tags = launch_tags()
    finally:
        fused.TIMER.enabled = False
        fused.TIMER.clear()
    # one fused launch per node covers all T steps: [T, B, C, H, W] -> T tensors [B, C, H, W]
    z_by_node = {}
    for n, lst in rec.items():
        assert len(lst) == 1, (n, len(lst))
        z_by_node[n] = list(lst[0])
    orc.zero_grad()
    plif, stats = {}, {}
    B = x.shape[0]
    narrow = narrowing_points(amp_dtype, x16_own, net.plan() if hasattr(net, 'plan') else None) if amp_dtype is not None else None
    if oracle_chunk is None or oracle_chunk >= B:
        res_o, _ = run_oracle_pinned(orc, x, z_by_node, float64=float64, stats=stats, plif=plif, narrow=narrow)
        d_o, s_o = res_o if returns_spikes else (res_o, [])
        L_o = rn.total_loss(d_o, gt, s_o, penalize_spikes=penalize_spikes, beta=beta)
        mde_o = rn.mean_depth_error(d_o[0].detach(), gt)
        L_o.backward()
    elif oracle_procs:
        assert not penalize_spikes, 'the chunked oracle forms the loss from detached spike tensors'
        d_o, s_o, L_o = _oracle_in_processes(orc, x, gt, z_by_node, oracle_chunk, returns_spikes, float64, amp_dtype, stats, plif)
        mde_o = rn.mean_depth_error(d_o[0].detach(), gt)
    else:
        assert not penalize_spikes, 'the chunked oracle forms the loss from detached spike tensors'
        chunks = [(c0, min(B, c0 + oracle_chunk)) for c0 in range(0, B, oracle_chunk)]
        zc = lambda c0, c1: {n: [z[c0:c1] for z in lst] for n, lst in z_by_node.items()}        # noqa: E731
        outs = []
        with torch.no_grad():
            for c0, c1 in chunks:
                outs.append(run_oracle_pinned(orc, x[c0:c1], zc(c0, c1), float64=float64, stats=stats, narrow=narrow)[0])
        outs = [o if returns_spikes else (o, []) for o in outs]
        d_o = [torch.cat([o[0][i] for o in outs]).requires_grad_() for i in range(4)]
        s_o = [torch.cat([o[1][i] for o in outs]) for i in range(len(outs[0][1]))]
        L_o = rn.total_loss(d_o, gt, s_o)
        mde_o = rn.mean_depth_error(d_o[0].detach(), gt)
        L_o.backward()
        for c0, c1 in chunks:
            o = run_oracle_pinned(orc, x[c0:c1], zc(c0, c1), float64=float64, stats={}, plif=plif, narrow=narrow)[0]
            dc = o[0] if returns_spikes else o
            torch.autograd.backward(list(dc), [g.grad[c0:c1] for g in d_o])
    scale = max(float(t.detach().abs().max()) for t in d_o)
    orc_p = dict(orc.named_parameters())
    grad_rel_l2 = {k: rel_l2(p.grad, orc_p[k].grad) for k, p in net.named_parameters()}
    # 0-dim PLIF w of node `n`: dL/dw = dL/dk * k (1 - k); the product's value against the float64 sum, in units of the magnitude sum
    plif_w = {}
    for n, acc in plif.items():
        k = float(torch.sigmoid(orc_p[n + '.w'].detach().double()))
        ref64, mag = acc['sum'] * k * (1 - k), acc['abs'] * k * (1 - k)
        g_prod, g_orc = float(dict(net.named_parameters())[n + '.w'].grad), float(orc_p[n + '.w'].grad)
        plif_w[n + '.w'] = dict(product=g_prod, oracle_autograd_fp32=g_orc, oracle_float64=ref64, magnitude_sum=mag,
                                condition=mag / max(abs(ref64), 1e-300), err_over_magnitude=abs(g_prod - ref64) / max(mag, 1e-300),
                                oracle_fp32_err_over_magnitude=abs(g_orc - ref64) / max(mag, 1e-300))
    rep = dict(
        layers={n: dict(flip_frac=st['flips'] / max(1, st['total']), flips=st['flips'], max_margin=st['max_margin']) for n, st in stats.items()},
        spike_out_mismatch=max([float((a.detach().float().cpu() != b.detach()).float().mean()) for a, b in zip(s, s_o)] or [0.0]),
        depth_max_abs_rel=max(float((a.detach().float().cpu() - b.detach()).abs().max()) for a, b in zip(d, d_o)) / scale,
        loss=[float(L.detach()), float(L_o.detach())], loss_rel=abs(float(L.detach()) - float(L_o.detach())) / abs(float(L_o.detach())),
        mde=[float(mde), float(mde_o)], mde_rel=abs(float(mde) - float(mde_o)) / abs(float(mde_o)),
        product_spike_density=[float(t.count_nonzero()) / t.numel() for t in s],
        grad_rel_l2=grad_rel_l2, plif_w=plif_w, launch_tags=tags, plan=net.plan() if hasattr(net, 'plan') else {})
    rep['flip_frac_max'] = max([v['flip_frac'] for v in rep['layers'].values()] or [0.0])
    rep['margin_max'] = max([v['max_margin'] for v in rep['layers'].values()] or [0.0])
    rep['grad_rel_l2_max'] = max(rep['grad_rel_l2'].values())
    rep['tensor_grad_rel_l2_max'] = max([v for k, v in grad_rel_l2.items() if k not in plif_w] or [0.0])
    rep['plif_w_err_over_magnitude_max'] = max([v['err_over_magnitude'] for v in plif_w.values()] or [0.0])
    return rep

#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc outputs (rocpd sqlite .db and/or counter_collection.csv) to per-launch HBM bytes for the
fused kernels (see collect_pmc.sh).  gfx950: FETCH_SIZE counts 64 B per 128-B request => doubled (MI355X_MICROARCH.md §HBM);
FETCH_SIZE / WRITE_SIZE are in KiB."""
import csv
import glob
import json
import os
import sqlite3
import sys

out_dir = sys.argv[1]
mode = sys.argv[2] if len(sys.argv) > 2 else 'rc'


def key_of(name):
    if 'neuron_fwd16_pk8_kernel' in name:                           # round 6: <KIND, DT, T, SKIP, DENSE>
        import re
        m = re.search(r'neuron_fwd16_pk8_kernel<([^>]*)>', name)
        args = [a.strip() for a in m.group(1).split(',')] if m else []
        skip = len(args) >= 4 and args[3] == 'true'
        if mode == 'x16c5':
            return 'neuron_fwd_skip_packed' if skip else 'neuron_fwd_packed'
        return 'neuron_fwd_x16_packed'
    if 'neuron_bwd16_seg_kernel' in name:                           # round 6: the low-rank form <KIND, SG, DT, T, VEC, NSEG, G2, LR, WAVES, HAS_G1, SUM, PASS>
        import re
        m = re.search(r'neuron_bwd16_seg_kernel<([^>]*)>', name)
        args = [a.strip() for a in m.group(1).split(',')] if m else []
        if len(args) >= 12 and args[11] == '1':                    # the exact pass behind the fast one: returns at once unless asked for (no traffic)
            return None
        return 'neuron_bwd_lr' if mode == 'x16c5' else 'neuron_bwd_x16_lr'
    if 'neuron_fwd16_kernel' in name:
        return 'neuron_fwd_x16_packed'
    if 'neuron_bwd16_rc_kernel' in name:
        return 'neuron_bwd_x16_lr'
    if 'neuron_fwd_kernel' in name:
        import re
        m = re.search(r'neuron_fwd_kernel<([^>]*)>', name)
        args = [a.strip() for a in m.group(1).split(',')] if m else []
        if len(args) >= 6 and args[5] == 'true':                                                     # 6th template argument = PK, 3rd = SKIP
            return 'neuron_fwd_skip_packed' if args[2] == 'true' else 'neuron_fwd_packed'
        return 'neuron_fwd'
    if 'conv_s2_dgrad_kernel' in name:
        return 'conv_s2_dgrad'
    if 'dense_conv_s1_wgrad_kernel' in name:
        return 'dense_conv_s1_wgrad'
    if 'head_proj_packed_kernel' in name:
        return 'head_proj_packed'
    if 'head_wgrad_packed_kernel' in name:
        return 'head_wgrad_packed'
    if 'spike_conv_fwd_kernel' in name:
        return 'spike_conv_fwd'
    if 'dense_conv_s1_fwd_kernel' in name:
        return 'dense_conv_s1_fwd'
    if 'upconv_sub_fwd_kernel' in name:
        return 'upconv_sub'
    if 'upconv_boxsum_kernel' in name:
        return 'upconv_boxsum'
    if 'upconv_box_dgrad_kernel' in name:
        return 'upconv_box_dgrad'
    if 'upconv_box_wgrad_kernel' in name:
        return 'upconv_box_wgrad'
    if 'neuron_bwd_kernel' in name:
        import re
        m = re.search(r'neuron_bwd_kernel<([^>]*)>', name)
        args = [a.strip() for a in m.group(1).split(',')] if m else []
        return 'neuron_bwd_lr' if (len(args) >= 7 and args[6] == 'true') else 'neuron_bwd'          # 7th template argument = LR (rank-9 second gradient)
    return None


def mean_counter(sub, counter):
    vals = {}
    for path in glob.glob(os.path.join(out_dir, sub, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get('Counter_Name') == counter and key_of(row.get('Kernel_Name', '')):
                vals.setdefault(key_of(row['Kernel_Name']), []).append(float(row['Counter_Value']))
    if not vals:
        for path in glob.glob(os.path.join(out_dir, sub, '**', '*.db'), recursive=True):
            cur = sqlite3.connect(path).cursor()
            q = ("select s.kernel_name, e.value from rocpd_pmc_event e join rocpd_info_pmc i on e.pmc_id = i.id "
                 "join rocpd_kernel_dispatch d on e.event_id = d.event_id "
                 "join rocpd_info_kernel_symbol s on d.kernel_id = s.id where i.name = ?")
            for name, value in cur.execute(q, (counter,)):
                if key_of(name):
                    vals.setdefault(key_of(name), []).append(float(value))
    return {k: sum(v) / len(v) for k, v in vals.items()}, {k: len(v) for k, v in vals.items()}


fetch, nf = mean_counter('fetch', 'FETCH_SIZE')
write, nw = mean_counter('write', 'WRITE_SIZE')
T, N = 5, 16 * 32 * 260 * 346
if mode == 'x16c5':
    T, N = 10, 32 * 32 * 260 * 346
    alg = {'neuron_fwd_packed': int(2.25 * T * N), 'neuron_fwd_skip_packed': int(2.5 * T * N), 'neuron_bwd_lr': int((6 + 36 / 32) * T * N)}
    res = {'workload': 'BASELINE config 5 per-GPU layer shape: fp16 activations, T = 10, batch 32, 32 x 260 x 346 (9.2e8 updates), tools/pmc_target.py x16c5',
           'mode': mode, 'shape': {'T': 10, 'batch': 32, 'dtype': 'f16'},
           'algorithmic_bytes_per_update': {'neuron_fwd_packed': 2.25, 'neuron_fwd_skip_packed': 2.5, 'neuron_bwd_lr': 6 + 36 / 32},
           'algorithmic_bytes_per_launch': alg,
           'note': 'FETCH_SIZE (KiB) doubled per the gfx950 note in MI355X_MICROARCH.md; WRITE_SIZE (KiB) as reported; separate --pmc passes; the forward leaves the '
                   'membrane unwritten (v_last == NULL, ABI 10), as the training step runs it'}
    for k in alg:
        if k in fetch and k in write:
            hbm = (2 * fetch[k] + write[k]) * 1024
            res[k] = {'FETCH_SIZE_KiB_raw': fetch[k], 'WRITE_SIZE_KiB_raw': write[k], 'dispatches': [nf[k], nw[k]], 'hbm_read_bytes': int(2 * fetch[k] * 1024),
                      'hbm_write_bytes': int(write[k] * 1024), 'hbm_bytes_per_launch': int(hbm), 'ratio_to_algorithmic': round(hbm / alg[k], 4)}
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from stereospike_amd import _lib
        res['source'] = {'lib_source_hash': _lib.source_hash(), 'tree_source_hash': _lib.tree_source_hash(), 'git_head': os.environ.get('SS_GIT_HEAD', 'unknown')}
    except Exception as e:                                                # noqa: BLE001
        res['source'] = {'error': repr(e)}
    print(json.dumps(res, indent=1))
    sys.exit(0)
if mode == 'x16':
    # the 16-bit activation modes' own kernels (bf16), same shapes as the fp32 table below: algorithmic bytes with 2-byte activations, ONE weight term, one box plane
    planes1 = 80 * 4 * 1 * 278 * 364 * 8 * 2
    alg = {'neuron_fwd_x16_packed': int(2.25 * T * N), 'neuron_bwd_x16_lr': int((6 + 36 / 32) * T * N),
           'dense_conv_s1_fwd': 4 * 80 * 260 * 346 * 4 + 2 * 80 * 260 * 346 * 32 + 4 * 25 * 4 * 32,
           'dense_conv_s1_wgrad': 2 * 80 * 260 * 346 * 32 + 4 * 80 * 260 * 346 * 4 + 4 * 1024 * 4 * 32 * 128,
           'spike_conv_fwd': 80 * 260 * 346 * 32 // 4 + 2 * 80 * 130 * 173 * 64 + 2 * 25 * 32 * 64,
           'conv_s2_dgrad': 2 * (80 * 130 * 173 * 64 + 80 * 260 * 346 * 32) + 2 * 25 * 32 * 64,
           'upconv_sub': 80 * 130 * 173 * 64 // 4 + 2 * 80 * 260 * 346 * 32 + 2 * 25 * 18 * 512 * 4,
           'upconv_boxsum': 2 * 80 * 260 * 346 * 32 + planes1, 'upconv_box_dgrad': planes1 + 2 * 80 * 130 * 173 * 64 + 2 * 26 * 64 * 32,
           'upconv_box_wgrad': planes1 + 2 * 80 * 130 * 176 * 64 + 4 * 128 * 25 * 32 * 64}
    res = {'workload': 'config-3 shapes (80 frames), bf16 activations, tools/pmc_target.py x16', 'mode': mode, 'algorithmic_bytes_per_launch': alg,
           'note': 'FETCH_SIZE (KiB) doubled per the gfx950 note in MI355X_MICROARCH.md; WRITE_SIZE (KiB) as reported; separate --pmc passes'}
    for k in alg:
        if k in fetch and k in write:
            hbm = (2 * fetch[k] + write[k]) * 1024
            res[k] = {'FETCH_SIZE_KiB_raw': fetch[k], 'WRITE_SIZE_KiB_raw': write[k], 'dispatches': [nf[k], nw[k]], 'hbm_read_bytes': int(2 * fetch[k] * 1024),
                      'hbm_write_bytes': int(write[k] * 1024), 'hbm_bytes_per_launch': int(hbm), 'ratio_to_algorithmic': round(hbm / alg[k], 4)}
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from stereospike_amd import _lib
        res['source'] = {'lib_source_hash': _lib.source_hash(), 'tree_source_hash': _lib.tree_source_hash(), 'git_head': os.environ.get('SS_GIT_HEAD', 'unknown')}
    except Exception as e:                                                # noqa: BLE001
        res['source'] = {'error': repr(e)}
    print(json.dumps(res, indent=1))
    sys.exit(0)
per_update = {'neuron_fwd': 8 if mode == 'rc' else 12, 'neuron_bwd': 16 if mode == 'rc' else 12}
res = {'workload': f'B16 x T5 x 32x260x346 layer (config-3 bottom), IF, fp32, tools/pmc_target.py {mode}',
       'mode': mode, 'shape': {'T': 5, 'batch': 16, 'dtype': 'f32'}, 'algorithmic_bytes_per_update': per_update,
       'algorithmic_bytes_per_launch': {k: v * T * N for k, v in per_update.items()},
       'o_n_terms_bytes': {'neuron_fwd': 4 * N, 'neuron_bwd': 0},
       'note': 'FETCH_SIZE (KiB) doubled per the gfx950 note in MI355X_MICROARCH.md; WRITE_SIZE (KiB) as reported; '
               'separate --pmc passes; the dense forward also writes v_last (4 B x N, an O(N) term outside the per-update figure); the packed forwards leave it unwritten (ABI 10), as the training step runs them'}
alg = {k: v * T * N for k, v in per_update.items()}
alg['neuron_fwd_packed'] = int(4.25 * T * N)                        # x 4 B + 2-bit packed output 0.25 B per update (no dense output)
alg['neuron_bwd_lr'] = int((12 + 36 / 32) * T * N)                   # g_out, x, g_x + the head's rank-9 pair: 9 floats per 32-channel pixel
alg['spike_conv_fwd'] = 80 * 260 * 346 * 32 // 4 + 4 * 80 * 130 * 173 * 64 + 2 * 3 * 25 * 32 * 64        # conv1: packed spikes in + fp32 out + split weights
alg['dense_conv_s1_fwd'] = 4 * 80 * 260 * 346 * (4 + 32) + 4 * 25 * 4 * 32                                # bottom: voxel input + fp32 out + weights
alg['neuron_fwd_skip_packed'] = int(4.5 * T * N)                    # x 4 B + packed skip 0.25 B + packed output 0.25 B per update (deconv1 since the packed head)
alg['conv_s2_dgrad'] = 4 * (80 * 130 * 173 * 64 + 80 * 260 * 346 * 32) + 2 * 3 * 25 * 32 * 64       # conv1: g in + g_x out + split weights
alg['dense_conv_s1_wgrad'] = 4 * 80 * 260 * 346 * (32 + 4) + 4 * 1024 * 4 * 32 * 128                 # bottom: g + x in, wavefront partials out
alg['head_proj_packed'] = 80 * 260 * 346 * (32 // 4 + 9 * 4) + 4 * 32 * 9                            # packed spikes in + P out
alg['head_wgrad_packed'] = 80 * 260 * 346 * (32 // 4 + 9 * 4) + 4 * 4096 * 32 * 9                    # packed spikes + g_P in, partials out
# round 4, deconv1's backward on the box-sum image: 278 x 364 distinct (vertical, horizontal) ranges per frame (id 0 = empty), three bf16 planes of 32 channels
planes = 80 * 4 * 3 * 278 * 364 * 8 * 2
alg['upconv_boxsum'] = 4 * 80 * 260 * 346 * 32 + planes                                               # g_y in, planes out
alg['upconv_box_dgrad'] = planes + 4 * 80 * 130 * 173 * 64 + 2 * 3 * 26 * 64 * 32                     # planes in, g_x out, split weights
alg['upconv_box_wgrad'] = planes + 2 * 80 * 130 * 176 * 64 + 4 * 128 * 25 * 32 * 64                   # planes + bf16 transposed spikes in, 128 slices of partials out
alg['upconv_sub'] = 80 * 130 * 173 * 64 // 4 + 4 * 80 * 260 * 346 * 32 + 2 * 25 * 27 * 512 * 4            # deconv1 forward: packed spikes in, fp32 out, merged weights
res['algorithmic_bytes_per_launch'] = alg
for k in ('upconv_sub', 'upconv_boxsum', 'upconv_box_dgrad', 'upconv_box_wgrad', 'conv_s2_dgrad', 'dense_conv_s1_wgrad', 'head_proj_packed', 'head_wgrad_packed', 'neuron_fwd_skip_packed', 'spike_conv_fwd', 'dense_conv_s1_fwd', 'neuron_fwd', 'neuron_bwd', 'neuron_bwd_lr', 'neuron_fwd_packed'):
    if k in fetch and k in write:
        hbm = (2 * fetch[k] + write[k]) * 1024
        res[k] = {'FETCH_SIZE_KiB_raw': fetch[k], 'WRITE_SIZE_KiB_raw': write[k], 'dispatches': [nf[k], nw[k]],
                  'hbm_read_bytes': int(2 * fetch[k] * 1024), 'hbm_write_bytes': int(write[k] * 1024),
                  'hbm_bytes_per_launch': int(hbm), 'ratio_to_algorithmic': round(hbm / alg[k], 4)}
if 'neuron_fwd' in res:
    res['neuron_fwd_train_bytes_per_launch'] = res['neuron_fwd']['hbm_bytes_per_launch']
# which kernels produced these counters: the source hash compiled into the library the target loaded (csrc/Makefile: sha256 over the .hip
# units + headers), per-file hashes of the tree, and the commit the measurement script was launched from (SS_GIT_HEAD: the GPU box holds no .git)
try:
    import hashlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from stereospike_amd import _lib
    csrc = os.path.join(os.path.dirname(_lib.__file__), 'csrc')
    res['source'] = {'lib_source_hash': _lib.source_hash(), 'tree_source_hash': _lib.tree_source_hash(), 'git_head': os.environ.get('SS_GIT_HEAD', 'unknown'),
                     'csrc_sha256_16': {f: hashlib.sha256(open(os.path.join(csrc, f), 'rb').read()).hexdigest()[:16]
                                        for f in sorted(os.listdir(csrc)) if f.endswith(('.hip', '.hpp'))}}
except Exception as e:                                                    # noqa: BLE001 — the counters are still worth having
    res['source'] = {'error': repr(e)}
print(json.dumps(res, indent=1))

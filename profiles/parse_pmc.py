#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc CSVs to per-launch HBM bytes for the fused kernels (see collect_pmc.sh)."""
import csv
import glob
import json
import os
import sys

out_dir = sys.argv[1]


def mean_counter(sub, counter):
    vals = {}
    for path in glob.glob(os.path.join(out_dir, sub, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get('Counter_Name') != counter:
                continue
            name = row.get('Kernel_Name', '')
            key = 'neuron_fwd' if 'neuron_fwd_kernel' in name else 'neuron_bwd' if 'neuron_bwd_kernel' in name else None
            if key:
                vals.setdefault(key, []).append(float(row['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in vals.items()}, {k: len(v) for k, v in vals.items()}


fetch, nf = mean_counter('fetch', 'FETCH_SIZE')
write, nw = mean_counter('write', 'WRITE_SIZE')
T, N = 5, 16 * 32 * 260 * 346
res = {'workload': 'B16 x T5 x 32x260x346 layer, IF, fp32', 'algorithmic_bytes_per_launch': 12 * T * N,
       'note': 'FETCH_SIZE (KiB) doubled per the gfx950 note in MI355X_MICROARCH.md; WRITE_SIZE (KiB) as reported'}
for k in ('neuron_fwd', 'neuron_bwd'):
    if k in fetch and k in write:
        res[k] = {'FETCH_SIZE_KiB_raw': fetch[k], 'WRITE_SIZE_KiB_raw': write[k], 'dispatches': [nf[k], nw[k]],
                  'hbm_bytes_per_launch': int((2 * fetch[k] + write[k]) * 1024),
                  'ratio_to_algorithmic': round((2 * fetch[k] + write[k]) * 1024 / (12 * T * N), 3)}
if 'neuron_fwd' in res:
    res['neuron_fwd_train_bytes_per_launch'] = res['neuron_fwd']['hbm_bytes_per_launch']
print(json.dumps(res, indent=1))

#!/bin/bash
# SQ counters of the box-sum backward kernels at the deconv1 / deconv2 geometries (one launch each): where do the wave cycles go?
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04/pmc_box
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ONLY=${ONLY:-deconv1,deconv2} ROUNDS=1 REPS=1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT \
  --output-format csv -d "$OUT/a" -o pmc -- python "$REPO/tools/bench_upconv_bwd.py" > "$OUT/a.log" 2>&1
ONLY=${ONLY:-deconv1,deconv2} ROUNDS=1 REPS=1 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES \
  --output-format csv -d "$OUT/b" -o pmc -- python "$REPO/tools/bench_upconv_bwd.py" > "$OUT/b.log" 2>&1
python - <<PY
import csv, glob, collections
for sub in ('a', 'b'):
    for f in glob.glob('$OUT/' + sub + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'box' not in k and 'gemm6' not in k and 'spike_wgrad_kernel' not in k: continue
            k = k.split('(')[0][-60:]
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        for k, d in agg.items():
            print(sub, k, {c: f'{v:.3g}' for c, v in d.items()})
PY

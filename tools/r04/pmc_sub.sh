#!/bin/bash
# SQ counters of the sub-pixel forward kernel at the deconv1 geometry: where do the wave cycles go?
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04/pmc_sub
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ONLY=${ONLY:-deconv1} ROUNDS=1 REPS=1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT \
  --output-format csv -d "$OUT/a" -o pmc -- python "$REPO/tools/r04/bench_sub_fwd.py" > "$OUT/a.log" 2>&1
ONLY=${ONLY:-deconv1} ROUNDS=1 REPS=1 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES \
  --output-format csv -d "$OUT/b" -o pmc -- python "$REPO/tools/r04/bench_sub_fwd.py" > "$OUT/b.log" 2>&1
ONLY=${ONLY:-deconv1} ROUNDS=1 REPS=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/c" -o pmc -- python "$REPO/tools/r04/bench_sub_fwd.py" > "$OUT/c.log" 2>&1
python - <<PY
import csv, glob, collections
for sub in ('a', 'b', 'c'):
    for f in glob.glob('$OUT/' + sub + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'upconv_sub_fwd' not in k and 'upconv_fused2' not in k: continue
            k = k.split('(')[0][-50:]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
        for k, d in agg.items():
            print(sub, k, len(n[k]), {c: f'{v / len(n[k]):.3g}' for c, v in d.items()})
PY

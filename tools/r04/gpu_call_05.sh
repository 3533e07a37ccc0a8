#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_01_kernels.py -q -m gpu -k "upconv_box" > gpurun_out/r04/pytest_box2.log 2>&1
tail -3 gpurun_out/r04/pytest_box2.log
ONLY=deconv1,deconv2 timeout 600 python tools/bench_upconv_bwd.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r04/bench_box_bwd_v2.log
grep -E "^deconv|box:" gpurun_out/r04/bench_box_bwd_v2.log
bash tools/r04/pmc_box.sh > /dev/null 2>&1
python - <<'PY'
import csv, collections, re
for sub in ('a','b'):
    f=f'gpurun_out/r04/pmc_box/{sub}/pmc_counter_collection.csv'
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        m=re.search(r'(upconv_box\w+<[^>]*>)',r['Kernel_Name'])
        if not m: continue
        k=m.group(1)
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
    for k,d in agg.items():
        n=len(cnt[k]); print(sub,k,n,{c:f'{v/n:.3g}' for c,v in d.items()})
PY

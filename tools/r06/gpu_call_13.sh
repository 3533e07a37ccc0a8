#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c13; mkdir -p $O
# the DP path at one RCCL rank: process group, bucketed reducer with post-accumulate hooks, barrier + MAX timing, sustained leg's MIN all-reduce
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --force-dp --sustained-seconds 4 > $O/bench_force_dp.json 2> $O/bench_force_dp.err; echo "rc $?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06/c13/bench_force_dp.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['rccl_ranks'], j['sustained'])
PY
tail -3 $O/bench_force_dp.err

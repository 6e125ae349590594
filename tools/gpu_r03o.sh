#!/bin/bash
# round 3, call O: full GPU suite on the tree, N>1 dry runs (ranks share the GPU, gloo-staged exchange), 1-rank native sharded lines
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03o; mkdir -p $O; cd $R
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED" $O/pytest_gpu.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for ex in sync overlap; do
  for n in 2 4; do
    DFH_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29900+n)) bench.py --gpus $n --steps 10 --warmup 3 --min-time 0.05 --ids 2000000 --exchange $ex > $O/dry_${ex}_$n.json 2> $O/dry_${ex}_$n.err
    python -c "
import json
try:
  d=json.loads(open('$O/dry_${ex}_$n.json').read().strip().splitlines()[-1])
  print('dry $ex $n', round(d['value']/1e6,2), round(d['ms_per_step'],4), d['config']['exchange'][:8], d['config']['key_ranges'], 'logloss', round(d['train_logloss_per_example'],4))
except Exception as e: print('dry $ex $n ERR', e); print(open('$O/dry_${ex}_$n.err').read()[-1500:])"
  done
done
for ex in sync overlap; do
  timeout 200 python bench.py --force-sharded --exchange $ex --steps 200 --warmup 20 --min-time 1 > $O/bench_sharded_w1_$ex.json 2> $O/bench_sharded_w1_$ex.err
  python -c "
import json
d=json.loads(open('$O/bench_sharded_w1_$ex.json').read().strip().splitlines()[-1]); print('w1 $ex', round(d['value']/1e6,2), round(d['ms_per_step'],4), d.get('stage_ms_per_step'))"
done

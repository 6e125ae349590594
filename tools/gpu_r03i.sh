#!/bin/bash
# round 3, call I: the overlapped native exchange (callback transport, ranks share the GPU), host CLI, N>1 bench dry runs
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03i; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests/test_shard_native.py tests/test_host_cpp.py tests/test_sharded_gpu_ranks.py -m gpu -q -x ) > $O/pytest_shard.log 2>&1
tail -5 $O/pytest_shard.log | cut -c1-300; grep -E "^E |^FAILED" $O/pytest_shard.log | head -30
for ex in sync overlap; do
  for n in 2 4; do
    DFH_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29900+n)) bench.py --gpus $n --steps 10 --warmup 3 --min-time 0.05 --ids 2000000 --exchange $ex > $O/dry_${ex}_$n.json 2> $O/dry_${ex}_$n.err
    python -c "
import json
try:
  d=json.loads(open('$O/dry_${ex}_$n.json').read().strip().splitlines()[-1])
  print('dry $ex $n', round(d['value']/1e6,2), round(d['ms_per_step'],4), d['config']['exchange'][:8], d.get('stage_ms_per_step'), 'logloss', round(d['train_logloss_per_example'],4))
except Exception as e: print('dry $ex $n ERR', e); print(open('$O/dry_${ex}_$n.err').read()[-1500:])"
  done
done
timeout 200 python bench.py --force-sharded --steps 200 --warmup 20 --min-time 1 > $O/bench_sharded_w1.json 2> $O/bench_sharded_w1.err
python -c "
import json
d=json.loads(open('$O/bench_sharded_w1.json').read().strip().splitlines()[-1]); print('w1', round(d['value']/1e6,2), d['ms_per_step'], d.get('stage_ms_per_step'))"

#!/bin/bash
# chained steps (dfh_batch_chain): bit-exactness against the unchained steps, then the bench A/B
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03an; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -q -x -k "chained or pipelined" ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^E  " $O/pytest_gpu.log | head -30
run() {  # name args...
  n=$1; shift
  timeout 200 python bench.py --cpu-batches 0 --min-time 0.5 --no-secondary "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-14s' % '$n', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['kernel_ms_per_step'].items()}, 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round(d['roofline_backward']['avg_launch_ms'],4), 'logloss', round(d['train_logloss_per_example'],6))
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-800:])"
}
run plain; run chain --chain; run plain_b; run chain_b --chain; run chain_later --chain --later-epoch; run plain_later --later-epoch

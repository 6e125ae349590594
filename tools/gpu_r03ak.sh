#!/bin/bash
# forward: 4 against 5 row loads in flight per lane, pipelined / serial / C5 slice
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03ak; mkdir -p $O; cd $R
run() {  # name args...
  n=$1; shift
  timeout 300 python bench.py --cpu-batches 0 --min-time 1.0 --no-secondary "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-18s' % '$n', round(d['value']/1e6,2), round(d['ms_per_step'],4), 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round(d['roofline_backward']['avg_launch_ms'],4))
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-400:])"
}
run d5; run d4 --ctx-option fwd_depth=4; run d5_b; run d4_b --ctx-option fwd_depth=4; run d5_c; run d4_c --ctx-option fwd_depth=4
run d5_np --no-pipeline; run d4_np --no-pipeline --ctx-option fwd_depth=4
run d5_c5 --preset c5-slice; run d4_c5 --preset c5-slice --ctx-option fwd_depth=4

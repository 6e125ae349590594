#!/bin/bash
# launch tuning inside the PIPELINED step (the role block counts were chosen with the kernel alone): forward depth, update role blocks
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03aj; mkdir -p $O; cd $R
run() {  # name args...
  n=$1; shift
  timeout 200 python bench.py --cpu-batches 0 --min-time 0.5 --no-secondary "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-18s' % '$n', round(d['value']/1e6,2), round(d['ms_per_step'],4), 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round(d['roofline_backward']['avg_launch_ms'],4))
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-400:])"
}
run base
run fwd4 --ctx-option fwd_depth=4
run fwd8 --ctx-option fwd_depth=8
run fwd10 --ctx-option fwd_depth=10
run few512 --ctx-option upd_few_blocks=512
run few2048 --ctx-option upd_few_blocks=2048
run single2048 --ctx-option upd_single_blocks=2048
run single8192 --ctx-option upd_single_blocks=8192
run hot256 --ctx-option upd_hot_blocks=256
run hot1024 --ctx-option upd_hot_blocks=1024
run mid256 --ctx-option upd_mid_blocks=256
run mid1024 --ctx-option upd_mid_blocks=1024
run base_b

#!/bin/bash
# key-index probe inside the Localizer's emit pass (dfh_localize_lookup): tests, A/B against the probe as a launch of its own
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03ai; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -q -x -k "pipelined or host_cpp or cli or local" ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^E  " $O/pytest_gpu.log | head -20
run() {  # name args...
  n=$1; shift
  timeout 200 python bench.py --cpu-batches 0 --min-time 0.5 --no-secondary "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-14s' % '$n', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['kernel_ms_per_step'].items()}, 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round(d['roofline_backward']['avg_launch_ms'],4))
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-600:])"
}
run separate --no-fused-probe; run fused; run separate_b --no-fused-probe; run fused_b; run fused_later --later-epoch; run separate_later --no-fused-probe --later-epoch

#!/bin/bash
# round 3, call H: which hardware queue the preparation stream lands on (gaps on the main stream), env knobs
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03h; mkdir -p $O; cd $R
run() {  # name args...
  n=$1; shift
  timeout 200 python bench.py --cpu-batches 0 --min-time 0.4 --no-secondary "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-16s' % '$n', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['kernel_ms_per_step'].items()}, 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round(d['roofline_backward']['avg_launch_ms'],4))
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-600:])"
}
for k in 0 1 2 3 4; do run skip$k --ctx-option prep_queue_skip=$k; done
run prio0 --ctx-option prep_priority=0
run prio0_skip2 --ctx-option prep_priority=0 --ctx-option prep_queue_skip=2
GPU_MAX_HW_QUEUES=8 run hwq8
GPU_MAX_HW_QUEUES=8 run hwq8_skip3 --ctx-option prep_queue_skip=3
GPU_MAX_HW_QUEUES=2 run hwq2
HIP_FORCE_DEV_KERNARG=1 run devkernarg
DEBUG_CLR_USE_STDMUTEX_IN_AMD_MONITOR=1 run stdmutex
run base2

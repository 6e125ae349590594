#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r02ac; mkdir -p $O; cd $R
run() { # name, env...
  n=$1; shift
  env "$@" timeout 200 python bench.py --cpu-batches 0 --min-time 0.3 > $O/$n.json 2> $O/$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
  print('%-28s' % '$n', round(d['value']/1e6,2), round(d['ms_per_step'],4), 'enq', round(d['host_enqueue_ms_per_step'],4), 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round(d['roofline_backward']['avg_launch_ms'],4))
except Exception as e: print('$n ERR', e); print(open('$O/$n.err').read()[-400:])"
}
run base A=1
run devkernarg HIP_FORCE_DEV_KERNARG=1
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq2 GPU_MAX_HW_QUEUES=2
run nointerrupt HSA_ENABLE_INTERRUPT=0
run base2 A=1

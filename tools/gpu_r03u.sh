#!/bin/bash
# runtime switches against the main-stream dispatch gaps (second queue busy): one bench line per setting
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03u; mkdir -p $O; cd $R
run() {  # name env... -- args...
  n=$1; shift
  envs=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
  [ "$1" == "--" ] && shift
  env "${envs[@]}" timeout 200 python bench.py --cpu-batches 0 --min-time 0.5 --no-secondary "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-22s' % '$n', round(d['value']/1e6,2), round(d['ms_per_step'],4), 'enq', round(d['host_enqueue_ms_per_step'],4), 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round(d['roofline_backward']['avg_launch_ms'],4))
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-600:])"
}
run default
run sysscope0 ROC_SYSTEM_SCOPE_SIGNAL=0
run optflush0 AMD_OPT_FLUSH=0
run cpwait1 GPU_STREAMOPS_CP_WAIT=1
run dynq1 DEBUG_HIP_DYNAMIC_QUEUES=1
run dynq0 DEBUG_HIP_DYNAMIC_QUEUES=0
run activewait ROC_ACTIVE_WAIT_TIMEOUT=1000
run cpuwait ROC_CPU_WAIT_FOR_SIGNAL=1
run directdisp0 AMD_DIRECT_DISPATCH=0
run flushexec GPU_FLUSH_ON_EXECUTION=1
run notiming -- --no-timing
run prio_hi -- --ctx-option prep_priority=1
run default_b

#!/bin/bash
# where the main stream idles inside a step: DFH_GAP_TRACE=1 (every step's lookup / forward / update dispatch carries its own
# start / stop events; dfh_ctx_get_timing prints the time from the end of one to the start of the next).  usage: gpu_gaps.sh <tag>
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-gaps}; mkdir -p $O; cd $R
for mode in "" "--no-relocalize" "--no-pipeline" "--later-epoch"; do
  n=$(echo "x$mode" | tr -d ' -')
  DFH_GAP_TRACE=1 DFH_TIMING_EVERY=1 timeout 300 python bench.py --cpu-batches 0 --min-time 1 --no-secondary $mode > $O/b_$n.json 2> $O/b_$n.err
  echo "== [$mode]"; grep "gap trace" $O/b_$n.err | sort | uniq -c | sort -rn | head -8 | cut -c1-160
  python -c "
import json
d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
print('   ', round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,1), 'us/step; live fwd/upd', round(d['roofline']['avg_launch_ms']*1e3,1), round(d['roofline_backward']['avg_launch_ms']*1e3,1))"
done

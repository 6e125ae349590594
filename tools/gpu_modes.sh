#!/bin/bash
# bench.py lines of ONE library over several modes on ONE box, interleaved twice (A B A B): usage: gpu_modes.sh <tag> "<mode1>|<mode2>|..."
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$1; shift; IFS='|' read -ra MODES <<< "$1"; shift; mkdir -p $O; cd $R
for rep in 1 2; do
  i=0
  for mode in "${MODES[@]}"; do
    i=$((i+1)); n=m${i}_r$rep
    timeout 300 python bench.py --cpu-batches 0 --min-time 1 --no-secondary $mode > $O/b_$n.json 2> $O/b_$n.err
    python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('[%s]' % '$mode', round(d['value']/1e6,2), round(d['ms_per_step'],4), 'live fwd/upd', round(d['roofline']['avg_launch_ms'],4), round(d['roofline_backward']['avg_launch_ms'],4))
except Exception as e: print('[$mode] ERR', e); print(open('$O/b_$n.err').read()[-600:])"
  done
done

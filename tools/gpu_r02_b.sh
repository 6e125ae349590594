#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r02b; mkdir -p $O; cd $R
for v in "--prep-lookup" "--prep-streams 1" "--prep-streams 3" "--prep-streams 1 --prep-lookup" "--prep-streams 3 --prep-lookup"; do
  n=$(echo $v | tr -d ' -'); timeout 200 python bench.py --cpu-batches 0 $v > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
print('$v', round(d['value']/1e6,2), d['ms_per_step'], {k:round(x,4) for k,x in d['kernel_ms_per_step'].items()}, d['roofline']['avg_launch_ms'], d['roofline_backward']['avg_launch_ms'])"
done

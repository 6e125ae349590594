#!/bin/bash
# how do the kernels of the step scale with the minibatch size (is the Localizer latency-bound: would two minibatches per launch pay?)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03q; mkdir -p $O; cd $R
for r in 10000 20000 30000; do
  for mode in "--no-pipeline" ""; do
    timeout 200 python bench.py --cpu-batches 0 --min-time 0.3 --no-secondary $mode --rows $r > $O/b_$r$mode.json 2> $O/b_$r$mode.err
    python -c "
import json
d=json.loads(open('$O/b_$r$mode.json').read().strip().splitlines()[-1]); print($r, '$mode', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['kernel_ms_per_step'].items()}, 'U', round(d['config']['unique_keys_per_batch']))"
  done
done

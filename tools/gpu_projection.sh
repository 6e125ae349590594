#!/bin/bash
# the projection lines of DESIGN §6: one GPU as rank r of W (loop-back transport, wires modelled) — C4 at W = 2 / 4 / 8 and C5 at W = 8
# usage: gpurun -- bash tools/gpu_projection.sh [tag]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${1:-proj}; mkdir -p $O; cd $R
run() {  # name args...
  n=$1; shift
  timeout 1500 python bench.py --emulate-rank auto --cpu-batches 0 --min-time 1 "$@" > $O/emul_$n.json 2> $O/emul_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/emul_$n.json').read().strip().splitlines()[-1])
  print('$n', 'ms/step', round(d['ms_per_step'],4), {k:round(v/1e6,1) for k,v in d['projected_examples_per_sec'].items()}, 'ranks', d.get('emulated_ranks'))
except Exception as e: print('$n ERR', e); print(open('$O/emul_$n.err').read()[-600:])"
}
run c4_w2 --emulate-world 2
run c4_w4 --emulate-world 4
run c4_w8 --emulate-world 8
run c5_w8 --emulate-world 8 --preset c5-slice

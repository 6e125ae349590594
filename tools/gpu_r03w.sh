#!/bin/bash
# tags (pos << tb | row bits) through the sort instead of the rowid gather in k_loc_emit: Localizer / fused-step tests, then A/B
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03w; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -q -x -k "local or Local or fused or step or parity" ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^E  " $O/pytest_gpu.log | head -20
cp $R/difacto_amd/libdifacto_hip.so $R/tools/var_tags.so
run() {  # name variant args...
  n=$1; v=$2; shift 2
  cp $R/tools/var_$v.so $R/difacto_amd/libdifacto_hip.so
  timeout 200 python bench.py --cpu-batches 0 --min-time 0.5 --no-secondary "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-14s' % '$n', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['kernel_ms_per_step'].items()}, 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round(d['roofline_backward']['avg_launch_ms'],4))
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-600:])"
}
run base base; run tags tags; run base_np base --no-pipeline; run tags_np tags --no-pipeline; run base_b base; run tags_b tags
cp $R/tools/var_tags.so $R/difacto_amd/libdifacto_hip.so

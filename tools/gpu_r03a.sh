#!/bin/bash
# round 3, call A: parity suite on the new fused update kernel, then A/B of its builds on ONE box
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03a; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log | cut -c1-300; grep -E "^E |^FAILED" $O/pytest_gpu.log | head -30
cp $R/difacto_amd/libdifacto_hip.so /tmp/keep.so
run() {  # name variant args...
  n=$1; v=$2; shift 2
  cp $R/tools/var_$v.so $R/difacto_amd/libdifacto_hip.so
  timeout 200 python bench.py --cpu-batches 0 --min-time 0.3 "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-14s' % '$n', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['kernel_ms_per_step'].items()}, 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round(d['roofline_backward']['avg_launch_ms'],4), 'pen', d.get('train_logloss_per_example'))
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-600:])"
}
run old256 old256 --ctx-option upd_kernel=0
run old256_np old256 --ctx-option upd_kernel=0 --no-pipeline
run old64_np cur --ctx-option upd_kernel=0 --no-pipeline
for v in cur w4 w6 w8 t512w4; do
  run ${v} $v
  run ${v}_np $v --no-pipeline
done
run cur_ev0 cur --ctx-option event_flags=0
run cur_later cur --later-epoch
cp /tmp/keep.so $R/difacto_amd/libdifacto_hip.so

#!/bin/bash
# count / scatter in smaller blocks (they fit beside the update kernel's 5 waves per SIMD instead of waiting for its tail):
# does the Localizer then finish inside the update's window and leave lookup + forward alone?
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03v; mkdir -p $O; cd $R
cp $R/difacto_amd/libdifacto_hip.so /tmp/keep.so
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
run base base
run t256a t256a
run t256a_hi t256a --ctx-option prep_priority=1
run t256a_mid t256a --ctx-option prep_priority=0
run t256b t256b
run t256b_hi t256b --ctx-option prep_priority=1
run t512 t512
run t512_hi t512 --ctx-option prep_priority=1
run t256a_np t256a --no-pipeline
run base_b base
cp /tmp/keep.so $R/difacto_amd/libdifacto_hip.so

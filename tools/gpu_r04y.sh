#!/bin/bash
# round 4: InitV by the whole lane group at the end of upd_apply (difacto_amd/libdifacto_hip.so) against the serial loop of the
# group's first lane (tools/var_base_initv_serial.so): cold steps (empty table) and the warm default line, alternating
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04y; mkdir -p $O; cd $R; rm -f $O/summary2.txt
cp $R/difacto_amd/libdifacto_hip.so /tmp/keep.so
line() {  # name args...
  n=$1; shift
  timeout 300 python bench.py --cpu-batches 0 --no-secondary "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-26s' % '$n', round(d['value']/1e6,2), 'M ex/s', round(d['ms_per_step'],4), 'ms/step')
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-800:])" | tee -a $O/summary2.txt
}
COLD="--no-prefill --warmup 0 --max-reps 1 --min-time 0 --no-timing"
for rep in 1 2; do
  for v in new base; do
    if [ $v = base ]; then cp $R/tools/var_base_initv_serial.so $R/difacto_amd/libdifacto_hip.so; else cp /tmp/keep.so $R/difacto_amd/libdifacto_hip.so; fi
    line warm_${v}_$rep --min-time 1.5
    [ $rep = 1 ] && line cold64_$v $COLD --steps 64
    [ $rep = 1 ] && line cold256_$v $COLD --steps 256
  done
done
cp /tmp/keep.so $R/difacto_amd/libdifacto_hip.so
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3 | tee -a $O/summary2.txt

#!/bin/bash
# A/B of library variants on ONE box: tools/var_<name>.so swapped in for difacto_amd/libdifacto_hip.so
# usage: gpu_variants.sh <outdir-name> "<mode1>|<mode2>|..." variant...
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$1; shift; IFS='|' read -ra MODES <<< "$1"; shift; mkdir -p $O; cd $R
cp $R/difacto_amd/libdifacto_hip.so /tmp/keep.so
for v in "$@"; do
  cp $R/tools/var_$v.so $R/difacto_amd/libdifacto_hip.so
  for mode in "${MODES[@]}"; do
    n=$(echo "${v}_x$mode" | tr -d ' -')
    timeout 200 python bench.py --cpu-batches 0 --min-time 0.3 $mode > $O/b_${n}.json 2> $O/b_${n}.err
    python -c "
import json
try:
  d=json.loads(open('$O/b_${n}.json').read().strip().splitlines()[-1])
  print('%-8s [%s]' % ('$v', '$mode'), round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['kernel_ms_per_step'].items()}, 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round(d['roofline_backward']['avg_launch_ms'],4))
except Exception as e: print('$v [$mode] ERR', e); print(open('$O/b_${n}.err').read()[-600:])"
  done
done
cp /tmp/keep.so $R/difacto_amd/libdifacto_hip.so

#!/bin/bash
# round 4: launch options and unroll depths of k_update_fused re-swept after the {u, beg, end} list entries (same box)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04r; mkdir -p $O; cd $R
cp $R/difacto_amd/libdifacto_hip.so /tmp/keep.so
line() {  # name args...
  n=$1; shift
  timeout 300 python bench.py --cpu-batches 0 --min-time 1 --no-secondary "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-26s' % '$n', round(d['value']/1e6,2), round(d['ms_per_step'],4), 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round((d.get('roofline_backward') or {}).get('avg_launch_ms',0),4))
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-800:])"
}
line base
line few512 --ctx-option upd_few_blocks=512
line few2048 --ctx-option upd_few_blocks=2048
line mid256 --ctx-option upd_mid_blocks=256
line mid1024 --ctx-option upd_mid_blocks=1024
line hot256 --ctx-option upd_hot_blocks=256
line fwd4 --ctx-option fwd_depth=4
for v in fd8 fd2 ud4 ur3; do
  cp $R/tools/var_$v.so $R/difacto_amd/libdifacto_hip.so
  line var_$v
done
cp /tmp/keep.so $R/difacto_amd/libdifacto_hip.so
line base_again

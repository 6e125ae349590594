#!/bin/bash
# round 3, call F: per-role time of k_update_fused against the block counts of the role
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
cp $R/difacto_amd/libdifacto_hip.so /tmp/keep.so
run() {  # name variant args...
  n=$1; v=$2; shift 2
  cp $R/tools/var_$v.so $R/difacto_amd/libdifacto_hip.so
  timeout 200 python bench.py --cpu-batches 0 --min-time 0.15 --no-pipeline --ctx-option upd_interleave=0 "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-22s' % '$n', 'bwd', round(d['roofline_backward']['avg_launch_ms']*1e3,1))
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-600:])"
}
opt() { for kv in "$@"; do echo -n "--ctx-option $kv "; done; }
for nb in 128 256 512 1024 2048; do run hot_$nb role1 $(opt upd_hot_blocks=$nb); done
for nb in 256 512 1024 2048 4096; do run mid_$nb role2 $(opt upd_mid_blocks=$nb); done
for nb in 256 512 1024 2048 4096 8192; do run few_$nb role4 $(opt upd_few_blocks=$nb); done
for nb in 640 1280 2500 4096; do run single_$nb role8 $(opt upd_single_blocks=$nb); done
run list_default role7
run list_m3 role7 $(opt upd_hot_blocks=512 upd_mid_blocks=512 upd_few_blocks=1024)
cp /tmp/keep.so $R/difacto_amd/libdifacto_hip.so

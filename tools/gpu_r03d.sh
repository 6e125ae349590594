#!/bin/bash
# round 3, call D: block interleave and list-role block counts of k_update_fused (runtime options), one box
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03d; mkdir -p $O; cd $R
run() {  # name args...
  n=$1; shift
  timeout 200 python bench.py --cpu-batches 0 --min-time 0.25 "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-22s' % '$n', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['kernel_ms_per_step'].items()}, 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round(d['roofline_backward']['avg_launch_ms'],4))
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-600:])"
}
opt() { for kv in "$@"; do echo -n "--ctx-option $kv "; done; }
run il0_np --no-pipeline $(opt upd_interleave=0)
run il2_np --no-pipeline $(opt upd_interleave=2)
run il3_np --no-pipeline $(opt upd_interleave=3)
run il4_np --no-pipeline $(opt upd_interleave=4)
run il2_more_np --no-pipeline $(opt upd_interleave=2 upd_hot_blocks=512 upd_mid_blocks=1024 upd_few_blocks=2048)
run il3_more_np --no-pipeline $(opt upd_interleave=3 upd_hot_blocks=512 upd_mid_blocks=1024 upd_few_blocks=2048)
run il2_more2_np --no-pipeline $(opt upd_interleave=2 upd_hot_blocks=1024 upd_mid_blocks=2048 upd_few_blocks=4096)
run il4_more_np --no-pipeline $(opt upd_interleave=4 upd_hot_blocks=512 upd_mid_blocks=1024 upd_few_blocks=2048)
run il2_few_np --no-pipeline $(opt upd_interleave=2 upd_few_blocks=2048)
run il2 $(opt upd_interleave=2)
run il3 $(opt upd_interleave=3)
run il2_more $(opt upd_interleave=2 upd_hot_blocks=512 upd_mid_blocks=1024 upd_few_blocks=2048)
run il3_more $(opt upd_interleave=3 upd_hot_blocks=512 upd_mid_blocks=1024 upd_few_blocks=2048)
run il2_more2 $(opt upd_interleave=2 upd_hot_blocks=1024 upd_mid_blocks=2048 upd_few_blocks=4096)
timeout 300 python -m pytest tests/test_kernel_parity.py -m gpu -q -x 2>&1 | tail -3

#!/bin/bash
# round 3, call E: block interleave and list-role block counts of k_update_fused, mapping fixed; parity; task=predict
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03e; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_kernel_parity.py tests/test_host_cpp.py -m gpu -q -x 2>&1 | tail -5
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
M1="upd_hot_blocks=512 upd_mid_blocks=1024 upd_few_blocks=2048"
M2="upd_hot_blocks=1024 upd_mid_blocks=2048 upd_few_blocks=4096"
M3="upd_hot_blocks=512 upd_mid_blocks=512 upd_few_blocks=1024"
for il in 0 2 3; do
  run il${il}_np --no-pipeline $(opt upd_interleave=$il)
  run il${il}_m1_np --no-pipeline $(opt upd_interleave=$il $M1)
  run il${il}_m2_np --no-pipeline $(opt upd_interleave=$il $M2)
  run il${il}_m3_np --no-pipeline $(opt upd_interleave=$il $M3)
done
run il0 $(opt upd_interleave=0)
run il0_m1 $(opt upd_interleave=0 $M1)
run il2_m1 $(opt upd_interleave=2 $M1)
run il0_m2 $(opt upd_interleave=0 $M2)
run il2_m2 $(opt upd_interleave=2 $M2)

#!/bin/bash
# bucket areas filled XCD group by XCD group (write-combining of the scatter's short runs): Localizer tests, A/B, WRITE_SIZE / FETCH_SIZE of the Localizer kernels
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03x; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -q -x -k "local or Local or fused or step or parity" ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^E  " $O/pytest_gpu.log | head -20
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
run base base; run xcd xcd; run base_np base --no-pipeline; run xcd_np xcd --no-pipeline; run base_b base; run xcd_b xcd
cp /tmp/keep.so $R/difacto_amd/libdifacto_hip.so
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o pmc -- python $R/bench.py --steps 20 --warmup 5 --cpu-batches 0 --no-pipeline --no-timing --min-time 0.001 --max-reps 1 --no-secondary > $O/pmc_$c.log 2>&1
done
f() { ls $O/$1/*.db $O/$1/*/*.db 2>/dev/null | head -1; }
python $R/tools/pmc_summary.py $(f pmc_FETCH_SIZE) $(f pmc_WRITE_SIZE) $O/pmc_hbm_traffic.json $O/pmc_hbm_traffic.txt > /dev/null 2>&1
cat $O/pmc_hbm_traffic.txt | head -14
find $O -name "*.db" -delete; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE

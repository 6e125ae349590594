#!/bin/bash
# round 4: cold start (empty table, every key of the first minibatches is NEW): row allocation with one atomic per
# wavefront (find_or_insert) against one per key (tools/var_base_alloc.so = the tree before), same box; then the
# warm default line of both, the sharded-store tests and the whole GPU suite on the final tree
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/${OUT:-r04t}; mkdir -p $O; cd $R
cp $R/difacto_amd/libdifacto_hip.so /tmp/keep.so
line() {  # name args...
  n=$1; shift
  timeout 300 python bench.py --cpu-batches 0 --no-secondary "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-26s' % '$n', round(d['value']/1e6,2), 'M ex/s', round(d['ms_per_step'],4), 'ms/step')
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-800:])" | tee -a $O/summary.txt
}
COLD="--no-prefill --warmup 0 --max-reps 1 --min-time 0 --no-timing"
for v in new base; do
  [ $v = base ] && cp $R/tools/var_${VARIANT:-base_alloc}.so $R/difacto_amd/libdifacto_hip.so
  line cold16_$v $COLD --steps 16
  line cold64_$v $COLD --steps 64
  line cold256_$v $COLD --steps 256
  line warm_$v --min-time 1
  cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$v -o kt -- python $R/bench.py --cpu-batches 0 --no-secondary $COLD --steps 64 > $O/prof_$v.log 2>&1; cd $R
  python $R/tools/rocpd_stats.py $(ls $O/prof_$v/*.db $O/prof_$v/*/*.db 2>/dev/null | head -1) $O/kernel_stats_cold64_$v.txt > /dev/null 2>&1
  rm -rf $O/prof_$v
done
cp /tmp/keep.so $R/difacto_amd/libdifacto_hip.so
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log | tee -a $O/summary.txt

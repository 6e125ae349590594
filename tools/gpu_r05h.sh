#!/bin/bash
# round 5: memory-side requests per kernel and per step (serial + pipelined default step), the request-rate ceiling of the
# same box (tools/fresh_bench.bin), and the step's time on that box
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05h; mkdir -p $O; cd /tmp
[ -x $R/tools/fresh_bench.bin ] && timeout 120 $R/tools/fresh_bench.bin > $O/fresh_bench.txt 2>&1; head -14 $O/fresh_bench.txt
B="--steps 40 --warmup 10 --cpu-batches 0 --no-timing --min-time 0.001 --max-reps 1 --no-secondary"
for mode in serial pipelined; do
  X=""; [ $mode = serial ] && X="--no-pipeline"
  for c in TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${mode}_$c -o pmc -- python $R/bench.py $B $X > $O/pmc_${mode}_$c.log 2>&1
  done
  f() { ls $O/$1/*.db $O/$1/*/*.db 2>/dev/null | head -1; }
  python $R/tools/requests_table.py $(f pmc_${mode}_TCC_EA0_RDREQ_sum) $(f pmc_${mode}_TCC_EA0_WRREQ_sum) $O/requests_$mode.json $O/requests_$mode.txt "C3 default step, $mode"
done
cd $R
for X in "" "--no-pipeline"; do timeout 300 python bench.py --cpu-batches 0 --min-time 1 --no-secondary $X 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$X', round(d['value']/1e6,2), 'M ex/s', round(d['ms_per_step'],4), 'ms/step', d.get('roofline_requests'))"; done
find $O -name "*.db" -delete; rm -rf $O/pmc_*_TCC*
# C2 (rcv1 shape, batch 100): is the small minibatch better off on ONE stream?  (six launches around ~7 500 pairs)
for X in "" "--no-pipeline" "--no-prep-lookup"; do timeout 200 python bench.py --preset c2 --cpu-batches 0 --min-time 1 --no-secondary $X 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 [$X]', round(d['value']/1e6,3), 'M ex/s', round(d['ms_per_step']*1e3,1), 'us/step', d.get('kernel_ms_per_step'))"; done

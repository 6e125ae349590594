#!/bin/bash
# round 4, third call: GPU suite (growing table, dynamic device feed, ballot-compacted AUC units, ev_p ordering), the AUC A/B again
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
( time DFH_PARITY_RECORD=$O/parity.json timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log | cut -c1-300; grep -E "^E  |^FAILED" $O/pytest_gpu.log | head -40
line() {  # name args...
  n=$1; shift
  timeout 300 python bench.py --cpu-batches 0 --min-time 1 --no-secondary "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-26s' % '$n', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(x,4) for k,x in (d.get('kernel_ms_per_step') or {}).items()}, 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round((d.get('roofline_backward') or {}).get('avg_launch_ms',0),4))
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-800:])"
}
line auc_rides
line auc_own_launch --ctx-option auc_in_update=0
line no_auc --no-auc
line auc_rides_again
line serial --no-pipeline
line serial_auc_own --no-pipeline --ctx-option auc_in_update=0
line serial_no_auc --no-pipeline --no-auc
line c5_auc --preset c5-slice
line c5_no_auc --preset c5-slice --no-auc
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3_np -o kt -- python $R/bench.py --steps 100 --warmup 20 --cpu-batches 0 --no-pipeline --min-time 0.05 --no-secondary --ctx-option auc_in_update=0 > $O/prof_c3_np.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_c3_np/*.db $O/prof_c3_np/*/*.db 2>/dev/null | head -1) $O/kernel_stats_c3_serial_auc_own_launch.txt > /dev/null 2>&1
head -9 $O/kernel_stats_c3_serial_auc_own_launch.txt | cut -c1-200
find $O -name "*.db" -delete; rm -rf $O/prof_c3_np
du -sh $O

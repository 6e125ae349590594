#!/bin/bash
# round 4, second call: the GPU suite (C_SIGMA = 1, reference-generated FM fixtures, non-finite values, AUC inside the update
# launch), then the bench line with the AUC in the step: riding in k_update_fused (default) / as its own launch / off;
# the 1-rank sharded line with cpu_baseline + roofline_exchange; forward grid caps
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04b; mkdir -p $O; cd $R
( time DFH_PARITY_RECORD=$O/parity.json timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log | cut -c1-300; grep -E "^E  |^FAILED" $O/pytest_gpu.log | head -40
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
line() {  # name args...
  n=$1; shift
  timeout 300 python bench.py --cpu-batches 0 --min-time 1 --no-secondary "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-26s' % '$n', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(x,4) for k,x in (d.get('kernel_ms_per_step') or {}).items()}, 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round((d.get('roofline_backward') or {}).get('avg_launch_ms',0),4), d.get('stage_ms_per_step'))
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-800:])"
}
line auc_rides
line auc_own_launch --ctx-option auc_in_update=0
line no_auc --no-auc
line auc_rides_again
line no_auc_d64 --no-auc --distinct 64
line fwd_blocks_1280 --ctx-option fwd_blocks=1280
line fwd_blocks_640 --ctx-option fwd_blocks=640
line serial --no-pipeline
line serial_auc_own --no-pipeline --ctx-option auc_in_update=0
( time timeout 600 python bench.py --force-sharded --min-time 1 ) > $O/b_sharded_w1.json 2> $O/b_sharded_w1.err; tail -3 $O/b_sharded_w1.err
python -c "
import json
d=json.loads(open('$O/b_sharded_w1.json').read().strip().splitlines()[-1])
print('sharded_w1', round(d['value']/1e6,2), d['stage_ms_per_step'], 'cpu', (d['cpu_baseline'] or {}).get('value'), 'rx', d['roofline_exchange'], d['config']['transport_bound'])"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o kt -- python $R/bench.py --cpu-batches 0 --min-time 0.5 --no-secondary > $O/prof_c3.log 2>&1
DB=$(ls $O/prof_c3/*.db $O/prof_c3/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_stats.py $DB $O/kernel_stats_c3_pipelined.txt > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $DB k_forward 5 $O/timeline_c3_pipelined.txt > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3_np -o kt -- python $R/bench.py --steps 100 --warmup 20 --cpu-batches 0 --no-pipeline --min-time 0.05 --no-secondary --ctx-option auc_in_update=0 > $O/prof_c3_np.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_c3_np/*.db $O/prof_c3_np/*/*.db 2>/dev/null | head -1) $O/kernel_stats_c3_serial_auc_own_launch.txt > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmcs_$c -o pmc -- python $R/bench.py --force-sharded --exchange sync --steps 20 --warmup 5 --cpu-batches 0 --no-timing --min-time 0.001 --max-reps 1 > $O/pmcs_$c.log 2>&1
done
f() { ls $O/$1/*.db $O/$1/*/*.db 2>/dev/null | head -1; }
python $R/tools/pmc_summary.py $(f pmcs_FETCH_SIZE) $(f pmcs_WRITE_SIZE) $O/pmc_hbm_traffic_sharded_w1.json $O/pmc_hbm_traffic_sharded_w1.txt > /dev/null 2>&1
head -12 $O/kernel_stats_c3_pipelined.txt | cut -c1-200; head -12 $O/kernel_stats_c3_serial_auc_own_launch.txt | cut -c1-200; cat $O/timeline_c3_pipelined.txt; head -14 $O/pmc_hbm_traffic_sharded_w1.txt
find $O -name "*.db" -delete; rm -rf $O/pmcs_FETCH_SIZE $O/pmcs_WRITE_SIZE $O/prof_c3 $O/prof_c3_np
du -sh $O

#!/bin/bash
# round 5, final measured set on the tree's own library: parity suite (+ record), smoke, the default bench line (with `secondary`),
# later-epoch / serial / 1-rank sharded lines, kernel stats pipelined + serial, timeline, overlap, HBM traffic counters (c3, c5 slice)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05z; mkdir -p $O; cd $R
( time DFH_PARITY_RECORD=$O/parity.json DFH_PARITY_RECORD_STEPS=$O/parity_steps.json timeout 1800 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log | cut -c1-300; grep -E "^E |^FAILED" $O/pytest_gpu.log | head -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench_c3.json 2> $O/bench_c3.err; tail -3 $O/bench_c3.err
python -c "
import json
d=json.loads(open('$O/bench_c3.json').read().strip().splitlines()[-1])
print('default', round(d['value']/1e6,2), round(d['ms_per_step'],4), 'reps', d['repetitions'], 'fwd frac', round(d['roofline']['frac'],3), 'bwd frac', round(d['roofline_backward']['frac'],3), round(d['roofline_backward']['frac_hbm_necessary'],3), 'step frac', round(d['roofline_step']['frac'],3))
for k,v in (d.get('secondary') or {}).items(): print(' secondary', k, {a:(round(b/1e6,2) if a=='value' else b) for a,b in v.items() if a in ('value','ms_per_step','wall_seconds','error')})
print(' cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('scaled_threads',{}).get('value'))
"
line() {  # name args...
  n=$1; shift
  timeout 300 python bench.py --cpu-batches 0 --min-time 1 --no-secondary "$@" > $O/bench_$n.json 2> $O/bench_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1])
  print('%-22s' % '$n', round(d['value']/1e6,2), round(d['ms_per_step'],4), d.get('kernel_ms_per_step'), d.get('stage_ms_per_step'))
except Exception as e: print('$n ERR', e); print(open('$O/bench_$n.err').read()[-600:])"
}
line c3_later_epoch --later-epoch
line c3_serial --no-pipeline
line c3_no_relocalize --no-relocalize
line c2 --preset c2
line c3_no_auc --no-auc
line c3_cold --no-prefill --steps 256 --warmup 0 --max-reps 1 --min-time 0
line sharded_w1_native_overlap --force-sharded --exchange overlap
line sharded_w1_native_sync --force-sharded --exchange sync
( time timeout 600 python bench.py --force-sharded --min-time 1 ) > $O/bench_sharded_w1_full.json 2> $O/bench_sharded_w1_full.err; tail -2 $O/bench_sharded_w1_full.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o kt -- python $R/bench.py --cpu-batches 0 --min-time 0.5 --no-secondary > $O/prof_c3.log 2>&1
DB=$(ls $O/prof_c3/*.db $O/prof_c3/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_stats.py $DB $O/kernel_stats_c3_pipelined.txt > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $DB k_forward 5 $O/timeline_c3_pipelined.txt > /dev/null 2>&1
python $R/tools/rocpd_overlap.py $DB $O/overlap_c3_pipelined.txt > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3_np -o kt -- python $R/bench.py --steps 100 --warmup 20 --cpu-batches 0 --no-pipeline --min-time 0.05 --no-secondary > $O/prof_c3_np.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_c3_np/*.db $O/prof_c3_np/*/*.db 2>/dev/null | head -1) $O/kernel_stats_c3_serial.txt > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o pmc -- python $R/bench.py --steps 20 --warmup 5 --cpu-batches 0 --no-pipeline --no-timing --min-time 0.001 --max-reps 1 --no-secondary > $O/pmc_$c.log 2>&1
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/pmc5_$c -o pmc -- python $R/bench.py --preset c5-slice --steps 20 --warmup 5 --cpu-batches 0 --no-pipeline --no-timing --min-time 0.001 --max-reps 1 --no-secondary > $O/pmc5_$c.log 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmcs_$c -o pmc -- python $R/bench.py --force-sharded --exchange sync --steps 20 --warmup 5 --cpu-batches 0 --no-timing --min-time 0.001 --max-reps 1 > $O/pmcs_$c.log 2>&1
done
f() { ls $O/$1/*.db $O/$1/*/*.db 2>/dev/null | head -1; }
python $R/tools/pmc_summary.py $(f pmc_FETCH_SIZE) $(f pmc_WRITE_SIZE) $O/pmc_hbm_traffic.json $O/pmc_hbm_traffic.txt > /dev/null 2>&1
python $R/tools/pmc_summary.py $(f pmc5_FETCH_SIZE) $(f pmc5_WRITE_SIZE) $O/pmc_hbm_traffic_c5_slice.json $O/pmc_hbm_traffic_c5_slice.txt > /dev/null 2>&1
python $R/tools/pmc_summary.py $(f pmcs_FETCH_SIZE) $(f pmcs_WRITE_SIZE) $O/pmc_hbm_traffic_sharded_w1.json $O/pmc_hbm_traffic_sharded_w1.txt > /dev/null 2>&1
head -12 $O/kernel_stats_c3_pipelined.txt; cat $O/timeline_c3_pipelined.txt; cat $O/pmc_hbm_traffic.txt | head -20; cat $O/pmc_hbm_traffic_c5_slice.txt | head -12
DIFACTO_PROFILE=1 E2E_FORMATS=criteo,rec timeout 900 python $R/tools/e2e_cli.py 400000 48 > $O/e2e.jsonl 2> $O/e2e.err
DIFACTO_PROFILE=1 E2E_FORMATS=libsvm E2E_BATCH_SIZE=100 E2E_VDIM=8 timeout 600 python $R/tools/e2e_cli.py 100000 4 > $O/e2e_c2shape.jsonl 2> $O/e2e_c2shape.err; grep 'host loop over' $O/e2e_c2shape.err | tail -2
find $O -name "*.db" -delete; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc5_FETCH_SIZE $O/pmc5_WRITE_SIZE $O/pmcs_FETCH_SIZE $O/pmcs_WRITE_SIZE $O/prof_c3 $O/prof_c3_np
du -sh $O

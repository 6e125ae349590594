#!/bin/bash
# round 6, final measured set on the tree's own library: parity suite (+ record), smoke, the default bench line (with `secondary`),
# later-epoch / serial / no-Localizer / single-queue lines, 1-rank sharded line, the N = 8 projection, kernel stats pipelined + serial,
# HBM traffic counters (c3, c5 slice, sharded), memory-side requests, build/difacto end to end (+ process clock), the gloo dry run at 8 ranks
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06z; mkdir -p $O; cd $R
python -c "from difacto_amd.build import build_hip, build_host; build_hip(); build_host()" 2>&1 | tail -2
( time DFH_PARITY_RECORD=$O/parity.json DFH_PARITY_RECORD_STEPS=$O/parity_steps.json timeout 1800 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2 | cut -c1-300; grep -E "^E |^FAILED" $O/pytest_gpu.log | head -30
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
  timeout 400 python bench.py --cpu-batches 0 --min-time 1 --no-secondary "$@" > $O/bench_$n.json 2> $O/bench_$n.err
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
line c3_no_auc --no-auc
line c3_cold --no-prefill --steps 256 --warmup 0 --max-reps 1 --min-time 0
line c3_single_queue --single-queue
line c3_single_queue_best --single-queue --ahead 4 --ctx-option rider_slot_count=2 --ctx-option rider_slot_scatter=2 --ctx-option rider_slot_sort=2 --ctx-option rider_slot_emit=2 --ctx-option rider_start_update=50
line c2 --preset c2
line c2_single_queue --preset c2 --single-queue
line sharded_w1_native_overlap --force-sharded --exchange overlap
line sharded_w1_native_sync --force-sharded --exchange sync
( time timeout 600 python bench.py --force-sharded --min-time 1 ) > $O/bench_sharded_w1_full.json 2> $O/bench_sharded_w1_full.err; tail -2 $O/bench_sharded_w1_full.err
timeout 900 python bench.py --emulate-world 8 --emulate-rank auto --cpu-batches 0 --min-time 1 > $O/emul_c4_w8.json 2> $O/emul_c4_w8.err
python -c "
import json
try:
  d=json.loads(open('$O/emul_c4_w8.json').read().strip().splitlines()[-1])
  print('emulated N=8', {k:d.get(k) for k in ('value','ms_per_step','projection')}, json.dumps(d.get('wire_models') or d.get('projected') or {})[:400])
except Exception as e: print('emul ERR', e); print(open('$O/emul_c4_w8.err').read()[-600:])"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o kt -- python $R/bench.py --cpu-batches 0 --min-time 0.5 --no-secondary > $O/prof_c3.log 2>&1
DB=$(ls $O/prof_c3/*.db $O/prof_c3/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_stats.py $DB $O/kernel_stats_c3_pipelined.txt > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $DB k_forward 5 $O/timeline_c3_pipelined.txt > /dev/null 2>&1
python $R/tools/rocpd_overlap.py $DB $O/overlap_c3_pipelined.txt > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3_np -o kt -- python $R/bench.py --steps 100 --warmup 20 --cpu-batches 0 --no-pipeline --min-time 0.05 --no-secondary > $O/prof_c3_np.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_c3_np/*.db $O/prof_c3_np/*/*.db 2>/dev/null | head -1) $O/kernel_stats_c3_serial.txt > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3_cold -o kt -- python $R/bench.py --cpu-batches 0 --no-secondary --no-prefill --steps 256 --warmup 0 --max-reps 1 --min-time 0 > $O/prof_c3_cold.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_c3_cold/*.db $O/prof_c3_cold/*/*.db 2>/dev/null | head -1) $O/kernel_stats_c3_cold_256.txt > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o pmc -- python $R/bench.py --steps 20 --warmup 5 --cpu-batches 0 --no-pipeline --no-timing --min-time 0.001 --max-reps 1 --no-secondary > $O/pmc_$c.log 2>&1
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/pmc5_$c -o pmc -- python $R/bench.py --preset c5-slice --steps 20 --warmup 5 --cpu-batches 0 --no-pipeline --no-timing --min-time 0.001 --max-reps 1 --no-secondary > $O/pmc5_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmcs_$c -o pmc -- python $R/bench.py --force-sharded --exchange sync --steps 20 --warmup 5 --cpu-batches 0 --no-timing --min-time 0.001 --max-reps 1 > $O/pmcs_$c.log 2>&1
done
for c in TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/req_$c -o pmc -- python $R/bench.py --steps 20 --warmup 5 --cpu-batches 0 --no-timing --min-time 0.001 --max-reps 1 --no-secondary > $O/req_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/reqs_$c -o pmc -- python $R/bench.py --steps 20 --warmup 5 --cpu-batches 0 --no-pipeline --no-timing --min-time 0.001 --max-reps 1 --no-secondary > $O/reqs_$c.log 2>&1
done
f() { ls $O/$1/*.db $O/$1/*/*.db 2>/dev/null | head -1; }
python $R/tools/pmc_summary.py $(f pmc_FETCH_SIZE) $(f pmc_WRITE_SIZE) $O/pmc_hbm_traffic.json $O/pmc_hbm_traffic.txt > /dev/null 2>&1
python $R/tools/pmc_summary.py $(f pmc5_FETCH_SIZE) $(f pmc5_WRITE_SIZE) $O/pmc_hbm_traffic_c5_slice.json $O/pmc_hbm_traffic_c5_slice.txt > /dev/null 2>&1
python $R/tools/pmc_summary.py $(f pmcs_FETCH_SIZE) $(f pmcs_WRITE_SIZE) $O/pmc_hbm_traffic_sharded_w1.json $O/pmc_hbm_traffic_sharded_w1.txt > /dev/null 2>&1
python $R/tools/requests_table.py $(f req_TCC_EA0_RDREQ_sum) $(f req_TCC_EA0_WRREQ_sum) $O/requests_pipelined.json $O/requests_pipelined.txt "default step (pipelined)" > /dev/null 2>&1
python $R/tools/requests_table.py $(f reqs_TCC_EA0_RDREQ_sum) $(f reqs_TCC_EA0_WRREQ_sum) $O/requests_serial.json $O/requests_serial.txt "serial step" > /dev/null 2>&1
head -12 $O/kernel_stats_c3_pipelined.txt; cat $O/timeline_c3_pipelined.txt | head -30; cat $O/pmc_hbm_traffic.txt | head -20; cat $O/pmc_hbm_traffic_c5_slice.txt | head -12; head -14 $O/requests_pipelined.txt
cd $R
[ -x tools/fresh_bench.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/fresh_bench.bin tools/fresh_bench.hip 2>/dev/null
timeout 200 tools/fresh_bench.bin > $O/fresh_rows_bench.txt 2>&1; head -12 $O/fresh_rows_bench.txt
DIFACTO_PROFILE=1 E2E_FORMATS=criteo,rec timeout 900 python $R/tools/e2e_cli.py 400000 48 > $O/e2e.jsonl 2> $O/e2e.err
grep "process:" $O/e2e.err | tail -4
python - $O/e2e.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print(d["format"], "whole loop M rows/s %.1f steady %.1f loop_s small/big %.4f %.4f process wall small/big %.3f %.3f -> process M rows/s %.1f" % (
        d.get("loop_rows_per_s_big",0)/1e6, d.get("steady_rows_per_s_by_loop_clock",0)/1e6, d.get("loop_s",0), d.get("loop_s_big",0), d["wall_s"], d["wall_s_big"], d["rows_big"]/d["wall_s_big"]/1e6))
PY
DIFACTO_PROFILE=1 E2E_FORMATS=libsvm E2E_BATCH_SIZE=100 E2E_VDIM=8 timeout 600 python $R/tools/e2e_cli.py 100000 4 > $O/e2e_c2shape.jsonl 2> $O/e2e_c2shape.err; grep 'host loop over' $O/e2e_c2shape.err | tail -2
DFH_BENCH_BACKEND=gloo DFH_WIRE_PROBE_BYTES=100000,1000000 timeout 900 python bench.py --gpus 8 --steps 5 --warmup 2 --min-time 0 --ids 2000000 --rows 2000 --distinct 8 --cpu-batches 0 > $O/dryrun_shared_gpu_w8.json 2> $O/dryrun_shared_gpu_w8.err; tail -c 300 $O/dryrun_shared_gpu_w8.json; echo
find $O -name "*.db" -delete; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc5_FETCH_SIZE $O/pmc5_WRITE_SIZE $O/pmcs_FETCH_SIZE $O/pmcs_WRITE_SIZE $O/req_TCC* $O/reqs_TCC* $O/prof_c3 $O/prof_c3_np $O/prof_c3_cold
du -sh $O

#!/bin/bash
# The round's profile set on the final tree: parity suite, smoke, host tests, bench presets, kernel stats (pipelined, serial, sharded, C5 slice), HBM traffic counters, end-to-end CLI rate.  Output: gpurun_out/r02e (summaries only)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r02e; mkdir -p $O; cd $R
( time timeout 600 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log | cut -c1-300; grep -E "^E |^FAILED" $O/pytest_gpu.log | head -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 ./build/difacto_host_tests tests/golden/rcv1_100.libsvm > $O/host_tests.log 2>&1; tail -1 $O/host_tests.log
timeout 300 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
timeout 200 python bench.py --no-pipeline --cpu-batches 0 > $O/bench_c3_serial.json 2> $O/bench_c3_serial.err
timeout 200 python bench.py --preset c3-refdefaults --cpu-batches 0 > $O/bench_c3_refdefaults.json 2> $O/bench_c3_refdefaults.err
timeout 200 python bench.py --preset c2 > $O/bench_c2.json 2> $O/bench_c2.err
timeout 200 python bench.py --force-sharded --steps 200 --warmup 20 > $O/bench_sharded_w1.json 2> $O/bench_sharded_w1.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o kt -- python $R/bench.py --cpu-batches 0 > $O/prof_c3.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_c3/*.db $O/prof_c3/*/*.db 2>/dev/null | head -1) $O/kernel_stats_c3_pipelined.txt > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $(ls $O/prof_c3/*.db $O/prof_c3/*/*.db 2>/dev/null | head -1) k_forward 5 $O/timeline_c3_pipelined.txt > /dev/null 2>&1
python $R/tools/rocpd_overlap.py $(ls $O/prof_c3/*.db $O/prof_c3/*/*.db 2>/dev/null | head -1) $O/overlap_c3_pipelined.txt > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3_np -o kt -- python $R/bench.py --steps 100 --warmup 20 --cpu-batches 0 --no-pipeline --min-time 0.05 > $O/prof_c3_np.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_c3_np/*.db $O/prof_c3_np/*/*.db 2>/dev/null | head -1) $O/kernel_stats_c3_serial.txt > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o pmc -- python $R/bench.py --steps 20 --warmup 5 --cpu-batches 0 --no-pipeline --no-timing --min-time 0.001 --max-reps 1 > $O/pmc_$c.log 2>&1
done
python $R/tools/pmc_summary.py $(ls $O/pmc_FETCH_SIZE/*.db $O/pmc_FETCH_SIZE/*/*.db 2>/dev/null | head -1) $(ls $O/pmc_WRITE_SIZE/*.db $O/pmc_WRITE_SIZE/*/*.db 2>/dev/null | head -1) $O/pmc_hbm_traffic.json $O/pmc_hbm_traffic.txt > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_w1 -o kt -- python $R/bench.py --force-sharded --steps 100 --warmup 10 > $O/prof_w1.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_w1/*.db $O/prof_w1/*/*.db 2>/dev/null | head -1) $O/kernel_stats_sharded_w1.txt > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o kt -- python $R/bench.py --preset c5-slice --steps 100 --warmup 20 > $O/bench_c5_slice.json 2> $O/bench_c5_slice.err
python $R/tools/rocpd_stats.py $(ls $O/prof_c5/*.db $O/prof_c5/*/*.db 2>/dev/null | head -1) $O/kernel_stats_c5_slice_pipelined.txt > /dev/null 2>&1
cd $R
timeout 600 python tools/e2e_cli.py 200000 16 > $O/e2e.jsonl 2> $O/e2e.err; cat $O/e2e.jsonl; tail -2 $O/e2e.err
for f in bench_c3 bench_c3_serial bench_c3_refdefaults bench_c2 bench_sharded_w1 bench_c5_slice; do
  python -c "
import json
try:
  d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
  print('$f', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(x,4) for k,x in (d.get('kernel_ms_per_step') or {}).items()}, 'live fwd/bwd', (d.get('roofline') or {}).get('avg_launch_ms'), (d.get('roofline_backward') or {}).get('avg_launch_ms'))
except Exception as e: print('$f ERR', e); print(open('$O/$f.err').read()[-800:])"
done
find $O -name "*.db" -delete; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/prof_c3 $O/prof_c3_np $O/prof_w1 $O/prof_c5
du -sh $O

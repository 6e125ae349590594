#!/bin/bash
# Round 2, GPU call A: parity suite, smoke, the bench presets, exact-vs-fast AdaGrad A/B, kernel stats.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r02a; mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q -s ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 300 ./build/difacto_host_tests > $O/host_tests.log 2>&1; tail -3 $O/host_tests.log
timeout 300 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 600 $O/bench_c3.json
timeout 300 python bench.py --preset c3-refdefaults --cpu-batches 0 > $O/bench_c3_refdefaults.json 2> $O/bench_c3_refdefaults.err
timeout 300 python bench.py --preset c2 > $O/bench_c2.json 2> $O/bench_c2.err
timeout 300 python bench.py --no-pipeline --cpu-batches 0 > $O/bench_c3_serial.json 2> $O/bench_c3_serial.err
# A/B: hardware sqrt / rcp in the AdaGrad update
cp difacto_amd/libdifacto_hip.so /tmp/exact.so
cp tools/libdfh_fast_adagrad.so difacto_amd/libdifacto_hip.so
timeout 300 python bench.py --cpu-batches 0 > $O/bench_c3_fast_adagrad.json 2> $O/bench_c3_fast_adagrad.err
timeout 300 python bench.py --no-pipeline --cpu-batches 0 > $O/bench_c3_serial_fast_adagrad.json 2>> $O/bench_c3_fast_adagrad.err
cp /tmp/exact.so difacto_amd/libdifacto_hip.so
# kernel stats: serial C3 and the C5 slice (L = 32 instantiations)
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3_np -o kt -- python $R/bench.py --steps 100 --warmup 20 --cpu-batches 0 --no-pipeline > $O/prof_c3_np.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_c3_np/*/*.db | head -1) $O/kernel_stats_c3_serial.txt > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o kt -- python $R/bench.py --preset c5-slice --steps 100 --warmup 20 > $O/bench_c5_slice.json 2> $O/bench_c5_slice.err
python $R/tools/rocpd_stats.py $(ls $O/prof_c5/*/*.db | head -1) $O/kernel_stats_c5_slice.txt > /dev/null 2>&1
cd $R
for f in bench_c3_refdefaults bench_c2 bench_c3_serial bench_c3_fast_adagrad bench_c3_serial_fast_adagrad bench_c5_slice; do
  echo "== $f"; python -c "
import json,sys
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d.get('repetitions'), (d.get('roofline') or {}).get('avg_launch_ms'), (d.get('roofline_backward') or {}).get('avg_launch_ms'), d['config'].get('model_keys'))
except Exception as e: print('ERR', e)
"; done
rm -rf $O/prof_c3_np/*/*.db.tmp 2>/dev/null
du -sh $O

#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r02c; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -15 $O/pytest_gpu.log | cut -c1-400
for v in "" "--prep-lookup" "--prep-streams 1" "--prep-streams 1 --prep-lookup" "--prep-streams 3 --prep-lookup" "--no-pipeline"; do
  n=$(echo "x$v" | tr -d ' -'); timeout 200 python bench.py --cpu-batches 0 $v > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('[$v]', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['kernel_ms_per_step'].items()}, round(d['roofline']['avg_launch_ms'],4), round(d['roofline_backward']['avg_launch_ms'],4))
except Exception as e: print('[$v] ERR', e); print(open('$O/b_$n.err').read()[-800:])"
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_np -o kt -- python $R/bench.py --steps 100 --warmup 20 --cpu-batches 0 --no-pipeline > $O/prof_np.log 2>&1
python $R/tools/rocpd_stats.py $O/prof_np/kt_results.db $O/kernel_stats_serial.txt | cut -c1-60,90-200 | head -16

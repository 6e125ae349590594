#!/bin/bash
# Round 2, GPU call T: two-launch Localizer + pair-counting AUC: parity suite, serial / pipelined bench A/B against
# the four-launch form, kernel stats of the serial step
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r02t; mkdir -p $O; cd $R
( time timeout 600 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log | cut -c1-300; grep -E "^E " $O/pytest_gpu.log | head -20
for mode in "--no-pipeline" "--no-pipeline --loc-launches 4" "" "--loc-launches 4" "--prep-streams 2"; do
  n=$(echo "b$mode" | tr -d ' -')
  timeout 200 python bench.py --cpu-batches 0 $mode > $O/$n.json 2> $O/$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
  print('[%s]' % '$mode', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['kernel_ms_per_step'].items()}, 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round(d['roofline_backward']['avg_launch_ms'],4))
except Exception as e: print('[$mode] ERR', e); print(open('$O/$n.err').read()[-600:])"
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3_np -o kt -- python $R/bench.py --steps 100 --warmup 20 --cpu-batches 0 --no-pipeline --min-time 0.05 > $O/prof_c3_np.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_c3_np/*.db $O/prof_c3_np/*/*.db 2>/dev/null | head -1) $O/kernel_stats_c3_serial.txt > /dev/null 2>&1
cut -c1-60,90-150 $O/kernel_stats_c3_serial.txt | head -12
find $O -name "*.db" -size +20M -delete

#!/bin/bash
# round 5: C2 (batch 100, V_dim 8) on the device's own clock — per-kernel durations and the timeline of a few steps, to put a
# number on what a one-launch small-minibatch step could gain
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05x; mkdir -p $O; cd /tmp
for mode in pipelined serial; do
  fl=""; [ $mode = serial ] && fl="--no-pipeline"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c2_$mode -o kt -- python $R/bench.py --preset c2 --cpu-batches 0 --min-time 0.3 --no-secondary --no-timing $fl > $O/prof_c2_$mode.log 2>&1
  DB=$(ls $O/prof_c2_$mode/*.db $O/prof_c2_$mode/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_stats.py $DB $O/kernel_stats_c2_$mode.txt > /dev/null 2>&1
  python $R/tools/rocpd_timeline.py $DB k_forward 4 $O/timeline_c2_$mode.txt > /dev/null 2>&1
  tail -1 $O/prof_c2_$mode.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$mode (under the profiler)', round(d['value']/1e6, 3), 'M ex/s', round(d['ms_per_step']*1e3, 1), 'us/step, host enqueue', round(d.get('host_enqueue_ms_per_step', 0)*1e3, 1))"
  rm -rf $O/prof_c2_$mode
done
head -30 $O/kernel_stats_c2_pipelined.txt

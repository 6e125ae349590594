#!/bin/bash
# kernel stats of 64 cold steps (bench.py --no-prefill) on the final tree
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04y; mkdir -p $O; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_cold -o kt -- python $R/bench.py --cpu-batches 0 --no-secondary --no-prefill --warmup 0 --max-reps 1 --min-time 0 --no-timing --steps 64 > $O/prof_cold.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_cold/*.db $O/prof_cold/*/*.db 2>/dev/null | head -1) $O/kernel_stats_cold64_final.txt > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $(ls $O/prof_cold/*.db $O/prof_cold/*/*.db 2>/dev/null | head -1) k_forward 20 $O/timeline_cold.txt > /dev/null 2>&1
rm -rf $O/prof_cold; head -12 $O/kernel_stats_cold64_final.txt | cut -c1-150; cat $O/timeline_cold.txt | cut -c1-150

#!/bin/bash
# kernel stats + timeline of the N > 1 code path on one rank over RCCL (both exchange modes)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03am; mkdir -p $O; cd /tmp
for ex in overlap sync; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$ex -o kt -- python $R/bench.py --force-sharded --exchange $ex --steps 200 --warmup 20 --min-time 0.3 --no-secondary > $O/prof_$ex.log 2>&1
  DB=$(ls $O/prof_$ex/*.db $O/prof_$ex/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_stats.py $DB $O/kernel_stats_sharded_w1_$ex.txt > /dev/null 2>&1
  python $R/tools/rocpd_timeline.py $DB k_forward 5 $O/timeline_sharded_w1_$ex.txt > /dev/null 2>&1
  head -16 $O/kernel_stats_sharded_w1_$ex.txt | cut -c1-170; cat $O/timeline_sharded_w1_$ex.txt | cut -c1-150
done
find $O -name "*.db" -delete; rm -rf $O/prof_overlap $O/prof_sync

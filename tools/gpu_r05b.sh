#!/bin/bash
# round 5: one GPU as rank 0 of 8 (loop-back transport): per-kernel stats (sync exchange, no wire model: every kernel alone),
# overlap-mode stats, and the HBM traffic counters of the owner-side kernels
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05b; mkdir -p $O; cd /tmp
E="--emulate-world 8 --emulate-rank 0 --steps 50 --warmup 10 --min-time 0.3 --no-timing"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_sync -o kt -- python $R/bench.py $E --exchange sync > $O/prof_sync.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_sync/*.db $O/prof_sync/*/*.db 2>/dev/null | head -1) $O/kernel_stats_emulated_w8_sync.txt > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_ovl -o kt -- python $R/bench.py $E --exchange overlap > $O/prof_ovl.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_ovl/*.db $O/prof_ovl/*/*.db 2>/dev/null | head -1) $O/kernel_stats_emulated_w8_overlap.txt > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $(ls $O/prof_ovl/*.db $O/prof_ovl/*/*.db 2>/dev/null | head -1) k_forward 5 $O/timeline_emulated_w8_overlap.txt > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o pmc -- python $R/bench.py --emulate-world 8 --emulate-rank 0 --steps 10 --warmup 5 --min-time 0.001 --max-reps 1 --no-timing --exchange sync --emulate-wire off > $O/pmc_$c.log 2>&1
done
f() { ls $O/$1/*.db $O/$1/*/*.db 2>/dev/null | head -1; }
python $R/tools/pmc_summary.py $(f pmc_FETCH_SIZE) $(f pmc_WRITE_SIZE) $O/pmc_hbm_traffic_emulated_w8.json $O/pmc_hbm_traffic_emulated_w8.txt > /dev/null 2>&1
head -30 $O/kernel_stats_emulated_w8_sync.txt | cut -c1-200; head -24 $O/pmc_hbm_traffic_emulated_w8.txt
cat $O/timeline_emulated_w8_overlap.txt | head -60
find $O -name "*.db" -delete; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/prof_sync $O/prof_ovl

#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06z; mkdir -p $O; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o pmc -- python $R/bench.py --steps 20 --warmup 5 --cpu-batches 0 --no-pipeline --no-timing --min-time 0.001 --max-reps 1 --no-secondary > $O/pmc_$c.log 2>&1
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/pmc5_$c -o pmc -- python $R/bench.py --preset c5-slice --steps 20 --warmup 5 --cpu-batches 0 --no-pipeline --no-timing --min-time 0.001 --max-reps 1 --no-secondary > $O/pmc5_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmcs_$c -o pmc -- python $R/bench.py --force-sharded --exchange sync --steps 20 --warmup 5 --cpu-batches 0 --no-timing --min-time 0.001 --max-reps 1 > $O/pmcs_$c.log 2>&1
done
f() { ls $O/$1/*.db $O/$1/*/*.db 2>/dev/null | head -1; }
python $R/tools/pmc_summary.py $(f pmc_FETCH_SIZE) $(f pmc_WRITE_SIZE) $O/hbm_traffic.json $O/hbm_traffic.txt > /dev/null 2>&1
python $R/tools/pmc_summary.py $(f pmc5_FETCH_SIZE) $(f pmc5_WRITE_SIZE) $O/hbm_traffic_c5_slice.json $O/hbm_traffic_c5_slice.txt > /dev/null 2>&1
python $R/tools/pmc_summary.py $(f pmcs_FETCH_SIZE) $(f pmcs_WRITE_SIZE) $O/hbm_traffic_sharded_w1.json $O/hbm_traffic_sharded_w1.txt > /dev/null 2>&1
cd $R
( time timeout 600 python bench.py --force-sharded --min-time 1 ) > $O/bench_sharded_w1_full.json 2> $O/bench_sharded_w1_full.err; tail -2 $O/bench_sharded_w1_full.err
find $O -name "*.db" -delete; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc5_FETCH_SIZE $O/pmc5_WRITE_SIZE $O/pmcs_FETCH_SIZE $O/pmcs_WRITE_SIZE
ls $O | grep hbm; head -8 $O/hbm_traffic.txt

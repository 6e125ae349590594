#!/bin/bash
# memory-side counters for small random accesses: tools/pmc_calib.bin (known bytes) and the serial C3 step, with FETCH_SIZE /
# WRITE_SIZE next to gfx950's 32 B-granular DRAM request counters
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03z; mkdir -p $O; cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_DRAM_32B_sum" "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace -d $O/cal_$i -o p -- $R/tools/pmc_calib.bin 1000000 > $O/cal_$i.log 2>&1
done
python $R/tools/pmc_table.py $O/pmc_calibration_small_accesses.txt $(ls $O/cal_*/*.db $O/cal_*/*/*.db 2>/dev/null)
i=0
for set in "TCC_EA0_RDREQ_DRAM_32B_sum" "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $O/c3_$i -o p -- python $R/bench.py --steps 20 --warmup 5 --cpu-batches 0 --no-pipeline --no-timing --min-time 0.001 --max-reps 1 --no-secondary > $O/c3_$i.log 2>&1
done
python $R/tools/pmc_table.py $O/pmc_dram32_c3_serial.txt $(ls $O/c3_*/*.db $O/c3_*/*/*.db 2>/dev/null)
tail -3 $O/cal_3.log $O/cal_5.log
find $O -name "*.db" -delete; rm -rf $O/cal_? $O/c3_?

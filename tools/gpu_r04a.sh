#!/bin/bash
# round 4, first call: the GPU suite on the tree with k_forward_singles, then same-box A/B of the singles role in the
# forward's epilogue (fwd_singles = 1, 5 and 4 waves per SIMD) against k_forward + four-role k_update_fused (fwd_singles = 0):
# pipelined, serial and no-relocalize lines; kernel stats + HBM counters of the new default
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04a; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log | cut -c1-300; grep -E "^E |^FAILED" $O/pytest_gpu.log | head -30
cp $R/difacto_amd/libdifacto_hip.so /tmp/keep.so
line() {  # name args...
  n=$1; shift
  timeout 300 python bench.py --cpu-batches 0 --min-time 1 --no-secondary "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-26s' % '$n', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['kernel_ms_per_step'].items()}, 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round(d['roofline_backward']['avg_launch_ms'],4))
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-600:])"
}
for v in fs5 fs4; do
  cp $R/tools/var_$v.so $R/difacto_amd/libdifacto_hip.so
  line ${v}_on
  line ${v}_on_serial --no-pipeline
  line ${v}_on_norelocalize --no-relocalize
done
line off --ctx-option fwd_singles=0
line off_serial --no-pipeline --ctx-option fwd_singles=0
line off_norelocalize --no-relocalize --ctx-option fwd_singles=0
cp $R/tools/var_fs5.so $R/difacto_amd/libdifacto_hip.so
line fs5_on_again
line fs5_few2048 --ctx-option upd_few_blocks=2048
line fs5_hot1024_mid1024 --ctx-option upd_hot_blocks=1024 --ctx-option upd_mid_blocks=1024
line fs5_c5 --preset c5-slice
line off_c5 --preset c5-slice --ctx-option fwd_singles=0
cp /tmp/keep.so $R/difacto_amd/libdifacto_hip.so
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o kt -- python $R/bench.py --cpu-batches 0 --min-time 0.5 --no-secondary > $O/prof_c3.log 2>&1
DB=$(ls $O/prof_c3/*.db $O/prof_c3/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_stats.py $DB $O/kernel_stats_c3_pipelined.txt > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $DB k_forward 5 $O/timeline_c3_pipelined.txt > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3_np -o kt -- python $R/bench.py --steps 100 --warmup 20 --cpu-batches 0 --no-pipeline --min-time 0.05 --no-secondary > $O/prof_c3_np.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_c3_np/*.db $O/prof_c3_np/*/*.db 2>/dev/null | head -1) $O/kernel_stats_c3_serial.txt > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o pmc -- python $R/bench.py --steps 20 --warmup 5 --cpu-batches 0 --no-pipeline --no-timing --min-time 0.001 --max-reps 1 --no-secondary > $O/pmc_$c.log 2>&1
done
f() { ls $O/$1/*.db $O/$1/*/*.db 2>/dev/null | head -1; }
python $R/tools/pmc_summary.py $(f pmc_FETCH_SIZE) $(f pmc_WRITE_SIZE) $O/pmc_hbm_traffic.json $O/pmc_hbm_traffic.txt > /dev/null 2>&1
head -12 $O/kernel_stats_c3_pipelined.txt | cut -c1-200; head -12 $O/kernel_stats_c3_serial.txt | cut -c1-200; cat $O/timeline_c3_pipelined.txt; head -14 $O/pmc_hbm_traffic.txt
find $O -name "*.db" -delete; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/prof_c3 $O/prof_c3_np
du -sh $O

#!/bin/bash
# round 3, call C: header write granularity (fresh rows), per-role time of k_update_fused, timeline of the pipelined step
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03c; mkdir -p $O; cd $R
./tools/fresh_bench.bin 32 0 2>&1 | grep "blocks  4096\|nsets" | tee $O/fresh_bench.txt
cp $R/difacto_amd/libdifacto_hip.so /tmp/keep.so
run() {  # name variant args...
  n=$1; v=$2; shift 2
  cp $R/tools/var_$v.so $R/difacto_amd/libdifacto_hip.so
  timeout 200 python bench.py --cpu-batches 0 --min-time 0.3 "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-14s' % '$n', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['kernel_ms_per_step'].items()}, 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round(d['roofline_backward']['avg_launch_ms'],4))
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-600:])"
}
for r in 8 7 1 2 4; do run role${r}_np role$r --no-pipeline; done
cp /tmp/keep.so $R/difacto_amd/libdifacto_hip.so
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o kt -- python $R/bench.py --cpu-batches 0 --min-time 0.3 > $O/prof_c3.log 2>&1
DB=$(ls $O/prof_c3/*.db $O/prof_c3/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_stats.py $DB $O/kernel_stats_c3_pipelined.txt > /dev/null 2>&1
for k in 5 9 13; do python $R/tools/rocpd_timeline.py $DB k_forward $k $O/timeline_c3_pipelined_$k.txt > /dev/null 2>&1; done
python $R/tools/rocpd_overlap.py $DB $O/overlap_c3_pipelined.txt > /dev/null 2>&1
cat $O/kernel_stats_c3_pipelined.txt | head -30
cat $O/timeline_c3_pipelined_5.txt
find $O -name "*.db" -delete; rm -rf $O/prof_c3

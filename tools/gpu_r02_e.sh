#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r02e; mkdir -p $O; cd $R
cp difacto_amd/libdifacto_hip.so /tmp/keep.so
for v in v0; do
  cp $R/tools/var_$v.so $R/difacto_amd/libdifacto_hip.so
  cd $R
  timeout 200 python bench.py --cpu-batches 0 --no-pipeline --no-relocalize > $O/norel_$v.json 2> $O/norel_$v.err
  python -c "
import json
d=json.loads(open('$O/norel_$v.json').read().strip().splitlines()[-1])
print('$v main-only', round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['kernel_ms_per_step'].items()}, 'live', round(d['roofline']['avg_launch_ms'],4), round(d['roofline_backward']['avg_launch_ms'],4))"
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace -d $O/prof_pipe_$v -o kt -- python $R/bench.py --steps 100 --warmup 20 --cpu-batches 0 --no-timing --min-time 0.05 > $O/prof_pipe_$v.log 2>&1
  python $R/tools/rocpd_overlap.py $O/prof_pipe_$v/kt_results.db $O/overlap_$v.txt
  python $R/tools/rocpd_timeline.py $O/prof_pipe_$v/kt_results.db k_forward 5 $O/timeline_$v.txt > /dev/null
done
cp /tmp/keep.so $R/difacto_amd/libdifacto_hip.so

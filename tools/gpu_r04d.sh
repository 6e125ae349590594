#!/bin/bash
# round 4, fourth call: GPU suite (AUC units without atomics + lazy finalisation, fat list entries, growth test), then same-box A/B:
# head (before fat entries / XCD-aligned role layout) vs new, without AUC; new with AUC riding / own launch; aligned interleave; misalignment
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04d; mkdir -p $O; cd $R
( time DFH_PARITY_RECORD=$O/parity.json timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log | cut -c1-300; grep -E "^E  |^FAILED" $O/pytest_gpu.log | head -40
cp $R/difacto_amd/libdifacto_hip.so /tmp/keep.so
line() {  # name args...
  n=$1; shift
  timeout 300 python bench.py --cpu-batches 0 --min-time 1 --no-secondary "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-26s' % '$n', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(x,4) for k,x in (d.get('kernel_ms_per_step') or {}).items()}, 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round((d.get('roofline_backward') or {}).get('avg_launch_ms',0),4))
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-800:])"
}
cp $R/tools/var_head.so $R/difacto_amd/libdifacto_hip.so
line head_no_auc --no-auc
line head_no_auc_serial --no-auc --no-pipeline
cp $R/tools/var_new.so $R/difacto_amd/libdifacto_hip.so
line new_no_auc --no-auc
line new_no_auc_serial --no-auc --no-pipeline
line new_auc_rides
line new_auc_own --ctx-option auc_in_update=0
line new_auc_rides_serial --no-pipeline
line new_auc_own_serial --no-pipeline --ctx-option auc_in_update=0
line new_ileave2 --no-auc --ctx-option upd_interleave=2
line new_ileave3 --no-auc --ctx-option upd_interleave=3
line new_ileave2_serial --no-auc --no-pipeline --ctx-option upd_interleave=2
line new_misalign1 --no-auc --ctx-option upd_misalign=1
line new_misalign4_serial --no-auc --no-pipeline --ctx-option upd_misalign=4
cp $R/tools/var_head.so $R/difacto_amd/libdifacto_hip.so
line head_no_auc_again --no-auc
cp /tmp/keep.so $R/difacto_amd/libdifacto_hip.so
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3_np -o kt -- python $R/bench.py --steps 100 --warmup 20 --cpu-batches 0 --no-pipeline --min-time 0.05 --no-secondary --ctx-option auc_in_update=0 > $O/prof_c3_np.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_c3_np/*.db $O/prof_c3_np/*/*.db 2>/dev/null | head -1) $O/kernel_stats_c3_serial_auc_own_launch.txt > /dev/null 2>&1
head -9 $O/kernel_stats_c3_serial_auc_own_launch.txt | cut -c1-200
find $O -name "*.db" -delete; rm -rf $O/prof_c3_np
du -sh $O

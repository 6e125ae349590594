#!/bin/bash
# round 5: one overlapped N = 8 step with the wire model OFF as a timeline (what the main stream waits for)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05o; mkdir -p $O; cd /tmp
DFH_EMUL_MODELS=off timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o kt -- python $R/bench.py --emulate-world 8 --steps 50 --warmup 10 --min-time 0.3 --no-timing > $O/prof.log 2>&1
DB=$(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_timeline.py $DB k_forward 5 $O/timeline_emulated_w8_overlap_wires_off.txt > /dev/null 2>&1
cat $O/timeline_emulated_w8_overlap_wires_off.txt
find $O -name "*.db" -delete; rm -rf $O/prof

#!/bin/bash
# kernel stats of the C5 slice (pipelined); build/difacto end to end on longer files (criteo text, .rec), old binary vs new
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03ap; mkdir -p $O; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o kt -- python $R/bench.py --preset c5-slice --cpu-batches 0 --min-time 0.5 --no-secondary > $O/prof_c5.log 2>&1
DB=$(ls $O/prof_c5/*.db $O/prof_c5/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_stats.py $DB $O/kernel_stats_c5_slice_pipelined.txt > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $DB k_forward 5 $O/timeline_c5_slice_pipelined.txt > /dev/null 2>&1
head -12 $O/kernel_stats_c5_slice_pipelined.txt | cut -c1-170; cat $O/timeline_c5_slice_pipelined.txt | cut -c1-150
find $O -name "*.db" -delete; rm -rf $O/prof_c5
cd $R
E2E_FORMATS=criteo,rec E2E_EXES=difacto_old,difacto timeout 1500 python tools/e2e_cli.py 400000 48 > $O/e2e.jsonl 2> $O/e2e.err
python -c "
import json
for l in open('$O/e2e.jsonl'):
    d=json.loads(l); print(d['format'], d['exe'], 'steady %.2f M rows/s' % (d['steady_rows_per_s']/1e6), 'big %.2f s' % d['wall_s_big'], 'small %.2f s' % d['wall_s'], d['rc'], d['rc_big'])"
tail -2 $O/e2e.err

#!/bin/bash
# round 4: 1 / 2 / 3 preparation streams, end to end (where the preparation chain carries the row gather too) and in bench.py
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04x; mkdir -p $O; cd $R
E2E_FORMATS=rec,criteo E2E_VARIANTS="s1:DIFACTO_PREP_STREAMS=1,s2:DIFACTO_PREP_STREAMS=2,s3:DIFACTO_PREP_STREAMS=3" E2E_EXES=difacto@s1,difacto@s2,difacto@s3,difacto@s1 bash tools/gpu_r04w.sh 2>&1 | grep -v throttled
for n in 1 2 3 1; do
  timeout 300 python bench.py --cpu-batches 0 --min-time 1 --no-secondary --prep-streams $n > $O/b_ps$n.json 2> $O/b_ps$n.err
  python -c "
import json
d=json.loads(open('$O/b_ps$n.json').read().strip().splitlines()[-1]); print('bench prep-streams $n', round(d['value']/1e6,2), 'M ex/s', round(d['ms_per_step'],4))"
done

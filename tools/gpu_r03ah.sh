#!/bin/bash
# row-gather threads of the reader: in-loop profile of build/difacto on the .rec and criteo files
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03ah; mkdir -p $O; cd $R
for g in 1 2 4 8; do
  E2E_FORMATS=rec,criteo DIFACTO_GATHER_THREADS=$g DIFACTO_PROFILE=1 timeout 900 python tools/e2e_cli.py 400000 16 > $O/e2e_g$g.jsonl 2> $O/e2e_g$g.err
  echo "gather threads $g"
  python -c "
import json
for l in open('$O/e2e_g$g.jsonl'):
    d=json.loads(l); print(' ', d['format'], 'steady %.2f M rows/s' % (d['steady_rows_per_s']/1e6), 'big %.2f s' % d['wall_s_big'])"
  grep -E "host loop over 640|10000 rows, shuffle" $O/e2e_g$g.err | sed -e 's/^.*host loop/  host loop/' -e 's/^.*batch reader/  batch reader/' | cut -c1-200
done

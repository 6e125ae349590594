#!/bin/bash
# parser threads: criteo text end to end with 8 / 16 / 32 / 48 parser threads
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03as; mkdir -p $O; cd $R
for p in 16 32 48; do
  E2E_FORMATS=criteo,rec DIFACTO_PARSER_THREADS=$p DIFACTO_PROFILE=1 timeout 900 python tools/e2e_cli.py 400000 32 > $O/e2e_p$p.jsonl 2> $O/e2e_p$p.err
  echo "parser threads $p"
  python -c "
import json
for l in open('$O/e2e_p$p.jsonl'):
    d=json.loads(l); print(' ', d['format'], 'steady %.2f M rows/s' % (d['steady_rows_per_s']/1e6), 'big %.2f s' % d['wall_s_big'], 'small %.2f s' % d['wall_s'])"
  grep -E "host loop over 1280|reader: 1280|reader: [0-9]+ chunks" $O/e2e_p$p.err | sed -e 's/^.*host loop/  host loop/' -e 's/^.*reader: /  reader: /' | cut -c1-180 | tail -4
done

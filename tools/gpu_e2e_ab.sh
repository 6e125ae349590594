#!/bin/bash
# build/difacto end to end on the .rec file under environment variants (tools/e2e_cli.py's E2E_VARIANTS), one box
# usage: gpu_e2e_ab.sh <tag> "<name:K=V+K=V,name2:...>" [formats]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$1; mkdir -p $O; cd $R
python -c "from difacto_amd.build import build_hip, build_host; build_hip(); build_host()" 2>&1 | tail -1
names=$(echo "$2" | tr ',' '\n' | cut -d: -f1 | sed 's/^/difacto@/' | paste -sd, -)
DIFACTO_PROFILE=1 E2E_FORMATS=${3:-rec} E2E_VARIANTS="$2" E2E_EXES="difacto,$names,difacto" timeout 1500 python tools/e2e_cli.py 400000 48 > $O/e2e.jsonl 2> $O/e2e.err
python - $O/e2e.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print('%-8s %-22s loop clock: big %.1f M rows/s, steady %.1f M; wall steady %.1f M  rc %s %s' % (d['format'], d['exe'], d.get('loop_rows_per_s_big',0)/1e6, d.get('steady_rows_per_s_by_loop_clock',0)/1e6, d['steady_rows_per_s']/1e6, d['rc'], d['rc_big']))
PY

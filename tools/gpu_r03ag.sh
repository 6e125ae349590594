#!/bin/bash
# reader threads + 4-thread row gather: host + CLI tests, then build/difacto end to end, old binary vs new (same files, same box)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03ag; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -q -x -k "host_cpp or cli or ingest" ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^E  " $O/pytest_gpu.log | head -20
E2E_EXES=difacto_old,difacto DIFACTO_PROFILE=1 timeout 1500 python tools/e2e_cli.py 400000 16 > $O/e2e.jsonl 2> $O/e2e.err
python -c "
import json
for l in open('$O/e2e.jsonl'):
    d=json.loads(l); print(d['format'], d['exe'], 'steady %.2f M rows/s' % (d['steady_rows_per_s']/1e6), 'big %.2f s' % d['wall_s_big'], d['rc'], d['rc_big'], d['line_big'][-60:])"
grep -E "host loop|batch reader|reader:" $O/e2e.err | tail -8

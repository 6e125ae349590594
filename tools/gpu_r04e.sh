#!/bin/bash
# round 4: device feed with one preparation call per minibatch (dfh_batch_prepare_rows) against the three calls of round 3
# (DIFACTO_SPLIT_PREP=1), same files, same box; the unit / CLI tests of the feed first
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04e; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -q -k "row_gather or host_cpp or cli or growing or nonfinite or fused_step_auc" ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^E  " $O/pytest_gpu.log | head -20
E2E_FORMATS=criteo,rec E2E_VARIANTS="split:DIFACTO_SPLIT_PREP=1" E2E_EXES=difacto@split,difacto DIFACTO_PROFILE=1 timeout 1500 python tools/e2e_cli.py 400000 48 > $O/e2e.jsonl 2> $O/e2e.err
python -c "
import json
for l in open('$O/e2e.jsonl'):
    d=json.loads(l); print(d['format'], d['exe'], 'steady %.2f M rows/s' % (d['steady_rows_per_s']/1e6), 'big %.2f s' % d['wall_s_big'], 'small %.2f s' % d['wall_s'], d['rc'], d['rc_big'], d['line_big'][-50:])"
grep -E "host loop over 1920|10000 rows, shuffle|reader: " $O/e2e.err | cut -c1-260 | tail -24

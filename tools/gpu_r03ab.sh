#!/bin/bash
# new tall-ragged parity cases; build/difacto end to end from files on the final tree
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03ab; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -q -x -k "tall_ragged or host_cpp or cli" ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^E  " $O/pytest_gpu.log | head -20
timeout 900 python tools/e2e_cli.py 200000 16 > $O/e2e.jsonl 2> $O/e2e.err; cat $O/e2e.jsonl; tail -2 $O/e2e.err

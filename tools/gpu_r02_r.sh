#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out/r02r; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3; grep -E "^E |FAILED" $O/pytest_gpu.log | head -20
timeout 600 python tools/e2e_cli.py 200000 > $O/e2e.jsonl 2> $O/e2e.err; cat $O/e2e.jsonl; tail -2 $O/e2e.err

#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03ao; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -q -x -k "sample_sort" ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^E  " $O/pytest_gpu.log | head -20; tail -5 $O/pytest_gpu.log

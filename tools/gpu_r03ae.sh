#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03ae; mkdir -p $O; cd $R
python -c "from difacto_amd import build; build.build_host()" > $O/build.log 2>&1
timeout 600 python tools/determinism_cli.py 3 2>&1 | tee $O/determinism.txt
( time timeout 900 python -m pytest tests -m gpu -q -x -k "host_cpp or cli" ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^E  " $O/pytest_gpu.log | head -20

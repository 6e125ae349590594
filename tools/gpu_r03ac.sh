#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03ac; mkdir -p $O; cd $R
python -c "from difacto_amd import build; build.build_host()" > $O/build.log 2>&1
timeout 600 python tools/determinism_cli.py 4 2>&1 | tee $O/determinism.txt

#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R
bash tools/gpu_variants.sh r02ae "|--no-pipeline" base lateload
bash tools/gpu_variants.sh r02ae "" depth1 mid256 mid1024 small16

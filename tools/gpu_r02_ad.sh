#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R
bash tools/gpu_variants.sh r02ad "|--no-pipeline" base sort32

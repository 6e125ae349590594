#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R
bash tools/gpu_variants.sh r02aa "|--prep-streams 2|--ctx-option fwd_depth=4" bw8
bash tools/gpu_variants.sh r02aa "|--ctx-option fwd_depth=4" bw8fw6
bash tools/gpu_variants.sh r02aa "--ctx-option fwd_depth=4" bw8fw8
bash tools/gpu_variants.sh r02aa "" base

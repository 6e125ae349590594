#!/bin/bash
# Round 2, GPU call W: A/B of the overlap between the preparation stream and the step
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R
bash tools/gpu_variants.sh r02w "|--prep-gate|--prep-streams 2|--prep-streams 2 --prep-gate|--no-prep-lookup|--no-prep-lookup --prep-gate" base
bash tools/gpu_variants.sh r02w "--ctx-option bwd_small_blocks=4096|--ctx-option bwd_small_blocks=4096 --prep-gate|--ctx-option bwd_small_blocks=4096 --no-pipeline" bt256
bash tools/gpu_variants.sh r02w "|--prep-gate" cap256

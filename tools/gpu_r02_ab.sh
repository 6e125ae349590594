#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r02ab; mkdir -p $O; cd $R
( time timeout 600 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log | cut -c1-300; grep -E "^E |^FAILED" $O/pytest_gpu.log | head -30
bash tools/gpu_variants.sh r02ab "|--ctx-option bwd_small_blocks=1024|--ctx-option bwd_small_blocks=3072|--ctx-option bwd_small_blocks=4096|--no-pipeline|--preset c5-slice --ids 60000000|--preset c5-slice --ids 60000000 --no-pipeline" base
timeout 200 python bench.py --force-sharded --steps 200 --warmup 20 > $O/bench_sharded_w1.json 2> $O/bench_sharded_w1.err; python -c "
import json
d=json.loads(open('$O/bench_sharded_w1.json').read().strip().splitlines()[-1]); print('sharded w1', d['value']/1e6, d['ms_per_step'])"

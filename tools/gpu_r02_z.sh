#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r02z; mkdir -p $O; cd $R
( time timeout 600 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log | cut -c1-300; grep -E "^E |^FAILED" $O/pytest_gpu.log | head -30
for i in 1 2 3; do timeout 300 python -m pytest tests/test_host_cpp.py tests/test_shard_native.py -m gpu -q 2>&1 | tail -1; done
cp difacto_amd/libdifacto_hip.so tools/var_base.so
bash tools/gpu_variants.sh r02z "|--no-pipeline" base bw8
timeout 600 python tools/e2e_cli.py 200000 8 > $O/e2e.jsonl 2> $O/e2e.err; cat $O/e2e.jsonl | cut -c1-600; tail -2 $O/e2e.err

#!/bin/bash
# Round 2, GPU call X: counts of the following step exchanged inside the running one (dfh_shard_prefetch_counts),
# C++ sharded loop with two batch objects
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r02x; mkdir -p $O; cd $R
( time timeout 600 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log | cut -c1-300; grep -E "^E |^FAILED" $O/pytest_gpu.log | head -30
timeout 300 ./build/difacto_host_tests tests/golden/rcv1_100.libsvm > $O/host_tests.log 2>&1; tail -2 $O/host_tests.log
timeout 200 python bench.py --force-sharded --steps 100 --warmup 10 > $O/w1_native.json 2> $O/w1_native.err; tail -c 300 $O/w1_native.json | head -c 200; tail -2 $O/w1_native.err
DFH_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 4 --steps 6 --warmup 2 --ids 2000000 > $O/w4_dry.json 2> $O/w4_dry.err; python -c "
import json
d=json.loads(open('$O/w4_dry.json').read().strip().splitlines()[-1]); print('w4 dry', d['value'], d['ms_per_step'], d['train_logloss_per_example'])"; grep -v "NCCL\|longer_path\|^$\|amdgpu.ids\|socket.cpp\|Gloo\|\*\*\*" $O/w4_dry.err | tail -5 | cut -c1-300

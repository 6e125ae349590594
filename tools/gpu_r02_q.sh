#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); O=$R/gpurun_out/r02q; mkdir -p $O; cd $R
timeout 200 python bench.py --force-sharded --steps 100 --warmup 10 > $O/w1_native.json 2> $O/w1_native.err; tail -c 700 $O/w1_native.json; tail -3 $O/w1_native.err
DFH_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 4 --steps 6 --warmup 2 --ids 2000000 > $O/w4_dry.json 2> $O/w4_dry.err; tail -c 600 $O/w4_dry.json; grep -v "NCCL\|longer_path\|^$\|amdgpu.ids" $O/w4_dry.err | tail -8 | cut -c1-300
timeout 200 python bench.py --force-sharded --transport torch --steps 100 --warmup 10 > $O/w1_torch.json 2> $O/w1_torch.err; tail -c 300 $O/w1_torch.json

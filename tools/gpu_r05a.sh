#!/bin/bash
# round 5, first GPU call: the loop-back emulation works at all + the 8-rank shared-GPU replay + this box's default line
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests/test_shard_native.py -x -q -m gpu -k "share_one_gpu and 8-" > $O/pytest_w8.txt 2>&1; tail -5 $O/pytest_w8.txt
timeout 600 python bench.py --emulate-world 8 --emulate-rank 0 --steps 50 --warmup 10 --min-time 1.5 > $O/emul_w8_r0.json 2> $O/emul_w8_r0.err; tail -c 3000 $O/emul_w8_r0.json; tail -5 $O/emul_w8_r0.err
timeout 600 python bench.py --emulate-world 8 --emulate-rank 0 --exchange sync --steps 50 --warmup 10 --min-time 1.5 > $O/emul_w8_r0_sync.json 2> $O/emul_w8_r0_sync.err; tail -c 1500 $O/emul_w8_r0_sync.json; tail -5 $O/emul_w8_r0_sync.err
timeout 600 python bench.py --no-secondary --cpu-batches 0 > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 1500 $O/bench_c3.json

#!/bin/bash
# Round 2, GPU call V: sharded step with the rank's own keys on its table (k_forward<MIXED>, two backward launches),
# pair-counting AUC, four-launch Localizer restored
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r02v; mkdir -p $O; cd $R
( time timeout 600 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log | cut -c1-300; grep -E "^E |^FAILED" $O/pytest_gpu.log | head -30
timeout 200 python bench.py --force-sharded --steps 100 --warmup 10 > $O/w1_native.json 2> $O/w1_native.err; tail -c 500 $O/w1_native.json; tail -3 $O/w1_native.err
DFH_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 4 --steps 6 --warmup 2 --ids 2000000 > $O/w4_dry.json 2> $O/w4_dry.err; tail -c 400 $O/w4_dry.json; grep -v "NCCL\|longer_path\|^$\|amdgpu.ids" $O/w4_dry.err | tail -8 | cut -c1-300
for mode in "--no-pipeline" ""; do
  n=$(echo "b$mode" | tr -d ' -')
  timeout 200 python bench.py --cpu-batches 0 $mode > $O/$n.json 2> $O/$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
  print('[%s]' % '$mode', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['kernel_ms_per_step'].items()}, 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round(d['roofline_backward']['avg_launch_ms'],4))
except Exception as e: print('[$mode] ERR', e); print(open('$O/$n.err').read()[-600:])"
done

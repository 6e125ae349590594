#!/bin/bash
# Round profiles: kernel-trace stats of the default bench command, and the two PMC passes for HBM traffic.
# usage (on the GPU box, from the repo root): bash tools/collect_profiles.sh r01
TAG=${1:-r01}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp
# the default bench command itself (what the driver runs), under the kernel trace
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_final -o kt -- python $R/bench.py > $R/gpurun_out/prof_${TAG}_final.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_final_np -o kt -- python $R/bench.py --steps 100 --warmup 20 --cpu-batches 0 --no-pipeline > $R/gpurun_out/prof_${TAG}_final_np.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_${TAG}_$c -o pmc -- python $R/bench.py --steps 20 --warmup 5 --cpu-batches 0 --no-pipeline --no-timing > $R/gpurun_out/pmc_${TAG}_$c.log 2>&1
done
cd $R
python bench.py > gpurun_out/bench_${TAG}_default.json 2> gpurun_out/bench_${TAG}_default.err
tail -c 2500 gpurun_out/bench_${TAG}_default.json
# L2 behaviour of the hot kernels (hit / miss / request counts), serial run
cd /tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_${TAG}_l2_$i -o pmc -- python $R/bench.py --steps 20 --warmup 5 --cpu-batches 0 --no-pipeline --no-timing > $R/gpurun_out/pmc_${TAG}_l2_$i.log 2>&1
done
cd $R

export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcx_$i -o p -- python $R/bench.py --steps 10 --warmup 3 --cpu-batches 0 --no-pipeline --no-timing > $R/gpurun_out/pmcx_$i.log 2>&1
done
timeout 100 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace -d $R/gpurun_out/pmcx_gb -o p -- $R/tools/gather_bench.bin 147000 > $R/gpurun_out/pmcx_gb.log 2>&1
ls $R/gpurun_out | grep pmcx

#!/bin/bash
# round 5: owner-side kernels alone (tools/owner_bench.py) under the kernel trace, one library variant after the other
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05e; mkdir -p $O; cd /tmp
for v in "$@"; do
  L=$R/tools/var_$v.so; [ -f $L ] || L=$R/difacto_amd/libdifacto_hip.so   # an unknown name: the product library
  DIFACTO_HIP_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats -d $O/p_$v -o kt -- python $R/tools/owner_bench.py 120 > $O/$v.log 2>&1
  grep owner_bench $O/$v.log
  python $R/tools/rocpd_stats.py $(ls $O/p_$v/*.db $O/p_$v/*/*.db 2>/dev/null | head -1) $O/stats_$v.txt > /dev/null 2>&1
  echo "== $v"; grep -E "k_push_grad_multi|k_push_count_multi|k_pull_resolved|k_resolve_multi" $O/stats_$v.txt | cut -c1-60,90-150
  rm -rf $O/p_$v
done

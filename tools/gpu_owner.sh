#!/bin/bash
# the owner side of the sharded store per distinct key against per entry: parity, the kernels alone (tools/owner_bench.py under
# rocprofv3), the N = 8 projection both ways on one box.   usage: gpurun -- bash tools/gpu_owner.sh [tag]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-owner}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
python -c "from difacto_amd.build import build_hip; build_hip()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_shard_native.py tests/test_loopback_emulation.py -m gpu -q -x -k "owner_side or shard or loopback or resolve_multi" > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-300
grep -E "^E " $O/pytest.log | head -20
cd /tmp
for m in entry listed; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$m -o kt -- python $R/tools/owner_bench.py 200 64 8 $m > $O/owner_$m.log 2>&1
  tail -1 $O/owner_$m.log | cut -c1-400
  python $R/tools/rocpd_stats.py $(ls $O/prof_$m/*.db $O/prof_$m/*/*.db 2>/dev/null | head -1) $O/kernel_stats_owner_$m.txt > /dev/null 2>&1
  grep -E "k_resolve_multi|k_push_count|k_pull_resolved|k_push_grad|k_count_pull" $O/kernel_stats_owner_$m.txt | cut -c1-200
  rm -rf $O/prof_$m
done
cd $R
for m in entry listed; do
  if [ $m = listed ]; then export DFH_OWNER_PER_KEY=1; else unset DFH_OWNER_PER_KEY; fi
  timeout 900 python bench.py --emulate-world 8 --emulate-rank auto --cpu-batches 0 --min-time 1 > $O/emul_c4_w8_$m.json 2> $O/emul_c4_w8_$m.err
  python -c "
import json
try:
  d=json.loads(open('$O/emul_c4_w8_$m.json').read().strip().splitlines()[-1])
  print('emulated N=8 $m', round(d['value']/1e6,1), d['ms_per_step'], {k:round(v/1e6,1) for k,v in d['projected_examples_per_sec'].items()}, d.get('stage_ms_per_step'))
except Exception as e: print('emul ERR', e); print(open('$O/emul_c4_w8_$m.err').read()[-800:])"
done

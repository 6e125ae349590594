#!/bin/bash
# round 5: the bucket sort of the Localizer (k_loc_sort) with register ranking + interleaved merge searches (new; avg384: the
# same without the half-size buckets of small minibatches), against the build before it (tools/var_oldsort.so), same box: Localizer parity, then C2 and C3 lines and kernel stats for both
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05v; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_kernel_parity.py -m gpu -q -x -k "localizer or fused_first_step or hot_kernels" ) > $O/pytest_localizer.log 2>&1
grep -E "passed|failed" $O/pytest_localizer.log | tail -2; grep -E "^E |^FAILED" $O/pytest_localizer.log | head -10
for lib in new avg384 old; do
  unset DIFACTO_HIP_LIB; [ $lib = old ] && export DIFACTO_HIP_LIB=$R/tools/var_oldsort.so; [ $lib = avg384 ] && export DIFACTO_HIP_LIB=$R/tools/var_avg384.so
  for p in c2 c3; do
    timeout 300 python bench.py --preset $p --cpu-batches 0 --min-time 1.5 --no-secondary > $O/bench_${p}_$lib.json 2> $O/bench_${p}_$lib.err
    python -c "
import json
d=json.loads(open('$O/bench_${p}_$lib.json').read().strip().splitlines()[-1]); print('$lib $p', round(d['value']/1e6,3), 'M ex/s', round(d['ms_per_step']*1e3,1), 'us/step', {k: round(v*1e3,1) for k,v in d['kernel_ms_per_step'].items()})"
  done
done
cd /tmp
for lib in new avg384 old; do
  unset DIFACTO_HIP_LIB; [ $lib = old ] && export DIFACTO_HIP_LIB=$R/tools/var_oldsort.so; [ $lib = avg384 ] && export DIFACTO_HIP_LIB=$R/tools/var_avg384.so
  for p in c2 c3; do
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_${p}_$lib -o kt -- python $R/bench.py --preset $p --cpu-batches 0 --min-time 0.3 --no-secondary --no-timing > $O/prof_${p}_$lib.log 2>&1
    DB=$(ls $O/prof_${p}_$lib/*.db $O/prof_${p}_$lib/*/*.db 2>/dev/null | head -1)
    python $R/tools/rocpd_stats.py $DB $O/kernel_stats_${p}_$lib.txt > /dev/null 2>&1
    rm -rf $O/prof_${p}_$lib
    echo "$lib $p: $(grep k_loc_sort $O/kernel_stats_${p}_$lib.txt | awk '{print "k_loc_sort avg", $(NF-9), "us"}')"
  done
done

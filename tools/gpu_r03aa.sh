#!/bin/bash
# one 128 B header line per key, rewritten whole by the update kernel (-DDFH_HDR_BYTES=128) against the 32 B header: tests, A/B
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03aa; mkdir -p $O; cd $R
cp $R/difacto_amd/libdifacto_hip.so /tmp/keep.so
cp $R/tools/var_hdr128.so $R/difacto_amd/libdifacto_hip.so
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu_hdr128.log 2>&1
grep -E "passed|failed" $O/pytest_gpu_hdr128.log | tail -2; grep -E "^FAILED|^E  " $O/pytest_gpu_hdr128.log | head -20
run() {  # name variant args...
  n=$1; v=$2; shift 2
  cp $R/tools/var_$v.so $R/difacto_amd/libdifacto_hip.so
  timeout 300 python bench.py --cpu-batches 0 --min-time 0.5 --no-secondary "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-14s' % '$n', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['kernel_ms_per_step'].items()}, 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round(d['roofline_backward']['avg_launch_ms'],4))
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-600:])"
}
run h32 hdr32; run h128 hdr128; run h32_np hdr32 --no-pipeline; run h128_np hdr128 --no-pipeline; run h32_b hdr32; run h128_b hdr128
run h32_c5 hdr32 --preset c5-slice; run h128_c5 hdr128 --preset c5-slice
cp /tmp/keep.so $R/difacto_amd/libdifacto_hip.so

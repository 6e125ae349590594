#!/bin/bash
# round 4: large size class of the sample-sort Localizer (bit-exact tests), timing of B = 20 000 C3 rows (sample sort vs the library path), full GPU suite
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04q; mkdir -p $O; cd $R
( time DFH_PARITY_RECORD=$O/parity.json timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log | cut -c1-300; grep -E "^E  |^FAILED" $O/pytest_gpu.log | head -40
line() {  # name args...
  n=$1; shift
  timeout 300 python bench.py --cpu-batches 0 --min-time 1 --no-secondary "$@" > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
try:
  d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1])
  print('%-26s' % '$n', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k:round(x,4) for k,x in (d.get('kernel_ms_per_step') or {}).items()}, 'live fwd/bwd', round(d['roofline']['avg_launch_ms'],4), round((d.get('roofline_backward') or {}).get('avg_launch_ms',0),4))
except Exception as e: print('$n ERR', e); print(open('$O/b_$n.err').read()[-800:])"
}
line c3
line c3_rows20000 --rows 20000 --distinct 64
line c3_rows20000_serial --rows 20000 --distinct 64 --no-pipeline
line c3_rows40000_serial --rows 40000 --distinct 32 --no-pipeline
line c3_serial --no-pipeline

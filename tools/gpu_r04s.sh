#!/bin/bash
# round 4: AUC riding in the sharded step's own-keys update; the sharded tests; the 1-rank sharded lines (with cpu_baseline + roofline_exchange)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04s; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -q -k "shard or sharded or auc or host_cpp" ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^E  " $O/pytest_gpu.log | head -20
for m in overlap sync; do
  timeout 300 python bench.py --cpu-batches 0 --min-time 1 --no-secondary --force-sharded --exchange $m > $O/bench_sharded_w1_native_$m.json 2> $O/bench_sharded_w1_native_$m.err
  python -c "
import json
d=json.loads(open('$O/bench_sharded_w1_native_$m.json').read().strip().splitlines()[-1])
print('$m', round(d['value']/1e6,2), round(d['ms_per_step'],4), d.get('stage_ms_per_step'))"
done
( time timeout 600 python bench.py --force-sharded --min-time 1 ) > $O/bench_sharded_w1_full.json 2> $O/bench_sharded_w1_full.err
python -c "
import json
d=json.loads(open('$O/bench_sharded_w1_full.json').read().strip().splitlines()[-1])
print('full', round(d['value']/1e6,2), 'cpu', (d['cpu_baseline'] or {}).get('value'), 'rx', {k:d['roofline_exchange'][k] for k in ('bytes_per_gpu_step','exchange_ms_per_step','achieved','peak')}, 'traffic', d['roofline']['traffic'], d['roofline'].get('traffic_source'), d['config']['transport_bound'])"
line() { n=$1; shift; timeout 300 python bench.py --cpu-batches 0 --min-time 1 --no-secondary "$@" > $O/b_$n.json 2> $O/b_$n.err; python -c "
import json
d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1]); print('$n', round(d['value']/1e6,2))"; }
line c3

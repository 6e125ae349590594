#!/bin/bash
# round 5: last check of the final tree — the whole GPU suite with both parity records, smoke, the default bench line
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05zz; mkdir -p $O; cd $R
( time DFH_PARITY_RECORD=$O/parity.json DFH_PARITY_RECORD_STEPS=$O/parity_steps.json timeout 2400 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^E |^FAILED" $O/pytest_gpu.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench_c3.json 2> $O/bench_c3.err; tail -3 $O/bench_c3.err
python - <<PY
import json
d=json.loads(open('$O/bench_c3.json').read().strip().splitlines()[-1])
print('default', round(d['value']/1e6,2), round(d['ms_per_step'],4), 'fwd frac', round(d['roofline']['frac'],3), 'bwd frac', round(d['roofline_backward']['frac'],3), 'requests frac', round(d['roofline_requests']['frac'],3))
for k,v in (d.get('secondary') or {}).items(): print(' secondary', k, {a:(round(b/1e6,2) if a=='value' else b) for a,b in v.items() if a in ('value','ms_per_step','wall_seconds','error')}, (v.get('cpu_baseline') or {}).get('value'))
print(' cpu', d['cpu_baseline']['value'])
PY
timeout 600 python bench.py --force-sharded --min-time 1 > $O/bench_sharded_w1_full.json 2> $O/bench_sharded_w1_full.err; python -c "
import json
d=json.loads(open('$O/bench_sharded_w1_full.json').read().strip().splitlines()[-1]); print('sharded w1', round(d['value']/1e6,2), d['config']['key_ranges'])"

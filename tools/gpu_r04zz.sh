#!/bin/bash
# round 4, last check of the tree as committed: the whole GPU suite (+ parity record), smoke, the default bench line, the
# 1-rank sharded line, the host binary's own tests
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04zz; mkdir -p $O; cd $R
( time DFH_PARITY_RECORD=$O/parity.json timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3 | cut -c1-300; grep -E "^E |^FAILED" $O/pytest_gpu.log | head -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; grep -E "smoke ok|Error|error" $O/smoke.log | tail -2
( time timeout 900 python bench.py ) > $O/bench_c3.json 2> $O/bench_c3.err; tail -3 $O/bench_c3.err
python -c "
import json
d=json.loads(open('$O/bench_c3.json').read().strip().splitlines()[-1])
print('default', round(d['value']/1e6,2), round(d['ms_per_step'],4), 'reps', d['repetitions'], 'fwd frac', round(d['roofline']['frac'],3), 'bwd frac', round(d['roofline_backward']['frac'],3), round(d['roofline_backward']['frac_hbm_necessary'],3), 'step frac', round(d['roofline_step']['frac'],3))
for k,v in (d.get('secondary') or {}).items(): print(' secondary', k, {a:(round(b/1e6,2) if a=='value' else b) for a,b in v.items() if a in ('value','ms_per_step','wall_seconds','error')})
print(' cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('scaled_threads',{}).get('value'), d['cpu_baseline'].get('scaled_threads',{}).get('threads'), d['cpu_baseline'].get('cpu_quota'))
"
( time timeout 600 python bench.py --force-sharded --min-time 1 ) > $O/bench_sharded_w1_full.json 2> $O/bench_sharded_w1_full.err; tail -2 $O/bench_sharded_w1_full.err
python -c "
import json
d=json.loads(open('$O/bench_sharded_w1_full.json').read().strip().splitlines()[-1]); print('sharded w1', round(d['value']/1e6,2), round(d['ms_per_step'],4))"

#!/bin/bash
# round 5: the whole GPU suite (with the step-by-step parity record), smoke, the default bench line (new secondary: c3-cold, c2)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05g; mkdir -p $O; cd $R
( time DFH_PARITY_RECORD=$O/parity.json DFH_PARITY_RECORD_STEPS=$O/parity_steps.json timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log | cut -c1-300; grep -E "^E |^FAILED" $O/pytest_gpu.log | head -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench_c3.json 2> $O/bench_c3.err; tail -3 $O/bench_c3.err
python - <<PY
import json
d=json.loads(open('$O/bench_c3.json').read().strip().splitlines()[-1])
print('default', round(d['value']/1e6,2), round(d['ms_per_step'],4), 'fwd frac', round(d['roofline']['frac'],3), 'bwd frac', round(d['roofline_backward']['frac'],3))
for k,v in (d.get('secondary') or {}).items(): print(' secondary', k, {a:(round(b/1e6,2) if a=='value' else b) for a,b in v.items() if a in ('value','ms_per_step','wall_seconds','error','steps')}, (v.get('cpu_baseline') or {}).get('value'))
print(' cpu', d['cpu_baseline']['value'])
PY

#!/bin/bash
# final check of the round's tree: full GPU suite, smoke, default bench line, 2-rank dry run (ranks share the GPU, gloo-staged exchange), 1-rank sharded line
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03al; mkdir -p $O; cd $R
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^E  " $O/pytest_gpu.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench_c3.json 2> $O/bench_c3.err; tail -3 $O/bench_c3.err
python -c "
import json
d=json.loads(open('$O/bench_c3.json').read().strip().splitlines()[-1])
print('default', round(d['value']/1e6,2), round(d['ms_per_step'],4), 'reps', d['repetitions'], 'fwd frac', round(d['roofline']['frac'],3), 'bwd frac', round(d['roofline_backward']['frac'],3), round(d['roofline_backward']['frac_hbm_necessary'],3), d['roofline']['traffic_source'])
for k,v in (d.get('secondary') or {}).items(): print(' secondary', k, {a:(round(b/1e6,2) if a=='value' else b) for a,b in v.items() if a in ('value','ms_per_step','wall_seconds','error')})
"
for ex in overlap sync; do
  DFH_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29917 bench.py --gpus 2 --steps 10 --warmup 3 --min-time 0.05 --ids 2000000 --exchange $ex > $O/dry_${ex}_2.json 2> $O/dry_${ex}_2.err
  python -c "
import json
try:
  d=json.loads(open('$O/dry_${ex}_2.json').read().strip().splitlines()[-1])
  print('dry $ex 2', round(d['value']/1e6,2), round(d['ms_per_step'],4), d['config']['exchange'][:8], 'logloss', round(d['train_logloss_per_example'],4), d.get('stage_ms_per_step'))
except Exception as e: print('dry $ex ERR', e); print(open('$O/dry_${ex}_2.err').read()[-1500:])"
done
timeout 200 python bench.py --force-sharded --steps 200 --warmup 20 --min-time 1 --no-secondary > $O/bench_sharded_w1.json 2> $O/bench_sharded_w1.err
python -c "
import json
d=json.loads(open('$O/bench_sharded_w1.json').read().strip().splitlines()[-1]); print('w1 sharded', round(d['value']/1e6,2), round(d['ms_per_step'],4), d.get('stage_ms_per_step'))"

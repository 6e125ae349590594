#!/bin/bash
# round 5: one device allocation per batch object (61 hipMallocs before): the job's start-up, end to end
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05l; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_host_cpp.py -x -q -m gpu 2>&1 | tail -2
DIFACTO_PROFILE=1 E2E_FORMATS=rec,criteo timeout 1200 python tools/e2e_cli.py 400000 48 > $O/e2e.jsonl 2> $O/e2e.err
python - <<PY
import json
for l in open("$O/e2e.jsonl"):
    d=json.loads(l); print(d["format"], d["exe"], "whole loop M rows/s", round(d.get("loop_rows_per_s_big",0)/1e6,1), "steady", round(d.get("steady_rows_per_s_by_loop_clock",0)/1e6,1), "loop_s small/big", d.get("loop_s"), d.get("loop_s_big"))
PY
grep "host loop over 40 " $O/e2e.err | tail -2
timeout 300 python bench.py --cpu-batches 0 --min-time 1 --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', round(d['value']/1e6,2))"

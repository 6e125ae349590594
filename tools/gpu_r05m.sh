#!/bin/bash
# round 5: a job's start-up — the code object loaded by a helper thread at dfh_ctx_create, the loop's batch objects out of one
# allocation — against DFH_WARM_LOAD=0; the GPU tests that create / destroy many objects
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05m; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_host_cpp.py tests/test_shard_native.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
E2E_VARIANTS="warm:,cold:DFH_WARM_LOAD=0" E2E_EXES="difacto@warm,difacto@cold" DIFACTO_PROFILE=1 E2E_FORMATS=rec,criteo timeout 1200 python tools/e2e_cli.py 400000 48 > $O/e2e.jsonl 2> $O/e2e.err
python - <<PY
import json
for l in open("$O/e2e.jsonl"):
    d=json.loads(l); print(d["format"], d["exe"], "whole loop M rows/s", round(d.get("loop_rows_per_s_big",0)/1e6,1), "steady", round(d.get("steady_rows_per_s_by_loop_clock",0)/1e6,1), "loop_s small/big", d.get("loop_s"), d.get("loop_s_big"), "wall small/big", round(d["wall_s"],3), round(d["wall_s_big"],3))
PY
grep -E "start-up" $O/e2e.err | awk 'NR%3==1' | head -8

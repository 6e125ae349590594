#!/bin/bash
# round 5: a GROWING table (table_capacity = 0, the C++ host's default) — the row bound refreshed from the device without draining
# the streams (default) against the drain every 32 launches (DFH_NO_BOUND_REFRESH=1), and the fixed capacity; the growing-table tests
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05q; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_host_cpp.py -x -q -m gpu -k "growing or grow or criteo_conf or capacity" > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1
E2E_TABLE_CAPACITY=0 E2E_VARIANTS="refresh:DFH_TRACE_RESERVE=1,drain:DFH_NO_BOUND_REFRESH=1+DFH_TRACE_RESERVE=1" E2E_EXES="difacto@refresh,difacto@drain,difacto@refresh,difacto@drain" DIFACTO_PROFILE=1 E2E_FORMATS=rec timeout 1200 python tools/e2e_cli.py 400000 48 > $O/e2e_grow.jsonl 2> $O/e2e_grow.err
DIFACTO_PROFILE=1 E2E_FORMATS=rec timeout 600 python tools/e2e_cli.py 400000 48 > $O/e2e_fixed.jsonl 2> $O/e2e_fixed.err
python - <<PY
import json
for f in ("e2e_grow","e2e_fixed"):
    for l in open("$O/%s.jsonl" % f):
        d=json.loads(l); print(f, d["format"], d["exe"], "whole loop M rows/s", round(d.get("loop_rows_per_s_big",0)/1e6,1), "steady", round(d.get("steady_rows_per_s_by_loop_clock",0)/1e6,1), "loop_s small/big", round(d.get("loop_s"),4), round(d.get("loop_s_big"),4), d["rc"], d["rc_big"])
PY

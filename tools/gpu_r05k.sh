#!/bin/bash
# round 5: small first chunks (default) against DIFACTO_CHUNK_RAMP=0 on one box, criteo text, alternating
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05k; mkdir -p $O; cd $R
E2E_VARIANTS="ramp:,noramp:DIFACTO_CHUNK_RAMP=0" E2E_EXES="difacto@ramp,difacto@noramp,difacto@ramp,difacto@noramp" DIFACTO_PROFILE=1 E2E_FORMATS=criteo timeout 1200 python tools/e2e_cli.py 400000 48 > $O/e2e.jsonl 2> $O/e2e.err
python - <<PY
import json
for l in open("$O/e2e.jsonl"):
    d=json.loads(l); print(d["format"], d["exe"], "whole loop M rows/s", round(d.get("loop_rows_per_s_big",0)/1e6,1), "steady", round(d.get("steady_rows_per_s_by_loop_clock",0)/1e6,1), "loop_s small/big", d.get("loop_s"), d.get("loop_s_big"))
PY

#!/bin/bash
# round 4: the GPU boxes give a container 16 cores' worth of CPU time (cpu.max 1600000 100000) on a 256-thread host:
# parser threads 8 / 10 / 12 / 16 under that quota, end to end by the worker loop's clock
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04w; mkdir -p $O; cd $R
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) | nproc $(nproc)"
DIFACTO_PROFILE=1 E2E_FORMATS=${E2E_FORMATS:-criteo,rec} E2E_VARIANTS="${E2E_VARIANTS:-p8:DIFACTO_PARSER_THREADS=8,p10:DIFACTO_PARSER_THREADS=10,p12:DIFACTO_PARSER_THREADS=12,p16:DIFACTO_PARSER_THREADS=16}" \
  E2E_EXES=${E2E_EXES:-difacto@p8,difacto@p10,difacto@p12,difacto@p16} timeout 900 python tools/e2e_cli.py 400000 48 > $O/e2e.jsonl 2> $O/e2e.err
python - <<PY
import json
for l in open("$O/e2e.jsonl"):
    d = json.loads(l)
    print(d["format"], d["exe"], "wall", round(d["wall_s"], 3), "big", round(d["wall_s_big"], 3), "| loop clock: small", round(d.get("loop_s", 0), 3), "big", round(d.get("loop_s_big", 0), 3), "s =", round(d.get("loop_rows_per_s_big", 0) / 1e6, 1), "M rows/s, steady", round(d.get("steady_rows_per_s_by_loop_clock", 0) / 1e6, 1), "rc", d["rc"], d["rc_big"])
PY
grep -E "reader: (1920|311)" $O/e2e.err | cut -c1-200 | sort | uniq -c | sort -rn | head -0
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | grep -E "throttled|nr_periods"

#!/bin/bash
# round 4: end-to-end A/B on one box: build/difacto (RecordIO records as views of the mapped file) against
# build/difacto_prev (fread under the reader's lock), .rec and criteo text, with the worker loop's profile lines
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04v; mkdir -p $O; cd $R
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) | nproc $(nproc) | $(grep Cpus_allowed_list /proc/self/status)"; lscpu | grep -E "Model name|Socket|Thread|NUMA node\(s\)" | head -5
DIFACTO_PROFILE=1 E2E_FORMATS=${E2E_FORMATS:-rec,criteo} E2E_EXES=${E2E_EXES:-difacto,difacto_prev} timeout 900 python tools/e2e_cli.py 400000 48 > $O/e2e.jsonl 2> $O/e2e.err
python - <<PY
import json
for l in open("$O/e2e.jsonl"):
    d = json.loads(l)
    print(d["format"], d["exe"], "wall", round(d["wall_s"], 3), "big", round(d["wall_s_big"], 3), "steady M rows/s", round(d["steady_rows_per_s"] / 1e6, 1), "| loop clock: big", round(d.get("loop_s_big", 0), 3), "s =", round(d.get("loop_rows_per_s_big", 0) / 1e6, 1), "M rows/s, steady", round(d.get("steady_rows_per_s_by_loop_clock", 0) / 1e6, 1), "rc", d["rc"], d["rc_big"], d["line_big"][:60])
PY
grep -E "host loop|reader: |batch reader" $O/e2e.err | cut -c1-300 | tail -40

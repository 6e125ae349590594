#!/bin/bash
# round 5: the two layout questions as microbenchmarks (merged index + header line; slot-major index), build/difacto end to end
# (small first chunks), the N > 1 bench code with 4 ranks sharing the GPU (dry run of `bench.py --gpus 4`), C2 after the probe skip
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05i; mkdir -p $O; cd $R
timeout 120 tools/lookup_line_bench.bin > $O/lookup_line_bench.txt 2>&1; cat $O/lookup_line_bench.txt
timeout 120 tools/index_layout_bench.bin > $O/index_layout_bench.txt 2>&1; cat $O/index_layout_bench.txt
cd /tmp; timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_idx -o pmc -- $R/tools/index_layout_bench.bin > /dev/null 2>&1
python - <<PY
import sqlite3,glob
db=sqlite3.connect(glob.glob("$O/pmc_idx/**/*.db",recursive=True)[0])
for k,n,a in db.execute("select kernel_name,count(*),avg(value) from counters_collection where counter_name='WRITE_SIZE' group by kernel_name"):
    print("WRITE_SIZE KiB/launch", k[:60], n, round(a,1))
PY
rm -rf $O/pmc_idx; cd $R
for X in "" ; do timeout 200 python bench.py --preset c2 --cpu-batches 0 --min-time 1 --no-secondary $X 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 [$X]', round(d['value']/1e6,3), 'M ex/s', round(d['ms_per_step']*1e3,1), 'us/step', 'enqueue', round(d['host_enqueue_ms_per_step']*1e3,1))"; done
DIFACTO_PROFILE=1 E2E_FORMATS=criteo,rec timeout 900 python tools/e2e_cli.py 400000 48 > $O/e2e.jsonl 2> $O/e2e.err
python - <<PY
import json
for l in open("$O/e2e.jsonl"):
    d=json.loads(l); print(d["format"], "whole loop M rows/s", round(d.get("loop_rows_per_s_big",0)/1e6,1), "steady", round(d.get("steady_rows_per_s_by_loop_clock",0)/1e6,1), "loop_s small/big", d.get("loop_s"), d.get("loop_s_big"), d["rc"], d["rc_big"])
PY
grep -E "host loop over" $O/e2e.err | tail -4
DFH_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 4 --steps 20 --warmup 5 --min-time 0.5 --cpu-batches 0 > $O/dryrun_w4.json 2> $O/dryrun_w4.err; tail -c 600 $O/dryrun_w4.json; tail -3 $O/dryrun_w4.err

#!/bin/bash
# round 6: the fused worker loop on the single queue for small minibatches (prepared two ahead): CLI tests, then build/difacto on the C2 shape
# (libsvm, batch 100, V_dim 8) and on Criteo rows at batch 2000, single queue forced off / on, same files and box
cd "$(dirname "$0")/.." && R=$PWD && O=$R/gpurun_out/r06m && mkdir -p $O
export TMPDIR=/tmp
python -c "from difacto_amd.build import build_hip, build_host; build_hip(); build_host()" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_host_cpp.py tests/test_ingest.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -4 | tee $O/pytest_host.txt
export DIFACTO_PROFILE=1
E2E_FORMATS=libsvm E2E_BATCH_SIZE=100 E2E_VDIM=8 E2E_EXES=difacto@tq,difacto@sq E2E_VARIANTS="tq:DIFACTO_SINGLE_QUEUE=0,sq:DIFACTO_SINGLE_QUEUE=1" timeout 900 python tools/e2e_cli.py 100000 4 > $O/e2e_c2shape.jsonl 2> $O/e2e_c2shape.err
grep 'host loop over 4000' $O/e2e_c2shape.err | sed 's/^.*difacto/difacto/' | cut -c1-230
E2E_FORMATS=rec E2E_BATCH_SIZE=2000 E2E_EXES=difacto@tq,difacto@sq E2E_VARIANTS="tq:DIFACTO_SINGLE_QUEUE=0,sq:DIFACTO_SINGLE_QUEUE=1" timeout 900 python tools/e2e_cli.py 400000 16 > $O/e2e_b2000.jsonl 2> $O/e2e_b2000.err
grep 'host loop over 3200' $O/e2e_b2000.err | sed 's/^.*difacto/difacto/' | cut -c1-230
python - $O/e2e_c2shape.jsonl $O/e2e_b2000.jsonl <<'PY'
import json,sys
for f in sys.argv[1:]:
    for l in open(f):
        d=json.loads(l)
        print(d["format"], d["exe"], "rows/s whole loop %.2f M steady %.2f M; loop_s big %.4f; wall big %.3f" % (d.get("loop_rows_per_s_big",0)/1e6, d.get("steady_rows_per_s_by_loop_clock",0)/1e6, d.get("loop_s_big",0), d["wall_s_big"]))
PY

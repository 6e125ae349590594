#!/bin/bash
# end-to-end CLI rate with 1 / 8 / 16 parser threads + the CLI and ingest tests (after a change to the host reader)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/e2e; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_host_cpp.py tests/test_ingest.py -q 2>&1 | tail -2
for t in 1 8 16; do
  DIFACTO_PARSER_THREADS=$t timeout 600 python tools/e2e_cli.py 200000 16 > $O/e2e_t$t.jsonl 2> $O/e2e_t$t.err
  python -c "
import json
for l in open('$O/e2e_t$t.jsonl'):
    d=json.loads(l); print('threads $t', d['format'], 'steady rows/s', round(d['steady_rows_per_s']), 'MB/s', round(d['steady_mb_per_s']), 'big wall', round(d['wall_s_big'],3), d['rc'], d['rc_big'])"
done

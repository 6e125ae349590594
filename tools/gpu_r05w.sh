#!/bin/bash
# round 5: the default line against a stream of 1 000 distinct minibatches (SURVEY 8d's wording), same box
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05w; mkdir -p $O; cd $R
for nd in 256 1000; do
  ( time timeout 600 python bench.py --distinct $nd --no-secondary --cpu-batches 0 ) > $O/bench_c3_distinct$nd.json 2> $O/bench_c3_distinct$nd.err
  python - <<PY
import json
d=json.loads(open('$O/bench_c3_distinct$nd.json').read().strip().splitlines()[-1])
print('distinct', d['config']['distinct_batches'], round(d['value']/1e6,2), 'M ex/s', round(d['ms_per_step'],4), 'ms/step; fwd', round(d['roofline']['frac'],3), 'requests', round(d['roofline_requests']['frac'],3))
PY
  grep real $O/bench_c3_distinct$nd.err
done

#!/bin/bash
# round 5: minibatches localized two steps ahead in the N > 1 loops (the keys of t+1 ready when step t starts) against one ahead
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05p; mkdir -p $O; cd $R
for A in 1 2 3; do
timeout 600 python bench.py --emulate-world 8 --steps 50 --warmup 10 --min-time 2.0 --shard-prep-ahead $A > $O/emul_w8_a$A.json 2> $O/emul.err
python - <<PY
import json
d=json.loads(open("$O/emul_w8_a$A.json").read().strip().splitlines()[-1])
r=d["ranks"][0]
print("[ahead $A]", "proj M ex/s", {m:round(v/1e6,1) for m,v in d["projected_examples_per_sec"].items()}, {m:round(v["ms_per_step"],4) for m,v in r["models"].items()}, r["stage_ms_per_step"].get("off"))
PY
done
timeout 300 python bench.py --force-sharded --cpu-batches 0 --min-time 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sharded w1', round(d['value']/1e6,2))"

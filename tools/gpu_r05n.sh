#!/bin/bash
# round 5: the others' row words folded into the own keys' lookup launch (overlapped exchange): parity, then the N = 8 projection again
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05n; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_loopback_emulation.py tests/test_shard_native.py tests/test_sharded_gpu_ranks.py tests/test_host_cpp.py -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed|rror" $O/pytest.txt | tail -3
for X in "" "--exchange sync"; do
timeout 600 python bench.py --emulate-world 8 --steps 50 --warmup 10 --min-time 2.0 $X > $O/emul_w8$(echo $X | tr -d ' -').json 2> $O/emul.err
python - <<PY
import json
d=json.loads(open("$O/emul_w8$(echo $X | tr -d ' -').json").read().strip().splitlines()[-1])
r=d["ranks"][0]
print("[$X]", "proj M ex/s", {m:round(v/1e6,1) for m,v in d["projected_examples_per_sec"].items()}, {m:round(v["ms_per_step"],4) for m,v in r["models"].items()}, r["stage_ms_per_step"].get("off"))
PY
done

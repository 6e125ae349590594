#!/bin/bash
# round 6: keys with more than 1 024 occurrences go through the update kernel part by part (upd_split_role): parity, the default line, and a
# Criteo-shaped stream with a key in every row (bench.py --bias-slot) with the split role and without it (ctx option upd_split = 0)
cd "$(dirname "$0")/.." && R=$PWD && O=$R/gpurun_out/r06p && mkdir -p $O
export TMPDIR=/tmp
python -c "from difacto_amd.build import build_hip, build_host; build_hip(); build_host()" 2>&1 | tail -2
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -4 | tee $O/pytest.txt
B="python bench.py --no-secondary --cpu-batches 0 --min-time 2"
run() { name=$1; shift; timeout 300 $B "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d.get("kernel_ms_per_step",{})
    print("%-28s %8.2f M ex/s  %.4f ms  fwd %.1f upd %.1f us | bk %s" % (sys.argv[2], d["value"]/1e6, d["ms_per_step"],
          (d["roofline"] or {}).get("avg_launch_ms",0)*1e3, (d["roofline_backward"] or {}).get("avg_launch_ms",0)*1e3,
          {a:round(b*1e3,1) for a,b in k.items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json",".err")).read()[-500:])
PY
}
run c3_split
run c3_nosplit --ctx-option upd_split=0
run c3_bias_split --bias-slots 2
run c3_bias_nosplit --bias-slots 2 --ctx-option upd_split=0
run c3_split_again

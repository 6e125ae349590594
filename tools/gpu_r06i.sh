#!/bin/bash
# round 6: the cold first epoch after the batched InitV epilogue; parity of the touched paths
cd "$(dirname "$0")/.." && R=$PWD && O=$R/gpurun_out/r06j && mkdir -p $O
export TMPDIR=/tmp
python -c "from difacto_amd.build import build_hip, build_host; build_hip(); build_host()" 2>&1 | tail -2
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee $O/pytest.txt
B="python bench.py --no-secondary --cpu-batches 0"
run() { name=$1; shift; timeout 300 $B "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d.get("kernel_ms_per_step",{})
    print("%-28s %8.2f M ex/s  %.4f ms  fwd %.1f upd %.1f us | bk %s" % (sys.argv[2], d["value"]/1e6, d["ms_per_step"],
          (d["roofline"] or {}).get("avg_launch_ms",0)*1e3, (d["roofline_backward"] or {}).get("avg_launch_ms",0)*1e3,
          {a:round(b*1e3,1) for a,b in k.items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
COLD="--preset c3 --no-prefill --steps 256 --warmup 0 --max-reps 1 --min-time 0 --distinct 256"
run cold_a $COLD
run cold_b $COLD
run warm --min-time 2
run cold_refdefaults --preset c3-refdefaults --no-prefill --steps 256 --warmup 0 --max-reps 1 --min-time 0 --distinct 256

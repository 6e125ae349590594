#!/bin/bash
# round 6: the whole GPU suite + the default bench line after the has-V flag (the one deviation retired), the ADVICE r5 fixes and the
# riders' arguments read through the kernarg segment
cd "$(dirname "$0")/.." && R=$PWD && O=$R/gpurun_out/r06e && mkdir -p $O
export TMPDIR=/tmp
python -c "from difacto_amd.build import build_hip, build_host; build_hip(); build_host()" 2>&1 | tail -2
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
python - $O/bench_c3.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c3 %.2f M ex/s %.4f ms; fwd %.1f us frac %.3f; upd %.1f us frac %.3f" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["avg_launch_ms"]*1e3, d["roofline"]["frac"],
      d["roofline_backward"]["avg_launch_ms"]*1e3, d["roofline_backward"]["frac"]))
for k,v in (d.get("secondary") or {}).items():
    print(" ", k, v.get("value") and round(v["value"]/1e6,3), v.get("ms_per_step"), v.get("error"))
PY

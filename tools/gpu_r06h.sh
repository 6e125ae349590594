#!/bin/bash
# round 6: two queues, minibatches prepared further ahead on ONE preparation stream; process-phase clock of build/difacto
cd "$(dirname "$0")/.." && R=$PWD && O=$R/gpurun_out/r06h && mkdir -p $O
export TMPDIR=/tmp
python -c "from difacto_amd.build import build_hip, build_host; build_hip(); build_host()" 2>&1 | tail -2
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
    print(sys.argv[2], "FAILED", e)
PY
}
run tq_ahead1 --two-queues
run tq_ahead2 --two-queues --ahead 2
run tq_ahead3 --two-queues --ahead 3
run tq_ahead4 --two-queues --ahead 4
run tq_ahead2_notiming --two-queues --ahead 2 --no-timing
run tq_ahead1_notiming --two-queues --no-timing
run tq_ahead1_again --two-queues
DIFACTO_PROFILE=1 E2E_FORMATS=rec timeout 900 python tools/e2e_cli.py 400000 48 > $O/e2e.jsonl 2> $O/e2e.err
grep "process:" $O/e2e.err | tail -6
python - $O/e2e.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print(d["format"], d["exe"], "whole loop M rows/s %.1f steady %.1f loop_s small/big %.4f %.4f process wall small/big %.3f %.3f -> process M rows/s %.1f" % (
        d.get("loop_rows_per_s_big",0)/1e6, d.get("steady_rows_per_s_by_loop_clock",0)/1e6, d.get("loop_s",0), d.get("loop_s_big",0), d["wall_s"], d["wall_s_big"], d["rows_big"]/d["wall_s_big"]/1e6))
PY

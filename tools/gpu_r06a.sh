#!/bin/bash
# round 6, first GPU call: the single-queue step — parity tests, then same-box A/B of the two-queue step against rider policies
cd "$(dirname "$0")/.." && R=$PWD && O=$R/gpurun_out/r06a && mkdir -p $O
export TMPDIR=/tmp
python -c "from difacto_amd.build import build_hip; build_hip()" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_single_queue.py -x -q 2>&1 | tail -15 > $O/pytest_single_queue.txt
cat $O/pytest_single_queue.txt
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
run two_queues --two-queues
run sq_default --single-queue
run sq_ahead1 --single-queue --ahead 1
run sq_ahead3 --single-queue --ahead 3
run sq_allU_ahead4 --single-queue --ahead 4 --ctx-option rider_slot_count=2 --ctx-option rider_slot_scatter=2 --ctx-option rider_slot_sort=2 --ctx-option rider_slot_emit=2
run sq_sortU_ahead3 --single-queue --ahead 3 --ctx-option rider_slot_count=2 --ctx-option rider_slot_scatter=0 --ctx-option rider_slot_sort=2 --ctx-option rider_slot_emit=2
run sq_first --single-queue --ctx-option rider_period_lookup=1 --ctx-option rider_period_forward=1 --ctx-option rider_period_update=1
run sq_per2 --single-queue --ctx-option rider_period_lookup=2 --ctx-option rider_period_forward=2 --ctx-option rider_period_update=2
run sq_per8 --single-queue --ctx-option rider_period_lookup=8 --ctx-option rider_period_forward=8 --ctx-option rider_period_update=8
# partial variants: count + scatter riding only (sort, emit as launches of their own); sort + emit riding only
run sq_only_count_scatter --single-queue --ctx-option rider_slot_sort=5 --ctx-option rider_slot_emit=6
run sq_only_sort_emit --single-queue --ctx-option rider_slot_count=6 --ctx-option rider_slot_scatter=4
run sq_all_alone --single-queue --ctx-option rider_slot_count=6 --ctx-option rider_slot_scatter=4 --ctx-option rider_slot_sort=5 --ctx-option rider_slot_emit=6
run serial --no-pipeline
run two_queues_again --two-queues
run sq_default_again --single-queue

#!/bin/bash
# round 6, third GPU call: the single-queue step where launches are latency, not throughput — the C2 shape (batch 100, V_dim 8)
cd "$(dirname "$0")/.." && R=$PWD && O=$R/gpurun_out/r06c && mkdir -p $O
export TMPDIR=/tmp
python -c "from difacto_amd.build import build_hip; build_hip()" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_single_queue.py -x -q 2>&1 | tail -15 > $O/pytest_single_queue.txt
cat $O/pytest_single_queue.txt
B="python bench.py --no-secondary --cpu-batches 0 --min-time 1 --preset c2"
run() { name=$1; shift; timeout 300 $B "$@" > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d.get("kernel_ms_per_step",{})
    print("%-28s %8.3f M ex/s  %.4f ms  enqueue %.4f ms | bk %s" % (sys.argv[2], d["value"]/1e6, d["ms_per_step"], d["host_enqueue_ms_per_step"],
          {a:round(b*1e3,1) for a,b in k.items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run c2_two_queues --two-queues
run c2_sq --single-queue
run c2_sq_first --single-queue --ctx-option rider_period_lookup=1 --ctx-option rider_period_forward=1 --ctx-option rider_period_update=1
run c2_sq_ahead1 --single-queue --ahead 1
run c2_sq_ahead3 --single-queue --ahead 3
run c2_sq_notiming --single-queue --no-timing
run c2_two_queues_notiming --two-queues --no-timing
run c2_serial --no-pipeline
run c2_sq_rows1000 --single-queue --rows 1000
run c2_tq_rows1000 --two-queues --rows 1000
B="python bench.py --no-secondary --cpu-batches 0 --min-time 1"
run c3_rows2000_tq --two-queues --rows 2000
run c3_rows2000_sq --single-queue --rows 2000
run c3_rows2000_sq_first --single-queue --rows 2000 --ctx-option rider_period_lookup=1 --ctx-option rider_period_forward=1 --ctx-option rider_period_update=1

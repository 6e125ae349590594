#!/bin/bash
# round 6, fourth GPU call: riders placed in the TAIL of their carrier launch (rider_start_*: per cent of the carrier's blocks first)
cd "$(dirname "$0")/.." && R=$PWD && O=$R/gpurun_out/r06d && mkdir -p $O
export TMPDIR=/tmp
python -c "from difacto_amd.build import build_hip; build_hip()" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_single_queue.py -x -q 2>&1 | tail -5 > $O/pytest_single_queue.txt
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
ALLU="--ctx-option rider_slot_count=2 --ctx-option rider_slot_scatter=2 --ctx-option rider_slot_sort=2 --ctx-option rider_slot_emit=2"
P1="--ctx-option rider_period_lookup=1 --ctx-option rider_period_forward=1 --ctx-option rider_period_update=1"
run two_queues --two-queues
for st in 50 80 100; do
  run allU4_start${st}_p1 --single-queue --ahead 4 $ALLU $P1 --ctx-option rider_start_update=$st
  run allU4_start${st}_p2 --single-queue --ahead 4 $ALLU --ctx-option rider_period_update=2 --ctx-option rider_start_update=$st
done
for st in 50 100; do
  run def_start${st}_p1 --single-queue $P1 --ctx-option rider_start_lookup=$st --ctx-option rider_start_forward=$st --ctx-option rider_start_update=$st
done
run sortU3_start80_p1 --single-queue --ahead 3 $P1 --ctx-option rider_slot_count=2 --ctx-option rider_slot_scatter=0 --ctx-option rider_slot_sort=2 --ctx-option rider_slot_emit=2 --ctx-option rider_start_update=80
run sortU3_start100_p1 --single-queue --ahead 3 $P1 --ctx-option rider_slot_count=2 --ctx-option rider_slot_scatter=0 --ctx-option rider_slot_sort=2 --ctx-option rider_slot_emit=2 --ctx-option rider_start_update=100

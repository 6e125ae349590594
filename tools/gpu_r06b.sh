#!/bin/bash
# round 6, second GPU call: does hipExtAnyOrderLaunch overlap two dispatches of one stream on gfx950?  then the single-queue step with
# the Localizer's stages as any-order dispatches beside their carrier launches
cd "$(dirname "$0")/.." && R=$PWD && O=$R/gpurun_out/r06b && mkdir -p $O
export TMPDIR=/tmp
python -c "from difacto_amd.build import build_hip; build_hip()" 2>&1 | tail -2
timeout 120 tools/anyorder_bench.bin > $O/anyorder_bench.txt 2>&1; cat $O/anyorder_bench.txt
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
run ao_after --single-queue --ctx-option rider_mode=1
run ao_before --single-queue --ctx-option rider_mode=2
run ao_after_allU4 --single-queue --ahead 4 --ctx-option rider_mode=1 --ctx-option rider_slot_count=2 --ctx-option rider_slot_scatter=2 --ctx-option rider_slot_sort=2 --ctx-option rider_slot_emit=2
run ao_after_sortU3 --single-queue --ahead 3 --ctx-option rider_mode=1 --ctx-option rider_slot_count=2 --ctx-option rider_slot_scatter=0 --ctx-option rider_slot_sort=2 --ctx-option rider_slot_emit=2
run ao_before_sortU3 --single-queue --ahead 3 --ctx-option rider_mode=2 --ctx-option rider_slot_count=2 --ctx-option rider_slot_scatter=0 --ctx-option rider_slot_sort=2 --ctx-option rider_slot_emit=2
run ao_after_notiming --single-queue --ctx-option rider_mode=1 --no-timing
run two_queues_notiming --two-queues --no-timing
